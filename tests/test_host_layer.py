"""CPU: the C++ drop-in classes (msc-hpc-final-project_amd/host), driven through host_capi.cc."""
import ctypes
import glob
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_SO = os.path.join(ROOT, "msc-hpc-final-project_amd", "host", "libmschpc_host.so")
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))

_u32p = ctypes.POINTER(ctypes.c_uint32)
_f64p = ctypes.POINTER(ctypes.c_double)


@pytest.fixture(scope="module")
def host(pkg):
    pkg.lib()  # liblzx.so first (RTLD_GLOBAL), then the host library that links it
    H = ctypes.CDLL(HOST_SO)
    H.host_last_error.restype = ctypes.c_char_p
    H.host_symtridiag.argtypes = [ctypes.c_int, _f64p, _f64p, _f64p]
    H.host_expm_file.argtypes = [ctypes.c_char_p, ctypes.c_uint, ctypes.c_int, ctypes.c_int, _f64p, ctypes.c_uint, _f64p, _f64p]
    H.host_expm_file.restype = ctypes.c_long
    H.host_load_csr.argtypes = [ctypes.c_char_p, _u32p, _u32p, ctypes.c_uint]
    H.host_load_csr.restype = ctypes.c_long
    H.host_adaptive_file.argtypes = [ctypes.c_char_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_double, ctypes.c_int, _f64p,
                                     ctypes.c_uint, _u32p, _f64p, ctypes.c_uint, _u32p]
    H.host_adaptive_file.restype = ctypes.c_long
    H.host_gen_csr.argtypes = [ctypes.c_char, ctypes.c_uint, ctypes.c_uint, ctypes.c_ulonglong, ctypes.c_ulonglong, _u32p, _u32p, ctypes.c_uint]
    H.host_gen_csr.restype = ctypes.c_long
    H.host_load_path.argtypes = [ctypes.c_char_p, _u32p, _u32p, ctypes.c_uint, _u32p]
    H.host_load_path.restype = ctypes.c_long
    H.host_expm_options_file.argtypes = [ctypes.c_char_p, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_uint,
                                         ctypes.c_double, ctypes.c_int, ctypes.c_int, _f64p, ctypes.c_uint, _f64p, _f64p, _u32p,
                                         _f64p, ctypes.c_uint]
    H.host_expm_options_file.restype = ctypes.c_long
    return H


def p(a, t):
    return a.ctypes.data_as(t)


def write_pairs(path, n, pairs):
    with open(path, "w") as f:
        f.write(f"{n} {n} {len(pairs)}\n")
        np.savetxt(f, pairs, fmt="%d")


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(q)[:-4] for q in GOLDEN])
def test_loader_and_cpu_pipeline_match_reference_fixture(host, tmp_path, path):
    g = np.load(path)
    n, k = int(g["mtx_n"]), int(g["k"])
    mtx = str(tmp_path / "g.mtx")
    write_pairs(mtx, n, g["mtx_pairs"])
    ro = np.zeros(n + 1, dtype=np.uint32)
    ci = np.zeros(2 * len(g["mtx_pairs"]) + 1, dtype=np.uint32)
    edges = host.host_load_csr(mtx.encode(), p(ro, _u32p), p(ci, _u32p), len(ci))
    assert edges == int(g["ref_edge_count"]), host.host_last_error()
    assert np.array_equal(ro, g["ref_row_offset"]) and np.array_equal(ci[:2 * edges], g["ref_col_idx"])
    ans = np.zeros(n)
    alpha, beta = np.zeros(k), np.zeros(k)
    rc = host.host_expm_file(mtx.encode(), k, 0, 0, p(ans, _f64p), n, p(alpha, _f64p), p(beta, _f64p))
    assert rc == n, host.host_last_error()
    # CPU decompose(): same operation order as the reference -> identical coefficients
    assert np.array_equal(alpha, g["alpha"]) and np.array_equal(beta[:k - 1], g["beta"])
    # own QL eigensolver + plain-loop multOut vs LAPACK + BLAS: rounding level
    assert np.abs(ans - g["ans"]).max() <= 1e-12 * np.abs(g["ans"]).max()


def test_symtridiag_matches_lapack(host):
    from scipy.linalg import eigh_tridiagonal
    rng = np.random.default_rng(3)
    for k in (1, 2, 7, 50, 120):
        d = rng.normal(size=k) * 10
        e = rng.random(max(k - 1, 1)) * 5
        dd = d.copy()
        z = np.zeros((k, k))
        assert host.host_symtridiag(k, p(dd, _f64p), p(e, _f64p), p(z, _f64p)) == 0
        if k == 1:
            assert dd[0] == d[0] and z[0, 0] == 1.0
            continue
        lam, V = eigh_tridiagonal(d, e[:k - 1], lapack_driver="stev")
        assert np.abs(dd - lam).max() <= 1e-12 * max(1.0, np.abs(lam).max())
        T = np.diag(d) + np.diag(e[:k - 1], 1) + np.diag(e[:k - 1], -1)
        assert np.abs(T @ z - z * dd).max() <= 1e-11 * np.abs(lam).max()
        assert np.abs(z.T @ z - np.eye(k)).max() <= 1e-12


def test_generators_match_oracle_spec(host, oracle):
    O = oracle
    ro_ref, ci_ref = O.gen_rmat(12, 3000, 40000, 77)
    ro = np.zeros(3001, dtype=np.uint32)
    ci = np.zeros(80001, dtype=np.uint32)
    e = host.host_gen_csr(b"m", 12, 3000, 40000, 77, p(ro, _u32p), p(ci, _u32p), len(ci))
    assert e == len(ci_ref) // 2
    assert np.array_equal(ro, ro_ref.astype(np.uint32)) and np.array_equal(ci[:2 * e], ci_ref)
    ro_ref, ci_ref = O.gen_er(2000, 9000, 1234)
    ro = np.zeros(2001, dtype=np.uint32)
    ci = np.zeros(18001, dtype=np.uint32)
    e = host.host_gen_csr(b"r", 0, 2000, 9000, 1234, p(ro, _u32p), p(ci, _u32p), len(ci))
    assert e == len(ci_ref) // 2 and np.array_equal(ci[:2 * e], ci_ref)
    ro = np.zeros(501, dtype=np.uint32)
    ci = np.zeros(6001, dtype=np.uint32)
    e = host.host_gen_csr(b"b", 0, 500, 5, 1234, p(ro, _u32p), p(ci, _u32p), len(ci))
    deg = np.diff(ro.astype(np.int64))
    assert e > 0 and deg.min() >= 5 and deg.sum() == 2 * e     # every vertex attaches to >= m others


def test_missing_file_reports_error(host):
    ans = np.zeros(4)
    assert host.host_expm_file(b"/nonexistent/graph.mtx", 3, 0, 0, p(ans, _f64p), 4, None, None) == -1
    assert b"cannot open" in host.host_last_error()


def test_convergence_monitor_cpu(host, tmp_path):
    """multOutAdaptive: the change between successive Krylov dimensions falls monotonically to the tolerance and the
    answer at the stopping dimension matches the fixture (reference's open problem, writeup section 11)."""
    g = np.load(GOLDEN[1])
    n = int(g["mtx_n"])
    mtx = str(tmp_path / "g.mtx")
    write_pairs(mtx, n, g["mtx_pairs"])
    ans = np.zeros(n)
    ks = np.zeros(16, dtype=np.uint32)
    ch = np.zeros(16)
    used = ctypes.c_uint()
    m = host.host_adaptive_file(mtx.encode(), 40, 5, 1e-12, 0, p(ans, _f64p), n, p(ks, _u32p), p(ch, _f64p), 16,
                                ctypes.cast(ctypes.byref(used), _u32p))
    assert m > 2, host.host_last_error()
    assert list(ks[:m]) == [5 * (i + 1) for i in range(m)]
    assert ch[0] == 1.0 and np.all(np.diff(ch[1:m]) < 0) and ch[m - 1] <= 1e-12
    assert used.value == ks[m - 1] and used.value < 40          # stopped early
    assert np.abs(ans - g["expm_ref"]).max() <= 1e-10 * np.abs(g["expm_ref"]).max()


def _load_path(host, path, n, cap):
    ro = np.zeros(n + 1, dtype=np.uint32)
    ci = np.zeros(cap, dtype=np.uint32)
    info = np.zeros(4, dtype=np.uint32)
    edges = host.host_load_path(path.encode(), p(ro, _u32p), p(ci, _u32p), cap, p(info, _u32p))
    assert edges >= 0, host.host_last_error()
    return ro, ci[:info[3]], info


def test_parallel_parser_and_cache_match_the_sequential_loader(host, oracle, tmp_path, monkeypatch):
    """N1: the text is parsed by several threads and a binary side-car caches the CSR; both must give exactly the CSR
    of the one-thread loader (which the golden fixtures pin to the reference's std::set build)."""
    O = oracle
    n = 60000
    rp, ci = O.gen_rmat(16, n, 400000, 99)
    path = str(tmp_path / "big.mtx")
    O.write_mtx(path, n, rp, ci)
    assert os.path.getsize(path) > (1 << 20)          # large enough for the parser to split it
    monkeypatch.setenv("LZX_NO_CSR_CACHE", "1")
    monkeypatch.setenv("LZX_PARSE_THREADS", "1")
    ro1, ci1, info1 = _load_path(host, path, n, len(ci) + 8)
    assert info1[0] == 0 and info1[2] == 1
    assert np.array_equal(ro1, rp.astype(np.uint32)) and np.array_equal(ci1, ci)
    monkeypatch.setenv("LZX_PARSE_THREADS", "7")
    ro7, ci7, info7 = _load_path(host, path, n, len(ci) + 8)
    assert info7[2] == 7 and np.array_equal(ro7, ro1) and np.array_equal(ci7, ci1)
    assert not os.path.exists(path + ".lzxcsr")
    monkeypatch.delenv("LZX_NO_CSR_CACHE")
    _, _, a = _load_path(host, path, n, len(ci) + 8)   # parses and writes the side-car
    assert a[0] == 0 and os.path.exists(path + ".lzxcsr")
    roc, cic, b = _load_path(host, path, n, len(ci) + 8)
    assert b[0] == 1 and np.array_equal(roc, ro1) and np.array_equal(cic, ci1)
    # a changed text file invalidates the side-car (size + mtime are part of its header)
    with open(path, "a") as f:
        f.write("\n")
    _, _, c = _load_path(host, path, n, len(ci) + 8)
    assert c[0] == 0


def test_parser_token_stream_semantics(host, oracle, tmp_path, monkeypatch):
    """The reference reads `f >> col >> row` as ONE token stream (parallel-final/lib/adjMatrix.cc:29-31): line structure
    means nothing.  A large file with three numbers per line makes pairs straddle the parser's cuts: the pieces then
    pair differently from the stream and the loader must notice and read it as one piece.  An out-of-range
    pair directly behind the declared edge count is not an error (it is never read), one inside it is."""
    O = oracle
    n = 60000
    rp, ci = O.gen_rmat(16, n, 400000, 99)
    path = str(tmp_path / "ref.mtx")
    O.write_mtx(path, n, rp, ci)
    tok = open(path).read().split()
    body = tok[3:]
    odd = str(tmp_path / "three_per_line.mtx")
    with open(odd, "w") as f:
        f.write(" ".join(tok[:3]) + "\n")
        for i in range(0, len(body), 3):
            f.write(" ".join(body[i:i + 3]) + "\n")
    assert os.path.getsize(odd) > (1 << 20)
    monkeypatch.setenv("LZX_NO_CSR_CACHE", "1")
    monkeypatch.setenv("LZX_PARSE_THREADS", "7")
    ro, cj, info = _load_path(host, odd, n, len(ci) + 8)
    assert info[2] == 1                                 # fell back to one piece
    assert np.array_equal(ro, rp.astype(np.uint32)) and np.array_equal(cj, ci)
    # E good pairs, then a pair naming vertex n + 5: beyond the declared count, never read
    tail = str(tmp_path / "bad_tail.mtx")
    write_pairs(tail, 6, np.array([[2, 1], [5, 4], [6, 1]]))
    with open(tail, "a") as f:
        f.write("11 3\n")
    ro, cj, info = _load_path(host, tail, 6, 16)
    assert info[3] == 6
    inside = str(tmp_path / "bad_inside.mtx")
    write_pairs(inside, 6, np.array([[2, 1], [11, 3], [6, 1]]))
    ro2 = np.zeros(7, dtype=np.uint32)
    ci2 = np.zeros(16, dtype=np.uint32)
    assert host.host_load_path(inside.encode(), p(ro2, _u32p), p(ci2, _u32p), 16, None) < 0
    assert b"out of range" in host.host_last_error()
    # a side-car whose columns are not vertices is not trusted: the text is parsed again
    monkeypatch.delenv("LZX_NO_CSR_CACHE")
    good = str(tmp_path / "cached.mtx")
    write_pairs(good, 6, np.array([[2, 1], [5, 4], [6, 1]]))
    ro_a, ci_a, ia = _load_path(host, good, 6, 16)
    assert ia[0] == 0 and os.path.exists(good + ".lzxcsr")
    raw = bytearray(open(good + ".lzxcsr", "rb").read())
    raw[-4:] = (77).to_bytes(4, "little")               # last column index := 77
    st = os.stat(good)
    open(good + ".lzxcsr", "wb").write(raw)
    os.utime(good, ns=(st.st_atime_ns, st.st_mtime_ns))
    ro_b, ci_b, ib = _load_path(host, good, 6, 16)
    assert ib[0] == 0 and np.array_equal(ro_b, ro_a) and np.array_equal(ci_b, ci_a)


def test_self_loop_line_is_one_diagonal_entry(host, oracle, tmp_path, monkeypatch):
    """A `r r` line: the reference's std::set holds the key once, so the row gets ONE diagonal entry and
    row_offset[n] is odd (2 * edge_count would be one short / one over: ADVICE r1)."""
    monkeypatch.setenv("LZX_NO_CSR_CACHE", "1")
    path = str(tmp_path / "loop.mtx")
    write_pairs(path, 6, np.array([[2, 1], [3, 3], [5, 4], [6, 1]]))
    ro, ci, info = _load_path(host, path, 6, 16)
    assert info[3] == 7 and ro[-1] == 7
    assert list(ci[ro[2]:ro[3]]) == [2]
    n, e, ro_ref, ci_ref = oracle.load_mtx(path)
    assert np.array_equal(ro, ro_ref.astype(np.uint32)) and np.array_equal(ci, ci_ref)
    ans = np.zeros(6)
    alpha, beta = np.zeros(3), np.zeros(3)
    rc = host.host_expm_file(path.encode(), 3, 0, 0, p(ans, _f64p), 6, p(alpha, _f64p), p(beta, _f64p))
    assert rc == 6, host.host_last_error()
    assert np.isfinite(ans).all()


def test_arnoldi_and_reorthog_in_the_class_layer(host, oracle, tmp_path):
    """lanczosDecomp(A, k, x, cuda = false, lanczosOptions{arnoldi_every}) is serial/lib/lanczos.h:44-57's fourth-argument
    constructor (decompose_with_arnoldi, serial/lib/lanczos.cc:58-132) with the hard-coded 2 as a parameter: identical
    coefficients to the oracle's restatement (same operation order), for the reference's 2 and for 1.  reorthog()
    (serial/lib/lanczos.cc:202-207) leaves the answer of a well-conditioned run where it was."""
    O = oracle
    rp, ci = O.gen_rmat(12, 3000, 40000, 3)
    n, k = 3000, 16
    mtx = str(tmp_path / "g.mtx")
    O.write_mtx(mtx, n, rp, ci)
    _, _, rp, ci = O.load_mtx(mtx)
    x0 = np.ones(n)
    for every in (1, 2):
        a_ref, b_ref, Q_ref, xn = O.lanczos_arnoldi(rp, ci, k, x0, every=every)
        ans, alpha, beta = np.zeros(n), np.zeros(k), np.zeros(k)
        info = np.zeros(4, dtype=np.uint32)
        rc = host.host_expm_options_file(mtx.encode(), k, 0, 0, every, 0, 0.0, 0, 0, p(ans, _f64p), n, p(alpha, _f64p), p(beta, _f64p),
                                         p(info, _u32p), None, 0)
        assert rc == n, host.host_last_error()
        assert np.array_equal(alpha, a_ref) and np.array_equal(beta[:k - 1], b_ref), every
        assert info[0] == k and info[1] == k
        lam, V = O.eigen(a_ref, b_ref)
        want = O.mult_out(np.ascontiguousarray(Q_ref.T), V, lam, xn)
        assert np.abs(ans - want).max() <= 1e-12 * np.abs(want).max()
    # reorthog() on the plain decomposition: Q was orthonormal to rounding already
    plain, re = np.zeros(n), np.zeros(n)
    assert host.host_expm_options_file(mtx.encode(), 10, 0, 0, 0, 0, 0.0, 0, 0, p(plain, _f64p), n, None, None, None, None, 0) == n
    assert host.host_expm_options_file(mtx.encode(), 10, 0, 0, 0, 0, 0.0, 0, 1, p(re, _f64p), n, None, None, None, None, 0) == n
    assert np.abs(re - plain).max() <= 1e-9 * np.abs(plain).max() and np.abs(re - plain).max() > 0

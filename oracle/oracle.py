"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Python face of the CPU oracle: ctypes bindings to ``liboracle.so`` (our plain-C
restatement of the reference's serial/ hot path, see lanczos_oracle.c) plus the
small host steps the reference runs after the loop, restated in numpy.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product path (liblzx.so and everything under
``msc-hpc-final-project_amd/``) never does.

Reference lines restated here (paths relative to /root/reference):
  * load_mtx        serial/main.cc:33-41 + serial/lib/adjMatrix.cc:21-54
  * write_mtx       parallel-final/lib/adjMatrix.cc:53-69 (``n n E`` + ``col row``)
  * eigen           serial/lib/eigen.cc:12-15 (LAPACKE_dstevd; here LAPACK dstev
                    through scipy -- same symmetric-tridiagonal problem, the
                    result is invariant to eigenvector sign/order)
  * mult_out        serial/lib/multiplyOut.cc:17-37
  * eigen_dstevd, mult_out_blas: the reference's own LAPACKE / CBLAS CALLS executed from SciPy's bundled OpenBLAS
                    (eigen.cc:13; multiplyOut.cc:30,33) -- a pin for the two restatements above
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

_u64p = ctypes.POINTER(ctypes.c_uint64)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_f64p = ctypes.POINTER(ctypes.c_double)
SPMV_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, _f64p, _f64p)


def _p(a, ty):
    return a.ctypes.data_as(ty)


def build(ref: bool = False) -> None:
    """Compile liboracle.so (and, where /root/reference exists, oracle/_ref)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if ref and os.path.isdir("/root/reference/serial/lib"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        L.orc_spmv.argtypes = [ctypes.c_uint64, _u64p, _u32p, _f64p, _f64p]
        L.orc_spmv.restype = None
        L.orc_norm.argtypes = [ctypes.c_uint64, _f64p]
        L.orc_norm.restype = ctypes.c_double
        L.orc_inner_prod.argtypes = [ctypes.c_uint64, _f64p, _f64p]
        L.orc_inner_prod.restype = ctypes.c_double
        L.orc_lanczos.argtypes = [ctypes.c_uint64, _u64p, _u32p, ctypes.c_uint32, _f64p,
                                  _f64p, _f64p, _f64p, ctypes.c_int, _f64p,
                                  ctypes.c_void_p, ctypes.c_void_p]
        L.orc_lanczos.restype = ctypes.c_int
        L.orc_lanczos_arnoldi.argtypes = [ctypes.c_uint64, _u64p, _u32p, ctypes.c_uint32, ctypes.c_uint32, _f64p,
                                          _f64p, _f64p, _f64p, _f64p]
        L.orc_lanczos_arnoldi.restype = ctypes.c_int
        L.orc_csr_from_keys.argtypes = [ctypes.c_uint64, ctypes.c_uint64, _u64p, _u64p, _u32p]
        L.orc_csr_from_keys.restype = None
        L.orc_gen_er_keys.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, _u64p]
        L.orc_gen_er_keys.restype = ctypes.c_uint64
        L.orc_gen_rmat_keys.argtypes = [ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64,
                                        ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32,
                                        ctypes.c_uint32, _u64p]
        L.orc_gen_rmat_keys.restype = ctypes.c_uint64
        _LIB = L
    return _LIB


# --------------------------------------------------------------------------- graphs
def csr_from_keys(n: int, keys: np.ndarray):
    """Sorted-unique directed keys ((row<<32)|col) -> (row_offset u64[n+1], col_idx u32[nnz])."""
    keys = np.unique(np.ascontiguousarray(keys, dtype=np.uint64))  # std::set<Edge> order
    row_offset = np.empty(n + 1, dtype=np.uint64)
    col_idx = np.empty(max(len(keys), 1), dtype=np.uint32)
    lib().orc_csr_from_keys(n, len(keys), _p(keys, _u64p), _p(row_offset, _u64p), _p(col_idx, _u32p))
    return row_offset, col_idx[:len(keys)]


def rmat_thresholds(a=0.57, b=0.19, c=0.19):
    ta = int(round(a * 65536))
    tab = int(round((a + b) * 65536))
    tabc = int(round((a + b + c) * 65536))
    return ta, tab, tabc


def gen_er(n: int, draws: int, seed: int):
    keys = np.empty(2 * draws, dtype=np.uint64)
    m = lib().orc_gen_er_keys(n, draws, seed, _p(keys, _u64p))
    return csr_from_keys(n, keys[:m])


def gen_rmat(scale: int, n: int, draws: int, seed: int, a=0.57, b=0.19, c=0.19):
    assert n <= (1 << scale) and scale <= 32
    keys = np.empty(2 * draws, dtype=np.uint64)
    ta, tab, tabc = rmat_thresholds(a, b, c)
    m = lib().orc_gen_rmat_keys(scale, n, draws, seed, ta, tab, tabc, _p(keys, _u64p))
    return csr_from_keys(n, keys[:m])


def write_mtx(path: str, n: int, row_offset: np.ndarray, col_idx: np.ndarray) -> int:
    """Reference text format: ``n n E`` then one 1-indexed ``col row`` line per undirected edge
    (col > row), as parallel-final's write_matrix_to_file emits. Returns E."""
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(row_offset.astype(np.int64)))
    cols = col_idx.astype(np.int64)
    up = cols > rows
    r, c = rows[up] + 1, cols[up] + 1
    with open(path, "w") as f:
        f.write(f"{n} {n} {len(r)}\n")
        np.savetxt(f, np.stack([c, r], axis=1), fmt="%d")
    return int(len(r))


def load_mtx(path: str):
    """serial/main.cc:33-41 + populate_sparse_matrix: returns (n, edge_count, row_offset, col_idx)."""
    with open(path) as f:
        tok = np.array(f.read().split(), dtype=np.int64)
    n, declared = int(tok[0]), int(tok[2])
    pairs = tok[3:3 + 2 * declared].reshape(-1, 2)
    col = (pairs[:, 0] - 1).astype(np.uint64)
    row = (pairs[:, 1] - 1).astype(np.uint64)
    keys = np.concatenate([(row << np.uint64(32)) | col, (col << np.uint64(32)) | row])
    ro, ci = csr_from_keys(n, keys)
    return n, len(ci) // 2, ro, ci


# --------------------------------------------------------------------------- hot path
def spmv(row_offset, col_idx, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty(len(row_offset) - 1, dtype=np.float64)
    lib().orc_spmv(len(y), _p(row_offset, _u64p), _p(col_idx, _u32p), _p(x, _f64p), _p(y, _f64p))
    return y


def lanczos(row_offset, col_idx, k: int, x, want_q: bool = True, q_colmajor: bool = False,
            ext_spmv=None):
    """Returns (alpha[k], beta[k-1], Q or None, x_norm). Q is (n,k) row-major as serial/ keeps it,
    or (k,n) when q_colmajor (the layout parallel-final's device path produces)."""
    n = len(row_offset) - 1
    x = np.ascontiguousarray(x, dtype=np.float64)
    alpha = np.zeros(k)
    beta = np.zeros(max(k - 1, 1))
    Q = np.zeros((k, n) if q_colmajor else (n, k)) if want_q else None
    xn = ctypes.c_double(0.0)
    cb = ctypes.cast(ext_spmv, ctypes.c_void_p) if ext_spmv is not None else None
    rc = lib().orc_lanczos(n, _p(row_offset, _u64p), _p(col_idx, _u32p), k, _p(x, _f64p),
                           _p(alpha, _f64p), _p(beta, _f64p),
                           _p(Q, _f64p) if want_q else None, int(q_colmajor),
                           ctypes.byref(xn), cb, None)
    if rc != 0:
        raise MemoryError("orc_lanczos")
    return alpha, beta[:k - 1], Q, xn.value


_OMP = None


def lanczos_omp(row_offset, col_idx, k: int, x, threads: int = 0):
    """The all-core companion of `lanczos` (lanczos_oracle_omp.c: the same loop, OpenMP over rows and elements; inner products
    are OpenMP reductions, so alpha / beta carry another rounding).  A TIMING baseline (bench.py, kind "port-omp"), not a parity
    oracle.  Returns (alpha[k], beta[k-1], x_norm, threads used).  `threads` > 0 sets OMP_NUM_THREADS before the library is
    first loaded (the OpenMP runtime reads it once)."""
    global _OMP
    if _OMP is None:
        if threads > 0:
            os.environ["OMP_NUM_THREADS"] = str(threads)
        so = os.path.join(_HERE, "liboracle_omp.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        L.orc_lanczos_omp.argtypes = [ctypes.c_uint64, _u64p, _u32p, ctypes.c_uint32, _f64p, _f64p, _f64p, _f64p]
        L.orc_lanczos_omp.restype = ctypes.c_int
        L.orc_omp_threads.restype = ctypes.c_int
        L.orc_omp_set_threads.argtypes = [ctypes.c_int]
        L.orc_omp_set_threads.restype = None
        _OMP = L
    if threads > 0:
        _OMP.orc_omp_set_threads(int(threads))
    n = len(row_offset) - 1
    x = np.ascontiguousarray(x, dtype=np.float64)
    alpha = np.zeros(k)
    beta = np.zeros(max(k - 1, 1))
    xn = ctypes.c_double(0.0)
    rc = _OMP.orc_lanczos_omp(n, _p(row_offset, _u64p), _p(col_idx, _u32p), k, _p(x, _f64p), _p(alpha, _f64p), _p(beta, _f64p),
                              ctypes.byref(xn))
    if rc != 0:
        raise MemoryError("orc_lanczos_omp")
    return alpha, beta[:k - 1], xn.value, int(_OMP.orc_omp_threads())


def lanczos_arnoldi(row_offset, col_idx, k: int, x, every: int = 2):
    """serial/lib/lanczos.cc:58-132 (decompose_with_arnoldi; the reference's every = 2).  Returns (alpha[k],
    beta[k-1], Q (k, n), x_norm)."""
    n = len(row_offset) - 1
    x = np.ascontiguousarray(x, dtype=np.float64)
    alpha = np.zeros(k)
    beta = np.zeros(max(k - 1, 1))
    Q = np.zeros((k, n))
    xn = ctypes.c_double(0.0)
    rc = lib().orc_lanczos_arnoldi(n, _p(row_offset, _u64p), _p(col_idx, _u32p), k, every, _p(x, _f64p),
                                   _p(alpha, _f64p), _p(beta, _f64p), _p(Q, _f64p), ctypes.byref(xn))
    if rc != 0:
        raise MemoryError("orc_lanczos_arnoldi")
    return alpha, beta[:k - 1], Q, xn.value


def eigen(alpha, beta):
    """T = V diag(lam) V^T. V[i, j] = i-th component of eigenvector j (LAPACK_ROW_MAJOR 'V')."""
    from scipy.linalg import eigh_tridiagonal
    if len(alpha) == 1:
        return np.array(alpha, dtype=np.float64), np.ones((1, 1))
    lam, V = eigh_tridiagonal(np.asarray(alpha), np.asarray(beta), lapack_driver="stev")
    return lam, V


def mult_out(Q_rowmajor, V, lam, x_norm):
    """serial/lib/multiplyOut.cc:17-37: f = exp(lam) * x_norm * V[0, :]; ans = (Q V) f."""
    f = np.exp(lam)
    f = f * (x_norm * V[0, :])
    return (Q_rowmajor @ V) @ f


def expm_action(row_offset, col_idx, k: int, x):
    """The whole serial/main.cc pipeline (79-88): returns the centrality vector e^A x."""
    alpha, beta, Q, xn = lanczos(row_offset, col_idx, k, x)
    lam, V = eigen(alpha, beta)
    return mult_out(Q, V, lam, xn)


# --------------------------------------------------------------------------- the reference's own LIBRARY CALLS
# serial/lib/eigen.cc:13 is one call, LAPACKE_dstevd(LAPACK_ROW_MAJOR, 'V', k, eigenvalues, beta, eigenvectors, k); serial/lib/
# multiplyOut.cc:30,33 are two, cblas_dgemm (QV += Q V) and cblas_dgemv (ans += QV f).  lanczos.cc / eigen.cc / multiplyOut.cc cannot be
# COMPILED here (no lapacke.h / cblas.h in the image, and no stand-ins are written), but the routines they call exist in the OpenBLAS that
# SciPy bundles (exported with a scipy_ prefix): calling them with the reference's arguments executes the reference's post-loop
# arithmetic itself -- a pin for eigen() and mult_out() above that does not go through this file's restatement of them.
_BLAS = None


def blas_calls():
    """ctypes handle on SciPy's bundled OpenBLAS with the three routines the reference calls, or None where it is not found."""
    global _BLAS
    if _BLAS is None:
        import glob
        try:
            import scipy
            cands = glob.glob(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs", "libscipy_openblas*.so"))
            B = ctypes.CDLL(cands[0])
            B.scipy_LAPACKE_dstevd.argtypes = [ctypes.c_int, ctypes.c_char, ctypes.c_int, _f64p, _f64p, _f64p, ctypes.c_int]
            B.scipy_LAPACKE_dstevd.restype = ctypes.c_int
            B.scipy_cblas_dgemm.argtypes = [ctypes.c_int] * 3 + [ctypes.c_int] * 3 + [ctypes.c_double, _f64p, ctypes.c_int, _f64p, ctypes.c_int,
                                                                                      ctypes.c_double, _f64p, ctypes.c_int]
            B.scipy_cblas_dgemm.restype = None
            B.scipy_cblas_dgemv.argtypes = [ctypes.c_int] * 2 + [ctypes.c_int] * 2 + [ctypes.c_double, _f64p, ctypes.c_int, _f64p, ctypes.c_int,
                                                                                      ctypes.c_double, _f64p, ctypes.c_int]
            B.scipy_cblas_dgemv.restype = None
            _BLAS = B
        except (ImportError, IndexError, OSError, AttributeError):
            _BLAS = False
    return _BLAS or None


def eigen_dstevd(alpha, beta):
    """serial/lib/eigen.cc:13 executed: LAPACKE_dstevd(LAPACK_ROW_MAJOR = 101, 'V', k, d, e, z, k).  Returns (lam, V) as eigen() does."""
    B = blas_calls()
    assert B is not None, "SciPy's OpenBLAS not found"
    k = len(alpha)
    d = np.array(alpha, dtype=np.float64)
    e = np.zeros(max(k, 1))
    e[:k - 1] = np.asarray(beta, dtype=np.float64)[:k - 1]
    z = np.zeros((k, k))
    info = B.scipy_LAPACKE_dstevd(101, b"V", k, _p(d, _f64p), _p(e, _f64p), _p(z, _f64p), k)
    assert info == 0, info
    return d, z


def mult_out_blas(Q_rowmajor, V, lam, x_norm):
    """serial/lib/multiplyOut.cc:21-33 executed: exp, the elementwise product with ||x|| * (first row of the eigenvectors), then the
    reference's cblas_dgemm(RowMajor = 101, NoTrans = 111, NoTrans, n, k, k, 1, Q, k, V, k, 1, QV, k) and cblas_dgemv(RowMajor, NoTrans,
    n, k, 1, QV, k, f, 1, 1, ans, 1) -- QV and ans start as zeros (the reference accumulates with beta = 1 into fresh pages)."""
    B = blas_calls()
    assert B is not None, "SciPy's OpenBLAS not found"
    Q = np.ascontiguousarray(Q_rowmajor, dtype=np.float64)
    Vc = np.ascontiguousarray(V, dtype=np.float64)
    n, k = Q.shape
    f = np.exp(np.asarray(lam, dtype=np.float64))
    f = np.ascontiguousarray(f * (x_norm * Vc[0, :]))
    QV = np.zeros((n, k))
    ans = np.zeros(n)
    B.scipy_cblas_dgemm(101, 111, 111, n, k, k, 1.0, _p(Q, _f64p), k, _p(Vc, _f64p), k, 1.0, _p(QV, _f64p), k)
    B.scipy_cblas_dgemv(101, 111, n, k, 1.0, _p(QV, _f64p), k, _p(f, _f64p), 1, 1.0, _p(ans, _f64p), 1)
    return ans


# --------------------------------------------------------------------------- real reference
def ref():
    """oracle/_ref/libref_serial.so: the reference's own SPMV.cc + adjMatrix.cc, compiled here.
    Returns None where it has not been built (it cannot be built on the GPU box)."""
    global _REF
    if _REF is None:
        so = os.path.join(_HERE, "_ref", "libref_serial.so")
        if not os.path.exists(so):
            return None
        R = ctypes.CDLL(so)
        R.ref_load.argtypes = [ctypes.c_char_p]
        R.ref_load.restype = ctypes.c_void_p
        R.ref_info.argtypes = [ctypes.c_void_p, _u64p, _u64p, _u64p]
        R.ref_csr.argtypes = [ctypes.c_void_p, _u64p, _u64p]
        R.ref_spmv.argtypes = [ctypes.c_void_p, _f64p, _f64p]
        R.ref_free.argtypes = [ctypes.c_void_p]
        for f in (R.ref_info, R.ref_csr, R.ref_spmv, R.ref_free):
            f.restype = None
        _REF = R
    return _REF


class RefGraph:
    """A graph loaded by the reference's own adjMatrix file constructor."""

    def __init__(self, path: str):
        self.R = ref()
        assert self.R is not None, "oracle/_ref not built"
        self.h = self.R.ref_load(path.encode())
        assert self.h, f"reference loader could not open {path}"
        n, e, r0 = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        self.R.ref_info(self.h, ctypes.byref(n), ctypes.byref(e), ctypes.byref(r0))
        self.n, self.edge_count, self.row_offset0_as_loaded = n.value, e.value, r0.value

    def csr(self):
        ro = np.empty(self.n + 1, dtype=np.uint64)
        ci = np.empty(2 * self.edge_count + 1, dtype=np.uint64)
        self.R.ref_csr(self.h, _p(ro, _u64p), _p(ci, _u64p))
        return ro, ci[:int(ro[-1])]

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(self.n)
        self.R.ref_spmv(self.h, _p(x, _f64p), _p(y, _f64p))
        return y

    def spmv_callback(self):
        def cb(_user, pin, pout):
            self.R.ref_spmv(self.h, pin, pout)
        return SPMV_FN(cb)

    def close(self):
        if self.h:
            self.R.ref_free(self.h)
            self.h = None


# --------------------------------------------------------------------------- accuracy referee
_REFEREE = None


def referee_lib():
    """oracle/libreferee.so (referee.c): the same recurrence in x87 extended precision, optional full
    re-orthogonalisation.  Test infrastructure: more accurate than the oracle AND the engine, so k = 50 tolerances can
    be anchored to it."""
    global _REFEREE
    if _REFEREE is None:
        so = os.path.join(_HERE, "libreferee.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-s", "-C", _HERE, "referee"])
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # idle threads sleep instead of spinning on a shared CPU
        R = ctypes.CDLL(so)
        R.ref_expm_ld.argtypes = [ctypes.c_uint64, _u64p, _u32p, ctypes.c_uint32, _f64p, ctypes.c_int, ctypes.c_uint32,
                                  _f64p, _f64p, _f64p, _f64p, _f64p, _f64p]
        R.ref_expm_ld.restype = ctypes.c_int
        R.ref_spmv_ld.argtypes = [ctypes.c_uint64, _u64p, _u32p, _f64p, _f64p]
        R.ref_spmv_ld.restype = ctypes.c_int
        R.ref_set_threads.argtypes = [ctypes.c_int]
        R.ref_set_threads.restype = None
        # the job's CPU share, not the machine's thread count (a GPU box shows 256 hardware threads and grants ~16)
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        R.ref_set_threads(int(os.environ.get("LZX_REFEREE_THREADS", min(16, avail))))
        _REFEREE = R
    return _REFEREE


def referee_expm(row_offset, col_idx, k: int, x, caps=(0.0, 40.0), reorth: int = 0):
    """Extended-precision k-step Lanczos e^(s (A - theta_max)) x.  caps: one answer per entry -- cap > 0: s = min(1, cap /
    theta_max) (tests' shift_weights(cap)); 0: s = 1 (shift_weights(cap=None)); < 0: the unshifted e^A x.
    reorth: 0 none, 1 full (two MGS sweeps every iteration), 2 the reference's decompose_with_arnoldi schedule.
    Returns dict(ans=(len(caps), n) float64, alpha, beta, lam, orth_loss)."""
    n = len(row_offset) - 1
    x = np.ascontiguousarray(x, dtype=np.float64)
    caps = np.ascontiguousarray(caps, dtype=np.float64)
    ans = np.empty((len(caps), n))
    alpha, beta, lam = np.zeros(k), np.zeros(max(k - 1, 1)), np.zeros(k)
    loss = ctypes.c_double(0.0)
    rc = referee_lib().ref_expm_ld(n, _p(row_offset, _u64p), _p(col_idx, _u32p), k, _p(x, _f64p), reorth, len(caps),
                                   _p(caps, _f64p), _p(ans, _f64p), _p(alpha, _f64p), _p(beta, _f64p), _p(lam, _f64p),
                                   ctypes.byref(loss))
    if rc != 0:
        raise MemoryError("ref_expm_ld")
    return dict(ans=ans, alpha=alpha, beta=beta[:k - 1], lam=lam, orth_loss=loss.value)


def referee_spmv(row_offset, col_idx, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty(len(row_offset) - 1)
    if referee_lib().ref_spmv_ld(len(y), _p(row_offset, _u64p), _p(col_idx, _u32p), _p(x, _f64p), _p(y, _f64p)) != 0:
        raise MemoryError("ref_spmv_ld")
    return y

"""CPU: the oracle against (a) the committed golden vectors made from the reference's own compiled
SPMV.cc / adjMatrix.cc (tests/golden/make_golden.py), (b) that build itself where oracle/_ref exists,
(c) an analytic known-answer test in the style of the reference's serial/tests/numerical_test.cc."""
import glob
import os

import numpy as np
import pytest

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def write_pairs(path, n, pairs):
    with open(path, "w") as f:
        f.write(f"{n} {n} {len(pairs)}\n")
        np.savetxt(f, pairs, fmt="%d")


def test_golden_present():
    assert len(GOLDEN) >= 5


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_matches_reference_fixture(oracle, tmp_path, path):
    O = oracle
    g = np.load(path)
    n, k = int(g["mtx_n"]), int(g["k"])
    mtx = str(tmp_path / "g.mtx")
    write_pairs(mtx, n, g["mtx_pairs"])
    n2, edges, ro, ci = O.load_mtx(mtx)
    # loader: bit-exact with the reference's adjMatrix(N, E, ifstream&)
    assert n2 == n and edges == int(g["ref_edge_count"])
    assert np.array_equal(ro, g["ref_row_offset"].astype(np.uint64))
    assert np.array_equal(ci, g["ref_col_idx"])
    # spMV: bit-exact with the reference's spMV<double>
    assert np.array_equal(O.spmv(ro, ci, g["x"]), g["ref_spmv"])
    # Lanczos loop: same operation order over the same SpMV bits -> identical alpha / beta
    alpha, beta, Q, xn = O.lanczos(ro, ci, k, np.ones(n))
    assert np.array_equal(alpha, g["alpha"]) and np.array_equal(beta, g["beta"])
    lam, V = O.eigen(alpha, beta)
    ans = O.mult_out(Q, V, lam, xn)
    assert np.abs(ans - g["ans"]).max() <= 1e-13 * np.abs(g["ans"]).max()
    # and the whole pipeline against an independent e^A x
    assert np.abs(ans - g["expm_ref"]).max() <= 1e-11 * np.abs(g["expm_ref"]).max()


def test_oracle_matches_live_reference_build(oracle, tmp_path):
    O = oracle
    if O.ref() is None:
        pytest.skip("oracle/_ref not built here (needs /root/reference)")
    ro, ci = O.gen_er(10000, 100000, 1234)  # BASELINE C1
    n = 10000
    mtx = str(tmp_path / "c1.mtx")
    O.write_mtx(mtx, n, ro, ci)
    G = O.RefGraph(mtx)
    rro, rci = G.csr()
    assert np.array_equal(rro, ro) and np.array_equal(rci.astype(np.uint32), ci)
    x = np.random.default_rng(7).random(n)
    assert np.array_equal(G.spmv(x), O.spmv(ro, ci, x))
    a1, b1, Q1, _ = O.lanczos(ro, ci, 20, np.ones(n))
    a2, b2, Q2, _ = O.lanczos(ro, ci, 20, np.ones(n), ext_spmv=G.spmv_callback())
    assert np.array_equal(a1, a2) and np.array_equal(b1, b2) and np.array_equal(Q1, Q2)
    G.close()


def test_known_answer_eigen_expansion(oracle):
    """serial/tests/numerical_test.cc:74-116: x = sum c_i v_i over known eigenpairs, analytic
    e^A x = sum c_i e^{lambda_i} v_i.  The reference's fixtures (NotreDame_yeast eigenpairs) are not in the
    repository, so the eigenpairs come from a dense eigh of a small seeded graph; its recorded accuracy
    curve (2.1 at k=5 ... 3.5e-11 at k=20 ... 4e-15 at k=25, serial/output/numerical_test_output.txt) is
    what the shape below reproduces: monotone to rounding level."""
    O = oracle
    n = 400
    ro, ci = O.gen_er(n, 1600, 1234)
    A = np.zeros((n, n))
    rows = np.repeat(np.arange(n), np.diff(ro.astype(np.int64)))
    A[rows, ci.astype(np.int64)] = 1.0
    lam, Vec = np.linalg.eigh(A)
    c = np.random.default_rng(1234).random(100)
    top = Vec[:, -100:]
    x = top @ c
    exact = top @ (c * np.exp(lam[-100:]))
    errs = []
    for k in (5, 10, 20, 30):
        ans = O.expm_action(ro, ci, k, x)
        errs.append(np.linalg.norm(ans - exact) / np.linalg.norm(exact))
    assert errs[0] > errs[1] > errs[2]
    assert errs[2] < 1e-8 and errs[3] < 1e-12


def test_generators_edge_cases(oracle):
    O = oracle
    ro, ci = O.gen_er(50, 0, 1)                      # no edges at all
    assert len(ci) == 0 and np.all(ro == 0)
    ro, ci = O.gen_rmat(10, 700, 5000, 3)            # n not a power of two: endpoints re-drawn below n
    assert ci.max() < 700 and len(ro) == 701
    rows = np.repeat(np.arange(700), np.diff(ro.astype(np.int64)))
    assert not np.any(rows == ci)                     # no self loops
    keys = set(zip(rows.tolist(), ci.tolist()))
    assert all((c, r) in keys for r, c in keys)       # symmetric
    assert len(keys) == len(ci)                       # no duplicates
    for r in range(700):                              # ascending columns within a row
        seg = ci[int(ro[r]):int(ro[r + 1])]
        assert np.all(np.diff(seg.astype(np.int64)) > 0)


def test_referee_agrees_with_oracle_and_with_itself(oracle):
    """oracle/referee.c (x87 extended precision, test infrastructure): on small well-conditioned fixtures it must agree
    with the fp64 oracle to fp64 rounding, with and without full re-orthogonalisation, in all three functionals; its
    SpMV is the oracle's up to the last bit of a short positive sum."""
    O = oracle
    for rp, ci, k in ((*O.gen_er(4000, 40000, 5), 20), (*O.gen_rmat(12, 3000, 30000, 9), 16)):
        n = len(rp) - 1
        x0 = 0.5 + np.random.default_rng(1).random(n)
        a, b, Q, xn = O.lanczos(rp, ci, k, x0, q_colmajor=True)
        lam, V = O.eigen(a, b)
        plain = (V @ (np.exp(lam) * (xn * V[0, :]))) @ Q
        shifted = (V @ (np.exp(lam - lam.max()) * (xn * V[0, :]))) @ Q
        s = min(1.0, 40.0 / lam.max())
        capped = (V @ (np.exp(s * (lam - lam.max())) * (xn * V[0, :]))) @ Q
        for reorth in (0, 1):
            R = O.referee_expm(rp, ci, k, x0, caps=(-1.0, 0.0, 40.0), reorth=reorth)
            for got, want in zip(R["ans"], (plain, shifted, capped)):
                assert np.abs(got - want).max() <= 1e-11 * np.abs(want).max(), reorth
            assert abs(R["alpha"][0] - a[0]) <= 1e-14 * abs(a[0]) and abs(R["beta"][0] - b[0]) <= 1e-13 * abs(b[0])
            assert abs(R["lam"].max() - lam.max()) <= 1e-10 * lam.max()
        assert R["orth_loss"] <= 1e-15
        y = O.referee_spmv(rp, ci, x0)
        assert np.allclose(y, O.spmv(rp, ci, x0), rtol=1e-14, atol=0)


def test_arnoldi_restatement(oracle):
    """orc_lanczos_arnoldi restates serial/lib/lanczos.cc:58-132.  While the basis is still orthogonal the pass removes
    nothing but rounding, so it must reproduce the plain loop's answer; with every = 1 it keeps the basis orthogonal where
    the plain loop loses it (hub-heavy R-MAT), and the result then agrees with the fully re-orthogonalised referee."""
    O = oracle
    rp, ci = O.gen_er(4000, 40000, 5)
    x0 = np.ones(4000)
    a, b, Q, xn = O.lanczos(rp, ci, 12, x0, q_colmajor=True)
    for every in (1, 2, 3):
        a2, b2, Q2, xn2 = O.lanczos_arnoldi(rp, ci, 12, x0, every=every)
        assert xn2 == xn and np.array_equal(a2[:3], a[:3]) and np.array_equal(Q2[:4], Q[:4])   # untouched before j = 3 (or 4)
        assert np.allclose(a2, a, rtol=1e-9) and np.allclose(Q2[:8], Q[:8], atol=1e-8), np.abs(Q2 - Q).max(axis=1)
    rp, ci = O.gen_rmat(16, 65536, 1500000, 99, a=0.7, b=0.12, c=0.12)
    n, k = 65536, 30
    x0 = np.ones(n)
    a, b, Q, xn = O.lanczos(rp, ci, k, x0, q_colmajor=True)
    assert max(abs(Q[0] @ Q[j]) for j in range(2, k)) > 1e-3           # the plain loop has lost orthogonality here
    a1, b1, Q1, xn1 = O.lanczos_arnoldi(rp, ci, k, x0, every=1)
    assert max(abs(Q1[0] @ Q1[j]) for j in range(2, k)) < 1e-9
    R = O.referee_expm(rp, ci, k, x0, caps=(40.0,), reorth=1)
    lam, V = O.eigen(a1, b1)
    s = min(1.0, 40.0 / lam.max())
    got = (V @ (np.exp(s * (lam - lam.max())) * (xn1 * V[0, :]))) @ Q1
    assert np.abs(got - R["ans"][0]).max() <= 1e-10 * np.abs(R["ans"][0]).max()


def test_all_core_companion_agrees_with_the_single_thread_loop():
    """oracle/lanczos_oracle_omp.c (bench.py's optional "port-omp" CPU figure, SURVEY.md 8 d) is the same loop with OpenMP over rows and
    elements: its spMV rows are summed as orc_spmv sums them; its inner products are OpenMP reductions, so the coefficients carry
    another rounding -- a timing baseline, checked here only for being the same computation (leading coefficients to 1e-12, x_norm of
    the ones vector exactly)."""
    from oracle import oracle as O
    rp, ci = O.gen_rmat(15, 30000, 300000, 7)
    n = len(rp) - 1
    a, b, _, xn = O.lanczos(rp, ci, 8, np.ones(n), want_q=False)
    a2, b2, xn2, threads = O.lanczos_omp(rp, ci, 8, np.ones(n), threads=4)
    assert threads == 4 and xn2 == xn
    assert abs(a2[0] - a[0]) <= 1e-12 * abs(a[0]) and abs(b2[0] - b[0]) <= 1e-12 * abs(b[0])
    assert np.allclose(a2[:4], a[:4], rtol=1e-9, atol=1e-9 * np.abs(a).max()) and np.allclose(b2[:3], b[:3], rtol=1e-9)


def test_post_loop_steps_equal_the_reference_library_calls(oracle):
    """serial/lib/eigen.cc:13 and multiplyOut.cc:30,33 are three library calls -- LAPACKE_dstevd, cblas_dgemm, cblas_dgemv.  Their
    sources cannot be compiled here (no lapacke.h / cblas.h), but the routines themselves are in SciPy's bundled OpenBLAS: executed
    with the reference's arguments (oracle.eigen_dstevd / mult_out_blas) they reproduce the fixtures' `ans` from the fixtures' own
    alpha / beta, and the oracle's restatements of the two steps agree with them -- at the fixtures' k and at BASELINE's k = 50,
    where dstevd takes its divide-and-conquer path (k > 25) and dstev, which oracle.eigen calls, does not."""
    import glob
    O = oracle
    if O.blas_calls() is None:
        pytest.skip("SciPy's bundled OpenBLAS not found")
    for path in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))):
        g = np.load(path)
        n, k = int(g["mtx_n"]), int(g["k"])
        ro, ci = g["ref_row_offset"].astype(np.uint64), g["ref_col_idx"].astype(np.uint32)
        alpha, beta, Q, xn = O.lanczos(ro, ci, k, np.ones(n))
        assert np.array_equal(alpha, g["alpha"]) and np.array_equal(beta, g["beta"])
        lam, V = O.eigen_dstevd(g["alpha"], g["beta"])           # the reference's eigen step on the fixture's coefficients
        ans = O.mult_out_blas(Q, V, lam, xn)                     # ... and its multOut
        assert np.abs(ans - g["ans"]).max() <= 1e-13 * np.abs(g["ans"]).max(), path
        lam_o, V_o = O.eigen(alpha, beta)
        assert np.abs(lam - lam_o).max() <= 1e-13 * np.abs(lam).max(), path
        assert np.abs(O.mult_out(Q, V_o, lam_o, xn) - ans).max() <= 1e-13 * np.abs(ans).max(), path
    # BASELINE's k: a 1 M-entry graph at k = 50; the answer through e^(A - theta_max) so that fp64 holds it
    ro, ci = O.gen_rmat(14, 12000, 120000, 1234)
    n = len(ro) - 1
    alpha, beta, Q, xn = O.lanczos(ro, ci, 50, np.ones(n))
    lam, V = O.eigen_dstevd(alpha, beta)
    lam_o, V_o = O.eigen(alpha, beta)
    assert np.abs(lam - lam_o).max() <= 1e-13 * np.abs(lam).max()
    shift = lam.max()
    a1, a2 = O.mult_out_blas(Q, V, lam - shift, xn), O.mult_out(Q, V_o, lam_o - shift, xn)
    assert np.abs(a1 - a2).max() <= 1e-12 * np.abs(a1).max()

"""GPU: BASELINE's own Krylov dimension (k = 50) on C2 and C3 with tolerances anchored to something MORE accurate than the
thing under test -- the extended-precision referee (oracle/referee.c: the same recurrence in x87 long double, optionally
with full re-orthogonalisation).  VERDICT round 2, item 1.

What the referee established on C2 (CPU runs, tools/referee_c2.py, recorded in DESIGN.md section 4):
  * the k-step approximation itself has converged by k = 12 (referee_k vs referee_50: 2e-13);
  * the fp64 serial/ algorithm (the oracle) loses orthogonality there (max |q_0 . q_j| = 0.17 from j = 12 on) and is
    3.3e-8 / 8.4e-10 (e^(A - theta_max) x / the s theta_max = 40 functional) away from the referee at k = 50, 8e-11 at k = 10;
  * the referee with and without full re-orthogonalisation agree to 2.4e-12 / 6e-14: it IS accurate enough to judge.
So: err(engine vs referee) <= 1.5 x err(oracle vs referee) at k = 50, and <= 1e-10 at every k where the oracle itself is."""
import numpy as np
import pytest

from bench import C2_DRAWS, C3_DRAWS, ER_DRAWS   # BASELINE's configurations at their named edge counts (bench.py: WORKLOADS)

pytestmark = pytest.mark.gpu

CAPS = (0.0, 40.0)          # referee_expm's caps: 0 -> e^(A - theta_max) x (shift_weights(cap=None)), 40 -> shift_weights(cap=40)


def _protocol(O, eng, rp, ci, n, k, name, small_ks, want_q):
    """engine and oracle against the referee at k and at the prefixes small_ks (a k'-step decomposition is the prefix of a
    k-step one).  Returns the printed table rows."""
    from test_gpu_parity import REL_INF_TOL, rel_inf, shift_weights
    import sys
    import time

    def say(msg):          # a long CPU phase must not look like a hang to the GPU box's silence guard
        print(f"[{name} {time.strftime('%H:%M:%S')}] {msg}", flush=True)
        sys.stderr.write(".")
        sys.stderr.flush()

    x0 = np.ones(n)
    say(f"oracle: {k} iterations of serial/'s loop on one host thread")
    a_ref, b_ref, Q_ref, xn_ref = O.lanczos(rp, ci, k, x0, q_colmajor=True)
    say("engine")
    a, b, Q, xn, st = eng.lanczos(x0, k, want_q=want_q)
    assert xn == xn_ref and st["iters"] == k and np.isfinite(a).all() and np.isfinite(b).all()
    # the same graph with the product library's option `reference_order` -- SpMV one lane
    # per row of the caller's CSR, inner product / norm one left-to-right accumulator (serial/lib/SPMV.cc:24-27,
    # lanczos.cc:155-171).  With serial/'s reduction order the device loop must reproduce the oracle's restatement of serial/
    # BIT FOR BIT at BASELINE's k = 50: all 50 alpha, all 49 beta, every entry of the basis.  So "within 1e-10 of serial/" holds
    # literally (at 0) in that mode, and what separates the production mode from serial/ is the reduction order alone.
    say("engine, reference_order shape")
    import __graft_entry__ as ge
    pkg = ge.load_pkg()
    ref_eng = pkg.Engine(0, reference_order=1)
    ref_eng.set_graph_csr(rp, ci)
    ra, rb, _, rxn, _ = ref_eng.lanczos(x0, k, want_q=False)
    assert rxn == xn_ref
    assert np.array_equal(ra, a_ref), (name, "alpha", np.flatnonzero(ra != a_ref)[:4])
    assert np.array_equal(rb, b_ref), (name, "beta", np.flatnonzero(rb != b_ref)[:4])
    # the basis column by column (no second 4 GB copy at C3): unit coefficient vectors through the device multOut
    for j in (0, 1, k // 2, k - 1):
        e = np.zeros(k)
        e[j] = 1.0
        assert np.array_equal(ref_eng.multout(e), Q_ref[j]), (name, "basis column", j)
    for cap in (None, 40.0):
        d = rel_inf(ref_eng.multout(shift_weights(O, ra, rb, rxn, cap=cap)), shift_weights(O, a_ref, b_ref, xn_ref, cap=cap) @ Q_ref)
        p = rel_inf(eng.multout(shift_weights(O, a, b, xn, cap=cap)), shift_weights(O, a_ref, b_ref, xn_ref, cap=cap) @ Q_ref)
        print(f"{name} k={k} cap={cap}: reference_order shape vs oracle {d:.2e}; production mode vs oracle {p:.2e}; "
              f"alpha / beta of the production mode differ from serial/'s by up to {np.abs(a - a_ref).max() / np.abs(a_ref).max():.2e} / "
              f"{np.abs(b - b_ref).max() / np.abs(b_ref).max():.2e} (reduction order only)")
        assert d <= REL_INF_TOL, (name, cap, d)      # north star, literally: within 1e-10 of serial/ (the multOut sums differ: 1e-15)
    ref_eng.close()
    rows = []
    R0 = None
    for kk in list(small_ks) + [k]:
        say(f"referee, k = {kk}")
        R = R0 = O.referee_expm(rp, ci, kk, x0, caps=CAPS, reorth=0)
        for ci_, cap in enumerate((None, 40.0)):
            ref = R["ans"][ci_]
            assert np.isfinite(ref).all() and np.abs(ref).max() > 0
            e_orc = rel_inf(shift_weights(O, a_ref[:kk], b_ref[:kk - 1], xn_ref, cap=cap) @ Q_ref[:kk], ref)
            t = shift_weights(O, a[:kk], b[:kk - 1], xn, cap=cap)
            e_dev = rel_inf(eng.multout(t), ref)
            e_host = rel_inf(t @ Q[:kk], ref) if want_q else e_dev
            rows.append((kk, cap, e_orc, e_dev, e_host))
            print(f"{name} k={kk} cap={cap}: oracle vs referee {e_orc:.2e}; engine vs referee {e_dev:.2e} (device multOut) "
                  f"{e_host:.2e} (host multOut)")
            # the engine may not be worse than the serial/ algorithm, measured against something better than both
            assert max(e_dev, e_host) <= 1.5 * e_orc + 1e-13, (name, kk, cap, e_orc, e_dev, e_host)
            # ... and wherever serial/ itself achieves the north star's 1e-10, so does the engine
            if e_orc <= REL_INF_TOL:
                assert max(e_dev, e_host) <= REL_INF_TOL, (name, kk, cap, e_orc, e_dev, e_host)
    if k >= 50:
        # the referee is accurate enough to judge: with full re-orthogonalisation (a stand-in for exact-arithmetic
        # Lanczos) it gives the same answer to well below 1e-10
        say("referee with full re-orthogonalisation")
        R1 = O.referee_expm(rp, ci, k, x0, caps=CAPS, reorth=1)
        for ci_ in range(2):
            own = rel_inf(R0["ans"][ci_], R1["ans"][ci_])
            print(f"{name} k={k}: referee vs fully re-orthogonalised referee {own:.2e} (orthogonality lost to {R0['orth_loss']:.2e} / {R1['orth_loss']:.2e})")
            assert own <= 2e-11, (name, ci_, own)
        assert R1["orth_loss"] <= 1e-15
    return rows


def test_c2_k50_against_referee(pkg, oracle):
    O = oracle
    n, k = 1 << 20, 50
    eng = pkg.Engine(0)
    eng.gen_rmat(20, n, C2_DRAWS, 1234)          # BASELINE C2
    rp, ci = eng.get_graph_csr()
    rows = _protocol(O, eng, rp, ci, n, k, "C2", small_ks=(8, 10, 20), want_q=True)
    # the 1e-10 criterion was really exercised somewhere (k = 8 and 10, where serial/ achieves it)
    assert any(r[2] <= 1e-10 for r in rows)
    eng.close()


def test_c3_k50_against_referee(pkg, oracle):
    """The bench's own graph at BASELINE's k = 50: oracle 50 iterations ~35 s of one host core, the
    referee the same on all of them.  The engine's answers come through the device multOut (no 4 GB host copy of Q)."""
    O = oracle
    n, k = 10_000_000, 50
    eng = pkg.Engine(0)
    eng.gen_rmat(24, n, C3_DRAWS, 1234)         # BASELINE C3 / C4 graph
    rp, ci = eng.get_graph_csr()
    _protocol(O, eng, rp, ci, n, k, "C3", small_ks=(8,), want_q=False)
    eng.close()


def test_reorthogonalised_variant(pkg, oracle):
    """R1: the Arnoldi pass of serial/lib/lanczos.cc:58-132 on the device (option reorthogonalise = e) against the oracle's
    restatement of decompose_with_arnoldi, and against the referee.
      * e = 1 and the reference's own e = 2 on fixtures at k = 20: same centrality vector as the restatement (1e-10),
        recurrence-free check: the basis really is orthogonal with e = 1;
      * C2 at k = 50 with e = 1: 1e-10 from the extended-precision referee, where the plain loop (and serial/) are 3e-8 away;
      * the reference's e = 2 at C2, k = 50 is off by O(1) IN THE ORACLE TOO ("neither give good results",
        serial/tests/numerical_test_orthog.cc:3-4): orthogonality goes between two passes and the pass then removes
        components the tridiagonal T does not record."""
    from test_gpu_parity import REL_INF_TOL, graphs, rel_inf, shift_weights
    O = oracle
    asserted = 0
    for name, (rp, ci) in graphs(O):
        n = len(rp) - 1
        k = min(20, n - 1)
        x0 = np.ones(n)
        for e in (1, 2):
            a_ref, b_ref, Q_ref, xn_ref = O.lanczos_arnoldi(rp, ci, k, x0, every=e)
            loss_ref = max(abs(Q_ref[0] @ Q_ref[j]) for j in range(2, k))
            ref = shift_weights(O, a_ref, b_ref, xn_ref) @ Q_ref
            for mode in (dict(propagation_blocking=0), dict(propagation_blocking=1, hub_entries=64)):
                eng = pkg.Engine(0, reorthogonalise=e, **mode)
                eng.set_graph_csr(rp, ci)
                a, b, Q, xn, st = eng.lanczos(x0, k)
                assert xn == xn_ref and st["iters"] == k
                loss = max(abs(Q[0] @ Q[j]) for j in range(2, k))
                if e == 1:
                    assert loss <= 1e-9, (name, mode, loss)      # the pass does its job
                # where the restatement keeps its basis orthogonal the two must agree at the north star's tolerance (where it
                # does not -- e = 2 on the hub-heavy graph -- both are chaotic: see the C2 leg below)
                if loss_ref <= 1e-8:
                    got = rel_inf(eng.multout(shift_weights(O, a, b, xn)), ref)
                    assert got <= REL_INF_TOL, (name, e, mode, got)
                    assert abs(a[0] - a_ref[0]) <= 1e-12 * abs(a_ref[0]) and abs(b[0] - b_ref[0]) <= 1e-12 * abs(b_ref[0])
                    asserted += 1
                eng.close()
    assert asserted >= 10, asserted
    # the option may change between decompositions on one handle: off again = the plain loop
    rp, ci = O.gen_er(10000, 100000, 1234)
    eng = pkg.Engine(0)
    eng.set_graph_csr(rp, ci)
    a0, b0, Q0, _, _ = eng.lanczos(np.ones(10000), 12)
    eng.set_option("reorthogonalise", 1)
    a1, b1, Q1, _, _ = eng.lanczos(np.ones(10000), 12)
    eng.set_option("reorthogonalise", 0)
    a2, b2, Q2, _, _ = eng.lanczos(np.ones(10000), 12)
    assert np.array_equal(a0, a2) and np.array_equal(b0, b2) and np.array_equal(Q0, Q2)
    assert np.allclose(a0[:8], a1[:8], rtol=1e-9)
    eng.close()

    # C2, k = 50
    n, k = 1 << 20, 50
    eng = pkg.Engine(0, reorthogonalise=1)
    eng.gen_rmat(20, n, C2_DRAWS, 1234)
    rp, ci = eng.get_graph_csr()
    x0 = np.ones(n)
    R = O.referee_expm(rp, ci, k, x0, caps=CAPS, reorth=1)
    a, b, Q, xn, st = eng.lanczos(x0, k, want_q=False)
    for ci_, cap in enumerate((None, 40.0)):
        got = rel_inf(eng.multout(shift_weights(O, a, b, xn, cap=cap)), R["ans"][ci_])
        print(f"C2 k=50 reorthogonalise=1 cap={cap}: engine vs referee {got:.2e}; Arnoldi passes {st['vec_ms']:.1f} ms of {st['loop_ms']:.1f} ms")
        assert got <= REL_INF_TOL, (cap, got)
    eng.close()
    # The reference's own constant (every 2) and every 3: between two passes orthogonality keeps decaying (three- to tenfold per
    # pass pair on this graph) and once it is gone the pass removes components T does not record.  How soon depends on the
    # level the decay STARTS from: the oracle's left-to-right sums put it at 1e-11 and it is garbage by k = 50 already with
    # every 2; the engine starts at 1e-16 and is still fine at k = 50 with every 2 (2e-15; max |q_0 . q_j| has reached 0.14
    # by then) but not with every 3 (tools/reorth_probe.py, profiles/r3_reorth_probe.txt).
    for e, expect_engine_bad in ((2, None), (3, True)):
        eng = pkg.Engine(0, reorthogonalise=e)
        eng.gen_rmat(20, n, C2_DRAWS, 1234)
        a, b, _, xn, _ = eng.lanczos(x0, k, want_q=False)
        a_ref, b_ref, Q_ref, xn_ref = O.lanczos_arnoldi(rp, ci, k, x0, every=e)
        bad_engine = rel_inf(eng.multout(shift_weights(O, a, b, xn, cap=40.0)), R["ans"][1])
        bad_oracle = rel_inf(shift_weights(O, a_ref, b_ref, xn_ref, cap=40.0) @ Q_ref, R["ans"][1])
        print(f"C2 k=50 reorthogonalise={e}: engine vs referee {bad_engine:.2e}, oracle restatement vs referee {bad_oracle:.2e}")
        assert bad_oracle > 1e-3                              # the reference's own finding, reproduced by its restatement
        if expect_engine_bad:
            assert bad_engine > 1e-3
        eng.close()


def test_run_in_chunks_and_early_stop(pkg, oracle):
    """N3 proper: lzx_lanczos_run_steps continues a prepared decomposition from the state resident in HBM -- same bits as in
    one go, in every form of the loop -- and lzx_multout_change_f64 gives the host's stopping rule two scalars instead of an
    n-vector, so that a converged answer SAVES SpMVs (parallel-final/lib/multiplyOut.cu:25-49 can only look at a finished
    decomposition; writeup section 11)."""
    from test_gpu_parity import rel_inf
    O = oracle
    rp, ci = O.gen_rmat(14, 12000, 200000, 7)
    n, K = len(rp) - 1, 30
    x0 = np.ones(n)
    for mode in (dict(propagation_blocking=0), dict(propagation_blocking=1, hub_entries=256), dict(propagation_blocking=0, lazy_normalisation=1),
                 dict(propagation_blocking=1, hub_entries=256, lazy_normalisation=0), dict(propagation_blocking=0, reorthogonalise=1),
                 dict(propagation_blocking=1, hub_entries=256, lazy_normalisation=1, basis_fp32=1)):
        eng = pkg.Engine(0, **mode)
        eng.set_graph_csr(rp, ci)
        a1, b1, Q1, xn1, _ = eng.lanczos(x0, K)
        eng.lanczos_prepare(x0, K)
        done = 0
        for steps in (1, 2, 7, 5, 100):
            st = eng.lanczos_run_steps(steps)
            assert st["iters"] == min(steps, K - done), (mode, steps, st["iters"])
            done = min(K, done + steps)
            assert eng.lanczos_progress() == (done, K if done < K else 0), mode
            # what is there so far can be used: a prefix of the one-go run, bit for bit
            a, b, Q = eng.lanczos_fetch(done, want_q=True)
            assert np.array_equal(a, a1[:done]) and np.array_equal(b, b1[:done - 1]) and np.array_equal(Q, Q1[:done]), (mode, done)
            t = np.random.default_rng(done).random(done)
            assert np.allclose(eng.multout(t), t @ Q, rtol=1e-12, atol=1e-14) or mode.get("basis_fp32")
        with pytest.raises(pkg.LzxError):
            eng.lanczos_run_steps(1)                      # complete: nothing left
        eng.lanczos_prepare(x0, K)
        eng.lanczos_run_steps(3)
        eng.spmv(x0)                                      # overwrites the loop's work vectors: the preparation is void
        with pytest.raises(pkg.LzxError):
            eng.lanczos_run_steps(1)
        eng.close()

    # early stop on BASELINE C1 (theta_max ~ 21: e^A x representable): chunks of 5, stop when the answer moved < 1e-12
    rp, ci = O.gen_er(10000, 100000, 1234)
    n, K, step, tol = 10000, 50, 5, 1e-12
    x0 = np.ones(n)
    eng = pkg.Engine(0)
    eng.set_graph_csr(rp, ci)
    xn = eng.lanczos_prepare(x0, K)
    used, changes, spmvs = 0, [], 0
    for k in range(step, K + 1, step):
        spmvs += eng.lanczos_run_steps(step)["iters"]
        a, b, _ = eng.lanczos_fetch(k)
        lam, V = O.eigen(a, b)
        t = V @ (np.exp(lam) * (xn * V[0, :]))
        changes.append(eng.multout_change(t))
        used = k
        if changes[-1] <= tol:
            break
    assert changes[0] == 1.0 and used < K and spmvs == used, (used, changes)
    ans = eng.multout(t)
    # identical to K-then-truncate: the full run's leading block gives the same answer ...
    eng2 = pkg.Engine(0)
    eng2.set_graph_csr(rp, ci)
    a2, b2, Q2, xn2, _ = eng2.lanczos(x0, K)
    lam2, V2 = O.eigen(a2[:used], b2[:used - 1])
    assert np.array_equal(eng2.multout(V2 @ (np.exp(lam2) * (xn2 * V2[0, :]))), ans)
    # ... the monitored changes are those of the answers themselves, and the converged answer is the oracle's e^A x
    prev = None
    for i, k in enumerate(range(step, used + 1, step)):
        lam_k, V_k = O.eigen(a2[:k], b2[:k - 1])
        y = (V_k @ (np.exp(lam_k) * (xn2 * V_k[0, :]))) @ Q2[:k]
        if prev is not None:
            assert abs(changes[i] - np.linalg.norm(y - prev) / np.linalg.norm(y)) <= 1e-9 * max(changes[i], 1e-6), (k, changes[i])
        prev = y
    ans_ref = O.expm_action(rp, ci, K, x0)
    assert rel_inf(ans, ans_ref) <= 1e-10
    print(f"C1 early stop: k_used = {used} of {K} (changes {['%.1e' % c for c in changes]}), {spmvs} SpMVs run")
    eng.close()
    eng2.close()


def test_basis_stored_as_fp32(pkg, oracle):
    """N4 remainder: option basis_fp32 -- the resident basis in fp32, the recurrence's three live vectors in fp64.  alpha /
    beta are those of the fp64 loop BIT FOR BIT (the loop never reads a rounded column); the centrality vector carries the
    6e-8 rounding of the stored columns: ~1e-8, outside the 1e-10 criterion (the reference's float runs: 1.2e-6 at best,
    parallel-final/output/single_double.txt:58-63), for half the HBM."""
    from test_gpu_parity import rel_inf, shift_weights
    O = oracle
    for name, (rp, ci), k in (("er_c1", O.gen_er(10000, 100000, 1234), 20), ("rmat_iso", O.gen_rmat(15, 30000, 120000, 11), 14)):
        n = len(rp) - 1
        x0 = 0.5 + np.random.default_rng(3).random(n)
        a_ref, b_ref, Q_ref, xn_ref = O.lanczos(rp, ci, k, x0, q_colmajor=True)
        ref = shift_weights(O, a_ref, b_ref, xn_ref) @ Q_ref
        for mode in (dict(propagation_blocking=0, lazy_normalisation=1), dict(propagation_blocking=1, hub_entries=256)):
            e64 = pkg.Engine(0, **mode)
            e64.set_graph_csr(rp, ci)
            a64, b64, Q64, xn, _ = e64.lanczos(x0, k)
            e32 = pkg.Engine(0, basis_fp32=1, **mode)
            e32.set_graph_csr(rp, ci)
            xn32 = e32.lanczos_prepare(x0, k)
            e32.lanczos_run()
            t = shift_weights(O, a64, b64, xn)
            ans_factored = e32.multout(t)                     # rows without an edge still as scalars times q_0
            a32, b32, Q32 = e32.lanczos_fetch(k, want_q=True)
            assert xn32 == xn and np.array_equal(a32, a64) and np.array_equal(b32, b64), (name, mode)
            scale = np.abs(Q64).max(axis=1, keepdims=True)
            assert np.abs(Q32 - Q64).max() <= 1e-7 * scale.max() and np.abs(Q32 - Q64).max() > 0
            err = rel_inf(e32.multout(t), ref)
            err_f = rel_inf(ans_factored, ref)
            err64 = rel_inf(e64.multout(t), ref)
            print(f"{name} {mode}: centrality vector vs oracle: fp32-stored basis {err:.2e} ({err_f:.2e} before the fetch), fp64 basis {err64:.2e}")
            assert err64 <= 1e-10 and 1e-10 < err <= 1e-6 and err_f <= 1e-6
            e64.close()
            e32.close()
    # the reference-order loop keeps its basis in fp64: asking for both is refused, not ignored
    eng = pkg.Engine(0, propagation_blocking=0, lazy_normalisation=0, basis_fp32=1)
    eng.set_graph_csr(*O.gen_er(1000, 5000, 1))
    with pytest.raises(pkg.LzxError):
        eng.lanczos(np.ones(1000), 5)
    eng.close()
    # three in-process ranks
    rp, ci = O.gen_er(10000, 100000, 1234)
    x0 = np.ones(10000)
    a_ref, b_ref, Q_ref, xn_ref = O.lanczos(rp, ci, 20, x0, q_colmajor=True)
    ref = shift_weights(O, a_ref, b_ref, xn_ref) @ Q_ref
    grp = pkg.LocalGroup([0, 0, 0], basis_fp32=1)
    grp.set_graph_csr(rp, ci)
    a, b, Q, xn, _ = grp.lanczos(x0, 20)
    err = rel_inf(grp.multout(shift_weights(O, a, b, xn)), ref)
    assert 1e-10 < err <= 1e-6 and np.abs(Q[:4] - Q_ref[:4]).max() <= 1e-6
    grp.close()

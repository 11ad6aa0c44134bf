"""GPU: handle lifecycle and reproducibility through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_runs_are_reproducible_bit_for_bit(pkg, oracle):
    """Fixed-shape reductions, no global atomics: the same handle, a fresh handle and a repeated run agree bit for bit
    (plain mode by construction; blocked mode as observed -- see DESIGN.md section 3)."""
    O = oracle
    rp, ci = O.gen_rmat(16, 50000, 600000, 11)
    n = len(rp) - 1
    # the last mode: the gather pass's dynamic tail -- which workgroup draws which item differs from run to run, the bits
    # must not (every drawn item leaves its own share of alpha, closed in ticket order)
    for mode in (dict(propagation_blocking=0), dict(propagation_blocking=1, hub_entries=512),
                 dict(propagation_blocking=1, hub_entries=64, pb_target=1024, pb_gather_grid=8, pb_dyn_share=50)):
        e1 = pkg.Engine(0, **mode)
        e1.set_graph_csr(rp, ci)
        a1, b1, Q1, _, _ = e1.lanczos(np.ones(n), 15)
        a2, b2, Q2, _, _ = e1.lanczos(np.ones(n), 15)
        e2 = pkg.Engine(0, **mode)
        e2.set_graph_csr(rp, ci)
        a3, b3, Q3, _, _ = e2.lanczos(np.ones(n), 15)
        for a, b, Q in ((a2, b2, Q2), (a3, b3, Q3)):
            assert np.array_equal(a1, a) and np.array_equal(b1, b) and np.array_equal(Q1, Q), mode
        e1.close()
        e2.close()


def test_handle_reuse_and_growth(pkg, oracle):
    O = oracle
    rp, ci = O.gen_er(3000, 20000, 5)
    rp2, ci2 = O.gen_er(1200, 9000, 6)
    eng = pkg.Engine(0)
    eng.set_graph_csr(rp, ci)
    a5, b5, _, _, _ = eng.lanczos(np.ones(3000), 5)
    a30, b30, Q30, xn, _ = eng.lanczos(np.ones(3000), 30)       # the resident basis grows
    assert np.array_equal(a30[:5], a5) and np.array_equal(b30[:4], b5)   # a longer run extends a shorter one
    t = np.random.default_rng(0).random(12)
    assert np.allclose(eng.multout(t), t @ Q30[:12], rtol=1e-12, atol=1e-14)   # multOut on a prefix of the basis
    eng.set_graph_csr(rp2, ci2)                                  # replace the graph on a live handle
    with pytest.raises(pkg.LzxError):
        eng.multout(t)                                           # the old basis is gone
    x = np.random.default_rng(1).random(1200)
    assert np.array_equal(eng.spmv(x), O.spmv(rp2, ci2, x))
    other = pkg.Engine(0)                                        # two live handles on one GPU
    other.set_graph_csr(rp, ci)
    y = np.random.default_rng(2).random(3000)
    assert np.array_equal(other.spmv(y), O.spmv(rp, ci, y))
    assert np.array_equal(eng.spmv(x), O.spmv(rp2, ci2, x))
    other.close()
    eng.close()


def test_prepared_state_is_voided_by_other_calls(pkg, oracle):
    """lzx_lanczos_prepare_f64 leaves q_0 in work buffers; handing over another graph frees them, and
    lzx_spmv_f64 / lzx_multout_f64 / lzx_bench_spmv overwrite them.  Each of these now voids the preparation, so a
    following lzx_lanczos_run is an LZX_ERR_STATE instead of a device fault or silently wrong coefficients."""
    O = oracle
    rp, ci = O.gen_er(3000, 20000, 5)
    rp2, ci2 = O.gen_er(1200, 9000, 6)
    eng = pkg.Engine(0)
    eng.set_graph_csr(rp, ci)
    eng.lanczos_prepare(np.ones(3000), 6)
    eng.set_graph_csr(rp2, ci2)                                  # frees the basis the preparation sized
    with pytest.raises(pkg.LzxError):
        eng.lanczos_run()
    a_ref, b_ref, _, _, _ = eng.lanczos(np.ones(1200), 6)
    for clobber in (lambda: eng.spmv(np.ones(1200)), lambda: eng.bench_spmv(1)):
        eng.lanczos_prepare(np.ones(1200), 6)
        clobber()
        with pytest.raises(pkg.LzxError):
            eng.lanczos_run()
    eng.lanczos_prepare(np.ones(1200), 6)
    with pytest.raises(pkg.LzxError):
        eng.multout(np.ones(3))                                  # a preparation invalidates the previous basis at once
    eng.lanczos_prepare(np.ones(1200), 6)                        # the straight sequence still works, same numbers
    eng.lanczos_run()
    a, b, _ = eng.lanczos_fetch(6)
    assert np.array_equal(a, a_ref) and np.array_equal(b, b_ref)
    with pytest.raises(pkg.LzxError):
        eng.lanczos_run()                                        # a preparation is consumed by its run
    eng.close()


def test_csr_validation(pkg):
    eng = pkg.Engine(0)
    with pytest.raises(pkg.LzxError):                             # row_ptr decreases (sums still match)
        eng.set_graph_csr(np.array([0, 2, 1, 3], dtype=np.uint64), np.array([1, 2, 0], dtype=np.uint32))
    with pytest.raises(pkg.LzxError):                             # a column index >= n would index the reshaping tables out of bounds
        eng.set_graph_csr(np.array([0, 1, 2, 3], dtype=np.uint64), np.array([1, 7, 0], dtype=np.uint32))
    with pytest.raises(pkg.LzxError):                             # an edge endpoint >= n in the device ingest
        eng.set_graph_edges(4, np.array([0, 9], dtype=np.uint32), np.array([1, 2], dtype=np.uint32))
    eng.set_graph_csr(np.array([0, 1, 2, 2], dtype=np.uint64), np.array([1, 0], dtype=np.uint32))   # still usable afterwards
    assert np.array_equal(eng.spmv(np.array([1.0, 2.0, 3.0])), np.array([2.0, 1.0, 0.0]))
    with pytest.raises(pkg.LzxError):
        eng.set_graph_csr(np.array([0, 1, 3], dtype=np.uint64), np.array([1, 0], dtype=np.uint32))   # row_ptr[n] != nnz
    with pytest.raises(pkg.LzxError):
        eng.set_graph_csr(np.array([1, 1, 2], dtype=np.uint64), np.array([1, 0], dtype=np.uint32))   # row_ptr[0] != 0
    fresh = pkg.Engine(0)                                        # options go in before the graph
    fresh.set_option("hub_entries", 4096)
    with pytest.raises(pkg.LzxError):
        fresh.set_option("no_such_option", 1)
    with pytest.raises(pkg.LzxError):
        fresh.set_option("phase_mask", 3)                        # an experiment knob: liblzx_dbg.so only
    with pytest.raises(pkg.LzxError):                            # ... also through the product's test-only shape entry
        pkg._check(fresh.L.lzx_test_set_shape(fresh.h, b"phase_mask", 3), "lzx_test_set_shape", fresh.L)
    fresh.set_option("pb_reduce", 1)                             # a table SHAPE: served by liblzx.so itself (lzx_test_set_shape)
    fresh.close()
    dbg = pkg.Engine(0, phase_mask=3)                            # an experiment knob makes the package load the debug library
    assert dbg.L is not eng.L
    dbg.close()
    with pytest.raises(pkg.LzxError):
        pkg.Engine(99)                                            # no such device
    eng.close()


def test_larger_graph_on_a_handle_that_ran_the_optional_loop_forms(pkg, oracle):
    """the fp32-stored basis, its three live fp64 vectors and the convergence monitor's two answers are
    sized by the graph (ldq / n_loc_pad).  Handing a LARGER graph to the same handle must drop them with the graph -- otherwise
    prepare's clears, k_lazy_update's fp32 stores and lzx_multout_change's writes run past the old allocations.  Also the
    mode switch fp64 basis <-> fp32 basis on a resident graph (the representation not in use is freed) and a decomposition
    after a failed one."""
    from test_gpu_parity import shift_weights
    O = oracle
    small = O.gen_er(3000, 20000, 5)
    large = O.gen_rmat(16, 60000, 900000, 12)
    eng = pkg.Engine(0, propagation_blocking=1, hub_entries=64)
    for rp, ci in (small, large, small, large):
        n = len(rp) - 1
        x0 = np.ones(n)
        eng.set_graph_csr(rp, ci)
        eng.set_option("basis_fp32", 1)
        a32, b32, _, xn, _ = eng.lanczos(x0, 12, want_q=False)
        t = shift_weights(O, a32, b32, xn)
        assert eng.multout_change(t) == 1.0                    # first answer since the prepare
        assert eng.multout_change(t) == 0.0                    # the same coefficients again: no change
        y32 = eng.multout(t)
        eng.set_option("basis_fp32", 0)                        # back to the fp64 basis on the resident graph
        a64, b64, Q, xn64, _ = eng.lanczos(x0, 12)
        assert np.array_equal(a32, a64) and np.array_equal(b32, b64) and xn == xn64   # the loop never reads a rounded column
        y64 = eng.multout(t)
        assert np.abs(y64 - t @ Q).max() <= 1e-12 * np.abs(y64).max()
        assert np.abs(y32 - y64).max() <= 1e-6 * np.abs(y64).max()
        assert eng.multout_change(t) == 1.0 and eng.multout_change(t) == 0.0
        a_ref, b_ref, _, _ = O.lanczos(rp, ci, 3, x0, want_q=False)
        assert abs(a64[0] - a_ref[0]) <= 1e-12 * abs(a_ref[0]) and abs(b64[0] - b_ref[0]) <= 1e-12 * abs(b_ref[0])
    # a prepare that fails (k = 0) leaves a handle that reports "no decomposition", then works again
    with pytest.raises(pkg.LzxError):
        eng.lanczos_prepare(np.ones(eng.n), 0)
    a, b, _, _, _ = eng.lanczos(np.ones(eng.n), 12, want_q=False)
    assert np.array_equal(a, a64) and np.array_equal(b, b64)
    eng.close()


def test_placement_trials_change_no_bit(pkg):
    """Option placement_trials: the value stream of the blocked SpMV is allocated a few more times at the hand-over and the
    fastest placement kept (lzx_pb.hip: lzx_pb_place_values).  The winner is a matter of speed only: same SpMV bits, same
    coefficients as with the first allocation; a stream the caches hold is left alone."""
    n, draws = 2_000_000, 20_000_000
    x = np.random.default_rng(11).random(n)
    out = {}
    for trials in (0, 3):
        eng = pkg.Engine(0, propagation_blocking=1, placement_trials=trials)
        eng.gen_er(n, draws, 21)
        assert eng.info()["pb_values"] * 8 > (128 << 20)
        tried = eng.shape("placement_tried")
        assert tried == (0 if trials == 0 else trials + 1), tried
        if trials:
            times = [eng.shape(f"placement_us_{i}") for i in range(tried)]
            kept = eng.shape("placement_kept")
            assert all(t > 0 for t in times) and times[kept] == min(times), (times, kept)
        y = eng.spmv(x)
        a, b, _, xn, _ = eng.lanczos(np.ones(n), 6)
        out[trials] = (y, a, b)
        eng.close()
    assert np.array_equal(out[0][0], out[3][0]) and np.array_equal(out[0][1], out[3][1]) and np.array_equal(out[0][2], out[3][2])
    small = pkg.Engine(0, propagation_blocking=1, hub_entries=256)
    small.gen_er(200_000, 1_000_000, 3)
    assert small.shape("placement_tried") == 0
    small.close()


def test_start_vector_hand_over_fast_paths(pkg, oracle):
    """Round 5: lzx_lanczos_prepare_f64 looks at x0 once.  A constant vector -- the reference's own start
    vector is ones, parallel-final/main.cu:79 -- is filled on the device instead of crossing PCIe; a sum of squares that is exact
    in any order (integer entries) is formed by several host threads instead of serial/'s one dependent chain
    (serial/lib/lanczos.cc:155-161).  Neither may change a bit: compared with the plain hand-over (test shape
    start_vector_scan = 0) and with the oracle's norm."""
    O = oracle
    rp, ci = O.gen_rmat(17, 120000, 1500000, 3)
    n = len(rp) - 1
    rng = np.random.default_rng(5)
    cases = {
        "ones": (np.ones(n), True),
        "constant 0.3 (squares not exact)": (np.full(n, 0.3), True),
        "small integers": (rng.integers(-9, 10, n).astype(np.float64), False),
        "general": (rng.random(n), False),
        "one entry differs, at the end": (np.concatenate([np.ones(n - 1), [2.0]]), False),
        "too large for exact squares": (np.full(n, 2.0 ** 27), True),
    }
    for mode in (dict(propagation_blocking=0), dict(propagation_blocking=1, hub_entries=512)):
        fast = pkg.Engine(0, **mode)
        plain = pkg.Engine(0, start_vector_scan=0, **mode)
        fast.set_graph_csr(rp, ci)
        plain.set_graph_csr(rp, ci)
        for name, (x0, const) in cases.items():
            a1, b1, Q1, xn1, _ = fast.lanczos(x0, 8)
            assert fast.shape("start_vector_was_constant") == int(const), name
            a0, b0, Q0, xn0, _ = plain.lanczos(x0, 8)
            assert plain.shape("start_vector_was_constant") == 0
            assert xn1 == xn0 == O.lanczos(rp, ci, 1, x0, want_q=False)[3], name   # serial/'s left-to-right norm, to the bit
            assert np.array_equal(a1, a0) and np.array_equal(b1, b0) and np.array_equal(Q1, Q0), (name, mode)
        fast.close()
        plain.close()


def test_finish_launch_deferred_into_the_vector_kernel(pkg, oracle):
    """Round 5: on graphs whose row bands are cut into several gather items the blocked SpMV ended with a
    fourth launch, k_pb_finish, that added the per-item totals (and the split rows' item totals) to v.  In the lazy loop
    k_lazy_update now adds those totals where it reads v and the launch is left out; the rows' share of alpha is formed by the gather
    pass's items either way.  Same operands in the same order: every coefficient and basis column equals the undeferred form's
    (test shape defer_finish = 0) BIT FOR BIT -- on one rank and on three, with split rows inside multi-item bands -- and lzx_spmv_f64,
    which hands v out, still gets the launch."""
    O = oracle
    rp, ci = O.gen_rmat(17, 120000, 2500000, 9)
    n = len(rp) - 1
    x = np.random.default_rng(3).random(n)
    y_ref = O.spmv(rp, ci, x)
    # small gather items: many multi-item bands; a low split-row threshold: split rows inside them
    shape = dict(propagation_blocking=1, hub_entries=512, pb_target=1024, long_row=24)
    for world in (1, 3):
        runs = {}
        for defer in (1, 0):
            g = pkg.Engine(0, defer_finish=defer, **shape) if world == 1 else pkg.LocalGroup([0] * world, defer_finish=defer, **shape)
            g.set_graph_csr(rp, ci)
            e0 = g if world == 1 else g.engines[0]
            assert e0.shape("finish_launched") == 1 and e0.shape("finish_deferrable") == 1 and e0.info()["long_rows"] > 0
            assert np.allclose(g.spmv(x), y_ref, rtol=1e-13, atol=0)                # v complete for those who read it
            a, b, Q, xn, st = g.lanczos(np.ones(n), 12)
            assert st["spmv_kernels"] == (3 if defer else 4), st
            assert np.allclose(g.spmv(x), y_ref, rtol=1e-13, atol=0)                # ... also after a loop that deferred
            runs[defer] = (a, b, Q)
            g.close()
        for u, w in zip(runs[1], runs[0]):
            assert np.array_equal(u, w), world
        a_ref, b_ref, _, _ = O.lanczos(rp, ci, 12, np.ones(n), want_q=False)
        assert abs(runs[1][0][0] - a_ref[0]) <= 1e-12 * abs(a_ref[0]) and abs(runs[1][1][0] - b_ref[0]) <= 1e-12 * abs(b_ref[0])


def test_create_group_in_one_call(pkg, oracle):
    """SURVEY.md 8(b) sketched lzx_create(out, n_devices, device_ids); the ABI has one handle per GPU and three ways of wiring them.
    lzx_create_group (round 5) is the sketch's call: n handles on the named GPUs (ids may repeat), wired as one in-process group --
    the same numbers as handles made and wired one by one, and nothing left behind on a bad device id."""
    import ctypes
    O = oracle
    rp, ci = O.gen_rmat(15, 30000, 400000, 4)
    n = len(rp) - 1
    x = np.random.default_rng(1).random(n)
    one_call = pkg.LocalGroup.create([0, 0, 0])
    by_hand = pkg.LocalGroup([0, 0, 0])
    for g in (one_call, by_hand):
        g.set_graph_csr(rp, ci)
    assert [e.info()["rank"] for e in one_call.engines] == [0, 1, 2] and one_call.engines[2].info()["world"] == 3
    assert np.array_equal(one_call.spmv(x), by_hand.spmv(x)) and np.allclose(one_call.spmv(x), O.spmv(rp, ci, x), rtol=1e-13, atol=0)
    a1, b1, Q1, _, _ = one_call.lanczos(np.ones(n), 8)
    a2, b2, Q2, _, _ = by_hand.lanczos(np.ones(n), 8)
    assert np.array_equal(a1, a2) and np.array_equal(b1, b2) and np.array_equal(Q1, Q2)
    one_call.close()
    by_hand.close()
    L = pkg.lib()
    arr = (ctypes.c_void_p * 2)()
    assert L.lzx_create_group(arr, 2, (ctypes.c_int * 2)(0, 4096)) != 0 and not arr[0] and not arr[1]

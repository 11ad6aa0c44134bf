"""GPU: the sharded graph hand-over (option sharded_ingest; SURVEY.md 7.1 step 7 -- a loader for graphs one device need not
hold, parallel-final/lib/adjMatrix.cc:21-46).  A rank sweeps the source (seeded generator or edge list) in bounded batches and
keeps its OWN rows only.  The bar: the tables it builds are those of the whole-graph hand-over entry for entry, so the SpMV,
every Lanczos coefficient and every basis column agree with it BIT FOR BIT -- at one rank and at three, plain and blocked,
with the sparse second exchange chunk derived from the rank's own rows -- and with the oracle at the usual tolerances."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

INFO_KEYS = ("n", "nnz", "max_degree", "rows_local", "nnz_local", "long_rows", "sell_padded", "pb_entries", "active_vertices",
             "exchange_slice", "hub_entries", "pb_values", "pb_reduced_entries", "exchange_chunk0", "exchange_recv")


def _build(pkg, world, how, spec, **options):
    g = pkg.Engine(0, **options) if world == 1 else pkg.LocalGroup([0] * world, **options)
    if how == "rmat":
        g.gen_rmat(*spec)
    elif how == "er":
        g.gen_er(*spec)
    elif how == "csr":
        g.set_graph_csr(*spec)
    elif how == "csr32":
        engines = g.engines if world > 1 else [g]
        for e in engines:
            e.set_graph_csr32(spec[0].astype(np.uint32), spec[1])
        g.n = len(spec[0]) - 1
    else:
        g.set_graph_edges(*spec)
    return g


def _infos(g):
    engines = g.engines if hasattr(g, "engines") else [g]
    return [{k: e.info()[k] for k in INFO_KEYS} for e in engines]


@pytest.mark.parametrize("world", [1, 3])
def test_sharded_equals_whole_graph_bit_for_bit(pkg, oracle, world):
    O = oracle
    n, k = 40000, 12
    rng = np.random.default_rng(3)
    # an edge list with duplicates, both orientations and a few self loops (one diagonal entry each, as the std::set build leaves them)
    m = 300000
    src = (rng.random(m) ** 2 * n).astype(np.uint32)
    dst = (rng.random(m) ** 2 * n).astype(np.uint32)
    src[:50] = dst[:50]
    # ... and a CSR the caller holds in host memory (streamed past the device in row chunks; 64- and 32-bit offsets)
    rp_h, ci_h = O.gen_rmat(16, n, 400000, 23)
    cases = (("rmat", (16, n, 400000, 21)), ("er", (n, 250000, 22)), ("edges", (n, src, dst)), ("csr", (rp_h, ci_h)), ("csr32", (rp_h, ci_h)))
    # plain; blocked with the default second chunk; blocked with every run reduced and small gather items
    modes = (dict(propagation_blocking=0), dict(propagation_blocking=1, hub_entries=512),
             dict(propagation_blocking=1, hub_entries=256, pb_reduce=16, pb_target=2048))
    x = rng.random(n)
    x0 = np.ones(n)
    for how, spec in cases:
        for mode in modes:
            whole = _build(pkg, world, how, spec, **mode)
            y_w = whole.spmv(x)
            a_w, b_w, Q_w, xn_w, _ = whole.lanczos(x0, k)
            info_w = _infos(whole)
            for sweeps in (1, 3, 7):
                part = _build(pkg, world, how, spec, sharded_ingest=sweeps, **mode)
                assert _infos(part) == info_w, (how, mode, sweeps)
                assert np.array_equal(part.spmv(x), y_w), (how, mode, sweeps)
                a, b, Q, xn, _ = part.lanczos(x0, k)
                assert xn == xn_w and np.array_equal(a, a_w) and np.array_equal(b, b_w) and np.array_equal(Q, Q_w), (how, mode, sweeps)
                part.close()
            whole.close()
    # and the oracle, on the generated graph (the generators share their integer specification with it)
    rp, ci = O.gen_rmat(16, n, 400000, 21)
    g = _build(pkg, world, "rmat", (16, n, 400000, 21), sharded_ingest=3, propagation_blocking=1, hub_entries=512)
    assert np.allclose(g.spmv(x), O.spmv(rp, ci, x), rtol=1e-13, atol=0)
    a, b, _, _, _ = g.lanczos(x0, k)
    a_ref, b_ref, _, _ = O.lanczos(rp, ci, k, x0, q_colmajor=True)
    assert abs(a[0] - a_ref[0]) <= 1e-12 * abs(a_ref[0]) and abs(b[0] - b_ref[0]) <= 1e-12 * abs(b_ref[0])
    g.close()


def test_sharded_rank_holds_its_own_rows_only(pkg, oracle):
    """What the option is for: with several ranks a handle's device CSR holds the rank's rows alone (their entry counts add up
    to the matrix), the whole graph cannot be read back from it, and at one rank it can -- equal to the whole-graph CSR."""
    O = oracle
    n = 30000
    spec = (15, n, 300000, 5)
    rp, ci = O.gen_rmat(*spec)
    grp = pkg.LocalGroup([0] * 4, sharded_ingest=2)
    grp.gen_rmat(*spec)
    infos = [e.info() for e in grp.engines]
    assert all(i["nnz"] == len(ci) for i in infos) and sum(i["nnz_local"] for i in infos) == len(ci)
    deg = np.diff(rp.astype(np.int64))
    seen = np.zeros(n, dtype=bool)
    for e in grp.engines:                                     # every local row's sum of ones = its vertex's degree, rank by rank
        v, ids = e.rank_row_sums()
        assert np.array_equal(v.astype(np.int64), deg[ids])
        seen[ids] = True
    assert seen.all()
    with pytest.raises(pkg.LzxError):
        grp.engines[1].get_graph_csr()
    grp.close()
    one = pkg.Engine(0, sharded_ingest=5)
    one.gen_rmat(*spec)
    rp1, ci1 = one.get_graph_csr()
    assert np.array_equal(rp1, rp) and np.array_equal(ci1, ci)
    one.close()


def test_sharded_csr_rejects_a_bad_column(pkg, oracle):
    rp, ci = oracle.gen_er(5000, 30000, 9)
    bad = ci.copy()
    bad[len(bad) // 2] = 5000
    for mode in (dict(propagation_blocking=0), dict(propagation_blocking=1, hub_entries=256)):
        eng = pkg.Engine(0, sharded_ingest=3, **mode)
        with pytest.raises(pkg.LzxError):
            eng.set_graph_csr(rp, bad)
        eng.set_graph_csr(rp, ci)                                # the handle is still good
        x = np.random.default_rng(1).random(5000)
        assert np.allclose(eng.spmv(x), oracle.spmv(rp, ci, x), rtol=1e-13, atol=0)
        eng.close()


def test_sharded_edge_list_errors_and_empty(pkg):
    eng = pkg.Engine(0, sharded_ingest=1)
    with pytest.raises(pkg.LzxError):
        eng.set_graph_edges(100, np.array([1, 100], dtype=np.uint32), np.array([2, 3], dtype=np.uint32))   # endpoint >= n
    eng.set_graph_edges(100, np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=np.uint32))                    # no edge at all
    assert eng.info()["nnz"] == 0 and np.array_equal(eng.spmv(np.ones(100)), np.zeros(100))
    eng.set_graph_edges(64, np.array([0, 5, 5, 63], dtype=np.uint32), np.array([5, 0, 5, 1], dtype=np.uint32))
    y = eng.spmv(np.arange(64, dtype=np.float64))
    ref = np.zeros(64)
    ref[0] = 5.0; ref[5] = 0.0 + 5.0; ref[63] = 1.0; ref[1] = 63.0
    assert np.array_equal(y, ref)
    eng.close()


def test_sharded_hand_over_catches_an_asymmetric_pattern(pkg, oracle):
    """with the sharded hand-over a rank derives what it SENDS to each peer from its own rows through the matrix's
    symmetry, and until round 5 the hand-over compared only how MANY entries travel between every pair of ranks.  A caller's CSR
    that is not symmetric can keep every count and change the members: row i (rank A) references j' instead of j, both owned by
    rank B -- A still expects one value from B, B still packs one (x_j: its row j holds i), and A's product silently uses x_j for
    x_j'.  Now the lists are compared by content too (one order-dependent hash per pair): the sharded hand-over of such a CSR is
    an error at the group's first operation; the whole-graph hand-over, which sees every row, multiplies it correctly."""
    O = oracle
    n, world = 60000, 3
    rp, ci = O.gen_er(n, 150000, 17)
    opts = dict(propagation_blocking=1, hub_entries=512, overlap_exchange=1, sparse_exchange=1)
    grp = pkg.LocalGroup([0] * world, sharded_ingest=1, **opts)
    grp.set_graph_csr(rp, ci)
    assert grp.engines[0].info()["exchange_chunk0"] > 0          # two chunks, the second one sparse
    owner = np.empty(n, dtype=np.int64)
    for r, e in enumerate(grp.engines):
        owner[e.rank_row_sums()[1]] = r
    x = np.random.default_rng(2).random(n)
    assert np.allclose(grp.spmv(x), O.spmv(rp, ci, x), rtol=1e-13, atol=0)
    grp.close()
    # a row i of rank 0 with a column j of rank 1 that no other row of rank 0 references, and a j' of rank 1 that no row of rank 0
    # references at all; both of low degree (the sparse second chunk, not the dense first one)
    deg = np.diff(rp.astype(np.int64))
    rows = np.repeat(np.arange(n), deg)
    ref0 = np.bincount(ci[owner[rows] == 0], minlength=n)        # how many rows of rank 0 reference each column
    found = None
    for i in np.flatnonzero((owner == 0) & (deg >= 2)):
        for j in ci[rp[i]:rp[i + 1]]:
            if owner[j] == 1 and ref0[j] == 1 and deg[j] <= 4:
                cand = np.flatnonzero((owner == 1) & (ref0 == 0) & (deg > 0) & (deg <= 4))
                cand = cand[~np.isin(cand, ci[rp[i]:rp[i + 1]])]
                if len(cand):
                    found = (int(i), int(j), int(cand[0]))
                    break
        if found:
            break
    assert found, "the test graph has no such triple"
    i, j, j2 = found
    bad = ci.copy()
    row = bad[rp[i]:rp[i + 1]]
    row[row == j] = j2
    row.sort()                                                   # (a view: the row stays ascending, every degree stays what it was)
    y_ref = O.spmv(rp, bad, x)
    assert not np.array_equal(y_ref, O.spmv(rp, ci, x))
    whole = pkg.LocalGroup([0] * world, **opts)                  # every rank sees every row: the lists are right by themselves
    whole.set_graph_csr(rp, bad)
    assert np.allclose(whole.spmv(x), y_ref, rtol=1e-13, atol=0)
    whole.close()
    part = pkg.LocalGroup([0] * world, sharded_ingest=1, **opts)
    part.set_graph_csr(rp, bad)
    with pytest.raises(pkg.LzxError, match="OTHER ones"):
        part.spmv(x)
    with pytest.raises(pkg.LzxError):
        part.lanczos(np.ones(n), 5)
    part.set_graph_csr(rp, ci)                                   # the handles are still good
    assert np.allclose(part.spmv(x), O.spmv(rp, ci, x), rtol=1e-13, atol=0)
    part.close()

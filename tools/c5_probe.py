"""tools/c5_probe.py -- BASELINE C5 (R-MAT, 100 M vertices, 2 G draws -> ~3.9 G stored entries) as rank 0 of 8 sees it, on
ONE GPU: 8 handles wired as an in-process communicator, the whole graph generated and reshaped on rank 0 only (every
rank of a real run does the same on its own GPU), rank 0's local SpMV timed.  Shows that the 2^32-entry graph fits a
288 GB device without a distributed ingest, and what a rank's share of the compute costs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_pkg()
scale, n, draws = 27, 100_000_000, 2_000_000_000
args = [a for a in sys.argv[1:] if not a.startswith("--")]
if len(args) >= 3:
    scale, n, draws = int(args[0]), int(args[1]), int(args[2])
grp = pkg.LocalGroup([0] * 8)
e0 = grp.engines[0]
t = time.time()
e0.gen_rmat(scale, n, draws, 1234)
print(f"generated + reshaped in {time.time() - t:.1f} s", flush=True)
gi = e0.info()
print({k: gi[k] for k in ("n", "nnz", "max_degree", "rows_local", "nnz_local", "pb_entries", "pb_reduced_entries", "pb_values",
                          "active_vertices", "exchange_slice", "hub_entries")}, flush=True)
avg, mn = e0.bench_spmv(5)
byt = 4 * gi["nnz_local"] + 4 * (gi["rows_local"] + 1) + 8 * gi["n"] + 8 * gi["rows_local"]
print(f"rank 0 of 8: local SpMV avg {avg:.3f} ms min {mn:.3f} ms -> {byt / mn / 1e6:.0f} GB/s algorithmic", flush=True)
grp.close()

# The same graph on ONE handle in plain mode (a single rank's share of the blocked tables would exceed their 31-bit
# slots): checks ingest and SpMV beyond 2^32 / 2^31 entries through size-independent properties.
if "--check" in sys.argv:
    eng = pkg.Engine(0, propagation_blocking=0)
    t = time.time()
    eng.gen_rmat(scale, n, draws, 1234)
    gi1 = eng.info()
    print(f"one handle, plain mode: reshaped in {time.time() - t:.1f} s, nnz {gi1['nnz']}", flush=True)
    y = eng.spmv(np.ones(n))
    assert float(y.sum()) == float(gi1["nnz"]) and y.max() == gi1["max_degree"] and np.array_equal(y, np.rint(y)), "row sums"
    rng = np.random.default_rng(7)
    a, b = rng.random(n), rng.random(n)
    Aa, Ab = eng.spmv(a), eng.spmv(b)
    assert abs(a @ Ab - b @ Aa) <= 1e-12 * abs(a @ Ab), "symmetry"
    print(f"row sums = degrees (sum {int(y.sum())}, max {int(y.max())}), x'(Ay) = y'(Ax) to {abs(a @ Ab - b @ Aa) / abs(a @ Ab):.1e}: ok", flush=True)
    eng.close()

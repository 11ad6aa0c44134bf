"""tools/edge_count_probe.py -- VERDICT r4 item 7: BASELINE's configs name EDGES ("10 M-node / 200 M-edge R-MAT"), the generators
take DRAWS, and duplicates / self loops are dropped: 200 M draws leave 193.2 M distinct undirected edges.  Finds, by bisection on
the GPU generator (counter-based: draw i is the same whatever the total, so the count is monotone), the smallest number of draws
that reaches each workload's named edge count."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_pkg()
CASES = {"c2": ("rmat", 20, 1 << 20, 20_000_000), "c3": ("rmat", 24, 10_000_000, 200_000_000), "er": ("er", 0, 10_000_000, 100_000_000)}
for name in sys.argv[1:] or list(CASES):
    kind, scale, n, target = CASES[name]
    def edges(draws):
        e = pkg.Engine(0, propagation_blocking=0, placement_trials=0)
        (e.gen_er(n, draws, 1234) if kind == "er" else e.gen_rmat(scale, n, draws, 1234))
        m = e.info()["nnz"] // 2
        e.close()
        return m
    lo, hi = target, int(target * 1.2)
    assert edges(hi) >= target
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if edges(mid) >= target: hi = mid
        else: lo = mid
    print(f"{name}: {hi} draws -> {edges(hi)} distinct undirected edges (target {target}); {target} draws -> {edges(target)}", flush=True)

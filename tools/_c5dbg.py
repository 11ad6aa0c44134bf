import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_pkg()
scale, n, draws = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
e0 = pkg.Engine(0, propagation_blocking=0); e0.gen_rmat(scale, n, draws, 1234)
deg = e0.spmv(np.ones(n)); e0.close()
e1 = pkg.Engine(0); e1.gen_rmat(scale, n, draws, 1234)
y = e1.spmv(np.ones(n))
bad = np.nonzero(y != deg)[0]
gi = e1.info(); print({k_: gi[k_] for k_ in ("nnz", "pb_entries", "pb_values", "active_vertices")}); print("bad rows", len(bad), "sum diff", float((y - deg).sum()), flush=True)
order = np.argsort(-deg, kind="stable"); rank = np.empty(n, dtype=np.int64); rank[order] = np.arange(n)
br = np.sort(rank[bad]) if len(bad) else np.zeros(1, dtype=np.int64)
print("degree ranks of bad rows: min", br[:10], "max", br[-10:], flush=True)
d = (y - deg)[order[br[:20]]]
print("diffs of first bad ranks", d, "their degrees", deg[order[br[:20]]], flush=True)
h, _ = np.histogram(br, bins=20, range=(0, br.max() + 1)); print("hist of bad ranks", h, flush=True)

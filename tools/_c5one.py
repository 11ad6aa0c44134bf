import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_pkg()
scale, n, draws = 27, 100_000_000, 2_000_000_000
opts = dict(a.split('=') for a in sys.argv[1:])
opts = {k: int(v) for k, v in opts.items()}
eng = pkg.Engine(0, **opts)
t = time.time(); eng.gen_rmat(scale, n, draws, 1234); print(f"reshaped in {time.time()-t:.1f}s", flush=True)
gi = eng.info(); print(gi, flush=True)
y = eng.spmv(np.ones(n))
print("row sums", float(y.sum()) == float(gi["nnz"]), y.max() == gi["max_degree"], np.array_equal(y, np.rint(y)), flush=True)
rng = np.random.default_rng(7)
a, b = rng.random(n), rng.random(n)
Aa, Ab = eng.spmv(a), eng.spmv(b)
print("symmetry", abs(a @ Ab - b @ Aa) / abs(a @ Ab), flush=True)

import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_pkg()
scale, n, draws = 27, 100_000_000, 1_000_000_000
e0 = pkg.Engine(0, propagation_blocking=0); e0.gen_rmat(scale, n, draws, 1234)
deg = e0.spmv(np.ones(n)); e0.close()
order = np.argsort(-deg, kind="stable")
X = 32891136
for pm in (3 + 4 + 8, 3 + 8, 3):
    e1 = pkg.Engine(0, phase_mask=pm); e1.gen_rmat(scale, n, draws, 1234)
    y = e1.spmv(np.ones(n))
    ys = y[order]
    print("phase_mask", pm, "rows<X sum", ys[:X].sum(), "rows>=X sum", ys[X:].sum(), "row0", ys[0], "nonzero rows>=X", int((ys[X:] != 0).sum()),
          "first rows>=X", ys[X:X + 6], "rows just below X", ys[X - 6:X], flush=True)
    e1.close()
print("deg: rows<X", deg[order][:X].sum(), "rows>=X", deg[order][X:].sum(), flush=True)

"""tools/multout_probe.py [c2|c3] -- the device back-projection (N2: k_multout, ans = Q t on the resident basis) on a
bench graph, k = 50, timed from the host and meant to be run under rocprofv3 --kernel-trace --stats."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge

pkg = ge.load_pkg()
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
scale, n, draws = {"c2": (20, 1 << 20, 20_000_000), "c3": (24, 10_000_000, 200_000_000)}[name]
k = 50
eng = pkg.Engine(0)
eng.gen_rmat(scale, n, draws, 1234)
eng.lanczos(np.ones(n), k, want_q=False)
t = np.random.default_rng(0).random(k)
eng.multout(t)
t0 = time.perf_counter()
for _ in range(5):
    ans = eng.multout(t)
dt = (time.perf_counter() - t0) / 5
act = eng.info()["active_vertices"]
print(f"{name}: multout k={k} n={n}: {dt * 1e3:.2f} ms per call including the {8 * n / 1e6:.0f} MB download of the answer; "
      f"the kernel reads the columns of the {act} vertices that have an edge, {8 * act * k / 1e9:.2f} GB (the others: k scalars times q_0)")
eng.close()

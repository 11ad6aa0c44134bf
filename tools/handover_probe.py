"""tools/handover_probe.py -- where the milliseconds around the Lanczos loop go: lzx_lanczos_prepare_f64 with
and without round 5's look at x0 (test shape start_vector_scan), the first call (the basis is sized) and a repeated one, the loop,
the download of alpha / beta.  C3 by default."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, __graft_entry__ as ge
from bench import C2_DRAWS, C3_DRAWS
pkg = ge.load_pkg()
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
scale, n, draws, k = {"c3": (24, 10_000_000, C3_DRAWS, 50), "c2": (20, 1 << 20, C2_DRAWS, 50)}[wl]
for scan in (1, 0):
    e = pkg.Engine(0, start_vector_scan=scan)
    e.gen_rmat(scale, n, draws, 1234)
    for name, x0 in (("ones", np.ones(n)), ("random", np.random.default_rng(1).random(n))):
        for rep in range(3):
            t0 = time.perf_counter(); e.lanczos_prepare(x0, k); e.sync(); t1 = time.perf_counter()
            st = e.lanczos_run(); e.sync(); t2 = time.perf_counter()
            a, b, _ = e.lanczos_fetch(k); t3 = time.perf_counter()
            print(f"{wl} scan={scan} x0={name:6s} call {rep}: prepare {1e3 * (t1 - t0):7.2f} ms  loop {1e3 * (t2 - t1):7.2f} ms  fetch alpha/beta {1e3 * (t3 - t2):6.2f} ms"
                  f"  -> {k / (t3 - t0):7.1f} iter/s with the hand-over, {k / (t2 - t1):7.1f} without", flush=True)
    e.close()

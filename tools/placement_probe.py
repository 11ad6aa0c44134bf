"""tools/placement_probe.py <workload> [engines] -- option placement_trials at work: engines built one after the other in one
process (each lands where the driver puts it), with the candidates' SpMV times at the hand-over, the one kept, and the SpMV
time afterwards; beside an engine built with placement_trials = 0."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_pkg()
work = sys.argv[1] if len(sys.argv) > 1 else "er"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
WORK = {"c3": (24, 10_000_000, 200_000_000), "er": (0, 10_000_000, 100_000_000), "c2": (20, 1 << 20, 20_000_000),
        "mid": (22, 4_000_000, 70_000_000)}
scale, n, draws = WORK[work]
import time
for i in range(reps):
    for trials in [int(a) for a in os.environ.get("PLACEMENT_TRIALS", "0,7").split(",")]:
        e = pkg.Engine(0, placement_trials=trials)
        t = time.time()
        if scale == 0:
            e.gen_er(n, draws, 1234)
        else:
            e.gen_rmat(scale, n, draws, 1234)
        t = time.time() - t
        tried = e.shape("placement_tried")
        us = [e.shape(f"placement_us_{j}") for j in range(tried)]
        mn = min(e.bench_spmv(10)[1] for _ in range(3))
        print(f"{work} engine {i} placement_trials={trials}: built in {t:.2f} s; candidates {us} us, kept {e.shape('placement_kept')}; "
              f"spmv min {mn:.4f} ms", flush=True)
        e.close()

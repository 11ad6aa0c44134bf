"""tools/state_probe2.py <workload> -- do the two "process states" of the SpMV (profiles/NOTES.md: consecutive processes alternate
between a faster and a slower one, C3 1.5-5 %, ER 4-15 %) also separate ENGINES INSIDE ONE PROCESS?  If the state belongs to the
physical memory behind a graph's tables, two engines alive at once hold different memory and may differ; an engine rebuilt after
the first is closed gets the freed pages back.  Sequence: A; B (A alive); close A; C; close B; D; ... -- every SpMV time printed."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_pkg()
work = sys.argv[1] if len(sys.argv) > 1 else "er"
WORK = {"c3": (24, 10_000_000, 200_000_000), "er": (0, 10_000_000, 100_000_000), "c2": (20, 1 << 20, 20_000_000)}
scale, n, draws = WORK[work]


def build():
    e = pkg.Engine(0)
    if scale == 0:
        e.gen_er(n, draws, 1234)
    else:
        e.gen_rmat(scale, n, draws, 1234)
    return e


def t(e):
    return min(e.bench_spmv(10)[1] for _ in range(3)), e.bench_spmv(20)[0]


alive = []
log = []
for step in range(8):
    e = build()
    alive.append(e)
    mn, avg = t(e)
    log.append((step, len(alive), mn, avg))
    print(f"{work}: engine {step} ({len(alive)} alive): spmv min {mn:.4f} avg {avg:.4f} ms", flush=True)
    if len(alive) == 2:
        # the older one again, now that a second graph lives beside it
        mn0, avg0 = t(alive[0])
        print(f"{work}:   engine {step - 1} again: min {mn0:.4f} avg {avg0:.4f} ms", flush=True)
        alive.pop(0).close()
for e in alive:
    e.close()

"""tools/unit_sweep.py -- entries per scatter unit (shape pb_unit; every unit restages its 144 KiB column band) on C3, each value in two
engines, alternating with the default."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from bench import C3_DRAWS
pkg = ge.load_pkg()
for unit in (-1, 98304, -1, 163840, -1, 196608, -1, 262144, -1, 65536):
    e = pkg.Engine(0, placement_trials=3, **({} if unit < 0 else dict(pb_unit=unit)))
    e.gen_rmat(24, 10_000_000, C3_DRAWS, 1234)
    avg, mn = e.bench_spmv(40)
    print(f"c3 pb_unit={unit}: SpMV avg {avg:.4f} min {mn:.4f} ms", flush=True)
    e.close()

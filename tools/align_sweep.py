"""tools/align_sweep.py -- value-stream padding: every (row band, column band) run starts on a 64-byte boundary (8 values, shape
pb_run_align).  On the uniform graph a run holds ~18 values, so the padding is a fifth of the stream; 4 values (32 bytes: what a plain
quad's two 16-byte stores need) would halve it.  ER, C3, C2, each configuration in two engines."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from bench import C2_DRAWS, C3_DRAWS, ER_DRAWS
pkg = ge.load_pkg()
for name, gen in (("er", lambda e: e.gen_er(10_000_000, ER_DRAWS, 1234)), ("c3", lambda e: e.gen_rmat(24, 10_000_000, C3_DRAWS, 1234)),
                  ("c2", lambda e: e.gen_rmat(20, 1 << 20, C2_DRAWS, 1234))):
    for align in (8, 4, 8, 4, 8, 4):
        e = pkg.Engine(0, placement_trials=7, pb_run_align=align)
        gen(e)
        gi = e.info()
        avg, mn = e.bench_spmv(30)
        print(f"{name} pb_run_align={align}: values={gi['pb_values']} (blocked entries {gi['pb_entries']}, reduced {gi['pb_reduced_entries']}) | SpMV avg {avg:.4f} min {mn:.4f} ms", flush=True)
        e.close()

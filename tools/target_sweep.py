"""tools/target_sweep.py -- values per gather item (shape pb_target) on small problems: C2 on one rank and rank 0's share of C3 at 2, 4, 8
ranks (alone on the GPU).  Each configuration twice (two engines in a row)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from bench import C2_DRAWS, C3_DRAWS
pkg = ge.load_pkg()
targets = [-1, 12288, 16384, 24576, 32768, 65536]
for t in targets:
    for rep in range(2):
        e = pkg.Engine(0, placement_trials=0, **({} if t < 0 else dict(pb_target=t)))
        e.gen_rmat(20, 1 << 20, C2_DRAWS, 1234)
        avg, mn = e.bench_spmv(50)
        print(f"c2 world=1 pb_target={t}: finish launch {e.shape('finish_launched')} items {e.shape('gather_items_dealt')} | SpMV avg {avg:.4f} min {mn:.4f} ms", flush=True)
        e.close()
src = pkg.Engine(0, propagation_blocking=0, placement_trials=0)
src.gen_rmat(24, 10_000_000, C3_DRAWS, 1234)
rp, ci = src.get_graph_csr()
src.close()
for world in (2, 4, 8):
    for t in targets:
        for rep in range(2):
            grp = pkg.LocalGroup([0] * world, placement_trials=0, **({} if t < 0 else dict(pb_target=t)))
            e0 = grp.engines[0]
            e0.set_graph_csr(rp, ci)
            avg, mn = e0.bench_spmv(30)
            print(f"c3 world={world} pb_target={t}: finish launch {e0.shape('finish_launched')} items {e0.shape('gather_items_dealt')} | local SpMV avg {avg:.4f} min {mn:.4f} ms", flush=True)
            grp.close()

"""tools/share_sweep.py [world] -- rank 0's local SpMV of a `world`-rank run of C3 (alone on the GPU, lzx_bench_spmv with marks between
the kernels) for a few shapes of the gather pass's small-band groups: is the share's latency-bound gather pass
tunable?  Every configuration in two consecutive processes would be the parity-controlled form; this is the quick look."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from bench import C3_DRAWS
pkg = ge.load_pkg()
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
src = pkg.Engine(0, propagation_blocking=0, placement_trials=0)
src.gen_rmat(24, 10_000_000, C3_DRAWS, 1234)
rp, ci = src.get_graph_csr()
src.close()
for opts in (dict(), dict(pb_group_force=2), dict(pb_group_force=4), dict(pb_group_force=8), dict(pb_group=32768), dict(pb_group=8192),
             dict(pb_target=32768), dict(pb_target=131072), dict(pb_gather_nt=1)):
    for rep in range(2):
        grp = pkg.LocalGroup([0] * world, placement_trials=0, **opts)
        e0 = grp.engines[0]
        e0.set_graph_csr(rp, ci)
        gi = e0.info()
        avg, mn = e0.bench_spmv(30)
        print(f"world={world} {opts}: values={gi['pb_values']} items dealt {e0.shape('gather_items_dealt')} workgroups {e0.shape('gather_workgroups')} | local SpMV avg {avg:.4f} min {mn:.4f} ms", flush=True)
        grp.close()

"""tools/rank_share.py <world> [c3|c5] -- the program behind `rocprofv3 ... --` for VERDICT r4 item 4: rank 0's share of a
`world`-rank run, alone on the GPU (an in-process group whose other handles hold no graph; lzx_bench_spmv does no exchange), 20
local SpMVs.  Prints one line of table facts; the profiler around it gives kernel times and FETCH_SIZE / WRITE_SIZE per launch
(tools/profile_rank.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from bench import C3_DRAWS
pkg = ge.load_pkg()
world = int(sys.argv[1])
wl = sys.argv[2] if len(sys.argv) > 2 else "c3"
scale, n, draws = {"c3": (24, 10_000_000, C3_DRAWS), "c5": (27, 100_000_000, 2_000_000_000)}[wl]
opts = dict(sharded_ingest=1) if wl == "c5" else {}
if world == 1:
    e0 = pkg.Engine(0, **opts)
    grp = None
else:
    grp = pkg.LocalGroup([0] * world, **opts)
    e0 = grp.engines[0]
e0.gen_rmat(scale, n, draws, 1234)          # only rank 0 needs its share: the other handles stay empty
gi = e0.info()
avg, mn = e0.bench_spmv(20)
print(f"{wl} world={world} rank 0: rows_local={gi['rows_local']} nnz_local={gi['nnz_local']} pb_entries={gi['pb_entries']} reduced={gi['pb_reduced_entries']} "
      f"values={gi['pb_values']} hub={gi['hub_entries']} chunk0={gi['exchange_chunk0']} recv={gi['exchange_recv']} | local SpMV avg {avg:.4f} ms min {mn:.4f} ms "
      f"| algorithmic bytes of the share {4 * gi['nnz_local'] + 4 * (gi['rows_local'] + 1) + 8 * gi['n'] + 8 * gi['rows_local']}", flush=True)
(grp or e0).close()

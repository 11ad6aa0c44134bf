"""tools/rank_probe.py -- per-rank SpMV time of the row-partitioned layout, measured on ONE GPU: `world` handles wired as
an in-process communicator, the graph reshaped on each, then only rank 0's local SpMV is timed (lzx_bench_spmv does no
exchange).  Shows how the compute part scales with the rank count before any xGMI time is added."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_pkg()
from bench import C3_DRAWS
scale, n, draws = 24, 10_000_000, C3_DRAWS
src = pkg.Engine(0, propagation_blocking=0)
src.gen_rmat(scale, n, draws, 1234)
rp, ci = src.get_graph_csr()
src.close()
for world in [int(w) for w in os.environ.get("RANK_PROBE_WORLDS", "1,2,4,8").split(",")]:
    for opts in ([dict(), dict(pb_column_band=18432), dict(), dict(pb_column_band=18432)] if os.environ.get("RANK_PROBE_WIDE") else [dict(propagation_blocking=0), dict(sparse_exchange=0), dict()]):
        if world == 1:
            eng = pkg.Engine(0, **opts)
            eng.set_graph_csr(rp, ci)
            e0 = eng
            grp = None
        else:
            grp = pkg.LocalGroup([0] * world, **opts)
            # only rank 0 needs its share for the timing; the others stay empty handles
            grp.engines[0].set_graph_csr(rp, ci)
            e0 = grp.engines[0]
        gi = e0.info()
        avg, mn = e0.bench_spmv(10)
        print(f"world={world} {opts} rows_local={gi['rows_local']} nnz_local={gi['nnz_local']} pb={gi['pb_entries']} reduced={gi['pb_reduced_entries']} values={gi['pb_values']} "
              f"exchange: slice {gi['exchange_slice']} chunk0 {gi['exchange_chunk0']} received per iteration {gi['exchange_recv']} doubles = {8 * gi['exchange_recv'] / 1e6:.1f} MB | "
              f"spmv min {mn:.4f} ms (x{world} = {mn * world:.3f})", flush=True)
        (grp or e0).close()

"""tools/local8_check.py -- C3 on 8 in-process ranks sharing ONE GPU: a functional rehearsal of the 8-rank layout at
full size (both exchange modes), not a timing of xGMI."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_pkg()
n = 10_000_000
src = pkg.Engine(0, propagation_blocking=0)
src.gen_rmat(24, n, 200_000_000, 1234)
rp, ci = src.get_graph_csr()
a1, b1, _, _, st1 = src.lanczos(np.ones(n), 6, want_q=False)
src.close()
print("single", a1[:3], b1[:2], flush=True)
for overlap in (1, 0):
    grp = pkg.LocalGroup([0] * 8, overlap_exchange=overlap)
    t = time.time(); grp.set_graph_csr(rp, ci); ts = time.time() - t
    gi = grp.engines[3].info()
    a, b, _, _, st = grp.lanczos(np.ones(n), 6, want_q=False)
    print(f"overlap={overlap} setup {ts:.1f}s pb={gi['pb_entries']} xs={gi['exchange_slice']} active={gi['active_vertices']} "
          f"loop {st['loop_ms']:.2f} ms spmv {st['spmv_ms']:.2f} comm {st['comm_ms']:.2f} | rel diff alpha {np.abs(a - a1).max() / np.abs(a1).max():.2e} "
          f"beta {np.abs(b - b1).max() / np.abs(b1).max():.2e}", flush=True)
    grp.close()

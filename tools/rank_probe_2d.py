"""tools/rank_probe_2d.py -- what would a 2-D (row group x column group) partition of the SpMV buy at 8 ranks?  (VERDICT round 3,
next 4: decide the 8-GPU layout on evidence.)

Rank (i, j) of a Pr x Pc grid multiplies the block A[R_i, C_j] (R_i = degree ranks r with r % Pr == i, C_j = ranks with
r % Pc == j): it needs only x_j (n / Pc entries: an all-gather inside its column group) and leaves PARTIAL sums for the
n / Pr rows of R_i, which a reduce-scatter over the Pc ranks of its row group turns into the n / P rows it owns.

The local SpMV of block (0, 0) is timed on ONE GPU by handing the block to an ordinary one-rank engine as a general CSR
pattern (the engine multiplies any pattern; tests/test_gpu_parity.py::test_general_csr_patterns): rows = R_0 in descending
order of their degree INSIDE the block, column ids = popularity rank among C_0.  The engine's own degree ranking of the rows
is then the identity, so its x layout holds C_0 compactly in popularity order -- staged columns = the 16 Ki most popular
columns of C_0, column bands of 16 Ki consecutive ones -- exactly the tables rank (0, 0) would build.  Values are irrelevant
to the timing.  Beside it: rank 0 of 8 of the 1-D layout in the same process (tools/rank_probe.py's figure), the bytes each
layout receives per iteration, and what xGMI's point-to-point links make of them (7 links per GPU; a reduce-scatter among
the Pc ranks of a row group can only use the Pc - 1 links between them)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_pkg()
scale, n, draws = 24, 10_000_000, 200_000_000
LINK_GBS = float(os.environ.get("XGMI_LINK_GBS", "55"))   # one xGMI link, one direction, as DESIGN section 5 assumes (7 x ~64 peak)

src = pkg.Engine(0, propagation_blocking=0)
src.gen_rmat(scale, n, draws, 1234)
rp, ci = src.get_graph_csr()
src.close()
deg = np.diff(rp.astype(np.int64))
n_active = int((deg > 0).sum())
t0 = time.time()
order = np.argsort(-deg, kind="stable")          # degree rank -> vertex (ties by id, as the engine's stable radix sort)
rank_of = np.empty(n, dtype=np.int64)
rank_of[order] = np.arange(n)
row_of_entry = np.repeat(np.arange(n, dtype=np.int64), deg)
print(f"[2d] ranks ready in {time.time() - t0:.1f} s; n_active {n_active}", flush=True)


def block(pr, pc, i=0, j=0):
    """CSR of A[R_i, C_j] in the probe's numbering (rows sorted by block degree, columns = popularity rank in C_j)."""
    rr = rank_of[row_of_entry]
    cr = rank_of[ci]
    keep = (rr % pr == i) & (cr % pc == j)
    rows = rr[keep] // pr                      # row id inside R_i (by true degree rank)
    cols = (cr[keep] // pc).astype(np.uint32)  # popularity rank inside C_j
    nrow = (n - i + pr - 1) // pr
    bdeg = np.bincount(rows, minlength=nrow)
    rorder = np.argsort(-bdeg, kind="stable")  # rows by degree inside the block
    newid = np.empty(nrow, dtype=np.int64)
    newid[rorder] = np.arange(nrow)
    rows = newid[rows]
    o = np.lexsort((cols, rows))
    rows, cols = rows[o], cols[o]
    size = max(nrow, (n - j + pc - 1) // pc)
    brp = np.zeros(size + 1, dtype=np.uint64)
    brp[1:nrow + 1] = np.cumsum(bdeg[rorder])
    brp[nrow + 1:] = brp[nrow]
    return brp, cols, int(keep.sum())


def time_block(pr, pc):
    t = time.time()
    brp, bci, cnt = block(pr, pc)
    eng = pkg.Engine(0)
    eng.set_graph_csr(brp, bci)
    gi = eng.info()
    best = min(eng.bench_spmv(10)[1] for _ in range(3))
    eng.close()
    rows_i, cols_j = n_active / pr, n_active / pc       # active rows of R_i, active columns of C_j (rows are dealt round-robin)
    P = pr * pc
    own = n_active / P
    recv_x = (cols_j - own) * 8e-6                       # MB: the rest of x_j, from the pr - 1 ... other owners inside C_j
    recv_y = (pc - 1) * own * 8e-6                       # MB: partial sums of the own rows from the other pc - 1 ranks of the row group
    # point-to-point links: x_j comes from pr - 1 peers (one link each), the partial sums from pc - 1 peers (one link each)
    t_x = (own * 8e-6 / LINK_GBS) if pr > 1 else 0.0          # ms (MB / (GB/s)): every peer sends its own slice over its own link
    t_y = (own * 8e-6 / LINK_GBS) if pc > 1 else 0.0          # ms: every peer of the row group sends this rank's rows over its own link
    print(f"[2d] {pr} x {pc}: block (0,0) {cnt} entries, engine: rows_local {gi['rows_local']} pb {gi['pb_entries']} values {gi['pb_values']} staged {gi['hub_entries']} | "
          f"local SpMV min {best:.4f} ms | per iteration: x_j all-gather {recv_x:.1f} MB received ({t_x * 1e3:.0f} us over {max(pr - 1, 0)} links), partial-sum reduce-scatter "
          f"{recv_y:.1f} MB received ({t_y * 1e3:.0f} us over {max(pc - 1, 0)} links, AFTER the SpMV: exposed) | built in {time.time() - t:.0f} s", flush=True)
    return best, t_x, t_y


def time_1d(world, **opts):
    grp = pkg.LocalGroup([0] * world, **opts)
    grp.engines[0].set_graph_csr(rp, ci)
    gi = grp.engines[0].info()
    best = min(grp.engines[0].bench_spmv(10)[1] for _ in range(3))
    grp.close()
    print(f"[1d] {world} x 1 {opts}: rank 0 {gi['nnz_local']} entries, values {gi['pb_values']} | local SpMV min {best:.4f} ms | per iteration "
          f"{8e-6 * gi['exchange_recv']:.1f} MB received over 7 links ({8e-6 * gi['exchange_recv'] / 7 / LINK_GBS * 1e3:.0f} us), overlapping the SpMV but for chunk 0 "
          f"({8e-6 * 7 * gi['exchange_chunk0']:.1f} MB)", flush=True)
    return best


one = pkg.Engine(0)
one.set_graph_csr(rp, ci)
t1 = min(one.bench_spmv(10)[1] for _ in range(3))
one.close()
print(f"[1d] 1 x 1: local SpMV min {t1:.4f} ms", flush=True)
t8 = time_1d(8)
res = {(pr, pc): time_block(pr, pc) for pr, pc in ((2, 4), (4, 2), (8, 1))}
FIX = 0.010 + 0.025        # ms: vector kernel + two-double all-reduce (DESIGN section 5's assumptions)
print(f"[model] per iteration = local SpMV + {FIX * 1e3:.0f} us (vector kernel, scalar all-reduce) + exposed exchange; one GPU: {t1 + 0.037:.3f} ms", flush=True)
print(f"[model] 8 x 1 (built): {t8:.3f} + {FIX:.3f} + chunk 0 0.035 = {t8 + FIX + 0.035:.3f} ms -> {(t1 + 0.037) / (t8 + FIX + 0.035):.2f} x", flush=True)
for (pr, pc), (ts, tx, ty) in res.items():
    if pc == 1:
        continue
    it = ts + FIX + tx + ty
    print(f"[model] {pr} x {pc}: {ts:.3f} + {FIX:.3f} + x_j gather {tx:.3f} (before the SpMV) + reduce-scatter {ty:.3f} (after it) = {it:.3f} ms -> {(t1 + 0.037) / it:.2f} x; "
          f"with the x_j gather hidden entirely: {(t1 + 0.037) / (it - tx):.2f} x", flush=True)

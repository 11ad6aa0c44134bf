"""tools/pair_stats.py -- how many distinct (row, column band) pairs do the blocked entries of C3 form?  (Upper bound on
what a scatter-side pre-reduction per pair could save.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_pkg()
n = 10_000_000
e = pkg.Engine(0, propagation_blocking=0); e.gen_rmat(24, n, 200_000_000, 1234); rp, ci = e.get_graph_csr(); e.close()
deg = np.diff(rp.astype(np.int64))
order = np.argsort(-deg, kind="stable")
rank = np.empty(n, dtype=np.int64); rank[order] = np.arange(n)
H, CB = 16384, 16384
rows = np.repeat(np.arange(n, dtype=np.int64), deg)
cr = rank[ci.astype(np.int64)]
keep = cr >= H
band = (cr[keep] - H) // CB
pairs = rows[keep] * 1024 + band
total = int(keep.sum())
u = np.unique(pairs)
print("blocked entries", total, "distinct (row, band) pairs", len(u), "ratio", len(u) / total)
# by row degree class
d_of = deg[rows[keep]]
for lo, hi in ((0, 128), (128, 1024), (1024, 8192), (8192, 10**9)):
    m = (d_of >= lo) & (d_of < hi)
    uu = np.unique(pairs[m])
    print(f"  rows with degree in [{lo},{hi}): entries {int(m.sum())} pairs {len(uu)} ratio {len(uu) / max(1, int(m.sum())):.3f}")
# histogram of entries per pair, and the padding of a sliced-ELL over the pairs of each band taken in row (degree-rank) order
pk, cnt = np.unique(rank[rows[keep]] + band * (1 << 24), return_counts=True)   # sorted by (band, row rank)
for lo, hi in ((1, 2), (2, 3), (3, 5), (5, 9), (9, 17), (17, 33), (33, 65), (65, 257), (257, 10**9)):
    m = (cnt >= lo) & (cnt < hi)
    print(f"  pairs with {lo}..{hi - 1} entries: {int(m.sum())} pairs ({m.sum() / len(cnt):.3f}), entries {int(cnt[m].sum())} ({cnt[m].sum() / total:.3f})")
for S in (16, 32, 64):
    pad = (-len(cnt)) % S
    c2 = np.concatenate([cnt, np.zeros(pad, dtype=cnt.dtype)]).reshape(-1, S)
    w = c2.max(axis=1)
    w4 = (w + 3) // 4 * 4
    print(f"  slices of {S} pairs in (band, row) order: padded entries {int(w.sum()) * S} ({w.sum() * S / total:.2f}x), widths rounded to 4: {w4.sum() * S / total:.2f}x")
    capped = np.minimum(c2, 32)
    wc = capped.max(axis=1)
    print(f"     with pairs cut at 32 entries: {wc.sum() * S / total:.2f}x (+ {int(np.maximum(cnt - 32, 0).sum())} entries in overflow items)")

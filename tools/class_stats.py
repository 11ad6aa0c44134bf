"""tools/class_stats.py -- CPU only (the oracle's generator): where the entries of the C3 graph lie by degree class of row and
column, how many (row, column band) pairs -- values that have to cross the two passes of the blocked SpMV -- each block
forms, and the run-length / padding figures of DESIGN.md 3.1 (h, i).  About 4 minutes and 8 GB of host memory."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O

n, H, CB = 10_000_000, 16384, 16384
ro, ci = O.gen_rmat(24, n, 200_000_000, 1234)
ro = ro.astype(np.int64)
deg = np.diff(ro)
order = np.argsort(-deg, kind="stable")
rank = np.empty(n, dtype=np.int64)
rank[order] = np.arange(n)
rows = np.repeat(np.arange(n, dtype=np.int32), deg)
rr, cr = rank[rows].astype(np.int32), rank[ci].astype(np.int32)
del rows
n_act, n_m = int((deg > 0).sum()), int((deg >= 128).sum())
print(f"n {n}  stored entries {len(ci)}  vertices with an edge {n_act}  degree >= 128: {n_m}  staged columns T: {H}")


def cls(r):
    return np.where(r < H, 0, np.where(r < n_m, 1, 2))


a, b = cls(rr), cls(cr)
print("entries by class of row x class of column (T = staged hubs, M = degree >= 128, L = the rest), millions:")
for i, ni in enumerate("TML"):
    print("   rows", ni, " ".join(f"{nj}: {((a == i) & (b == j)).sum() / 1e6:7.1f}" for j, nj in enumerate("TML")))
keep = cr >= H
r, band = rr[keep].astype(np.int64), (cr[keep] - H) // CB
pairs = np.unique(r * 512 + band)
prow = pairs >> 9
pc = cls(prow)
print(f"blocked entries {int(keep.sum())}, distinct (row, column band) pairs {len(pairs)}: rows T {int((pc == 0).sum())}, M {int((pc == 1).sum())}, L {int((pc == 2).sum())}")
for i, ni in enumerate("TML"):
    m = a[keep] == i
    for j, nj in ((1, "M"), (2, "L")):
        mm = m & (b[keep] == j)
        u = len(np.unique(r[mm] * 512 + band[mm]))
        print(f"   rows {ni} x columns {nj}: entries {int(mm.sum()) / 1e6:6.1f} M  pairs {u / 1e6:6.1f} M  ({mm.sum() / max(u, 1):.1f} entries per pair)")
# values per 1024-row band of the gather pass
cnt = np.bincount(prow // 1024)
for lo, hi in ((1, 4096), (4096, 16384), (16384, 65536), (65536, 1 << 30)):
    m = (cnt >= lo) & (cnt < hi)
    print(f"row bands with {lo}..{hi} pairs: {int(m.sum())} bands, {cnt[m].sum() / len(pairs):.3f} of the pairs")
# staged-only slices: width by 64 rows in (degree, staged count) order
s = np.bincount(rr[cr < H], minlength=n)
degr = np.bincount(rr, minlength=n)
idx = np.arange(H, n_act)
for name, o in (("degree, then id", idx), ("degree, then staged count", idx[np.lexsort((-s[idx], -degr[idx]))])):
    body = s[o][s[o] <= 256]
    body = np.concatenate([body, np.zeros((-len(body)) % 64, dtype=body.dtype)]).reshape(-1, 64)
    w = (body.max(axis=1) + 3) // 4
    print(f"staged-only slices, rows ranked by {name}: {len(w)} slices, {int(w.sum()) * 256} code slots for {int(body.sum())} entries; "
          f"slices of 0 / 1 / 2 packets: {int((w == 0).sum())} / {int((w == 1).sum())} / {int((w == 2).sum())}")

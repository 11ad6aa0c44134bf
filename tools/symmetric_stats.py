"""tools/symmetric_stats.py -- CPU only (the oracle's generator).  Prices VERDICT r4's item 1: stream only one orientation
of every edge (A = A^T), keep a y tile next to the x tile of the column band in LDS, add the transposed contribution
y_B[j] += x_i there, and let the (row, band) partial sums cross the passes as today.

What the proposal leaves open is where x_i comes from: the scatter unit of band B holds x_B, not x of the rows that stream
past it.  This tool counts, on the real C3 graph (or --workload er / c2), for the built layout and for the symmetric one:

  * code entries streamed,
  * (row, column band) pairs = fp64 values that cross the passes,
  * for the symmetric layout: the x_i fetches (one per pair), how many 128-byte lines of x they touch (the unit's rows are
    a sparse, sorted subset of the vertex order: a fetched line that holds one needed double still costs a fabric
    transaction), and the same if x_i crossed as a third sequential stream (8 B written by a pre-pass + 8 B read),

and prices each with the byte costs DESIGN.md uses (2 B per code, 8 B written + 8 B read + 2 B slot per crossing value).
About 6 minutes and 12 GB of host memory for C3."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3", choices=["c3", "c2", "er"])
args = ap.parse_args()
if args.workload == "c3":
    n = 10_000_000
    ro, ci = O.gen_rmat(24, n, 200_000_000, 1234)
elif args.workload == "c2":
    n = 1 << 20
    ro, ci = O.gen_rmat(20, n, 20_000_000, 1234)
else:
    n = 10_000_000
    ro, ci = O.gen_er(n, 100_000_000, 1234)
ro = ro.astype(np.int64)
deg = np.diff(ro)
order = np.argsort(-deg, kind="stable")
rank = np.empty(n, dtype=np.int64)
rank[order] = np.arange(n)
rows = np.repeat(np.arange(n, dtype=np.int32), deg)
rr, cr = rank[rows].astype(np.int64), rank[ci].astype(np.int64)
del rows, ci
nnz = len(rr)
n_act, n_m = int((deg > 0).sum()), int((deg >= 128).sum())
print(f"workload {args.workload}: n {n}  stored entries {nnz}  vertices with an edge {n_act}  degree >= 128: {n_m}")
GB = 1e9


def npairs(r, band, nb):
    return len(np.unique(r * nb + band))


def price(tag, H, CB, keep, fetch_x):
    """keep: mask of the entries streamed.  Entries whose column is staged (rank < H) cross nothing."""
    r, c = rr[keep], cr[keep]
    st = c < H
    e_st, e_bl = int(st.sum()), int((~st).sum())
    nb = (n - H) // CB + 2
    rb, band = r[~st], (c[~st] - H) // CB
    key = np.unique(rb * nb + band)
    P = len(key)
    codes = 2.0 * (e_st + e_bl)
    values = 18.0 * P
    line = f"{tag:58s} codes {e_st / 1e6:6.1f} staged + {e_bl / 1e6:6.1f} blocked M = {codes / GB:5.2f} GB; pairs {P / 1e6:6.1f} M = {values / GB:5.2f} GB"
    extra = 0.0
    if fetch_x:
        # one x_i per pair of the blocked part and one per (row, staged tile) of the staged part
        prow, pband = key // nb, key % nb
        lines = len(np.unique((prow // 16) * nb + pband))
        srow = np.unique(r[st])
        slines = len(np.unique(srow // 16))
        dense = 8.0 * (P + len(srow))
        gathered = 128.0 * (lines + slines)
        third = 16.0 * P + 8.0 * len(srow)
        yflush = 16.0 * n_act                     # every band owner writes its y tile once and somebody adds it: 8 B + 8 B
        line += (f"; x_i: {P / 1e6:.1f} M fetches touching {lines / 1e6:.1f} M lines ({P / max(lines, 1):.2f} doubles per 128 B line) = "
                 f"{gathered / GB:.2f} GB as gathers, {third / GB:.2f} GB as a third stream, {dense / GB:.2f} GB if they were dense; y tiles {yflush / GB:.2f} GB")
        extra = min(gathered, third) + yflush
    tot = codes + values + extra
    print(line + f"  ==> {tot / GB:5.2f} GB")
    return tot


print("\n-- whole matrix, priced as 2 B per code + 18 B per crossing value (+ x_i and y tiles for the symmetric forms) --")
allm = np.ones(nnz, dtype=bool)
base = price("built: both orientations, H = CB = 18 Ki", 18432, 18432, allm, False)
price("built layout with 9 Ki bands and 9 Ki staged (what the second tile costs by itself)", 9216, 9216, allm, False)
lo = cr < rr                                   # column = the higher-degree end
up = cr > rr                                   # column = the lower-degree end
for tag, m in (("symmetric, column = higher-degree end (j < i)", lo), ("symmetric, column = lower-degree end (j > i)", up)):
    for H, CB in ((9216, 9216), (18432, 18432)):
        t = price(f"{tag}, H = CB = {H // 1024} Ki" + (" (LDS would need 288 KiB)" if H > 9216 else ""), H, CB, m, True)
        print(f"      against the built layout: {t / base:.2f} x")

print("\n-- by block (T = the 18 Ki staged hubs, M = the other vertices of degree >= 128, L = the rest) --")
H = 18432


def cls(r):
    return np.where(r < H, 0, np.where(r < n_m, 1, 2))


a, b = cls(rr), cls(cr)
names = "TML"
for i in range(3):
    for j in range(i, 3):
        blk = ((a == i) & (b == j)) | ((a == j) & (b == i))
        e_all = int(blk.sum())
        if e_all == 0:
            continue
        # built: both orientations, 18 Ki bands
        m_now = blk & (cr >= H)
        nb18 = (n - H) // 18432 + 2
        p_now = npairs(rr[m_now], (cr[m_now] - H) // 18432, nb18)
        now = 2.0 * e_all + 18.0 * p_now
        out = f"block {names[i]}-{names[j]}: {e_all / 1e6:6.1f} M entries, built {p_now / 1e6:5.1f} M pairs = {now / GB:5.2f} GB"
        for tag, half in (("col = high end", blk & lo), ("col = low end", blk & up)):
            m_s = half & (cr >= 9216)
            nb9 = (n - 9216) // 9216 + 2
            r_s, band_s = rr[m_s], (cr[m_s] - 9216) // 9216
            key = np.unique(r_s * nb9 + band_s)
            lines = len(np.unique((key // nb9 // 16) * nb9 + key % nb9))
            sym = 2.0 * int(half.sum()) + 18.0 * len(key) + min(128.0 * lines, 16.0 * len(key))
            out += f" | sym, {tag}: {len(key) / 1e6:5.1f} M pairs, x_i {len(key) / max(lines, 1):.2f} per line = {sym / GB:5.2f} GB"
        print(out)

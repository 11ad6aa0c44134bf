"""tools/state_probe.py <workload> -- one process: the block -> XCD placement of a few small launches (debug library,
lzx_dbg_xcc_map) beside the SpMV time of this process.  Run several times in a row (a scratch job script): does the process
state of the SpMV (NOTES 3.1 i: 3-5 % on C3, 10-20 % on ER, for a process's life) follow where workgroup 0 lands?"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_pkg()
work = sys.argv[1] if len(sys.argv) > 1 else "er"
WORK = {"c3": (24, 10_000_000, 200_000_000), "er": (0, 10_000_000, 100_000_000), "c2": (20, 1 << 20, 20_000_000)}
scale, n, draws = WORK[work]
eng = pkg.Engine(0, phase_mask=3)     # debug library
L = eng.L
L.lzx_dbg_xcc_map.restype = ctypes.c_int
L.lzx_dbg_xcc_map.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
m = np.zeros((6, 16), dtype=np.uint32)
pkg._check(L.lzx_dbg_xcc_map(eng.h, 6, m.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))), "xcc", L)
if scale == 0:
    eng.gen_er(n, draws, 1234)
else:
    eng.gen_rmat(scale, n, draws, 1234)
avg, mn = eng.bench_spmv(20)
m2 = np.zeros((3, 16), dtype=np.uint32)
pkg._check(L.lzx_dbg_xcc_map(eng.h, 3, m2.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))), "xcc", L)
print(f"{work}: spmv avg {avg:.4f} min {mn:.4f} ms | XCC of block 0 in 6 launches before: {m[:, 0].tolist()}, after: {m2[:, 0].tolist()} | blocks 0..15 of the first launch: {m[0].tolist()}", flush=True)
eng.close()

"""tools/ab_lib.py <liblzx.so> <workload> [reps] -- SpMV / loop timing of ONE library build on a synthetic workload, through a minimal
ctypes binding of its own (so that an older build without the newer entry points can be measured beside the current one on
the same box: a scratch job script alternates processes).  GPU box only."""
import ctypes
import sys

import numpy as np

path, work = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
WORK = {"c2": (1, 20, 1 << 20, 21_615_022), "c3": (1, 24, 10_000_000, 207_184_357), "er": (0, 0, 10_000_000, 100_000_104),
        "er1m": (0, 0, 1_000_000, 10_000_000)}
kind, scale, n, draws = WORK[work]
L = ctypes.CDLL(path)
h = ctypes.c_void_p()
L.lzx_last_error.restype = ctypes.c_char_p


def chk(rc, what):
    if rc != 0:
        raise SystemExit(f"{what}: {L.lzx_last_error().decode()}")


chk(L.lzx_create(ctypes.byref(h), 0), "create")
L.lzx_gen_graph.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                            ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
ta, tab, tabc = int(round(0.57 * 65536)), int(round(0.76 * 65536)), int(round(0.95 * 65536))
chk(L.lzx_gen_graph(h, kind, scale, n, draws, 1234, ta, tab, tabc), "gen")
avg, mn = ctypes.c_double(), ctypes.c_double()
L.lzx_bench_spmv.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
chk(L.lzx_bench_spmv(h, reps, ctypes.byref(avg), ctypes.byref(mn)), "bench_spmv")


class St(ctypes.Structure):
    _fields_ = [("loop_ms", ctypes.c_double), ("spmv_ms", ctypes.c_double), ("spmv_ms_min", ctypes.c_double), ("vec_ms", ctypes.c_double),
                ("comm_ms", ctypes.c_double), ("iters", ctypes.c_uint32), ("spmv_kernels", ctypes.c_uint32), ("spmv_bytes", ctypes.c_uint64)]


x0 = np.ones(n)
st, xn = St(), ctypes.c_double()
k = 20
f64p = ctypes.POINTER(ctypes.c_double)
L.lzx_lanczos_prepare_f64.argtypes = [ctypes.c_void_p, f64p, ctypes.c_uint32, f64p]
L.lzx_lanczos_run.argtypes = [ctypes.c_void_p, ctypes.POINTER(St)]
best = 1e9
for _ in range(3):
    chk(L.lzx_lanczos_prepare_f64(h, x0.ctypes.data_as(f64p), k, ctypes.byref(xn)), "prepare")
    chk(L.lzx_lanczos_run(h, ctypes.byref(st)), "run")
    best = min(best, st.loop_ms)
print(f"{path.split('/')[-1]:20s} {work}: spmv avg {avg.value:.4f} ms min {mn.value:.4f} | loop k={k}: {best:.3f} ms ({k / best * 1e3:.1f} it/s), spmv {st.spmv_ms / k:.4f} ms/iter", flush=True)
L.lzx_destroy.argtypes = [ctypes.c_void_p]
L.lzx_destroy(h)

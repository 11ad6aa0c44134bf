"""ctypes face of liblzx.so (include/lzx.h) for the parity tests, bench.py and the smoke check.

This is plumbing, not the product: the product is the C-ABI library built from csrc/ and the C++
drop-in classes in host/.  There is no CPU fallback here -- if liblzx.so is missing or a call fails the
error is raised; nothing in this package imports the test oracle.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblzx.so")
# the same code built with -DLZX_DEBUG_KNOBS: experiment knobs, test hooks, ablation switches (Makefile, `make debug`)
DBG_LIB_PATH = os.path.join(_HERE, "liblzx_dbg.so")
# what the product library's lzx_set_option knows (include/lzx.h); any other option name selects the debug library
PRODUCT_OPTIONS = ("hub_entries", "propagation_blocking", "overlap_exchange", "sparse_exchange", "exchange_fp32",
                   "lazy_normalisation", "timing_marks_every", "reorthogonalise", "basis_fp32", "reference_order", "placement_trials",
                   "sharded_ingest")

# test-only shapes the product library accepts through lzx_test_set_shape (csrc/lzx_test_hooks.h): they select among code
# paths the product contains (what large graphs get by themselves), so tests that force them still run liblzx.so
SHAPE_OPTIONS = ("pb_reduce", "pb_target", "pb_unit", "pb_column_band", "pb_run_align", "pb_taper", "pb_dyn_share", "pb_carry_scan", "pb_scatter_nt", "pb_gather_grid", "pb_gather_nt", "spmv_wgs", "pb_group", "pb_group_force",
                 "narrow_slices", "tie_sort", "long_row", "item_len", "exchange_at_world_1", "isolated_rows", "unnormalised_basis", "fuse_staged", "start_vector_scan", "defer_finish")

_u64p = ctypes.POINTER(ctypes.c_uint64)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_f64p = ctypes.POINTER(ctypes.c_double)
_h = ctypes.c_void_p
_hp = ctypes.POINTER(ctypes.c_void_p)


class LzxStats(ctypes.Structure):
    _fields_ = [("loop_ms", ctypes.c_double), ("spmv_ms", ctypes.c_double),
                ("spmv_ms_min", ctypes.c_double), ("vec_ms", ctypes.c_double),
                ("comm_ms", ctypes.c_double), ("iters", ctypes.c_uint32),
                ("spmv_kernels", ctypes.c_uint32), ("spmv_bytes", ctypes.c_uint64)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


class LzxGraphInfo(ctypes.Structure):
    _fields_ = [("n", ctypes.c_uint64), ("nnz", ctypes.c_uint64), ("max_degree", ctypes.c_uint64),
                ("rows_local", ctypes.c_uint64), ("nnz_local", ctypes.c_uint64),
                ("long_rows", ctypes.c_uint64), ("sell_padded", ctypes.c_uint64),
                ("pb_entries", ctypes.c_uint64), ("active_vertices", ctypes.c_uint64),
                ("exchange_slice", ctypes.c_uint64), ("hub_entries", ctypes.c_uint32), ("world", ctypes.c_uint32), ("rank", ctypes.c_uint32),
                ("reserved_", ctypes.c_uint32), ("pb_values", ctypes.c_uint64), ("pb_reduced_entries", ctypes.c_uint64),
                ("exchange_chunk0", ctypes.c_uint64), ("exchange_recv", ctypes.c_uint64),
                ("placement_tried", ctypes.c_uint32), ("placement_kept", ctypes.c_uint32), ("placement_us", ctypes.c_uint32 * 8)]

    def as_dict(self):
        d = {f: getattr(self, f) for f, _ in self._fields_}
        d["placement_us"] = list(d["placement_us"])[:d["placement_tried"]]
        return d


# every symbol include/lzx.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("lzx_create", ctypes.c_int, [_hp, ctypes.c_int]),
    ("lzx_destroy", None, [_h]),
    ("lzx_last_error", ctypes.c_char_p, []),
    ("lzx_comm_unique_id", ctypes.c_int, [_u8p]),
    ("lzx_comm_init_rank", ctypes.c_int, [_h, _u8p, ctypes.c_int, ctypes.c_int]),
    ("lzx_comm_init_local", ctypes.c_int, [_hp, ctypes.c_int]),
    ("lzx_create_group", ctypes.c_int, [_hp, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    ("lzx_comm_ipc_export", ctypes.c_int, [_h, _u8p]),
    ("lzx_comm_ipc_init", ctypes.c_int, [_h, _u8p, ctypes.c_int, ctypes.c_int]),
    ("lzx_set_graph_csr", ctypes.c_int, [_h, ctypes.c_uint64, ctypes.c_uint64, _u64p, _u32p]),
    ("lzx_set_graph_csr32", ctypes.c_int, [_h, ctypes.c_uint32, ctypes.c_uint32, _u32p, _u32p]),
    ("lzx_set_graph_edges", ctypes.c_int, [_h, ctypes.c_uint64, ctypes.c_uint64, _u32p, _u32p]),
    ("lzx_gen_graph", ctypes.c_int, [_h, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64,
                                     ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]),
    ("lzx_get_graph_info", ctypes.c_int, [_h, ctypes.POINTER(LzxGraphInfo)]),
    ("lzx_get_graph_csr", ctypes.c_int, [_h, _u64p, _u32p]),
    ("lzx_spmv_f64", ctypes.c_int, [_h, _f64p, _f64p]),
    ("lzx_spmv_f64_local", ctypes.c_int, [_hp, ctypes.c_int, _f64p, _f64p]),
    ("lzx_lanczos_f64", ctypes.c_int, [_h, _f64p, ctypes.c_uint32, _f64p, _f64p, _f64p, _f64p,
                                       ctypes.POINTER(LzxStats)]),
    ("lzx_lanczos_f64_local", ctypes.c_int, [_hp, ctypes.c_int, _f64p, ctypes.c_uint32, _f64p, _f64p, _f64p,
                                             _f64p, ctypes.POINTER(LzxStats)]),
    ("lzx_lanczos_prepare_f64", ctypes.c_int, [_h, _f64p, ctypes.c_uint32, _f64p]),
    ("lzx_lanczos_run", ctypes.c_int, [_h, ctypes.POINTER(LzxStats)]),
    ("lzx_lanczos_fetch_f64", ctypes.c_int, [_h, ctypes.c_uint32, _f64p, _f64p, _f64p]),
    ("lzx_lanczos_fetch_f64_local", ctypes.c_int, [_hp, ctypes.c_int, ctypes.c_uint32, _f64p, _f64p, _f64p]),
    ("lzx_lanczos_run_steps", ctypes.c_int, [_h, ctypes.c_uint32, ctypes.POINTER(LzxStats)]),
    ("lzx_lanczos_run_steps_local", ctypes.c_int, [_hp, ctypes.c_int, ctypes.c_uint32, ctypes.POINTER(LzxStats)]),
    ("lzx_lanczos_prepare_f64_local", ctypes.c_int, [_hp, ctypes.c_int, _f64p, ctypes.c_uint32, _f64p]),
    ("lzx_lanczos_progress", ctypes.c_int, [_h, _u32p, _u32p]),
    ("lzx_multout_change_f64", ctypes.c_int, [_h, _f64p, ctypes.c_uint32, _f64p]),
    ("lzx_multout_change_f64_local", ctypes.c_int, [_hp, ctypes.c_int, _f64p, ctypes.c_uint32, _f64p]),
    ("lzx_device_count", ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    ("lzx_sync", ctypes.c_int, [_h]),
    ("lzx_multout_f64", ctypes.c_int, [_h, _f64p, ctypes.c_uint32, _f64p]),
    ("lzx_multout_f64_local", ctypes.c_int, [_hp, ctypes.c_int, _f64p, ctypes.c_uint32, _f64p]),
    ("lzx_bench_spmv", ctypes.c_int, [_h, ctypes.c_uint32, _f64p, _f64p]),
    ("lzx_bench_stream", ctypes.c_int, [_h, ctypes.c_uint64, ctypes.c_uint32, _f64p, _f64p]),
    ("lzx_set_option", ctypes.c_int, [_h, ctypes.c_char_p, ctypes.c_int64]),
]

_LIB = None
_DBG_LIB = None


class LzxError(RuntimeError):
    pass


def _load(path: str, mode: int) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise LzxError(f"{path} is missing: run __graft_entry__.build() (make -C {_HERE})")
    L = ctypes.CDLL(path, mode=mode)
    for name, res, args in SYMBOLS:
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    L.lzx_test_set_shape.restype = ctypes.c_int       # test hook, not in include/lzx.h
    L.lzx_test_set_shape.argtypes = [_h, ctypes.c_char_p, ctypes.c_int64]
    L.lzx_test_get_shape.restype = ctypes.c_int
    L.lzx_test_get_shape.argtypes = [_h, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]
    L.lzx_test_allreduce_latency.restype = ctypes.c_int
    L.lzx_test_allreduce_latency.argtypes = [_h, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double)]
    L.lzx_test_rank_row_sums.restype = ctypes.c_int
    L.lzx_test_rank_row_sums.argtypes = [_h, _f64p, _u32p, _u64p]
    return L


def lib(debug: bool = False) -> ctypes.CDLL:
    """liblzx.so (the product), or liblzx_dbg.so when debug-only options are wanted; raises if it has not been
    built (no fallback).  Both are linked -Bsymbolic, so they can be loaded side by side."""
    global _LIB, _DBG_LIB
    if debug:
        if _DBG_LIB is None:
            _DBG_LIB = _load(DBG_LIB_PATH, ctypes.RTLD_LOCAL)
        return _DBG_LIB
    if _LIB is None:
        _LIB = _load(LIB_PATH, ctypes.RTLD_GLOBAL)
    return _LIB


def _p(a, ty):
    return a.ctypes.data_as(ty)


def _check(rc: int, what: str, L=None):
    if rc != 0:
        raise LzxError(f"{what} failed ({rc}): {(L or lib()).lzx_last_error().decode(errors='replace')}")


def rmat_thresholds(a=0.57, b=0.19, c=0.19):
    return int(round(a * 65536)), int(round((a + b) * 65536)), int(round((a + b + c) * 65536))


class Engine:
    """One lzx handle (one GPU)."""

    def __init__(self, device: int = 0, **options):
        self.h = ctypes.c_void_p()
        self.debug = any(k not in PRODUCT_OPTIONS and k not in SHAPE_OPTIONS for k in options)   # experiment knobs: liblzx_dbg.so
        self.L = lib(debug=self.debug)
        _check(self.L.lzx_create(ctypes.byref(self.h), device), "lzx_create", self.L)
        for k, v in options.items():
            self.set_option(k, v)
        self.n = 0

    def set_option(self, name: str, value: int):
        if name in SHAPE_OPTIONS and not self.debug:
            _check(self.L.lzx_test_set_shape(self.h, name.encode(), int(value)), f"lzx_test_set_shape({name})", self.L)
            return
        _check(self.L.lzx_set_option(self.h, name.encode(), int(value)), f"lzx_set_option({name})", self.L)

    def shape(self, name: str) -> int:
        """what shape the blocked tables took (test hook lzx_test_get_shape): gather_items_dealt / _drawn, gather_workgroups"""
        v = ctypes.c_int64()
        _check(self.L.lzx_test_get_shape(self.h, name.encode(), ctypes.byref(v)), f"lzx_test_get_shape({name})", self.L)
        return int(v.value)

    def rank_row_sums(self):
        """test hook lzx_test_rank_row_sums: (row sums of this rank's own rows from its local SpMV of x = 1, caller's vertex id of
        every local row) -- a rank's share checked without its peers"""
        gi = self.info()
        v = np.empty(max(gi["rows_local"], 1))
        pos = np.empty(gi["n"], dtype=np.uint32)
        pad = ctypes.c_uint64()
        _check(self.L.lzx_test_rank_row_sums(self.h, _p(v, _f64p), _p(pos, _u32p), ctypes.byref(pad)), "lzx_test_rank_row_sums", self.L)
        lo = gi["rank"] * pad.value
        mine = np.flatnonzero((pos >= lo) & (pos < lo + gi["rows_local"]))
        ids = np.empty(gi["rows_local"], dtype=np.int64)
        ids[pos[mine].astype(np.int64) - lo] = mine
        return v[:gi["rows_local"]], ids

    def close(self):
        if self.h:
            self.L.lzx_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- communicator ----
    @staticmethod
    def unique_id() -> np.ndarray:
        uid = np.zeros(128, dtype=np.uint8)
        _check(lib().lzx_comm_unique_id(_p(uid, _u8p)), "lzx_comm_unique_id")
        return uid

    def comm_init_rank(self, uid: np.ndarray, rank: int, world: int):
        uid = np.ascontiguousarray(uid, dtype=np.uint8)
        _check(self.L.lzx_comm_init_rank(self.h, _p(uid, _u8p), rank, world), "lzx_comm_init_rank", self.L)

    IPC_BLOB = 128   # LZX_IPC_BLOB

    def allreduce_latency(self, reps: int = 1000) -> float:
        """test hook lzx_test_allreduce_latency: microseconds per two-double all-reduce of the wired communicator (collective)."""
        us = ctypes.c_double()
        _check(self.L.lzx_test_allreduce_latency(self.h, reps, ctypes.byref(us)), "lzx_test_allreduce_latency", self.L)
        return us.value

    def comm_ipc_export(self) -> np.ndarray:
        """This rank's window for the peer-window transport (include/lzx.h): 128 bytes to be gathered from all ranks."""
        blob = np.zeros(self.IPC_BLOB, dtype=np.uint8)
        _check(self.L.lzx_comm_ipc_export(self.h, _p(blob, _u8p)), "lzx_comm_ipc_export", self.L)
        return blob

    def comm_ipc_init(self, blobs: np.ndarray, rank: int, world: int):
        blobs = np.ascontiguousarray(blobs, dtype=np.uint8).reshape(-1)
        if blobs.size != world * self.IPC_BLOB:
            raise ValueError(f"comm_ipc_init: {world} ranks need {world * self.IPC_BLOB} bytes of exports, got {blobs.size}")
        _check(self.L.lzx_comm_ipc_init(self.h, _p(blobs, _u8p), rank, world), "lzx_comm_ipc_init", self.L)

    # ---- graph ----
    def set_graph_csr(self, row_ptr, col_idx):
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        col_idx = np.ascontiguousarray(col_idx, dtype=np.uint32)
        n, nnz = len(row_ptr) - 1, len(col_idx)
        ci = col_idx if nnz else np.zeros(1, dtype=np.uint32)
        _check(self.L.lzx_set_graph_csr(self.h, n, nnz, _p(row_ptr, _u64p), _p(ci, _u32p)), "lzx_set_graph_csr", self.L)
        self.n = n

    def set_graph_csr32(self, row_ptr, col_idx):
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint32)
        col_idx = np.ascontiguousarray(col_idx, dtype=np.uint32)
        n, nnz = len(row_ptr) - 1, len(col_idx)
        ci = col_idx if nnz else np.zeros(1, dtype=np.uint32)
        _check(self.L.lzx_set_graph_csr32(self.h, n, nnz, _p(row_ptr, _u32p), _p(ci, _u32p)), "lzx_set_graph_csr32", self.L)
        self.n = n

    def set_graph_edges(self, n, src, dst):
        src = np.ascontiguousarray(src, dtype=np.uint32)
        dst = np.ascontiguousarray(dst, dtype=np.uint32)
        assert len(src) == len(dst)
        s = src if len(src) else np.zeros(1, dtype=np.uint32)
        d = dst if len(dst) else np.zeros(1, dtype=np.uint32)
        _check(self.L.lzx_set_graph_edges(self.h, n, len(src), _p(s, _u32p), _p(d, _u32p)), "lzx_set_graph_edges", self.L)
        self.n = n

    def gen_er(self, n, draws, seed):
        _check(self.L.lzx_gen_graph(self.h, 0, 0, n, draws, seed, 0, 0, 0), "lzx_gen_graph(er)", self.L)
        self.n = n

    def gen_rmat(self, scale, n, draws, seed, a=0.57, b=0.19, c=0.19):
        ta, tab, tabc = rmat_thresholds(a, b, c)
        _check(self.L.lzx_gen_graph(self.h, 1, scale, n, draws, seed, ta, tab, tabc), "lzx_gen_graph(rmat)", self.L)
        self.n = n

    def info(self) -> dict:
        gi = LzxGraphInfo()
        _check(self.L.lzx_get_graph_info(self.h, ctypes.byref(gi)), "lzx_get_graph_info", self.L)
        return gi.as_dict()

    def get_graph_csr(self):
        gi = self.info()
        rp = np.empty(gi["n"] + 1, dtype=np.uint64)
        ci = np.empty(max(gi["nnz"], 1), dtype=np.uint32)
        _check(self.L.lzx_get_graph_csr(self.h, _p(rp, _u64p), _p(ci, _u32p)), "lzx_get_graph_csr", self.L)
        return rp, ci[:gi["nnz"]]

    # ---- hot path ----
    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert len(x) == self.n
        y = np.empty(self.n)
        _check(self.L.lzx_spmv_f64(self.h, _p(x, _f64p), _p(y, _f64p)), "lzx_spmv_f64", self.L)
        return y

    def lanczos(self, x0, k: int, want_q: bool = True):
        """Returns (alpha[k], beta[k-1], Q (k, n) or None, x_norm, stats dict)."""
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        assert len(x0) == self.n
        alpha = np.zeros(k)
        beta = np.zeros(max(k - 1, 1))
        Q = np.empty((k, self.n)) if want_q else None
        xn = ctypes.c_double()
        st = LzxStats()
        _check(self.L.lzx_lanczos_f64(self.h, _p(x0, _f64p), k, _p(alpha, _f64p), _p(beta, _f64p),
                                     _p(Q, _f64p) if want_q else None, ctypes.byref(xn), ctypes.byref(st)),
               "lzx_lanczos_f64", self.L)
        return alpha, beta[:k - 1], Q, xn.value, st.as_dict()

    def lanczos_prepare(self, x0, k: int) -> float:
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        assert len(x0) == self.n
        xn = ctypes.c_double()
        _check(self.L.lzx_lanczos_prepare_f64(self.h, _p(x0, _f64p), k, ctypes.byref(xn)), "lzx_lanczos_prepare_f64", self.L)
        return xn.value

    def lanczos_run(self) -> dict:
        st = LzxStats()
        _check(self.L.lzx_lanczos_run(self.h, ctypes.byref(st)), "lzx_lanczos_run", self.L)
        return st.as_dict()

    def lanczos_run_steps(self, steps: int) -> dict:
        """Up to `steps` more iterations of the prepared decomposition (lzx_lanczos_run_steps)."""
        st = LzxStats()
        _check(self.L.lzx_lanczos_run_steps(self.h, steps, ctypes.byref(st)), "lzx_lanczos_run_steps", self.L)
        return st.as_dict()

    def lanczos_progress(self):
        done, prep = ctypes.c_uint32(), ctypes.c_uint32()
        _check(self.L.lzx_lanczos_progress(self.h, ctypes.byref(done), ctypes.byref(prep)), "lzx_lanczos_progress", self.L)
        return done.value, prep.value

    def multout_change(self, t) -> float:
        t = np.ascontiguousarray(t, dtype=np.float64)
        rc = ctypes.c_double()
        _check(self.L.lzx_multout_change_f64(self.h, _p(t, _f64p), len(t), ctypes.byref(rc)), "lzx_multout_change_f64", self.L)
        return rc.value

    def lanczos_fetch(self, k: int, want_q: bool = False):
        alpha = np.zeros(k)
        beta = np.zeros(max(k - 1, 1))
        Q = np.empty((k, self.n)) if want_q else None
        _check(self.L.lzx_lanczos_fetch_f64(self.h, k, _p(alpha, _f64p), _p(beta, _f64p),
                                           _p(Q, _f64p) if want_q else None), "lzx_lanczos_fetch_f64", self.L)
        return alpha, beta[:k - 1], Q

    def sync(self):
        _check(self.L.lzx_sync(self.h), "lzx_sync", self.L)

    def multout(self, t):
        t = np.ascontiguousarray(t, dtype=np.float64)
        ans = np.empty(self.n)
        _check(self.L.lzx_multout_f64(self.h, _p(t, _f64p), len(t), _p(ans, _f64p)), "lzx_multout_f64", self.L)
        return ans

    def bench_stream(self, nbytes: int = 1 << 30, reps: int = 5):
        rd, cp = ctypes.c_double(), ctypes.c_double()
        _check(self.L.lzx_bench_stream(self.h, nbytes, reps, ctypes.byref(rd), ctypes.byref(cp)), "lzx_bench_stream", self.L)
        return rd.value, cp.value

    def bench_spmv(self, reps: int = 20):
        avg, mn = ctypes.c_double(), ctypes.c_double()
        _check(self.L.lzx_bench_spmv(self.h, reps, ctypes.byref(avg), ctypes.byref(mn)), "lzx_bench_spmv", self.L)
        return avg.value, mn.value


class LocalGroup:
    """`world` handles wired as an in-process communicator (lzx_comm_init_local)."""

    def __init__(self, devices, **options):
        self.engines = [Engine(d, **options) for d in devices]
        self.L = self.engines[0].L
        self.world = len(devices)
        self.arr = (ctypes.c_void_p * self.world)(*[e.h for e in self.engines])
        _check(self.L.lzx_comm_init_local(self.arr, self.world), "lzx_comm_init_local", self.L)
        self.n = 0

    @classmethod
    def create(cls, devices):
        """the same group made by ONE call of the C ABI (lzx_create_group: SURVEY.md 8(b)'s `lzx_create(out, n_devices, device_ids)`)"""
        self = cls.__new__(cls)
        self.L = lib()
        self.world = len(devices)
        self.arr = (ctypes.c_void_p * self.world)()
        ids = (ctypes.c_int * self.world)(*devices)
        _check(self.L.lzx_create_group(self.arr, self.world, ids), "lzx_create_group", self.L)
        self.engines = []
        for h in self.arr:
            e = Engine.__new__(Engine)
            e.h, e.debug, e.L, e.n = ctypes.c_void_p(h), False, self.L, 0
            self.engines.append(e)
        self.n = 0
        return self

    def set_graph_csr(self, row_ptr, col_idx):
        for e in self.engines:
            e.set_graph_csr(row_ptr, col_idx)
        self.n = self.engines[0].n

    def gen_rmat(self, scale, n, draws, seed, a=0.57, b=0.19, c=0.19):
        for e in self.engines:
            e.gen_rmat(scale, n, draws, seed, a, b, c)
        self.n = self.engines[0].n

    def gen_er(self, n, draws, seed):
        for e in self.engines:
            e.gen_er(n, draws, seed)
        self.n = self.engines[0].n

    def set_graph_edges(self, n, src, dst):
        for e in self.engines:
            e.set_graph_edges(n, src, dst)
        self.n = self.engines[0].n

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(self.n)
        _check(self.L.lzx_spmv_f64_local(self.arr, self.world, _p(x, _f64p), _p(y, _f64p)), "lzx_spmv_f64_local", self.L)
        return y

    def lanczos(self, x0, k: int, want_q: bool = True):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        alpha = np.zeros(k)
        beta = np.zeros(max(k - 1, 1))
        Q = np.empty((k, self.n)) if want_q else None
        xn = ctypes.c_double()
        st = LzxStats()
        _check(self.L.lzx_lanczos_f64_local(self.arr, self.world, _p(x0, _f64p), k, _p(alpha, _f64p),
                                           _p(beta, _f64p), _p(Q, _f64p) if want_q else None,
                                           ctypes.byref(xn), ctypes.byref(st)), "lzx_lanczos_f64_local", self.L)
        return alpha, beta[:k - 1], Q, xn.value, st.as_dict()

    def multout(self, t):
        t = np.ascontiguousarray(t, dtype=np.float64)
        ans = np.empty(self.n)
        _check(self.L.lzx_multout_f64_local(self.arr, self.world, _p(t, _f64p), len(t), _p(ans, _f64p)), "lzx_multout_f64_local", self.L)
        return ans

    def close(self):
        for e in self.engines:
            e.close()

"""tools/shard_probe.py <workload> [world] [rank ...] -- whole-graph against sharded hand-over (option sharded_ingest) on the
GPU box: seconds per hand-over, the device-memory high-water mark while it runs (rocm-smi style: sampled from a thread
through hipMemGetInfo of a second context-free call, 5 ms period), what stays resident afterwards, and that the SpMV of the
two agree bit for bit.  `world` > 1 builds the share of the given ranks of an in-process group one at a time
(lzx_comm_init_local on GPU 0), the way every process of an N-GPU run would build its own.

    python tools/shard_probe.py c3            # one rank
    python tools/shard_probe.py c3 8 0 5      # ranks 0 and 5 of 8
    python tools/shard_probe.py c5 8 0        # rank 0 of 8 on the 100 M-vertex graph (sharded only: see WHOLE_TOO)
"""
import ctypes
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge

pkg = ge.load_pkg()
WORK = {"c3": (1, 24, 10_000_000, 200_000_000, 1234), "c2": (1, 20, 1 << 20, 20_000_000, 1234),
        "er": (0, 0, 10_000_000, 100_000_000, 1234), "c5": (1, 27, 100_000_000, 2_000_000_000, 1234)}
work = sys.argv[1] if len(sys.argv) > 1 else "c3"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ranks = [int(a) for a in sys.argv[3:]] or [0]
kind, scale, n, draws, seed = WORK[work]
WHOLE_TOO = os.environ.get("SHARD_PROBE_WHOLE", "0" if work == "c5" and world > 1 else "1") == "1"

hip = ctypes.CDLL("libamdhip64.so")


def mem_used():
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
    return total.value - free.value


class Peak:
    def __enter__(self):
        self.stop, self.peak = False, mem_used()

        def loop():
            while not self.stop:
                self.peak = max(self.peak, mem_used())
                time.sleep(0.005)
        self.t = threading.Thread(target=loop, daemon=True)
        self.t.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        self.t.join()


HOST_CSR = None
if os.environ.get("SHARD_PROBE_SOURCE") == "csr":   # the same graph as a CSR in host memory (lzx_set_graph_csr), streamed when sharded
    helper = pkg.Engine(0)
    (helper.gen_er(n, draws, seed) if kind == 0 else helper.gen_rmat(scale, n, draws, seed))
    HOST_CSR = helper.get_graph_csr()
    helper.close()


def build(rank, **options):
    if world == 1:
        engines = [pkg.Engine(0, **options)]
    else:
        grp = pkg.LocalGroup([0] * world, **options)
        engines = grp.engines
    e = engines[rank]
    base = mem_used()
    t = time.perf_counter()
    with Peak() as pk:
        if HOST_CSR is not None:
            e.set_graph_csr(*HOST_CSR)
        elif kind == 0:
            e.gen_er(n, draws, seed)
        else:
            e.gen_rmat(scale, n, draws, seed)
    dt = time.perf_counter() - t
    return engines, e, dt, pk.peak - base, mem_used() - base


for rank in ranks:
    res = {}
    for tag, opts in (("whole", dict()), ("sharded", dict(sharded_ingest=1))):
        if tag == "whole" and not WHOLE_TOO:
            continue
        engines, e, dt, peak, resident = build(rank, **opts)
        gi = e.info()
        sums = None
        if world > 1:
            sums = e.rank_row_sums()[0]
        else:
            sums = e.spmv(np.ones(n))
        res[tag] = sums
        print(f"{work}{' (host CSR)' if HOST_CSR is not None else ''} rank {rank} of {world} [{tag}]: hand-over {dt:.2f} s, peak {peak / 1e9:.2f} GB above the idle handle, "
              f"resident afterwards {resident / 1e9:.2f} GB; nnz={gi['nnz']} nnz_local={gi['nnz_local']} pb_values={gi['pb_values']}", flush=True)
        for x in engines:
            x.close()
    if len(res) == 2:
        assert np.array_equal(res["whole"], res["sharded"])
        print(f"{work} rank {rank} of {world}: row sums of the two hand-overs identical", flush=True)

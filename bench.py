#!/usr/bin/env python3
"""bench.py -- Lanczos iterations/sec + SpMV effective HBM GB/s on synthetic R-MAT graphs (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W [--workload c3|c2|c1|er|er1m|c5]

A "step" is one Lanczos iteration (serial/lib/lanczos.cc:21-53) on a graph already reshaped and resident in
HBM.  The decomposition that is timed is BASELINE's: k = the workload's Krylov dimension (50; C5 30; C1 20 -- what
parallel-final/main.cu:104-116 times), prepared with lzx_lanczos_prepare_f64(x0, k); the timed region is exactly K = --steps
iterations of it (lzx_lanczos_run_steps), bracketed by a barrier and a device synchronisation on both sides, MAX over
ranks; the remaining k - K iterations then complete the decomposition outside the clock (config.k and steps are reported
separately; --steps above the workload's k lengthens the decomposition to K).  N > 1: one process per GPU (torch.distributed.run), rows
dealt to ranks by degree rank, the new Lanczos vector re-assembled each iteration by an RCCL all-gather
inside liblzx.so (torch.distributed only carries the 128-byte communicator id, the barrier and the max).
The same graph is used at every N, so scaling is "strong".

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

# multi-process RCCL on this platform needs dmabuf IPC (the host driver has no legacy IPC); must be set before HIP starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# peer-window transport (--transport ipc, or the fallback when RCCL cannot build an engine): a rank waits this long for a peer
# at a synchronisation point before it fails with LZX_ERR_COMM.  Graph hand-overs of different ranks may end many seconds apart
# (C5: 13-45 s each), so the bench allows more than the library's default 20 s -- a hang still ends, with an error, after 3 min.
os.environ.setdefault("LZX_IPC_TIMEOUT_MS", "180000")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

# BASELINE.json names EDGES ("10 M-node / 200 M-edge R-MAT"); the generators take DRAWS and drop duplicates and self loops (200 M
# draws leave 193.2 M distinct edges).  Round 5: the draw counts below are the smallest that reach the named
# number of distinct undirected edges (tools/edge_count_probe.py: bisection over the counter-based generator, whose draw i does not
# depend on the total; profiles/r5_edge_counts.txt).  C5 keeps its 2 G draws (1.96 G distinct edges: a bisection over 100 M-vertex
# graphs is minutes of GPU time per probe) and says so.
C2_DRAWS, C3_DRAWS, ER_DRAWS = 21_615_022, 207_184_357, 100_000_104
# name -> (description, generator kind, scale, n, draws, seed, default k)
WORKLOADS = {
    "c1": ("C1: Erdos-Renyi n=10k, 100k draws (99,906 distinct undirected edges: the committed fixture of tests/golden), seed 1234", "er", 0, 10_000, 100_000, 1234, 20),
    "c2": ("C2: R-MAT scale 20 (a,b,c,d)=(.57,.19,.19,.05), n=1,048,576, 20,000,000 distinct undirected edges "
           f"({C2_DRAWS:,} draws), seed 1234", "rmat", 20, 1 << 20, C2_DRAWS, 1234, 50),
    "c3": ("C3/C4: R-MAT scale 24 (a,b,c,d)=(.57,.19,.19,.05), endpoints >= n re-drawn, n=10,000,000, "
           f"200,000,000 distinct undirected edges ({C3_DRAWS:,} draws), seed 1234", "rmat", 24, 10_000_000, C3_DRAWS, 1234, 50),
    # the uniform family north_star names beside R-MAT ("a 100 M-edge graph"): no hubs, every (row, column band) pair holds
    # about one entry, so nothing can be summed before it crosses the two passes
    "er": (f"ER: Erdos-Renyi G(n, M) n=10,000,000, 100,000,000 distinct undirected edges ({ER_DRAWS:,} draws; north_star's 100 M-edge "
           "graph), seed 1234", "er", 0, 10_000_000, ER_DRAWS, 1234, 50),
    "er1m": ("ER: Erdos-Renyi n=1,000,000, 10M draws, seed 1234", "er", 0, 1_000_000, 10_000_000, 1234, 50),
    # 3.9e9 stored entries (> 2^32).  At N > 1 every rank sweeps the generator in bounded batches and keeps its own rows only
    # (option sharded_ingest: 19 GB at the peak for a rank of 8 instead of 102 GB); the whole graph also fits one MI355X (about
    # 110 iter/s there).  Use --no-cpu-baseline at N = 1: the host loop takes minutes per iteration.
    "c5": ("C5: R-MAT scale 27 (a,b,c,d)=(.57,.19,.19,.05), endpoints >= n re-drawn, n=100,000,000, 2G draws "
           "(1.96 G distinct undirected edges), seed 1234", "rmat", 27, 100_000_000, 2_000_000_000, 1234, 30),
}
HBM_PEAK_GBS = 8000.0  # MI355X datasheet, /opt/skills/guides/MI355X_MICROARCH.md
EXIT_TRIAL_HUNG = 75   # the overlapped-exchange trial never completed (a hung collective): the held line is printed, the run is NOT a success


def build_id() -> str:
    """Hash of the kernel sources this run's liblzx.so was built from (csrc/*.hip, *.h, include/lzx.h; they travel with
    the snapshot, .git does not): ties a bench line to the build, and the committed PMC traffic figure to the build it was
    measured on (profiles/pmc_traffic.json carries the id of the profiled run's own bench line)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "msc-hpc-final-project_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "msc-hpc-final-project_amd", "csrc", "*.h")) +
                   [os.path.join(ROOT, "include", "lzx.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(workload: str, world: int):
    """HBM bytes per SpMV launch from the committed rocprofv3 PMC passes (bench.py cannot wrap itself in the
    profiler): (bytes, source, build id of the profiled run or None); all None when no pass exists for this workload /
    rank count."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if world != 1 or not os.path.exists(path):
        return None, None, None
    try:
        d = json.load(open(path)).get(workload)
        return (d["hbm_bytes_per_launch"], d["source"], d.get("build_id")) if d else (None, None, None)
    except Exception:
        return None, None, None


def byte_cost_model(workload: str, world: int, stream, spmv_ms: float):
    """What the SpMV's MEASURED bytes would cost at this box's own streaming rates, the two directions priced apart: a read byte
    at 1 / (the read-only probe's rate), a written byte at the rate the copy probe implies for its written half
    (1 / w = 2 / copy - 1 / read: the copy moves as many bytes each way).  `measured_over_model` > 1 is what the kernels lose
    against their bytes; `traffic` / algorithmic bytes is what the format loses against the matrix.  None without a PMC pass."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(path)).get(workload) if world == 1 else None
        rd, wr = d["read_bytes_per_launch"], d["written_bytes_per_launch"]
        r, cp = stream[0] * 1e9, stream[1] * 1e9
        inv_w = 2.0 / cp - 1.0 / r
        if not (r > 0 and cp > 0 and inv_w > 0 and spmv_ms > 0):
            return None
        ms = (rd / r + wr * inv_w) * 1e3
        return {"read_bytes": rd, "written_bytes": wr, "read_GBps": r / 1e9, "write_GBps": 1e-9 / inv_w, "ms": ms,
                "measured_over_model": spmv_ms / ms}
    except Exception:
        return None


def cpu_baseline(eng, O, budget_s: float = 10.0, omp_budget_s: float = 5.0):
    """The oracle's restatement of serial/ (single thread) on the SAME graph, a bounded number of iterations -- and, beside it
    ("optionally an OpenMP all-core SpMV number labelled as such"), the same loop with OpenMP over rows and
    elements on the box's CPU share (oracle/lanczos_oracle_omp.c, kind "port-omp": a timing baseline whose inner products are
    OpenMP reductions, not the reference's arithmetic)."""
    rp, ci = eng.get_graph_csr()
    n = len(rp) - 1
    x0 = np.ones(n)
    t = time.perf_counter()
    O.lanczos(rp, ci, 2, x0, want_q=False)
    per_iter = max((time.perf_counter() - t) / 2.0, 1e-6)
    k = int(max(2, min(50, budget_s / per_iter)))
    t = time.perf_counter()
    O.lanczos(rp, ci, k, x0, want_q=False)
    dt = time.perf_counter() - t
    out = {"value": k / dt, "unit": "iter/s", "cores": 1, "kind": "port",
           "sample": f"{k} Lanczos iterations of the same graph (oracle/lanczos_oracle.c, -O3 -ffp-contract=off as the reference builds, 1 thread, "
                     f"{dt:.1f} s; host has {os.cpu_count()} hardware threads)"}
    if omp_budget_s > 0:
        try:
            threads = max(1, min(int(os.environ.get("LZX_BENCH_OMP_THREADS", "16")), os.cpu_count() or 1))   # a one-GPU box's CPU share
            O.lanczos_omp(rp, ci, 2, x0, threads=threads)                      # threads started, pages touched
            t = time.perf_counter()
            _, _, _, used = O.lanczos_omp(rp, ci, 2, x0)
            per_iter = max((time.perf_counter() - t) / 2.0, 1e-6)
            ko = int(max(2, min(50, omp_budget_s / per_iter)))
            t = time.perf_counter()
            O.lanczos_omp(rp, ci, ko, x0)
            dto = time.perf_counter() - t
            out["all_cores"] = {"value": ko / dto, "unit": "iter/s", "cores": used, "kind": "port-omp",
                                "sample": f"{ko} iterations of the same loop with OpenMP over rows and elements (oracle/lanczos_oracle_omp.c; inner products "
                                          f"are OpenMP reductions: another rounding than serial/'s), {used} threads, {dto:.1f} s"}
        except Exception as exc:   # no OpenMP runtime on the box: the reference-faithful number stands alone
            out["all_cores"] = {"value": None, "kind": "port-omp", "sample": f"not measured: {exc}"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=os.environ.get("LZX_BENCH_WORKLOAD", "c3"), choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=10.0,
                    help="seconds of single-thread oracle work the cpu_baseline leg may take (N = 1; the all-core leg takes half of it)")
    # N > 1 only.  rccl: ncclAllGather / grouped send-receive / ncclAllReduce (the default: the collective library is the
    # safer choice on a node this code has never seen).  ipc: the peer-window transport of csrc/lzx_ipc.hip (buffers mapped
    # across processes, pushed slices, mailbox all-reduce) -- measured on request until it has run on two physical GPUs.
    # LZX_BENCH_ONE_GPU=1 puts all ranks on GPU 0 (ipc only; torch's own group over gloo): a functional run of the N > 1
    # flow on a one-GPU box, not a measurement of scaling.
    ap.add_argument("--transport", default=os.environ.get("LZX_BENCH_TRANSPORT", "rccl"), choices=["rccl", "ipc"])
    args = ap.parse_args()
    one_gpu = os.environ.get("LZX_BENCH_ONE_GPU") == "1"
    if one_gpu:
        args.transport = "ipc"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start one rank per GPU as CHILD processes (nothing has touched the GPU
        # yet, and nothing is exec'ed from a GPU process) and relay rank 0's JSON line and the return code
        import socket
        import subprocess
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        child = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
        # rank 0's JSON line on stdout; whatever else the ranks printed there (gloo announces its connections on stdout) on stderr
        for ln in child.stdout.splitlines():
            print(ln, file=sys.stdout if ln.startswith("{") else sys.stderr)
        sys.stdout.flush()
        sys.exit(child.returncode)
    if world != args.gpus:
        args.gpus = world

    import torch  # device plumbing + torch.distributed only
    dist = None
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ   # under torch.distributed.run, also at N = 1
    if world > 1 or launched:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    tdev = "cpu" if one_gpu else "cuda"   # where the few scalars torch.distributed reduces live

    pkg = ge.load_pkg()
    desc, kind, scale, n, draws, seed, k_workload = WORKLOADS[args.workload]
    K, W = args.steps, args.warmup
    k_cfg = max(k_workload, K)   # the decomposition timed is BASELINE's k-step one; K of its iterations are on the clock
    bid = build_id()

    # LZX_BENCH_REHEARSE_MULTI=1 (with torch.distributed.run --nproc-per-node 1): the N > 1 flow -- exchange tuning, both
    # engines, RCCL collectives on the 1-rank communicator -- on a box with one GPU.  A rehearsal of the code path, not
    # a measurement of anything.
    rehearse = world == 1 and launched and os.environ.get("LZX_BENCH_REHEARSE_MULTI") == "1"

    def all_ok(ok: bool) -> bool:
        """Every rank learns whether a LOCAL step failed anywhere, before anybody enters the next collective."""
        if dist is None:
            return ok
        flag = torch.tensor([0.0 if ok else 1.0], dtype=torch.float64, device=tdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return flag.item() == 0.0

    # Sharded hand-over (every rank sweeps the generator in bounded batches and keeps its own rows; same tables bit for bit):
    # on by default where the whole graph per rank is what hurts -- C5 at N > 1 -- or by LZX_BENCH_SHARDED_INGEST=<sweeps>.
    shard_opt = int(os.environ.get("LZX_BENCH_SHARDED_INGEST", "1" if (args.workload == "c5" and world > 1) else "0"))

    def make_engine(**options):
        """(engine, seconds of graph build), or (None, 0) on EVERY rank when a local step failed on any of them.
        Local steps (handle creation, graph generation + reshaping) and the communicator's creation are guarded and
        agreed on with an all-reduce."""
        # the library's default is 2 candidates for the value stream (lzx.h, "placement_trials"); the bench asks for the full seven
        # and reports what each read (config.placement)
        options = dict(options, placement_trials=int(os.environ.get("LZX_BENCH_PLACEMENT_TRIALS", "7")))
        if rehearse:
            options = dict(options, exchange_at_world_1=1)
        if shard_opt:
            options = dict(options, sharded_ingest=shard_opt)
        e = None
        try:
            e = pkg.Engine(local_rank, **options)
        except Exception as exc:
            print(f"[bench rank {rank}] engine: {exc}", file=sys.stderr, flush=True)
        if not all_ok(e is not None):
            if e is not None:
                e.close()
            return None, 0.0
        if dist is not None and args.transport == "ipc":
            blob, ok = np.zeros(pkg.Engine.IPC_BLOB, dtype=np.uint8), True
            try:
                blob = e.comm_ipc_export()
            except Exception as exc:
                print(f"[bench rank {rank}] window export: {exc}", file=sys.stderr, flush=True)
                ok = False
            if not all_ok(ok):
                e.close()
                return None, 0.0
            blobs = [torch.zeros(pkg.Engine.IPC_BLOB, dtype=torch.uint8, device=tdev) for _ in range(world)]
            dist.all_gather(blobs, torch.from_numpy(blob.copy()).to(tdev))
            try:
                e.comm_ipc_init(torch.cat(blobs).cpu().numpy(), rank, world)   # collective (every rank maps every window)
            except Exception as exc:
                print(f"[bench rank {rank}] peer windows: {exc}", file=sys.stderr, flush=True)
                ok = False
            if not all_ok(ok):
                e.close()
                return None, 0.0
        elif dist is not None:
            uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                uid.copy_(torch.from_numpy(pkg.Engine.unique_id()))
            dist.broadcast(uid, 0)
            ok = True
            try:
                e.comm_init_rank(uid.cpu().numpy(), rank, world)
            except Exception as exc:   # (a rank that fails here alone leaves its peers inside ncclCommInitRank: nothing to be done about that)
                print(f"[bench rank {rank}] RCCL communicator: {exc}", file=sys.stderr, flush=True)
                ok = False
            if not all_ok(ok):
                e.close()
                return None, 0.0
        t = time.perf_counter()
        ok = True
        try:
            if kind == "er":
                e.gen_er(n, draws, seed)
            else:
                e.gen_rmat(scale, n, draws, seed)
        except Exception as exc:
            print(f"[bench rank {rank}] graph: {exc}", file=sys.stderr, flush=True)
            ok = False
        if not all_ok(ok):
            e.close()
            return None, 0.0
        return e, time.perf_counter() - t

    def barrier(e):
        e.sync()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    x0 = np.ones(n)

    def transport_self_check(e):
        """N > 1, before anything is timed (round 5, ADVICE r4: the scored run may be the first time a transport moves data
        between two physical GPUs, and alpha_0 against its closed form does not exercise the exchange at all).  Three products
        through the exchange under test, none needing the oracle or the whole graph on a rank:
          * A = A^T: z . (A x) = x . (A z) for two seeded random vectors -- an entry that arrived late, stale or in the wrong
            place breaks it;
          * A 1 is integer-valued and sums to nnz;
          * every rank holds the same A x, bit for bit (a 64-bit checksum of its bytes, gathered).
        Returns the figures for the JSON line; raises on a violation (the caller agrees on the outcome across ranks)."""
        rng = np.random.default_rng(20260705)
        x, z = rng.random(n), rng.random(n)
        ax, az = e.spmv(x), e.spmv(z)
        lhs, rhs = float(z @ ax), float(x @ az)
        sym = abs(lhs - rhs) / max(abs(lhs), 1e-300)
        ones = e.spmv(x0)
        rows_exact = bool(np.array_equal(ones, np.rint(ones)) and float(ones.sum()) == float(e.info()["nnz"]))
        words = ax.view(np.uint64)
        digest = np.array([int(np.bitwise_xor.reduce(words)) & 0x7FFFFFFFFFFFFFFF, int(words.sum(dtype=np.uint64)) & 0x7FFFFFFFFFFFFFFF], dtype=np.int64)
        same = True
        if dist is not None:
            mine = torch.from_numpy(digest).to(tdev)
            everyone = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(everyone, mine)
            same = all(torch.equal(t, everyone[0]) for t in everyone)
        res = {"symmetry_rel": sym, "row_sums_exact": rows_exact, "ranks_bit_equal": same}
        if not (sym < 1e-11 and rows_exact and same):
            raise RuntimeError(f"transport self-check failed on rank {rank}: {res}")
        return res

    def measure(e):
        """W untimed iterations (of a k-step decomposition of their own), then the workload's k-step decomposition is prepared and
        exactly K of its iterations run between barrier + synchronise on both sides, max over ranks; the other k - K follow
        outside the clock."""
        if W > 0:
            # a decomposition of the timed one's shape (the resident basis is sized for k columns once, here), W of its iterations
            e.lanczos_prepare(x0, k_cfg)
            e.lanczos_run_steps(W)
        t_in = time.perf_counter()
        e.lanczos_prepare(x0, k_cfg)     # x0 uploaded, q_0 in HBM, basis sized for k columns: inputs resident before the clock starts
        barrier(e)
        t_in = time.perf_counter() - t_in
        t0 = time.perf_counter()
        st = e.lanczos_run_steps(K)      # exactly K iterations; returns after a stream synchronise
        barrier(e)
        elapsed = time.perf_counter() - t0
        assert st["iters"] == K, st
        t_rest = time.perf_counter()
        if k_cfg > K:
            e.lanczos_run_steps(k_cfg - K)
        barrier(e)
        t_rest = time.perf_counter() - t_rest
        m = {}
        if dist is not None:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
            agg = torch.tensor([st["spmv_ms"], float(st["spmv_bytes"]), st["comm_ms"], st["vec_ms"]],
                               dtype=torch.float64, device=tdev)
            mx, mn = agg.clone(), agg.clone()
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            dist.all_reduce(mn, op=dist.ReduceOp.MIN)
            dist.all_reduce(agg, op=dist.ReduceOp.SUM)
            m.update(spmv_ms_max=float(mx[0]), comm_ms=float(mx[2]), comm_ms_min=float(mn[2]), spmv_ms_min=float(mn[0]), vec_ms=float(mx[3]),
                     spmv_bytes_total=float(agg[1]))
        else:
            m.update(spmv_ms_max=st["spmv_ms"], comm_ms=st["comm_ms"], comm_ms_min=st["comm_ms"], spmv_ms_min=st["spmv_ms"], vec_ms=st["vec_ms"],
                     spmv_bytes_total=float(st["spmv_bytes"]))
        t_out = time.perf_counter()
        alpha, beta, _ = e.lanczos_fetch(k_cfg)
        t_out = time.perf_counter() - t_out
        m.update(elapsed=elapsed, t_in=t_in, t_out=t_out, t_rest=t_rest,
                 finite=bool(np.isfinite(alpha).all() and np.isfinite(beta).all()),
                 head=np.concatenate([alpha[:3], beta[:2]]))   # alpha_0..2, beta_0..1: compared across exchange modes / with the closed form
        return m

    stream = None

    def line(e, m, tune):
        """rank 0's JSON line for measurement m of engine e"""
        nonlocal stream
        gi = e.info()
        if stream is None:
            stream = e.bench_stream(2 << 30, 5)   # 2 GiB: eight times the Infinity Cache
        elapsed = m["elapsed"]
        spmv_avg_ms = m["spmv_ms_max"] / K
        tried, place_us = gi["placement_tried"], gi["placement_us"]
        achieved = m["spmv_bytes_total"] / (spmv_avg_ms * 1e-3) / 1e9 if spmv_avg_ms > 0 else 0.0
        return {
            "metric": "lanczos_iterations_per_sec",
            "value": K / elapsed,
            "unit": "iter/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": desc,
                "n": gi["n"], "undirected_edges": gi["nnz"] // 2, "nnz": gi["nnz"], "max_degree": gi["max_degree"],
                "k": k_cfg, "x0": "ones",
                "timed": f"iterations 0..{K - 1} of the k = {k_cfg} decomposition (lzx_lanczos_prepare_f64(x0, {k_cfg}), then "
                         f"lzx_lanczos_run_steps({K}) on the clock; the other {k_cfg - K} run afterwards, outside it)",
                "build_id": bid,
                "partition": "single GPU" if world == 1 else
                             f"rows dealt round-robin by degree rank over {world} " + ("ranks sharing ONE GPU" if one_gpu else "GPUs") +
                             f"; per iteration {8 * gi['exchange_recv']} B received per rank "
                             + ("pushed by the peers' kernels into mapped buffers (peer-window transport)" if args.transport == "ipc" else "over RCCL") +
                             f" (slices of {8 * gi['exchange_slice']} B: only the {gi['active_vertices']} "
                             f"vertices that have an edge are exchanged, unnormalised; "
                             + (f"two chunks overlapping the SpMV, the second one sparse: each peer sends only what this rank's rows reference"
                                if gi.get("exchange_chunk0") else "one all-gather") + ") + 1 two-double all-reduce" + (" through mailboxes in device memory" if args.transport == "ipc" else ""),
                "transport": args.transport if world > 1 or launched else None,
                "transport_fallback": tune.get("transport_fallback"),
                "transport_self_check": m.get("self_check"),
                # option placement_trials = 7 (this bench's choice; the library's default is 2): the value stream between the
                # SpMV's two passes was allocated `tried` times at the hand-over, the SpMV timed with each, the fastest kept
                "placement": {"tried": tried, "kept": gi["placement_kept"], "spmv_us_per_candidate": place_us,
                              "chosen_us": min(place_us) if place_us else None, "worst_us": max(place_us) if place_us else None} if tried else None,
                "exchange_tuning_ms_per_iter": dict(tune) or None,
                "exchange_chunk0_doubles_per_rank": gi.get("exchange_chunk0", 0),
                "graph_build_s": round(t_gen, 3),
                "graph_hand_over": "sharded: each rank sweeps the generator in bounded batches and keeps its own rows" if shard_opt else "whole graph on every rank",
                # not `value`: the whole k-step decomposition with the host hand-over (x0 upload, basis set-up) and the
                # download of alpha / beta included -- what a caller holding host buffers sees (rank 0's clock)
                "iters_per_sec_including_host_transfers": k_cfg / (elapsed + m["t_rest"] + m["t_in"] + m["t_out"]),
                # ... its parts on rank 0's clock, ms: lzx_lanczos_prepare_f64 (x0 in, q_0, basis sized) + barrier; the K timed iterations;
                # the other k - K (a second lzx_lanczos_run_steps call + barrier); lzx_lanczos_fetch_f64 of alpha / beta
                "host_hand_over_ms": {"prepare": 1e3 * m["t_in"], "timed_iterations": 1e3 * elapsed, "other_iterations": 1e3 * m["t_rest"],
                                      "fetch_alpha_beta": 1e3 * m["t_out"]},
                "lanczos_coefficients_finite": m["finite"],
                # x0 = ones: alpha_0 = q_0' A q_0 = nnz / n exactly (a check of the run itself, not of parity -- that is tests/)
                "alpha0_vs_closed_form_rel": abs(float(m["head"][0]) * gi["n"] / gi["nnz"] - 1.0) if gi["nnz"] else None,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": ("k_pb_scatter_spmv (scatter pass + staged columns from LDS, one launch; k_spmv + k_pb_scatter when they "
                           "are launched separately) + k_pb_gather: propagation-blocked CSR SpMV (partial row sums cross the two "
                           "passes) fused with the alpha partial; the totals of row bands cut into several gather items are added "
                           "by k_lazy_update in this loop, by k_pb_finish elsewhere") if gi["pb_entries"] else
                          "k_spmv (+ k_long_finish): CSR SpMV fused with the alpha partial",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS * (1 if one_gpu else world),
                "unit": "GB/s",
                "frac": achieved / (HBM_PEAK_GBS * (1 if one_gpu else world)),
                # PMC bytes per SpMV from the committed rocprofv3 passes (a bench run cannot wrap itself in the profiler);
                # traffic_build_id is the build_id of the profiled run's own bench line: equal to config.build_id when the
                # figure was measured on this very build
                "traffic": pmc_traffic(args.workload, world)[0],
                "traffic_source": pmc_traffic(args.workload, world)[1],
                "traffic_build_id": pmc_traffic(args.workload, world)[2],
                "traffic_measured_on_this_build": pmc_traffic(args.workload, world)[2] == bid,
                # those bytes, read and written apart, priced at this box's own streaming rates (below): what is left between
                # the model and the measured SpMV is the kernels', what is between the traffic and the algorithmic bytes the format's
                "byte_cost_model": byte_cost_model(args.workload, world, stream, spmv_avg_ms),
                # the same box's own streaming rates (read-only sum / copy over 1 GiB, rank 0) and the SpMV against them
                "measured_stream_read_GBps": stream[0], "measured_stream_copy_GBps": stream[1],
                "frac_of_measured_stream_read": achieved / (stream[0] * world) if stream[0] else None,
                "algorithmic_bytes_per_spmv": m["spmv_bytes_total"],
                "avg_spmv_ms": spmv_avg_ms,
                "spmv_share_of_loop": m["spmv_ms_max"] / (elapsed * 1e3),
                "vector_kernels_ms_per_iter": m["vec_ms"] / K,
                "exchange_ms_per_iter": m["comm_ms"] / K,
                # over the ranks (the figures above are the slowest rank's)
                "exchange_ms_per_iter_max_min": [m["comm_ms"] / K, m["comm_ms_min"] / K],
                "spmv_ms_per_iter_max_min": [m["spmv_ms_max"] / K, m["spmv_ms_min"] / K],
            },
        }

    tune = {}
    if world == 1 and not rehearse:
        eng, t_gen = make_engine()
        if eng is None:
            sys.exit("bench.py: could not build the engine")
        out = line(eng, measure(eng), tune)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(eng, ge.load_oracle(), args.cpu_budget_s, args.cpu_budget_s / 2)
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    else:
        # Several ranks.  The exchange can run as one all-gather before the SpMV, or as two chunks overlapping the blocked
        # SpMV, the second one sparse.  Which is faster depends on the node's xGMI and on the rank
        # count; the second has never run on two physical GPUs.  So: (1) the whole measurement -- warm-up, K timed
        # iterations, rank aggregation -- is done in the single-all-gather mode FIRST and rank 0 holds its line; (2) the
        # overlapped mode is then tried for a few iterations under a watchdog and, if it is faster on the slowest rank,
        # measured in full and reported instead.  Whatever happens in (2) -- an error (every rank falls back) or a
        # collective that never completes (the watchdog prints the held line and ends the process with EXIT_TRIAL_HUNG: a
        # hang is a failure and shows in the return code) -- the run still delivers the line of (1).  Nothing is re-executed
        # from a GPU process.
        def timed(e):
            best = float("inf")
            for _ in range(2):
                e.lanczos_prepare(x0, k_cfg)
                e.sync()
                dist.barrier()
                t = time.perf_counter()
                e.lanczos_run_steps(6)
                dt = torch.tensor([time.perf_counter() - t], dtype=torch.float64, device=tdev)
                dist.all_reduce(dt, op=dist.ReduceOp.MAX)
                best = min(best, float(dt.item()))
            return best / 6 * 1e3

        eng, t_gen = make_engine(overlap_exchange=0)
        if eng is None and args.transport == "rccl" and not rehearse and os.environ.get("LZX_BENCH_IPC_FALLBACK") == "1":
            # OPT-IN (round 5): the collective library would not start on this node and the caller allows the transport
            # that needs none (csrc/lzx_ipc.hip).  Not the default: that transport has only ever run with all ranks on one GPU, and
            # a scored run is no place for its first contact with xGMI.  Whatever transport runs is checked below before it is timed.
            print(f"[bench rank {rank}] no engine over RCCL: trying the peer-window transport (LZX_BENCH_IPC_FALLBACK=1)", file=sys.stderr, flush=True)
            args.transport = "ipc"
            tune["transport_fallback"] = "rccl -> ipc"
            eng, t_gen = make_engine(overlap_exchange=0)
        if eng is None:
            sys.exit(f"bench.py rank {rank}: could not build the engine (see stderr of the failing rank; LZX_BENCH_IPC_FALLBACK=1 allows "
                     f"the peer-window transport when RCCL cannot start)")

        def checked(e, what):
            """the transport's self-check on engine e; every rank learns the verdict before anybody enters the next collective"""
            res, ok = None, True
            try:
                res = transport_self_check(e)
            except Exception as exc:
                print(f"[bench rank {rank}] {what}: {exc}", file=sys.stderr, flush=True)
                ok = False
            return res if all_ok(ok) else None

        check_single = checked(eng, "single all-gather")
        if check_single is None:
            sys.exit(f"bench.py rank {rank}: the exchange failed its self-check (see stderr): nothing is measured over a transport that moves wrong data")
        m_single = measure(eng)
        m_single["self_check"] = check_single
        tune["single"] = timed(eng)
        out = line(eng, m_single, tune) if rank == 0 else None
        import threading
        trial_done = threading.Event()
        held = json.dumps(dict(out, config=dict(out["config"], exchange_tuning_ms_per_iter=dict(
            tune, overlapped="did not finish (watchdog): the line is the single all-gather mode's")))) if rank == 0 else None

        def watchdog(limit_s=float(os.environ.get("LZX_BENCH_TRIAL_LIMIT_S", "120"))):
            if not trial_done.wait(limit_s):
                print(f"[bench rank {rank}] the overlapped-exchange trial did not finish within {limit_s:.0f} s: reporting the "
                      f"single all-gather measurement and leaving (LZX_BENCH_SKIP_OVERLAP_TRIAL=1 skips the trial)",
                      file=sys.stderr, flush=True)
                if held is not None:
                    print(held, flush=True)
                os._exit(EXIT_TRIAL_HUNG)   # every rank: a collective that never completes is not a successful run

        if os.environ.get("LZX_BENCH_SKIP_OVERLAP_TRIAL") == "1":
            alt = None
        else:
            threading.Thread(target=watchdog, daemon=True).start()
            alt, _ = make_engine(overlap_exchange=1, sparse_exchange=1)
        # a rank count / graph that does not qualify for the overlapped mode on some rank: nothing to compare
        check_alt = checked(alt, "overlapped / sparse exchange") if alt is not None and all_ok(bool(alt.info()["pb_entries"])) else None
        if alt is not None and check_alt is not None:
            t_alt, ok = float("inf"), True
            try:
                t_alt = timed(alt)
            except Exception as exc:
                print(f"[bench rank {rank}] overlapped exchange failed: {exc}", file=sys.stderr, flush=True)
                ok = False
            if all_ok(ok):
                tune["overlapped"] = t_alt
                tune["overlapped_received_MB_per_rank"] = 8e-6 * alt.info()["exchange_recv"]
                # The trial's six iterations went through the two-chunk / sparse exchange itself (the self-check's products do not:
                # lzx_spmv_f64 re-lays x out locally): their leading coefficients must be the single all-gather's, whether or not
                # the overlapped form is then chosen (round 5, ADVICE r4: this comparison used to run only when it won)
                dev = float("inf")
                try:
                    a6, b6, _ = alt.lanczos_fetch(6)
                    head6 = np.concatenate([a6[:3], b6[:2]])
                    dev = float(np.max(np.abs(head6 - m_single["head"]) / np.maximum(np.abs(m_single["head"]), 1e-300)))
                except Exception as exc:
                    print(f"[bench rank {rank}] overlapped exchange: cannot read the trial's coefficients: {exc}", file=sys.stderr, flush=True)
                tune["overlapped_vs_single_coefficients_rel"] = dev
                trial_agrees = all_ok(dev < 1e-9)
                if not trial_agrees:
                    print(f"[bench rank {rank}] overlapped exchange REJECTED after its trial: leading coefficients differ from the single "
                          f"all-gather's by {dev:.2e}", file=sys.stderr, flush=True)
                if rank == 0:
                    out["config"]["exchange_tuning_ms_per_iter"] = dict(tune)
                if trial_agrees and (tune["overlapped"] < tune["single"] or os.environ.get("LZX_BENCH_FORCE_ALT") == "1"):   # (the variable: tests run the comparison below)
                    m_alt, ok = None, True
                    try:
                        m_alt = measure(alt)
                        m_alt["self_check"] = check_alt
                    except Exception as exc:
                        print(f"[bench rank {rank}] overlapped exchange failed in the timed run: {exc}", file=sys.stderr, flush=True)
                        ok = False
                    # the two exchange forms must describe the same decomposition: alpha_0..2, beta_0..1 to rounding
                    if ok and m_alt is not None:
                        dev = float(np.max(np.abs(m_alt["head"] - m_single["head"]) / np.maximum(np.abs(m_single["head"]), 1e-300)))
                        tune["overlapped_vs_single_coefficients_rel"] = dev
                        if not dev < 1e-9:
                            print(f"[bench rank {rank}] overlapped exchange REJECTED: leading coefficients differ from the single all-gather's by {dev:.2e}",
                                  file=sys.stderr, flush=True)
                            ok = False
                        if rank == 0:
                            out["config"]["exchange_tuning_ms_per_iter"] = dict(tune)
                    if all_ok(ok) and m_alt["elapsed"] < m_single["elapsed"]:
                        if rank == 0:
                            out = line(alt, m_alt, tune)
                        eng.close()
                        eng, alt = alt, None
        if alt is not None:
            alt.close()
        trial_done.set()
    if rank == 0:
        print(json.dumps(out), flush=True)

    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""tests/rccl_ranks.py -- run under torch.distributed.run with N >= 2 ranks, one GPU each: the row-partitioned Lanczos
loop over a real RCCL communicator (all-gather, two-double all-reduce, the two-chunk exchange with its sparse second
chunk: grouped ncclSend / ncclRecv) against the oracle.  Rank 0 prints "RCCL_RANKS_OK <world>" when every mode agrees.
Started by tests/test_gpu_parity.py::test_rccl_several_gpus (skipped on a one-GPU box)."""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import torch.distributed as dist
import __graft_entry__ as ge
from test_gpu_parity import REL_INF_TOL, check_leading_coefficients, check_recurrence, rel_inf, shift_weights

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
# LZX_RANKS_ONE_GPU=1: all ranks on GPU 0 -- a probe of whether RCCL lets several ranks of one
# communicator share a device (torch's own group runs over gloo then).  It does not: ncclCommInitRank fails with "Duplicate
# GPU detected"; the probe prints the refusal and exits 0, so the answer is on record (profiles/r4_rccl_one_gpu.txt).
one_gpu = os.environ.get("LZX_RANKS_ONE_GPU") == "1"
if one_gpu:
    local = 0
torch.cuda.set_device(local)
if one_gpu:
    dist.init_process_group("gloo", rank=rank, world_size=world)
else:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
pkg, O = ge.load_pkg(), ge.load_oracle()
if one_gpu:
    eng = pkg.Engine(0)
    uid = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        uid.copy_(torch.from_numpy(pkg.Engine.unique_id()))
    dist.broadcast(uid, 0)
    try:
        eng.comm_init_rank(uid.numpy(), rank, world)
        print(f"[rccl_ranks] rank {rank}: RCCL ACCEPTED {world} ranks on one GPU", flush=True)
    except pkg.LzxError as exc:
        print(f"[rccl_ranks] rank {rank}: RCCL_ONE_GPU_REFUSED: {exc}", flush=True)
        sys.exit(0)
rp, ci = O.gen_er(600000, 3000000, 77)
n, k = len(rp) - 1, 10
x0 = np.ones(n)
a_ref, b_ref, Q_ref, xn_ref = O.lanczos(rp, ci, k, x0, q_colmajor=True)
ref = shift_weights(O, a_ref, b_ref, xn_ref) @ Q_ref
x = np.random.default_rng(9).random(n)
y_ref = O.spmv(rp, ci, x)
ok = True
# the safest forms first (single all-gather), then the two-chunk exchange -- over RCCL it is on request only until this very
# script has passed once on two or more GPUs -- dense, then with the sparse second chunk (grouped ncclSend / ncclRecv on the
# exchange stream's own communicator; its send / receive counts were checked pairwise when the graph was reshaped)
for mode in (dict(propagation_blocking=0), dict(propagation_blocking=1, hub_entries=1024, overlap_exchange=0),
             dict(propagation_blocking=1, hub_entries=1024, overlap_exchange=1, sparse_exchange=0),
             dict(propagation_blocking=1, hub_entries=1024, overlap_exchange=1, sparse_exchange=1),
             dict(propagation_blocking=1, hub_entries=1024, overlap_exchange=1, lazy_normalisation=0),
             dict(propagation_blocking=0, reorthogonalise=1), dict(propagation_blocking=1, hub_entries=1024, basis_fp32=1)):
    eng = pkg.Engine(local, **mode)
    uid = torch.zeros(128, dtype=torch.uint8, device="cpu" if one_gpu else "cuda")
    if rank == 0:
        uid.copy_(torch.from_numpy(pkg.Engine.unique_id()))
    dist.broadcast(uid, 0)
    eng.comm_init_rank(uid.cpu().numpy(), rank, world)
    eng.set_graph_csr(rp, ci)
    gi = eng.info()
    assert gi["world"] == world and gi["rank"] == rank
    assert np.allclose(eng.spmv(x), y_ref, rtol=1e-13, atol=0), mode
    a, b, Q, xn, st = eng.lanczos(x0, k)
    check_leading_coefficients(a, b, a_ref, b_ref, ("rccl", world, mode), n=n)
    check_recurrence(O, rp, ci, a, b, Q, ("rccl", world, mode))
    assert rel_inf(eng.multout(shift_weights(O, a, b, xn)), ref) <= (1e-6 if mode.get("basis_fp32") else REL_INF_TOL), mode
    if rank == 0:
        print(f"[rccl_ranks] world={world} {mode}: exchange_recv={gi['exchange_recv']} chunk0={gi['exchange_chunk0']} "
              f"loop {st['loop_ms']:.2f} ms (comm {st['comm_ms']:.2f})", flush=True)
    eng.close()
    dist.barrier()
if rank == 0:
    print(f"RCCL_RANKS_OK {world}", flush=True)
dist.destroy_process_group()

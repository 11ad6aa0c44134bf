"""tests/ipc_ranks.py -- run under torch.distributed.run (gloo: the side channel only) with N >= 2 ranks that may all sit on
ONE GPU: the row-partitioned Lanczos loop over the peer-window transport (csrc/lzx_ipc.hip: receive buffers mapped across
processes, data pushed by the sender's kernel, sequence numbers in device memory, mailbox all-reduce) against the oracle,
in every exchange form the loop has.  Rank 0 prints "IPC_RANKS_OK <world>" when every mode agrees on every rank.
Started by tests/test_gpu_parity.py::test_peer_windows_across_processes (2 and 4 processes on the box's one GPU)."""
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

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
n_dev = torch.cuda.device_count()
device = int(os.environ.get("LOCAL_RANK", "0")) % max(n_dev, 1) if os.environ.get("LZX_IPC_SPREAD") == "1" else 0
dist.init_process_group("gloo", rank=rank, world_size=world)
pkg, O = ge.load_pkg(), ge.load_oracle()


def wire(eng):
    mine = torch.from_numpy(eng.comm_ipc_export().copy())
    blobs = [torch.zeros(pkg.Engine.IPC_BLOB, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(blobs, mine)
    eng.comm_ipc_init(torch.cat(blobs).numpy(), rank, world)


rp, ci = O.gen_er(200000, 1000000, 77)
n, k = len(rp) - 1, 10
x0 = np.ones(n)
a_ref, b_ref, Q_ref, xn_ref = O.lanczos(rp, ci, k, x0, q_colmajor=True)
ref = shift_weights(O, a_ref, b_ref, xn_ref) @ Q_ref
x = np.random.default_rng(9).random(n)
y_ref = O.spmv(rp, ci, x)
coeffs = {}
for mode in (dict(propagation_blocking=0), dict(propagation_blocking=1, hub_entries=1024, overlap_exchange=0),
             dict(propagation_blocking=1, hub_entries=1024, overlap_exchange=1, sparse_exchange=0),
             dict(propagation_blocking=1, hub_entries=1024),   # the transport's defaults: two chunks, the second one sparse
             dict(propagation_blocking=1, hub_entries=1024, overlap_exchange=1, lazy_normalisation=0),
             dict(propagation_blocking=0, reorthogonalise=1), dict(propagation_blocking=1, hub_entries=1024, basis_fp32=1),
             dict(propagation_blocking=0, exchange_fp32=1)):
    eng = pkg.Engine(device, **mode)
    wire(eng)
    eng.set_graph_csr(rp, ci)
    gi = eng.info()
    assert gi["world"] == world and gi["rank"] == rank
    fp32 = bool(mode.get("basis_fp32") or mode.get("exchange_fp32"))
    if not mode.get("exchange_fp32"):
        assert np.allclose(eng.spmv(x), y_ref, rtol=1e-13, atol=0), mode
    a, b, Q, xn, st = eng.lanczos(x0, k)
    if not fp32:
        check_leading_coefficients(a, b, a_ref, b_ref, ("ipc", world, mode), n=n)
        check_recurrence(O, rp, ci, a, b, Q, ("ipc", world, mode))
    assert rel_inf(eng.multout(shift_weights(O, a, b, xn)), ref) <= (1e-6 if fp32 else REL_INF_TOL), mode
    # every rank holds the same coefficients, bit for bit (the mailbox sums in rank order on every rank)
    mine = torch.from_numpy(np.concatenate([a, b]).copy())
    everyone = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(everyone, mine)
    assert all(torch.equal(e, everyone[0]) for e in everyone), ("coefficients differ between ranks", mode)
    # a second decomposition on the same handle, advanced in chunks: the same bits as in one go
    eng.lanczos_prepare(x0, k)
    eng.lanczos_run_steps(4)
    eng.lanczos_run_steps(k)
    a2, b2, _ = eng.lanczos_fetch(k)
    assert np.array_equal(a2, a) and np.array_equal(b2, b), ("chunked run differs", mode)
    if rank == 0:
        print(f"[ipc_ranks] world={world} {mode}: exchange_recv={gi['exchange_recv']} chunk0={gi['exchange_chunk0']} "
              f"loop {st['loop_ms']:.2f} ms (comm {st['comm_ms']:.2f})", flush=True)
    eng.close()
    dist.barrier()
# what the mailbox all-reduce costs between processes (here: ranks sharing one GPU; profiles/r4_ipc_one_gpu.txt)
eng = pkg.Engine(device)
wire(eng)
us = eng.allreduce_latency(2000)
if rank == 0:
    print(f"[ipc_ranks] world={world}: two-double all-reduce through the mailboxes {us:.2f} us each (2000 back to back)", flush=True)
assert 0.0 < us < 5000.0
eng.close()
dist.barrier()
# a larger graph on fresh handles: the receive buffers are published again; an R-MAT graph with split rows
eng = pkg.Engine(device, propagation_blocking=1)
wire(eng)
eng.gen_rmat(18, 200000, 3000000, 5)
rp2, ci2 = eng.get_graph_csr()
x2 = np.random.default_rng(10).random(len(rp2) - 1)
assert np.allclose(eng.spmv(x2), O.spmv(rp2, ci2, x2), rtol=1e-12, atol=0)
eng.set_graph_csr(rp, ci)      # and a second graph on the SAME wired handle
assert np.allclose(eng.spmv(x), y_ref, rtol=1e-13, atol=0)
eng.close()
dist.barrier()
# the sharded hand-over across processes (option sharded_ingest): every rank sweeps the seeded generator and keeps its OWN rows
# only; the sparse second chunk's send lists come from those rows through the matrix's symmetry and are checked pairwise by
# the transport at the hand-over (a disagreement is LZX_ERR_STATE on every rank).  Same bits as the whole-graph hand-over.
rp3, ci3 = O.gen_rmat(18, 200000, 3000000, 5)
x3 = np.random.default_rng(11).random(len(rp3) - 1)
y3 = O.spmv(rp3, ci3, x3)
got = {}
for sharded in (0, 3):
    eng = pkg.Engine(device, propagation_blocking=1, hub_entries=1024, sharded_ingest=sharded)
    wire(eng)
    eng.gen_rmat(18, 200000, 3000000, 5)
    gi = eng.info()
    assert gi["nnz"] == len(ci3) and gi["exchange_chunk0"] > 0 and gi["exchange_recv"] < (world - 1) * gi["exchange_slice"], gi
    y = eng.spmv(x3)
    assert np.allclose(y, y3, rtol=1e-12, atol=0)
    a, b, _, _, _ = eng.lanczos(np.ones(len(rp3) - 1), 8, want_q=False)
    got[sharded] = (gi["nnz_local"], gi["exchange_recv"], y, a, b)
    if sharded:
        try:
            eng.get_graph_csr()
            raise AssertionError("a sharded rank handed out a whole graph")
        except pkg.LzxError:
            pass
    eng.close()
    dist.barrier()
assert got[0][0] == got[3][0] and got[0][1] == got[3][1] and all(np.array_equal(p, q) for p, q in zip(got[0][2:], got[3][2:]))
if rank == 0:
    print(f"[ipc_ranks] world={world}: sharded hand-over == whole-graph hand-over bit for bit (nnz_local {got[3][0]}, receives {got[3][1]} doubles)", flush=True)
# a peer that never arrives is an error after the deadline, not a hang: rank 0 enters a collective alone
import time
os.environ["LZX_IPC_TIMEOUT_MS"] = "400"
eng = pkg.Engine(device)
wire(eng)
if rank == 0:
    t = time.time()
    try:
        eng.allreduce_latency(4)
        raise AssertionError("an all-reduce without the peers returned")
    except pkg.LzxError as exc:
        waited = time.time() - t
        assert "did not arrive" in str(exc) and waited < 10.0, (str(exc), waited)
        print(f"[ipc_ranks] world={world}: a collective entered by rank 0 alone failed after {waited:.2f} s: {str(exc)[-110:]}", flush=True)
dist.barrier()
# ... and it is FATAL for the communicator (round 5): the ranks' sequence numbers no longer describe the same operations,
# so rank 0 refuses every later collective at once, and the peers -- whose windows it poisoned -- fail fast with "gave up", not
# after a deadline of their own and never with another operation's mailbox values
t = time.time()
try:
    eng.allreduce_latency(4)
    raise AssertionError("a collective on a broken communicator returned")
except pkg.LzxError as exc:
    assert ("is broken" in str(exc)) if rank == 0 else ("gave up" in str(exc)), (rank, str(exc))
    assert time.time() - t < 10.0
try:
    eng.allreduce_latency(4)
    raise AssertionError("a collective on a broken communicator returned")
except pkg.LzxError as exc:
    assert "is broken" in str(exc), (rank, str(exc))
dist.barrier()
eng.close()
dist.barrier()
if rank == 0:
    print(f"IPC_RANKS_OK {world}", flush=True)
dist.destroy_process_group()

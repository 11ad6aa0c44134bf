"""The N > 1 Lanczos loop with the SAME partition, layout and exchange pattern as liblzx.so's multi-rank
path (csrc/lzx_api.hip: lanczos_loop; csrc/lzx_comm.hip), stated over torch.distributed so that it can be
rehearsed on CPU with the gloo backend.  The local SpMV here is the test oracle's row sums -- this file is
test infrastructure, not a product path."""
import numpy as np
import torch
import torch.distributed as dist


def run_rank(pkg_partition, row_ptr, col_idx, x0, k, xs0=None, lazy=False, sparse=False):
    """Returns (alpha, beta, gathered full-length q vectors (k, n) in the caller's order) on every rank.
    lazy: the product's default at N > 1 -- the unnormalised vector is exchanged and ONE 2-double all-reduce per
    iteration carries u.(A u) and ||u||^2 (csrc/lzx_kernels.hip: k_lazy_update).
    sparse (with xs0): chunk 1 travels point to point -- every rank derives, from the whole graph it holds, which
    chunk-1 entries of each peer its rows reference and which of its own entries each peer's rows reference, packs per
    peer and sends (csrc/lzx_graph.hip: k_sx_mark; csrc/lzx_comm.hip: lzx_comm_sparse_chunk1).  Entries nobody on
    this rank references stay 0 in its copy: they are never read."""
    P = pkg_partition
    world, rank = dist.get_world_size(), dist.get_rank()
    n = len(row_ptr) - 1
    order = P.degree_order(row_ptr)
    L = P.slice_len(n, world)                   # local vector length (all owned rows)
    X = P.exchange_len(row_ptr, world)          # exchanged prefix: vertices with an edge only
    deg = np.diff(row_ptr.astype(np.int64))
    n_active = int((deg > 0).sum())
    X0 = X if xs0 is None else xs0              # two-chunk exchange layout: [world][X0] then [world][X - X0]
    pos_of_old = np.full(n, -1, dtype=np.int64)  # exchange-layout position; isolated vertices are never gathered
    pos_of_old[order[:n_active]] = P.chunked_positions(n, world, X, X0)[:n_active]
    mine = P.local_vertices(order, world, rank)  # caller's ids of my rows, local order
    rp = row_ptr.astype(np.int64)
    # my rows' columns translated to exchange-layout positions, caller's column order kept
    seg = [pos_of_old[col_idx[rp[o]:rp[o + 1]].astype(np.int64)] for o in mine]

    def spmv_local(xfull):
        out = np.zeros(L)
        for l, cols in enumerate(seg):
            acc = 0.0
            for c in cols:                       # left-to-right, as serial/lib/SPMV.cc
                acc += xfull[c]
            out[l] = acc
        return out

    def allreduce(v):
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t.item())

    def allgather(loc, count, first=0):
        outs = [torch.empty(count, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(outs, torch.from_numpy(np.ascontiguousarray(loc[first:first + count])))
        return torch.cat(outs).numpy()

    # sparse chunk 1: what I need from every peer and what every peer needs from me, from the graph alone
    L1 = X - X0
    need = want = None
    if sparse and L1 > 0:
        c1 = world * X0
        need = [np.zeros(L1, dtype=bool) for _ in range(world)]      # need[r][l]: my rows reference peer r's entry X0 + l
        want = [np.zeros(L1, dtype=bool) for _ in range(world)]      # want[p][l]: rank p's rows reference MY entry X0 + l
        for p in range(world):
            for o in P.local_vertices(order, world, p):
                cols = pos_of_old[col_idx[rp[o]:rp[o + 1]].astype(np.int64)]
                cols = cols[cols >= c1] - c1
                owner, l = cols // L1, cols % L1
                if p == rank:
                    for r in range(world):
                        need[r][l[owner == r]] = True
                want[p][l[owner == rank]] = True

    def exchange(loc):
        """chunk 0 of every slice, then chunk 1: two all-gathers into one buffer (in the product the SpMV starts on
        chunk 0 while chunk 1 is still travelling); or chunk 1 packed per peer and sent point to point"""
        if X0 == X:
            return allgather(loc, X)
        if need is None:
            return np.concatenate([allgather(loc, X0), allgather(loc, L1, X0)])
        head = allgather(loc, X0)
        tail = np.zeros(world * L1)
        mine1 = loc[X0:X0 + L1]
        tail[rank * L1:(rank + 1) * L1][need[rank]] = mine1[need[rank]]          # my own segment: no transport
        ops, bufs = [], {}
        for peer in range(world):
            if peer == rank:
                continue
            send = torch.from_numpy(np.ascontiguousarray(mine1[want[peer]]))
            bufs[peer] = torch.empty(int(need[peer].sum()), dtype=torch.float64)
            if send.numel():
                ops.append(dist.P2POp(dist.isend, send, peer))
            if bufs[peer].numel():
                ops.append(dist.P2POp(dist.irecv, bufs[peer], peer))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for peer, b in bufs.items():
            tail[peer * L1:(peer + 1) * L1][need[peer]] = b.numpy()
        return np.concatenate([head, tail])

    xn = np.sqrt(np.sum(x0 * x0))
    q = np.zeros(L)
    q[:len(mine)] = x0[mine] / xn               # every rank holds all of x0: no exchange to start
    xfull = np.zeros(world * X)
    act = pos_of_old >= 0
    xfull[pos_of_old[act]] = x0[act] / xn
    q_prev = np.zeros(L)
    alpha, beta = np.zeros(k), np.zeros(max(k - 1, 1))
    Q = np.zeros((k, n))
    io_pos = np.empty(n, dtype=np.int64)         # result layout: stride L
    io_pos[order] = P.positions(n, world)
    def allreduce2(a, b):
        t = torch.tensor([a, b], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t[0].item()), float(t[1].item())

    if lazy:
        u, own_sq = q.copy(), 0.0                # u_0 = q_0 (already normalised)
        for j in range(k):
            w = spmv_local(xfull)                # A u_j on the exchanged, unnormalised u_j
            D, B = allreduce2(float(w @ u), own_sq)
            if j == 0:
                b, alpha[j], qj = 1.0, D, u
            else:
                b = np.sqrt(B)
                beta[j - 1], alpha[j], qj, w = b, D / B, u / b, w / b
            Q[j] = allgather(qj, L)[io_pos]      # result gather (not part of the iteration's exchange)
            if j == k - 1:
                break
            t = w - alpha[j] * qj
            if j > 0:
                t = t - b * q_prev
            q_prev, u, own_sq = qj, t, float(t @ t)
            xfull = exchange(u)
        return alpha, beta[:k - 1], Q, xn

    for j in range(k):
        Q[j] = allgather(q, L)[io_pos]           # result gather (not part of the iteration's exchange)
        v = spmv_local(xfull)
        alpha[j] = allreduce(float(v @ q))
        if j == k - 1:
            break
        v = v - alpha[j] * q
        if j > 0:
            v = v - beta[j - 1] * q_prev
        beta[j] = np.sqrt(allreduce(float(v @ v)))
        q_prev, q = q, v / beta[j]
        xfull = exchange(q)                      # the per-iteration exchange: the prefix with edges only
    return alpha, beta[:k - 1], Q, xn

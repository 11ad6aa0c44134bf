"""TEST INFRASTRUCTURE (moved out of the product package in round 4).  Row partition of the multi-GPU Lanczos loop, stated in numpy (host-side mirror of the rules in
csrc/lzx_graph.hip: lzx_graph_prepare steps 1-2).  Used by the gloo tests (tests/test_distributed_gloo.py, tests/dist_model.py) that
rehearse the N > 1 exchange pattern on CPU.

Rules (identical on every rank, no communication needed to agree on them):
  * vertices are ranked by degree, descending, ties by the caller's id (stable);
  * degree rank r is owned by rank r % world, at local row r // world;
  * a rank's slice of a full-length vector is n_loc_pad = round_up(ceil(n / world), 64) long; degree rank r sits at
    position (r % world) * n_loc_pad + r // world;
  * the per-iteration exchange only moves vertices that have an edge: they are the first n_active degree ranks, hence
    a prefix of every slice, xs = round_up(ceil(n_active / world), 64) long (R-MAT graphs are ~40 % isolated
    vertices); in that exchange layout degree rank r sits at (r % world) * xs + r // world.
Replaces the reference's split at rows0 = 0.5 * n (parallel-two-cards/lib/cu_lanczos.cu:62-64), which
balances rows, not work.
"""
from __future__ import annotations

import numpy as np

SLICE = 64


def degree_order(row_ptr: np.ndarray) -> np.ndarray:
    """order[r] = caller's id of the vertex with degree rank r."""
    deg = np.diff(np.asarray(row_ptr).astype(np.int64))
    return np.argsort(-deg, kind="stable")


def slice_len(n: int, world: int) -> int:
    per = -(-n // world)
    return -(-per // SLICE) * SLICE


def exchange_len(row_ptr: np.ndarray, world: int) -> int:
    """xs: how many entries of each rank's slice the per-iteration all-gather moves."""
    deg = np.diff(np.asarray(row_ptr).astype(np.int64))
    n_active = int((deg > 0).sum())
    per = -(-n_active // world)
    return min(max(SLICE, -(-per // SLICE) * SLICE), slice_len(len(deg), world))


def exchange_positions(row_ptr: np.ndarray, world: int) -> np.ndarray:
    """pos[r] = position of degree rank r in the exchange layout (meaningful for r < n_active)."""
    n = len(row_ptr) - 1
    r = np.arange(n, dtype=np.int64)
    return (r % world) * exchange_len(row_ptr, world) + r // world


PB_CB = 16384


def chunk0_len(row_ptr: np.ndarray, world: int, hub: int = 16384) -> int:
    """xs0: the exchange is cut into the first xs0 entries of every slice (high-degree end) and the rest, laid out
    [world][xs0] then [world][xs - xs0], so that the blocked SpMV can start on chunk 0 while chunk 1 travels.
    xs0 = xs (one chunk) when the slices are too short for that."""
    xs = exchange_len(row_ptr, world)
    x0 = -(-max(xs // 8, -(-hub // world)) // PB_CB) * PB_CB
    return x0 if x0 < xs else xs


def chunked_positions(n: int, world: int, xs: int, xs0: int) -> np.ndarray:
    """pos[r] = position of degree rank r in the two-chunk exchange layout."""
    r = np.arange(n, dtype=np.int64)
    p, l = r % world, r // world
    return np.where(l < xs0, p * xs0 + l, world * xs0 + p * (xs - xs0) + (l - xs0))


def positions(n: int, world: int) -> np.ndarray:
    """pos[r] = position of degree rank r in the full-length exchange layout."""
    r = np.arange(n, dtype=np.int64)
    return (r % world) * slice_len(n, world) + r // world


def local_vertices(order: np.ndarray, world: int, rank: int) -> np.ndarray:
    """Caller's ids of the rows rank `rank` owns, in local row order."""
    return order[rank::world]


def nnz_per_rank(row_ptr: np.ndarray, world: int) -> np.ndarray:
    deg = np.diff(np.asarray(row_ptr).astype(np.int64))
    order = degree_order(row_ptr)
    return np.array([deg[order[p::world]].sum() for p in range(world)])


def _chunk1_codes(row_ptr, world, xs, xs0):
    """code1[o] = position of vertex o inside chunk 1 of the two-chunk exchange layout (owner * L1 + l), or -1 when the vertex
    lies in chunk 0 / has no edge (lzx_graph.hip: k_rank_maps)."""
    n = len(row_ptr) - 1
    order = degree_order(row_ptr)
    deg = np.diff(np.asarray(row_ptr).astype(np.int64))
    r = np.arange(n, dtype=np.int64)
    p, l = r % world, r // world
    L1 = xs - xs0
    code_of_rank = np.where((l >= xs0) & (deg[order] > 0), p * L1 + (l - xs0), -1)
    code1 = np.empty(n, dtype=np.int64)
    code1[order] = code_of_rank
    owner = np.empty(n, dtype=np.int64)
    owner[order] = p
    return code1, owner, L1


def sparse_lists_whole(row_ptr, col_idx, world, rank, xs, xs0):
    """The sparse second chunk's two sides derived from the WHOLE graph (lzx_graph.hip: k_sx_mark):
    ref[r * L1 + l]  : a row of `rank` has an entry in the column that is chunk-1 entry l of rank r   (what it receives)
    want[p * L1 + l] : a row of rank p has an entry in the column that is chunk-1 entry l of `rank`   (what it sends to p)"""
    code1, owner, L1 = _chunk1_codes(row_ptr, world, xs, xs0)
    n = len(row_ptr) - 1
    deg = np.diff(np.asarray(row_ptr).astype(np.int64))
    row_of_entry = np.repeat(np.arange(n, dtype=np.int64), deg)
    cc = code1[np.asarray(col_idx).astype(np.int64)]
    ok = cc >= 0
    ref = np.zeros(world * L1, dtype=bool)
    want = np.zeros(world * L1, dtype=bool)
    mine = ok & (owner[row_of_entry] == rank)
    ref[cc[mine]] = True
    to_me = ok & (cc // max(L1, 1) == rank)
    want[owner[row_of_entry[to_me]] * L1 + cc[to_me] % max(L1, 1)] = True
    return ref, want


def sparse_lists_own(row_ptr, col_idx, world, rank, xs, xs0):
    """The same two sides from the rank's OWN rows alone (sharded hand-over, lzx_graph.hip: k_sx_mark_own): `ref` as above; `want`
    through the symmetry of the matrix -- a row of rank p has an entry in my column v exactly when my row v has an entry in a
    column that rank p owns."""
    code1, owner, L1 = _chunk1_codes(row_ptr, world, xs, xs0)
    n = len(row_ptr) - 1
    deg = np.diff(np.asarray(row_ptr).astype(np.int64))
    row_of_entry = np.repeat(np.arange(n, dtype=np.int64), deg)
    cols = np.asarray(col_idx).astype(np.int64)
    own = owner[row_of_entry] == rank            # the only entries this rank holds
    ref = np.zeros(world * L1, dtype=bool)
    want = np.zeros(world * L1, dtype=bool)
    cc = code1[cols[own]]
    ref[cc[cc >= 0]] = True
    my_code = code1[row_of_entry[own]]           # the row's own place in chunk 1 (its owner is `rank`)
    in1 = my_code >= 0
    want[owner[cols[own][in1]] * L1 + my_code[in1] % max(L1, 1)] = True
    return ref, want

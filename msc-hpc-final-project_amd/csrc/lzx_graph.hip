// lzx_graph.hip -- graph hand-over and the one-time reshaping of the caller's CSR into the layout the
// SpMV kernel streams:
//   1. vertices are ranked by degree (descending, ties by caller's id) with a device radix sort;
//   2. degree rank r is owned by rank r % world at local row r / world, so every rank holds the same
//      mix of heavy and light rows and local rows are again sorted by degree;
//   3. local rows with more than LZX_LONG_ROW entries become "split rows" (wave-sized items);
//      the rest form a sliced-ELL body with 64-row slices whose width is the slice's first (largest)
//      degree -- almost no padding because neighbours in the order have near-equal degree;
//   4. column indices are rewritten to codes: c < hub means "x value staged in LDS slot c" (the hub
//      highest-degree vertices), otherwise hub + position in the full-length exchange layout.
// Within a row the caller's ascending column order is kept, so the body reproduces the reference's
// summation order exactly (serial/lib/SPMV.cc:24-27).
//
// Also here: device-side ingest of an edge list (replaces the std::set build of
// adjMatrix::populate_sparse_matrix, parallel-final/lib/adjMatrix.cc:21-46) and the seeded ER / R-MAT
// generators used by bench.py (integer spec shared with oracle/lanczos_oracle.c).
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <string>

#include "lzx_internal.h"

template <typename T>
static int dev_alloc(T **p, u64 count)
{
    *p = nullptr;
    if (count == 0) count = 1;
    LZX_HIP(hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T)));
    return LZX_OK;
}

template <typename T>
static void dev_free(T *&p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}

int lzx_graph_release(lzx_ctx *c)
{
    c->iso_cols_filled = 1;
    c->iso_on = c->iso_filled = c->basis_u = false;   // the factored rows belong to a decomposition on the old graph

    dev_free(c->d_row_ptr);
    dev_free(c->d_col_idx);
    dev_free(c->d_shard_deg);
    c->shard = lzx_ctx::lzx_key_source();
    c->sharded = false;
    dev_free(c->d_gidx_of_old);
    dev_free(c->d_sx_send_idx);
    dev_free(c->d_sx_sendbuf);
    dev_free(c->d_sx_map);
    dev_free(c->d_xf32_send);
    dev_free(c->d_xf32_full);
    c->xfp32 = false;
    c->sparse = false;
    c->xc1 = 0;
    c->sx_recv_off.clear();
    c->sx_send_off.clear();
    c->sx_send_hash.clear();
    c->sx_recv_hash.clear();
    dev_free(c->d_slice_off);
    dev_free(c->d_slice_w);
    dev_free(c->d_slice_perm);
    c->ns_wide = c->ns_w8 = c->ns_w4 = 0;
    c->h_slice_w0.clear();
    dev_free(c->d_sell_cols);
    dev_free(c->d_long_cols);
    dev_free(c->d_item_beg);
    dev_free(c->d_item_len);
    dev_free(c->d_item_first);
    dev_free(c->d_long_partial);
    dev_free(c->d_v);
    dev_free(c->d_u[0]);
    dev_free(c->d_u[1]);
    dev_free(c->d_Q);
    // the other per-graph buffers of the loop's optional forms are sized by this graph's ldq / n_loc_pad too: a larger graph
    // on the same handle must not inherit them (fp32-stored basis, its three live vectors, the convergence monitor's answers)
    dev_free(c->d_Qf);
    for (double *&r : c->d_ring) dev_free(r);
    for (double *&y : c->d_ymon) dev_free(y);
    c->qf_cols = 0;
    c->qf32 = false;
    c->k_done = 0;
    c->ymon_valid = 0;
    lzx_comm_ipc_unpublish(c);   // peer windows: the peers' mappings of this graph's receive buffers are closed first
    dev_free(c->d_xbuf);
    dev_free(c->d_ybuf);
    dev_free(c->d_io);
    dev_free(c->d_partials);
    dev_free(c->d_partials2);
    dev_free(c->d_partials3);
    lzx_pb_release(c);
    c->q_cols = 0;
    c->k_last = 0;
    c->k_prep = 0;   // a prepared start vector lived in the buffers just freed
    c->n = c->nnz = 0;
    return LZX_OK;
}

// --------------------------------------------------------------------------------------------------
// kernels of the reshaping pass
__global__ void k_degrees(const u64 *row_ptr, u32 *deg, u32 *ids, u64 n)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        deg[i] = (u32)(row_ptr[i + 1] - row_ptr[i]);
        ids[i] = (u32)i;
    }
}

// Blocked mode: rows of EQUAL degree are ordered by their number of staged (top-`hub`) columns, so that the 64 rows of a
// staged-only slice are about equally wide (degree alone leaves that count Poisson-scattered: a low-degree slice was
// two to three times as wide as its mean row).  The staged set itself is fixed by the first ranking.
__global__ void k_rank_of_old(const u32 *sorted_ids, u32 *rank_of_old, u64 n)
{
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) rank_of_old[sorted_ids[r]] = (u32)r;
}

// one wavefront per ranked vertex (strided over the grid): key = degree << 32 | staged count; the staged vertices
// themselves keep their order (low word counts down from 2^32 - 1, above any count)
__global__ void __launch_bounds__(64)
k_staged_key(const u64 *row_ptr, const u32 *col_idx, const u32 *sorted_ids, const u32 *sorted_deg, const u32 *rank_of_old,
             u64 n_active, u64 n, u32 hub, int count_major, u64 *key)
{
    const u32 lane = threadIdx.x;
    for (u64 r = blockIdx.x; r < n; r += gridDim.x) {
        const u32 d = sorted_deg[r];
        u32 low = 0;
        if (r < hub) {
            if (lane == 0) key[r] = count_major ? ~(u64)r : ((u64)d << 32) | (0xffffffffu - (u32)r);
            continue;
        } else if (r < n_active) {
            const u64 base = row_ptr[sorted_ids[r]];
            u32 cnt = 0;
            for (u32 k = lane; k < d; k += 64) cnt += rank_of_old[col_idx[base + k]] < hub ? 1u : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
            low = cnt;
        }
        if (lane == 0) key[r] = count_major ? ((u64)low << 32) | d : ((u64)d << 32) | low;
    }
}

__global__ void k_degree_of(const u32 *deg_of_old, const u32 *sorted_ids, u32 *sorted_deg, u64 n)
{
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) sorted_deg[r] = deg_of_old[sorted_ids[r]];
}

// degree rank r -> position in the exchange layout and column code
__global__ void k_rank_maps(const u32 *sorted_ids, u32 *gidx_of_old, u32 *code_of_old, u64 n,
                            u32 world, u32 n_loc_pad, u32 xs, u32 xs0, u32 hub)
{
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const u32 o = sorted_ids[r];
    const u32 p = (u32)(r % world), l = (u32)(r / world);
    gidx_of_old[o] = p * n_loc_pad + l;   // hand-over / result layout
    // exchange layout (meaningful for degree > 0): chunk 0 = [world][xs0], chunk 1 = [world][xs - xs0]
    const u32 x = l < xs0 ? p * xs0 + l : world * xs0 + p * (xs - xs0) + (l - xs0);
    code_of_old[o] = (r < hub) ? (u32)r : hub + x;
}

// ---- sparse exchange of chunk 1 -------------------------------------------------------------------------------
// Every rank holds the whole graph, so each one derives BOTH sides from it without talking to anybody: which chunk-1
// entries of every peer its own rows reference (ref: its receive layout), and which of its own chunk-1 entries every
// peer's rows reference (want: its send lists).  One wavefront per row of the WHOLE graph, rows strided over the grid.
//   ref [r * L1 + l] = 1: a row of rank `me` has an entry in the column that is local entry xs0 + l of rank r
//   want[p * L1 + l] = 1: a row of rank p has an entry in the column that is local entry xs0 + l of rank `me`
__global__ void __launch_bounds__(64)
k_sx_mark(const u64 *row_ptr, const u32 *col_idx, const u32 *sorted_ids, const u32 *code_of_old, u64 n_active, u32 world, u32 me,
          u32 hub, u32 xs0, u32 L1, uint8_t *ref, uint8_t *want)
{
    const u32 lane = threadIdx.x;
    const u32 c1 = hub + world * xs0;   // first chunk-1 code
    for (u64 r = blockIdx.x; r < n_active; r += gridDim.x) {
        const u32 p = (u32)(r % world);
        const u32 o = sorted_ids[r];
        const u64 beg = row_ptr[o], end = row_ptr[o + 1];
        for (u64 k = beg + lane; k < end; k += 64) {
            const u32 code = code_of_old[col_idx[k]];
            if (code < c1) continue;
            const u32 x = code - c1, owner = x / L1, l = x % L1;
            if (p == me) ref[x] = 1;
            if (owner == me) want[(size_t)p * L1 + l] = 1;
        }
    }
}

__global__ void k_widen_u8(const uint8_t *in, u64 count, u32 *out)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = in[i];
}

// packed positions: codes of referenced chunk-1 columns move to c1 + newpos; map[newpos] = hand-over position
__global__ void k_sx_remap(const u32 *sorted_ids, u32 *code_of_old, u64 n_active, u32 world, u32 hub, u32 xs0, u32 L1, u32 n_loc_pad,
                           const uint8_t *ref, const u32 *newpos, u32 *map)
{
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_active) return;
    const u32 o = sorted_ids[r];
    const u32 c1 = hub + world * xs0;
    const u32 code = code_of_old[o];
    if (code < c1) return;
    const u32 x = code - c1;
    if (ref[x]) {
        code_of_old[o] = c1 + newpos[x];
        map[newpos[x]] = (x / L1) * n_loc_pad + xs0 + (x % L1);
    } else {
        code_of_old[o] = c1;   // never referenced by this rank's rows
    }
}

// send list: entry (p, l) with want set -> idx[sendpos] = xs0 + l
__global__ void k_sx_lists(const uint8_t *want, const u32 *sendpos, u64 count, u32 L1, u32 xs0, u32 *idx)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count && want[i]) idx[sendpos[i]] = xs0 + (u32)(i % L1);
}

__global__ void k_count_active(const u32 *sorted_deg, u64 n, unsigned long long *count)
{
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    // degrees are sorted descending: the active vertices are a prefix; one thread finds its end
    if (r < n && sorted_deg[r] > 0 && (r + 1 == n || sorted_deg[r + 1] == 0)) *count = r + 1;
}

// local row l <-> degree rank l * world + rank
__global__ void k_local_rows(const u32 *sorted_ids, const u32 *sorted_deg, u32 *old_of_local,
                             u32 *deg_local, u32 n_loc_real, u32 world, u32 rank)
{
    const u32 l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n_loc_real) return;
    const u64 r = (u64)l * world + rank;
    old_of_local[l] = sorted_ids[r];
    deg_local[l] = sorted_deg[r];
}

// Body: thread = local row; writes its row's codes as 16-byte packets, packet p of lane l of slice s
// at ((uint4*)(cols + slice_off[s]))[p * 64 + l].
__global__ void k_fill_sell(const u64 *row_ptr, const u32 *col_idx, const u32 *code_of_old,
                            const u32 *old_of_local, const u32 *deg_local, u32 n_loc_real, u32 row0,
                            u32 n_loc_pad, const u64 *slice_off, const u32 *slice_w, u32 *cols,
                            u32 sentinel)
{
    const u32 l = row0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n_loc_pad) return;
    const u32 s = (l - row0) >> 6, lane = (l - row0) & 63;
    const u32 w = slice_w[s];
    uint4 *out = reinterpret_cast<uint4 *>(cols + slice_off[s]) + lane;
    u32 d = 0;
    u64 base = 0;
    if (l < n_loc_real) {
        d = deg_local[l];
        base = row_ptr[old_of_local[l]];
    }
    for (u32 k = 0; k < w; k += 4) {
        uint4 c;
        c.x = (k + 0 < d) ? code_of_old[col_idx[base + k + 0]] : sentinel;
        c.y = (k + 1 < d) ? code_of_old[col_idx[base + k + 1]] : sentinel;
        c.z = (k + 2 < d) ? code_of_old[col_idx[base + k + 2]] : sentinel;
        c.w = (k + 3 < d) ? code_of_old[col_idx[base + k + 3]] : sentinel;
        out[(size_t)(k >> 2) * 64] = c;
    }
}

// Split rows: one wavefront per row copies (and recodes) its entries contiguously, padded to 4.
__global__ void k_fill_long(const u64 *row_ptr, const u32 *col_idx, const u32 *code_of_old,
                            const u32 *old_of_local, const u32 *deg_local, u32 n_loc_real,
                            const u64 *long_ptr, u32 *long_cols, u32 sentinel)
{
    const u32 r = blockIdx.x;
    const u64 beg = long_ptr[r], end = long_ptr[r + 1];
    u32 d = 0;
    u64 base = 0;
    if (r < n_loc_real) {
        d = deg_local[r];
        base = row_ptr[old_of_local[r]];
    }
    for (u64 k = threadIdx.x; k < end - beg; k += blockDim.x)
        long_cols[beg + k] = (k < d) ? code_of_old[col_idx[base + k]] : sentinel;
}

// entries of each local row whose column is staged in LDS (code < hub); one wavefront per row, rows strided over the
// grid (a launch holds at most 2^32 work-items: one block per row wrapped silently from 67 M rows)
__global__ void __launch_bounds__(64)
k_row_hub_count(const u64 *row_ptr, const u32 *col_idx, const u32 *code_of_old, const u32 *old_of_local,
                const u32 *deg_local, u32 n_rows, u32 hub, u32 *hub_deg, u32 *nonhub_deg)
{
    const u32 lane = threadIdx.x;
    for (u32 l = blockIdx.x; l < n_rows; l += gridDim.x) {
        const u32 d = deg_local[l];
        const u64 base = row_ptr[old_of_local[l]];
        u32 cnt = 0;
        for (u32 k = lane; k < d; k += 64) cnt += code_of_old[col_idx[base + k]] < hub;
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        if (lane == 0) {
            hub_deg[l] = cnt;
            nonhub_deg[l] = d - cnt;
        }
    }
}

// Hub-only variants of the two fills (propagation-blocking mode): keep, in order, the entries with
// code < hub; pad with `pad_code` (an LDS slot that holds 0).
__global__ void k_fill_sell_hub(const u64 *row_ptr, const u32 *col_idx, const u32 *code_of_old,
                                const u32 *old_of_local, const u32 *deg_local, u32 n_loc_real, u32 row0,
                                u32 n_loc_pad, const u64 *slice_off, const u32 *slice_w, uint16_t *cols, u32 hub,
                                u32 pad_code)
{
    const u32 l = row0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n_loc_pad) return;
    const u32 s = (l - row0) >> 6, lane = (l - row0) & 63;
    const u32 w = slice_w[s];
    uint16_t *out = cols + slice_off[s] + (size_t)lane * 4;   // packet p of this lane: out[p * 256 .. +4), 16-bit codes
    u32 kept = 0;
    if (l < n_loc_real) {
        const u32 d = deg_local[l];
        const u64 base = row_ptr[old_of_local[l]];
        for (u32 k = 0; k < d; ++k) {
            const u32 cde = code_of_old[col_idx[base + k]];
            if (cde < hub) {
                out[(size_t)(kept >> 2) * 256 + (kept & 3)] = (uint16_t)cde;
                ++kept;
            }
        }
    }
    for (; kept < w; ++kept) out[(size_t)(kept >> 2) * 256 + (kept & 3)] = (uint16_t)pad_code;
}

__global__ void __launch_bounds__(64)
k_fill_long_hub(const u64 *row_ptr, const u32 *col_idx, const u32 *code_of_old, const u32 *old_of_local,
                const u32 *deg_local, u32 n_loc_real, const u64 *long_ptr, uint16_t *long_cols, u32 hub, u32 pad_code)
{
    const u32 r = blockIdx.x, lane = threadIdx.x;
    const u64 beg = long_ptr[r], end = long_ptr[r + 1];
    u64 out = beg;
    if (r < n_loc_real) {
        const u32 d = deg_local[r];
        const u64 base = row_ptr[old_of_local[r]];
        for (u32 k0 = 0; k0 < d; k0 += 64) {
            const u32 k = k0 + lane;
            u32 cde = 0;
            bool keep = false;
            if (k < d) {
                cde = code_of_old[col_idx[base + k]];
                keep = cde < hub;
            }
            const unsigned long long m = __ballot(keep);
            if (keep) long_cols[out + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)cde;
            out += __popcll(m);
        }
    }
    for (u64 k = out + lane; k < end; k += 64) long_cols[k] = (uint16_t)pad_code;
}


// ==================================================================================================================
// Sharded hand-over (option sharded_ingest; SURVEY.md 7.1 step 7, the loader of parallel-final/lib/adjMatrix.cc:21-46 for
// graphs a single device need not hold).  The whole-graph hand-over sorts all 2 m directed keys at once and leaves the
// whole CSR on every rank (C5: 32 GB of keys, twice more for the sort, then 17 GB of CSR -- per rank).  Here a rank never
// holds more than one bounded batch of keys besides ITS OWN rows:
//   sweep 1  degrees: for each of B key classes (row % B) the source is re-drawn, the class's keys sorted and
//            de-duplicated, and the run lengths are the degrees of the class's vertices;
//   sweep 2  (blocked mode) the same over the keys whose column is one of the staged top-degree vertices: the count that
//            breaks ties of the degree ranking (k_staged_key);
//   sweep 3  rows: for contiguous vertex ranges, the keys whose row this rank OWNS under the final ranking, sorted and
//            de-duplicated straight into its col_idx.
// Nothing is exchanged: every rank re-draws the same seeded source (or sweeps the same edge list), so all ranks compute
// the same degrees and the same ranking without talking, exactly as they do from a whole graph.  The resulting tables are
// those of the whole-graph hand-over entry for entry (tests/test_gpu_sharded.py compares SpMV and Lanczos bits).
// The sparse second exchange chunk derives its send lists from the rank's own rows through the matrix's symmetry.
__device__ __forceinline__ u64 gen_word(u64 seed, u64 ctr)
{
    u64 z = seed + (ctr + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// draw e of the seeded generators (integer specification: oracle/lanczos_oracle.c, orc_gen_er_keys / orc_gen_rmat_keys)
__device__ __forceinline__ bool gen_draw(int kind, u32 scale, u64 n, u64 seed, u32 ta, u32 tab, u32 tabc, u64 e, u64 &u, u64 &v)
{
    if (kind == 0) {
        u = ((gen_word(seed, 2 * e) >> 32) * n) >> 32;
        v = ((gen_word(seed, 2 * e + 1) >> 32) * n) >> 32;
        return u != v;
    }
    for (u32 t = 0; t < 8; ++t) {
        u64 w = 0;
        u = 0; v = 0;
        for (u32 l = 0; l < scale; ++l) {
            if ((l & 3) == 0) w = gen_word(seed, (e * 8 + t) * 8 + (l >> 2));
            const u32 r = (u32)(w & 0xffff);
            w >>= 16;
            const u32 ub = r >= tab;
            const u32 vb = (r >= ta && r < tab) || r >= tabc;
            u = (u << 1) | ub;
            v = (v << 1) | vb;
        }
        if (u < n && v < n) return u != v;
    }
    return false;
}

struct ShardFilter {
    int mode = 0;              // 0: row % mod == cls;  1: that, and the column is staged (rank < hub);  2: lo <= row < hi and the row is mine
    u32 mod = 1, cls = 0;
    const u32 *rank_of_old = nullptr;
    u32 hub = 0;
    u64 lo = 0, hi = 0;
    const u32 *gidx_of_old = nullptr;
    u32 n_loc_pad = 1, me = 0;
};

__device__ __forceinline__ bool shard_pass(const ShardFilter &f, u64 row, u64 col)
{
    if (f.mode == 2) return row >= f.lo && row < f.hi && f.gidx_of_old[row] / f.n_loc_pad == f.me;
    if (row % f.mod != f.cls) return false;
    return f.mode == 0 || f.rank_of_old[col] < f.hub;
}

// The directed keys of the source that pass the filter, appended in no particular order (they are sorted next).  out ==
// nullptr: count only.  One atomic per wavefront and round.
__global__ void __launch_bounds__(256)
k_shard_emit(lzx_ctx::lzx_key_source s, ShardFilter f, u64 *out, unsigned long long *count)
{
    const u64 total = s.kind == 2 ? s.m : s.draws;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const u32 lane = threadIdx.x & 63;
    unsigned long long mine = 0;
    for (u64 e0 = (u64)blockIdx.x * blockDim.x; e0 < total; e0 += stride) {
        const u64 e = e0 + threadIdx.x;
        u64 u = 0, v = 0;
        bool ok = false;
        if (e < total) {
            if (s.kind == 2) {
                u = s.d_src[e]; v = s.d_dst[e];
                ok = true;                      // endpoints were range-checked when the list was uploaded
            } else {
                ok = gen_draw(s.kind, s.scale, s.n, s.seed, s.ta, s.tab, s.tabc, e, u, v);
            }
        }
        // a self loop (edge lists only) is ONE diagonal entry, as in the std::set build
        const bool k0 = ok && shard_pass(f, u, v), k1 = ok && u != v && shard_pass(f, v, u);
        const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1);
        const u32 n0 = (u32)__popcll(m0), n1 = (u32)__popcll(m1);
        if (!out) { if (lane == 0) mine += n0 + n1; continue; }
        if (n0 + n1 == 0) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(count, (unsigned long long)(n0 + n1));
        base = __shfl(base, 0, 64);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (k0) out[base + __popcll(m0 & below)] = (u << 32) | v;
        if (k1) out[base + n0 + __popcll(m1 & below)] = (v << 32) | u;
    }
    if (!out && lane == 0 && mine) atomicAdd(count, mine);
}

// Sweeps 1 and 2: how many keys each of the `mod` classes will hold, all classes in ONE pass over the source (a count pass
// per class would re-draw the source as often again as the fill passes do).  Block-private histogram in LDS.
__global__ void __launch_bounds__(256)
k_shard_class_counts(lzx_ctx::lzx_key_source s, ShardFilter f, unsigned long long *counts)
{
    extern __shared__ u32 hist[];
    for (u32 j = threadIdx.x; j < f.mod; j += blockDim.x) hist[j] = 0;
    __syncthreads();
    const u64 total = s.kind == 2 ? s.m : s.draws;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        u64 u = 0, v = 0;
        bool ok = true;
        if (s.kind == 2) { u = s.d_src[e]; v = s.d_dst[e]; }
        else ok = gen_draw(s.kind, s.scale, s.n, s.seed, s.ta, s.tab, s.tabc, e, u, v);
        if (!ok) continue;
        if (f.mode == 0 || f.rank_of_old[v] < f.hub) atomicAdd(&hist[u % f.mod], 1u);
        if (u != v && (f.mode == 0 || f.rank_of_old[u] < f.hub)) atomicAdd(&hist[v % f.mod], 1u);
    }
    __syncthreads();
    for (u32 j = threadIdx.x; j < f.mod; j += blockDim.x)
        if (hist[j]) atomicAdd(&counts[j], (unsigned long long)hist[j]);
}

// sorted unique keys: every row is one run; its length goes to out[row] (two atomics per row, none contended)
__global__ void k_shard_run_lengths(const u64 *keys, u64 cnt, u32 *out)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    const u32 row = (u32)(keys[i] >> 32);
    if (i == 0 || (u32)(keys[i - 1] >> 32) != row) atomicSub(&out[row], (u32)i);
    if (i + 1 == cnt || (u32)(keys[i + 1] >> 32) != row) atomicAdd(&out[row], (u32)(i + 1));
}

__global__ void k_shard_cols(const u64 *keys, u64 cnt, u32 *col_idx)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) col_idx[i] = (u32)(keys[i] & 0xffffffffu);
}

__global__ void k_shard_owned_deg(const u32 *deg, const u32 *gidx_of_old, u32 n_loc_pad, u32 me, u64 n, u64 *out)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gidx_of_old[i] / n_loc_pad == me ? deg[i] : 0u;
    if (i == n) out[i] = 0;
}

__global__ void k_shard_sum_deg(const u32 *deg, u64 n, unsigned long long *total)
{
    const u64 nthreads = (u64)gridDim.x * blockDim.x;
    unsigned long long s = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nthreads) s += deg[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(total, s);
}

__global__ void k_shard_iota(u32 *ids, u64 n)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ids[i] = (u32)i;
}

// first vertex whose row_ptr reaches target[j]
__global__ void k_shard_bounds(const u64 *row_ptr, u64 n, const u64 *target, u32 count, u64 *bound)
{
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    u64 lo = 0, hi = n;
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (row_ptr[mid] < target[j]) lo = mid + 1; else hi = mid;
    }
    bound[j] = lo;
}

// k_staged_key with the staged-column counts already known per vertex (sharded hand-over, sweep 2)
__global__ void k_staged_key_counted(const u32 *sorted_ids, const u32 *sorted_deg, const u32 *staged_of_old, u64 n_active, u64 n,
                                     u32 hub, int count_major, u64 *key)
{
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const u32 d = sorted_deg[r];
    if (r < hub) { key[r] = count_major ? ~(u64)r : ((u64)d << 32) | (0xffffffffu - (u32)r); return; }
    const u32 low = r < n_active ? staged_of_old[sorted_ids[r]] : 0u;
    key[r] = count_major ? ((u64)low << 32) | d : ((u64)d << 32) | low;
}

// k_sx_mark from this rank's rows alone: `ref` as there; `want` through the symmetry of the matrix -- a row of rank p has an
// entry in my column v exactly when my row v has an entry in a column that rank p owns.
__global__ void __launch_bounds__(64)
k_sx_mark_own(const u64 *row_ptr, const u32 *col_idx, const u32 *sorted_ids, const u32 *code_of_old, const u32 *gidx_of_old,
              u64 n_active, u32 world, u32 me, u32 hub, u32 xs0, u32 L1, u32 n_loc_pad, uint8_t *ref, uint8_t *want)
{
    const u32 lane = threadIdx.x;
    const u32 c1 = hub + world * xs0;
    for (u64 l = blockIdx.x; l * world + me < n_active; l += gridDim.x) {
        const u32 o = sorted_ids[l * world + me];
        const u32 own = code_of_old[o];
        const bool in_chunk1 = own >= c1;
        const u32 my_l = in_chunk1 ? (own - c1) % L1 : 0u;
        const u64 beg = row_ptr[o], end = row_ptr[o + 1];
        for (u64 k = beg + lane; k < end; k += 64) {
            const u32 col = col_idx[k];
            const u32 code = code_of_old[col];
            if (code >= c1) ref[code - c1] = 1;
            if (in_chunk1) want[(size_t)(gidx_of_old[col] / n_loc_pad) * L1 + my_l] = 1;
        }
    }
}

// Sorts d_keys[nkeys] and drops duplicates.  Consumes (frees) d_keys; on success *d_uniq (caller frees) holds *cnt keys.
static int sorted_unique_keys(lzx_ctx *c, u64 *d_keys, u64 nkeys, u64 **d_uniq_out, u64 *cnt_out)
{
    hipStream_t st = c->stream;
    u64 *d_sorted = nullptr, *d_uniq = nullptr, *d_count = nullptr;
    void *d_tmp = nullptr;
    int rc = LZX_OK;
    *d_uniq_out = nullptr;
    *cnt_out = 0;
    auto cleanup = [&]() {
        dev_free(d_keys); dev_free(d_sorted); dev_free(d_uniq); dev_free(d_count);
        if (d_tmp) (void)hipFree(d_tmp);
        d_tmp = nullptr;
    };
#define SUK(call) do { rc = (call); if (rc != LZX_OK) { cleanup(); return rc; } } while (0)
#define SUK_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
        lzx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); cleanup(); \
        return e_ == hipErrorOutOfMemory ? LZX_ERR_NOMEM : LZX_ERR_HIP; } } while (0)
    // one work-item per key below, and a launch holds at most 2^32 of them
    if (nkeys >= (1ull << 32) - (1ull << 24)) { cleanup(); LZX_FAIL(LZX_ERR_LIMIT, "edge list too long for one ingest (%llu directed entries)", (unsigned long long)nkeys); }
    SUK(dev_alloc(&d_sorted, nkeys)); SUK(dev_alloc(&d_count, 1));
    size_t tb = 0;
    SUK_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, d_keys, d_sorted, (u64)nkeys, 0, 64, st));
    SUK_HIP(hipMalloc(&d_tmp, tb ? tb : 16));
    SUK_HIP(hipcub::DeviceRadixSort::SortKeys(d_tmp, tb, d_keys, d_sorted, (u64)nkeys, 0, 64, st));
    SUK_HIP(hipStreamSynchronize(st));
    (void)hipFree(d_tmp); d_tmp = nullptr;
    dev_free(d_keys);
    SUK(dev_alloc(&d_uniq, nkeys));
    tb = 0;
    SUK_HIP(hipcub::DeviceSelect::Unique(nullptr, tb, d_sorted, d_uniq, d_count, (u64)nkeys, st));
    SUK_HIP(hipMalloc(&d_tmp, tb ? tb : 16));
    SUK_HIP(hipcub::DeviceSelect::Unique(d_tmp, tb, d_sorted, d_uniq, d_count, (u64)nkeys, st));
    u64 cnt = 0;
    SUK_HIP(hipMemcpyAsync(&cnt, d_count, sizeof(u64), hipMemcpyDeviceToHost, st));
    SUK_HIP(hipStreamSynchronize(st));
    *d_uniq_out = d_uniq;
    *cnt_out = cnt;
    d_uniq = nullptr;
    cleanup();
#undef SUK
#undef SUK_HIP
    return LZX_OK;
}

static const u64 LZX_SHARD_BATCH = 1ull << 28;   // directed keys per sweep batch when the option leaves the choice here (2 GiB; about 8 GiB with the sort's buffers)

// One batch of a sweep: the filtered keys of the source, sorted, unique.  *d_uniq is null when the batch is empty.
// `known` >= 0: the batch's key count is already known (k_shard_class_counts); otherwise a count pass finds it.
static int shard_batch(lzx_ctx *c, const ShardFilter &f, u64 **d_uniq, u64 *cnt, long long known = -1)
{
    hipStream_t st = c->stream;
    *d_uniq = nullptr;
    *cnt = 0;
    const u64 total = c->shard.kind == 2 ? c->shard.m : c->shard.draws;
    if (total == 0 || known == 0) return LZX_OK;
    unsigned long long *d_cnt = nullptr, h_cnt = known > 0 ? (unsigned long long)known : 0;
    LZX_TRY(dev_alloc(&d_cnt, 1));
    const u32 grid = (u32)std::min<u64>((total + 255) / 256, (u64)c->cu_count * 32);
    hipError_t e = hipSuccess;
    if (known < 0) {
        e = hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_shard_emit, dim3(grid), dim3(256), 0, st, c->shard, f, (u64 *)nullptr, d_cnt);
            e = hipMemcpyAsync(&h_cnt, d_cnt, sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (e != hipSuccess) { dev_free(d_cnt); LZX_FAIL(LZX_ERR_HIP, "sharded hand-over, count pass: %s", hipGetErrorString(e)); }
    if (h_cnt == 0) { dev_free(d_cnt); return LZX_OK; }
    u64 *d_keys = nullptr;
    if (dev_alloc(&d_keys, h_cnt) != LZX_OK) { dev_free(d_cnt); return LZX_ERR_NOMEM; }
    unsigned long long h_cnt2 = 0;
    e = hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_shard_emit, dim3(grid), dim3(256), 0, st, c->shard, f, d_keys, d_cnt);
        e = hipMemcpyAsync(&h_cnt2, d_cnt, sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    dev_free(d_cnt);
    if (e != hipSuccess) { dev_free(d_keys); LZX_FAIL(LZX_ERR_HIP, "sharded hand-over, fill pass: %s", hipGetErrorString(e)); }
    if (h_cnt2 != h_cnt) { dev_free(d_keys); LZX_FAIL(LZX_ERR_STATE, "sharded hand-over: the source gave %llu keys, then %llu", h_cnt, h_cnt2); }
    return sorted_unique_keys(c, d_keys, h_cnt, d_uniq, cnt);
}

// ---- kind 3: the caller's CSR stays in host memory and passes through the device in chunks of whole rows -----------------
// one wavefront per row of the chunk (rows strided over the grid); cols = the chunk's entries, row_ptr = device copy of ALL offsets.
// mode 1: out[row] = entries whose column is staged (rank < hub);  mode 2: this rank's rows are copied to col_out[own_ptr[row] ..]
__global__ void __launch_bounds__(64)
k_shard_csr_chunk(int mode, const u64 *row_ptr, u64 r0, u64 r1, const u32 *cols, u64 n, const u32 *rank_of_old, u32 hub, u32 *out,
                  const u32 *gidx_of_old, u32 n_loc_pad, u32 me, const u64 *own_ptr, u32 *col_out, u32 *bad)
{
    const u32 lane = threadIdx.x;
    const u64 base = row_ptr[r0];
    for (u64 r = r0 + blockIdx.x; r < r1; r += gridDim.x) {
        const u64 beg = row_ptr[r] - base, end = row_ptr[r + 1] - base;
        if (mode == 1) {
            u32 cnt = 0;
            for (u64 k = beg + lane; k < end; k += 64) {
                const u32 cl = cols[k];
                if (cl >= n) { bad[0] = 1u; continue; }
                cnt += rank_of_old[cl] < hub ? 1u : 0u;
            }
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
            if (lane == 0) out[r] = cnt;
        } else if (gidx_of_old[r] / n_loc_pad == me) {
            const u64 dst = own_ptr[r];
            for (u64 k = beg + lane; k < end; k += 64) {
                const u32 cl = cols[k];
                if (cl >= n) bad[0] = 1u;
                col_out[dst + (k - beg)] = cl < n ? cl : 0u;
            }
        }
    }
}

// Streams the host CSR through the device: rows in chunks of at most LZX_SHARD_BATCH / 2 entries (or shard_opt chunks).
static int shard_csr_sweep(lzx_ctx *c, int mode, const u32 *d_rank_of_old, u32 hub, u32 *d_out)
{
    hipStream_t st = c->stream;
    const u64 n = c->n;
    const u64 *rp = c->shard.h_row_ptr;
    const u64 nnz = rp[n];
    if (mode == 1) LZX_HIP(hipMemsetAsync(d_out, 0, sizeof(u32) * n, st));
    if (nnz == 0) return LZX_OK;
    u64 cap = c->shard_opt >= 2 ? (nnz + (u64)c->shard_opt - 1) / (u64)c->shard_opt : LZX_SHARD_BATCH / 2;
    u64 widest = 0;
    for (u64 r = 0; r < n; ++r) widest = std::max(widest, rp[r + 1] - rp[r]);
    cap = std::max(cap, widest);   // a chunk holds whole rows
    u64 *d_rp = nullptr;
    u32 *d_cols = nullptr, *d_bad = nullptr, bad = 0;
    LZX_TRY(dev_alloc(&d_rp, n + 1));
    int rc = dev_alloc(&d_cols, cap);
    if (rc == LZX_OK) rc = dev_alloc(&d_bad, 1);
    hipError_t e = hipSuccess;
    if (rc == LZX_OK) {
        e = hipMemcpyAsync(d_rp, rp, sizeof(u64) * (n + 1), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, sizeof(u32), st);
        for (u64 r0 = 0; e == hipSuccess && r0 < n;) {
            u64 r1 = r0 + 1;
            // the last row r1 with rp[r1] - rp[r0] <= cap (binary search on the host's offsets)
            u64 lo = r0 + 1, hi = n;
            while (lo < hi) {
                const u64 mid = (lo + hi + 1) >> 1;
                if (rp[mid] - rp[r0] <= cap) lo = mid; else hi = mid - 1;
            }
            r1 = lo;
            const u64 cnt = rp[r1] - rp[r0];
            if (cnt) {
                e = hipMemcpyAsync(d_cols, c->shard.h_col_idx + rp[r0], sizeof(u32) * cnt, hipMemcpyHostToDevice, st);
                if (e != hipSuccess) break;
                hipLaunchKernelGGL(k_shard_csr_chunk, dim3((u32)std::min<u64>(r1 - r0, 1u << 20)), dim3(64), 0, st, mode, d_rp, r0, r1, d_cols, n,
                                   d_rank_of_old, hub, d_out, c->d_gidx_of_old, c->n_loc_pad, (u32)c->rank, c->d_row_ptr, c->d_col_idx, d_bad);
                e = hipStreamSynchronize(st);   // d_cols is reused by the next chunk
            }
            r0 = r1;
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof(u32), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    dev_free(d_rp); dev_free(d_cols); dev_free(d_bad);
    if (rc != LZX_OK) return rc;
    if (e != hipSuccess) LZX_FAIL(LZX_ERR_HIP, "sharded hand-over of a host CSR: %s", hipGetErrorString(e));
    if (bad) LZX_FAIL(LZX_ERR_ARG, "lzx_set_graph_csr: a column index is >= n");
    return LZX_OK;
}

static u32 shard_classes(const lzx_ctx *c)
{
    if (c->shard_opt >= 2) return (u32)std::min<int64_t>(c->shard_opt, 4096);
    const u64 total = 2 * (c->shard.kind == 2 ? c->shard.m : c->shard.draws);
    return (u32)std::max<u64>(1, (total + LZX_SHARD_BATCH - 1) / LZX_SHARD_BATCH);
}

// Sweeps 1 and 2: d_out[v] (zeroed here) = number of distinct entries of row v (mode 0), or of those whose column is staged (mode 1).
static int shard_count_sweep(lzx_ctx *c, int mode, const u32 *d_rank_of_old, u32 hub, u32 *d_out)
{
    if (c->shard.kind == 3) return shard_csr_sweep(c, 1, d_rank_of_old, hub, d_out);   // (mode 0 -- degrees -- comes from row_ptr itself)
    LZX_HIP(hipMemsetAsync(d_out, 0, sizeof(u32) * c->n, c->stream));
    const u32 classes = shard_classes(c);
    const u64 total = c->shard.kind == 2 ? c->shard.m : c->shard.draws;
    std::vector<unsigned long long> per_class(classes, 0);
    if (total) {
        unsigned long long *d_counts = nullptr;
        LZX_TRY(dev_alloc(&d_counts, classes));
        ShardFilter f;
        f.mode = mode; f.mod = classes; f.rank_of_old = d_rank_of_old; f.hub = hub;
        hipError_t e = hipMemsetAsync(d_counts, 0, sizeof(unsigned long long) * classes, c->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_shard_class_counts, dim3((u32)std::min<u64>((total + 255) / 256, (u64)c->cu_count * 32)), dim3(256),
                               sizeof(u32) * classes, c->stream, c->shard, f, d_counts);
            e = hipMemcpyAsync(per_class.data(), d_counts, sizeof(unsigned long long) * classes, hipMemcpyDeviceToHost, c->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        dev_free(d_counts);
        if (e != hipSuccess) LZX_FAIL(LZX_ERR_HIP, "sharded hand-over, class counts: %s", hipGetErrorString(e));
    }
    for (u32 cls = 0; cls < classes; ++cls) {
        ShardFilter f;
        f.mode = mode; f.mod = classes; f.cls = cls; f.rank_of_old = d_rank_of_old; f.hub = hub;
        u64 *d_uniq = nullptr, cnt = 0;
        LZX_TRY(shard_batch(c, f, &d_uniq, &cnt, (long long)per_class[cls]));
        if (cnt) hipLaunchKernelGGL(k_shard_run_lengths, dim3((u32)((cnt + 255) / 256)), dim3(256), 0, c->stream, d_uniq, cnt, d_out);
        hipError_t e = hipStreamSynchronize(c->stream);
        dev_free(d_uniq);
        if (e != hipSuccess) LZX_FAIL(LZX_ERR_HIP, "sharded hand-over, degree sweep: %s", hipGetErrorString(e));
    }
    return LZX_OK;
}

// Sweep 3: this rank's rows.  d_row_ptr[n + 1] over ALL vertices (a row of another rank is empty), d_col_idx = own entries.
static int shard_build_rows(lzx_ctx *c)
{
    hipStream_t st = c->stream;
    const u64 n = c->n;
    LZX_TRY(dev_alloc(&c->d_row_ptr, n + 1));
    hipLaunchKernelGGL(k_shard_owned_deg, dim3((u32)((n + 256) / 256)), dim3(256), 0, st, c->d_shard_deg, c->d_gidx_of_old, c->n_loc_pad,
                       (u32)c->rank, n, c->d_row_ptr);
    {
        size_t sb = 0;
        void *tmp = nullptr;
        LZX_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, sb, c->d_row_ptr, c->d_row_ptr, n + 1, st));
        LZX_HIP(hipMalloc(&tmp, sb ? sb : 16));
        hipError_t e = hipcub::DeviceScan::ExclusiveSum(tmp, sb, c->d_row_ptr, c->d_row_ptr, n + 1, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        (void)hipFree(tmp);
        LZX_HIP(e);
    }
    u64 own = 0;
    LZX_HIP(hipMemcpy(&own, c->d_row_ptr + n, sizeof(u64), hipMemcpyDeviceToHost));
    if (own >= (1ull << 32)) LZX_FAIL(LZX_ERR_LIMIT, "%llu entries in this rank's rows: use more ranks (limit 2^32 per rank)", (unsigned long long)own);
    LZX_TRY(dev_alloc(&c->d_col_idx, own));
    if (own == 0) return LZX_OK;
    if (c->shard.kind == 3) return shard_csr_sweep(c, 2, nullptr, 0, nullptr);
    // vertex ranges holding about equal shares of the rank's entries
    u32 blocks = c->shard_opt >= 2 ? (u32)std::min<int64_t>(c->shard_opt, 4096) : (u32)std::max<u64>(1, (own + LZX_SHARD_BATCH / 2 - 1) / (LZX_SHARD_BATCH / 2));
    std::vector<u64> target(blocks + 1), bound(blocks + 1), at(blocks + 1);
    for (u32 j = 0; j <= blocks; ++j) target[j] = own / blocks * j;
    target[blocks] = own;
    u64 *d_t = nullptr, *d_b = nullptr;
    LZX_TRY(dev_alloc(&d_t, blocks + 1));
    if (dev_alloc(&d_b, blocks + 1) != LZX_OK) { dev_free(d_t); return LZX_ERR_NOMEM; }
    hipError_t e = hipMemcpyAsync(d_t, target.data(), sizeof(u64) * (blocks + 1), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_shard_bounds, dim3((blocks + 64) / 64), dim3(64), 0, st, c->d_row_ptr, n, d_t, blocks + 1, d_b);
        e = hipMemcpyAsync(bound.data(), d_b, sizeof(u64) * (blocks + 1), hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    dev_free(d_t); dev_free(d_b);
    LZX_HIP(e);
    bound[0] = 0;
    bound[blocks] = n;
    for (u32 j = 0; j <= blocks; ++j)
        LZX_HIP(hipMemcpy(&at[j], c->d_row_ptr + bound[j], sizeof(u64), hipMemcpyDeviceToHost));
    for (u32 j = 0; j < blocks; ++j) {
        if (bound[j + 1] <= bound[j]) continue;
        ShardFilter f;
        f.mode = 2; f.lo = bound[j]; f.hi = bound[j + 1]; f.gidx_of_old = c->d_gidx_of_old; f.n_loc_pad = c->n_loc_pad; f.me = (u32)c->rank;
        u64 *d_uniq = nullptr, cnt = 0;
        LZX_TRY(shard_batch(c, f, &d_uniq, &cnt));
        if (cnt != at[j + 1] - at[j]) {
            dev_free(d_uniq);
            LZX_FAIL(LZX_ERR_STATE, "sharded hand-over: vertices %llu..%llu hold %llu entries of this rank where the degree sweep counted %llu",
                     (unsigned long long)bound[j], (unsigned long long)bound[j + 1], (unsigned long long)cnt, (unsigned long long)(at[j + 1] - at[j]));
        }
        if (cnt) hipLaunchKernelGGL(k_shard_cols, dim3((u32)((cnt + 255) / 256)), dim3(256), 0, st, d_uniq, cnt, c->d_col_idx + at[j]);
        e = hipStreamSynchronize(st);
        dev_free(d_uniq);
        LZX_HIP(e);
    }
    return LZX_OK;
}

static u32 round_up(u32 a, u32 m) { return (a + m - 1) / m * m; }

int lzx_graph_prepare(lzx_ctx *c)
{
    const u64 n = c->n;
    const u32 world = (u32)c->world, rank = (u32)c->rank;
    if (n == 0) LZX_FAIL(LZX_ERR_ARG, "graph has no vertices");
    if (n >= (1ull << 31)) LZX_FAIL(LZX_ERR_LIMIT, "n = %llu: this build indexes vertices with 31 bits", (unsigned long long)n);

    hipStream_t st = c->stream;
    const u32 per = (u32)((n + world - 1) / world);
    c->n_loc_pad = round_up(per, LZX_SLICE);
    c->ldq = c->n_loc_pad + LZX_TAIL;
    c->iolen = (u64)world * c->n_loc_pad + LZX_TAIL;
    c->xs = c->n_loc_pad;          // refined below once the number of vertices with an edge is known
    c->xlen = c->iolen;
    c->n_loc_real = (rank < n) ? (u32)((n - rank + world - 1) / world) : 0;
    if (c->iolen + 65536 >= (1ull << 32)) LZX_FAIL(LZX_ERR_LIMIT, "exchange layout does not fit 32-bit codes");

    // Propagation blocking (lzx_pb.hip) for every entry whose column is not staged in LDS.  -1 = decide here.
    // Measured (DESIGN.md section 3, 1 GPU, plain / blocked SpMV): C3 (10 M vertices, x far beyond the L2s) 2.13 / 1.05 ms;
    // C2 (1 M vertices, x = 8 MB, L2-resident) 0.136 / 0.103 ms -- since high-degree rows cross the two passes as partial
    // sums the blocked form also wins while x still fits the caches.  Below about 8 Mi entries per rank its extra
    // launches and the per-unit 128 KiB column band cost more than the gathers they replace.
    bool pb = c->pb_opt > 0 || (c->pb_opt < 0 && n >= (512u << 10) && c->nnz / (u64)world >= (8u << 20));
    // hub entries staged in LDS by k_spmv: 8192 (64 KiB, two workgroups per CU) when k_spmv also gathers from
    // memory; 16384 (128 KiB, one per CU) in propagation-blocking mode, where it only ever reads LDS.
    // (round 4, second session: 18 Ki staged values beside 18 Ki column bands on ONE rank from 4 Mi vertices -- 10 M-vertex graph
    //  SpMV - 1.9 % in two parity-controlled sweeps, profiles/r4_wide_band.txt; the 1 M-vertex graph loses 2.5 % with more staged
    //  values and keeps 16 Ki of them)
    u64 hub = (c->hub_opt >= 0) ? (u64)c->hub_opt : (pb ? (world == 1 && !c->force_multi && n >= (4u << 20) ? LZX_PB_CB_WIDE : 16384) : 8192);
    hub = std::min<u64>(hub, n);
    hub = std::min<u64>(hub, 20000);  // 160 KiB LDS per CU
    hub &= ~1ull;
    if (hub == 0 || hub >= n) pb = false;   // nothing staged, or everything staged: nothing to block
    c->hub_real = (u32)hub;
    c->hub = (u32)hub + (pb ? 2 : 0);       // PB mode: two zero slots behind the staged values (padding target)
    c->spmv_lds = ((size_t)c->hub + LZX_SPMV_BLOCK / 64) * sizeof(double);

    // ---- 1. degree ranking ----
    u32 *d_deg = nullptr, *d_ids = nullptr, *d_sdeg = nullptr, *d_sids = nullptr, *d_code = nullptr;
    u32 *d_old_of_local = nullptr, *d_deg_local = nullptr;
    u32 *d_hub_deg = nullptr, *d_nh_deg = nullptr, *d_nh_off = nullptr;
    void *d_tmp = nullptr;
    u64 *d_long_ptr = nullptr;
    int rc = LZX_OK;
    std::vector<u32> degl, h_nh;
    std::vector<u64> h_slice_off, h_long_ptr, h_item_beg;
    std::vector<u32> h_slice_w, h_item_len, h_item_first;
    auto cleanup = [&]() {
        dev_free(d_deg); dev_free(d_ids); dev_free(d_sdeg); dev_free(d_sids); dev_free(d_code);
        dev_free(d_old_of_local); dev_free(d_deg_local); dev_free(d_long_ptr);
        dev_free(d_hub_deg); dev_free(d_nh_deg); dev_free(d_nh_off);
        if (d_tmp) (void)hipFree(d_tmp);
        d_tmp = nullptr;
    };
#define PREP(call) do { rc = (call); if (rc != LZX_OK) { cleanup(); return rc; } } while (0)
#define PREP_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
        lzx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); cleanup(); \
        return e_ == hipErrorOutOfMemory ? LZX_ERR_NOMEM : LZX_ERR_HIP; } } while (0)

    PREP(dev_alloc(&d_deg, n)); PREP(dev_alloc(&d_ids, n));
    PREP(dev_alloc(&d_sdeg, n)); PREP(dev_alloc(&d_sids, n));
    PREP(dev_alloc(&d_code, n));
    PREP(dev_alloc(&c->d_gidx_of_old, n));
    const u32 gb = (u32)((n + 255) / 256);
    const bool sharded = c->shard.kind >= 0;   // sharded hand-over: no CSR yet -- degrees were counted by its first sweep
    if (sharded) {
        PREP_HIP(hipMemcpyAsync(d_deg, c->d_shard_deg, sizeof(u32) * n, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_shard_iota, dim3(gb), dim3(256), 0, st, d_ids, n);
    } else {
        hipLaunchKernelGGL(k_degrees, dim3(gb), dim3(256), 0, st, c->d_row_ptr, d_deg, d_ids, n);
    }
    size_t tmp_bytes = 0;
    PREP_HIP(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tmp_bytes, d_deg, d_sdeg, d_ids, d_sids,
                                                         (u64)n, 0, 32, st));
    PREP_HIP(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16));
    PREP_HIP(hipcub::DeviceRadixSort::SortPairsDescending(d_tmp, tmp_bytes, d_deg, d_sdeg, d_ids, d_sids,
                                                         (u64)n, 0, 32, st));
    {
        u32 md = 0;
        unsigned long long *d_cnt = nullptr, h_cnt = 0;
        PREP_HIP(hipMalloc(reinterpret_cast<void **>(&d_cnt), sizeof(unsigned long long)));
        hipError_t e = hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_count_active, dim3(gb), dim3(256), 0, st, d_sdeg, n, d_cnt);
            e = hipMemcpyAsync(&h_cnt, d_cnt, sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&md, d_sdeg, sizeof(u32), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        (void)hipFree(d_cnt);
        PREP_HIP(e);
        c->max_degree = md;
        c->n_active = h_cnt;
        // How many x values to stage is a per-GRAPH choice, not a per-size one (round 3): on a uniform graph the 16 Ki
        // highest-degree columns hold next to nothing (Erdos-Renyi, 10 M vertices / 200 M entries: 0.25 % of the entries), yet
        // the staged-columns kernel still sweeps every row for them, and 16 Ki fewer columns are what the blocked passes
        // would have taken anyway.  Measured (profiles/r3_er_probe.txt, SpMV ms with 16384 / 1024 staged): ER 10 M 1.31 /
        // 1.05, ER 4 M 0.474 / 0.437, ER 1 M (2.5 % of the entries staged) 0.102 / 0.109; R-MAT (31 %) needs them all.
        // Rule: below 1 % of the entries in the staged columns, stage 1 Ki.  Every rank holds the whole graph and computes
        // the same share, so all ranks agree without talking.
        if (pb && c->hub_opt < 0 && c->nnz > 0) {
            const u32 top = (u32)std::min<u64>(c->hub_real, n);
            std::vector<u32> hd(top);
            PREP_HIP(hipMemcpy(hd.data(), d_sdeg, sizeof(u32) * top, hipMemcpyDeviceToHost));
            u64 staged = 0;
            for (u32 dgr : hd) staged += dgr;
            if (staged * 100 < c->nnz && n > 2048) {
                c->hub_real = 1024;
                c->hub = c->hub_real + 2;
                c->spmv_lds = ((size_t)c->hub + LZX_SPMV_BLOCK / 64) * sizeof(double);
            }
        }
        const u64 live = h_cnt > rank ? (h_cnt - rank + world - 1) / world : 0;
        c->rows_live = std::min<u32>(c->n_loc_pad, round_up((u32)live, LZX_SLICE));
    }
    // Only vertices with at least one edge are ever gathered by an SpMV, and (degree-sorted, dealt round-robin)
    // they are a prefix of every rank's slice: the per-iteration exchange moves just that prefix.  (R-MAT: 41 % of
    // the 10 M-vertex benchmark graph is isolated.)  One rank exchanges nothing and gathers from the basis itself.
    if (world > 1) {
        c->xs = std::max<u32>(LZX_SLICE, round_up((u32)((c->n_active + world - 1) / world), LZX_SLICE));
        c->xs = std::min(c->xs, c->n_loc_pad);
        c->xlen = (u64)world * c->xs + LZX_TAIL;
    }
    // Two chunks (see lzx_internal.h: xs0) when the blocked SpMV will run on several ranks: chunk 0 is the first eighth
    // of every slice, rounded so that chunk 0 ends on a column-band boundary, and must hold the LDS-staged hub entries.
    c->xs0 = c->xs;
    c->overlap = false;
    c->xfp32 = (world > 1 || c->force_multi) && c->xfp32_opt > 0;   // N4: fp32 exchange uses the single all-gather
    // (over RCCL only on request: the two-chunk exchange with its sparse second chunk -- a grouped ncclSend / ncclRecv on
    //  the exchange stream -- has never run on two or more physical GPUs; bench.py asks for it in its guarded tuning
    //  phase.  In-process groups take it by default: every form of it is covered there, tests/test_gpu_parity.py.)
    const bool overlap_wanted = c->overlap_opt > 0 || (c->overlap_opt < 0 && c->comm_kind != 2);
    if ((world > 1 || c->force_multi) && pb && overlap_wanted && !c->xfp32) {
        // (rounded to the column band the blocked passes will take -- lzx_pb.hip: 18 Ki unless 16 Ki is forced or x is small)
        const u32 band = (c->pb_cb_opt == 8192 || c->pb_cb_opt == 16384 || c->xlen * sizeof(double) <= (16u << 20)) ? LZX_PB_CB : LZX_PB_CB_WIDE;
        u32 x0 = round_up(std::max<u32>(c->xs / 8, (c->hub_real + world - 1) / world), band);
        if (x0 < c->xs) {
            c->xs0 = x0;
            c->overlap = true;
        }
    }
    if (c->hub_real > c->n_active) {
        // staged slots beyond the vertices that have edges would never be referenced (and, with several ranks, would
        // lie outside the exchanged prefix)
        c->hub_real = (u32)(c->n_active & ~1ull);
        if (c->hub_real == 0) pb = false;
        c->hub = c->hub_real + (pb ? 2 : 0);
        c->spmv_lds = ((size_t)c->hub + LZX_SPMV_BLOCK / 64) * sizeof(double);
    }
    if (c->hub_real >= c->n_active) pb = false;   // every referenced column is staged: nothing left to block
    if (!pb) {
        c->xs0 = c->xs;
        c->overlap = false;
    }
    if (pb && c->n_active > c->hub_real && c->tie_sort_opt != 0) {
        // ---- 1a. ties of the degree ranking broken by the staged-column count (see k_staged_key); d_code is free until
        //          k_rank_maps fills it and serves as the rank-of-vertex scratch, d_ids as the second value buffer
        u64 *d_key = nullptr, *d_skey = nullptr;
        rc = dev_alloc(&d_key, n);
        if (rc == LZX_OK) rc = dev_alloc(&d_skey, n);
        if (rc != LZX_OK) { dev_free(d_key); dev_free(d_skey); cleanup(); return rc; }
        hipLaunchKernelGGL(k_rank_of_old, dim3(gb), dim3(256), 0, st, d_sids, d_code, n);
        if (sharded) {   // sweep 2: every vertex's count of staged columns, d_ids (free until the sort below) as the scratch
            rc = shard_count_sweep(c, 1, d_code, c->hub_real, d_ids);
            if (rc != LZX_OK) { dev_free(d_key); dev_free(d_skey); cleanup(); return rc; }
            hipLaunchKernelGGL(k_staged_key_counted, dim3(gb), dim3(256), 0, st, d_sids, d_sdeg, d_ids, c->n_active, n, c->hub_real,
                               c->tie_sort_opt == 2 ? 1 : 0, d_key);
        } else {
            hipLaunchKernelGGL(k_staged_key, dim3((u32)std::min<u64>(n, 1u << 20)), dim3(64), 0, st, c->d_row_ptr, c->d_col_idx, d_sids,
                               d_sdeg, d_code, c->n_active, n, c->hub_real, c->tie_sort_opt == 2 ? 1 : 0, d_key);
        }
        size_t tb = 0;
        hipError_t e = hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tb, d_key, d_skey, d_sids, d_ids, (u64)n, 0, 64, st);
        if (e == hipSuccess && tb > tmp_bytes) {
            (void)hipFree(d_tmp);
            d_tmp = nullptr;
            e = hipMalloc(&d_tmp, tb);
            if (e == hipSuccess) tmp_bytes = tb;
        }
        if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairsDescending(d_tmp, tb, d_key, d_skey, d_sids, d_ids, (u64)n, 0, 64, st);
        if (e == hipSuccess) {
            e = hipMemcpyAsync(d_sids, d_ids, sizeof(u32) * n, hipMemcpyDeviceToDevice, st);
            hipLaunchKernelGGL(k_degree_of, dim3(gb), dim3(256), 0, st, d_deg, d_sids, d_sdeg, n);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        dev_free(d_key); dev_free(d_skey);
        PREP_HIP(e);
    }
    const u32 sentinel = pb ? c->hub_real : c->hub + (u32)((u64)world * c->xs);
    hipLaunchKernelGGL(k_rank_maps, dim3(gb), dim3(256), 0, st, d_sids, c->d_gidx_of_old, d_code, n,
                       world, c->n_loc_pad, c->xs, c->xs0, c->hub_real);
    if (sharded) {   // sweep 3: the ranking is final, so ownership is: this rank's rows become its CSR
        PREP_HIP(hipStreamSynchronize(st));
        PREP(shard_build_rows(c));
    }

    // ---- 1b. sparse exchange of chunk 1: this rank's packed x layout and its send lists (see k_sx_mark) ----
    c->sparse = false;
    if (c->overlap && (world > 1 || c->force_multi) && c->sparse_opt != 0 && c->xs > c->xs0) {
        const u32 L1 = c->xs - c->xs0;
        const u64 cnt = (u64)world * L1;
        uint8_t *d_ref = nullptr, *d_want = nullptr;
        u32 *d_w32 = nullptr, *d_pos = nullptr, *d_spos = nullptr;
        auto sx_free = [&]() { dev_free(d_ref); dev_free(d_want); dev_free(d_w32); dev_free(d_pos); dev_free(d_spos); };
#define SX(call) do { rc = (call); if (rc != LZX_OK) { sx_free(); cleanup(); return rc; } } while (0)
#define SX_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
        lzx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); sx_free(); cleanup(); \
        return e_ == hipErrorOutOfMemory ? LZX_ERR_NOMEM : LZX_ERR_HIP; } } while (0)
        SX(dev_alloc(&d_ref, cnt)); SX(dev_alloc(&d_want, cnt));
        SX(dev_alloc(&d_w32, cnt + 1)); SX(dev_alloc(&d_pos, cnt + 1)); SX(dev_alloc(&d_spos, cnt + 1));
        SX_HIP(hipMemsetAsync(d_ref, 0, cnt, st));
        SX_HIP(hipMemsetAsync(d_want, 0, cnt, st));
        if (sharded && world > 1)
            hipLaunchKernelGGL(k_sx_mark_own, dim3((u32)std::min<u64>(std::max<u64>(c->n_active / world + 1, 1), 1u << 20)), dim3(64), 0, st, c->d_row_ptr,
                               c->d_col_idx, d_sids, d_code, c->d_gidx_of_old, c->n_active, world, rank, c->hub_real, c->xs0, L1, c->n_loc_pad, d_ref, d_want);
        else
            hipLaunchKernelGGL(k_sx_mark, dim3((u32)std::min<u64>(std::max<u64>(c->n_active, 1), 1u << 20)), dim3(64), 0, st, c->d_row_ptr, c->d_col_idx,
                               d_sids, d_code, c->n_active, world, rank, c->hub_real, c->xs0, L1, d_ref, d_want);
        auto scan = [&](const uint8_t *flags, u32 *out) -> int {
            hipLaunchKernelGGL(k_widen_u8, dim3((u32)((cnt + 255) / 256)), dim3(256), 0, st, flags, cnt, d_w32);
            LZX_HIP(hipMemsetAsync(d_w32 + cnt, 0, sizeof(u32), st));
            size_t sb = 0;
            LZX_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, sb, d_w32, out, cnt + 1, st));
            void *tmp = nullptr;
            LZX_HIP(hipMalloc(&tmp, sb ? sb : 16));
            hipError_t e = hipcub::DeviceScan::ExclusiveSum(tmp, sb, d_w32, out, cnt + 1, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            (void)hipFree(tmp);
            LZX_HIP(e);
            return LZX_OK;
        };
        SX(scan(d_ref, d_pos));
        SX(scan(d_want, d_spos));
        c->sx_recv_off.assign(world + 1, 0);
        c->sx_send_off.assign(world + 1, 0);
        for (u32 p = 0; p <= world; ++p) {
            SX_HIP(hipMemcpyAsync(&c->sx_recv_off[p], d_pos + (u64)p * L1, sizeof(u32), hipMemcpyDeviceToHost, st));
            SX_HIP(hipMemcpyAsync(&c->sx_send_off[p], d_spos + (u64)p * L1, sizeof(u32), hipMemcpyDeviceToHost, st));
        }
        SX_HIP(hipStreamSynchronize(st));
        c->xc1 = c->sx_recv_off[world];
        SX(dev_alloc(&c->d_sx_map, std::max<u64>(c->xc1, 1)));
        SX(dev_alloc(&c->d_sx_send_idx, std::max<u32>(c->sx_send_off[world], 1)));
        SX(dev_alloc(&c->d_sx_sendbuf, std::max<u32>(c->sx_send_off[world], 1)));
        hipLaunchKernelGGL(k_sx_remap, dim3((u32)((c->n_active + 255) / 256)), dim3(256), 0, st, d_sids, d_code, c->n_active, world,
                           c->hub_real, c->xs0, L1, c->n_loc_pad, d_ref, d_pos, c->d_sx_map);
        hipLaunchKernelGGL(k_sx_lists, dim3((u32)((cnt + 255) / 256)), dim3(256), 0, st, d_want, d_spos, cnt, L1, c->xs0, c->d_sx_send_idx);
        SX_HIP(hipStreamSynchronize(st));
        // What travels is agreed on by CONTENT, not only by length (round 5, ADVICE r4: with the sharded hand-over the send lists come
        // from this rank's own rows through the matrix's symmetry, so a caller's CSR that is not symmetric could give lists of the
        // right lengths and the wrong members): one order-dependent 64-bit hash per peer of what this rank packs for it and of
        // what it expects from it, compared pairwise beside the counts (lzx_comm_check_sparse; in-process groups: lzx_api.hip).
        {
            std::vector<u32> h_idx(c->sx_send_off[world]), h_map(c->xc1);
            if (!h_idx.empty()) SX_HIP(hipMemcpy(h_idx.data(), c->d_sx_send_idx, sizeof(u32) * h_idx.size(), hipMemcpyDeviceToHost));
            if (!h_map.empty()) SX_HIP(hipMemcpy(h_map.data(), c->d_sx_map, sizeof(u32) * h_map.size(), hipMemcpyDeviceToHost));
            c->sx_send_hash.assign(world, 0);
            c->sx_recv_hash.assign(world, 0);
            for (u32 p = 0; p < world; ++p) {
                u64 hs = 0, hr = 0;
                for (u32 i = c->sx_send_off[p]; i < c->sx_send_off[p + 1]; ++i) hs += lzx_mix64((u64)h_idx[i] + ((u64)(i - c->sx_send_off[p]) << 32));
                for (u32 i = c->sx_recv_off[p]; i < c->sx_recv_off[p + 1]; ++i)
                    hr += lzx_mix64((u64)(h_map[i] - p * c->n_loc_pad) + ((u64)(i - c->sx_recv_off[p]) << 32));
                c->sx_send_hash[p] = hs;
                c->sx_recv_hash[p] = hr;
            }
        }
        sx_free();
#undef SX
#undef SX_HIP
        c->sparse = true;
        c->xlen = (u64)world * c->xs0 + round_up((u32)c->xc1, LZX_SLICE) + LZX_TAIL;
    }
    // ---- RCCL: the hand-over's one sync point.  Everything above is rank-local (allocations, sorts, scans) and may fail on
    //      one rank alone; what follows it for the sparse chunk is a collective.  Every rank votes here -- one that failed
    //      earlier through its hand-over entry point's lzx_agree_guard -- and all of them go on, or none does.
    if (c->agree_pending) {
        bool all_ok = false;
        c->agree_pending = false;
        rc = lzx_comm_agree(c, true, &all_ok);
        if (rc != LZX_OK) { cleanup(); return rc; }
        if (!all_ok) { cleanup(); LZX_FAIL(LZX_ERR_STATE, "graph hand-over: a peer rank failed while reshaping its share (see that rank's error)"); }
    }
    if (c->sparse) {
        rc = lzx_comm_check_sparse(c);   // RCCL, peer windows: every pair of ranks agrees on what travels, or all of them fail here
        if (rc != LZX_OK) { cleanup(); return rc; }
    }
    // peer windows: the hand-over ends with a second collective (the receive buffers allocated below become reachable for the
    // peers); a rank that fails before it still takes part, through the entry point's lzx_agree_guard
    c->publish_pending = c->comm_kind == 3 && lzx_exchanges(c);

    // ---- 2. this rank's rows ----
    PREP(dev_alloc(&d_old_of_local, c->n_loc_real)); PREP(dev_alloc(&d_deg_local, c->n_loc_real));
    if (c->n_loc_real) {
        hipLaunchKernelGGL(k_local_rows, dim3((c->n_loc_real + 255) / 256), dim3(256), 0, st, d_sids, d_sdeg,
                           d_old_of_local, d_deg_local, c->n_loc_real, world, rank);
    }
    degl.assign(c->n_loc_pad, 0);
    if (c->n_loc_real)
        PREP_HIP(hipMemcpyAsync(degl.data(), d_deg_local, sizeof(u32) * c->n_loc_real, hipMemcpyDeviceToHost, st));
    PREP_HIP(hipStreamSynchronize(st));
    c->nnz_local = 0;
    for (u32 l = 0; l < c->n_loc_real; ++l) c->nnz_local += degl[l];
    // the whole graph may hold more than 2^32 entries (every rank keeps all of it: C5's 4e9 entries are 17 GB of the
    // 288 GB); a rank's own share is indexed with 32 bits
    if (c->nnz_local >= (1ull << 32)) { cleanup(); LZX_FAIL(LZX_ERR_LIMIT, "%llu entries in this rank's rows: use more ranks (limit 2^32 per rank)", (unsigned long long)c->nnz_local); }

    // propagation-blocking mode: k_spmv keeps only the hub entries of each row; `degl` becomes that count
    u64 pb_total = 0;
    if (pb && c->n_loc_real) {
        PREP(dev_alloc(&d_hub_deg, c->n_loc_real)); PREP(dev_alloc(&d_nh_deg, (u64)c->n_loc_real + 1));
        PREP(dev_alloc(&d_nh_off, (u64)c->n_loc_real + 1));
        PREP_HIP(hipMemsetAsync(d_nh_deg, 0, sizeof(u32) * ((u64)c->n_loc_real + 1), st));
        hipLaunchKernelGGL(k_row_hub_count, dim3(std::min<u32>(c->n_loc_real, 1u << 22)), dim3(64), 0, st, c->d_row_ptr, c->d_col_idx,
                           d_code, d_old_of_local, d_deg_local, c->n_loc_real, c->hub_real, d_hub_deg, d_nh_deg);
        h_nh.assign(c->n_loc_real, 0);
        PREP_HIP(hipMemcpyAsync(degl.data(), d_hub_deg, sizeof(u32) * c->n_loc_real, hipMemcpyDeviceToHost, st));
        PREP_HIP(hipMemcpyAsync(h_nh.data(), d_nh_deg, sizeof(u32) * c->n_loc_real, hipMemcpyDeviceToHost, st));
        PREP_HIP(hipStreamSynchronize(st));
        u64 hub_total = 0;
        for (u32 l = 0; l < c->n_loc_real; ++l) hub_total += degl[l];
        pb_total = c->nnz_local - hub_total;
        if (pb_total >= LZX_PB_SLOT_LIMIT) {
            lzx_set_error("propagation blocking: %llu entries on this rank exceed the 32-bit slot range", (unsigned long long)pb_total);
            cleanup();
            return LZX_ERR_LIMIT;
        }
        // pb_total == 0: every entry of this rank's rows has a staged column -- the hub-only structures are the whole
        // matrix and no blocked tables are built (the layout decisions above stay as they are: other ranks block).
        if (pb_total > 0) {
            size_t sb = 0;
            PREP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, sb, d_nh_deg, d_nh_off, (u64)c->n_loc_real + 1, st));
            if (d_tmp) { (void)hipFree(d_tmp); d_tmp = nullptr; }
            PREP_HIP(hipMalloc(&d_tmp, sb ? sb : 16));
            PREP_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp, sb, d_nh_deg, d_nh_off, (u64)c->n_loc_real + 1, st));
        }
    }

    // ---- 3. host: split rows / slices (local rows are sorted by degree, descending) ----
    // blocked mode: the tables hold staged columns only and a lane's packets come from LDS-only sums, so wider slices
    // cost less than more split-row items: 256 staged entries per row -- and 1024 where a wavefront has many slices to
    // even out the long ones (from 2 Mi rows with an edge per rank).  Round 4, parity-controlled A/B (every configuration in
    // both process states, profiles/r4_long_row.txt): 10 M vertices 0.572 -> 0.556 ms per SpMV (768: 0.551; 512 and
    // 1536: - 1 %), 4 M vertices 0.213 -> 0.206; the 1 M-vertex graph (2.5 slices per wavefront) LOSES with wider slices
    // (512: + 4 %, 1024: + 42 %: a wavefront's one long slice is the launch's tail), hence the size rule.
    const u32 long_thr = c->long_row_opt > 0 ? (u32)c->long_row_opt
                         : (pb ? (c->rows_live >= (1u << 21) ? 8 * LZX_LONG_ROW : 2 * LZX_LONG_ROW) : LZX_LONG_ROW);
    u32 n_long = 0;
    for (u32 l = 0; l < c->n_loc_real; ++l)
        if (degl[l] > long_thr) n_long = l + 1;
    c->n_long_true = n_long;
    c->n_long64 = std::min(round_up(n_long, LZX_SLICE), c->n_loc_pad);
    c->n_slices = (c->n_loc_pad - c->n_long64) / LZX_SLICE;

    h_long_ptr.assign((size_t)c->n_long64 + 1, 0);
    h_item_first.assign((size_t)c->n_long64 + 1, 0);
    for (u32 r = 0; r < c->n_long64; ++r) {
        const u32 dp = round_up(degl[r], 4);
        h_long_ptr[r + 1] = h_long_ptr[r] + dp;
        h_item_first[r] = (u32)h_item_beg.size();
        const u32 item_cap = c->item_opt > 0 ? (u32)c->item_opt : LZX_ITEM;
        for (u32 b = 0; b < dp; b += item_cap) {
            h_item_beg.push_back(h_long_ptr[r] + b);
            h_item_len.push_back(std::min(item_cap, dp - b));
        }
    }
    h_item_first[c->n_long64] = (u32)h_item_beg.size();
    c->n_items = (u32)h_item_beg.size();
    c->long_elems = h_long_ptr[c->n_long64];

    h_slice_off.assign(c->n_slices, 0);
    h_slice_w.assign(c->n_slices, 0);
    u64 off = 0;
    for (u32 s = 0; s < c->n_slices; ++s) {
        u32 widest = 0;  // rows are degree-sorted, but in PB mode `degl` counts hub entries only: take the max
        for (u32 l = 0; l < LZX_SLICE; ++l) widest = std::max(widest, degl[c->n_long64 + s * LZX_SLICE + l]);
        const u32 w = round_up(widest, 4);
        h_slice_off[s] = off;
        h_slice_w[s] = w;
        off += (u64)w * LZX_SLICE;
    }
    c->sell_elems = off;

    // ---- 4. upload tables, fill column codes ----
    PREP(dev_alloc(&c->d_slice_off, c->n_slices)); PREP(dev_alloc(&c->d_slice_w, c->n_slices));
    // + 1 KiB: the SpMV's pipelined loads read one (clamped) packet row past a zero-width last slice / short item
    // (staged-only tables of the propagation-blocking mode hold 16-bit codes: half the bytes)
    c->codes16 = pb;
    PREP(dev_alloc(&c->d_sell_cols, pb ? (c->sell_elems + 1025) / 2 + 512 : c->sell_elems + 1024));
    PREP(dev_alloc(&c->d_long_cols, pb ? (c->long_elems + 1025) / 2 + 512 : c->long_elems + 1024));
    PREP(dev_alloc(&c->d_item_beg, c->n_items)); PREP(dev_alloc(&c->d_item_len, c->n_items));
    PREP(dev_alloc(&c->d_item_first, (u64)c->n_long64 + 1));
    PREP(dev_alloc(&c->d_long_partial, c->n_items));
    PREP(dev_alloc(&d_long_ptr, (u64)c->n_long64 + 1));
    {   // processing order of the slices: by class in blocked mode (k_spmv), as they lie otherwise
        std::vector<u32> perm;
        perm.reserve(c->n_slices);
        c->h_slice_w0.clear();
        // classes pay from a few slices per wavefront on (each class loop starts with its own descriptor round trips:
        // on the 1 M-vertex graph, 2.4 slices per wavefront, they cost 1.6 us)
        const bool classes = pb && c->narrow_opt != 0 && (c->narrow_opt > 0 || c->n_slices >= 8u * (u32)c->cu_count * (LZX_SPMV_BLOCK / 64));
        for (u32 s2 = 0; s2 < c->n_slices; ++s2)
            if (!classes || h_slice_w[s2] > 8) perm.push_back(s2);
        c->ns_wide = (u32)perm.size();
        for (u32 s2 = 0; classes && s2 < c->n_slices; ++s2)
            if (h_slice_w[s2] == 8) perm.push_back(s2);
        c->ns_w8 = (u32)perm.size() - c->ns_wide;
        for (u32 s2 = 0; classes && s2 < c->n_slices; ++s2)
            if (h_slice_w[s2] == 4) perm.push_back(s2);
        c->ns_w4 = (u32)perm.size() - c->ns_wide - c->ns_w8;
        for (u32 s2 = 0; classes && s2 < c->n_slices; ++s2)
            if (h_slice_w[s2] == 0) { perm.push_back(s2); c->h_slice_w0.push_back(s2); }
        PREP(dev_alloc(&c->d_slice_perm, c->n_slices));
        if (c->n_slices)
            PREP_HIP(hipMemcpyAsync(c->d_slice_perm, perm.data(), sizeof(u32) * c->n_slices, hipMemcpyHostToDevice, st));
        PREP_HIP(hipStreamSynchronize(st));   // perm is a local
    }
    if (c->n_slices) {
        PREP_HIP(hipMemcpyAsync(c->d_slice_off, h_slice_off.data(), sizeof(u64) * c->n_slices, hipMemcpyHostToDevice, st));
        PREP_HIP(hipMemcpyAsync(c->d_slice_w, h_slice_w.data(), sizeof(u32) * c->n_slices, hipMemcpyHostToDevice, st));
    }
    if (c->n_items) {
        PREP_HIP(hipMemcpyAsync(c->d_item_beg, h_item_beg.data(), sizeof(u64) * c->n_items, hipMemcpyHostToDevice, st));
        PREP_HIP(hipMemcpyAsync(c->d_item_len, h_item_len.data(), sizeof(u32) * c->n_items, hipMemcpyHostToDevice, st));
    }
    PREP_HIP(hipMemcpyAsync(c->d_item_first, h_item_first.data(), sizeof(u32) * ((size_t)c->n_long64 + 1), hipMemcpyHostToDevice, st));
    PREP_HIP(hipMemcpyAsync(d_long_ptr, h_long_ptr.data(), sizeof(u64) * ((size_t)c->n_long64 + 1), hipMemcpyHostToDevice, st));
    if (c->n_slices) {
        const u32 rows = c->n_loc_pad - c->n_long64;
        if (pb)
            hipLaunchKernelGGL(k_fill_sell_hub, dim3((rows + 255) / 256), dim3(256), 0, st, c->d_row_ptr, c->d_col_idx,
                               d_code, d_old_of_local, d_deg_local, c->n_loc_real, c->n_long64, c->n_loc_pad,
                               c->d_slice_off, c->d_slice_w, reinterpret_cast<uint16_t *>(c->d_sell_cols), c->hub_real, sentinel);
        else
            hipLaunchKernelGGL(k_fill_sell, dim3((rows + 255) / 256), dim3(256), 0, st, c->d_row_ptr, c->d_col_idx,
                               d_code, d_old_of_local, d_deg_local, c->n_loc_real, c->n_long64, c->n_loc_pad,
                               c->d_slice_off, c->d_slice_w, c->d_sell_cols, sentinel);
    }
    if (c->n_long64) {
        if (pb)
            hipLaunchKernelGGL(k_fill_long_hub, dim3(c->n_long64), dim3(64), 0, st, c->d_row_ptr, c->d_col_idx, d_code,
                               d_old_of_local, d_deg_local, c->n_loc_real, d_long_ptr, reinterpret_cast<uint16_t *>(c->d_long_cols),
                               c->hub_real, sentinel);
        else
            hipLaunchKernelGGL(k_fill_long, dim3(c->n_long64), dim3(256), 0, st, c->d_row_ptr, c->d_col_idx, d_code,
                               d_old_of_local, d_deg_local, c->n_loc_real, d_long_ptr, c->d_long_cols, sentinel);
    }
    PREP_HIP(hipGetLastError());
    if (pb && pb_total > 0) {
        if (d_tmp) { (void)hipFree(d_tmp); d_tmp = nullptr; }
        PREP(lzx_pb_prepare(c, d_code, d_old_of_local, d_deg_local, d_nh_off, h_nh, pb_total));
    }

    // ---- 5. vectors ----
    PREP(dev_alloc(&c->d_v, c->ldq));
    {
        PREP(dev_alloc(&c->d_u[0], c->ldq)); PREP(dev_alloc(&c->d_u[1], c->ldq));
        PREP_HIP(hipMemsetAsync(c->d_u[0], 0, sizeof(double) * c->ldq, st));
        PREP_HIP(hipMemsetAsync(c->d_u[1], 0, sizeof(double) * c->ldq, st));
    }
    PREP(dev_alloc(&c->d_xbuf, c->xlen)); PREP(dev_alloc(&c->d_ybuf, c->iolen));
    if (c->xfp32) { PREP(dev_alloc(&c->d_xf32_send, c->xs)); PREP(dev_alloc(&c->d_xf32_full, (u64)world * c->xs)); }
    PREP(dev_alloc(&c->d_io, n));
    PREP_HIP(hipMemsetAsync(c->d_v, 0, sizeof(double) * c->ldq, st));
    PREP_HIP(hipMemsetAsync(c->d_xbuf, 0, sizeof(double) * c->xlen, st));
    PREP_HIP(hipMemsetAsync(c->d_ybuf, 0, sizeof(double) * c->iolen, st));

    // launch shape: persistent workgroups, as many as stay resident (LDS-limited), never more than
    // there is work for.
    // 4 workgroups per CU: two are resident at the default 64 KiB of staged x, the rest queue behind them, which evens
    // out the tail (measured: C3 2.16 -> 2.15 ms, C2 0.141 -> 0.133 ms against exactly-resident grids).
    // In propagation-blocking mode the kernel stages 128 KiB per workgroup and only one is resident: one per CU.
    u32 per_cu = pb ? 1 : 4;
    if (c->wgs_per_cu_opt > 0) per_cu = (u32)c->wgs_per_cu_opt;
    const u32 units = c->n_slices + c->n_items;
    const u32 waves_per_wg = LZX_SPMV_BLOCK / 64;
    u32 grid = (u32)c->cu_count * per_cu;
    grid = std::min(grid, std::max(1u, (units + waves_per_wg - 1) / waves_per_wg));
    // Blocked mode, little staged-columns work per workgroup (below 512 slices + items each: graphs up to a few million vertices,
    // rank shares from P = 2): half as many, twice as long.  They share the scatter pass's launch; the CUs they leave free start on scatter units
    // at once, and 128 KiB of staged x are fetched half as often (1 M-vertex graph: SpMV 0.0652 -> 0.0595 ms; rank 0 of 8 / of 4:
    // 0.109 -> 0.100, 0.179 -> 0.173 ms; 4 M vertices and rank 0 of 2: - 2 %; neutral on the 10 M-vertex graph -- profiles/r3_small_units.txt)
    if (pb && units < (u32)c->cu_count * 512u) grid = std::min(grid, std::max(1u, (u32)c->cu_count / 2));
    if (c->spmv_wgs_opt > 0) grid = std::min<u32>((u32)c->cu_count * per_cu, std::min<u32>(std::max(1u, (units + waves_per_wg - 1) / waves_per_wg), (u32)c->spmv_wgs_opt));   // test shape spmv_wgs
    c->spmv_grid = grid;
    c->fin_grid = (c->n_long64 + LZX_VEC_BLOCK - 1) / LZX_VEC_BLOCK;
    c->np_cap = std::max<u32>(c->spmv_grid + c->fin_grid + lzx_pb_partials(c), (u32)c->cu_count * 8) + 8;
    PREP(dev_alloc(&c->d_partials, c->np_cap)); PREP(dev_alloc(&c->d_partials2, c->np_cap)); PREP(dev_alloc(&c->d_partials3, c->np_cap));

    PREP_HIP(hipStreamSynchronize(st));
    cleanup();
#undef PREP
#undef PREP_HIP
    rc = lzx_pb_place_values(c);   // option placement_trials: the value stream where the SpMV runs fastest (rank-local, no collective)
    if (c->publish_pending) {
        c->publish_pending = false;
        const int rp = lzx_comm_ipc_publish(c, rc == LZX_OK);
        return rc != LZX_OK ? rc : rp;
    }
    return rc;
}

// --------------------------------------------------------------------------------------------------
// ingest: sorted unique directed keys (row << 32 | col) -> CSR
__global__ void k_keys_to_csr(const u64 *keys, u64 nkeys, u64 n, u64 *row_ptr, u32 *col_idx)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nkeys) col_idx[i] = (u32)(keys[i] & 0xffffffffu);
    if (i <= n) {
        // first key whose row >= i
        u64 lo = 0, hi = nkeys;
        while (lo < hi) {
            const u64 mid = (lo + hi) >> 1;
            if ((keys[mid] >> 32) < i) lo = mid + 1; else hi = mid;
        }
        row_ptr[i] = lo;
    }
}

// d_keys[nkeys] unsorted, ~0 marks a dropped entry.  Consumes (frees) d_keys.
static int csr_from_keys_dev(lzx_ctx *c, u64 n, u64 *d_keys, u64 nkeys)
{
    hipStream_t st = c->stream;
    u64 *d_sorted = nullptr, *d_uniq = nullptr, *d_count = nullptr;
    void *d_tmp = nullptr;
    int rc = LZX_OK;
    auto cleanup = [&]() {
        dev_free(d_keys); dev_free(d_sorted); dev_free(d_uniq); dev_free(d_count);
        if (d_tmp) (void)hipFree(d_tmp);
        d_tmp = nullptr;
    };
#define ING(call) do { rc = (call); if (rc != LZX_OK) { cleanup(); return rc; } } while (0)
#define ING_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
        lzx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); cleanup(); \
        return e_ == hipErrorOutOfMemory ? LZX_ERR_NOMEM : LZX_ERR_HIP; } } while (0)
    // one work-item per key below, and a launch holds at most 2^32 of them
    if (nkeys >= (1ull << 32) - (1ull << 24)) { cleanup(); LZX_FAIL(LZX_ERR_LIMIT, "edge list too long for one ingest (%llu directed entries)", (unsigned long long)nkeys); }
    ING(dev_alloc(&d_sorted, nkeys)); ING(dev_alloc(&d_count, 1));
    size_t tb = 0;
    ING_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, d_keys, d_sorted, (u64)nkeys, 0, 64, st));
    ING_HIP(hipMalloc(&d_tmp, tb ? tb : 16));
    ING_HIP(hipcub::DeviceRadixSort::SortKeys(d_tmp, tb, d_keys, d_sorted, (u64)nkeys, 0, 64, st));
    (void)hipFree(d_tmp); d_tmp = nullptr;
    dev_free(d_keys);
    ING(dev_alloc(&d_uniq, nkeys));
    tb = 0;
    ING_HIP(hipcub::DeviceSelect::Unique(nullptr, tb, d_sorted, d_uniq, d_count, (u64)nkeys, st));
    ING_HIP(hipMalloc(&d_tmp, tb ? tb : 16));
    ING_HIP(hipcub::DeviceSelect::Unique(d_tmp, tb, d_sorted, d_uniq, d_count, (u64)nkeys, st));
    u64 cnt = 0, last = 0;
    ING_HIP(hipMemcpyAsync(&cnt, d_count, sizeof(u64), hipMemcpyDeviceToHost, st));
    ING_HIP(hipStreamSynchronize(st));
    if (cnt > 0) {
        ING_HIP(hipMemcpyAsync(&last, d_uniq + (cnt - 1), sizeof(u64), hipMemcpyDeviceToHost, st));
        ING_HIP(hipStreamSynchronize(st));
        if (last == ~0ull) --cnt;  // the dropped-entry marker sorts last
    }
    lzx_graph_release(c);
    c->n = n;
    c->nnz = cnt;
    ING(dev_alloc(&c->d_row_ptr, n + 1)); ING(dev_alloc(&c->d_col_idx, cnt));
    const u64 work = std::max<u64>(cnt, n + 1);
    hipLaunchKernelGGL(k_keys_to_csr, dim3((u32)((work + 255) / 256)), dim3(256), 0, st, d_uniq, cnt, n,
                       c->d_row_ptr, c->d_col_idx);
    ING_HIP(hipGetLastError());
    ING_HIP(hipStreamSynchronize(st));
    cleanup();
#undef ING
#undef ING_HIP
    return lzx_graph_prepare(c);
}

__global__ void k_edges_to_keys(const u32 *src, const u32 *dst, u64 m, u64 n, u64 *keys, u32 *bad)
{
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m) return;
    const u64 u = src[e], v = dst[e];
    // a self loop is ONE diagonal entry, as in the std::set build (both inserted keys are equal there)
    const bool ok = u < n && v < n;
    if (!ok) bad[0] = 1u;
    keys[2 * e] = ok ? ((u << 32) | v) : ~0ull;
    keys[2 * e + 1] = ok && u != v ? ((v << 32) | u) : ~0ull;
}

// oracle/lanczos_oracle.c: orc_gen_er_keys / orc_gen_rmat_keys, one thread per draw.
__global__ void k_gen_keys(int kind, u32 scale, u64 n, u64 draws, u64 seed, u32 ta, u32 tab, u32 tabc,
                           u64 *keys)
{
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= draws) return;
    u64 u = 0, v = 0;
    const bool ok = gen_draw(kind, scale, n, seed, ta, tab, tabc, e, u, v);
    keys[2 * e] = ok ? ((u << 32) | v) : ~0ull;
    keys[2 * e + 1] = ok ? ((v << 32) | u) : ~0ull;
}

// The sharded hand-over itself: sweep 1 here (the reshaping needs the entry count before it starts), sweeps 2 and 3 inside
// lzx_graph_prepare, at the points where the whole-graph hand-over reads the CSR.  `src` is valid during this call only.
static int shard_handover(lzx_ctx *c, const lzx_ctx::lzx_key_source &src)
{
    lzx_graph_release(c);
    c->n = src.n;
    c->shard = src;
    int rc = dev_alloc(&c->d_shard_deg, src.n);
    if (rc == LZX_OK && src.kind == 3) {   // a CSR carries its degrees: row_ptr alone crosses for them
        u64 *d_rp = nullptr;
        u32 *d_ids = nullptr;
        rc = dev_alloc(&d_rp, src.n + 1);
        if (rc == LZX_OK) rc = dev_alloc(&d_ids, src.n);
        if (rc == LZX_OK) {
            hipError_t e = hipMemcpyAsync(d_rp, src.h_row_ptr, sizeof(u64) * (src.n + 1), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_degrees, dim3((u32)((src.n + 255) / 256)), dim3(256), 0, c->stream, d_rp, c->d_shard_deg, d_ids, src.n);
                e = hipStreamSynchronize(c->stream);
            }
            if (e != hipSuccess) { lzx_set_error("sharded hand-over: %s", hipGetErrorString(e)); rc = LZX_ERR_HIP; }
        }
        dev_free(d_rp); dev_free(d_ids);
    } else if (rc == LZX_OK) rc = shard_count_sweep(c, 0, nullptr, 0, c->d_shard_deg);
    if (rc == LZX_OK) {
        unsigned long long *d_tot = nullptr, tot = 0;
        rc = dev_alloc(&d_tot, 1);
        if (rc == LZX_OK) {
            hipError_t e = hipMemsetAsync(d_tot, 0, sizeof(unsigned long long), c->stream);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_shard_sum_deg, dim3(1024), dim3(256), 0, c->stream, c->d_shard_deg, src.n, d_tot);
                e = hipMemcpyAsync(&tot, d_tot, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            dev_free(d_tot);
            if (e != hipSuccess) { lzx_set_error("sharded hand-over: %s", hipGetErrorString(e)); rc = LZX_ERR_HIP; }
            c->nnz = tot;
        }
    }
    if (rc == LZX_OK) rc = lzx_graph_prepare(c);
    c->shard = lzx_ctx::lzx_key_source();
    dev_free(c->d_shard_deg);
    if (rc != LZX_OK) {
        const std::string msg = lzx_last_error();   // the release below must not lose the reason
        lzx_graph_release(c);
        lzx_set_error("%s", msg.c_str());
        return rc;
    }
    c->sharded = c->world > 1;   // at one rank the CSR on the device is the whole graph after all
    return LZX_OK;
}

// sharded hand-over of an edge list: the range check alone (the keys are formed batch by batch later)
__global__ void k_check_endpoints(const u32 *src, const u32 *dst, u64 m, u64 n, u32 *bad)
{
    const u64 nthreads = (u64)gridDim.x * blockDim.x;
    bool b = false;
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += nthreads) b |= src[e] >= n || dst[e] >= n;
    if (b) bad[0] = 1u;
}

extern "C" int lzx_set_graph_edges(lzx_handle c, uint64_t n, uint64_t m, const uint32_t *src, const uint32_t *dst)
{
    if (!c || (m && (!src || !dst)) || n == 0) LZX_FAIL(LZX_ERR_ARG, "lzx_set_graph_edges: bad argument");
    lzx_agree_guard guard(c);   // RCCL: a local failure below still votes at the hand-over's sync point (lzx_internal.h)
    LZX_HIP(hipSetDevice(c->device));
    const bool shard = c->shard_opt > 0;   // the endpoint pairs stay resident (8 bytes per edge) and are swept; the 2 m keys are never formed at once
    u32 *d_src = nullptr, *d_dst = nullptr, *d_bad = nullptr, bad = 0;
    u64 *d_keys = nullptr;
    LZX_TRY(dev_alloc(&d_src, m));
    if (dev_alloc(&d_dst, m) != LZX_OK) { dev_free(d_src); return LZX_ERR_NOMEM; }
    if (!shard && dev_alloc(&d_keys, 2 * m) != LZX_OK) { dev_free(d_src); dev_free(d_dst); return LZX_ERR_NOMEM; }
    if (dev_alloc(&d_bad, 1) != LZX_OK) { dev_free(d_src); dev_free(d_dst); dev_free(d_keys); return LZX_ERR_NOMEM; }
    hipError_t e = hipSuccess;
    if (m) {
        e = hipMemcpyAsync(d_src, src, sizeof(u32) * m, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_dst, dst, sizeof(u32) * m, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, sizeof(u32), c->stream);
        if (e == hipSuccess) {
            if (shard) hipLaunchKernelGGL(k_check_endpoints, dim3(1024), dim3(256), 0, c->stream, d_src, d_dst, m, n, d_bad);
            else hipLaunchKernelGGL(k_edges_to_keys, dim3((u32)((m + 255) / 256)), dim3(256), 0, c->stream, d_src, d_dst, m, n, d_keys, d_bad);
            e = hipMemcpyAsync(&bad, d_bad, sizeof(u32), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        }
    }
    dev_free(d_bad);
    int rc = LZX_OK;
    if (e != hipSuccess) { lzx_set_error("edge upload: %s", hipGetErrorString(e)); rc = LZX_ERR_HIP; }
    else if (bad) { lzx_set_error("lzx_set_graph_edges: an endpoint is >= n"); rc = LZX_ERR_ARG; }
    else if (shard) {
        lzx_ctx::lzx_key_source ks;
        ks.kind = 2; ks.n = n; ks.d_src = d_src; ks.d_dst = d_dst; ks.m = m;
        rc = shard_handover(c, ks);
    }
    dev_free(d_src); dev_free(d_dst);
    if (rc != LZX_OK || shard) { dev_free(d_keys); return rc; }
    return csr_from_keys_dev(c, n, d_keys, 2 * m);
}

extern "C" int lzx_gen_graph(lzx_handle c, int kind, uint32_t scale, uint64_t n, uint64_t draws, uint64_t seed,
                             uint32_t ta, uint32_t tab, uint32_t tabc)
{
    if (!c || n == 0 || (kind != 0 && kind != 1)) LZX_FAIL(LZX_ERR_ARG, "lzx_gen_graph: bad argument");
    if (kind == 1 && (scale == 0 || scale > 32 || n > (1ull << scale))) LZX_FAIL(LZX_ERR_ARG, "lzx_gen_graph: scale/n mismatch");
    lzx_agree_guard guard(c);   // RCCL: a local failure below still votes at the hand-over's sync point (lzx_internal.h)
    LZX_HIP(hipSetDevice(c->device));
    if (c->shard_opt > 0) {
        lzx_ctx::lzx_key_source ks;
        ks.kind = kind; ks.scale = scale; ks.n = n; ks.draws = draws; ks.seed = seed; ks.ta = ta; ks.tab = tab; ks.tabc = tabc;
        return shard_handover(c, ks);
    }
    u64 *d_keys = nullptr;
    LZX_TRY(dev_alloc(&d_keys, 2 * draws));
    if (draws) {
        hipLaunchKernelGGL(k_gen_keys, dim3((u32)((draws + 255) / 256)), dim3(256), 0, c->stream, kind, scale, n,
                           draws, seed, ta, tab, tabc, d_keys);
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { dev_free(d_keys); LZX_FAIL(LZX_ERR_HIP, "generator: %s", hipGetErrorString(e)); }
    }
    return csr_from_keys_dev(c, n, d_keys, 2 * draws);
}

// flag[0] = 1 if any column index is out of range (the reshaping kernels index tables by it unchecked)
__global__ void k_check_cols(const u32 *col_idx, u64 nnz, u32 n, u32 *flag)
{
    const u64 nthreads = (u64)gridDim.x * blockDim.x;
    bool bad = false;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += nthreads) bad |= col_idx[i] >= n;
    if (bad) flag[0] = 1u;
}

static int set_csr_common(lzx_ctx *c, u64 n, u64 nnz, const u64 *row_ptr64, const u32 *row_ptr32, const u32 *col_idx)
{
    if (!c || n == 0 || (!row_ptr64 && !row_ptr32) || (nnz && !col_idx)) LZX_FAIL(LZX_ERR_ARG, "lzx_set_graph_csr: bad argument");
    lzx_agree_guard guard(c);   // RCCL: a local failure below still votes at the hand-over's sync point (lzx_internal.h)
    LZX_HIP(hipSetDevice(c->device));
    std::vector<u64> widened;
    if (!row_ptr64) {
        widened.resize(n + 1);
        for (u64 i = 0; i <= n; ++i) widened[i] = row_ptr32[i];
        row_ptr64 = widened.data();
    }
    if (row_ptr64[0] != 0 || row_ptr64[n] != nnz) LZX_FAIL(LZX_ERR_ARG, "lzx_set_graph_csr: row_ptr[0] must be 0 and row_ptr[n] must equal nnz");
    if (n > 0xffffffffull) LZX_FAIL(LZX_ERR_LIMIT, "lzx_set_graph_csr: more than 2^32 vertices");
    for (u64 i = 0; i < n; ++i)
        if (row_ptr64[i] > row_ptr64[i + 1]) LZX_FAIL(LZX_ERR_ARG, "lzx_set_graph_csr: row_ptr decreases at row %llu", (unsigned long long)i);
    if (c->shard_opt > 0) {   // sharded hand-over: the CSR stays with the caller and is streamed; this rank keeps its own rows
        lzx_ctx::lzx_key_source ks;
        ks.kind = 3; ks.n = n; ks.h_row_ptr = row_ptr64; ks.h_col_idx = col_idx;
        return shard_handover(c, ks);
    }
    lzx_graph_release(c);
    c->n = n;
    c->nnz = nnz;
    LZX_TRY(dev_alloc(&c->d_row_ptr, n + 1));
    LZX_TRY(dev_alloc(&c->d_col_idx, nnz));
    LZX_HIP(hipMemcpyAsync(c->d_row_ptr, row_ptr64, sizeof(u64) * (n + 1), hipMemcpyHostToDevice, c->stream));
    if (nnz) LZX_HIP(hipMemcpyAsync(c->d_col_idx, col_idx, sizeof(u32) * nnz, hipMemcpyHostToDevice, c->stream));
    if (nnz) {   // a caller's out-of-range column would index the reshaping tables out of bounds: refuse it here
        u32 *d_flag = nullptr, flag = 0;
        LZX_TRY(dev_alloc(&d_flag, 1));
        LZX_HIP(hipMemsetAsync(d_flag, 0, sizeof(u32), c->stream));
        k_check_cols<<<1024, 256, 0, c->stream>>>(c->d_col_idx, nnz, (u32)n, d_flag);
        LZX_HIP(hipMemcpyAsync(&flag, d_flag, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
        LZX_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(d_flag);
        if (flag) {
            lzx_graph_release(c);
            LZX_FAIL(LZX_ERR_ARG, "lzx_set_graph_csr: a column index is >= n");
        }
    }
    LZX_HIP(hipStreamSynchronize(c->stream));
    return lzx_graph_prepare(c);
}

extern "C" int lzx_set_graph_csr(lzx_handle c, uint64_t n, uint64_t nnz, const uint64_t *row_ptr, const uint32_t *col_idx)
{
    return set_csr_common(c, n, nnz, row_ptr, nullptr, col_idx);
}

extern "C" int lzx_set_graph_csr32(lzx_handle c, uint32_t n, uint32_t nnz, const uint32_t *row_ptr, const uint32_t *col_idx)
{
    return set_csr_common(c, n, nnz, nullptr, row_ptr, col_idx);
}

extern "C" int lzx_get_graph_csr(lzx_handle c, uint64_t *row_ptr, uint32_t *col_idx)
{
    if (!c || !row_ptr || !col_idx) LZX_FAIL(LZX_ERR_ARG, "lzx_get_graph_csr: bad argument");
    if (!c->d_row_ptr) LZX_FAIL(LZX_ERR_STATE, "no graph set");
    if (c->sharded) LZX_FAIL(LZX_ERR_STATE, "lzx_get_graph_csr: the graph came through the sharded hand-over -- no rank holds all of it");
    LZX_HIP(hipSetDevice(c->device));
    LZX_HIP(hipMemcpy(row_ptr, c->d_row_ptr, sizeof(u64) * (c->n + 1), hipMemcpyDeviceToHost));
    if (c->nnz) LZX_HIP(hipMemcpy(col_idx, c->d_col_idx, sizeof(u32) * c->nnz, hipMemcpyDeviceToHost));
    return LZX_OK;
}

extern "C" int lzx_get_graph_info(lzx_handle c, lzx_graph_info *o)
{
    if (!c || !o) LZX_FAIL(LZX_ERR_ARG, "lzx_get_graph_info: bad argument");
    if (!c->d_row_ptr) LZX_FAIL(LZX_ERR_STATE, "no graph set");
    o->n = c->n;
    o->nnz = c->nnz;
    o->max_degree = c->max_degree;
    o->rows_local = c->n_loc_real;
    o->nnz_local = c->nnz_local;
    o->long_rows = c->n_long_true;
    o->sell_padded = c->sell_elems + c->long_elems;
    o->hub_entries = c->hub_real;
    o->pb_entries = c->pb ? c->pb_entries : 0;
    o->pb_values = c->pb ? c->pb_values : 0;
    o->pb_reduced_entries = c->pb ? c->pbr_entries : 0;
    o->reserved_ = 0;
    o->exchange_chunk0 = c->overlap ? c->xs0 : 0;
    o->exchange_recv = 0;
    if (c->world > 1 || c->sparse) {
        o->exchange_recv = (u64)(c->world - 1) * c->xs / (c->xfp32 ? 2 : 1);   // in doubles: fp32 entries count half
        if (c->sparse)
            o->exchange_recv = (u64)(c->world - 1) * c->xs0 + c->xc1 - (c->sx_recv_off[c->rank + 1] - c->sx_recv_off[c->rank]);
    }
    o->active_vertices = c->n_active;
    o->exchange_slice = c->world > 1 ? c->xs : 0;
    o->world = (uint32_t)c->world;
    o->rank = (uint32_t)c->rank;
    o->placement_tried = c->place_tried;
    o->placement_kept = c->place_kept;
    for (u32 t = 0; t < 8; ++t) o->placement_us[t] = t < c->place_tried ? (uint32_t)(c->place_ms[t] * 1e3f) : 0u;
    return LZX_OK;
}

// lzx_internal.h -- shared state of liblzx.so (not part of the public boundary; see include/lzx.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "lzx.h"

typedef uint32_t u32;
typedef uint64_t u64;

// ---- error plumbing -------------------------------------------------------------------------------
void lzx_set_error(const char *fmt, ...);

#define LZX_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            lzx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return (e_ == hipErrorOutOfMemory) ? LZX_ERR_NOMEM : LZX_ERR_HIP;                   \
        }                                                                                      \
    } while (0)

#define LZX_TRY(call)              \
    do {                           \
        int rc_ = (call);          \
        if (rc_ != LZX_OK) return rc_; \
    } while (0)

#define LZX_FAIL(code, ...)       \
    do {                          \
        lzx_set_error(__VA_ARGS__); \
        return (code);            \
    } while (0)

// ---- layout constants -----------------------------------------------------------------------------
// Sliced-ELL body: slices of 64 rows (one wavefront), column indices stored so that lane l of the
// wave reads 16 contiguous bytes (4 indices of ITS row) per step and the wave reads 1 KiB contiguous.
static constexpr u32 LZX_SLICE = 64;
// Rows with more entries than this leave the sliced-ELL body and are split into wave-sized items.
static constexpr u32 LZX_LONG_ROW = 128;
// Entries one wavefront sums per split-row item (multiple of 256 = 64 lanes x 4 indices).
static constexpr u32 LZX_ITEM = 2048;
// Threads per workgroup of the SpMV kernel (16 wavefronts share one LDS copy of the hub entries).
static constexpr u32 LZX_SPMV_BLOCK = 1024;
static constexpr u32 LZX_VEC_BLOCK = 256;
// Zero tail behind every full-length vector: the padding column index points here.
static constexpr u32 LZX_TAIL = 64;
// Propagation blocking (csrc/lzx_pb.hip): column band staged in LDS by the scatter phase (doubles),
// rows per wavefront-private LDS y tile in the gather phase.
static constexpr u32 LZX_PB_CB = 16384;
static constexpr u32 LZX_PB_CB_WIDE = 18432;   // test shape pb_column_band: 144 KiB of x per tile (nine 2 Ki-value staging rounds of the 1 024 threads)
static constexpr u32 LZX_PB_RB = 1024;
static constexpr u32 LZX_PB_TARGET = 32768;   // upper limit of the values per gather item (one wavefront each)
static constexpr u32 LZX_PB_ALIGN = 4;        // (row band, column band) runs are padded to this many values: one plain quad's two 16-byte stores (round 5; 8 = a 64-byte line until then: 4 % more values on the 10 M-vertex R-MAT graph, 5 % on the uniform one, profiles/r5_align_sweep.txt)
static constexpr u32 LZX_PB_GATHER_BLOCK = 512;
static constexpr u32 LZX_PB_GROUP = 16384;    // a row band of at most this many values is gathered by one wavefront (k_pb_gather)
static constexpr unsigned long long LZX_PB_NT_BYTES = 128ull << 20;   // a gather value stream (values + slots) above this is read with non-temporal loads
static constexpr u32 LZX_PB_DYN_SHARE = 0;    // per cent of the gather pass's estimated cost whose items are drawn at run time (k_pb_gather's dynamic tail): off --
                                               // it evens the workgroups' end times out (170-194 us instead of 122-192 on the 10 M-vertex graph) and the pass ends when it did
                                               // before: its bound is the aggregate streaming rate, early finishers only leave their bandwidth to the others (DESIGN 3.1 j)
static constexpr u32 LZX_PB_ITEM_GROUP = 0xfffffffeu, LZX_PB_ITEM_NONE = 0xfffffffdu;   // item.w markers (0xffffffff: adds into v)
// entries, padded entries and values of one rank's blocked tables are indexed with 32 bits (a margin is left for the
// kernels' look-ahead)
static constexpr u64 LZX_PB_SLOT_LIMIT = (1ull << 32) - (1ull << 24);
// reduced runs (partial row sums instead of single values cross the two passes)
static constexpr u32 LZX_PBR_STEP = 512;      // entries per wavefront step; reduced runs are padded to whole steps
static constexpr u32 LZX_PBR_MIN_RUN = 384;   // a (row band, column band) run of at least this many entries is reduced
static constexpr bool LZX_PB_SCATTER_NT = true;   // the scatter pass's code / slot tables as non-temporal loads on large, mostly reduced value streams (test shape pb_scatter_nt)
static constexpr bool LZX_PB_CARRY_SCAN = true;   // reduced steps: rows that span lanes are summed by a cross-lane scan in registers (fixed order); false: LDS carry slots

// ---- peer-window transport (lzx_ipc.hip) ----
static constexpr int LZX_IPC_BUFS = 3;   // receive buffers a rank publishes per graph: d_xbuf, d_ybuf, d_xf32_full
struct lzx_ipc_state {
    void *win = nullptr;               // this rank's window (flags, mailboxes, board)
    bool finegrained = false;
    void *peer_win[64] = {};           // every rank's window as this process sees it ([rank] = win)
    bool win_opened[64] = {};          // ... mapped with hipIpcOpenMemHandle (to be closed)
    u64 seq[2] = {0, 0};               // last sequence number used on the main / the exchange stream's channel
    u64 mail_seq = 0, board_seq = 0;
    u64 deadline = 0;                  // ticks of the 100 MHz wall clock a wait may take
    bool broken = false;               // a deadline expired (here, or on a peer that then poisoned this window): every later collective fails at once
    struct Buf { void *mine = nullptr; size_t bytes = 0; void *peer[64] = {}; size_t peer_bytes[64] = {}; bool opened[64] = {}; } buf[LZX_IPC_BUFS];
    std::vector<u64> sx_dst_off;       // [world] where this rank's packed piece starts inside every peer's chunk 1
};

struct lzx_ctx {
    int device = 0;
    int cu_count = 256;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;                    // exchange stream (chunked all-gather overlapping the SpMV)
    hipEvent_t ev_q = nullptr, ev_c0 = nullptr, ev_c1 = nullptr;   // q_{j+1} slice ready / chunk 0 / chunk 1 arrived
    hipStream_t stream3 = nullptr;                    // side stream of the blocked SpMV (staged-columns kernel)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_phase2 = nullptr;                   // cross-handle ordering on stream2 (local communicator)
    hipEvent_t ev_a = nullptr, ev_b = nullptr;       // scratch timing pair
    hipEvent_t ev_phase = nullptr;                    // cross-handle ordering in local-comm mode
    std::vector<hipEvent_t> ev_pool;                  // per-iteration timing events
    hipEvent_t trace_ev[6] = {};                      // lzx_bench_spmv: marks between the SpMV's kernels
    bool trace = false;

    // ---- communicator ----
    int world = 1, rank = 0;
    int comm_kind = 0;                 // 0 none, 1 local (one process, peers[]), 2 RCCL, 3 peer windows (one process per rank, lzx_ipc.hip)
    struct lzx_ipc_state *ipc = nullptr;   // peer windows: this rank's window, the peers' mapped ones, sequence numbers
    bool publish_pending = false;      // peer windows: the hand-over is past its sync point and has not yet published its receive buffers
    lzx_ctx **peers = nullptr;         // local mode: all handles, index = rank (owned by rank 0's array copy)
    void *nccl_comm = nullptr;
    // in-process groups: the two-double reduction of the lazy loop through a peer-written mailbox -- every handle's reduce
    // kernel stores its [D, B] straight into slot [rank] of every peer's mailbox, the consumer (k_lazy_update) adds the
    // world pairs in rank order: one cross-handle barrier per iteration, no copies, no extra kernel.  Two mailboxes used
    // alternately (iteration parity): a fast handle's next write cannot overtake a slow handle's read.
    double *d_mail = nullptr;          // [2][64][2]
    bool mail_ok = false;              // every peer's mailbox can be written from this handle's GPU
    void *nccl_comm2 = nullptr;        // the exchange stream's own communicator (ncclCommSplit of nccl_comm), or null
    bool agree_pending = false;        // RCCL: a graph hand-over is under way and has not yet cast its vote at the sync point (lzx_agree_guard)

    // ---- whole graph, caller's vertex order (device) ----
    // (option sharded_ingest, several ranks: d_row_ptr still has n + 1 entries, but the rows of OTHER ranks are empty and
    //  d_col_idx holds this rank's rows only; nnz stays the whole matrix's count)
    u64 n = 0, nnz = 0, max_degree = 0;
    u64 *d_row_ptr = nullptr;
    u32 *d_col_idx = nullptr;
    // Sharded hand-over (lzx_graph.hip, "sharded hand-over"): where the directed entries come from while the hand-over runs
    // -- a seeded generator or an edge list resident on the device -- and every vertex's degree, counted in bounded sweeps
    // before lzx_graph_prepare starts.  Both live only for the duration of the entry point.
    struct lzx_key_source {
        int kind = -1;                 // -1: none (whole-graph hand-over); 0 Erdos-Renyi, 1 R-MAT (lzx_gen_graph); 2 edge list;
                                       // 3: the caller's CSR in HOST memory, streamed through the device in row chunks
        u32 scale = 0, ta = 0, tab = 0, tabc = 0;
        u64 n = 0, draws = 0, seed = 0;
        const u32 *d_src = nullptr, *d_dst = nullptr;   // kind 2: m endpoint pairs
        u64 m = 0;
        const u64 *h_row_ptr = nullptr;                 // kind 3: row_ptr[n + 1], col_idx[row_ptr[n]] (host, valid during the call)
        const u32 *h_col_idx = nullptr;
    } shard;
    u32 *d_shard_deg = nullptr;        // [n] degree of every vertex, caller's order
    bool sharded = false;              // the graph on this handle came through the sharded hand-over
    int64_t shard_opt = 0;             // option sharded_ingest: 0 off, 1 on (sweeps sized automatically), >= 2 on with that many sweeps

    // ---- this rank's share, internal order ----
    u32 n_loc_real = 0;                // rows owned
    u32 n_loc_pad = 0;                 // slice stride of the exchanged vector (multiple of 64, same on all ranks)
    u32 ldq = 0;                       // stride between basis vectors = n_loc_pad + LZX_TAIL
    u32 xs = 0;                        // slice stride of the PER-ITERATION exchange: only vertices with an edge are
                                       // exchanged (local rows are degree-sorted, so they are a prefix of every slice);
                                       // = n_loc_pad at one rank, round_up(ceil(n_active / world), 64) otherwise
    u32 xs0 = 0;                       // the exchange is cut in two chunks: the first xs0 entries of every slice (the
                                       // high-degree end: nearly all gathers land there) and the remaining xs - xs0;
                                       // layout [world][xs0] then [world][xs - xs0] so each chunk is one all-gather and
                                       // the second one can travel while the SpMV already works on the first
    // Sparse exchange of chunk 1 (overlapped mode): a rank receives from each peer only the entries its own rows
    // reference, packed; its x layout is [world][xs0] (chunk 0, dense) followed by one packed segment per peer.
    int64_t sparse_opt = -1;           // -1 auto (on with the two-chunk exchange), 0 off
    bool sparse = false;
    u64 xc1 = 0;                       // packed length of chunk 1 on this rank (all peers' segments)
    std::vector<u32> sx_recv_off;      // [world + 1] offsets of the peers' segments inside the packed chunk 1
    std::vector<u32> sx_send_off;      // [world + 1] offsets of what goes to each peer inside d_sx_sendbuf
    std::vector<u64> sx_send_hash, sx_recv_hash;   // [world] order-dependent hash of the local indices packed for / expected from every peer
    u32 *d_sx_send_idx = nullptr;      // [sx_send_off[world]] local row index (>= xs0) of every packed entry, peer-major
    double *d_sx_sendbuf = nullptr;    // [sx_send_off[world]]
    u32 *d_sx_map = nullptr;           // [xc1] packed chunk-1 position -> position in the hand-over layout (p * n_loc_pad + l)
    // N4 (SURVEY 8 f): the exchanged vector travels as fp32 (option exchange_fp32, off by default; single all-gather
    // path only): every rank multiplies the SAME fp32-rounded vector, sums stay fp64
    int64_t xfp32_opt = -1;
    bool xfp32 = false;
    float *d_xf32_send = nullptr;      // [xs] this rank's slice, rounded
    float *d_xf32_full = nullptr;      // [world * xs] the gathered slices
    u64 n_active = 0;                  // vertices of degree > 0
    u32 rows_live = 0;                 // this rank's rows that have an edge (a prefix of its rows), rounded up to whole slices
    u64 xlen = 0;                      // world * xs + LZX_TAIL: length of the vector the SpMV gathers from
    u64 iolen = 0;                     // world * n_loc_pad + LZX_TAIL: full-length layout of hand-over / results
    u64 nnz_local = 0;
    u32 *d_gidx_of_old = nullptr;      // [n] position of caller's vertex o in the full-length (iolen) layout
    u32 hub = 0;                       // entries staged in LDS
    int64_t hub_opt = -1;              // user override (-1: default)
    int64_t wgs_per_cu_opt = -1;
    int64_t nt_opt = -1;
    int64_t pb_opt = -1;               // propagation blocking: -1 auto, 0 off, 1 on
    int64_t overlap_opt = -1;          // chunked exchange overlapping the blocked SpMV: -1 auto, 0 off, 1 on
    bool overlap = false;
    u32 pb_units0 = 0;                 // scatter units whose column band lies wholly in chunk 0
    int64_t pb_target_opt = -1;        // entries per row band override
    int64_t pb_align_opt = -1;         // run padding override (4, 8, 16)
    int64_t marks_every_opt = -1;      // iterations between timing marks in the Lanczos loop (-1: 4, or 1 for short runs)
    bool force_multi = false;          // test hook: a 1-rank RCCL communicator runs the several-rank code path, collectives included
    int64_t lazy_opt = -1;             // lazy normalisation (lzx_api.hip): -1 = with several ranks and in blocked mode, 0 off, 1 on
    int64_t side_opt = -1;             // staged-columns kernel on a side stream next to the scatter passes: 1 on, else off
    int64_t pb_cb_opt = -1;            // column band override (8192, 16384 or, on one rank, 18432)
    u32 pb_cb = LZX_PB_CB;             // x values per column band = per LDS tile of the scatter pass
    int64_t pb_taper_opt = -1;         // tapered scatter unit sizes: -1/1 on, 0 off
    int64_t pb_unit_opt = -1;          // entries per scatter unit override
    int64_t pb_reduce_opt = -1;        // reduced bands: -1 auto (on), 0 off, > 1: minimum average run length
    int64_t long_row_opt = -1;         // split-row threshold override
    int64_t phase_mask_opt = 3;        // debug: 1 = split rows only, 2 = body only

    // sliced-ELL body
    u32 n_long64 = 0;                  // local rows [0, n_long64) go through the split-row path
    u32 n_long_true = 0;
    u32 n_slices = 0;
    u64 sell_elems = 0;
    u64 *d_slice_off = nullptr;        // [n_slices] element offset of slice s in d_sell_cols
    u32 *d_slice_w = nullptr;          // [n_slices] width (multiple of 4)
    // blocked mode: the staged-only slices are mostly 0 or 4 codes wide; they are processed class by class
    u32 *d_slice_perm = nullptr;       // [n_slices] slice ids: wide ones (> 8 codes) in order, then width 8, width 4, width 0
    u32 ns_wide = 0, ns_w8 = 0, ns_w4 = 0;
    std::vector<u32> h_slice_w0;       // ids of the width-0 slices, ascending (their live prefix is counted per launch)
    int64_t fuse_opt = -1;             // debug knob fuse_staged: 0 = staged-columns kernel in its own launch, 1 = fused behind the scatter units (default: ahead of them)
    int64_t narrow_opt = -1;           // debug knob narrow_slices: 0 = every slice through the general loop
    bool codes16 = false;              // staged-only tables (propagation-blocking mode): 16-bit codes
    u32 *d_sell_cols = nullptr;

    // split rows
    u32 n_items = 0;
    u64 long_elems = 0;
    u32 *d_long_cols = nullptr;
    u64 *d_item_beg = nullptr;         // [n_items] element offset into d_long_cols
    u32 *d_item_len = nullptr;         // [n_items] entries (multiple of 4)
    u32 *d_item_first = nullptr;       // [n_long64 + 1] first item of each split row
    double *d_long_partial = nullptr;  // [n_items]

    // propagation-blocked part (every entry whose column is not staged in LDS), see lzx_pb.hip
    bool pb = false;
    u32 hub_real = 0;                  // hub entries that carry x values (PB mode adds zero slots behind them)
    u64 pb_entries = 0;
    u32 pb_nr = 0;                     // row bands of LZX_PB_RB local rows
    u32 pb_units = 0;                  // scatter work units
    uint16_t *d_pb_lcol = nullptr;     // [pb_entries] scatter order: column within its band
    u32 *d_pb_dst = nullptr;           // [pb_entries] scatter order: slot in d_pb_val (gather order)
    uint16_t *d_pb_lrow = nullptr;     // [pb_entries] gather order: slot in the band's y tile (row * rep + replica)
    double *d_pb_val = nullptr;        // [pb_entries] gathered x values in gather order (scratch)
    u32 *d_pb_unit = nullptr;          // [pb_units][5] band, first / last step (reduced part), first / last quad (plain part)
    u32 *d_pb_row0 = nullptr;          // [pb_nr + 1] first local row of each row band
    u32 *d_pb_rep = nullptr;           // [pb_nr] LDS slots per row in that band's y tile
    u32 *d_pb_beg = nullptr;           // [pb_nr + 1] first value of each row band in gather order
    u32 *d_pb_items = nullptr;         // [pb_n_items][4] row band, begin, end (gather order), slot or ~0
    u32 *d_pb_multi = nullptr;         // [pb_n_multi][4] row, first slot, items, slot stride: rows of bands cut into several items
    double *d_pb_part = nullptr;       // item totals of those rows
    uint8_t *d_pb_long_multi = nullptr; // [n_long64] who closes the split row: 0 the gather fold, 1 its multi thread (d_pb_multi), 2 k_pb_finish's split-row thread
    int64_t pb_stamps_opt = -1;
    unsigned long long *d_pb_gstamps = nullptr;  // [pb_gather_grid][8] debug library, option pb_stamps: the product gather pass's sections
    int64_t tie_sort_opt = -1;         // blocked mode: ties of the degree ranking broken by staged-column count (debug knob; 0 = by id)
    int64_t item_opt = -1;             // entries per split-row item (debug knob; default LZX_ITEM)
    int64_t vec_per_cu_opt = -1;       // blocks per CU of the vector kernels (debug knob; default 8)
    int64_t burst_opt = -1;            // staged-columns kernel: staging loads all in flight (debug knob; 0 = one per iteration)
    int64_t deep_opt = -1;             // staged-columns kernel: 1 = four slices in flight instead of two (debug knob; no gain)
    int64_t pb_order_opt = -1;         // kernel order of the blocked SpMV (debug knob): -1/1 scatter, staged columns, gather; 0 staged columns first
    int64_t pb_group_opt = -1;         // values up to which a row band is gathered by ONE wavefront, eight such bands per workgroup item (debug knob; 0 = off)
    int64_t pb_group_force_opt = -1;   // test hook: this many bands per group (2 .. 8; 1 = 8) whatever the graph's size
    int64_t pb_gwaves_opt = -1;        // wavefronts per gather workgroup (debug knob): 8 (default) or 4
    u32 pb_gather_block = 512;
    u32 pb_n_items = 0, pb_n_multi = 0;
    u32 pb_gather_grid = 0, pb_finish_grid = 0;
    // Round 5: in the lazy loop the sums k_pb_finish adds to v (rows of multi-item bands: their per-item totals + their split-row
    // totals) are added by k_lazy_update where it reads w = v, and the launch is skipped.  pb_defer_ok: the graph's tables allow it
    // (there is a finish launch, it serves multi-item bands only, no dynamic tail); pb_deferring: the loop that does it is running
    // (set / reset by lanczos_loop: every other consumer of v -- lzx_spmv_f64, the plain loop, the re-orthogonalised one -- gets
    // the launch).  d_pb_mrow[row] for row < pb_multi_limit: {row, first slot, items, slot stride}, items = 0: not a multi row.
    uint4 *d_pb_mrow = nullptr;
    u32 pb_multi_limit = 0;
    bool pb_defer_ok = false, pb_deferring = false;
    int64_t defer_opt = -1;            // test shape defer_finish: 0 = never defer
    double *d_pb_item_dot = nullptr;   // [pb_n_dyn] alpha partial of every drawn item of the gather pass's dynamic tail (closed in ticket order by k_pb_finish)
    u32 pb_n_static = 0, pb_n_dyn = 0; // gather items dealt to workgroups by the host / drawn from d_pb_gcounter at run time (k_pb_gather)
    u32 *d_pb_gcounter = nullptr;      // the dynamic tail's ticket counter (back to 0 at the end of every launch)
    int64_t spmv_wgs_opt = -1;         // test shape spmv_wgs: at most this many workgroups of k_spmv / staged-columns workgroups of the shared launch
    int64_t pb_gather_nt_opt = -1;     // test shape pb_gather_nt: the gather pass's stream loads non-temporal (1) or cached (0); -1: by the stream's size
    int64_t pb_grid_cap_opt = -1;      // test shape pb_gather_grid: at most this many gather workgroups
    int64_t pb_scatter_nt_opt = -1;    // test shape pb_scatter_nt: the scatter pass's table loads non-temporal (1) / cached (0); -1: LZX_PB_SCATTER_NT by size
    int64_t pb_scan_opt = -1;          // test shape pb_carry_scan: 1 = cross-lane scan in registers, 0 = LDS carry slots, -1 = LZX_PB_CARRY_SCAN
    int64_t pb_dyn_opt = -1;           // test shape pb_dyn_share: per cent of the gather pass's cost left to the dynamic tail (-1: default)
    u64 pb_values = 0;                 // values the scatter passes hand to the gather pass per SpMV (incl. padding)
    u64 pbr_entries = 0;               // entries of the reduced runs
    uint4 *d_pbr_code = nullptr;       // [pbr_steps][64] scatter order: 8 x (column in band | piece-end flag) per lane
    u32 *d_pbr_base = nullptr;         // [pbr_steps] first value slot of the step
    u32 pbr_steps = 0;

    // vectors and scalars
    double *d_v = nullptr;             // [ldq]
    u32 np2_last = 0;                  // partials the last k_lazy_update left in d_partials2
    // Rows without an edge (the tail of every slice, [rows_live, n_loc_pad)) in the lazy loop: (A u)_i = 0 there, so
    // q_j[i] = c_j q_0[i] with ONE scalar recurrence for all of them (k_lazy_update, block 0); the loop neither reads nor
    // writes them, lzx_multout uses the scalars, a host fetch of the basis materialises them first.
    // Lazy loop: the resident basis holds the UNNORMALISED u_j = beta_{j-1} q_j (column 0: q_0): the loop then reads w, u_j,
    // u_{j-1} and writes u_{j+1} -- 32 bytes per row instead of 40 -- and q_j = u_j / beta_{j-1} is formed where it is used
    // (the same division as before: same bits); a fetch divides on the way out, lzx_multout folds 1 / beta into t.
    bool basis_u = false;
    int64_t basis_u_opt = -1;          // debug knob unnormalised_basis: 0 = columns hold q_j, u_j alternates between d_u[0 / 1]
    bool iso_on = false;               // the last prepared decomposition runs in that form
    bool iso_filled = false;           // ... and some of its basis columns have been materialised for those rows (a host fetch)
    u32 iso_cols_filled = 1;           // columns [1, iso_cols_filled) are; column 0 (q_0) always is
    double *d_iso = nullptr;           // [2 k_cap + 4]: c_j at [j], d_j (the same for the unnormalised u_j) at [k_cap + 1 + j]; last: sum of q_0[i]^2 over those rows
    u32 iso_cap = 0;
    int64_t iso_opt = -1;              // debug knob isolated_rows: 0 = elementwise like every other row
    double *d_u[2] = {nullptr, nullptr};   // [ldq] each, several ranks: the unnormalised Lanczos vector, alternating (lzx_api.hip)
    double *d_Q = nullptr;             // [q_cols][ldq]
    u32 q_cols = 0;
    u32 k_last = 0;                    // valid basis vectors from the last decomposition
    u32 k_prep = 0;                    // iterations a prepared decomposition will take in all (0: nothing prepared)
    u32 k_done = 0;                    // ... of which this many have run (lzx_lanczos_run_steps continues from here)
    // R1 (serial/lib/lanczos.cc:58-132, decompose_with_arnoldi): every reorth_opt iterations (j % e == 0 && j > 2) A q_j is
    // orthogonalised against q_0 .. q_{j-2} (modified Gram-Schmidt, the reference's order) before alpha_j is taken; the
    // reference hard-codes e = 2.  0 / -1: off.  Runs the reference-order loop (normalised basis, every row elementwise).
    int64_t reorth_opt = -1;
    // Option reference_order (one rank): SpMV one lane per row of the caller's CSR, inner product and
    // norm one left-to-right accumulator in the caller's vertex order -- the reduction orders serial/ fixes (SPMV.cc:24-27,
    // lanczos.cc:155-171), so alpha / beta / Q meet the oracle's bit for bit at any k.  A parity instrument, not a fast path.
    int64_t ref_order_opt = -1;
    // Option placement_trials (blocked mode): at the end of a graph hand-over the value stream between the two passes is
    // allocated this many more times, the SpMV timed with each, the fastest kept -- where the driver puts that buffer
    // decides 15 % of the Erdos-Renyi SpMV and 1-2 % of the R-MAT one (lzx_pb.hip: lzx_pb_place_values).  -1: default (2), 0: off, at most 7.
    int64_t place_opt = -1;
    // test shape start_vector_scan (lzx_api.hip: scan_start_vector): 0 = x0 always crosses PCIe, its norm always the serial chain
    int64_t x0_scan_opt = -1;
    bool x0_was_constant = false;      // the last prepared start vector was filled on the device, not uploaded
    u32 place_tried = 0;               // allocations timed at the last hand-over (incl. the first one), and their SpMV times
    float place_ms[8] = {};
    u32 place_kept = 0;
    // N4 remainder: the resident basis is STORED as fp32 (d_Qf), the recurrence keeps its three live vectors in fp64
    // (d_ring); lzx_multout and the host fetch read the fp32 columns.  Lazy loop only.  Off by default.
    int64_t qf32_opt = -1;
    bool qf32 = false;                 // the last prepared decomposition stores its basis that way
    float *d_Qf = nullptr;             // [qf_cols][ldq]
    u32 qf_cols = 0;
    double *d_ring[3] = {nullptr, nullptr, nullptr};   // [ldq] each: u_{j-1}, u_j, u_{j+1} (column j lives in d_ring[j % 3])
    // convergence monitor on the device (lzx_multout_change_f64): the last two evaluated answers, this rank's slice
    double *d_ymon[2] = {nullptr, nullptr};            // [n_loc_pad] each
    u32 ymon_valid = 0;                // answers evaluated since the last prepare (0: no previous one to compare with)
    double *d_xbuf = nullptr;          // [xlen] exchange buffer the SpMV gathers from (world > 1, hooks)
    double *d_ybuf = nullptr;          // [iolen] full-length buffer in hand-over layout (hooks, multout, fetch)
    double *d_io = nullptr;            // [n] staging in the caller's order
    double *d_partials = nullptr;      // [np_cap] block partials of the running reduction
    double *d_partials2 = nullptr;     // [np_cap]
    double *d_partials3 = nullptr;     // [np_cap] one rank, lazy normalisation: norm partials alternate with d_partials2
    u32 np_cap = 0;
    double *d_alpha = nullptr, *d_beta = nullptr;  // [k_cap]
    u32 k_cap = 0;
    double *d_scal = nullptr;          // [8 + world] reduced scalars for the exchange path

    // launch shape of the SpMV kernel
    u32 spmv_grid = 0;
    u32 fin_grid = 0;
    size_t spmv_lds = 0;
};

// ---- lzx_graph.hip ----
int lzx_graph_release(lzx_ctx *c);
int lzx_graph_prepare(lzx_ctx *c);   // builds this rank's share from d_row_ptr/d_col_idx (sharded hand-over: from c->shard)

// ---- lzx_pb.hip ----
// Builds the propagation-blocked structure for this rank's non-hub entries. d_nh_off: exclusive prefix of the
// per-local-row count of non-hub entries (u32, n_loc_real + 1 values).
int lzx_pb_prepare(lzx_ctx *c, const u32 *d_code, const u32 *d_old_of_local, const u32 *d_deg_local,
                   const u32 *d_nh_off, const std::vector<u32> &h_nh, u64 total);
void lzx_pb_release(lzx_ctx *c);
int lzx_pb_place_values(lzx_ctx *c);   // end of a hand-over: option placement_trials
// chunk1_ready (may be null): event after which the second chunk of the exchange layout is valid in x
// phases: 1 = scatter pass, 2 = gather (+ finish) pass, 3 = both
struct SpmvArgs;   // lzx_spmv_body.h
bool lzx_pb_can_fuse(const lzx_ctx *c);
int lzx_pb_launch(lzx_ctx *c, const double *x, const double *q_loc, double *v, double *partials, hipEvent_t chunk1_ready,
                  hipEvent_t v_ready, int phases = 3, const struct SpmvArgs *fuse = nullptr, u32 fuse_blocks = 0, bool *fused = nullptr);
u32 lzx_pb_partials(const lzx_ctx *c);

// ---- lzx_kernels.hip ----
struct SpmvLaunch {
    const double *x;       // full-length input vector (xlen layout)
    const double *q_loc;   // this rank's slice of it (for the fused alpha partial)
    double *v;             // [n_loc_pad] output
    double *partials;      // block partials of v.q_loc ; count returned by lzx_spmv_partials()
    hipEvent_t chunk1_ready = nullptr;   // overlapped exchange: columns of chunk 1 may be read only after this event
    bool live_rows_only = false;         // the caller never reads v beyond the rows that have an edge (lzx_ctx::rows_live)
};
int lzx_launch_spmv(lzx_ctx *c, const SpmvLaunch &a);
u32 lzx_spmv_partials(const lzx_ctx *c);
int lzx_launch_reduce(lzx_ctx *c, const double *partials, u32 np, double *out, int do_sqrt);
int lzx_launch_reduce2(lzx_ctx *c, const double *pa, u32 na, const double *pb, u32 nb, double *out2);
struct MailPeers { double *slot[64]; };   // where this handle's pair goes in every peer's mailbox (parity applied)
int lzx_launch_reduce2_mail(lzx_ctx *c, const double *pa, u32 na, const double *pb, u32 nb, const MailPeers &peers, u32 world);
int lzx_launch_iso_prepare(lzx_ctx *c, u32 k);                       // sum of squares of q_0 over the rows without an edge, c_0 = d_0 = 1
int lzx_launch_iso_fill(lzx_ctx *c, u32 k);                          // q_j[i] = c_j q_0[i] for those rows, the columns below k that are not there yet
// what k_lazy_update adds to w for the rows of multi-item bands when k_pb_finish was not launched (limit = 0: nothing)
struct LazyDeferred {
    const uint4 *mrow = nullptr;
    u32 limit = 0;
    const double *part = nullptr;
    const u32 *item_first = nullptr;
    const double *long_partial = nullptr;
    const uint8_t *long_mode = nullptr;
    u32 n_long = 0;
};
int lzx_launch_lazy_update(lzx_ctx *c, const double *w, u32 w_rows, const double *u, const double *q_prev, const double *scal2, int first,
                           double *alpha_out, double *beta_out, double *q_out, double *u_next, double *partials_out, u32 *np_out, const double *prev_div = nullptr,
                           float *f32_next = nullptr, u32 mail_world = 0);   // mail_world > 0: scal2 is a mailbox of that many [D, B] pairs
int lzx_launch_lazy_update_local(lzx_ctx *c, const double *w, u32 w_rows, const double *u, const double *q_prev, const double *pa, u32 na,
                                 const double *pb, u32 nb, int first, double *alpha_out, double *beta_out, double *q_out,
                                 double *u_next, double *partials_out, u32 *np_out, const double *prev_div = nullptr, float *f32_next = nullptr);
// v -= alpha q_j (+ beta_prev q_jm1); alpha = sum(partials_in); writes alpha_out; partial ||v||^2 out.
int lzx_launch_axpy_norm(lzx_ctx *c, double *v, const double *qj, const double *qjm1,
                         const double *partials_in, u32 np_in, double *alpha_out,
                         const double *beta_prev, double *partials_out, u32 *np_out);
// q_next = v / sqrt(sum(partials_in)); writes beta_out.  in_is_sqrt: partials_in[0] already holds beta^2 summed.
int lzx_launch_scale(lzx_ctx *c, const double *v, double *q_next, const double *partials_in,
                     u32 np_in, double *beta_out);
int lzx_launch_permute_in(lzx_ctx *c, const double *io_old_order, double *full, double scale);
int lzx_launch_fill(lzx_ctx *c, double *out, double value, u64 count);
int lzx_launch_permute_out(lzx_ctx *c, const double *full, double *io_old_order, const double *div = nullptr);   // div: device scalar every entry is divided by
// hand-over layout (stride n_loc_pad) -> exchange layout (stride xs): the active prefix of every rank's slice
int lzx_launch_relayout(lzx_ctx *c, const double *io_layout, double *exchange_layout);
int lzx_launch_multout(lzx_ctx *c, const double *t_dev, u32 k, double *out_loc);
// one step of the Arnoldi pass (serial/lib/lanczos.cc:86-90): v -= d q_m (d = sum of d_partials[0..d_np), or *d_scal when
// d_np == 0; q_m == nullptr: no update), then per-block partials of v . q_next
int lzx_launch_mgs_step(lzx_ctx *c, double *v, const double *q_m, const double *d_partials, u32 d_np, const double *d_scal,
                        const double *q_next, double *partials_out, u32 *np_out);
// partial sums of |y - y_prev|^2 and |y|^2 over this rank's rows -> out2[0..1] (device)
int lzx_launch_change(lzx_ctx *c, const double *y, const double *y_prev, double *out2);
// reference-order test shape: y = A x one lane per caller's row (y, x in the internal layout); out = sum over the caller's
// vertex order of a[i] * b[i], one accumulator (uses d_io as scratch)
int lzx_launch_ref_spmv(lzx_ctx *c, const double *x, double *y);
int lzx_launch_ref_dot(lzx_ctx *c, const double *a, const double *b, double *out);
// column `col` of the fp32 basis, widened to fp64 into out[0..n_loc_pad)
int lzx_launch_widen_col(lzx_ctx *c, u32 col, double *out);

// ---- lzx_comm.hip ----
// does the Lanczos loop exchange vectors / reduce scalars through the communicator?
// splitmix64 finaliser: the hash of the sparse exchange's index lists (lzx_graph.hip, lzx_comm_check_sparse)
inline u64 lzx_mix64(u64 z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// One rank's message of the sparse-exchange check, 6 * world words: [send counts | receive counts | send hashes (lo, hi) | receive
// hashes (lo, hi)]; lzx_sx_check_pairs: every pair of ranks agrees on how many entries travel AND on which ones, or an error that
// is the same on every rank (all of them hold all messages).
void lzx_sx_check_message(const lzx_ctx *c, std::vector<u32> &mine);
int lzx_sx_check_pairs(const std::vector<u32> &all, u32 world);
inline bool lzx_exchanges(const lzx_ctx *c) { return c->world > 1 || (c->force_multi && c->comm_kind >= 2); }
int lzx_comm_allreduce_sum(std::vector<lzx_ctx *> &cs, u32 slot, u32 count = 1);   // d_scal[slot .. slot+count) on every handle
// peers_idle: the caller knows that no rank still reads what the gather overwrites (the loop: an all-reduce lies between
// every SpMV and the next exchange); otherwise the peer-window transport puts a barrier in front (the others order by
// construction)
int lzx_comm_allgather(std::vector<lzx_ctx *> &cs, const double *const *src_loc, double *const *dst_full, size_t count,
                       bool on_stream2 = false, bool peers_idle = false);
// chunk 1 of the exchange, sparse form: every handle packs what each peer's rows reference out of slice_loc[i] (its
// own slice of the new vector) and the packed pieces land in the peers' d_xbuf behind chunk 0; on the exchange streams
int lzx_comm_sparse_chunk1(std::vector<lzx_ctx *> &cs, const double *const *slice_loc, bool peers_idle = false);
int lzx_launch_sx_pack(lzx_ctx *c, const double *slice_loc, hipStream_t st);
int lzx_comm_check_sparse(lzx_ctx *c);   // RCCL, peer windows: all ranks' send / receive lists of the sparse chunk agree pairwise (lengths and content hashes), or LZX_ERR_STATE everywhere
int lzx_comm_check_sparse_local(std::vector<lzx_ctx *> &cs);   // the same for the handles of an in-process group (host memory, no collective)
// RCCL: every rank learns whether a rank-LOCAL step (an allocation, a sort, ...) failed on ANY rank -- a 1-value all-reduce
// (min) on the handle's pre-allocated scalars -- before any of them enters the next collective; other transports: *all_ok = ok.
int lzx_comm_agree(lzx_ctx *c, bool ok, bool *all_ok);
// A graph hand-over over RCCL contains ONE such sync point (inside lzx_graph_prepare, ahead of the pairwise check of the
// sparse lists).  A rank whose local steps fail BEFORE it must still cast its vote, or its peers wait there for ever: every
// hand-over entry point holds a guard, whose destructor votes "failed" if the sync point was never reached.
int lzx_comm_ipc_publish(lzx_ctx *c, bool ok);   // lzx_ipc.hip, see below
struct lzx_agree_guard {
    lzx_ctx *c;
    explicit lzx_agree_guard(lzx_ctx *c_) : c(c_) { if (c) c->agree_pending = c->comm_kind >= 2 && lzx_exchanges(c); }
    ~lzx_agree_guard()
    {
        if (c && c->agree_pending) {
            bool all = false;
            c->agree_pending = false;
            (void)lzx_comm_agree(c, false, &all);
        }
        if (c && c->publish_pending) {   // peer windows: failed between the sync point and the publication of its receive buffers
            c->publish_pending = false;
            (void)lzx_comm_ipc_publish(c, false);
        }
    }
    lzx_agree_guard(const lzx_agree_guard &) = delete;
    lzx_agree_guard &operator=(const lzx_agree_guard &) = delete;
};
// N4: all-gather of the slices as fp32 into every handle's d_xbuf (converted back to fp64 there), main streams
int lzx_comm_allgather_fp32(std::vector<lzx_ctx *> &cs, const double *const *slice_loc, bool peers_idle = false);
int lzx_launch_to_f32(lzx_ctx *c, const double *in, float *out, u64 count);
int lzx_launch_to_f64(lzx_ctx *c, const float *in, double *out, u64 count);
// everything queued so far on every handle's `from` stream happens before what is queued next on every `to` stream
int lzx_comm_order(std::vector<lzx_ctx *> &cs, bool from_stream2, bool to_stream2);
// in-process group, lazy loop: every handle's [D, B] into every peer's mailbox (parity = iteration & 1), then one barrier;
// returns false (nothing queued) when the group cannot use mailboxes (no peer access): the caller takes the all-reduce
bool lzx_comm_mail_usable(std::vector<lzx_ctx *> &cs);
int lzx_comm_mail_reduce2(std::vector<lzx_ctx *> &cs, u32 parity, bool first);
// peer windows, lazy loop: [sum pa, sum pb] of this rank closed from the block partials AND all-reduced into d_scal[0..1] by one launch
bool lzx_comm_fused_reduce2(const lzx_ctx *c);
int lzx_comm_reduce2_allreduce(lzx_ctx *c, const double *pa, u32 na, const double *pb, u32 nb);
void lzx_comm_release(lzx_ctx *c);
// ---- lzx_ipc.hip: the peer-window transport behind the operations above ----
void lzx_comm_ipc_release(lzx_ctx *c);
int lzx_comm_ipc_check(lzx_ctx *c);                          // LZX_ERR_COMM if one of this rank's waits ran into its deadline
int lzx_comm_ipc_agree(lzx_ctx *c, bool ok, bool *all_ok);
int lzx_comm_ipc_publish(lzx_ctx *c, bool ok);               // collective, end of a graph hand-over: the receive buffers become reachable for the peers
void lzx_comm_ipc_unpublish(lzx_ctx *c);                     // before those buffers are freed
int lzx_comm_ipc_allreduce(lzx_ctx *c, u32 slot, u32 count, int op, const double *pa = nullptr, u32 na = 0, const double *pb = nullptr, u32 nb = 0);
int lzx_comm_ipc_allgather(lzx_ctx *c, const double *src_loc, double *dst_full, size_t cnt, bool s2, bool peers_idle);
int lzx_comm_ipc_allgather_f32(lzx_ctx *c, const float *src_loc, float *dst_full, size_t cnt, bool peers_idle);
int lzx_comm_ipc_sparse_chunk1(lzx_ctx *c, bool peers_idle);
int lzx_comm_ipc_check_sparse(lzx_ctx *c);

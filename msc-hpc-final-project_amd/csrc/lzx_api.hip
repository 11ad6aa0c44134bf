// lzx_api.hip -- C ABI entry points (include/lzx.h) and the Lanczos loop driver.
//
// The loop is written once over a list of handles: one handle (single GPU, or one rank of an RCCL
// communicator) or all handles of an in-process communicator.  Everything between the upload of x0
// and the download of alpha/beta is enqueued on HIP streams without a host synchronisation; the
// scalars alpha_j / beta_j never leave the device (as in the reference, where later kernels read
// *alpha_d: parallel-final/lib/cu_lanczos.cu:108,113,123).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#include "lzx_internal.h"

static thread_local std::string g_err;

void lzx_set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

extern "C" const char *lzx_last_error(void) { return g_err.c_str(); }

extern "C" int lzx_create(lzx_handle *out, int device_id)
{
    if (!out) LZX_FAIL(LZX_ERR_ARG, "lzx_create: null out");
    *out = nullptr;
    int count = 0;
    LZX_HIP(hipGetDeviceCount(&count));
    if (device_id < 0 || device_id >= count)
        LZX_FAIL(LZX_ERR_ARG, "lzx_create: device %d not present (%d visible)", device_id, count);
    LZX_HIP(hipSetDevice(device_id));
    lzx_ctx *c = new (std::nothrow) lzx_ctx;
    if (!c) LZX_FAIL(LZX_ERR_NOMEM, "lzx_create: host allocation failed");
    c->device = device_id;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device_id);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_a);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_b);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_phase, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_phase2, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_q, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_c0, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_c1, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->d_scal), sizeof(double) * (8 + 64));
    if (e != hipSuccess) {
        lzx_set_error("lzx_create: %s", hipGetErrorString(e));
        lzx_destroy(c);
        return LZX_ERR_HIP;
    }
    c->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    *out = c;
    return LZX_OK;
}

extern "C" void lzx_destroy(lzx_handle c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    lzx_graph_release(c);
    lzx_comm_release(c);
    if (c->d_alpha) (void)hipFree(c->d_alpha);
    if (c->d_beta) (void)hipFree(c->d_beta);
    if (c->d_scal) (void)hipFree(c->d_scal);
    if (c->d_iso) (void)hipFree(c->d_iso);
    if (c->d_Qf) (void)hipFree(c->d_Qf);
    for (double *&r : c->d_ring) { if (r) (void)hipFree(r); r = nullptr; }
    for (double *&r : c->d_ymon) { if (r) (void)hipFree(r); r = nullptr; }
    for (hipEvent_t ev : c->ev_pool) (void)hipEventDestroy(ev);
    if (c->ev_a) (void)hipEventDestroy(c->ev_a);
    if (c->ev_b) (void)hipEventDestroy(c->ev_b);
    if (c->ev_phase) (void)hipEventDestroy(c->ev_phase);
    if (c->ev_phase2) (void)hipEventDestroy(c->ev_phase2);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_q) (void)hipEventDestroy(c->ev_q);
    if (c->ev_c0) (void)hipEventDestroy(c->ev_c0);
    if (c->ev_c1) (void)hipEventDestroy(c->ev_c1);
    if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
    if (c->stream3) { (void)hipStreamSynchronize(c->stream3); (void)hipStreamDestroy(c->stream3); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int lzx_set_option(lzx_handle c, const char *name, int64_t value)
{
    if (!c || !name) LZX_FAIL(LZX_ERR_ARG, "lzx_set_option: bad argument");
    // the two options that shape the LOOP, not the graph, may change between decompositions
    const bool loop_option = !strcmp(name, "reorthogonalise") || !strcmp(name, "basis_fp32") || !strcmp(name, "reference_order");
    if (c->d_row_ptr && !loop_option) LZX_FAIL(LZX_ERR_STATE, "options must be set before the graph is handed over");
    if (loop_option) c->k_prep = 0;   // a decomposition that was being advanced in chunks is abandoned (what it has done stays usable)
    if (!strcmp(name, "hub_entries")) c->hub_opt = value;
    else if (!strcmp(name, "propagation_blocking")) c->pb_opt = value;
    else if (!strcmp(name, "overlap_exchange")) c->overlap_opt = value;
    else if (!strcmp(name, "lazy_normalisation")) c->lazy_opt = value;
    else if (!strcmp(name, "sparse_exchange")) c->sparse_opt = value;
    else if (!strcmp(name, "exchange_fp32")) c->xfp32_opt = value;
    else if (!strcmp(name, "timing_marks_every")) c->marks_every_opt = value;
    else if (!strcmp(name, "reorthogonalise")) c->reorth_opt = value;
    else if (!strcmp(name, "basis_fp32")) c->qf32_opt = value;
    else if (!strcmp(name, "reference_order")) c->ref_order_opt = value;
    else if (!strcmp(name, "placement_trials")) c->place_opt = value;
    else if (!strcmp(name, "sharded_ingest")) c->shard_opt = value < 0 ? 0 : value;
#ifdef LZX_DEBUG_KNOBS
    // experiment knobs and test hooks: only in liblzx_dbg.so (make debug), which tests/ and tools/perf_probe.py load
    // when they ask for one of these; the product library does not know the names
    else if (!strcmp(name, "wgs_per_cu")) c->wgs_per_cu_opt = value;
    else if (!strcmp(name, "nt_index_loads")) c->nt_opt = value;
    else if (!strcmp(name, "long_row")) c->long_row_opt = value;
    else if (!strcmp(name, "pb_target")) c->pb_target_opt = value;
    else if (!strcmp(name, "pb_run_align")) c->pb_align_opt = value;
    else if (!strcmp(name, "pb_reduce")) c->pb_reduce_opt = value;
    else if (!strcmp(name, "pb_unit")) c->pb_unit_opt = value;
    else if (!strcmp(name, "pb_taper")) c->pb_taper_opt = value;
    else if (!strcmp(name, "pb_dyn_share")) c->pb_dyn_opt = value;
    else if (!strcmp(name, "pb_gather_grid")) c->pb_grid_cap_opt = value;
    else if (!strcmp(name, "pb_gather_nt")) c->pb_gather_nt_opt = value;
    else if (!strcmp(name, "spmv_wgs")) c->spmv_wgs_opt = value;
    else if (!strcmp(name, "pb_column_band")) c->pb_cb_opt = value;
    else if (!strcmp(name, "side_stream")) c->side_opt = value;
    else if (!strcmp(name, "pb_stamps")) c->pb_stamps_opt = value;
    else if (!strcmp(name, "pb_order")) c->pb_order_opt = value;
    else if (!strcmp(name, "spmv_deep")) c->deep_opt = value;
    else if (!strcmp(name, "tie_sort")) c->tie_sort_opt = value;
    else if (!strcmp(name, "pb_group")) c->pb_group_opt = value;
    else if (!strcmp(name, "pb_group_force")) c->pb_group_force_opt = value;
    else if (!strcmp(name, "item_len")) c->item_opt = value;
    else if (!strcmp(name, "stage_burst")) c->burst_opt = value;
    else if (!strcmp(name, "vec_blocks_per_cu")) c->vec_per_cu_opt = value;
    else if (!strcmp(name, "narrow_slices")) c->narrow_opt = value;
    else if (!strcmp(name, "fuse_staged")) c->fuse_opt = value;
    else if (!strcmp(name, "isolated_rows")) c->iso_opt = value;
    else if (!strcmp(name, "unnormalised_basis")) c->basis_u_opt = value;
    else if (!strcmp(name, "pb_gather_waves")) c->pb_gwaves_opt = value;
    else if (!strcmp(name, "exchange_at_world_1")) c->force_multi = value > 0;
    else if (!strcmp(name, "phase_mask")) c->phase_mask_opt = value;
    else if (!strcmp(name, "start_vector_scan")) c->x0_scan_opt = value;
    else if (!strcmp(name, "defer_finish")) c->defer_opt = value;
#endif
    else LZX_FAIL(LZX_ERR_ARG, "lzx_set_option: unknown option '%s'", name);
    return LZX_OK;
}

// Test-only entry of the PRODUCT library (declared in csrc/lzx_test_hooks.h, not in include/lzx.h): forces, on small test
// graphs, the table shapes large graphs get by themselves -- run formats, band / item / unit sizes, slice classes, gather
// groups, tie-break -- and the hook that runs the several-rank loop on a 1-rank RCCL communicator, so that the parity
// tests exercise the same machine code the bench runs.  Every name below only selects among
// code paths the product build contains; the experiment knobs stay in liblzx_dbg.so.
extern "C" int lzx_test_set_shape(lzx_handle c, const char *name, int64_t value)
{
    if (!c || !name) LZX_FAIL(LZX_ERR_ARG, "lzx_test_set_shape: bad argument");
    if (c->d_row_ptr) LZX_FAIL(LZX_ERR_STATE, "shapes must be set before the graph is handed over");
    if (!strcmp(name, "pb_reduce")) c->pb_reduce_opt = value;
    else if (!strcmp(name, "pb_target")) c->pb_target_opt = value;
    else if (!strcmp(name, "pb_unit")) c->pb_unit_opt = value;
    else if (!strcmp(name, "pb_column_band")) c->pb_cb_opt = value;
    else if (!strcmp(name, "pb_run_align")) c->pb_align_opt = value;
    else if (!strcmp(name, "pb_taper")) c->pb_taper_opt = value;
    else if (!strcmp(name, "pb_dyn_share")) c->pb_dyn_opt = value;
    else if (!strcmp(name, "pb_carry_scan")) c->pb_scan_opt = value;
    else if (!strcmp(name, "pb_scatter_nt")) c->pb_scatter_nt_opt = value;
    else if (!strcmp(name, "pb_gather_grid")) c->pb_grid_cap_opt = value;
    else if (!strcmp(name, "pb_gather_nt")) c->pb_gather_nt_opt = value;
    else if (!strcmp(name, "spmv_wgs")) c->spmv_wgs_opt = value;
    else if (!strcmp(name, "pb_group")) c->pb_group_opt = value;
    else if (!strcmp(name, "pb_group_force")) c->pb_group_force_opt = value;
    else if (!strcmp(name, "narrow_slices")) c->narrow_opt = value;
    else if (!strcmp(name, "tie_sort")) c->tie_sort_opt = value;
    else if (!strcmp(name, "long_row")) c->long_row_opt = value;
    else if (!strcmp(name, "item_len")) c->item_opt = value;
    else if (!strcmp(name, "exchange_at_world_1")) c->force_multi = value > 0;
    // the lazy loop's two other forms (rows without an edge elementwise instead of one scalar recurrence; q_j stored instead of
    // the unnormalised u_j) and the staged-columns workgroups' place in the shared launch: all compiled into this library
    else if (!strcmp(name, "isolated_rows")) c->iso_opt = value;
    else if (!strcmp(name, "unnormalised_basis")) c->basis_u_opt = value;
    else if (!strcmp(name, "fuse_staged")) c->fuse_opt = value;
    // 0: the start vector always crosses PCIe and its norm is always the serial chain (what round 5's look at x0 is compared with)
    else if (!strcmp(name, "start_vector_scan")) c->x0_scan_opt = value;
    // 0: the blocked SpMV always launches k_pb_finish (default: the lazy loop's vector kernel stands in for it)
    else if (!strcmp(name, "defer_finish")) c->defer_opt = value;
    else LZX_FAIL(LZX_ERR_ARG, "lzx_test_set_shape: unknown shape '%s'", name);
    return LZX_OK;
}

// what shape the blocked tables took (test hook, beside lzx_test_set_shape): "gather_items_dealt", "gather_items_drawn",
// "gather_workgroups"
extern "C" int lzx_test_get_shape(lzx_handle c, const char *name, int64_t *value)
{
    if (!c || !name || !value) LZX_FAIL(LZX_ERR_ARG, "lzx_test_get_shape: bad argument");
    if (!strcmp(name, "gather_items_dealt")) *value = c->pb ? c->pb_n_static : 0;
    else if (!strcmp(name, "gather_items_drawn")) *value = c->pb ? c->pb_n_dyn : 0;
    else if (!strcmp(name, "gather_workgroups")) *value = c->pb ? c->pb_gather_grid : 0;
    else if (!strcmp(name, "placement_tried")) *value = c->place_tried;
    else if (!strcmp(name, "placement_kept")) *value = c->place_kept;
    else if (!strcmp(name, "start_vector_was_constant")) *value = c->x0_was_constant ? 1 : 0;
    else if (!strcmp(name, "finish_deferrable")) *value = c->pb && c->pb_defer_ok ? 1 : 0;
    else if (!strcmp(name, "finish_launched")) *value = c->pb && c->pb_finish_grid ? 1 : 0;
    else if (!strncmp(name, "placement_us_", 13) && name[13] >= '0' && name[13] <= '7' && !name[14]) *value = (int64_t)(c->place_ms[name[13] - '0'] * 1e3f);
    else LZX_FAIL(LZX_ERR_ARG, "lzx_test_get_shape: unknown shape '%s'", name);
    return LZX_OK;
}

// test hook (csrc/lzx_test_hooks.h): what one two-double all-reduce of the wired communicator costs, back to back
extern "C" int lzx_test_allreduce_latency(lzx_handle c, uint32_t reps, double *us_each)
{
    if (!c || !us_each || reps == 0) LZX_FAIL(LZX_ERR_ARG, "lzx_test_allreduce_latency: bad argument");
    if (c->comm_kind < 2 || !lzx_exchanges(c)) LZX_FAIL(LZX_ERR_STATE, "lzx_test_allreduce_latency: needs a one-process-per-rank communicator");
    std::vector<lzx_ctx *> cs(1, c);
    LZX_HIP(hipSetDevice(c->device));
    LZX_HIP(hipMemsetAsync(c->d_scal, 0, sizeof(double) * 2, c->stream));
    for (u32 i = 0; i < 5; ++i) LZX_TRY(lzx_comm_allreduce_sum(cs, 0, 2));   // warm-up, and the ranks fall into step
    LZX_HIP(hipEventRecord(c->ev_a, c->stream));
    for (u32 i = 0; i < reps; ++i) LZX_TRY(lzx_comm_allreduce_sum(cs, 0, 2));
    LZX_HIP(hipEventRecord(c->ev_b, c->stream));
    LZX_HIP(hipStreamSynchronize(c->stream));
    LZX_TRY(lzx_comm_ipc_check(c));
    float ms = 0.f;
    LZX_HIP(hipEventElapsedTime(&ms, c->ev_a, c->ev_b));
    *us_each = (double)ms * 1e3 / reps;
    return LZX_OK;
}

// test hook (csrc/lzx_test_hooks.h): one rank's local SpMV of x = 1, and where every vertex sits in the full-length layout
extern "C" int lzx_test_rank_row_sums(lzx_handle c, double *v_local, uint32_t *layout_pos, uint64_t *n_loc_pad)
{
    if (!c || !v_local || !layout_pos || !n_loc_pad) LZX_FAIL(LZX_ERR_ARG, "lzx_test_rank_row_sums: bad argument");
    if (!c->d_row_ptr || !c->d_v) LZX_FAIL(LZX_ERR_STATE, "no graph has been handed over");
    double avg = 0.0;
    LZX_TRY(lzx_bench_spmv(c, 1, &avg, nullptr));   // x = 1 in the layout this rank gathers from; leaves v = A x for its rows
    LZX_HIP(hipSetDevice(c->device));
    LZX_HIP(hipMemcpy(v_local, c->d_v, sizeof(double) * c->n_loc_real, hipMemcpyDeviceToHost));
    LZX_HIP(hipMemcpy(layout_pos, c->d_gidx_of_old, sizeof(u32) * c->n, hipMemcpyDeviceToHost));
    *n_loc_pad = c->n_loc_pad;
    return LZX_OK;
}

// --------------------------------------------------------------------------------------------------
static int gather_handles(lzx_handle h, std::vector<lzx_ctx *> &cs)
{
    if (!h) LZX_FAIL(LZX_ERR_ARG, "null handle");
    if (h->comm_kind == 1 && h->world > 1)
        LZX_FAIL(LZX_ERR_STATE, "handle belongs to an in-process communicator: use the *_local entry point");
    cs.assign(1, h);
    return LZX_OK;
}

static int gather_handles_local(lzx_handle *hs, int world, std::vector<lzx_ctx *> &cs)
{
    if (!hs || world < 1) LZX_FAIL(LZX_ERR_ARG, "bad handle list");
    cs.clear();
    for (int p = 0; p < world; ++p) {
        if (!hs[p]) LZX_FAIL(LZX_ERR_ARG, "null handle in list");
        if (hs[p]->world != world || hs[p]->rank != p || (world > 1 && hs[p]->comm_kind != 1))
            LZX_FAIL(LZX_ERR_STATE, "handles are not wired as an in-process communicator of %d", world);
        cs.push_back(hs[p]);
    }
    return LZX_OK;
}

static int check_graphs(std::vector<lzx_ctx *> &cs)
{
    for (lzx_ctx *c : cs) {
        if (!c->d_row_ptr || !c->d_v) LZX_FAIL(LZX_ERR_STATE, "no graph has been handed over");
        if (c->n != cs[0]->n || c->nnz != cs[0]->nnz) LZX_FAIL(LZX_ERR_STATE, "handles hold different graphs");
    }
    // an in-process group has no collective at the hand-over: its sparse exchange lists are compared here, pairwise, in length
    // and in content (host memory; the other transports do it inside the hand-over, lzx_comm_check_sparse)
    if (cs.size() > 1 && cs[0]->comm_kind == 1) LZX_TRY(lzx_comm_check_sparse_local(cs));
    return LZX_OK;
}

static int sync_all(std::vector<lzx_ctx *> &cs)
{
    for (lzx_ctx *c : cs) {
        LZX_HIP(hipSetDevice(c->device));
        LZX_HIP(hipStreamSynchronize(c->stream));
        LZX_HIP(hipStreamSynchronize(c->stream2));
        LZX_TRY(lzx_comm_ipc_check(c));   // peer windows: a peer that never arrived ended a wait at its deadline
    }
    return LZX_OK;
}

static int ensure_capacity(lzx_ctx *c, u32 k, bool qf32)
{
    LZX_HIP(hipSetDevice(c->device));
    if (k > c->k_cap) {
        if (c->d_alpha) (void)hipFree(c->d_alpha);
        if (c->d_beta) (void)hipFree(c->d_beta);
        c->d_alpha = c->d_beta = nullptr;
        c->k_cap = 0;
        LZX_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_alpha), sizeof(double) * k));
        LZX_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_beta), sizeof(double) * k));
        c->k_cap = k;
    }
    if (qf32) {
        // N4: the basis is stored as fp32; the recurrence's three live vectors stay fp64 (d_ring).  The fp64 basis of an
        // earlier decomposition on this handle goes first: the mode's point is half the HBM
        if (c->d_Q) (void)hipFree(c->d_Q);
        c->d_Q = nullptr;
        c->q_cols = 0;
        if (k > c->qf_cols) {
            if (c->d_Qf) (void)hipFree(c->d_Qf);
            c->d_Qf = nullptr;
            c->qf_cols = 0;
            LZX_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_Qf), sizeof(float) * (size_t)k * c->ldq));
            c->qf_cols = k;
        }
        for (double *&r : c->d_ring)
            if (!r) LZX_HIP(hipMalloc(reinterpret_cast<void **>(&r), sizeof(double) * c->ldq));
        return LZX_OK;
    }
    // ... and the fp32 form's buffers when the handle goes back to the fp64 basis
    if (c->d_Qf) (void)hipFree(c->d_Qf);
    c->d_Qf = nullptr;
    c->qf_cols = 0;
    for (double *&r : c->d_ring) {
        if (r) (void)hipFree(r);
        r = nullptr;
    }
    if (k > c->q_cols) {
        if (c->d_Q) (void)hipFree(c->d_Q);
        c->d_Q = nullptr;
        c->q_cols = 0;
        LZX_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_Q), sizeof(double) * (size_t)k * c->ldq));
        c->q_cols = k;
    }
    return LZX_OK;
}

// which form of the loop a decomposition on these handles takes (lanczos_loop)
static bool loop_is_lazy(const lzx_ctx *c0)
{
    if (c0->reorth_opt > 0) return false;   // R1 runs the reference's operation order on the normalised basis
    if (c0->ref_order_opt > 0) return false;   // option reference_order: the reference's operation AND reduction order
    if (c0->qf32_opt > 0 && c0->lazy_opt != 0) return true;   // the fp32-stored basis lives in the lazy loop: asking for it selects it
    return c0->lazy_opt > 0 || (c0->lazy_opt < 0 && (lzx_exchanges(c0) || c0->codes16));
}
// fp64 column j of the basis: a column of d_Q, or -- basis stored as fp32 -- one of the three live vectors
static inline double *basis_col(const lzx_ctx *c, u32 j)
{
    return c->qf32 ? c->d_ring[j % 3] : c->d_Q + (size_t)j * c->ldq;
}

// timing marks on the first handle's stream: the interval that ENDS at a mark is billed to its category
enum { CAT_NONE = 0, CAT_SPMV = 1, CAT_VEC = 2, CAT_COMM = 3 };
struct Marks {
    lzx_ctx *c;
    size_t used = 0;
    std::vector<int> cat;
    bool on = true;   // every mark is a barrier packet (~5 us of drained pipeline): only every few iterations carry marks
    u32 sampled = 0;  // iterations that carried marks
    int begin_iteration(u32 j, u32 every)
    {
        on = every <= 1 || j % every == 0;
        if (on) ++sampled;
        return tick(CAT_NONE);   // what ran since the last marked iteration is billed to nobody
    }
    int tick(int category)
    {
        if (!on) return LZX_OK;
        LZX_HIP(hipSetDevice(c->device));
        if (used == c->ev_pool.size()) {
            hipEvent_t ev;
            LZX_HIP(hipEventCreate(&ev));
            c->ev_pool.push_back(ev);
        }
        LZX_HIP(hipEventRecord(c->ev_pool[used], c->stream));
        cat.push_back(category);
        ++used;
        return LZX_OK;
    }
};

static u64 spmv_algorithmic_bytes(const lzx_ctx *c)
{
    // column indices once, row pointers once (W_p = 4), every x element once, y written once.
    return 4ull * c->nnz_local + 4ull * ((u64)c->n_loc_real + 1) + 8ull * c->n + 8ull * c->n_loc_real;
}

// What one pass over the caller's start vector tells the hand-over (round 5: the reference-comparable figure times the
// constructor, parallel-final/main.cu:104-116, and at n = 10 M the 80 MB pageable upload of x0 and the 10 M-term dependent
// chain of ||x0||^2 were 6 ms of a 29 ms decomposition):
//   constant  every entry equals x0[0] -- the reference's own start vector is ones (parallel-final/main.cu:79,
//             serial/main.cc:79) -- so nothing needs to cross PCIe: the device fills the vector itself;
//   exact     every entry is an integer of magnitude <= 2^26 and the squares add up to <= 2^53: every partial sum of
//             serial/'s left-to-right loop (serial/lib/lanczos.cc:155-161) is then an exactly representable integer, the loop
//             rounds nowhere, and the sum may be formed in any order -- in parallel -- with the same bits.
// Several host threads, each its own stretch; a stretch gives up as soon as neither property can hold any more, so a
// general vector costs one cache line's look, not a pass.
struct X0Scan {
    bool constant = false, exact = false;
    double sum_sq = 0.0;
};
static X0Scan scan_start_vector(const double *x0, u64 n)
{
    X0Scan out;
    if (n == 0) return out;
    const u32 hw = std::max(1u, std::thread::hardware_concurrency());
    const u32 T = (u32)std::min<u64>(std::min<u32>(8u, hw), n / (1u << 18) + 1);
    std::vector<double> part(T, 0.0);
    std::atomic<bool> all_const{true}, all_int{true};
    const double first = x0[0];
    auto work = [&](u32 t) {
        const u64 a = n * t / T, b = n * (t + 1) / T;
        // four accumulators: used only when every square is an exact integer, and then any order gives the same bits -- so the
        // inner loop is free of the one dependent chain (and of libm calls: an integer is what survives the round trip through
        // int64, magnitudes <= 2^26 checked first)
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        bool cst = true, itg = true;
        for (u64 i0 = a; i0 < b; i0 += 4096) {
            const u64 i1 = std::min<u64>(b, i0 + 4096);
            u64 i = i0;
            for (; i + 4 <= i1; i += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double v = x0[i + u];
                    cst &= v == first;
                    const bool small = std::fabs(v) <= 67108864.0;   // (NaN fails)
                    itg &= small && (double)(int64_t)(small ? v : 0.0) == v;
                    acc[u] += v * v;
                }
            }
            for (; i < i1; ++i) {
                const double v = x0[i];
                cst &= v == first;
                const bool small = std::fabs(v) <= 67108864.0;
                itg &= small && (double)(int64_t)(small ? v : 0.0) == v;
                acc[0] += v * v;
            }
            if (!cst) all_const.store(false, std::memory_order_relaxed);
            if (!itg) all_int.store(false, std::memory_order_relaxed);
            if (!all_const.load(std::memory_order_relaxed) && !all_int.load(std::memory_order_relaxed)) return;
        }
        part[t] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    };
    {
        std::vector<std::thread> th;
        for (u32 t = 1; t < T; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &t : th) t.join();
    }
    out.constant = all_const.load();
    if (all_int.load()) {
        double s = 0.0;
        for (u32 t = 0; t < T; ++t) s += part[t];   // integers: exact as long as the total is
        if (s < 9007199254740992.0) {   // strictly below 2^53: no partial sum of any order was rounded
            out.exact = true;
            out.sum_sq = s;
        }
    }
    return out;
}

// Upload x0, normalise it into q_0 and size the resident basis for k vectors.
static int lanczos_prepare(std::vector<lzx_ctx *> &cs, const double *x0, u32 k, double *x_norm_out)
{
    if (!x0 || k == 0) LZX_FAIL(LZX_ERR_ARG, "lzx_lanczos: bad argument");
    LZX_TRY(check_graphs(cs));
    lzx_ctx *c0 = cs[0];
    const u64 n = c0->n;
    const bool multi = lzx_exchanges(c0);
    const bool lazy = loop_is_lazy(c0);
    // LZX_TRACE_PREPARE=1: where this call's milliseconds go (stderr), for tools/handover_probe.py
    static const bool trace = getenv("LZX_TRACE_PREPARE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        for (lzx_ctx *c : cs) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); }
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[lzx prepare] %-34s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
        t_last = t;
    };
    for (lzx_ctx *c : cs) {
        if (c->reorth_opt != c0->reorth_opt || c->qf32_opt != c0->qf32_opt) LZX_FAIL(LZX_ERR_STATE, "handles carry different loop options");
        if (c->qf32_opt > 0 && (!lazy || c->basis_u_opt == 0))
            LZX_FAIL(LZX_ERR_STATE, "basis_fp32 needs the lazy loop (not with lazy_normalisation = 0, not together with reorthogonalise)");
        if (c->ref_order_opt > 0 && (multi || cs.size() > 1))
            LZX_FAIL(LZX_ERR_STATE, "option reference_order runs on one rank");
        if (c->ref_order_opt > 0 && c->qf32_opt > 0)
            LZX_FAIL(LZX_ERR_STATE, "option reference_order keeps the fp64 basis (not together with basis_fp32)");
    }
    // From here on the resident basis of an earlier decomposition is gone (its buffers may be reallocated or change form):
    // nothing is resident and nothing prepared until this call has succeeded on every handle -- a failure part-way leaves
    // handles on which lzx_multout / lzx_lanczos_fetch report "no decomposition", not a stale or null basis.
    for (lzx_ctx *c : cs) {
        c->k_prep = 0;
        c->k_last = c->k_done = 0;
        c->ymon_valid = 0;
        c->qf32 = false;
    }
    const bool want_qf32 = c0->qf32_opt > 0;
    for (lzx_ctx *c : cs) LZX_TRY(ensure_capacity(c, k, want_qf32));
    for (lzx_ctx *c : cs) c->qf32 = want_qf32;
    lap("checks, basis sized");

    // ||x0||: left-to-right sum of squares on the host, then sqrt (serial/lib/lanczos.cc:155-161) -- one dependent chain of
    // n additions (7 ms at n = 10 M), on a helper thread while this one sizes the basis, clears it and uploads x0 (a pageable
    // copy that blocks its caller for about as long)
    // (round 5) unless one look at x0 shows that the sum is exact in any order -- then it was formed by scan_start_vector's
    // threads -- and, for a constant vector, that there is nothing to upload
    const X0Scan scan = c0->x0_scan_opt != 0 ? scan_start_vector(x0, n) : X0Scan{};
    lap("look at x0");
    double ss = scan.sum_sq;
    struct Joiner {
        std::thread t;
        ~Joiner() { if (t.joinable()) t.join(); }
    } norm_thread;
    if (!scan.exact)
        norm_thread.t = std::thread([&ss, x0, n]() {
            double acc = 0.0;
            for (u64 i = 0; i < n; ++i) acc += x0[i] * x0[i];
            ss = acc;
        });

    for (lzx_ctx *c : cs) {
        LZX_HIP(hipSetDevice(c->device));
        if (c->qf32) {
            for (double *r : c->d_ring) LZX_HIP(hipMemsetAsync(r, 0, sizeof(double) * c->ldq, c->stream));
        } else {
            // column 0 (its padded rows must be 0) and the zero tail behind every column; columns 1.. are written whole by
            // the loop (clearing all k columns cost 4 GB of memset at C3)
            LZX_HIP(hipMemsetAsync(c->d_Q, 0, sizeof(double) * c->ldq, c->stream));
            if (k > 1)
                LZX_HIP(hipMemset2DAsync(c->d_Q + c->ldq + c->n_loc_pad, sizeof(double) * c->ldq, 0, sizeof(double) * LZX_TAIL, k - 1, c->stream));
        }
        if (scan.constant) LZX_TRY(lzx_launch_fill(c, c->d_io, x0[0], n));
        else LZX_HIP(hipMemcpyAsync(c->d_io, x0, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
        c->x0_was_constant = scan.constant;
    }
    lap("clears + x0 on the device");
    if (norm_thread.t.joinable()) norm_thread.t.join();
    lap("serial norm chain (if any)");
    const double x_norm = std::sqrt(ss);
    if (x_norm_out) *x_norm_out = x_norm;
    for (lzx_ctx *c : cs) {
        LZX_HIP(hipSetDevice(c->device));
        double *col0 = basis_col(c, 0);
        // q_0 = x0 / ||x0|| (serial/lib/lanczos.cc:16-17), scattered into the internal order
        if (!multi) {
            LZX_TRY(lzx_launch_permute_in(c, c->d_io, col0, x_norm));
        } else {
            // every rank holds all of x0: full vector in hand-over layout, own slice into the basis, and the
            // prefix of every slice that carries vertices with an edge into the exchange buffer (no communication)
            LZX_TRY(lzx_launch_permute_in(c, c->d_io, c->d_ybuf, x_norm));
            LZX_HIP(hipMemcpyAsync(col0, c->d_ybuf + (size_t)c->rank * c->n_loc_pad,
                                   sizeof(double) * c->n_loc_pad, hipMemcpyDeviceToDevice, c->stream));
            LZX_TRY(lzx_launch_relayout(c, c->d_ybuf, c->d_xbuf));
        }
        if (c->qf32) LZX_TRY(lzx_launch_to_f32(c, col0, c->d_Qf, c->n_loc_pad));   // column 0 of the stored basis
    }
    // the lazy loop (lanczos_loop) carries the rows without an edge as one scalar recurrence
    {
        for (lzx_ctx *c : cs) {
            LZX_HIP(hipSetDevice(c->device));
            c->iso_on = lazy && c->iso_opt != 0;
            c->iso_filled = false;
            c->iso_cols_filled = 1;
            c->basis_u = lazy && c->basis_u_opt != 0;
            // columns 1.. are written by the loop up to rows_live only; with several ranks the exchanged prefix may reach
            // one slice further (into rows without an edge, which nobody reads): keep that slice clean
            if (!c->qf32 && c->basis_u && c->iso_on && k > 1 && c->rows_live < c->n_loc_pad)
                LZX_HIP(hipMemset2DAsync(c->d_Q + c->ldq + c->rows_live, sizeof(double) * c->ldq, 0,
                                         sizeof(double) * std::min<u32>(LZX_SLICE, c->n_loc_pad - c->rows_live), k - 1, c->stream));
            if (c->iso_on) LZX_TRY(lzx_launch_iso_prepare(c, k));
        }
    }
    lap("q_0, rows without an edge");
    LZX_TRY(sync_all(cs));
    for (lzx_ctx *c : cs) c->k_prep = k;
    return LZX_OK;
}

// Iterations [k_done, min(k_prep, k_done + steps)) of the prepared decomposition, on the vectors lanczos_prepare (or the
// previous call) left in HBM.  Everything an iteration needs from its predecessor lives on the device -- basis columns,
// alpha / beta, the norm partials, the exchanged vector and its events -- so a decomposition can be advanced in chunks
// (a convergence monitor that SAVES iterations) with the same bits as in one go.
static int lanczos_loop(std::vector<lzx_ctx *> &cs, u32 steps, lzx_stats *stats)
{
    LZX_TRY(check_graphs(cs));
    lzx_ctx *c0 = cs[0];
    const u32 k = c0->k_prep;
    for (lzx_ctx *c : cs)
        if (c->k_prep == 0 || c->k_prep != k || c->k_done != c0->k_done || (c->qf32 ? (!c->d_Qf || c->qf_cols < k) : (!c->d_Q || c->q_cols < k)))
            LZX_FAIL(LZX_ERR_STATE, "lzx_lanczos_run: no prepared start vector (prepare, then run, with no SpMV / benchmark call on the handle between them)");
    const u32 j0 = c0->k_done;
    const u32 j1 = (u32)std::min<u64>(k, (u64)j0 + steps);
    const bool multi = lzx_exchanges(c0);
    LZX_TRY(sync_all(cs));

    Marks mk{c0};
    LZX_HIP(hipSetDevice(c0->device));
    const auto t0 = std::chrono::steady_clock::now();
    LZX_TRY(mk.tick(CAT_NONE));

    std::vector<const double *> src(cs.size());
    std::vector<double *> dst(cs.size());
    const u32 np = lzx_spmv_partials(c0);

    const bool overlap = multi && c0->overlap;
    // Several ranks: exchange and multiply the UNNORMALISED vector u_j = beta_{j-1} q_j, so that beta_{j-1}^2 =
    // ||u_j||^2 travels in the same all-reduce as u_j . (A u_j) (k_lazy_update): one 2-double all-reduce per
    // iteration instead of two dependent 1-double ones, and the exchange of u_{j+1} starts straight after the vector
    // kernel instead of after a second reduction.  Same recurrence, operands rounded at slightly different places
    // (w / beta instead of A (u / beta)); one rank keeps the reference's exact operation order below.
    // One rank in blocked mode (whose sums are already ordered differently from the reference's) takes the same form:
    // it saves k_scale's pass over v and a launch; in plain mode one rank keeps the reference's order bit for bit.
    const bool lazy = loop_is_lazy(c0);
    // the lazy loop's vector kernel completes v for the rows of multi-item gather bands itself, so the blocked SpMV leaves its
    // k_pb_finish launch out (lzx_ctx::pb_deferring; test shape defer_finish = 0 keeps the launch); every other consumer of v gets it
    struct Deferring {
        std::vector<lzx_ctx *> &cs;
        Deferring(std::vector<lzx_ctx *> &h, bool on) : cs(h) { for (lzx_ctx *c : cs) c->pb_deferring = on && c->defer_opt != 0; }
        ~Deferring() { for (lzx_ctx *c : cs) c->pb_deferring = false; }
    } deferring(cs, lazy);
    const bool mail_ok = multi && lzx_comm_mail_usable(cs);
    // timing marks on every 4th iteration (every one when k is small); the sums below are scaled to all iterations run
    const u32 every = c0->marks_every_opt > 0 ? (u32)c0->marks_every_opt : (k >= 8 ? 4u : 1u);
    for (u32 j = j0; lazy && j < j1; ++j) {
        const bool first = j == 0, last = j == k - 1;
        LZX_TRY(mk.begin_iteration(j, every));
        if (overlap && j > 0) {
            for (lzx_ctx *c : cs) {
                LZX_HIP(hipSetDevice(c->device));
                LZX_HIP(hipStreamWaitEvent(c->stream, c->ev_c0, 0));
            }
            LZX_TRY(mk.tick(CAT_COMM));
        }
        for (lzx_ctx *c : cs) {
            LZX_HIP(hipSetDevice(c->device));
            const double *uj = first ? basis_col(c, 0) : (c->basis_u ? basis_col(c, j) : c->d_u[j & 1]);   // u_0 = q_0
            SpmvLaunch l{multi ? c->d_xbuf : uj, uj, c->d_v, c->d_partials};
            l.live_rows_only = true;   // k_lazy_update below takes (A u)_i = 0 for rows without an edge
            if (overlap && j > 0) l.chunk1_ready = c->ev_c1;
            LZX_TRY(lzx_launch_spmv(c, l));
        }
        LZX_TRY(mk.tick(CAT_SPMV));
        u32 np2 = 0;
        const bool mail = multi && mail_ok;
        if (mail) {
            // in-process group: each rank's pair goes straight into every peer's mailbox (one barrier, no copies)
            LZX_TRY(lzx_comm_mail_reduce2(cs, j & 1u, first));
        } else if (multi && lzx_comm_fused_reduce2(c0)) {
            // peer windows: the rank's two sums are closed and all-reduced by one single-workgroup launch
            LZX_TRY(lzx_comm_reduce2_allreduce(c0, c0->d_partials, lzx_spmv_partials(c0), c0->d_partials2, first ? 0 : c0->np2_last));
        } else if (multi) {
            for (lzx_ctx *c : cs) {
                LZX_HIP(hipSetDevice(c->device));
                // [u_j . w, ||u_j||^2] of this rank (the second from the previous iteration's k_lazy_update)
                LZX_TRY(lzx_launch_reduce2(c, c->d_partials, lzx_spmv_partials(c), c->d_partials2, first ? 0 : c->np2_last, c->d_scal + 0));
            }
            LZX_TRY(lzx_comm_allreduce_sum(cs, 0, 2));
        }
        // (no timing mark here: every mark is a barrier packet, ~5 us of drained pipeline between dependent kernels;
        //  the reduction and its all-reduce are billed to the vector work, the exposed part of the all-gather to comm)
        for (lzx_ctx *c : cs) {
            LZX_HIP(hipSetDevice(c->device));
            const double *uj = first ? basis_col(c, 0) : (c->basis_u ? basis_col(c, j) : c->d_u[j & 1]);
            // the resident basis holds u_j (basis_u): nothing normalised is stored, the previous column is divided by
            // beta_{j-2} on the way in, the next one written in place
            double *q_store = (first || c->basis_u) ? nullptr : basis_col(c, j);
            double *u_store = last ? nullptr : (c->basis_u ? basis_col(c, j + 1) : c->d_u[(j + 1) & 1]);
            float *f_store = (c->qf32 && u_store) ? c->d_Qf + (size_t)(j + 1) * c->ldq : nullptr;   // N4: the column as it is stored
            const double *pdiv = (c->basis_u && j >= 2) ? c->d_beta + (j - 2) : nullptr;
            // one rank: both sums are closed in the kernel's prologue from the partials themselves (no reduce launch);
            // the norm partials alternate between two arrays, the kernel reads one while writing the other
            double *p_out = multi ? c->d_partials2 : ((j & 1) ? c->d_partials3 : c->d_partials2);
            const double *p_in = multi ? nullptr : ((j & 1) ? c->d_partials2 : c->d_partials3);
            if (!multi)
                LZX_TRY(lzx_launch_lazy_update_local(c, c->d_v, c->rows_live, uj, first ? nullptr : basis_col(c, j - 1), c->d_partials,
                                                     lzx_spmv_partials(c), p_in, first ? 0 : c->np2_last, first ? 1 : 0, c->d_alpha + j,
                                                     first ? nullptr : c->d_beta + (j - 1), q_store, u_store, p_out, &np2, pdiv, f_store));
            else
            LZX_TRY(lzx_launch_lazy_update(c, c->d_v, c->rows_live, uj, first ? nullptr : basis_col(c, j - 1),
                                           mail ? c->d_mail + (size_t)(j & 1u) * 64 * 2 : c->d_scal + 0, first ? 1 : 0,
                                           c->d_alpha + j, first ? nullptr : c->d_beta + (j - 1), q_store, u_store, c->d_partials2, &np2, pdiv, f_store,
                                           mail ? (u32)cs.size() : 0u));
            c->np2_last = np2;
        }
        LZX_TRY(mk.tick(CAT_VEC));
        if (last || !multi) {
            if (last) break;
            continue;
        }
        auto next_u = [&](lzx_ctx *c) -> const double * {
            return c->basis_u ? basis_col(c, j + 1) : c->d_u[(j + 1) & 1];
        };
        for (size_t i = 0; i < cs.size(); ++i) {
            src[i] = next_u(cs[i]);
            dst[i] = cs[i]->d_xbuf;
        }
        if (!overlap) {
            if (c0->xfp32) LZX_TRY(lzx_comm_allgather_fp32(cs, src.data(), true));   // N4: the same prefix, rounded to fp32 on the wire
            else
            LZX_TRY(lzx_comm_allgather(cs, src.data(), dst.data(), c0->xs, false, true));   // only the prefix that has edges
            LZX_HIP(hipSetDevice(c0->device));
            LZX_TRY(mk.tick(CAT_COMM));
        } else {
            LZX_TRY(lzx_comm_order(cs, /*from main*/ false, /*to exchange*/ true));
            LZX_TRY(lzx_comm_allgather(cs, src.data(), dst.data(), c0->xs0, true, true));
            for (lzx_ctx *c : cs) {
                LZX_HIP(hipSetDevice(c->device));
                LZX_HIP(hipEventRecord(c->ev_c0, c->stream2));
            }
            if (c0->sparse) {   // every peer gets only what its rows reference
                for (size_t i = 0; i < cs.size(); ++i) src[i] = next_u(cs[i]);
                LZX_TRY(lzx_comm_sparse_chunk1(cs, src.data(), true));
            } else {
                for (size_t i = 0; i < cs.size(); ++i) {
                    src[i] = next_u(cs[i]) + cs[i]->xs0;
                    dst[i] = cs[i]->d_xbuf + (size_t)cs[i]->world * cs[i]->xs0;
                }
                LZX_TRY(lzx_comm_allgather(cs, src.data(), dst.data(), c0->xs - c0->xs0, true, true));
            }
            for (lzx_ctx *c : cs) {
                LZX_HIP(hipSetDevice(c->device));
                LZX_HIP(hipEventRecord(c->ev_c1, c->stream2));
            }
        }
    }
    const u32 reorth = (!lazy && c0->reorth_opt > 0) ? (u32)c0->reorth_opt : 0u;
    // Option reference_order: the three reductions in serial/'s order (lzx_kernels.hip: k_ref_*).  The scalars then
    // arrive in d_scal[0 / 1] the way the several-rank loop's all-reduced ones do, and the vector kernels are the same.
    const bool ref = !lazy && !multi && c0->ref_order_opt > 0;
    const bool scal = multi || ref;   // alpha_j / ||v||^2 come as ONE device scalar each instead of block partials
    for (u32 j = j0; !lazy && j < j1; ++j) {
        LZX_TRY(mk.begin_iteration(j, every));
        // v = A q_j ; partials of alpha_j
        if (overlap && j > 0) {
            // chunk 0 of q_j (the high-degree end of every slice, where the staged hub entries and nearly all
            // gathers are) has to be here; chunk 1 may still be on the wire while the SpMV works on chunk 0
            for (lzx_ctx *c : cs) {
                LZX_HIP(hipSetDevice(c->device));
                LZX_HIP(hipStreamWaitEvent(c->stream, c->ev_c0, 0));
            }
            LZX_TRY(mk.tick(CAT_COMM));
        }
        for (lzx_ctx *c : cs) {
            LZX_HIP(hipSetDevice(c->device));
            const double *qj = c->d_Q + (size_t)j * c->ldq;
            if (ref) {
                LZX_TRY(lzx_launch_ref_spmv(c, qj, c->d_v));
                LZX_TRY(lzx_launch_ref_dot(c, c->d_v, qj, c->d_scal + 0));   // alpha_j = <v, q_j>, left to right
                continue;
            }
            SpmvLaunch l{multi ? c->d_xbuf : qj, qj, c->d_v, c->d_partials};
            if (overlap && j > 0) l.chunk1_ready = c->ev_c1;
            LZX_TRY(lzx_launch_spmv(c, l));
        }
        LZX_TRY(mk.tick(CAT_SPMV));

        // R1: the Arnoldi pass of serial/lib/lanczos.cc:85-90 -- A q_j against q_0 .. q_{j-2}, modified Gram-Schmidt in the
        // reference's order (every inner product over the v the previous update left), as a chain of j launches: the
        // first forms <v, q_0>, launch t applies vector t - 1 and forms the next inner product, the last one's is
        // <v, q_j>: the alpha_j partials, which replace those the SpMV formed before the pass.  The partial buffers
        // alternate so that the last launch writes d_partials3 (k_axpy_norm writes d_partials2 while reading them).
        const bool re = reorth && j % reorth == 0 && j > 2;
        u32 np_re = 0;
        if (re) {
            for (u32 t = 0; t < j; ++t) {
                for (lzx_ctx *c : cs) {
                    LZX_HIP(hipSetDevice(c->device));
                    double *out = ((j - 1 - t) & 1) ? c->d_partials2 : c->d_partials3;
                    const double *in = ((j - 1 - t) & 1) ? c->d_partials3 : c->d_partials2;
                    const double *qm = t == 0 ? nullptr : c->d_Q + (size_t)(t - 1) * c->ldq;
                    const double *qn = c->d_Q + (size_t)(t + 1 < j ? t : j) * c->ldq;
                    u32 npo = 0;
                    LZX_TRY(lzx_launch_mgs_step(c, c->d_v, qm, scal ? nullptr : in, scal ? 0u : np_re, scal ? c->d_scal + 2 : nullptr, qn, out, &npo));
                    if (multi) LZX_TRY(lzx_launch_reduce(c, out, npo, c->d_scal + (t + 1 < j ? 2 : 0), 0));
                    // reference_order: the inner product over the updated v, left to right (serial/lib/lanczos.cc:86-90, 163-171)
                    if (ref) LZX_TRY(lzx_launch_ref_dot(c, c->d_v, qn, c->d_scal + (t + 1 < j ? 2 : 0)));
                    if (c == cs.back()) np_re = npo;
                }
                if (multi) LZX_TRY(lzx_comm_allreduce_sum(cs, t + 1 < j ? 2 : 0));
            }
            LZX_TRY(mk.tick(CAT_VEC));
        }

        if (multi && !re) {
            for (lzx_ctx *c : cs) {
                LZX_HIP(hipSetDevice(c->device));
                LZX_TRY(lzx_launch_reduce(c, c->d_partials, lzx_spmv_partials(c), c->d_scal + 0, 0));
            }
            LZX_TRY(lzx_comm_allreduce_sum(cs, 0));
            LZX_TRY(mk.tick(CAT_COMM));
        }

        if (j == k - 1) {
            // last step: only alpha_{k-1} is an output (the reference also updates v, then drops it)
            for (lzx_ctx *c : cs) {
                LZX_HIP(hipSetDevice(c->device));
                if (scal) LZX_HIP(hipMemcpyAsync(c->d_alpha + j, c->d_scal + 0, sizeof(double), hipMemcpyDeviceToDevice, c->stream));
                else LZX_TRY(lzx_launch_reduce(c, re ? c->d_partials3 : c->d_partials, re ? np_re : np, c->d_alpha + j, 0));
            }
            LZX_TRY(mk.tick(CAT_VEC));
            break;
        }

        u32 np2 = 0;
        for (lzx_ctx *c : cs) {
            LZX_HIP(hipSetDevice(c->device));
            const double *qj = c->d_Q + (size_t)j * c->ldq;
            const double *qjm1 = j > 0 ? c->d_Q + (size_t)(j - 1) * c->ldq : nullptr;
            LZX_TRY(lzx_launch_axpy_norm(c, c->d_v, qj, qjm1, scal ? c->d_scal + 0 : (re ? c->d_partials3 : c->d_partials),
                                         scal ? 1 : (re ? np_re : lzx_spmv_partials(c)), c->d_alpha + j,
                                         j > 0 ? c->d_beta + (j - 1) : nullptr, c->d_partials2, &np2));
            if (ref) LZX_TRY(lzx_launch_ref_dot(c, c->d_v, c->d_v, c->d_scal + 1));   // ||v||^2, left to right (k_scale takes the root)
        }
        if (multi) LZX_TRY(mk.tick(CAT_VEC));   // one rank: a single mark after k_scale covers both vector kernels

        if (multi) {
            for (lzx_ctx *c : cs) {
                LZX_HIP(hipSetDevice(c->device));
                LZX_TRY(lzx_launch_reduce(c, c->d_partials2, np2, c->d_scal + 1, 0));
            }
            LZX_TRY(lzx_comm_allreduce_sum(cs, 1));
            LZX_TRY(mk.tick(CAT_COMM));
        }

        for (lzx_ctx *c : cs) {
            LZX_HIP(hipSetDevice(c->device));
            LZX_TRY(lzx_launch_scale(c, c->d_v, c->d_Q + (size_t)(j + 1) * c->ldq,
                                     scal ? c->d_scal + 1 : c->d_partials2, scal ? 1 : np2, c->d_beta + j));
        }
        LZX_TRY(mk.tick(CAT_VEC));

        if (multi && !overlap) {
            for (size_t i = 0; i < cs.size(); ++i) {
                src[i] = cs[i]->d_Q + (size_t)(j + 1) * cs[i]->ldq;
                dst[i] = cs[i]->d_xbuf;
            }
            if (c0->xfp32) LZX_TRY(lzx_comm_allgather_fp32(cs, src.data(), true));
            else
            LZX_TRY(lzx_comm_allgather(cs, src.data(), dst.data(), c0->xs, false, true));   // only the prefix that has edges
            LZX_HIP(hipSetDevice(c0->device));
            LZX_TRY(mk.tick(CAT_COMM));
        }
        if (overlap) {
            // exchange streams: both chunks of q_{j+1} back to back, an event after each.  They start once every
            // handle's k_scale is done (which also means every SpMV that read the previous x is done).
            LZX_TRY(lzx_comm_order(cs, /*from main*/ false, /*to exchange*/ true));
            for (size_t i = 0; i < cs.size(); ++i) {
                src[i] = cs[i]->d_Q + (size_t)(j + 1) * cs[i]->ldq;
                dst[i] = cs[i]->d_xbuf;
            }
            LZX_TRY(lzx_comm_allgather(cs, src.data(), dst.data(), c0->xs0, true, true));
            for (lzx_ctx *c : cs) {
                LZX_HIP(hipSetDevice(c->device));
                LZX_HIP(hipEventRecord(c->ev_c0, c->stream2));
            }
            if (c0->sparse) {
                for (size_t i = 0; i < cs.size(); ++i) src[i] = cs[i]->d_Q + (size_t)(j + 1) * cs[i]->ldq;
                LZX_TRY(lzx_comm_sparse_chunk1(cs, src.data(), true));
            } else {
                for (size_t i = 0; i < cs.size(); ++i) {
                    src[i] = cs[i]->d_Q + (size_t)(j + 1) * cs[i]->ldq + cs[i]->xs0;
                    dst[i] = cs[i]->d_xbuf + (size_t)cs[i]->world * cs[i]->xs0;
                }
                LZX_TRY(lzx_comm_allgather(cs, src.data(), dst.data(), c0->xs - c0->xs0, true, true));
            }
            for (lzx_ctx *c : cs) {
                LZX_HIP(hipSetDevice(c->device));
                LZX_HIP(hipEventRecord(c->ev_c1, c->stream2));
            }
        }
    }
    mk.on = true;
    LZX_TRY(sync_all(cs));
    const auto t1 = std::chrono::steady_clock::now();
    for (lzx_ctx *c : cs) {
        c->k_done = c->k_last = j1;
        if (j1 == k) c->k_prep = 0;   // complete: nothing left to resume
    }

    if (stats) {
        const u32 ran = j1 - j0;
        memset(stats, 0, sizeof *stats);
        stats->loop_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        stats->iters = ran;
        stats->spmv_kernels = c0->pb ? 3 + (c0->pb_finish_grid && !(c0->pb_deferring && c0->pb_defer_ok) ? 1 : 0) : 1 + (c0->fin_grid > 0 ? 1 : 0);
        stats->spmv_bytes = spmv_algorithmic_bytes(c0);
        stats->spmv_ms_min = 1e300;
        for (size_t i = 1; i < mk.used; ++i) {
            float ms = 0.f;
            LZX_HIP(hipEventElapsedTime(&ms, c0->ev_pool[i - 1], c0->ev_pool[i]));
            switch (mk.cat[i]) {
                case CAT_SPMV: stats->spmv_ms += ms; if (ms < stats->spmv_ms_min) stats->spmv_ms_min = ms; break;
                case CAT_VEC: stats->vec_ms += ms; break;
                case CAT_COMM: stats->comm_ms += ms; break;
                default: break;
            }
        }
        if (stats->spmv_ms_min == 1e300) stats->spmv_ms_min = 0.0;
        if (mk.sampled > 0 && mk.sampled < ran) {   // marks were carried by `sampled` of the iterations: scale the sums
            const double f = (double)ran / (double)mk.sampled;
            stats->spmv_ms *= f;
            stats->vec_ms *= f;
            stats->comm_ms *= f;
        }
    }
    return LZX_OK;
}

// Download alpha, beta and (optionally) the basis in the caller's vertex order.
static int lanczos_fetch(std::vector<lzx_ctx *> &cs, u32 k, double *alpha, double *beta, double *Q)
{
    if (!alpha || k == 0 || (k > 1 && !beta)) LZX_FAIL(LZX_ERR_ARG, "lzx_lanczos_fetch: bad argument");
    LZX_TRY(check_graphs(cs));
    lzx_ctx *c0 = cs[0];
    if (c0->k_last < k) LZX_FAIL(LZX_ERR_STATE, "lzx_lanczos_fetch: only %u iterations are resident", c0->k_last);
    const u64 n = c0->n;
    const bool multi = lzx_exchanges(c0);
    std::vector<const double *> src(cs.size());
    std::vector<double *> dst(cs.size());
    LZX_HIP(hipSetDevice(c0->device));
    LZX_HIP(hipMemcpy(alpha, c0->d_alpha, sizeof(double) * k, hipMemcpyDeviceToHost));
    if (k > 1) LZX_HIP(hipMemcpy(beta, c0->d_beta, sizeof(double) * (k - 1), hipMemcpyDeviceToHost));

    if (Q) {
        // rows without an edge are kept as scalars times q_0 by the lazy loop: written out once somebody wants the basis
        for (lzx_ctx *c : cs) {
            if (!c->iso_on || c->iso_cols_filled >= c->k_last) continue;   // (a decomposition advanced in chunks: the new columns)
            LZX_HIP(hipSetDevice(c->device));
            LZX_TRY(lzx_launch_iso_fill(c, c->k_last));
            c->iso_filled = true;
        }
        // k contiguous vectors in the caller's vertex order (cu_lanczos.cu:126 layout)
        for (u32 j = 0; j < k; ++j) {
            if (multi) {
                for (size_t i = 0; i < cs.size(); ++i) {
                    lzx_ctx *c = cs[i];
                    if (c->qf32) {   // the stored column, widened (this rank's slice goes through d_v)
                        LZX_HIP(hipSetDevice(c->device));
                        LZX_TRY(lzx_launch_widen_col(c, j, c->d_v));
                    }
                    src[i] = c->qf32 ? c->d_v : c->d_Q + (size_t)j * c->ldq;
                    dst[i] = c->d_ybuf;
                }
                LZX_TRY(lzx_comm_allgather(cs, src.data(), dst.data(), c0->n_loc_pad));
            }
            LZX_HIP(hipSetDevice(c0->device));
            if (!multi && c0->qf32) LZX_TRY(lzx_launch_widen_col(c0, j, c0->d_ybuf));
            // columns of the lazy loop hold u_j = beta_{j-1} q_j: the division the loop itself applies, on the way out
            LZX_TRY(lzx_launch_permute_out(c0, (multi || c0->qf32) ? c0->d_ybuf : c0->d_Q + (size_t)j * c0->ldq, c0->d_io,
                                           (c0->basis_u && j > 0) ? c0->d_beta + (j - 1) : nullptr));
            LZX_HIP(hipMemcpyAsync(Q + (size_t)j * n, c0->d_io, sizeof(double) * n, hipMemcpyDeviceToHost, c0->stream));
            LZX_TRY(sync_all(cs));
        }
    }
    return LZX_OK;
}

static int lanczos_run(std::vector<lzx_ctx *> &cs, const double *x0, u32 k, double *alpha, double *beta,
                       double *Q, double *x_norm_out, lzx_stats *stats)
{
    if (!x0 || !alpha || k == 0 || (k > 1 && !beta)) LZX_FAIL(LZX_ERR_ARG, "lzx_lanczos_f64: bad argument");
    LZX_TRY(lanczos_prepare(cs, x0, k, x_norm_out));
    LZX_TRY(lanczos_loop(cs, k, stats));
    return lanczos_fetch(cs, k, alpha, beta, Q);
}

extern "C" int lzx_lanczos_f64(lzx_handle h, const double *x0, uint32_t k, double *alpha, double *beta,
                               double *Q, double *x_norm, lzx_stats *stats)
{
    std::vector<lzx_ctx *> cs;
    LZX_TRY(gather_handles(h, cs));
    return lanczos_run(cs, x0, k, alpha, beta, Q, x_norm, stats);
}

extern "C" int lzx_lanczos_prepare_f64(lzx_handle h, const double *x0, uint32_t k, double *x_norm)
{
    std::vector<lzx_ctx *> cs;
    LZX_TRY(gather_handles(h, cs));
    return lanczos_prepare(cs, x0, k, x_norm);
}

extern "C" int lzx_lanczos_run(lzx_handle h, lzx_stats *stats)
{
    std::vector<lzx_ctx *> cs;
    LZX_TRY(gather_handles(h, cs));
    return lanczos_loop(cs, 0xffffffffu, stats);
}

extern "C" int lzx_lanczos_run_steps(lzx_handle h, uint32_t steps, lzx_stats *stats)
{
    std::vector<lzx_ctx *> cs;
    LZX_TRY(gather_handles(h, cs));
    if (steps == 0) LZX_FAIL(LZX_ERR_ARG, "lzx_lanczos_run_steps: steps == 0");
    return lanczos_loop(cs, steps, stats);
}

extern "C" int lzx_lanczos_run_steps_local(lzx_handle *hs, int world, uint32_t steps, lzx_stats *stats)
{
    std::vector<lzx_ctx *> cs;
    LZX_TRY(gather_handles_local(hs, world, cs));
    if (steps == 0) LZX_FAIL(LZX_ERR_ARG, "lzx_lanczos_run_steps: steps == 0");
    return lanczos_loop(cs, steps, stats);
}

extern "C" int lzx_lanczos_prepare_f64_local(lzx_handle *hs, int world, const double *x0, uint32_t k, double *x_norm)
{
    std::vector<lzx_ctx *> cs;
    LZX_TRY(gather_handles_local(hs, world, cs));
    return lanczos_prepare(cs, x0, k, x_norm);
}

extern "C" int lzx_lanczos_progress(lzx_handle h, uint32_t *done, uint32_t *prepared)
{
    if (!h) LZX_FAIL(LZX_ERR_ARG, "lzx_lanczos_progress: null handle");
    if (done) *done = h->k_done;
    if (prepared) *prepared = h->k_prep;
    return LZX_OK;
}

extern "C" int lzx_lanczos_fetch_f64(lzx_handle h, uint32_t k, double *alpha, double *beta, double *Q)
{
    std::vector<lzx_ctx *> cs;
    LZX_TRY(gather_handles(h, cs));
    return lanczos_fetch(cs, k, alpha, beta, Q);
}

extern "C" int lzx_lanczos_fetch_f64_local(lzx_handle *hs, int world, uint32_t k, double *alpha, double *beta, double *Q)
{
    std::vector<lzx_ctx *> cs;
    LZX_TRY(gather_handles_local(hs, world, cs));
    return lanczos_fetch(cs, k, alpha, beta, Q);
}

extern "C" int lzx_device_count(int *count)
{
    if (!count) LZX_FAIL(LZX_ERR_ARG, "lzx_device_count: null pointer");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;   // no driver / no GPU: zero devices, not an error
    *count = n;
    return LZX_OK;
}

extern "C" int lzx_sync(lzx_handle h)
{
    if (!h) LZX_FAIL(LZX_ERR_ARG, "lzx_sync: null handle");
    LZX_HIP(hipSetDevice(h->device));
    LZX_HIP(hipStreamSynchronize(h->stream));
    return LZX_OK;
}

extern "C" int lzx_lanczos_f64_local(lzx_handle *hs, int world, const double *x0, uint32_t k, double *alpha,
                                     double *beta, double *Q, double *x_norm, lzx_stats *stats)
{
    std::vector<lzx_ctx *> cs;
    LZX_TRY(gather_handles_local(hs, world, cs));
    return lanczos_run(cs, x0, k, alpha, beta, Q, x_norm, stats);
}

// --------------------------------------------------------------------------------------------------
static int spmv_run(std::vector<lzx_ctx *> &cs, const double *x, double *y)
{
    if (!x || !y) LZX_FAIL(LZX_ERR_ARG, "lzx_spmv_f64: bad argument");
    LZX_TRY(check_graphs(cs));
    lzx_ctx *c0 = cs[0];
    const bool multi = lzx_exchanges(c0);
    std::vector<const double *> src(cs.size());
    std::vector<double *> dst(cs.size());
    // this call overwrites the work vectors a prepared decomposition keeps its start vector in: the preparation is void
    for (lzx_ctx *c : cs) c->k_prep = 0;
    for (lzx_ctx *c : cs) {
        LZX_HIP(hipSetDevice(c->device));
        LZX_HIP(hipMemcpyAsync(c->d_io, x, sizeof(double) * c->n, hipMemcpyHostToDevice, c->stream));
        // x in hand-over layout (d_ybuf); the SpMV gathers from the exchange layout (d_xbuf; the same thing at one rank)
        LZX_TRY(lzx_launch_permute_in(c, c->d_io, c->d_ybuf, 1.0));
        if (multi) LZX_TRY(lzx_launch_relayout(c, c->d_ybuf, c->d_xbuf));
        SpmvLaunch l{multi ? c->d_xbuf : c->d_ybuf, c->d_ybuf + (size_t)c->rank * c->n_loc_pad, c->d_v, c->d_partials};
        LZX_TRY(lzx_launch_spmv(c, l));
    }
    if (multi) {
        for (size_t i = 0; i < cs.size(); ++i) { src[i] = cs[i]->d_v; dst[i] = cs[i]->d_ybuf; }
        LZX_TRY(lzx_comm_allgather(cs, src.data(), dst.data(), c0->n_loc_pad));
    }
    LZX_HIP(hipSetDevice(c0->device));
    LZX_TRY(lzx_launch_permute_out(c0, multi ? c0->d_ybuf : c0->d_v, c0->d_io));
    LZX_HIP(hipMemcpyAsync(y, c0->d_io, sizeof(double) * c0->n, hipMemcpyDeviceToHost, c0->stream));
    return sync_all(cs);
}

extern "C" int lzx_spmv_f64(lzx_handle h, const double *x, double *y)
{
    std::vector<lzx_ctx *> cs;
    LZX_TRY(gather_handles(h, cs));
    return spmv_run(cs, x, y);
}

extern "C" int lzx_spmv_f64_local(lzx_handle *hs, int world, const double *x, double *y)
{
    std::vector<lzx_ctx *> cs;
    LZX_TRY(gather_handles_local(hs, world, cs));
    return spmv_run(cs, x, y);
}

// --------------------------------------------------------------------------------------------------
static int multout_run(std::vector<lzx_ctx *> &cs, const double *t, u32 k, double *ans)
{
    if (!t || !ans || k == 0) LZX_FAIL(LZX_ERR_ARG, "lzx_multout_f64: bad argument");
    LZX_TRY(check_graphs(cs));
    lzx_ctx *c0 = cs[0];
    const bool multi = lzx_exchanges(c0);
    for (lzx_ctx *c : cs)
        if (c->k_last < k) LZX_FAIL(LZX_ERR_STATE, "lzx_multout_f64: the resident basis has %u vectors, %u asked", c->k_last, k);
    // (touches d_v, d_partials and d_ybuf only: all dead between two iterations of the loop, so a decomposition that is
    //  being advanced in chunks -- lzx_lanczos_run_steps -- stays resumable across this call)
    std::vector<const double *> src(cs.size());
    std::vector<double *> dst(cs.size());
    for (lzx_ctx *c : cs) {
        LZX_HIP(hipSetDevice(c->device));
        // the k coefficients are staged in the (idle) block-partials buffer; the unnormalised basis needs k more behind them
        if ((u64)k * (c->basis_u ? 2u : 1u) > c->np_cap)
            LZX_FAIL(LZX_ERR_LIMIT, "k = %u exceeds the staging capacity %u", k, c->np_cap / (c->basis_u ? 2u : 1u));
        LZX_HIP(hipMemcpyAsync(c->d_partials, t, sizeof(double) * k, hipMemcpyHostToDevice, c->stream));
        LZX_TRY(lzx_launch_multout(c, c->d_partials, k, c->d_v));
    }
    if (multi) {
        for (size_t i = 0; i < cs.size(); ++i) { src[i] = cs[i]->d_v; dst[i] = cs[i]->d_ybuf; }
        LZX_TRY(lzx_comm_allgather(cs, src.data(), dst.data(), c0->n_loc_pad));
    }
    LZX_HIP(hipSetDevice(c0->device));
    LZX_TRY(lzx_launch_permute_out(c0, multi ? c0->d_ybuf : c0->d_v, c0->d_io));
    LZX_HIP(hipMemcpyAsync(ans, c0->d_io, sizeof(double) * c0->n, hipMemcpyDeviceToHost, c0->stream));
    return sync_all(cs);
}

extern "C" int lzx_multout_f64(lzx_handle h, const double *t, uint32_t k, double *ans)
{
    std::vector<lzx_ctx *> cs;
    LZX_TRY(gather_handles(h, cs));
    return multout_run(cs, t, k, ans);
}

extern "C" int lzx_multout_f64_local(lzx_handle *hs, int world, const double *t, uint32_t k, double *ans)
{
    std::vector<lzx_ctx *> cs;
    LZX_TRY(gather_handles_local(hs, world, cs));
    return multout_run(cs, t, k, ans);
}

// Convergence monitor on the device: y = Q_k t is formed as in lzx_multout_f64 but stays in HBM; what comes back is
// ||y - y_prev|| / ||y|| against the answer of the previous call since the last prepare (1.0 for the first one).
static int multout_change_run(std::vector<lzx_ctx *> &cs, const double *t, u32 k, double *rel_change)
{
    if (!t || !rel_change || k == 0) LZX_FAIL(LZX_ERR_ARG, "lzx_multout_change_f64: bad argument");
    LZX_TRY(check_graphs(cs));
    lzx_ctx *c0 = cs[0];
    for (lzx_ctx *c : cs)
        if (c->k_last < k) LZX_FAIL(LZX_ERR_STATE, "lzx_multout_change_f64: the resident basis has %u vectors, %u asked", c->k_last, k);
    const bool have_prev = c0->ymon_valid > 0;
    for (lzx_ctx *c : cs) {
        LZX_HIP(hipSetDevice(c->device));
        for (double *&y : c->d_ymon)
            if (!y) LZX_HIP(hipMalloc(reinterpret_cast<void **>(&y), sizeof(double) * c->n_loc_pad));
        if ((u64)k * (c->basis_u ? 2u : 1u) > c->np_cap)
            LZX_FAIL(LZX_ERR_LIMIT, "k = %u exceeds the staging capacity %u", k, c->np_cap / (c->basis_u ? 2u : 1u));
        double *cur = c->d_ymon[c->ymon_valid & 1], *prev = c->d_ymon[(c->ymon_valid + 1) & 1];
        LZX_HIP(hipMemcpyAsync(c->d_partials, t, sizeof(double) * k, hipMemcpyHostToDevice, c->stream));
        LZX_TRY(lzx_launch_multout(c, c->d_partials, k, cur));
        LZX_TRY(lzx_launch_change(c, cur, have_prev ? prev : cur, c->d_scal + 5));
        ++c->ymon_valid;
    }
    if (lzx_exchanges(c0)) LZX_TRY(lzx_comm_allreduce_sum(cs, 5, 2));
    double two[2] = {0.0, 0.0};
    LZX_HIP(hipSetDevice(c0->device));
    LZX_HIP(hipMemcpyAsync(two, c0->d_scal + 5, sizeof two, hipMemcpyDeviceToHost, c0->stream));
    LZX_TRY(sync_all(cs));
    *rel_change = have_prev ? std::sqrt(two[0]) / std::sqrt(two[1]) : 1.0;
    return LZX_OK;
}

extern "C" int lzx_multout_change_f64(lzx_handle h, const double *t, uint32_t k, double *rel_change)
{
    std::vector<lzx_ctx *> cs;
    LZX_TRY(gather_handles(h, cs));
    return multout_change_run(cs, t, k, rel_change);
}

extern "C" int lzx_multout_change_f64_local(lzx_handle *hs, int world, const double *t, uint32_t k, double *rel_change)
{
    std::vector<lzx_ctx *> cs;
    LZX_TRY(gather_handles_local(hs, world, cs));
    return multout_change_run(cs, t, k, rel_change);
}

// --------------------------------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(256) k_stream_read(const double2 *p, u64 n16, double *out)
{
    // grid-interleaved, one 16-byte load in flight per lane, 32 wavefronts per CU
    const u64 nthreads = (u64)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += nthreads) {
        const double2 c = p[i];
        acc += c.x + c.y;
    }
    if (acc == 1.2345e-300) out[0] = acc;   // keeps the loads alive
}
// the other shape that tools/unit_bench.hip found at the top (6.5 TB/s): every workgroup streams its own contiguous
// share, four loads in flight per lane
__global__ void __launch_bounds__(1024) k_stream_read_chunk(const double2 *p, u64 n16, double *out)
{
    const u64 per = n16 / gridDim.x;
    const double2 *q = p + per * blockIdx.x;
    double acc = 0.0;
    u64 i = threadIdx.x;
    for (; i + 3 * 1024 < per; i += 4 * 1024) {
        const double2 a = q[i], b = q[i + 1024], c = q[i + 2048], d = q[i + 3072];
        acc += (a.x + a.y) + (b.x + b.y) + (c.x + c.y) + (d.x + d.y);
    }
    for (; i < per; i += 1024) acc += q[i].x + q[i].y;
    if (acc == 1.2345e-300) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_stream_copy(const double2 *src, double2 *dst, u64 n16)
{
    const u64 nthreads = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += nthreads) dst[i] = src[i];
}
}  // namespace

extern "C" int lzx_bench_stream(lzx_handle c, uint64_t bytes, uint32_t reps, double *read_gbs, double *copy_gbs)
{
    if (!c || bytes < 32 || reps == 0 || !read_gbs || !copy_gbs) LZX_FAIL(LZX_ERR_ARG, "lzx_bench_stream: bad argument");
    LZX_HIP(hipSetDevice(c->device));
    const u64 n16 = bytes / 16, half = n16 / 2;
    double2 *buf = nullptr;
    LZX_HIP(hipMalloc(reinterpret_cast<void **>(&buf), n16 * 16));
    hipError_t e = hipMemsetAsync(buf, 0x11, n16 * 16, c->stream);   // not zeros: the clock the chip holds depends on the data
    const u32 grid = (u32)c->cu_count * 8;
    float best_r = 1e30f, best_c = 1e30f;
    for (u32 r = 0; r < reps + 1 && e == hipSuccess; ++r) {   // first round warms up
        float ms = 0.f;
        e = hipEventRecord(c->ev_a, c->stream);
        hipLaunchKernelGGL(k_stream_read, dim3(grid), dim3(256), 0, c->stream, buf, n16, c->d_scal + 4);
        if (e == hipSuccess) e = hipEventRecord(c->ev_b, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(c->ev_b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev_a, c->ev_b);
        if (r > 0 && ms < best_r) best_r = ms;
        if (e == hipSuccess) e = hipEventRecord(c->ev_a, c->stream);
        hipLaunchKernelGGL(k_stream_read_chunk, dim3((u32)c->cu_count * 2), dim3(1024), 0, c->stream, buf, n16, c->d_scal + 4);
        if (e == hipSuccess) e = hipEventRecord(c->ev_b, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(c->ev_b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev_a, c->ev_b);
        if (r > 0 && ms < best_r) best_r = ms;   // the better of the two shapes
        // ... and the shape at the top of tools/unit_bench.hip's table (6.6 TB/s there): the same chunked kernel with four
        // (and eight) grids' worth of workgroups, so that the chip never drains while workgroups finish
        for (u32 mult : {4u, 8u}) {
            if (e == hipSuccess) e = hipEventRecord(c->ev_a, c->stream);
            hipLaunchKernelGGL(k_stream_read_chunk, dim3((u32)c->cu_count * mult), dim3(1024), 0, c->stream, buf, n16, c->d_scal + 4);
            if (e == hipSuccess) e = hipEventRecord(c->ev_b, c->stream);
            if (e == hipSuccess) e = hipEventSynchronize(c->ev_b);
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev_a, c->ev_b);
            if (r > 0 && ms < best_r) best_r = ms;
        }
        if (e == hipSuccess) e = hipEventRecord(c->ev_a, c->stream);
        hipLaunchKernelGGL(k_stream_copy, dim3(grid), dim3(256), 0, c->stream, buf, buf + half, half);
        if (e == hipSuccess) e = hipEventRecord(c->ev_b, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(c->ev_b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev_a, c->ev_b);
        if (r > 0 && ms < best_c) best_c = ms;
    }
    (void)hipFree(buf);
    LZX_HIP(e);
    *read_gbs = (double)(n16 * 16) / (best_r * 1e-3) / 1e9;
    *copy_gbs = (double)(half * 32) / (best_c * 1e-3) / 1e9;
    return LZX_OK;
}

#ifdef LZX_DEBUG_KNOBS
namespace {
// which XCD does workgroup b of a launch land on?  (HW_REG_XCC_ID; block -> XCD placement is round-robin from a start that
// HIP does not promise: MI355X_MICROARCH.md)  A probe for the two process states of DESIGN.md / NOTES 3.1 i.
__global__ void k_xcc_of_block(u32 *out)
{
    if (threadIdx.x == 0) {
        u32 x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        out[blockIdx.x] = x & 0xfu;
    }
}
}  // namespace
// debug library only: XCC ids of workgroups 0 .. 15 of `launches` consecutive 16-block launches on the handle's stream
extern "C" int lzx_dbg_xcc_map(lzx_handle c, uint32_t launches, uint32_t *out /* [launches][16] */)
{
    if (!c || !out || launches == 0 || launches > 64) LZX_FAIL(LZX_ERR_ARG, "lzx_dbg_xcc_map: bad argument");
    LZX_HIP(hipSetDevice(c->device));
    u32 *d = nullptr;
    LZX_HIP(hipMalloc(reinterpret_cast<void **>(&d), sizeof(u32) * 16 * launches));
    for (u32 l = 0; l < launches; ++l) hipLaunchKernelGGL(k_xcc_of_block, dim3(16), dim3(64), 0, c->stream, d + 16 * l);
    hipError_t e = hipMemcpyAsync(out, d, sizeof(u32) * 16 * launches, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    LZX_HIP(e);
    return LZX_OK;
}
#endif

extern "C" int lzx_bench_spmv(lzx_handle c, uint32_t reps, double *avg_ms, double *min_ms)
{
    if (!c || reps == 0 || !avg_ms) LZX_FAIL(LZX_ERR_ARG, "lzx_bench_spmv: bad argument");
    if (!c->d_row_ptr || !c->d_v) LZX_FAIL(LZX_ERR_STATE, "no graph has been handed over");
    c->k_prep = 0;   // uses the work vectors (see spmv_run)
    LZX_HIP(hipSetDevice(c->device));
    // a non-trivial resident input: x = 1 everywhere
    std::vector<double> ones(c->n, 1.0);
    LZX_HIP(hipMemcpyAsync(c->d_io, ones.data(), sizeof(double) * c->n, hipMemcpyHostToDevice, c->stream));
    LZX_TRY(lzx_launch_permute_in(c, c->d_io, c->d_ybuf, 1.0));
    if (lzx_exchanges(c)) LZX_TRY(lzx_launch_relayout(c, c->d_ybuf, c->d_xbuf));
    SpmvLaunch l{lzx_exchanges(c) ? c->d_xbuf : c->d_ybuf, c->d_ybuf + (size_t)c->rank * c->n_loc_pad, c->d_v, c->d_partials};
    LZX_TRY(lzx_launch_spmv(c, l));  // warm-up
    LZX_HIP(hipStreamSynchronize(c->stream));
    double total = 0.0, best = 1e300;
    for (u32 r = 0; r < reps; ++r) {
        LZX_HIP(hipEventRecord(c->ev_a, c->stream));
        LZX_TRY(lzx_launch_spmv(c, l));
        LZX_HIP(hipEventRecord(c->ev_b, c->stream));
        LZX_HIP(hipEventSynchronize(c->ev_b));
        float ms = 0.f;
        LZX_HIP(hipEventElapsedTime(&ms, c->ev_a, c->ev_b));
        total += ms;
        if (ms < best) best = ms;
    }
    *avg_ms = total / reps;
    if (min_ms) *min_ms = best;
#ifdef LZX_DEBUG_KNOBS
    if (const char *rt = getenv("LZX_RELOC_TEST")) {
        // Does the SpMV's time depend on WHERE its big streamed arrays lie?  Each array named in the variable (v: the value
        // stream, s: its slots, c: the reduced codes) is moved to a fresh allocation at a series of offsets and the SpMV timed
        // with everything else in place (the "fast" / "slow" process states of DESIGN.md 3.1 i).
        auto time_spmv = [&](double *avg, double *mn) -> int {
            double tot = 0.0, best = 1e300;
            for (int r = 0; r < 8; ++r) {
                LZX_HIP(hipEventRecord(c->ev_a, c->stream));
                LZX_TRY(lzx_launch_spmv(c, l));
                LZX_HIP(hipEventRecord(c->ev_b, c->stream));
                LZX_HIP(hipEventSynchronize(c->ev_b));
                float ms = 0.f;
                LZX_HIP(hipEventElapsedTime(&ms, c->ev_a, c->ev_b));
                tot += ms;
                if (ms < best) best = ms;
            }
            *avg = tot / 8;
            *mn = best;
            return LZX_OK;
        };
        const size_t offs[] = {0, 256, 4096, 65536, 1u << 20, (1u << 20) + 4096, 2u << 20, (3u << 20) + 65536, 16u << 20, (32u << 20) + 3 * 4096,
                               (5u << 20) + 512, 48u << 20};
        const size_t pad = 64u << 20;
        for (const char *w = rt; *w && c->pb && c->pb_values; ++w) {
            void **slot = nullptr;
            size_t bytes = 0;
            if (*w == 'v') { slot = reinterpret_cast<void **>(&c->d_pb_val); bytes = sizeof(double) * (c->pb_values + 8); }
            else if (*w == 's') { slot = reinterpret_cast<void **>(&c->d_pb_lrow); bytes = sizeof(uint16_t) * (c->pb_values + 8); }
            else if (*w == 'c' && c->pbr_steps) { slot = reinterpret_cast<void **>(&c->d_pbr_code); bytes = sizeof(uint4) * ((size_t)c->pbr_steps * 64 + 1); }
            if (!slot || !*slot) continue;
            char *buf = nullptr;
            LZX_HIP(hipMalloc(&buf, bytes + pad));
            void *orig = *slot;
            double a0 = 0, m0 = 0;
            LZX_TRY(time_spmv(&a0, &m0));
            fprintf(stderr, "[lzx reloc] %c at %p (%zu MB): spmv avg %.4f min %.4f ms; moved to %p + offset:\n", *w, orig, bytes >> 20, a0, m0, (void *)buf);
            for (size_t off : offs) {
                LZX_HIP(hipMemcpyAsync(buf + off, orig, bytes, hipMemcpyDeviceToDevice, c->stream));
                *slot = buf + off;
                double a = 0, m = 0;
                int rc = time_spmv(&a, &m);
                *slot = orig;
                LZX_TRY(rc);
                fprintf(stderr, "[lzx reloc]   %c + %9zu: avg %.4f min %.4f ms\n", *w, off, a, m);
            }
            LZX_HIP(hipStreamSynchronize(c->stream));
            (void)hipFree(buf);
        }
    }
    if (getenv("LZX_RELOC_SHOP")) {
        // Which array's PLACEMENT carries the SpMV's "state" (profiles/NOTES.md: engines built one after the other in one
        // process differ by 1.5-5 % on C3, 4-20 % on ER, each keeping its time for life)?  Every large array in turn is copied
        // into several fresh allocations -- all kept alive, so they are physically distinct -- and the SpMV timed with the
        // array in each of them, everything else in place.
        auto time_spmv = [&](double *avg, double *mn) -> int {
            double tot = 0.0, best = 1e300;
            for (int r = 0; r < 8; ++r) {
                LZX_HIP(hipEventRecord(c->ev_a, c->stream));
                LZX_TRY(lzx_launch_spmv(c, l));
                LZX_HIP(hipEventRecord(c->ev_b, c->stream));
                LZX_HIP(hipEventSynchronize(c->ev_b));
                float ms = 0.f;
                LZX_HIP(hipEventElapsedTime(&ms, c->ev_a, c->ev_b));
                tot += ms;
                if (ms < best) best = ms;
            }
            *avg = tot / 8;
            *mn = best;
            return LZX_OK;
        };
        struct Named { const char *name; void **slot; };
        Named arrays[] = {{"pb_val", (void **)&c->d_pb_val}, {"pb_lrow", (void **)&c->d_pb_lrow}, {"pbr_code", (void **)&c->d_pbr_code},
                          {"pb_lcol", (void **)&c->d_pb_lcol}, {"pb_dst", (void **)&c->d_pb_dst}, {"sell_cols", (void **)&c->d_sell_cols},
                          {"long_cols", (void **)&c->d_long_cols}, {"pbr_base", (void **)&c->d_pbr_base}, {"v", (void **)&c->d_v}};
        double a0 = 0, m0 = 0;
        LZX_TRY(time_spmv(&a0, &m0));
        fprintf(stderr, "[lzx shop] as built: spmv avg %.4f min %.4f ms\n", a0, m0);
        for (const Named &nm : arrays) {
            if (!*nm.slot) continue;
            size_t bytes = 0;
            if (hipMemPtrGetInfo(*nm.slot, &bytes) != hipSuccess || bytes < (8u << 20)) { (void)hipGetLastError(); continue; }
            void *orig = *nm.slot;
            void *fresh[5] = {};
            fprintf(stderr, "[lzx shop] %s (%zu MB at %p):", nm.name, bytes >> 20, orig);
            for (int t = 0; t < 5; ++t) {
                if (hipMalloc(&fresh[t], bytes) != hipSuccess) { (void)hipGetLastError(); fresh[t] = nullptr; break; }
                LZX_HIP(hipMemcpyAsync(fresh[t], orig, bytes, hipMemcpyDeviceToDevice, c->stream));
                *nm.slot = fresh[t];
                if (nm.slot == (void **)&c->d_v) l.v = c->d_v;
                double a = 0, m = 0;
                int rc = time_spmv(&a, &m);
                *nm.slot = orig;
                if (nm.slot == (void **)&c->d_v) l.v = c->d_v;
                LZX_TRY(rc);
                fprintf(stderr, " %.4f", m);
                if (nm.slot == (void **)&c->d_pb_val && c->pb) {   // which pass feels the placement: the gather pass alone, the scatter pass alone
                    const int64_t keep = c->phase_mask_opt;
                    double ag = 0, mg = 0, as = 0, msc = 0;
                    *nm.slot = fresh[t];
                    c->phase_mask_opt = keep | 4;
                    rc = time_spmv(&ag, &mg);
                    c->phase_mask_opt = keep | 8;
                    if (rc == LZX_OK) rc = time_spmv(&as, &msc);
                    c->phase_mask_opt = keep;
                    *nm.slot = orig;
                    LZX_TRY(rc);
                    fprintf(stderr, " (gather alone %.4f, scatter alone %.4f)", mg, msc);
                }
            }
            fprintf(stderr, " ms (min of 8 each)\n");
            LZX_HIP(hipStreamSynchronize(c->stream));
            for (void *f : fresh) if (f) (void)hipFree(f);
        }
    }
    if (getenv("LZX_TRACE_SPMV")) {
        // one more SpMV with marks between its kernels: hub/body, split-row finish, scatter, gather (+finish)
        for (auto &ev : c->trace_ev)
            if (!ev) LZX_HIP(hipEventCreate(&ev));
        float t[4] = {0, 0, 0, 0};
        const bool scatter_first = c->pb && c->pb_order_opt != 0 && !(c->side_opt > 0);
        constexpr int TR = 8;   // traced SpMVs, averaged (the marks cost a few microseconds each)
        for (int rep = 0; rep < TR; ++rep) {
            c->trace = true;
            for (auto &ev : c->trace_ev) LZX_HIP(hipEventRecord(ev, c->stream));
            int rc = lzx_launch_spmv(c, l);
            c->trace = false;
            LZX_TRY(rc);
            LZX_HIP(hipStreamSynchronize(c->stream));
            float d[4] = {0, 0, 0, 0};
            (void)hipEventElapsedTime(&d[0], c->trace_ev[0], c->trace_ev[1]);
            if (scatter_first) {
                (void)hipEventElapsedTime(&d[2], c->trace_ev[2], c->trace_ev[3]);
                (void)hipEventElapsedTime(&d[3], c->trace_ev[5], c->trace_ev[4]);
            } else {
                (void)hipEventElapsedTime(&d[1], c->trace_ev[1], c->trace_ev[2]);
                if (c->pb) {
                    (void)hipEventElapsedTime(&d[2], c->trace_ev[2], c->trace_ev[3]);
                    (void)hipEventElapsedTime(&d[3], c->trace_ev[3], c->trace_ev[4]);
                }
            }
            for (int q = 0; q < 4; ++q) t[q] += d[q] / TR;
        }
        fprintf(stderr, "[lzx trace] val %p code %p lslot %p sell %p v %p xbuf %p ybuf %p\n", (void *)c->d_pb_val, (void *)c->d_pbr_code,
                (void *)c->d_pb_lrow, (void *)c->d_sell_cols, (void *)c->d_v, (void *)c->d_xbuf, (void *)c->d_ybuf);
        fprintf(stderr, "[lzx trace] k_spmv %.4f ms  long_finish %.4f ms  pb_scatter %.4f ms  pb_gather(+finish) %.4f ms  (mean of %d)\n",
                t[0], t[1], t[2], t[3], TR);
        if (c->d_pb_gstamps) {
            const u32 G = c->pb_gather_grid;
            std::vector<unsigned long long> h(8 * (size_t)G);
            LZX_HIP(hipMemcpy(h.data(), c->d_pb_gstamps, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull;
            for (u32 w = 0; w < G; ++w) if (h[8 * w + 1]) t0 = std::min(t0, h[8 * w]);
            std::vector<double> en, zr, sm, br, fd, rate, clk;
            double items = 0, vals = 0;
            for (u32 w = 0; w < G; ++w) {
                const unsigned long long *r = &h[8 * (size_t)w];
                if (!r[1]) continue;
                en.push_back((r[1] - t0) * 0.01); zr.push_back(r[2] * 0.01); sm.push_back(r[3] * 0.01); br.push_back(r[4] * 0.01); fd.push_back(r[5] * 0.01);
                items += (double)(r[6] & 0xffffull); vals += (double)r[7];
                if (r[1] > r[0]) clk.push_back((double)(r[6] >> 16) / ((double)(r[1] - r[0]) * 10.0));   // shader cycles per ns = GHz
                if (r[3]) rate.push_back((double)r[7] * 10.0 / (r[3] * 0.01) * 1e-3);   // GB/s while streaming (wavefront 0's clock; group items: its own band)
            }
            auto srt = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); };
            srt(en); srt(zr); srt(sm); srt(br); srt(fd); srt(rate); srt(clk);
            auto pct = [](const std::vector<double> &v, double p) { return v.empty() ? 0.0 : v[(size_t)(p * (v.size() - 1))]; };
            fprintf(stderr, "[lzx gstamps] gather: %zu workgroups, %.0f items, %.0f values | end us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f\n", en.size(), items, vals,
                    en.front(), pct(en, 0.1), pct(en, 0.5), pct(en, 0.9), en.back());
            {   // the workgroups that end last: what did they do?
                std::vector<u32> idx;
                for (u32 w = 0; w < G; ++w) if (h[8 * (size_t)w + 1]) idx.push_back(w);
                std::sort(idx.begin(), idx.end(), [&](u32 a, u32 b) { return h[8 * (size_t)a + 1] > h[8 * (size_t)b + 1]; });
                for (size_t q = 0; q < idx.size(); q += (q < 8 ? 1 : idx.size() / 8)) {
                    const unsigned long long *r = &h[8 * (size_t)idx[q]];
                    fprintf(stderr, "[lzx gstamps]   rank %zu: workgroup %u start %.1f end %.1f us | zero %.1f stream %.1f barrier %.1f fold %.1f | items %llu values %llu\n", q, idx[q],
                            (r[0] - t0) * 0.01, (r[1] - t0) * 0.01, r[2] * 0.01, r[3] * 0.01, r[4] * 0.01, r[5] * 0.01, r[6] & 0xffffull, r[7]);
                }
            }
            fprintf(stderr, "[lzx gstamps]   in-kernel shader clock (s_memtime / s_memrealtime), GHz: p10 %.3f p50 %.3f p90 %.3f\n", pct(clk, 0.1), pct(clk, 0.5), pct(clk, 0.9));
            fprintf(stderr, "[lzx gstamps]   wavefront 0, us per workgroup (p10 / p50 / p90): records+zeroing %.1f / %.1f / %.1f | streaming %.1f / %.1f / %.1f | barrier before fold %.1f / %.1f / %.1f | fold %.1f / %.1f / %.1f | GB/s per workgroup while streaming p50 %.1f\n",
                    pct(zr, 0.1), pct(zr, 0.5), pct(zr, 0.9), pct(sm, 0.1), pct(sm, 0.5), pct(sm, 0.9), pct(br, 0.1), pct(br, 0.5), pct(br, 0.9),
                    pct(fd, 0.1), pct(fd, 0.5), pct(fd, 0.9), pct(rate, 0.5));
        }
    }
#endif
    return LZX_OK;
}

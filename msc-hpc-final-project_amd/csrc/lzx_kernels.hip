// lzx_kernels.hip -- the per-iteration device kernels of the Lanczos loop, written for gfx950
// (64-wide wavefronts, 160 KiB LDS per CU, 256 CUs in 8 XCDs).  No MFMA: every kernel here is
// bandwidth / gather bound.
//
// One Lanczos iteration (serial/lib/lanczos.cc:21-53; parallel-final/lib/cu_lanczos.cu:97-128 runs
// it as 9 launches) is three launches here:
//   k_spmv        v = A q_j  and per-workgroup partials of alpha_j = v . q_j     (cu_spMV1 + cu_dot_prod)
//   k_axpy_norm   alpha_j = sum(partials); v -= alpha_j q_j; v -= beta_{j-1} q_{j-1};
//                 per-workgroup partials of ||v||^2                (cu_reduce + 2x cu_dpax + cu_norm_sq)
//   k_scale       beta_j = sqrt(sum(partials)); q_{j+1} = v / beta_j          (cu_reduce_sqrt + cu_dvexda)
// (+ k_long_finish when the graph has split rows; on graphs whose x exceeds the caches k_spmv keeps only
// the LDS-staged hub columns and lzx_pb.hip's two streaming passes add the rest into v).  The grid-wide
// reductions are closed in the PROLOGUE of the next kernel: every workgroup sums the same partials in the
// same fixed order, so all of them hold bit-identical alpha / beta without global atomics, a separate
// reduce launch or a grid barrier, and results are reproducible run to run.
#include "lzx_internal.h"

// --------------------------------------------------------------------------------------------------
// reductions: fixed shape, no atomics
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;  // every lane holds the same total
}

// Sum p[0..np) identically in every workgroup of a 256-thread launch. sh: >= 4 doubles of LDS.
__device__ __forceinline__ double block_sum_fixed_256(const double *p, u32 np, double *sh)
{
    double s = 0.0;
    for (u32 i = threadIdx.x; i < np; i += LZX_VEC_BLOCK) s += p[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    const double t = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    __syncthreads();
    return t;
}

// --------------------------------------------------------------------------------------------------
// SpMV on the sliced-ELL body + split rows, fused with the alpha partial.
struct SpmvArgs {
    const u32 *sell_cols;
    const u64 *slice_off;
    const u32 *slice_w;
    u32 n_slices;
    u32 row0;  // first local row of the body (= rows handled as split rows)
    const u32 *long_cols;
    const u64 *item_beg;
    const u32 *item_len;
    u32 n_items;
    double *long_partial;
    const double *x;
    const double *q_loc;
    double *v;
    double *partials;
    u32 hub;       // LDS slots (staged values + zero slots)
    u32 hub_real;  // slots that carry x values
    u32 world;
    u32 xs0;       // slice stride of chunk 0 of the exchange layout (the staged hub entries all live there)
};

// Column code c: c < hub -> value staged in LDS slot c; otherwise x[c - hub].
template <bool HUB>
__device__ __forceinline__ double gather(u32 c, const double *__restrict__ x, const double *hubv, u32 hub)
{
    if (HUB) {
        if (c < hub) return hubv[c];
        return x[c - hub];
    }
    return x[c];
}

template <bool NT>
__device__ __forceinline__ uint4 load_idx4(const uint4 *p)
{
    if (NT) {
        uint4 r;
        r.x = __builtin_nontemporal_load(&p->x);
        r.y = __builtin_nontemporal_load(&p->y);
        r.z = __builtin_nontemporal_load(&p->z);
        r.w = __builtin_nontemporal_load(&p->w);
        return r;
    }
    return *p;
}

template <bool HUB, bool NT>
__global__ void __launch_bounds__(LZX_SPMV_BLOCK) k_spmv(const SpmvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *hubv = lds;
    double *wsum = lds + a.hub;  // 16 doubles behind the staged entries

    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    if (HUB) {
        // Stage x of the `hub` highest-degree vertices once per workgroup (coalesced at world == 1;
        // `world` strided segments otherwise: degree rank r lives at (r % world) * xs0 + r / world, in chunk 0).
        for (u32 i = tid; i < a.hub; i += LZX_SPMV_BLOCK) {
            const u32 g = (a.world == 1) ? i : (i % a.world) * a.xs0 + i / a.world;
            hubv[i] = i < a.hub_real ? a.x[g] : 0.0;
        }
        __syncthreads();
    }

    const u32 waves = gridDim.x * (LZX_SPMV_BLOCK / 64);
    const u32 w0 = blockIdx.x * (LZX_SPMV_BLOCK / 64) + wv;

    // Both loops below are three-stage software pipelines over the wavefront's units (unit i = w0 + i * waves):
    // while unit i is summed, the index packets of unit i+1 are in flight and the descriptor of unit i+2 is being
    // fetched.  Without it every unit pays two dependent memory round trips (descriptor, then packets) and, after
    // the hub split leaves most units only a few packets long, the kernel is bound by those latencies, not by
    // bandwidth (measured: 0.25 ms for 0.63 GB of indices on the 10 M-vertex graph).
    // Units are dealt to wavefronts round-robin: neighbouring wavefronts stream neighbouring memory, and because
    // widths fall monotonically (and are capped by the split-row threshold) every wavefront gets the same work.

    // ---- split rows: one wavefront sums one item of <= LZX_ITEM entries, lanes striding 16-byte index
    //      packets; the item totals are combined in row order by k_long_finish.
    {
        u32 it = w0;
        bool h0 = it < a.n_items, h1 = it + waves < a.n_items;
        u64 beg0 = 0, beg1 = 0;
        u32 len0 = 0, len1 = 0;
        if (h0) { beg0 = a.item_beg[it]; len0 = a.item_len[it]; }
        if (h1) { beg1 = a.item_beg[it + waves]; len1 = a.item_len[it + waves]; }
        uint4 pk0 = make_uint4(0, 0, 0, 0), pk1 = make_uint4(0, 0, 0, 0);
        if (h0) {
            const uint4 *p = reinterpret_cast<const uint4 *>(a.long_cols + beg0);
            const u32 packets = len0 >> 2;
            if (lane < packets) pk0 = load_idx4<NT>(p + lane);
            if (lane + 64 < packets) pk1 = load_idx4<NT>(p + lane + 64);
        }
        while (h0) {
            // stage A: descriptor of item i+2
            const bool h2 = it + 2 * waves < a.n_items;
            u64 beg2 = 0;
            u32 len2 = 0;
            if (h2) { beg2 = a.item_beg[it + 2 * waves]; len2 = a.item_len[it + 2 * waves]; }
            // stage B: first two packets per lane of item i+1
            uint4 nk0 = make_uint4(0, 0, 0, 0), nk1 = make_uint4(0, 0, 0, 0);
            if (h1) {
                const uint4 *pn = reinterpret_cast<const uint4 *>(a.long_cols + beg1);
                const u32 pkn = len1 >> 2;
                if (lane < pkn) nk0 = load_idx4<NT>(pn + lane);
                if (lane + 64 < pkn) nk1 = load_idx4<NT>(pn + lane + 64);
            }
            // stage C: sum item i
            const uint4 *p = reinterpret_cast<const uint4 *>(a.long_cols + beg0);
            const u32 packets = len0 >> 2;
            double acc = 0.0;
            if (lane < packets) {
                const double x0 = gather<HUB>(pk0.x, a.x, hubv, a.hub);
                const double x1 = gather<HUB>(pk0.y, a.x, hubv, a.hub);
                const double x2 = gather<HUB>(pk0.z, a.x, hubv, a.hub);
                const double x3 = gather<HUB>(pk0.w, a.x, hubv, a.hub);
                acc += x0; acc += x1; acc += x2; acc += x3;
            }
            if (lane + 64 < packets) {
                const double x0 = gather<HUB>(pk1.x, a.x, hubv, a.hub);
                const double x1 = gather<HUB>(pk1.y, a.x, hubv, a.hub);
                const double x2 = gather<HUB>(pk1.z, a.x, hubv, a.hub);
                const double x3 = gather<HUB>(pk1.w, a.x, hubv, a.hub);
                acc += x0; acc += x1; acc += x2; acc += x3;
            }
            u32 q = lane + 128;
            for (; q + 64 < packets; q += 128) {
                const uint4 c = load_idx4<NT>(p + q), e = load_idx4<NT>(p + q + 64);
                const double x0 = gather<HUB>(c.x, a.x, hubv, a.hub);
                const double x1 = gather<HUB>(c.y, a.x, hubv, a.hub);
                const double x2 = gather<HUB>(c.z, a.x, hubv, a.hub);
                const double x3 = gather<HUB>(c.w, a.x, hubv, a.hub);
                const double x4 = gather<HUB>(e.x, a.x, hubv, a.hub);
                const double x5 = gather<HUB>(e.y, a.x, hubv, a.hub);
                const double x6 = gather<HUB>(e.z, a.x, hubv, a.hub);
                const double x7 = gather<HUB>(e.w, a.x, hubv, a.hub);
                acc += x0; acc += x1; acc += x2; acc += x3;
                acc += x4; acc += x5; acc += x6; acc += x7;
            }
            for (; q < packets; q += 64) {
                const uint4 c = load_idx4<NT>(p + q);
                const double x0 = gather<HUB>(c.x, a.x, hubv, a.hub);
                const double x1 = gather<HUB>(c.y, a.x, hubv, a.hub);
                const double x2 = gather<HUB>(c.z, a.x, hubv, a.hub);
                const double x3 = gather<HUB>(c.w, a.x, hubv, a.hub);
                acc += x0; acc += x1; acc += x2; acc += x3;
            }
            acc = wave_sum(acc);
            if (lane == 0) a.long_partial[it] = acc;
            // rotate
            it += waves;
            h0 = h1; h1 = h2;
            beg0 = beg1; len0 = len1;
            beg1 = beg2; len1 = len2;
            pk0 = nk0; pk1 = nk1;
        }
    }

    // ---- sliced-ELL body: one wavefront per slice of 64 rows, lane = row.  Each lane adds its row's
    //      entries one at a time in the caller's column order: the same left-to-right sum as the
    //      reference's spMV (serial/lib/SPMV.cc:24-27), so these rows come out bit-identical to it.
    double dot = 0.0;
    {
        u32 s = w0;
        bool h0 = s < a.n_slices, h1 = s + waves < a.n_slices;
        u64 off0 = 0, off1 = 0;
        u32 st0 = 0, st1 = 0;   // packets per lane
        if (h0) { off0 = a.slice_off[s]; st0 = a.slice_w[s] >> 2; }
        if (h1) { off1 = a.slice_off[s + waves]; st1 = a.slice_w[s + waves] >> 2; }
        uint4 pf[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) pf[u] = make_uint4(0, 0, 0, 0);
        if (h0) {
            const uint4 *p = reinterpret_cast<const uint4 *>(a.sell_cols + off0) + lane;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if ((u32)u < st0) pf[u] = load_idx4<NT>(p + (size_t)u * 64);
        }
        while (h0) {
            // stage A: descriptor of slice i+2
            const bool h2 = s + 2 * waves < a.n_slices;
            u64 off2 = 0;
            u32 st2 = 0;
            if (h2) { off2 = a.slice_off[s + 2 * waves]; st2 = a.slice_w[s + 2 * waves] >> 2; }
            // stage B: first four packets of slice i+1
            uint4 nf[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) nf[u] = make_uint4(0, 0, 0, 0);
            if (h1) {
                const uint4 *pn = reinterpret_cast<const uint4 *>(a.sell_cols + off1) + lane;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if ((u32)u < st1) nf[u] = load_idx4<NT>(pn + (size_t)u * 64);
            }
            // stage C: sum slice i
            const uint4 *p = reinterpret_cast<const uint4 *>(a.sell_cols + off0) + lane;
            const u32 row = a.row0 + s * 64 + lane;
            const double qrow = a.q_loc[row];
            double acc = 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if ((u32)u < st0) {
                    const double x0 = gather<HUB>(pf[u].x, a.x, hubv, a.hub);
                    const double x1 = gather<HUB>(pf[u].y, a.x, hubv, a.hub);
                    const double x2 = gather<HUB>(pf[u].z, a.x, hubv, a.hub);
                    const double x3 = gather<HUB>(pf[u].w, a.x, hubv, a.hub);
                    acc += x0; acc += x1; acc += x2; acc += x3;   // left to right: the reference's order
                }
            }
            u32 i = 4;
            for (; i + 4 <= st0; i += 4) {
                uint4 c[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) c[u] = load_idx4<NT>(p + (size_t)(i + u) * 64);
                double xv[16];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    xv[4 * u + 0] = gather<HUB>(c[u].x, a.x, hubv, a.hub);
                    xv[4 * u + 1] = gather<HUB>(c[u].y, a.x, hubv, a.hub);
                    xv[4 * u + 2] = gather<HUB>(c[u].z, a.x, hubv, a.hub);
                    xv[4 * u + 3] = gather<HUB>(c[u].w, a.x, hubv, a.hub);
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += xv[u];
            }
            for (; i < st0; ++i) {
                const uint4 c0 = load_idx4<NT>(p + (size_t)i * 64);
                const double x0 = gather<HUB>(c0.x, a.x, hubv, a.hub);
                const double x1 = gather<HUB>(c0.y, a.x, hubv, a.hub);
                const double x2 = gather<HUB>(c0.z, a.x, hubv, a.hub);
                const double x3 = gather<HUB>(c0.w, a.x, hubv, a.hub);
                acc += x0; acc += x1; acc += x2; acc += x3;
            }
            a.v[row] = acc;
            dot += acc * qrow;
            // rotate
            s += waves;
            h0 = h1; h1 = h2;
            off0 = off1; st0 = st1;
            off1 = off2; st1 = st2;
#pragma unroll
            for (int u = 0; u < 4; ++u) pf[u] = nf[u];
        }
    }

    dot = wave_sum(dot);
    if (lane == 0) wsum[wv] = dot;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (u32 i = 0; i < LZX_SPMV_BLOCK / 64; ++i) s += wsum[i];
        a.partials[blockIdx.x] = s;
    }
}

// Split rows: v[r] = sum of the row's item totals in item order; alpha partials for those rows.
__global__ void __launch_bounds__(LZX_VEC_BLOCK)
k_long_finish(const u32 *item_first, const double *long_partial, u32 n_long, const double *q_loc,
              double *v, double *partials)
{
    __shared__ double sh[4];
    const u32 r = blockIdx.x * LZX_VEC_BLOCK + threadIdx.x;
    double dot = 0.0;
    if (r < n_long) {
        double s = 0.0;
        for (u32 it = item_first[r]; it < item_first[r + 1]; ++it) s += long_partial[it];
        v[r] = s;
        dot = s * q_loc[r];
    }
    dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

// --------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(LZX_VEC_BLOCK)
k_reduce(const double *partials, u32 np, double *out, int do_sqrt)
{
    __shared__ double sh[4];
    const double t = block_sum_fixed_256(partials, np, sh);
    if (threadIdx.x == 0) *out = do_sqrt ? sqrt(t) : t;
}

// serial/lib/lanczos.cc:26-37: alpha_j closes here; v -= alpha_j q_j; then v -= beta_{j-1} q_{j-1}
// (two separately rounded updates, multiply then subtract, as the reference's generic x86-64 build does).
__global__ void __launch_bounds__(LZX_VEC_BLOCK)
k_axpy_norm(double *v, const double *__restrict__ qj, const double *__restrict__ qjm1,
            const double *partials_in, u32 np_in, double *alpha_out, const double *beta_prev,
            double *partials_out, u32 n)
{
    __shared__ double sh[4];
    const double alpha = block_sum_fixed_256(partials_in, np_in, sh);
    if (blockIdx.x == 0 && threadIdx.x == 0) *alpha_out = alpha;
    const double beta = qjm1 ? *beta_prev : 0.0;

    double nrm = 0.0;
    const u32 stride = gridDim.x * LZX_VEC_BLOCK * 2;
    for (u32 i = (blockIdx.x * LZX_VEC_BLOCK + threadIdx.x) * 2; i < n; i += stride) {
        double2 w = *reinterpret_cast<const double2 *>(v + i);
        const double2 q = *reinterpret_cast<const double2 *>(qj + i);
        w.x -= alpha * q.x;
        w.y -= alpha * q.y;
        if (qjm1) {
            const double2 p = *reinterpret_cast<const double2 *>(qjm1 + i);
            w.x -= beta * p.x;
            w.y -= beta * p.y;
        }
        *reinterpret_cast<double2 *>(v + i) = w;
        nrm += w.x * w.x;
        nrm += w.y * w.y;
    }
    nrm = wave_sum(nrm);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = nrm;
    __syncthreads();
    if (threadIdx.x == 0) partials_out[blockIdx.x] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

// serial/lib/lanczos.cc:39-44: beta_j = sqrt(sum v^2); q_{j+1} = v / beta_j (a true division).
__global__ void __launch_bounds__(LZX_VEC_BLOCK)
k_scale(const double *__restrict__ v, double *q_next, const double *partials_in, u32 np_in,
        double *beta_out, u32 n)
{
    __shared__ double sh[4];
    const double beta = sqrt(block_sum_fixed_256(partials_in, np_in, sh));
    if (blockIdx.x == 0 && threadIdx.x == 0) *beta_out = beta;
    const u32 stride = gridDim.x * LZX_VEC_BLOCK * 2;
    for (u32 i = (blockIdx.x * LZX_VEC_BLOCK + threadIdx.x) * 2; i < n; i += stride) {
        double2 w = *reinterpret_cast<const double2 *>(v + i);
        w.x /= beta;
        w.y /= beta;
        *reinterpret_cast<double2 *>(q_next + i) = w;
    }
}

// caller's order -> internal full-length layout (and back); div = ||x|| for the start vector.
__global__ void k_permute_in(const double *io, const u32 *gidx, double *full, double div, u64 n)
{
    const u64 o = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (o < n) full[gidx[o]] = io[o] / div;
}

// hand-over layout [world][n_loc_pad] -> exchange layout [world][xs]: the first xs entries of every slice
__global__ void k_relayout(const double *io_layout, double *x, u32 world, u32 n_loc_pad, u32 xs, u32 xs0)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (u64)world * xs) return;
    u32 p, l;
    if (i < (u64)world * xs0) {
        p = (u32)(i / xs0);
        l = (u32)(i % xs0);
    } else {
        const u64 j = i - (u64)world * xs0;
        p = (u32)(j / (xs - xs0));
        l = xs0 + (u32)(j % (xs - xs0));
    }
    x[i] = io_layout[(size_t)p * n_loc_pad + l];
}

__global__ void k_permute_out(const double *full, const u32 *gidx, double *io, u64 n)
{
    const u64 o = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (o < n) io[o] = full[gidx[o]];
}

// ans_loc = Q_loc t: the second dgemv of multOut (parallel-final/lib/multiplyOut.cu:42-44), on the
// device-resident basis; one thread per row, basis vectors streamed with unit stride.
__global__ void __launch_bounds__(LZX_VEC_BLOCK)
k_multout(const double *Q, u32 ldq, const double *t, u32 k, double *out, u32 n)
{
    const u32 i = blockIdx.x * LZX_VEC_BLOCK + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (u32 j = 0; j < k; ++j) s += Q[(size_t)j * ldq + i] * t[j];
    out[i] = s;
}

// --------------------------------------------------------------------------------------------------
// launch wrappers
static u32 vec_grid(const lzx_ctx *c)
{
    const u32 need = (c->n_loc_pad / 2 + LZX_VEC_BLOCK - 1) / LZX_VEC_BLOCK;
    const u32 cap = (u32)c->cu_count * 8;
    return need < 1 ? 1 : (need < cap ? need : cap);
}

u32 lzx_spmv_partials(const lzx_ctx *c) { return c->spmv_grid + c->fin_grid + lzx_pb_partials(c); }

template <bool HUB, bool NT>
static int launch_spmv_t(lzx_ctx *c, const SpmvArgs &a)
{
    auto kern = k_spmv<HUB, NT>;
    LZX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->spmv_lds));
    hipLaunchKernelGGL(kern, dim3(c->spmv_grid), dim3(LZX_SPMV_BLOCK), c->spmv_lds, c->stream, a);
    return LZX_OK;
}

int lzx_launch_spmv(lzx_ctx *c, const SpmvLaunch &l)
{
    SpmvArgs a;
    a.sell_cols = c->d_sell_cols;
    a.slice_off = c->d_slice_off;
    a.slice_w = c->d_slice_w;
    a.n_slices = (c->phase_mask_opt & 2) ? c->n_slices : 0;
    a.row0 = c->n_long64;
    a.long_cols = c->d_long_cols;
    a.item_beg = c->d_item_beg;
    a.item_len = c->d_item_len;
    a.n_items = (c->phase_mask_opt & 1) ? c->n_items : 0;
    a.long_partial = c->d_long_partial;
    a.x = l.x;
    a.q_loc = l.q_loc;
    a.v = l.v;
    a.partials = l.partials;
    a.hub = c->hub;
    a.hub_real = c->hub_real;
    a.world = (u32)c->world;
    a.xs0 = c->xs0;
    const bool nt = c->nt_opt > 0;
    if (c->trace) LZX_HIP(hipEventRecord(c->trace_ev[0], c->stream));
    if (c->hub > 0) {
        if (nt) LZX_TRY((launch_spmv_t<true, true>(c, a)));
        else    LZX_TRY((launch_spmv_t<true, false>(c, a)));
    } else {
        if (nt) LZX_TRY((launch_spmv_t<false, true>(c, a)));
        else    LZX_TRY((launch_spmv_t<false, false>(c, a)));
    }
    if (c->trace) LZX_HIP(hipEventRecord(c->trace_ev[1], c->stream));
    if (c->fin_grid > 0) {
        hipLaunchKernelGGL(k_long_finish, dim3(c->fin_grid), dim3(LZX_VEC_BLOCK), 0, c->stream,
                           c->d_item_first, c->d_long_partial, c->n_long64, l.q_loc, l.v,
                           l.partials + c->spmv_grid);
    }
    LZX_HIP(hipGetLastError());
    if (c->trace) LZX_HIP(hipEventRecord(c->trace_ev[2], c->stream));
    // entries whose column is not staged in LDS: two streaming passes that add into v (lzx_pb.hip)
    LZX_TRY(lzx_pb_launch(c, l.x, l.q_loc, l.v, l.partials + c->spmv_grid + c->fin_grid, l.chunk1_ready));
    if (c->trace) LZX_HIP(hipEventRecord(c->trace_ev[4], c->stream));
    return LZX_OK;
}

int lzx_launch_reduce(lzx_ctx *c, const double *partials, u32 np, double *out, int do_sqrt)
{
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(LZX_VEC_BLOCK), 0, c->stream, partials, np, out, do_sqrt);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

int lzx_launch_axpy_norm(lzx_ctx *c, double *v, const double *qj, const double *qjm1,
                         const double *partials_in, u32 np_in, double *alpha_out,
                         const double *beta_prev, double *partials_out, u32 *np_out)
{
    const u32 g = vec_grid(c);
    hipLaunchKernelGGL(k_axpy_norm, dim3(g), dim3(LZX_VEC_BLOCK), 0, c->stream, v, qj, qjm1,
                       partials_in, np_in, alpha_out, beta_prev, partials_out, c->n_loc_pad);
    LZX_HIP(hipGetLastError());
    *np_out = g;
    return LZX_OK;
}

int lzx_launch_scale(lzx_ctx *c, const double *v, double *q_next, const double *partials_in,
                     u32 np_in, double *beta_out)
{
    hipLaunchKernelGGL(k_scale, dim3(vec_grid(c)), dim3(LZX_VEC_BLOCK), 0, c->stream, v, q_next,
                       partials_in, np_in, beta_out, c->n_loc_pad);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

int lzx_launch_permute_in(lzx_ctx *c, const double *io, double *full, double div)
{
    if (c->n == 0) return LZX_OK;
    const u32 g = (u32)((c->n + 255) / 256);
    hipLaunchKernelGGL(k_permute_in, dim3(g), dim3(256), 0, c->stream, io, c->d_gidx_of_old, full, div, c->n);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

int lzx_launch_permute_out(lzx_ctx *c, const double *full, double *io)
{
    if (c->n == 0) return LZX_OK;
    const u32 g = (u32)((c->n + 255) / 256);
    hipLaunchKernelGGL(k_permute_out, dim3(g), dim3(256), 0, c->stream, full, c->d_gidx_of_old, io, c->n);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

int lzx_launch_relayout(lzx_ctx *c, const double *io_layout, double *exchange_layout)
{
    const u64 cnt = (u64)c->world * c->xs;
    hipLaunchKernelGGL(k_relayout, dim3((u32)((cnt + 255) / 256)), dim3(256), 0, c->stream, io_layout, exchange_layout,
                       (u32)c->world, c->n_loc_pad, c->xs, c->xs0);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

int lzx_launch_multout(lzx_ctx *c, const double *t_dev, u32 k, double *out_loc)
{
    const u32 g = (c->n_loc_pad + LZX_VEC_BLOCK - 1) / LZX_VEC_BLOCK;
    hipLaunchKernelGGL(k_multout, dim3(g), dim3(LZX_VEC_BLOCK), 0, c->stream, c->d_Q, c->ldq, t_dev, k,
                       out_loc, c->n_loc_pad);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

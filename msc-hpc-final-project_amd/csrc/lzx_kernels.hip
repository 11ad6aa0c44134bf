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
#include <type_traits>
#include <algorithm>
#include "lzx_internal.h"
#include "lzx_spmv_body.h"
#include "lzx_reduce.h"


template <int HUB, bool NT>
__global__ void __launch_bounds__(LZX_SPMV_BLOCK) k_spmv(const SpmvArgs a)
{
    spmv_body<HUB, NT>(a, blockIdx.x, gridDim.x);
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

// Several ranks, lazy normalisation (lzx_api.hip): the vector that was exchanged and multiplied is the UNNORMALISED
// u_j = beta_{j-1} q_j, so one all-reduce carries both D = u_j . (A u_j) and B = ||u_j||^2:
//   beta_{j-1} = sqrt(B); alpha_j = D / B; q_j = u_j / beta_{j-1}; A q_j = w / beta_{j-1};
//   u_{j+1} = A q_j - alpha_j q_j - beta_{j-1} q_{j-1}   (serial/lib/lanczos.cc:26-37, same two rounded updates)
// first: u_0 = q_0 is already normalised (B := 1).  u_next == nullptr on the last iteration (only q_j is still needed).
// scal2 == nullptr (one rank): D and B are the fixed-order sums of pa[0..na) and pb[0..nb), closed here by every workgroup.
__global__ void __launch_bounds__(LZX_VEC_BLOCK)
k_lazy_update(const double *__restrict__ w, u32 w_rows, const double *__restrict__ u, const double *__restrict__ q_prev,
              const double *scal2, const double *pa, u32 na, const double *pb, u32 nb, int first, double *alpha_out,
              double *beta_out, double *q_out, double *u_next, double *partials_out, u32 n, double *iso, u32 iso_k, u32 iso_j,
              const double *prev_div, float *f32_next, u32 mail_world, const LazyDeferred df)
{
    __shared__ double sh[4];
    // df.limit > 0 (blocked SpMV whose k_pb_finish launch was skipped, round 5): row r of a multi-item band still lacks the
    // totals that launch adds to v -- its band's per-item sums in item order, then its split-row item totals -- exactly the
    // operands, in the order, of k_pb_finish (lzx_pb.hip); their share of alpha was formed by the gather pass
    auto deferred = [&](u32 row) -> double {
        const uint4 m = df.mrow[row];
        if (m.z == 0u) return 0.0;
        double s = 0.0;
        for (u32 k = 0; k < m.z; ++k) s += df.part[m.y + (size_t)k * m.w];
        if (row < df.n_long && df.long_mode[row] == 1)
            for (u32 it = df.item_first[row]; it < df.item_first[row + 1]; ++it) s += df.long_partial[it];
        return s;
    };
    double D, B;
    if (scal2 && mail_world) {   // in-process group: the ranks' pairs as their reduce kernels wrote them, added in rank order
        D = B = 0.0;
        for (u32 p = 0; p < mail_world; ++p) {
            D += scal2[2 * p];
            B += scal2[2 * p + 1];
        }
    } else if (scal2) {
        D = scal2[0];
        B = scal2[1];
    } else {
        D = block_sum_fixed_256(pa, na, sh);
        B = first ? 1.0 : block_sum_fixed_256(pb, nb, sh);
    }
    if (first) B = 1.0;
    const double beta = first ? 1.0 : sqrt(B);
    const double alpha = first ? D : D / B;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *alpha_out = alpha;
        if (beta_out) *beta_out = beta;
        if (iso) {
            // the rows without an edge (this launch covers [0, n) only): q_j = c_j q_0, u_j = d_j q_0 there, and the update
            // below with w = 0 is one scalar recurrence, rounded where the elementwise form rounds; their share of
            // ||u_{j+1}||^2 goes into one more partial
            const double cj = first ? 1.0 : iso[iso_k + 1 + iso_j] / beta;
            iso[iso_j] = cj;
            if (u_next) {
                double t = 0.0;
                t -= alpha * cj;
                if (q_prev) t -= beta * iso[iso_j - 1];
                iso[iso_k + 1 + iso_j + 1] = t;
                partials_out[gridDim.x] = (t * t) * iso[2 * iso_k + 3];
            }
        }
    }
    double nrm = 0.0;
    const u32 stride = gridDim.x * LZX_VEC_BLOCK * 2;
    const double bprev = prev_div ? *prev_div : 1.0;
    if (!q_out && !u_next) n = 0;   // last iteration with the unnormalised basis: only the scalars were wanted
    for (u32 i = (blockIdx.x * LZX_VEC_BLOCK + threadIdx.x) * 2; i < n; i += stride) {
        double2 q = *reinterpret_cast<const double2 *>(u + i);
        if (!first) {
            q.x /= beta;
            q.y /= beta;
        }
        if (q_out) *reinterpret_cast<double2 *>(q_out + i) = q;
        if (u_next) {
            // rows without an edge (the tail beyond w_rows) have (A u)_i = 0: not read, and the SpMV did not write them
            double2 t = i < w_rows ? *reinterpret_cast<const double2 *>(w + i) : make_double2(0.0, 0.0);
            if (i < df.limit) {
                t.x += deferred(i);
                if (i + 1 < df.limit) t.y += deferred(i + 1);
            }
            if (!first) {
                t.x /= beta;
                t.y /= beta;
            }
            t.x -= alpha * q.x;
            t.y -= alpha * q.y;
            if (q_prev) {
                double2 p = *reinterpret_cast<const double2 *>(q_prev + i);
                if (prev_div) {   // the column holds u_{j-1}: q_{j-1} = u_{j-1} / beta_{j-2}, as it was formed one iteration ago
                    p.x /= bprev;
                    p.y /= bprev;
                }
                t.x -= beta * p.x;
                t.y -= beta * p.y;
            }
            *reinterpret_cast<double2 *>(u_next + i) = t;
            if (f32_next) *reinterpret_cast<float2 *>(f32_next + i) = make_float2((float)t.x, (float)t.y);   // basis stored as fp32 (N4)
            nrm += t.x * t.x;
            nrm += t.y * t.y;
        }
    }
    nrm = wave_sum(nrm);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = nrm;
    __syncthreads();
    if (threadIdx.x == 0) partials_out[blockIdx.x] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

// the same two sums, stored into slot [rank] of every peer's mailbox (in-process groups; lzx_internal.h: d_mail)
__global__ void __launch_bounds__(LZX_VEC_BLOCK)
k_reduce2_mail(const double *pa, u32 na, const double *pb, u32 nb, const MailPeers peers, u32 world)
{
    __shared__ double sh[4];
    const double a = block_sum_fixed_256(pa, na, sh);
    __syncthreads();
    const double b = block_sum_fixed_256(pb, nb, sh);
    if (threadIdx.x < world) {
        double *o = peers.slot[threadIdx.x];
        o[0] = a;
        o[1] = b;
    }
}

// two fixed-order sums in one launch: out2[0] = sum pa, out2[1] = sum pb
__global__ void __launch_bounds__(LZX_VEC_BLOCK)
k_reduce2(const double *pa, u32 na, const double *pb, u32 nb, double *out2)
{
    __shared__ double sh[4];
    const double a = block_sum_fixed_256(pa, na, sh);
    __syncthreads();
    const double b = block_sum_fixed_256(pb, nb, sh);
    if (threadIdx.x == 0) {
        out2[0] = a;
        out2[1] = b;
    }
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

// a constant start vector is made where it is used (lzx_api.hip: scan_start_vector) instead of crossing PCIe
__global__ void k_fill(double *out, double value, u64 n)
{
    const u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 < n) *reinterpret_cast<double2 *>(out + i) = make_double2(value, value);
    else if (i < n) out[i] = value;
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

// the same with chunk 1 in its sparse form: [world][xs0] then this rank's packed segments (map = hand-over position)
__global__ void k_relayout_sparse(const double *io_layout, double *x, u32 world, u32 n_loc_pad, u32 xs0, const u32 *map, u64 xc1)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 c0 = (u64)world * xs0;
    if (i < c0) x[i] = io_layout[(size_t)(i / xs0) * n_loc_pad + (u32)(i % xs0)];
    else if (i - c0 < xc1) x[i] = io_layout[map[i - c0]];
}

// sendbuf[i] = slice[idx[i]]: what every peer's rows reference of this rank's slice, peer by peer
__global__ void k_sx_pack(const double *slice, const u32 *idx, u32 count, double *out)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = slice[idx[i]];
}

__global__ void k_to_f32(const double *in, float *out, u64 count)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = (float)in[i];
}
__global__ void k_to_f64(const float *in, double *out, u64 count)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = (double)in[i];
}

__global__ void k_permute_out(const double *full, const u32 *gidx, double *io, u64 n, const double *div)
{
    const u64 o = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (o < n) io[o] = div ? full[gidx[o]] / *div : full[gidx[o]];
}

// ans_loc = Q_loc t: the second dgemv of multOut (parallel-final/lib/multiplyOut.cu:42-44), on the
// device-resident basis; one thread per row, basis vectors streamed with unit stride.
// iso != nullptr: rows [rows_act, n) are kept as q_j[i] = iso[j] q_0[i] (lzx_internal.h: iso_on).
// tq: the coefficients the stored columns are multiplied by (t itself, or t_j / beta_{j-1} when the columns hold u_j).
template <typename QT>
__global__ void __launch_bounds__(LZX_VEC_BLOCK)
k_multout(const QT *Q, u32 ldq, const double *t, const double *tq, u32 k, double *out, u32 n, u32 rows_act, const double *iso)
{
    const u32 i = blockIdx.x * LZX_VEC_BLOCK + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    if (iso && i >= rows_act) {
        const double z = Q[i];
        for (u32 j = 0; j < k; ++j) s += (j ? iso[j] * z : z) * t[j];
    } else {
        // eight columns' loads in flight before the first multiply (the sum keeps its order); every column is read once per
        // call and nothing of it is worth keeping in the caches: non-temporal loads (10 M vertices, k = 50: 0.456 -> 0.425 ms,
        // 5.6 TB/s; profiles/r3_multout.txt)
        u32 j = 0;
        for (; j + 8 <= k; j += 8) {
            QT q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = __builtin_nontemporal_load(Q + (size_t)(j + u) * ldq + i);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += q[u] * tq[j + u];
        }
        for (; j < k; ++j) s += __builtin_nontemporal_load(Q + (size_t)j * ldq + i) * tq[j];
    }
    out[i] = s;
}

__global__ void k_multout_coeff(const double *t, const double *beta, u32 k, double *tq)
{
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < k) tq[j] = j ? t[j] / beta[j - 1] : t[j];
}

// sum of q_0[i]^2 over rows [r0, r1) in two fixed-shape steps (per-block partials, then one block); also c_0 = d_0 = 1
__global__ void __launch_bounds__(LZX_VEC_BLOCK)
k_iso_sumsq(const double *q0, u32 r0, u32 r1, double *partials)
{
    __shared__ double sh[4];
    double s = 0.0;
    for (u32 i = r0 + blockIdx.x * LZX_VEC_BLOCK + threadIdx.x; i < r1; i += gridDim.x * LZX_VEC_BLOCK) s += q0[i] * q0[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

__global__ void __launch_bounds__(LZX_VEC_BLOCK)
k_iso_prepare(const double *partials, u32 np, double *iso, u32 iso_k)
{
    __shared__ double sh[4];
    const double t = block_sum_fixed_256(partials, np, sh);
    if (threadIdx.x == 0) {
        iso[2 * iso_k + 3] = t;
        iso[0] = 1.0;
        iso[iso_k + 1] = 1.0;
    }
}

// basis columns 1 .. k-1 of the rows without an edge, materialised: q_j[i] = c_j q_0[i]
template <typename QT>
__global__ void k_iso_fill(QT *Q, u32 ldq, u32 j0, u32 k, u32 r0, u32 r1, const double *coeff)
{
    const u32 i = r0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= r1) return;
    const double z = Q[i];
    for (u32 j = j0; j < k; ++j) Q[(size_t)j * ldq + i] = (QT)(coeff[j] * z);
}

// One step of the Arnoldi pass of serial/lib/lanczos.cc:85-90 (decompose_with_arnoldi), fused the way the dependent
// chain allows: this launch applies the update of basis vector m (v -= dot_m q_m, dot_m closed here from the previous
// launch's partials, in every workgroup the same fixed order) and forms the partials of the NEXT inner product
// <v, q_next> over the updated v -- q_{m+1} inside the pass, q_j (the alpha_j partials) behind its last vector.
// q_m == nullptr: the pass's first launch, inner product only.  Modified Gram-Schmidt, the reference's order: every
// inner product sees the v the previous update left.
__global__ void __launch_bounds__(LZX_VEC_BLOCK)
k_mgs_step(double *v, const double *__restrict__ q_m, const double *d_partials, u32 d_np, const double *d_scal,
           const double *__restrict__ q_next, double *partials_out, u32 n)
{
    __shared__ double sh[4];
    double d = 0.0;
    if (q_m) d = d_np ? block_sum_fixed_256(d_partials, d_np, sh) : *d_scal;
    double acc = 0.0;
    const u32 stride = gridDim.x * LZX_VEC_BLOCK * 2;
    for (u32 i = (blockIdx.x * LZX_VEC_BLOCK + threadIdx.x) * 2; i < n; i += stride) {
        double2 w = *reinterpret_cast<const double2 *>(v + i);
        if (q_m) {
            const double2 q = *reinterpret_cast<const double2 *>(q_m + i);
            w.x -= d * q.x;
            w.y -= d * q.y;
            *reinterpret_cast<double2 *>(v + i) = w;
        }
        const double2 p = *reinterpret_cast<const double2 *>(q_next + i);
        acc += w.x * p.x;
        acc += w.y * p.y;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials_out[blockIdx.x] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

// convergence monitor: per-block partials of |y - y_prev|^2 (first half of `partials`) and |y|^2 (second half)
__global__ void __launch_bounds__(LZX_VEC_BLOCK)
k_change_partials(const double *__restrict__ y, const double *__restrict__ yp, double *partials, u32 n)
{
    __shared__ double sh[8];
    double d2 = 0.0, y2 = 0.0;
    for (u32 i = blockIdx.x * LZX_VEC_BLOCK + threadIdx.x; i < n; i += gridDim.x * LZX_VEC_BLOCK) {
        const double a = y[i], df = a - yp[i];
        d2 += df * df;
        y2 += a * a;
    }
    d2 = wave_sum(d2);
    y2 = wave_sum(y2);
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = d2; sh[4 + (threadIdx.x >> 6)] = y2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
        partials[gridDim.x + blockIdx.x] = ((sh[4] + sh[5]) + sh[6]) + sh[7];
    }
}

__global__ void k_widen_col(const float *in, double *out, u32 n)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)in[i];
}

// --------------------------------------------------------------------------------------------------
// Option "reference_order" (include/lzx.h; one rank): the loop's three reductions in the order
// serial/ fixes, so that alpha, beta and the basis can be compared with it BIT FOR BIT at any k.  Not a fast path: a parity
// instrument, run by tests only.
//   k_ref_spmv      one lane per row of the caller's CSR, entries added one at a time in ascending column order starting from
//                   0.0 -- serial/lib/SPMV.cc:24-27 and the thread-per-row cu_spMV1 (parallel-final/lib/cu_SPMV.cu:31-41).
//                   The vectors live in the engine's internal order; gidx maps a caller's vertex to its position there.
//   k_ref_products  prod[o] = a[gidx[o]] * b[gidx[o]], caller's order (the product rounded on its own: no FMA, as in the
//                   reference's generic x86-64 build)
//   k_ref_seqsum    ONE lane adds prod[0 .. n) left to right into one accumulator -- lanczosDecomp::inner_prod / norm,
//                   serial/lib/lanczos.cc:155-171 -- the other lanes of its workgroup only stage chunks of prod in LDS
__global__ void __launch_bounds__(256)
k_ref_spmv(const u64 *row_ptr, const u32 *col_idx, const u32 *gidx, const double *x, double *y, u64 n)
{
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    u64 e = row_ptr[r];
    const u64 end = row_ptr[r + 1];
    double acc = 0.0;
    // sixteen entries' look-ups in flight, then added in order (the sum's order is the entries' order whatever the loads do)
    for (; e + 16 <= end; e += 16) {
        double t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = x[gidx[col_idx[e + u]]];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += t[u];
    }
    for (; e < end; ++e) acc += x[gidx[col_idx[e]]];
    y[gidx[r]] = acc;
}

__global__ void __launch_bounds__(256)
k_ref_products(const double *a, const double *b, const u32 *gidx, double *prod, u64 n)
{
    const u64 o = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (o < n) {
        const u32 g = gidx[o];
        prod[o] = a[g] * b[g];
    }
}

static constexpr u32 LZX_REF_CHUNK = 8192;   // doubles staged per round (64 KiB of LDS)
__global__ void __launch_bounds__(1024)
k_ref_seqsum(const double *prod, u64 n, double *out)
{
    __shared__ double sh[LZX_REF_CHUNK];
    double acc = 0.0;
    for (u64 base = 0; base < n; base += LZX_REF_CHUNK) {
        const u32 cnt = (u32)(n - base < LZX_REF_CHUNK ? n - base : LZX_REF_CHUNK);
        for (u32 i = threadIdx.x; i < cnt; i += 1024) sh[i] = prod[base + i];
        __syncthreads();
        if (threadIdx.x == 0) {
            u32 i = 0;
            for (; i + 16 <= cnt; i += 16) {
                double t[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) t[u] = sh[i + u];
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += t[u];
            }
            for (; i < cnt; ++i) acc += sh[i];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = acc;
}

int lzx_launch_ref_spmv(lzx_ctx *c, const double *x, double *y)
{
    // rows of the padding (and nothing else) are not written by the kernel
    LZX_HIP(hipMemsetAsync(y, 0, sizeof(double) * c->ldq, c->stream));
    if (c->n == 0) return LZX_OK;
    hipLaunchKernelGGL(k_ref_spmv, dim3((u32)((c->n + 255) / 256)), dim3(256), 0, c->stream, c->d_row_ptr, c->d_col_idx, c->d_gidx_of_old, x, y, c->n);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

int lzx_launch_ref_dot(lzx_ctx *c, const double *a, const double *b, double *out)
{
    if (c->n) hipLaunchKernelGGL(k_ref_products, dim3((u32)((c->n + 255) / 256)), dim3(256), 0, c->stream, a, b, c->d_gidx_of_old, c->d_io, c->n);
    hipLaunchKernelGGL(k_ref_seqsum, dim3(1), dim3(1024), 0, c->stream, c->d_io, c->n, out);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

// --------------------------------------------------------------------------------------------------
// launch wrappers
static u32 vec_grid(const lzx_ctx *c)
{
    const u32 need = (c->n_loc_pad / 2 + LZX_VEC_BLOCK - 1) / LZX_VEC_BLOCK;
    // every block first sums the SpMV's ~1300 partials: on a small vector half as many blocks do half as much of that
    // (1 M rows: 13.5 -> 11.5 us with 4 per CU; 10 M rows: no difference, 2 per CU loses)
    const u32 cap = (u32)c->cu_count * (c->vec_per_cu_opt > 0 ? (u32)c->vec_per_cu_opt : (c->n_loc_pad <= (4u << 20) ? 4u : 8u));
    return need < 1 ? 1 : (need < cap ? need : cap);
}

// (the partials of k_pb_finish come last: while its launch is deferred into k_lazy_update they are not there)
u32 lzx_spmv_partials(const lzx_ctx *c) { return c->spmv_grid + (c->pb ? 0 : c->fin_grid) + lzx_pb_partials(c) - (c->pb_deferring && c->pb_defer_ok ? c->pb_finish_grid : 0u); }

static LazyDeferred lazy_deferred(const lzx_ctx *c)
{
    LazyDeferred d;
    if (c->pb && c->pb_deferring && c->pb_defer_ok) {
        d.mrow = c->d_pb_mrow;
        d.limit = c->pb_multi_limit;
        d.part = c->d_pb_part;
        d.item_first = c->d_item_first;
        d.long_partial = c->d_long_partial;
        d.long_mode = c->d_pb_long_multi;
        d.n_long = c->n_long64;
    }
    return d;
}

template <int HUB, bool NT>
static int launch_spmv_t(lzx_ctx *c, const SpmvArgs &a, hipStream_t st)
{
    auto kern = k_spmv<HUB, NT>;
    LZX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->spmv_lds));
    hipLaunchKernelGGL(kern, dim3(c->spmv_grid), dim3(LZX_SPMV_BLOCK), c->spmv_lds, st, a);
    return LZX_OK;
}

int lzx_launch_spmv(lzx_ctx *c, const SpmvLaunch &l)
{
    SpmvArgs a;
    a.sell_cols = c->d_sell_cols;
    a.slice_off = c->d_slice_off;
    a.slice_w = c->d_slice_w;
    // slices of rows without an edge (live_rows_only): nothing to sum, and the caller does not read their v
    const u32 live_slices = l.live_rows_only ? (c->rows_live > c->n_long64 ? (c->rows_live - c->n_long64) / LZX_SLICE : 0u) : c->n_slices;
    a.slice_perm = c->d_slice_perm;
    a.ns_w8 = c->ns_w8;
    a.ns_w4 = c->ns_w4;
    // the width-0 slices with rows that are read later: a prefix of their ascending id list
    a.ns_w0 = (u32)(std::lower_bound(c->h_slice_w0.begin(), c->h_slice_w0.end(), live_slices) - c->h_slice_w0.begin());
    a.n_slices = c->ns_wide;
    if (c->ns_wide == c->n_slices) a.n_slices = std::min<u32>(c->n_slices, live_slices);   // no classes: slices as they lie
    if (!(c->phase_mask_opt & 2)) a.n_slices = a.ns_w8 = a.ns_w4 = a.ns_w0 = 0;
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
    a.n_zero = c->pb ? c->n_long64 : 0;
    a.burst = c->burst_opt > 0 ? (u32)c->burst_opt : 0u;   // debug knob stage_burst = loads in flight per thread: the empty kernel takes 9.5 us with one, 8.7 with two, 10.8 with four, 14.5 with eight (256 CUs hit the same 128 KiB at once): off
    a.deep = c->deep_opt > 0 ? 1u : 0u;   // debug knob spmv_deep: no gain measured (DESIGN.md 3.1 g), off
    const bool nt = c->nt_opt > 0;
    // Blocked mode, option "side_stream": the staged-columns kernel and the scatter passes are independent (both only
    // read x), so the former can run on a side stream, its drain overlapping the scatter's ramp-up; the gather pass,
    // which adds into the v it wrote, waits for it.  Measured: C3 +1..2 %, C2 -4 % (two more event operations per
    // SpMV) -- off by default.
    const bool side = c->pb && c->side_opt > 0 && !c->trace;
    // Blocked mode.  The scatter pass and the staged-columns kernel both only read x and write different things; the
    // gather pass adds into the v the latter wrote and reads the values the former wrote.  Product form: the two share ONE
    // launch (lzx_pb.hip: k_pb_scatter_spmv; with the two-chunk exchange: the staged-columns workgroups + the chunk-0
    // units, then the rest of the scatter pass once chunk 1 has arrived).  With launches of their own (debug knobs
    // fuse_staged = 0, tracing, non-temporal loads) the scatter pass goes first and the staged-columns kernel between it
    // and the gather pass: the gather pass reads 0.6 GB of values just written and, straight behind the scatter pass,
    // competes with the write-back of those still dirty in the Infinity Cache (0.25 instead of 0.21 ms on C3); the
    // staged-columns kernel in between lets that write-back drain in its shadow.
    const bool scatter_first = c->pb && !side && c->pb_order_opt != 0 && !l.chunk1_ready;
    double *pb_partials = l.partials + c->spmv_grid + (c->pb ? 0 : c->fin_grid);
    hipStream_t hs = c->stream;
    if (side) {
        LZX_HIP(hipEventRecord(c->ev_fork, c->stream));
        LZX_HIP(hipStreamWaitEvent(c->stream3, c->ev_fork, 0));
        hs = c->stream3;
    }
    bool fused = false;
    // two-chunk exchange: the staged-columns workgroups share the launch of the chunk-0 scatter units (they come first in
    // that grid, as the kernel of its own did); the rest of the scatter pass follows once chunk 1 has arrived
    const bool fuse_chunked = c->pb && !side && l.chunk1_ready && c->codes16 && !nt && !c->trace && c->fuse_opt != 0 && c->fuse_opt != 1 &&
                              lzx_pb_can_fuse(c);
    if (fuse_chunked) {
        LZX_TRY(lzx_pb_launch(c, l.x, l.q_loc, l.v, pb_partials, l.chunk1_ready, nullptr, 3, &a, c->spmv_grid, &fused));
        if (fused) {
            LZX_HIP(hipGetLastError());
            return LZX_OK;
        }
        // (not fused after all -- a debug form of the scatter pass: the passes ran without the staged-columns kernel, which
        //  cannot happen in the product library; fall through would double the passes, so this is an error)
        LZX_FAIL(LZX_ERR_STATE, "fused launch was refused");
    }
    if (scatter_first) {
        if (c->trace) LZX_HIP(hipEventRecord(c->trace_ev[2], c->stream));
        // the staged-columns workgroups in the scatter pass's launch, behind its units (lzx_pb.hip: k_pb_scatter_spmv)
        const bool fuse = c->codes16 && !nt && !c->trace && c->fuse_opt != 0 && lzx_pb_can_fuse(c);
        LZX_TRY(lzx_pb_launch(c, l.x, l.q_loc, l.v, pb_partials, l.chunk1_ready, nullptr, 1, fuse ? &a : nullptr, c->spmv_grid, &fused));
    }
    if (c->trace) LZX_HIP(hipEventRecord(c->trace_ev[0], c->stream));
    if (fused) {
    } else if (c->codes16) {
        if (nt) LZX_TRY((launch_spmv_t<2, true>(c, a, hs)));
        else    LZX_TRY((launch_spmv_t<2, false>(c, a, hs)));
    } else if (c->hub > 0) {
        if (nt) LZX_TRY((launch_spmv_t<1, true>(c, a, hs)));
        else    LZX_TRY((launch_spmv_t<1, false>(c, a, hs)));
    } else {
        if (nt) LZX_TRY((launch_spmv_t<0, true>(c, a, hs)));
        else    LZX_TRY((launch_spmv_t<0, false>(c, a, hs)));
    }
    if (c->trace) LZX_HIP(hipEventRecord(c->trace_ev[1], c->stream));
    if (c->fin_grid > 0 && !c->pb) {   // blocked mode: k_pb_finish adds the split rows' totals
        hipLaunchKernelGGL(k_long_finish, dim3(c->fin_grid), dim3(LZX_VEC_BLOCK), 0, hs,
                           c->d_item_first, c->d_long_partial, c->n_long64, l.q_loc, l.v,
                           l.partials + c->spmv_grid);
    }
    LZX_HIP(hipGetLastError());
    if (c->trace && !scatter_first) LZX_HIP(hipEventRecord(c->trace_ev[2], c->stream));
    // entries whose column is not staged in LDS: two streaming passes that add into v (lzx_pb.hip)
    if (side) LZX_HIP(hipEventRecord(c->ev_join, c->stream3));
    if (scatter_first) {
        if (c->trace) LZX_HIP(hipEventRecord(c->trace_ev[5], c->stream));
        LZX_TRY(lzx_pb_launch(c, l.x, l.q_loc, l.v, pb_partials, nullptr, nullptr, 2));
    } else {
        LZX_TRY(lzx_pb_launch(c, l.x, l.q_loc, l.v, pb_partials, l.chunk1_ready, side ? c->ev_join : nullptr, 3));
    }
    if (c->trace) LZX_HIP(hipEventRecord(c->trace_ev[4], c->stream));
    return LZX_OK;
}

int lzx_launch_reduce(lzx_ctx *c, const double *partials, u32 np, double *out, int do_sqrt)
{
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(LZX_VEC_BLOCK), 0, c->stream, partials, np, out, do_sqrt);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

int lzx_launch_reduce2_mail(lzx_ctx *c, const double *pa, u32 na, const double *pb, u32 nb, const MailPeers &peers, u32 world)
{
    hipLaunchKernelGGL(k_reduce2_mail, dim3(1), dim3(LZX_VEC_BLOCK), 0, c->stream, pa, na, pb, nb, peers, world);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

int lzx_launch_reduce2(lzx_ctx *c, const double *pa, u32 na, const double *pb, u32 nb, double *out2)
{
    hipLaunchKernelGGL(k_reduce2, dim3(1), dim3(LZX_VEC_BLOCK), 0, c->stream, pa, na, pb, nb, out2);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

int lzx_launch_lazy_update(lzx_ctx *c, const double *w, u32 w_rows, const double *u, const double *q_prev, const double *scal2, int first,
                           double *alpha_out, double *beta_out, double *q_out, double *u_next, double *partials_out, u32 *np_out,
                           const double *prev_div, float *f32_next, u32 mail_world)
{
    const u32 g = vec_grid(c);
    hipLaunchKernelGGL(k_lazy_update, dim3(g), dim3(LZX_VEC_BLOCK), 0, c->stream, w, w_rows, u, q_prev, scal2, nullptr, 0u, nullptr, 0u,
                       first, alpha_out, beta_out, q_out, u_next, partials_out, c->iso_on ? c->rows_live : c->n_loc_pad,
                       c->iso_on ? c->d_iso : nullptr, c->iso_cap, (u32)(alpha_out - c->d_alpha), prev_div, f32_next, mail_world, lazy_deferred(c));
    LZX_HIP(hipGetLastError());
    *np_out = g + (c->iso_on ? 1u : 0u);
    return LZX_OK;
}

int lzx_launch_lazy_update_local(lzx_ctx *c, const double *w, u32 w_rows, const double *u, const double *q_prev, const double *pa, u32 na,
                                 const double *pb, u32 nb, int first, double *alpha_out, double *beta_out, double *q_out,
                                 double *u_next, double *partials_out, u32 *np_out, const double *prev_div, float *f32_next)
{
    const u32 g = vec_grid(c);
    hipLaunchKernelGGL(k_lazy_update, dim3(g), dim3(LZX_VEC_BLOCK), 0, c->stream, w, w_rows, u, q_prev, nullptr, pa, na, pb, nb, first,
                       alpha_out, beta_out, q_out, u_next, partials_out, c->iso_on ? c->rows_live : c->n_loc_pad,
                       c->iso_on ? c->d_iso : nullptr, c->iso_cap, (u32)(alpha_out - c->d_alpha), prev_div, f32_next, 0u, lazy_deferred(c));
    LZX_HIP(hipGetLastError());
    *np_out = g + (c->iso_on ? 1u : 0u);
    return LZX_OK;
}

int lzx_launch_mgs_step(lzx_ctx *c, double *v, const double *q_m, const double *d_partials, u32 d_np, const double *d_scal,
                        const double *q_next, double *partials_out, u32 *np_out)
{
    const u32 g = vec_grid(c);
    hipLaunchKernelGGL(k_mgs_step, dim3(g), dim3(LZX_VEC_BLOCK), 0, c->stream, v, q_m, d_partials, d_np, d_scal, q_next,
                       partials_out, c->n_loc_pad);
    LZX_HIP(hipGetLastError());
    *np_out = g;
    return LZX_OK;
}

int lzx_launch_change(lzx_ctx *c, const double *y, const double *y_prev, double *out2)
{
    // two fixed-shape steps: per-block partials (both sums side by side in d_partials, which is dead between two
    // iterations of the loop -- d_partials2 / 3 are not: they carry the norm partials), then one block closes them
    const u32 g = std::min<u32>(vec_grid(c), c->np_cap / 2);
    hipLaunchKernelGGL(k_change_partials, dim3(g), dim3(LZX_VEC_BLOCK), 0, c->stream, y, y_prev, c->d_partials, c->n_loc_pad);
    hipLaunchKernelGGL(k_reduce2, dim3(1), dim3(LZX_VEC_BLOCK), 0, c->stream, c->d_partials, g, c->d_partials + g, g, out2);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

int lzx_launch_widen_col(lzx_ctx *c, u32 col, double *out)
{
    hipLaunchKernelGGL(k_widen_col, dim3((c->n_loc_pad + 255) / 256), dim3(256), 0, c->stream, c->d_Qf + (size_t)col * c->ldq, out, c->n_loc_pad);
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

int lzx_launch_fill(lzx_ctx *c, double *out, double value, u64 count)
{
    if (count == 0) return LZX_OK;
    hipLaunchKernelGGL(k_fill, dim3((u32)((count / 2 + 256) / 256)), dim3(256), 0, c->stream, out, value, count);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

int lzx_launch_permute_out(lzx_ctx *c, const double *full, double *io, const double *div)
{
    if (c->n == 0) return LZX_OK;
    const u32 g = (u32)((c->n + 255) / 256);
    hipLaunchKernelGGL(k_permute_out, dim3(g), dim3(256), 0, c->stream, full, c->d_gidx_of_old, io, c->n, div);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

int lzx_launch_to_f32(lzx_ctx *c, const double *in, float *out, u64 count)
{
    if (count) hipLaunchKernelGGL(k_to_f32, dim3((u32)((count + 255) / 256)), dim3(256), 0, c->stream, in, out, count);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}
int lzx_launch_to_f64(lzx_ctx *c, const float *in, double *out, u64 count)
{
    if (count) hipLaunchKernelGGL(k_to_f64, dim3((u32)((count + 255) / 256)), dim3(256), 0, c->stream, in, out, count);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

int lzx_launch_sx_pack(lzx_ctx *c, const double *slice_loc, hipStream_t st)
{
    const u32 cnt = c->sx_send_off.empty() ? 0u : c->sx_send_off[c->world];
    if (cnt == 0) return LZX_OK;
    hipLaunchKernelGGL(k_sx_pack, dim3((cnt + 255) / 256), dim3(256), 0, st, slice_loc, c->d_sx_send_idx, cnt, c->d_sx_sendbuf);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

int lzx_launch_relayout(lzx_ctx *c, const double *io_layout, double *exchange_layout)
{
    if (c->sparse) {
        const u64 cnt = (u64)c->world * c->xs0 + c->xc1;
        hipLaunchKernelGGL(k_relayout_sparse, dim3((u32)((cnt + 255) / 256)), dim3(256), 0, c->stream, io_layout, exchange_layout,
                           (u32)c->world, c->n_loc_pad, c->xs0, c->d_sx_map, c->xc1);
        LZX_HIP(hipGetLastError());
        return LZX_OK;
    }
    const u64 cnt = (u64)c->world * c->xs;
    hipLaunchKernelGGL(k_relayout, dim3((u32)((cnt + 255) / 256)), dim3(256), 0, c->stream, io_layout, exchange_layout,
                       (u32)c->world, c->n_loc_pad, c->xs, c->xs0);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

int lzx_launch_multout(lzx_ctx *c, const double *t_dev, u32 k, double *out_loc)
{
    const u32 g = (c->n_loc_pad + LZX_VEC_BLOCK - 1) / LZX_VEC_BLOCK;
    // rows without an edge: always from the scalars (valid whether or not a fetch has materialised some columns meanwhile)
    const bool factored = c->iso_on;
    const double *tq = t_dev;
    if ((u64)k * (c->basis_u ? 2u : 1u) > c->np_cap)   // t (and, with the unnormalised basis, k more doubles) live in a partials buffer
        LZX_FAIL(LZX_ERR_LIMIT, "lzx_launch_multout: k = %u does not fit the coefficient staging (%u doubles)", k, c->np_cap);
    if (c->basis_u) {   // the caller left room for k more doubles behind t
        hipLaunchKernelGGL(k_multout_coeff, dim3((k + 255) / 256), dim3(256), 0, c->stream, t_dev, c->d_beta, k, const_cast<double *>(t_dev) + k);
        tq = t_dev + k;
    }
    if (c->qf32)
        hipLaunchKernelGGL(k_multout<float>, dim3(g), dim3(LZX_VEC_BLOCK), 0, c->stream, c->d_Qf, c->ldq, t_dev, tq, k,
                           out_loc, c->n_loc_pad, c->rows_live, factored ? c->d_iso : nullptr);
    else
        hipLaunchKernelGGL(k_multout<double>, dim3(g), dim3(LZX_VEC_BLOCK), 0, c->stream, c->d_Q, c->ldq, t_dev, tq, k,
                           out_loc, c->n_loc_pad, c->rows_live, factored ? c->d_iso : nullptr);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

int lzx_launch_iso_prepare(lzx_ctx *c, u32 k)
{
    if (c->iso_cap < k || !c->d_iso) {
        if (c->d_iso) (void)hipFree(c->d_iso);
        c->d_iso = nullptr;
        c->iso_cap = 0;
        LZX_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_iso), sizeof(double) * (2 * (size_t)k + 4)));
        c->iso_cap = k;
    }
    LZX_HIP(hipMemsetAsync(c->d_iso, 0, sizeof(double) * (2 * (size_t)c->iso_cap + 4), c->stream));
    const u32 g = std::min<u32>(256u, c->np_cap);
    hipLaunchKernelGGL(k_iso_sumsq, dim3(g), dim3(LZX_VEC_BLOCK), 0, c->stream, c->qf32 ? c->d_ring[0] : c->d_Q, c->rows_live, c->n_loc_pad, c->d_partials3);
    hipLaunchKernelGGL(k_iso_prepare, dim3(1), dim3(LZX_VEC_BLOCK), 0, c->stream, c->d_partials3, g, c->d_iso, c->iso_cap);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

int lzx_launch_iso_fill(lzx_ctx *c, u32 k)
{
    const u32 j0 = c->iso_cols_filled < 1 ? 1u : c->iso_cols_filled;   // columns below j0 are there already
    if (k > j0 && c->n_loc_pad > c->rows_live) {
        const u32 rows = c->n_loc_pad - c->rows_live;
        const double *coeff = c->basis_u ? c->d_iso + c->iso_cap + 1 : c->d_iso;   // d_j where the columns hold u_j, c_j otherwise
        if (c->qf32)
            hipLaunchKernelGGL(k_iso_fill<float>, dim3((rows + 255) / 256), dim3(256), 0, c->stream, c->d_Qf, c->ldq, j0, k, c->rows_live, c->n_loc_pad, coeff);
        else
        hipLaunchKernelGGL(k_iso_fill<double>, dim3((rows + 255) / 256), dim3(256), 0, c->stream, c->d_Q, c->ldq, j0, k, c->rows_live, c->n_loc_pad, coeff);
        LZX_HIP(hipGetLastError());
    }
    if (k > c->iso_cols_filled) c->iso_cols_filled = k;
    return LZX_OK;
}

// lzx_reduce.h -- the fixed-order block reduction every vector kernel closes the loop's partial sums with (device code;
// include after lzx_spmv_body.h, which has wave_sum).
#pragma once

// Sum p[0..np) identically in every workgroup of a 256-thread launch. sh: >= 4 doubles of LDS.
__device__ __forceinline__ double block_sum_fixed_256(const double *p, u32 np, double *sh)
{
    // A thread's values are added in index order (the result does not depend on how the loads are scheduled), but they
    // are FETCHED eight at a time: one load per loop iteration was five dependent round trips for the ~1300 partials of
    // a blocked SpMV, at the head of every vector kernel (on the 1 M-vertex graph a third of k_lazy_update's 14 us).
    double s = 0.0;
    u32 i = threadIdx.x;
    for (; i + 7 * LZX_VEC_BLOCK < np; i += 8 * LZX_VEC_BLOCK) {
        double t[8];
#pragma unroll
        for (u32 u = 0; u < 8; ++u) t[u] = p[i + u * LZX_VEC_BLOCK];
#pragma unroll
        for (u32 u = 0; u < 8; ++u) s += t[u];
    }
    if (i < np) {   // up to seven more: clamped, unconditional loads
        double t[7];
#pragma unroll
        for (u32 u = 0; u < 7; ++u) {
            const u32 j = i + u * LZX_VEC_BLOCK;
            t[u] = p[j < np ? j : i];
        }
#pragma unroll
        for (u32 u = 0; u < 7; ++u)
            if (i + u * LZX_VEC_BLOCK < np) s += t[u];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    const double t = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    __syncthreads();
    return t;
}

// lzx_pb_shared.h -- helpers of the propagation-blocked SpMV (lzx_pb.hip): the code-word helpers, the wavefront sum, the
// allocation helpers.  (Until round 4 a second translation unit of the debug library, lzx_pb_dbg.hip, held the round-2 / 3
// experiments on the two passes -- persistent and ticketed forms, ablation switches; their results are in profiles/NOTES.md,
// their code in the repository's history.)
#pragma once

#include <vector>

#include "lzx_internal.h"

__device__ __forceinline__ u32 pbr_flag(const uint4 &c, int e)
{
    const u32 w = (e >> 1) == 0 ? c.x : (e >> 1) == 1 ? c.y : (e >> 1) == 2 ? c.z : c.w;
    return (w >> ((e & 1) ? 31 : 15)) & 1u;
}
__device__ __forceinline__ u32 pbr_half(const uint4 &c, int e)
{
    const u32 w = (e >> 1) == 0 ? c.x : (e >> 1) == 1 ? c.y : (e >> 1) == 2 ? c.z : c.w;
    return (e & 1) ? (w >> 16) : (w & 0xffffu);
}
__device__ __forceinline__ u32 lanes_below(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}

__device__ __forceinline__ double wave_sum_pb(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename T>
static inline int pb_alloc(T **p, u64 count)
{
    *p = nullptr;
    LZX_HIP(hipMalloc(reinterpret_cast<void **>(p), (count ? count : 1) * sizeof(T)));
    return LZX_OK;
}
template <typename T>
static inline void pb_free(T *&p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}

// lzx_pb_shared.h -- what lzx_pb.hip (the propagation-blocked SpMV, both libraries) and lzx_pb_dbg.hip (its experiments,
// debug library only) have in common: the code-word helpers, the wavefront sum, the allocation helpers, the record kinds of
// the ticketed gather pass, and the entry points of the experiments.
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

// record kinds of the ticketed gather pass (k_pb_gather3, lzx_pb_dbg.hip)
enum : u32 { LZX_G3_NORMAL = 0, LZX_G3_ONE_ROW = 1, LZX_G3_GROUP = 2, LZX_G3_IDLE = 3 };

#ifdef LZX_DEBUG_KNOBS
// ---- lzx_pb_dbg.hip: the experiments behind DESIGN.md section 3.1 (knobs pb_persistent, pb_gather_tickets, LZX_ABLATE) ----
int lzx_pbdbg_ablate();   // value of the environment switch LZX_ABLATE (0: off)
int lzx_pbdbg_segments(lzx_ctx *c, hipStream_t st, const std::vector<u32> &sstart, const std::vector<u32> &qstart, u32 nb, u32 nsteps);
int lzx_pbdbg_persist_records(lzx_ctx *c, hipStream_t st, const std::vector<u32> &items, const std::vector<u32> &row0, const std::vector<u32> &rep);
int lzx_pbdbg_g3_records(lzx_ctx *c, hipStream_t st, std::vector<u32> &g3_out, std::vector<u64> &g3_cost, const std::vector<u32> &items,
                         const std::vector<u32> &row0, const std::vector<u32> &rep, const std::vector<u32> &rstart);
// true: an experimental form of the pass was launched (or failed: *rc) instead of the product kernel
bool lzx_pbdbg_scatter(lzx_ctx *c, u32 u0, u32 u1, const double *x, int *rc);
bool lzx_pbdbg_gather(lzx_ctx *c, double *v, const double *q_loc, double *partials, int *rc);
#endif

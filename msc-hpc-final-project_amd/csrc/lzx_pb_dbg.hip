// lzx_pb_dbg.hip -- experiments on the propagation-blocked SpMV (lzx_pb.hip), compiled into liblzx_dbg.so ONLY.
//
// Everything here was built, measured and NOT adopted: the persistent scatter / gather passes of
// round 2 (knob pb_persistent: 1 = static schedule, 2 = tickets), the ticketed gather pass of round 3 (knob
// pb_gather_tickets), and a copy of the scatter pass's body with the LZX_ABLATE switches behind the ablation table.  The
// product library does not contain this translation unit; lzx_pb.hip calls into it at a handful of places under
// LZX_DEBUG_KNOBS (lzx_pb_shared.h).
#ifdef LZX_DEBUG_KNOBS
#include <algorithm>
#include <cstdlib>
#include <queue>
#include <utility>

#include "lzx_internal.h"
#include "lzx_spmv_body.h"
#include "lzx_pb_shared.h"

namespace {
// pieces per step (a copy of lzx_pb.hip's build kernel: the static scatter schedule weighs steps by them)
__global__ void __launch_bounds__(64) k_pbrd_count(const uint4 *rcode, u32 *cnt)
{
    const uint4 c = rcode[(size_t)blockIdx.x * 64 + threadIdx.x];
    u32 n = __popc(c.x & 0x80008000u) + __popc(c.y & 0x80008000u) + __popc(c.z & 0x80008000u) + __popc(c.w & 0x80008000u);
    for (int o = 32; o > 0; o >>= 1) n += (u32)__shfl_xor((int)n, o, 64);
    if (threadIdx.x == 0) cnt[blockIdx.x] = n;
}

// ---- the scatter pass's body as lzx_pb.hip has it, plus the ablation switches (LZX_ABLATE = 1 .. 12) ----------------
template <u32 CB>
__device__ __forceinline__ void
pb_scatter_body_abl(const u32 *unit, const uint4 *scode, const u32 *sbase, const uint2 *q_lcol, const u32 *q_dst,
                const double *__restrict__ x, u64 xlen, double *val, int ablate_arg, const u32 ublock)
{
    const int ablate = ablate_arg;
    const bool ab_store = ablate == 6 || ablate == 9, ab_lds = ablate == 7 || ablate == 9, ab_carry = ablate == 8 || ablate == 9;
    extern __shared__ __attribute__((aligned(16))) double tile[];   // CB staged values + a zero for padding
    const u32 band = unit[5 * ublock];
    const u64 base = (u64)band * CB;
    // staging is dead time for this CU (the tile leaves room for one workgroup): all eight 16-byte loads of a
    // thread are issued before the first LDS write, so it costs one memory round trip
    if (ablate == 10) {
    } else if (ablate == 11 && base + CB <= xlen) {   // one 16-byte load per round trip (staging experiment)
        const double2 *src = reinterpret_cast<const double2 *>(x + base);
        for (u32 u = 0; u < CB / 2048; ++u) {
            const double2 t = src[threadIdx.x + u * 1024];
            reinterpret_cast<double2 *>(tile)[threadIdx.x + u * 1024] = t;
            __builtin_amdgcn_s_waitcnt(0);
        }
    } else if (ablate == 12 && base + CB <= xlen) {   // two round trips of four loads
        const double2 *src = reinterpret_cast<const double2 *>(x + base);
#pragma unroll
        for (u32 h = 0; h < 2; ++h) {
            double2 t[CB / 4096];
#pragma unroll
            for (u32 u = 0; u < CB / 4096; ++u) t[u] = src[threadIdx.x + (h * (CB / 4096) + u) * 1024];
#pragma unroll
            for (u32 u = 0; u < CB / 4096; ++u) reinterpret_cast<double2 *>(tile)[threadIdx.x + (h * (CB / 4096) + u) * 1024] = t[u];
            __builtin_amdgcn_s_waitcnt(0);
        }
    } else if (base + CB <= xlen) {
        const double2 *src = reinterpret_cast<const double2 *>(x + base);   // band starts are 128 KiB aligned
        double2 t[CB / 2048];
#pragma unroll
        for (u32 u = 0; u < CB / 2048; ++u) t[u] = src[threadIdx.x + u * 1024];
#pragma unroll
        for (u32 u = 0; u < CB / 2048; ++u) reinterpret_cast<double2 *>(tile)[threadIdx.x + u * 1024] = t[u];
    } else {
        for (u32 j = threadIdx.x; j < CB; j += 1024) tile[j] = base + j < xlen ? x[base + j] : 0.0;
    }
    if (threadIdx.x < 2) tile[CB + threadIdx.x] = 0.0;
    for (u32 j = threadIdx.x; j < 16 * 66; j += 1024) tile[CB + 2 + j] = 0.0;   // the wavefronts' carry slots
    __syncthreads();
    const u32 lane = threadIdx.x & 63;
    const u32 wv = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr u32 W = 1024 / 64;

    {   // ---- reduced bands
        const u32 beg = unit[5 * ublock + 1], end = unit[5 * ublock + 2];
        // A row whose entries span several lanes is summed across them through 65 wave-private LDS slots: every lane
        // adds what follows its last piece end (its whole sum if it has none) to the slot named after the last lane
        // before it that holds a piece end; the lane holding the row's end starts its running sum from that slot.
        // One ds_add + one ds_read per lane and step, no shuffles, no scan.
        double *carry = tile + CB + 2 + wv * 66;
        auto body = [&](const uint4 &c, u32 pos) {
            double xv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[e] = ab_lds ? (double)pbr_half(c, e) : tile[pbr_half(c, e) & 0x7fffu];
            u32 ends = 0;   // bit e: entry e closes a piece
#pragma unroll
            for (int e = 0; e < 8; ++e) ends |= pbr_flag(c, e) << e;
            const bool has = ends != 0;
            const int last = has ? 31 - __clz((int)ends) : -1;
            double tail = 0.0;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (e > last) tail += xv[e];
            const unsigned long long holders = __ballot(has);
            const unsigned long long before = holders & ((1ull << lane) - 1ull);
            const u32 from = before ? 64u - (u32)__clzll((long long)before) : 0u;   // 1 + last holder before this lane
            double s = 0.0;
            if (!ab_carry) {
            atomicAdd(&carry[has ? lane + 1 : from], tail);
            __builtin_amdgcn_wave_barrier();
            s = has ? carry[from] : 0.0;
            __builtin_amdgcn_wave_barrier();
            if (has) carry[from] = 0.0;
            }
            double *out = val + pos;   // wave-uniform: the step's first value slot
            u32 done = 0;              // pieces of the planes before this one
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool f = (ends >> e) & 1u;
                s += xv[e];
                const unsigned long long m = __ballot(f);
                if (m) {               // scalar branch: steps of few long rows have mostly empty planes
                    if (f) {
                        if (!ab_store || s == 1.2345e-300) out[done + lanes_below(m)] = s;
                        s = 0.0;
                    }
                    done += (u32)__popcll(m);
                }
            }
        };
        u32 s = beg + wv;
        for (; s + 3 * W < end; s += 4 * W) {
            uint4 c[4];
            u32 b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c[u] = scode[(size_t)(s + u * W) * 64 + lane];
                b[u] = (u32)__builtin_amdgcn_readfirstlane((int)sbase[s + u * W]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) body(c[u], b[u]);
        }
        if (s < end) {   // up to three more steps: requested together (wave-uniform predicates), not one round trip each
            uint4 c[3];
            u32 b[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (s + u * W < end) {
                    c[u] = scode[(size_t)(s + u * W) * 64 + lane];
                    b[u] = (u32)__builtin_amdgcn_readfirstlane((int)sbase[s + u * W]);
                }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (s + u * W < end) body(c[u], b[u]);
        }
    }

    {   // ---- plain bands
        const u32 beg = unit[5 * ublock + 3], end = unit[5 * ublock + 4];
        // wavefront w takes the unit's 256-quad blocks w, w + 16, ...: the workgroup reads one stream and its writes
        // move through the value array together
        for (u32 blk = beg + wv * 256u; blk < end; blk += W * 256u) {
        const u32 wend = blk + 256u < end ? blk + 256u : end;
        u32 j = blk + lane;
        for (; j + 3 * 64 < wend; j += 4 * 64) {
            uint2 c[4];
            u32 d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c[u] = q_lcol[j + u * 64];
                d[u] = q_dst[j + u * 64];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                double2 lo, hi;
                if (ablate != 2) {
                    lo.x = tile[c[u].x & 0xffffu];
                    lo.y = tile[c[u].x >> 16];
                    hi.x = tile[c[u].y & 0xffffu];
                    hi.y = tile[c[u].y >> 16];
                }
                if (ablate == 2) { lo.x = c[u].x; lo.y = c[u].y; hi = lo; }
                double2 *out = reinterpret_cast<double2 *>(val + d[u]);   // 32-byte aligned: slots of a quad
                if (ablate == 1) { if (lo.x + lo.y + hi.x + hi.y == 1.2345e-300) out[0] = lo; continue; }
                if (ablate == 4) out = reinterpret_cast<double2 *>(val + (size_t)(j + u * 64) * 4);
                out[0] = lo;
                if (ablate != 3) out[1] = hi;
            }
        }
        if (wend < blk + 256u) {   // a partly filled block: up to three more quads per lane, requested together
            uint2 c[3];
            u32 d[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const u32 jj = j + u * 64 < wend ? j + u * 64 : blk;     // clamped: unconditional loads
                c[u] = q_lcol[jj];
                d[u] = q_dst[jj];
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (j + u * 64 < wend) {
                    double2 lo, hi;
                    lo.x = tile[c[u].x & 0xffffu];
                    lo.y = tile[c[u].x >> 16];
                    hi.x = tile[c[u].y & 0xffffu];
                    hi.y = tile[c[u].y >> 16];
                    double2 *out = reinterpret_cast<double2 *>(val + d[u]);
                    out[0] = lo;
                    out[1] = hi;
                }
            }
        }
        }
    }
}


template <u32 CB>
__global__ void __launch_bounds__(1024)
k_pb_scatter_abl(const u32 *unit, const uint4 *scode, const u32 *sbase, const uint2 *q_lcol, const u32 *q_dst,
                 const double *__restrict__ x, u64 xlen, double *val, int ablate_arg)
{
    pb_scatter_body_abl<CB>(unit, scode, sbase, q_lcol, q_dst, x, xlen, val, ablate_arg, blockIdx.x);
}

// ==== round-2 experiments (DESIGN.md section 3.1 g): compiled into liblzx_dbg.so only, selected by the debug knob
// pb_persistent (1 static schedule, 2 tickets); none was faster than the per-unit kernels above, which both libraries run.
// One reduced step, lean form (the scatter pass turned out to be bound by its instruction stream, not by memory:
// with stores, LDS look-ups and the carry all switched off it still took 0.27 of its 0.31 ms, profiles/README.md).
// Same format, same value order and the same sums as k_pb_scatter's step body, in about half the instructions:
//   * the eight piece-end flags are the sign bits of the eight half-words: one 16- or 32-bit signed compare each
//     gives the lane mask that is at once the per-lane flag, the ballot of the plane and the branch condition;
//   * what a lane hands on (the sum behind its last piece end, or its whole sum) falls out of one running sum that is
//     reset at every flag, instead of being selected by the position of the last flag;
//   * no mask of ends per lane, no count-leading-zeros, no per-plane vote.
__device__ __forceinline__ void pbr_step(const uint4 &c, u32 pos, const double *tile, double *carry, u32 lane, double *val)
{
    const u32 w[4] = {c.x, c.y, c.z, c.w};
    double xv[8];
    bool f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const u32 h = (e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu);
        xv[e] = tile[h & 0x7fffu];
        // sign of the half-word: the high one is the sign of the 32-bit word
        f[e] = (e & 1) ? ((int)w[e >> 1] < 0) : ((w[e >> 1] & 0x8000u) != 0u);
    }
    const bool has = f[0] | f[1] | f[2] | f[3] | f[4] | f[5] | f[6] | f[7];
    // running sum, reset behind every piece end: at the end it is what this lane hands on
    double t = 0.0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        t += xv[e];
        t = f[e] ? 0.0 : t;
    }
    const unsigned long long holders = __ballot(has);
    const unsigned long long before = holders & ((1ull << lane) - 1ull);
    const u32 from = before ? 64u - (u32)__clzll((long long)before) : 0u;   // 1 + last holder before this lane
    atomicAdd(&carry[has ? lane + 1 : from], t);
    __builtin_amdgcn_wave_barrier();
    double s = has ? carry[from] : 0.0;
    __builtin_amdgcn_wave_barrier();
    if (has) carry[from] = 0.0;
    double *out = val + pos;
    u32 done = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        s += xv[e];
        const unsigned long long m = __ballot(f[e]);
        if (m) {                          // scalar: steps of few long rows have mostly empty planes
            if (f[e]) out[done + lanes_below(m)] = s;
            s = f[e] ? 0.0 : s;
            done += (u32)__popcll(m);
        }
    }
}

// ---- persistent forms of the two passes (experiments of round 2; debug library only) ---------------------------
// Both passes above start every work unit cold: a unit record, then the tables it points to, then the first loads --
// three dependent memory round trips (plus, in the scatter pass, the 128 KiB band) before a workgroup streams, with one
// (scatter) or two (gather) workgroups per CU to hide them behind.  unit_bench (tools/unit_bench.hip) shows the inner
// loops alone reach 6.1 TB/s (gather) and 5.1 TB/s (scatter, read + written) at this very occupancy, against 3.8 and
// 4.3 TB/s of the passes.  Here a workgroup stays resident and draws units from a ticket counter (so the hardware's
// dynamic balancing is kept): the next ticket and the next unit's record arrive while the current unit streams, and in
// the scatter pass the next unit's x band is fetched into registers meanwhile (and not at all when the band stays).
// Tickets: t = atomicAdd(counter, 1) - base; a workgroup stops at its first t >= n, so a launch of G workgroups
// advances the counter by exactly n + G, which the host adds to `base` for the next launch: no reset between launches.

template <u32 CB>
__global__ void __launch_bounds__(1024)
k_pb_scatter2(const u32 *unit, u32 n_units, u32 *queue, u32 qbase, const uint4 *scode, const u32 *sbase, const uint2 *q_lcol,
              const u32 *q_dst, const double *__restrict__ x, u64 xlen, double *val, unsigned long long *stamps)
{
    // stamps (debug library only, else null): [4 * workgroup] start, end (100 MHz ticks), units done | restagings << 32,
    // ticks in the reduced part | ticks in the plain part << 32 (wavefront 0's clock)
    unsigned long long t_start = 0, t_red = 0, t_plain = 0, t_mark = 0;
    u32 n_done = 0, n_restaged = 0;
    if (stamps) t_start = wall_clock64();
    extern __shared__ __attribute__((aligned(16))) double tile[];   // CB staged values + a zero for padding
    u32 *tick = reinterpret_cast<u32 *>(tile + CB + 2 + 16 * 66);   // [2]
    const u32 lane = threadIdx.x & 63;
    const u32 wv = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr u32 W = 1024 / 64;
    constexpr u32 PRE = CB / 2048;                                  // 16-byte loads per thread for one band
    const u64 xlen2 = xlen / 2;                                      // xlen is even: a double2 is inside or outside

    if (threadIdx.x == 0) {
        const u32 t0 = atomicAdd(queue, 1u) - qbase;
        tick[0] = t0;
        tick[1] = t0 < n_units ? atomicAdd(queue, 1u) - qbase : 0xffffffffu;
    }
    if (threadIdx.x < 2) tile[CB + threadIdx.x] = 0.0;
    for (u32 j = threadIdx.x; j < 16 * 66; j += 1024) tile[CB + 2 + j] = 0.0;   // the wavefronts' carry slots
    __syncthreads();
    u32 cur = (u32)__builtin_amdgcn_readfirstlane((int)tick[0]);
    u32 nxt = (u32)__builtin_amdgcn_readfirstlane((int)tick[1]);
    if (cur >= n_units) return;
    u32 band = (u32)__builtin_amdgcn_readfirstlane((int)unit[5 * cur]);
    {   // the first band: staged directly (one round trip)
        const double2 *src = reinterpret_cast<const double2 *>(x) + (u64)band * (CB / 2);
        const u64 b2 = (u64)band * (CB / 2);
        double2 t[PRE];
#pragma unroll
        for (u32 u = 0; u < PRE; ++u) {
            const u32 j = threadIdx.x + u * 1024;
            t[u] = b2 + j < xlen2 ? src[j] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (u32 u = 0; u < PRE; ++u) reinterpret_cast<double2 *>(tile)[threadIdx.x + u * 1024] = t[u];
    }
    __syncthreads();
    double *carry = tile + CB + 2 + wv * 66;
    for (;;) {
        const u32 s_beg = (u32)__builtin_amdgcn_readfirstlane((int)unit[5 * cur + 1]);
        const u32 s_end = (u32)__builtin_amdgcn_readfirstlane((int)unit[5 * cur + 2]);
        const u32 q_beg = (u32)__builtin_amdgcn_readfirstlane((int)unit[5 * cur + 3]);
        const u32 q_end = (u32)__builtin_amdgcn_readfirstlane((int)unit[5 * cur + 4]);
        const bool have_next = nxt < n_units;
        u32 t2 = 0xffffffffu;
        if (threadIdx.x == 0 && have_next) t2 = atomicAdd(queue, 1u) - qbase;   // arrives while this unit streams
        const u32 band_next = have_next ? (u32)__builtin_amdgcn_readfirstlane((int)unit[5 * nxt]) : band;
        const bool restage = band_next != band;
        double2 pre[PRE];
        if (restage) {
            const double2 *src = reinterpret_cast<const double2 *>(x) + (u64)band_next * (CB / 2);
            const u64 b2 = (u64)band_next * (CB / 2);
#pragma unroll
            for (u32 u = 0; u < PRE; ++u) {
                const u32 j = threadIdx.x + u * 1024;
                pre[u] = b2 + j < xlen2 ? src[j] : make_double2(0.0, 0.0);
            }
        }
        if (stamps) t_mark = wall_clock64();
        {   // ---- reduced steps (see k_pb_scatter; lean step body: pbr_step)
            u32 s = s_beg + wv;
            for (; s + 3 * W < s_end; s += 4 * W) {
                uint4 c[4];
                u32 b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    c[u] = scode[(size_t)(s + u * W) * 64 + lane];
                    b[u] = (u32)__builtin_amdgcn_readfirstlane((int)sbase[s + u * W]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) pbr_step(c[u], b[u], tile, carry, lane, val);
            }
            for (; s < s_end; s += W) pbr_step(scode[(size_t)s * 64 + lane], (u32)__builtin_amdgcn_readfirstlane((int)sbase[s]), tile, carry, lane, val);
        }
        if (stamps) { const unsigned long long t = wall_clock64(); t_red += t - t_mark; t_mark = t; }
        // ---- plain quads (see k_pb_scatter)
        for (u32 blk = q_beg + wv * 256u; blk < q_end; blk += W * 256u) {
            const u32 wend = blk + 256u < q_end ? blk + 256u : q_end;
            u32 j = blk + lane;
            for (; j + 3 * 64 < wend; j += 4 * 64) {
                uint2 c[4];
                u32 d[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    c[u] = q_lcol[j + u * 64];
                    d[u] = q_dst[j + u * 64];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    double2 lo, hi;
                    lo.x = tile[c[u].x & 0xffffu];
                    lo.y = tile[c[u].x >> 16];
                    hi.x = tile[c[u].y & 0xffffu];
                    hi.y = tile[c[u].y >> 16];
                    double2 *out = reinterpret_cast<double2 *>(val + d[u]);
                    out[0] = lo;
                    out[1] = hi;
                }
            }
            for (; j < wend; j += 64) {
                const uint2 c = q_lcol[j];
                double2 lo, hi;
                lo.x = tile[c.x & 0xffffu];
                lo.y = tile[c.x >> 16];
                hi.x = tile[c.y & 0xffffu];
                hi.y = tile[c.y >> 16];
                double2 *out = reinterpret_cast<double2 *>(val + q_dst[j]);
                out[0] = lo;
                out[1] = hi;
            }
        }
        if (stamps) t_plain += wall_clock64() - t_mark;
        if (threadIdx.x == 0) tick[0] = t2;
        __syncthreads();                      // every wavefront is done with the band in LDS
        ++n_done;
        if (!have_next) break;
        n_restaged += restage ? 1u : 0u;
        if (restage) {
#pragma unroll
            for (u32 u = 0; u < PRE; ++u) reinterpret_cast<double2 *>(tile)[threadIdx.x + u * 1024] = pre[u];
        }
        const u32 t2_all = (u32)__builtin_amdgcn_readfirstlane((int)tick[0]);
        __syncthreads();
        cur = nxt;
        nxt = t2_all;
        band = band_next;
    }
    if (stamps && threadIdx.x == 0) {
        stamps[4 * blockIdx.x] = t_start;
        stamps[4 * blockIdx.x + 1] = wall_clock64();
        stamps[4 * blockIdx.x + 2] = n_done | ((unsigned long long)n_restaged << 32);
        stamps[4 * blockIdx.x + 3] = t_red | (t_plain << 32);
    }
}

// Scatter pass, static form (default).  Time stamps of the ticket-driven form (tools/perf_probe.py @st, C3) showed a
// wavefront working 173 of the 281 us its workgroup is resident: the rest it waits, at the two barriers around every
// unit, for the slowest of the sixteen wavefronts (their steps differ in pieces) and for the band to be restaged --
// and a unit is only ~15 steps per wavefront.  Here every workgroup owns ONE contiguous stretch of the scatter order,
// cut by the host so that all stretches cost the same (bytes read + written, counted per step); a stretch lies in one
// column band or a few, and inside a band its wavefronts run through all their steps and quads with no barrier at
// all: per workgroup two or three barrier pairs instead of fourteen, and the band is restaged 2.4 instead of 7 times.
// segment: {band, first step, last step, first quad, last quad}; seg_begin[w]: first segment of workgroup w.
template <u32 CB>
__global__ void __launch_bounds__(1024)
k_pb_scatter3(const u32 *seg, const u32 *seg_begin, const uint4 *scode, const u32 *sbase, const uint2 *q_lcol, const u32 *q_dst,
              const double *__restrict__ x, u64 xlen, double *val, unsigned long long *stamps)
{
    extern __shared__ __attribute__((aligned(16))) double tile[];   // CB staged values + a zero for padding
    const u32 lane = threadIdx.x & 63;
    const u32 wv = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr u32 W = 1024 / 64;
    constexpr u32 PRE = CB / 2048;
    const u64 xlen2 = xlen / 2;
    unsigned long long t_start = 0, t_red = 0, t_plain = 0, t_mark = 0;
    if (stamps) t_start = wall_clock64();
    const u32 i0 = (u32)__builtin_amdgcn_readfirstlane((int)seg_begin[blockIdx.x]);
    const u32 i1 = (u32)__builtin_amdgcn_readfirstlane((int)seg_begin[blockIdx.x + 1]);
    if (i0 >= i1) return;
    if (threadIdx.x < 2) tile[CB + threadIdx.x] = 0.0;
    for (u32 j = threadIdx.x; j < 16 * 66; j += 1024) tile[CB + 2 + j] = 0.0;   // the wavefronts' carry slots
    auto fetch_band = [&](u32 band, double2 (&t)[PRE]) {
        const double2 *src = reinterpret_cast<const double2 *>(x) + (u64)band * (CB / 2);
        const u64 b2 = (u64)band * (CB / 2);
#pragma unroll
        for (u32 u = 0; u < PRE; ++u) {
            const u32 j = threadIdx.x + u * 1024;
            t[u] = b2 + j < xlen2 ? src[j] : make_double2(0.0, 0.0);
        }
    };
    double2 pre[PRE];
    fetch_band((u32)__builtin_amdgcn_readfirstlane((int)seg[5 * i0]), pre);
#pragma unroll
    for (u32 u = 0; u < PRE; ++u) reinterpret_cast<double2 *>(tile)[threadIdx.x + u * 1024] = pre[u];
    __syncthreads();
    double *carry = tile + CB + 2 + wv * 66;
    for (u32 i = i0; i < i1; ++i) {
        const u32 s_beg = (u32)__builtin_amdgcn_readfirstlane((int)seg[5 * i + 1]);
        const u32 s_end = (u32)__builtin_amdgcn_readfirstlane((int)seg[5 * i + 2]);
        const u32 q_beg = (u32)__builtin_amdgcn_readfirstlane((int)seg[5 * i + 3]);
        const u32 q_end = (u32)__builtin_amdgcn_readfirstlane((int)seg[5 * i + 4]);
        const bool more = i + 1 < i1;
        if (more) fetch_band((u32)__builtin_amdgcn_readfirstlane((int)seg[5 * (i + 1)]), pre);   // lands while this segment runs
        if (stamps) t_mark = wall_clock64();
        {   // ---- reduced steps
            u32 s = s_beg + wv;
            for (; s + 3 * W < s_end; s += 4 * W) {
                uint4 c[4];
                u32 b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    c[u] = scode[(size_t)(s + u * W) * 64 + lane];
                    b[u] = (u32)__builtin_amdgcn_readfirstlane((int)sbase[s + u * W]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) pbr_step(c[u], b[u], tile, carry, lane, val);
            }
            for (; s < s_end; s += W) pbr_step(scode[(size_t)s * 64 + lane], (u32)__builtin_amdgcn_readfirstlane((int)sbase[s]), tile, carry, lane, val);
        }
        if (stamps) { const unsigned long long t = wall_clock64(); t_red += t - t_mark; t_mark = t; }
        // ---- plain quads: wavefront w takes the segment's 256-quad blocks w, w + 16, ...
        for (u32 blk = q_beg + wv * 256u; blk < q_end; blk += W * 256u) {
            const u32 wend = blk + 256u < q_end ? blk + 256u : q_end;
            u32 j = blk + lane;
            for (; j + 3 * 64 < wend; j += 4 * 64) {
                uint2 c[4];
                u32 d[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    c[u] = q_lcol[j + u * 64];
                    d[u] = q_dst[j + u * 64];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    double2 lo, hi;
                    lo.x = tile[c[u].x & 0xffffu];
                    lo.y = tile[c[u].x >> 16];
                    hi.x = tile[c[u].y & 0xffffu];
                    hi.y = tile[c[u].y >> 16];
                    double2 *out = reinterpret_cast<double2 *>(val + d[u]);
                    out[0] = lo;
                    out[1] = hi;
                }
            }
            for (; j < wend; j += 64) {
                const uint2 c = q_lcol[j];
                double2 lo, hi;
                lo.x = tile[c.x & 0xffffu];
                lo.y = tile[c.x >> 16];
                hi.x = tile[c.y & 0xffffu];
                hi.y = tile[c.y >> 16];
                double2 *out = reinterpret_cast<double2 *>(val + q_dst[j]);
                out[0] = lo;
                out[1] = hi;
            }
        }
        if (stamps) t_plain += wall_clock64() - t_mark;
        if (!more) break;
        __syncthreads();                      // every wavefront is done with the band in LDS
#pragma unroll
        for (u32 u = 0; u < PRE; ++u) reinterpret_cast<double2 *>(tile)[threadIdx.x + u * 1024] = pre[u];
        __syncthreads();
    }
    if (stamps && threadIdx.x == 0) {
        stamps[4 * blockIdx.x] = t_start;
        stamps[4 * blockIdx.x + 1] = wall_clock64();
        stamps[4 * blockIdx.x + 2] = (i1 - i0) | ((unsigned long long)(i1 - i0 - 1) << 32);
        stamps[4 * blockIdx.x + 3] = t_red | (t_plain << 32);
    }
}

// The gather pass keeps a FIXED item list per workgroup instead of tickets (lists balanced by the host, longest item
// first): which workgroup adds an item's share of alpha = v . q must not change from run to run, or alpha would not be
// reproducible bit for bit.  The next record is fetched while the current item streams, the rows of the fold ahead of it.
// item record (two uint4): {begin, end, first row, rows} {slots per row, total slot or ~0, -, -}
template <u32 BLOCK>
__global__ void __launch_bounds__(BLOCK)
k_pb_gather2(const uint4 *items2, const u32 *wg_begin, const uint16_t *lslot, const double *val, double *v,
             const double *__restrict__ q_loc, double *part, double *partials, unsigned long long *stamps)
{
    unsigned long long t_start = 0;
    if (stamps) t_start = wall_clock64();
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr u32 WAVES = BLOCK / 64;
    constexpr u32 FOLD = LZX_PB_RB / BLOCK;   // rows of the fold per thread
    constexpr u32 TILE = LZX_PB_RB + 8;
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 wv = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    double *ytile = lds + (size_t)wv * TILE;
    double *wsum = lds + (size_t)WAVES * TILE;
    double dot = 0.0;
    u32 it = (u32)__builtin_amdgcn_readfirstlane((int)wg_begin[blockIdx.x]);
    const u32 n_items = (u32)__builtin_amdgcn_readfirstlane((int)wg_begin[blockIdx.x + 1]);
    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0;
    if (it < n_items) {
        r0 = items2[2 * (size_t)it];
        r1 = items2[2 * (size_t)it + 1];
    }
    while (it < n_items) {
        const u32 beg = (u32)__builtin_amdgcn_readfirstlane((int)r0.x), end = (u32)__builtin_amdgcn_readfirstlane((int)r0.y);
        const u32 row0 = (u32)__builtin_amdgcn_readfirstlane((int)r0.z), rows = (u32)__builtin_amdgcn_readfirstlane((int)r0.w);
        const u32 rep = (u32)__builtin_amdgcn_readfirstlane((int)r1.x), slot = (u32)__builtin_amdgcn_readfirstlane((int)r1.y);
        const u32 nx = it + 1;
        uint4 n0 = make_uint4(0, 0, 0, 0), n1 = n0;
        if (nx < n_items) {                                    // the next record travels while this item streams
            n0 = items2[2 * (size_t)nx];
            n1 = items2[2 * (size_t)nx + 1];
        }
        double acc = 0.0;                                      // rows == 1
        double vv[FOLD], qq[FOLD];
#pragma unroll
        for (u32 u = 0; u < FOLD; ++u) vv[u] = qq[u] = 0.0;        // this thread's rows of the fold, fetched ahead
        if (rows == 1) {
            u32 i = beg + tid;
            for (; i + 7 * BLOCK < end; i += 8 * BLOCK) {
                double a[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] = val[i + u * BLOCK];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += a[u];
            }
            for (; i < end; i += BLOCK) acc += val[i];
            acc = wave_sum_pb(acc);
            if (lane == 0) wsum[wv] = acc;
        } else {
            if (slot == 0xffffffffu) {
#pragma unroll
                for (u32 u = 0; u < FOLD; ++u) {
                    const u32 j = tid + u * BLOCK;
                    if (j < rows) {
                        vv[u] = v[row0 + j];
                        qq[u] = q_loc[row0 + j];
                    }
                }
            }
            const u32 slots = rows * rep;
            for (u32 j = lane; j < slots; j += 64) ytile[j] = 0.0;
            __builtin_amdgcn_wave_barrier();
            const u32 blocks = (end - beg) / 128u;
            // the band's tail (< 128 values), wavefront 0: fetched first, added last
            double tv[2] = {0.0, 0.0};
            u32 ts[2] = {LZX_PB_RB, LZX_PB_RB};
            if (wv == 0) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const u32 i = beg + blocks * 128u + lane + u * 64;
                    if (i < end) {
                        tv[u] = val[i];
                        ts[u] = lslot[i];
                    }
                }
            }
            u32 kb = wv;
            for (; kb + 7 * WAVES < blocks; kb += 8 * WAVES) {
                double2 av[8];
                u32 sv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const u32 p = beg + (kb + u * WAVES) * 128u + lane * 2;
                    av[u] = *reinterpret_cast<const double2 *>(val + p);
                    sv[u] = *reinterpret_cast<const u32 *>(lslot + p);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    atomicAdd(&ytile[sv[u] & 0xffffu], av[u].x);
                    atomicAdd(&ytile[sv[u] >> 16], av[u].y);
                }
            }
            {   // up to 7 more blocks of this wavefront: all fetched before the first add
                double2 av[7];
                u32 sv[7];
#pragma unroll
                for (int u = 0; u < 7; ++u) {
                    const u32 k = kb + u * WAVES;
                    if (k < blocks) {
                        const u32 p = beg + k * 128u + lane * 2;
                        av[u] = *reinterpret_cast<const double2 *>(val + p);
                        sv[u] = *reinterpret_cast<const u32 *>(lslot + p);
                    }
                }
#pragma unroll
                for (int u = 0; u < 7; ++u) {
                    const u32 k = kb + u * WAVES;
                    if (k < blocks) {
                        atomicAdd(&ytile[sv[u] & 0xffffu], av[u].x);
                        atomicAdd(&ytile[sv[u] >> 16], av[u].y);
                    }
                }
            }
            if (wv == 0) {
                atomicAdd(&ytile[ts[0]], tv[0]);
                atomicAdd(&ytile[ts[1]], tv[1]);
            }
        }
        __syncthreads();
        if (rows == 1) {
            if (tid == 0) {
                double t = 0.0;
                for (u32 w = 0; w < WAVES; ++w) t += wsum[w];
                if (slot == 0xffffffffu) {
                    v[row0] += t;
                    dot += t * q_loc[row0];
                } else {
                    part[slot] = t;
                }
            }
        } else if (slot == 0xffffffffu) {
#pragma unroll
            for (u32 u = 0; u < FOLD; ++u) {
                const u32 j = tid + u * BLOCK;
                if (j < rows) {
                    double y = 0.0;
                    for (u32 w = 0; w < WAVES; ++w)
                        for (u32 t = 0; t < rep; ++t) y += lds[(size_t)w * TILE + j * rep + t];
                    v[row0 + j] = vv[u] + y;
                    dot += y * qq[u];
                }
            }
        } else {
            for (u32 j = tid; j < rows; j += BLOCK) {
                double y = 0.0;
                for (u32 w = 0; w < WAVES; ++w)
                    for (u32 t = 0; t < rep; ++t) y += lds[(size_t)w * TILE + j * rep + t];
                part[slot + j] = y;
            }
        }
        __syncthreads();                      // tiles and wsum are free
        it = nx;
        r0 = n0;
        r1 = n1;
    }
    dot = wave_sum_pb(dot);
    __syncthreads();
    if (lane == 0) wsum[wv] = dot;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (u32 i = 0; i < WAVES; ++i) s += wsum[i];
        partials[blockIdx.x] = s;
        if (stamps) {
            stamps[4 * blockIdx.x] = t_start;
            stamps[4 * blockIdx.x + 1] = wall_clock64();
            stamps[4 * blockIdx.x + 2] = wg_begin[blockIdx.x + 1] - wg_begin[blockIdx.x];
            stamps[4 * blockIdx.x + 3] = 0;
        }
    }
}


// ---- gather pass drawn from a ticket counter (round 3 experiment; debug knob pb_gather_tickets = 1) ------------------------------------------------------------------------
// What the section stamps of the static form showed on the 10 M-vertex graph (profiles/r3_gather_stamps.txt): while a
// workgroup streams it runs at the rate of the isolated loop (11 GB/s per workgroup = 5.6 TB/s chip-wide), but only 70 % of
// its time is streaming -- 14 us per workgroup go to dependent record loads (item -> band tables) before the first value
// is requested, 16 us to the fold (a dependent read-modify-write of v behind 8 * rep serial LDS reads per row, on a
// handful of threads for the top bands, whose rows have up to 64 replicas), 7 us to the barrier in between -- and the
// longest-first static schedule ends 20 % later than its median workgroup (189 vs 152 us).  So:
//   * items are drawn from a ticket counter in longest-first order (dynamic longest-processing-time: the tail is one small
//     item long), the next ticket and the next item's record both in flight while the current item streams;
//   * one fat record per (item, wavefront) holds everything the pass needs -- no dependent table look-ups;
//   * the fold's v / q operands are requested before the streaming starts; bands with replicas are folded by all 512
//     threads (row x share of the (tile, replica) pairs, partial sums through LDS, closed in fixed order);
//   * alpha partials are per ITEM (item_dot[item], closed by k_pb_finish in item order), so which workgroup happened to
//     draw an item changes no bit of any result.
// The counter is never reset: items 0 .. 2 G - 1 (G = grid size <= n_items) are dealt statically, the counter hands out the
// rest, and a workgroup draws while its next item exists and stops at its first ticket past the end -- n_items - 2 G draws
// that find an item plus one per workgroup (n_items >= 2 G), or one per workgroup whose second static item exists: n_items
// - G draws either way, which the host adds to the base it passes to the next launch.

template <bool STAMP>
__global__ void __launch_bounds__(LZX_PB_GATHER_BLOCK, 4)   // two workgroups per CU: at most 128 VGPRs
k_pb_gather3(const uint4 *__restrict__ recs /*[n_items][8][2]: beg, end, row0, rows | rep, part slot or ~0, kind, -*/, u32 n_items, u32 *queue,
             u32 qbase, const uint16_t *lslot, const double *val, double *v, const double *__restrict__ q_loc, double *part,
             double *item_dot, unsigned long long *stamps)
{
    unsigned long long t_start = 0, t_mark = 0, t_zero = 0, t_stream = 0, t_bar = 0, t_fold = 0, n_vals = 0;
    u32 n_it = 0;
#define GSTAMP(acc) do { if (STAMP) { const unsigned long long t_ = wall_clock64(); acc += t_ - t_mark; t_mark = t_; } } while (0)
    if (STAMP) t_start = t_mark = wall_clock64();
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr u32 WAVES = LZX_PB_GATHER_BLOCK / 64;
    constexpr u32 TILE = LZX_PB_RB + 8;                  // + spare slot for padding entries
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 wv = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    double *ytile = lds + (size_t)wv * TILE;             // private to this wavefront
    double *wsum = lds + (size_t)WAVES * TILE;           // [WAVES] the item's alpha partial, by wavefront
    double *fscr = wsum + WAVES;                         // [LZX_PB_GATHER_BLOCK] fold scratch (+ the one-row sum)
    u32 *tick = reinterpret_cast<u32 *>(fscr + LZX_PB_GATHER_BLOCK);   // [2]

    // the first two items of a workgroup are its index and its index + grid size (no round trip before the first value is
    // requested: on the 1 M-vertex graph a workgroup has one item and two dependent atomics were a third of its time); the
    // counter hands out the items from 2 * grid on
    u32 cur = blockIdx.x, nxt = blockIdx.x + gridDim.x;
    const u32 qoff = qbase - 2u * gridDim.x;          // ticket = counter value - qoff
    if (cur < n_items) {
        // (records come through the scalar cache -- the array is read-only and the address wave-uniform; the compiler waits
        //  for a scalar load where it issues it, so the next record costs one scalar round trip per item instead of the static
        //  form's two dependent vector ones; attempts to keep it in flight in VGPRs ended in waits the register allocator
        //  introduced by re-using the destination registers)
        uint4 ra = recs[((size_t)cur * WAVES + wv) * 2], rb = recs[((size_t)cur * WAVES + wv) * 2 + 1];
        for (;;) {
            const bool have_next = nxt < n_items;
            // The next ticket and the next item's record travel while this item streams.  Both are issued UNCONDITIONALLY:
            // a load or a returning atomic under an `if` is waited for at the end of that `if` (the compiler cannot carry
            // an unknown number of outstanding operations across the join: the ISA showed s_waitcnt vmcnt(0) right behind
            // the atomic and the record loads, two exposed round trips per item).  So every lane of wavefront 0 adds --
            // lane 0 one to the counter (zero when no ticket is wanted), the others zero to words of their own in a dummy
            // line -- and the record is loaded from a clamped index.
            const u32 nx = have_next ? nxt : cur;
            const uint4 na = recs[((size_t)nx * WAVES + wv) * 2], nb = recs[((size_t)nx * WAVES + wv) * 2 + 1];
            const u32 beg = ra.x, end = ra.y, row0 = ra.z, rows = ra.w, rep = rb.x, pslot = rb.y, kind = rb.z;
            u32 t2 = 0xffffffffu;
            if (wv == 0) t2 = atomicAdd(lane == 0 ? queue : queue + 64 + lane, (lane == 0 && have_next) ? 1u : 0u);   // (- qbase where it is used)
            // the fold's operands of a one-item band, requested now (rep == 1: a thread folds rows tid and tid + 512); loaded
            // whether or not they will be used (clamped to the band's first row), for the same reason
            double vv[2], qq[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const u32 j = tid + u * LZX_PB_GATHER_BLOCK;
                const u32 jc = j < rows ? j : 0;
                vv[u] = v[row0 + jc];
                qq[u] = q_loc[row0 + jc];
            }
            double dot = 0.0;
            if (kind == LZX_G3_GROUP) {
                // a small band, this wavefront's own (up to eight consecutive ones per item): streamed into its tile and
                // folded by the wavefront itself -- no workgroup barrier inside
                const u32 slots = rows * rep;
                for (u32 j = lane; j < slots; j += 64) ytile[j] = 0.0;
                __builtin_amdgcn_wave_barrier();
                if (STAMP) { ++n_it; n_vals += end - beg; }
                GSTAMP(t_zero);
                const u32 blocks = (end - beg) / 128u;
                u32 kb = 0;
                for (; kb + 8 <= blocks; kb += 8) {
                    double2 av[8];
                    u32 sv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const u32 p = beg + (kb + u) * 128u + lane * 2;
                        av[u] = *reinterpret_cast<const double2 *>(val + p);
                        sv[u] = *reinterpret_cast<const u32 *>(lslot + p);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        atomicAdd(&ytile[sv[u] & 0xffffu], av[u].x);
                        atomicAdd(&ytile[sv[u] >> 16], av[u].y);
                    }
                }
                {   // up to seven more blocks and the band's tail (< 128 values): all fetched before the first add
                    double2 av[7];
                    u32 sv[7];
                    double tv[2] = {0.0, 0.0};
                    u32 ts[2] = {LZX_PB_RB, LZX_PB_RB};
#pragma unroll
                    for (int u = 0; u < 7; ++u) {
                        if (kb + u < blocks) {           // wave-uniform
                            const u32 p = beg + (kb + u) * 128u + lane * 2;
                            av[u] = *reinterpret_cast<const double2 *>(val + p);
                            sv[u] = *reinterpret_cast<const u32 *>(lslot + p);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const u32 i = beg + blocks * 128u + lane + u * 64;
                        if (i < end) {
                            tv[u] = val[i];
                            ts[u] = lslot[i];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 7; ++u) {
                        if (kb + u < blocks) {
                            atomicAdd(&ytile[sv[u] & 0xffffu], av[u].x);
                            atomicAdd(&ytile[sv[u] >> 16], av[u].y);
                        }
                    }
                    atomicAdd(&ytile[ts[0]], tv[0]);
                    __builtin_amdgcn_wave_barrier();   // the tail goes 64 consecutive values per instruction, in order
                    atomicAdd(&ytile[ts[1]], tv[1]);
                }
                __builtin_amdgcn_wave_barrier();
                GSTAMP(t_stream);
                // fold: replicas in order; four rows per lane at a time, loads before stores
                for (u32 j0 = lane; j0 < rows; j0 += 256) {
                    double vv[4], qq[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const u32 j = j0 + u * 64;
                        vv[u] = j < rows ? v[row0 + j] : 0.0;
                        qq[u] = j < rows ? q_loc[row0 + j] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const u32 j = j0 + u * 64;
                        if (j < rows) {
                            double y = 0.0;
                            for (u32 t = 0; t < rep; ++t) y += ytile[j * rep + t];
                            v[row0 + j] = vv[u] + y;
                            dot += y * qq[u];
                        }
                    }
                }
                GSTAMP(t_fold);
            } else if (kind == LZX_G3_ONE_ROW) {
                // one heavy row: plain strided sum (padding holds zeros), fixed reduction order
                if (STAMP) { ++n_it; n_vals += end - beg; }
                double acc = 0.0;
                u32 i = beg + tid;
                for (; i + 7 * LZX_PB_GATHER_BLOCK < end; i += 8 * LZX_PB_GATHER_BLOCK) {
                    double a[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) a[u] = val[i + u * LZX_PB_GATHER_BLOCK];
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc += a[u];
                }
                for (; i < end; i += LZX_PB_GATHER_BLOCK) acc += val[i];
                acc = wave_sum_pb(acc);
                if (lane == 0) fscr[wv] = acc;
                GSTAMP(t_stream);
                __syncthreads();
                if (tid == 0) {
                    double t = 0.0;
                    for (u32 w = 0; w < WAVES; ++w) t += fscr[w];
                    if (pslot == 0xffffffffu) {
                        v[row0] += t;
                        dot = t * q_loc[row0];
                    } else {
                        part[pslot] = t;
                    }
                }
                GSTAMP(t_fold);
            } else if (kind == LZX_G3_NORMAL) {
                if (STAMP) { ++n_it; n_vals += end - beg; }
                const bool into_v = pslot == 0xffffffffu;
                const u32 slots = rows * rep;
                for (u32 j = lane; j < slots; j += 64) ytile[j] = 0.0;
                __builtin_amdgcn_wave_barrier();
                GSTAMP(t_zero);
                // whole blocks of 128 values (an item begins on a block boundary of its band): lane l owns values 2 l, 2 l + 1
                // of its wavefront's blocks; eight blocks in flight per wavefront
                const u32 blocks = (end - beg) / 128u;
                u32 kb = wv;
                for (; kb + 7 * WAVES < blocks; kb += 8 * WAVES) {
                    double2 av[8];
                    u32 sv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const u32 p = beg + (kb + u * WAVES) * 128u + lane * 2;
                        av[u] = *reinterpret_cast<const double2 *>(val + p);
                        sv[u] = *reinterpret_cast<const u32 *>(lslot + p);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        atomicAdd(&ytile[sv[u] & 0xffffu], av[u].x);
                        atomicAdd(&ytile[sv[u] >> 16], av[u].y);
                    }
                }
                {   // up to seven more blocks of this wavefront and (wavefront 0) the band's tail of < 128 values: ALL requested
                    // before the first add -- one block per round trip here was 3.5 serial round trips per item on average,
                    // a sixth of an item's streaming time on the 10 M-vertex graph
                    double2 av[7];
                    u32 sv[7];
                    double tv[2] = {0.0, 0.0};
                    u32 ts[2] = {LZX_PB_RB, LZX_PB_RB};
#pragma unroll
                    for (int u = 0; u < 7; ++u) {
                        if (kb + u * WAVES < blocks) {           // wave-uniform
                            const u32 p = beg + (kb + u * WAVES) * 128u + lane * 2;
                            av[u] = *reinterpret_cast<const double2 *>(val + p);
                            sv[u] = *reinterpret_cast<const u32 *>(lslot + p);
                        }
                    }
                    if (wv == 0) {
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const u32 i = beg + blocks * 128u + lane + u * 64;
                            if (i < end) {
                                tv[u] = val[i];
                                ts[u] = lslot[i];
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 7; ++u) {
                        if (kb + u * WAVES < blocks) {
                            atomicAdd(&ytile[sv[u] & 0xffffu], av[u].x);
                            atomicAdd(&ytile[sv[u] >> 16], av[u].y);
                        }
                    }
                    if (wv == 0) {
                        atomicAdd(&ytile[ts[0]], tv[0]);
                        __builtin_amdgcn_wave_barrier();   // the tail goes 64 consecutive values per instruction, in order
                        atomicAdd(&ytile[ts[1]], tv[1]);
                    }
                }
                GSTAMP(t_stream);
                __syncthreads();
                GSTAMP(t_bar);
                if (rep == 1) {
                    // a thread per row: the eight wavefront tiles in order
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const u32 j = tid + u * LZX_PB_GATHER_BLOCK;
                        if (j < rows) {
                            double y = 0.0;
#pragma unroll
                            for (u32 w = 0; w < WAVES; ++w) y += lds[(size_t)w * TILE + j];
                            if (into_v) {
                                v[row0 + j] = vv[u] + y;
                                dot += y * qq[u];
                            } else {
                                part[pslot + j] = y;
                            }
                        }
                    }
                } else {
                    // rows with replicas (rows * rep <= 1024, so rows <= 512): all threads share the 8 * rep (tile, replica)
                    // pairs of every row -- thread = (row, share); shares closed per row in share order
                    u32 rows_p = 1;
                    while (rows_p < rows) rows_p <<= 1;
                    const u32 shares = LZX_PB_GATHER_BLOCK / rows_p;
                    const u32 row = tid & (rows_p - 1u), share = tid / rows_p;
                    if (row < rows) {
                        double sacc = 0.0;
                        u32 w = share / rep, t = share % rep;
                        while (w < WAVES) {
                            sacc += lds[(size_t)w * TILE + row * rep + t];
                            t += shares;
                            while (t >= rep) { t -= rep; ++w; }
                        }
                        fscr[share * rows_p + row] = sacc;
                    }
                    __syncthreads();
                    if (tid < rows) {
                        double y = 0.0;
                        for (u32 sh = 0; sh < shares; ++sh) y += fscr[sh * rows_p + tid];
                        if (into_v) {
                            const double qv = q_loc[row0 + tid];
                            v[row0 + tid] += y;
                            dot += y * qv;
                        } else {
                            part[pslot + tid] = y;
                        }
                    }
                }
                GSTAMP(t_fold);
            }
            // the item's alpha partial: wavefronts in order; the next ticket travels with the same barrier
            dot = wave_sum_pb(dot);
            if (lane == 0) wsum[wv] = dot;
            if (tid == 0) tick[0] = have_next ? t2 - qoff : 0xffffffffu;
            __syncthreads();
            double sdot = 0.0;
            if (tid == 0)
                for (u32 w = 0; w < WAVES; ++w) sdot += wsum[w];
            const u32 t2_all = (u32)__builtin_amdgcn_readfirstlane((int)tick[0]);
            __syncthreads();                  // tiles, wsum, fscr and tick are free again
            if (tid == 0) item_dot[cur] = sdot;   // (behind the barrier: a barrier waits for the stores before it)
            if (!have_next) break;
            cur = nxt;
            nxt = t2_all;
            ra = na;
            rb = nb;
        }
    }
    if (STAMP && tid == 0) {
        unsigned long long *o = stamps + 8 * (size_t)blockIdx.x;
        o[0] = t_start; o[1] = wall_clock64(); o[2] = t_zero; o[3] = t_stream; o[4] = t_bar; o[5] = t_fold; o[6] = n_it; o[7] = n_vals;
    }
#undef GSTAMP
}

}  // namespace

int lzx_pbdbg_ablate()
{
    static const int ablate = getenv("LZX_ABLATE") ? atoi(getenv("LZX_ABLATE")) : 0;
    return ablate;
}

// Static scatter schedule: the scatter order (band by band: a band's steps, then its quads) is cut into `groups`
// stretches of equal cost -- bytes read + written: per step its 1 KiB of codes + 8 B per piece, per quad 12 B + 32 B --
// one per workgroup; a stretch is stored as segments {band, steps, quads}, one per band it touches.  Two schedules:
// the bands of chunk 0 of the exchange (all of them without the two-chunk exchange) and the rest.
int lzx_pbdbg_segments(lzx_ctx *c, hipStream_t st, const std::vector<u32> &sstart, const std::vector<u32> &qstart, u32 nb, u32 nsteps)
{
    std::vector<u32> cnt(nsteps, 0u);
    if (nsteps) {
        u32 *d_cnt = nullptr;
        LZX_TRY(pb_alloc(&d_cnt, nsteps));
        hipLaunchKernelGGL(k_pbrd_count, dim3(nsteps), dim3(64), 0, st, c->d_pbr_code, d_cnt);
        hipError_t e = hipMemcpyAsync(cnt.data(), d_cnt, sizeof(u32) * nsteps, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        (void)hipFree(d_cnt);
        LZX_HIP(e);
    }
    const u32 per_cu = c->pb_cb == 8192 ? 2u : 1u;
    const u32 groups_max = (u32)c->cu_count * per_cu;
    u32 b_split = nb;   // first band of the second schedule
    if (c->overlap) {
        const u64 chunk0_end = (u64)c->world * c->xs0;
        b_split = 0;
        while (b_split < nb && ((u64)b_split + 1) * c->pb_cb <= chunk0_end) ++b_split;
    }
    std::vector<u32> segs, begin;
    auto band_steps = [&](u32 b, u32 &s0, u32 &s1) { s0 = sstart.empty() ? 0 : sstart[b]; s1 = sstart.empty() ? 0 : sstart[b + 1]; };
    auto band_quads = [&](u32 b, u32 &q0, u32 &q1) { q0 = qstart.empty() ? 0 : qstart[b]; q1 = qstart.empty() ? 0 : qstart[b + 1]; };
    constexpr u64 QUAD_COST = 44;
    auto step_cost = [&](u32 s) { return 1024ull + 8ull * cnt[s] + 64ull; };
    for (int part = 0; part < 2; ++part) {
        const u32 b0 = part == 0 ? 0 : b_split, b1 = part == 0 ? b_split : nb;
        u64 total = 0;
        for (u32 b = b0; b < b1; ++b) {
            u32 s0, s1, q0, q1;
            band_steps(b, s0, s1);
            band_quads(b, q0, q1);
            for (u32 s = s0; s < s1; ++s) total += step_cost(s);
            total += (u64)(q1 - q0) * QUAD_COST;
        }
        const u32 groups = total ? (u32)std::min<u64>(groups_max, std::max<u64>(1, total / 65536)) : 0;
        c->pb_seg_groups[part] = groups;
        c->pb_seg_first[part] = (u32)begin.size();
        if (!groups) { begin.push_back((u32)(segs.size() / 5)); continue; }
        u64 done = 0;
        u32 g = 0;   // current group; its share ends at total * (g + 1) / groups
        begin.push_back((u32)(segs.size() / 5));
        auto limit = [&]() { return total * (u64)(g + 1) / groups; };
        auto close_group = [&]() {
            while (g + 1 < groups && done >= limit()) {
                ++g;
                begin.push_back((u32)(segs.size() / 5));
            }
        };
        for (u32 b = b0; b < b1; ++b) {
            u32 s0, s1, q0, q1;
            band_steps(b, s0, s1);
            band_quads(b, q0, q1);
            u32 s = s0, q = q0;
            while (s < s1 || q < q1) {
                // the part of this band that still fits the current group: steps first, then quads
                u32 se = s;
                while (se < s1 && (g + 1 == groups || done < limit())) done += step_cost(se++);
                u32 qe = q;
                if (se == s1 && q < q1) {
                    if (g + 1 == groups) {
                        done += (u64)(q1 - q) * QUAD_COST;
                        qe = q1;
                    } else if (done < limit()) {
                        const u64 room = limit() - done;
                        const u32 take = (u32)std::min<u64>(q1 - q, (room + QUAD_COST - 1) / QUAD_COST);
                        done += (u64)take * QUAD_COST;
                        qe = q + take;
                    }
                }
                if (se > s || qe > q) {
                    segs.push_back(b); segs.push_back(s); segs.push_back(se); segs.push_back(q); segs.push_back(qe);
                }
                s = se;
                q = qe;
                close_group();
            }
        }
        while (g + 1 < groups) {   // groups the rounding left empty
            ++g;
            begin.push_back((u32)(segs.size() / 5));
        }
        begin.push_back((u32)(segs.size() / 5));   // end of the last group of this part
    }
    LZX_TRY(pb_alloc(&c->d_pb_seg, segs.size()));
    LZX_TRY(pb_alloc(&c->d_pb_seg_begin, begin.size()));
    if (!segs.empty()) LZX_HIP(hipMemcpyAsync(c->d_pb_seg, segs.data(), sizeof(u32) * segs.size(), hipMemcpyHostToDevice, st));
    LZX_HIP(hipMemcpyAsync(c->d_pb_seg_begin, begin.data(), sizeof(u32) * begin.size(), hipMemcpyHostToDevice, st));
    LZX_HIP(hipStreamSynchronize(st));
    return LZX_OK;
}


int lzx_pbdbg_persist_records(lzx_ctx *c, hipStream_t st, const std::vector<u32> &items, const std::vector<u32> &row0, const std::vector<u32> &rep)
{   // records of the persistent gather pass: everything an item needs in one place.  Items are dealt to the
        // workgroups here, longest first, each to the workgroup with the least work so far (cost = bytes streamed + a
        // fixed share for the fold), and laid out workgroup by workgroup.
        const u32 G = c->pb_gather_grid;
        std::vector<u32> order(c->pb_n_items);
        for (u32 i = 0; i < c->pb_n_items; ++i) order[i] = i;
        auto cost = [&](u32 i) {
            const u32 R = items[4 * (size_t)i];
            return 10ull * (items[4 * (size_t)i + 2] - items[4 * (size_t)i + 1]) + 16ull * (row0[R + 1] - row0[R]) + 24000ull;
        };
        std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b) { return cost(a) > cost(b); });
        std::vector<std::vector<u32>> lists(G);
        std::priority_queue<std::pair<u64, u32>, std::vector<std::pair<u64, u32>>, std::greater<std::pair<u64, u32>>> heap;
        for (u32 w = 0; w < G; ++w) heap.push({0ull, w});
        for (u32 i : order) {
            auto [load, w] = heap.top();
            heap.pop();
            lists[w].push_back(i);
            heap.push({load + cost(i), w});
        }
        std::vector<u32> rec((size_t)c->pb_n_items * 8, 0u), begin((size_t)G + 1, 0u);
        size_t k = 0;
        for (u32 w = 0; w < G; ++w) {
            begin[w] = (u32)k;
            for (u32 i : lists[w]) {
                const u32 *it = &items[4 * (size_t)i];
                const u32 R = it[0];
                u32 *o = &rec[8 * k++];
                o[0] = it[1]; o[1] = it[2]; o[2] = row0[R]; o[3] = row0[R + 1] - row0[R]; o[4] = rep[R]; o[5] = it[3];
            }
        }
        begin[G] = (u32)k;
        LZX_TRY(pb_alloc(&c->d_pb_items2, rec.size()));
        LZX_TRY(pb_alloc(&c->d_pb_wg_begin, begin.size()));
        if (!rec.empty()) LZX_HIP(hipMemcpyAsync(c->d_pb_items2, rec.data(), sizeof(u32) * rec.size(), hipMemcpyHostToDevice, st));
        LZX_HIP(hipMemcpyAsync(c->d_pb_wg_begin, begin.data(), sizeof(u32) * begin.size(), hipMemcpyHostToDevice, st));
        if (c->pb_stamps_opt > 0) {   // debug library: per-workgroup time stamps of the persistent passes
            LZX_TRY(pb_alloc(&c->d_pb_stamps, 3 * 4096));
            LZX_HIP(hipMemsetAsync(c->d_pb_stamps, 0, sizeof(unsigned long long) * 3 * 4096, st));
        }
        LZX_TRY(pb_alloc(&c->d_pb_queue, 4));
        LZX_HIP(hipMemsetAsync(c->d_pb_queue, 0, sizeof(u32) * 4, st));
        for (u32 &b : c->pb_qbase) b = 0;
        LZX_HIP(hipStreamSynchronize(st));
    return LZX_OK;
}

int lzx_pbdbg_g3_records(lzx_ctx *c, hipStream_t st, std::vector<u32> &g3_out, std::vector<u64> &g3_cost, const std::vector<u32> &items,
                         const std::vector<u32> &row0, const std::vector<u32> &rep, const std::vector<u32> &rstart)
{
        // fat records, longest item first: {beg, end, row0, rows | rep, part slot or ~0, kind, bands} per (item, wavefront)
        if (g3_out.empty()) {     // no grouping: the items as they were made (items[] is still in that form)
            for (size_t i = 0; i + 3 < items.size(); i += 4) {
                if (items[i + 3] == LZX_PB_ITEM_NONE || items[i + 3] == LZX_PB_ITEM_GROUP) continue;
                const u32 R = items[i];
                g3_out.insert(g3_out.end(), items.begin() + i, items.begin() + i + 4);
                g3_cost.push_back(10ull * (items[i + 2] - items[i + 1]) + 24ull * (row0[R + 1] - row0[R]) + 40000ull);
            }
        }
        const size_t no = g3_cost.size();
        std::vector<u32> order(no);
        for (size_t i = 0; i < no; ++i) order[i] = (u32)i;
        std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b2) { return g3_cost[a] > g3_cost[b2]; });
        std::vector<u32> recs(no * 8 * 8, 0u);
        for (size_t k = 0; k < no; ++k) {
            const u32 *it = &g3_out[4 * (size_t)order[k]];
            for (u32 w = 0; w < 8; ++w) {
                u32 *o = &recs[(k * 8 + w) * 8];
                if (it[3] == LZX_PB_ITEM_GROUP) {
                    if (w < it[1]) {
                        const u32 R = it[0] + w;
                        o[0] = rstart[R]; o[1] = rstart[R + 1]; o[2] = row0[R]; o[3] = row0[R + 1] - row0[R];
                        o[4] = rep[R]; o[5] = 0xffffffffu; o[6] = LZX_G3_GROUP; o[7] = it[1];
                    } else {
                        o[6] = LZX_G3_IDLE;
                    }
                } else {
                    const u32 R = it[0], rows = row0[R + 1] - row0[R];
                    o[0] = it[1]; o[1] = it[2]; o[2] = row0[R]; o[3] = rows;
                    o[4] = rep[R]; o[5] = it[3]; o[6] = rows == 1 ? LZX_G3_ONE_ROW : LZX_G3_NORMAL; o[7] = 1;
                }
            }
        }
        c->pb_g3_items = (u32)no;
        c->pb_gather_grid = std::min<u32>((u32)c->cu_count * 2u, std::max<u32>(1u, (u32)no));
        LZX_TRY(pb_alloc(reinterpret_cast<u32 **>(&c->d_pb_grec), recs.size()));
        if (!recs.empty()) LZX_HIP(hipMemcpyAsync(c->d_pb_grec, recs.data(), sizeof(u32) * recs.size(), hipMemcpyHostToDevice, st));
        LZX_TRY(pb_alloc(&c->d_pb_item_dot, (u64)no));
        LZX_HIP(hipMemsetAsync(c->d_pb_item_dot, 0, sizeof(double) * std::max<size_t>(no, 1), st));
        LZX_TRY(pb_alloc(&c->d_pb_gqueue, 128));   // word 0: the ticket counter; words 64 ..: the dummy line (see k_pb_gather3)
        LZX_HIP(hipMemsetAsync(c->d_pb_gqueue, 0, sizeof(u32) * 128, st));
        c->pb_gq_base = 0;
        LZX_HIP(hipStreamSynchronize(st));
    return LZX_OK;
}

bool lzx_pbdbg_scatter(lzx_ctx *c, u32 u0, u32 u1, const double *x, int *rc)
{
    *rc = LZX_OK;
    const int ablate = lzx_pbdbg_ablate();
    const bool persistent = c->pb_persist_opt > 0 && !ablate;
    const bool fixed = persistent && c->pb_persist_opt != 2;   // 2 = the ticket-driven form
    if (!ablate && !persistent) return false;
    const size_t lds1 = ((size_t)c->pb_cb + 2 + 16 * 66) * sizeof(double), lds1p = lds1 + 16;
    auto fail = [&](hipError_t e) { lzx_set_error("lzx_pb_dbg: %s", hipGetErrorString(e)); *rc = LZX_ERR_HIP; return true; };
    if (ablate) {
        if (u1 <= u0) return true;
        auto kern = c->pb_cb == 8192 ? k_pb_scatter_abl<8192> : k_pb_scatter_abl<LZX_PB_CB>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        if (e != hipSuccess) return fail(e);
        hipLaunchKernelGGL(kern, dim3(u1 - u0), dim3(1024), lds1, c->stream, c->d_pb_unit + 5 * (size_t)u0, c->d_pbr_code,
                           c->d_pbr_base, reinterpret_cast<const uint2 *>(c->d_pb_lcol), c->d_pb_dst, x, c->xlen, c->d_pb_val, ablate);
        return true;
    }
    if (fixed) {
        // schedule 0 = the bands of chunk 0 (units [0, pb_units0)), schedule 1 = the rest; a call for all units runs both
        auto kern3 = c->pb_cb == 8192 ? k_pb_scatter3<8192> : k_pb_scatter3<LZX_PB_CB>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1p);
        if (e != hipSuccess) return fail(e);
        for (int part = 0; part < 2; ++part) {
            const bool wanted = part == 0 ? u0 == 0 : u1 == c->pb_units && c->pb_units0 < c->pb_units;
            const u32 grid = c->pb_seg_groups[part];
            if (!wanted || !grid) continue;
            hipLaunchKernelGGL(kern3, dim3(grid), dim3(1024), lds1p, c->stream, c->d_pb_seg, c->d_pb_seg_begin + c->pb_seg_first[part],
                               c->d_pbr_code, c->d_pbr_base, reinterpret_cast<const uint2 *>(c->d_pb_lcol), c->d_pb_dst, x, c->xlen,
                               c->d_pb_val, c->d_pb_stamps ? c->d_pb_stamps + 4096 * part : nullptr);
        }
        return true;
    }
    if (u1 <= u0) return true;
    auto kern2 = c->pb_cb == 8192 ? k_pb_scatter2<8192> : k_pb_scatter2<LZX_PB_CB>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1p);
    if (e != hipSuccess) return fail(e);
    const u32 q = u0 == 0 ? 0u : 1u, n = u1 - u0;
    const u32 grid = std::min<u32>(n, (u32)c->cu_count * (c->pb_cb == 8192 ? 2u : 1u));
    hipLaunchKernelGGL(kern2, dim3(grid), dim3(1024), lds1p, c->stream, c->d_pb_unit + 5 * (size_t)u0, n, c->d_pb_queue + q,
                       c->pb_qbase[q], c->d_pbr_code, c->d_pbr_base, reinterpret_cast<const uint2 *>(c->d_pb_lcol),
                       c->d_pb_dst, x, c->xlen, c->d_pb_val, c->d_pb_stamps ? c->d_pb_stamps + 4096 * q : nullptr);
    c->pb_qbase[q] += n + grid;
    return true;
}

bool lzx_pbdbg_gather(lzx_ctx *c, double *v, const double *q_loc, double *partials, int *rc)
{
    *rc = LZX_OK;
    const bool persistent = c->pb_persist_opt > 0 && !lzx_pbdbg_ablate();
    auto fail = [&](hipError_t e) { lzx_set_error("lzx_pb_dbg: %s", hipGetErrorString(e)); *rc = LZX_ERR_HIP; return true; };
    if (persistent) {
        const u32 block = c->pb_gather_block;
        const size_t lds2p = ((size_t)(block / 64) * (LZX_PB_RB + 8) + block / 64) * sizeof(double) + 16;
        auto gk = block == 256 ? k_pb_gather2<256> : k_pb_gather2<512>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2p);
        if (e != hipSuccess) return fail(e);
        hipLaunchKernelGGL(gk, dim3(c->pb_gather_grid), dim3(block), lds2p, c->stream,
                           reinterpret_cast<const uint4 *>(c->d_pb_items2), c->d_pb_wg_begin, c->d_pb_lrow, c->d_pb_val, v, q_loc,
                           c->d_pb_part, partials, c->d_pb_stamps ? c->d_pb_stamps + 8192 : nullptr);
        return true;
    }
    if (!c->pb_g3) return false;
    const size_t lds2 = ((size_t)(LZX_PB_GATHER_BLOCK / 64) * (LZX_PB_RB + 8) + LZX_PB_GATHER_BLOCK / 64) * sizeof(double);
    const size_t lds3 = lds2 + ((size_t)LZX_PB_GATHER_BLOCK + 2) * sizeof(double);   // + fold scratch and the two ticket words
    const bool stamped = c->pb_stamps_opt > 0 && c->d_pb_gstamps;
    auto gk = stamped ? k_pb_gather3<true> : k_pb_gather3<false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
    if (e != hipSuccess) return fail(e);
    hipLaunchKernelGGL(gk, dim3(c->pb_gather_grid), dim3(LZX_PB_GATHER_BLOCK), lds3, c->stream,
                       reinterpret_cast<const uint4 *>(c->d_pb_grec), c->pb_g3_items, c->d_pb_gqueue, c->pb_gq_base, c->d_pb_lrow,
                       c->d_pb_val, v, q_loc, c->d_pb_part, c->d_pb_item_dot, stamped ? c->d_pb_gstamps : nullptr);
    c->pb_gq_base += c->pb_g3_items - c->pb_gather_grid;   // what the launch advances the ticket counter by (k_pb_gather3)
    return true;
}
#endif  // LZX_DEBUG_KNOBS

// lzx_spmv_body.h -- the SpMV on the sliced-ELL body + split rows (k_spmv of lzx_kernels.hip) as a device function, so
// that lzx_pb.hip can run the staged-columns part of the blocked SpMV in the SAME launch as its scatter pass (the last
// workgroups of that grid: they fill the scatter pass's tail instead of waiting for it to drain).
#pragma once
#include <type_traits>
#include "lzx_internal.h"

// --------------------------------------------------------------------------------------------------
// reductions: fixed shape, no atomics
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;  // every lane holds the same total
}

// --------------------------------------------------------------------------------------------------
// SpMV on the sliced-ELL body + split rows, fused with the alpha partial.
struct SpmvArgs {
    const u32 *sell_cols;
    const u64 *slice_off;
    const u32 *slice_w;
    u32 n_slices;  // slices the general loop takes: entries [0, n_slices) of slice_perm
    const u32 *slice_perm;   // processing order (blocked mode: wide slices, then the 8-, 4- and 0-code ones; identity otherwise)
    u32 ns_w8, ns_w4, ns_w0; // blocked mode: how many of each narrow class follow in slice_perm
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
    u32 n_zero;    // blocked mode: v[0 .. n_zero) (the split rows) starts at 0; the blocked passes and k_pb_finish add to it
    u32 deep;      // blocked mode: slices pipelined four deep instead of two
    u32 burst;     // staging: eight loads per thread in flight (1) or one per loop iteration (0)
};

// Column code c: c < hub -> value staged in LDS slot c; otherwise x[c - hub].
// HUB: 0 = nothing staged, 1 = both kinds of code, 2 = staged codes only (the other entries went to the blocked
// passes): no global gather is compiled in, so the summing loops wait on LDS alone and never drain the index prefetch.
template <int HUB>
__device__ __forceinline__ double gather(u32 c, const double *__restrict__ x, const double *hubv, u32 hub)
{
#ifdef LZX_ABL_NOLDS   // ablation build (tools/): the staged value is not looked up
    if (HUB == 2) return (double)c;
#endif
    if (HUB == 2) return hubv[c];
    if (HUB) {
        if (c < hub) return hubv[c];
        return x[c - hub];
    }
    return x[c];
}

// value held by lane l (wave-uniform l) of a per-lane register
__device__ __forceinline__ u32 lane_u32(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ u64 lane_u64(u64 v, u32 l)
{
    return ((u64)lane_u32((u32)(v >> 32), l) << 32) | lane_u32((u32)v, l);
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

// staged-only tables (HUB == 2) hold 16-bit codes: 8-byte packets of four
template <bool NT>
__device__ __forceinline__ uint4 load_idx4(const uint2 *p)
{
    uint2 r;
    if (NT) {
        r.x = __builtin_nontemporal_load(&p->x);
        r.y = __builtin_nontemporal_load(&p->y);
    } else {
        r = *p;
    }
    return make_uint4(r.x & 0xffffu, r.x >> 16, r.y & 0xffffu, r.y >> 16);
}
// the packet as it travels (blocked mode: 8 bytes = two registers) and its four codes, unpacked at the point of use:
// twice as many packets fit the registers of a software pipeline
template <bool NT>
__device__ __forceinline__ uint4 load_raw(const uint4 *p) { return load_idx4<NT>(p); }
template <bool NT>
__device__ __forceinline__ uint2 load_raw(const uint2 *p)
{
#ifdef LZX_ABL_NOLOAD   // ablation build (tools/): the packet is not loaded
    return make_uint2((u32)(size_t)p & 0x3fff3fffu, ((u32)(size_t)p >> 3) & 0x3fff3fffu);
#endif
    uint2 r;
    if (NT) {
        r.x = __builtin_nontemporal_load(&p->x);
        r.y = __builtin_nontemporal_load(&p->y);
    } else {
        r = *p;
    }
    return r;
}
__device__ __forceinline__ uint4 codes_of(const uint4 &r) { return r; }
__device__ __forceinline__ uint4 codes_of(const uint2 &r) { return make_uint4(r.x & 0xffffu, r.x >> 16, r.y & 0xffffu, r.y >> 16); }
template <int HUB> struct CodeTable { using packet = uint4; using code = u32; };
template <> struct CodeTable<2> { using packet = uint2; using code = uint16_t; };

// block / nblocks: this workgroup's place among the workgroups that share the work (a launch of its own: blockIdx.x, gridDim.x)
template <int HUB, bool NT>
__device__ __forceinline__ void spmv_body(const SpmvArgs &a, const u32 block, const u32 nblocks)
{
    using PK = typename CodeTable<HUB>::packet;
    using CODE = typename CodeTable<HUB>::code;
    const CODE *long_cols = reinterpret_cast<const CODE *>(a.long_cols);
    const CODE *sell_cols = reinterpret_cast<const CODE *>(a.sell_cols);
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *hubv = lds;
    double *wsum = lds + a.hub;  // 16 doubles behind the staged entries

    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 wv = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));

    if (HUB) {
        // Stage x of the `hub` highest-degree vertices once per workgroup (coalesced at world == 1;
        // `world` strided segments otherwise: degree rank r lives at (r % world) * xs0 + r / world, in chunk 0).
        // Staging is dead time for the CU (one workgroup fits beside the tile): every thread issues eight loads before
        // its first LDS write, so 16 Ki values cost one or two memory round trips instead of sixteen (a load per loop
        // iteration, each waited for, was 20 us of the 110 us this kernel took on the 10 M-vertex graph).
        if (!a.burst) {
            for (u32 i = tid; i < a.hub_real; i += LZX_SPMV_BLOCK) {
                const u32 g = (a.world == 1) ? i : (i % a.world) * a.xs0 + i / a.world;
                hubv[i] = a.x[g];
            }
        } else if (a.world == 1) {
            const u32 pairs = a.hub_real >> 1;               // hub_real is even; x is 16-byte aligned
            const double2 *src = reinterpret_cast<const double2 *>(a.x);
            // clamped index, unconditional load AND store (a store under `if` pulls its load into the branch, and the
            // round trips are serial again); threads past the end rewrite the last pair with its own value
            auto stage = [&](auto depth_tag) {
                constexpr u32 D = decltype(depth_tag)::value;   // 16-byte loads per thread in flight
                for (u32 i0 = 0; i0 < pairs; i0 += D * LZX_SPMV_BLOCK) {
                    double2 t[D];
#pragma unroll
                    for (u32 u = 0; u < D; ++u) {
                        const u32 i = i0 + tid + u * LZX_SPMV_BLOCK;
                        t[u] = src[i < pairs ? i : pairs - 1];
                    }
#pragma unroll
                    for (u32 u = 0; u < D; ++u) {
                        const u32 i = i0 + tid + u * LZX_SPMV_BLOCK;
                        reinterpret_cast<double2 *>(hubv)[i < pairs ? i : pairs - 1] = t[u];
                    }
                }
            };
            if (a.burst >= 8) stage(std::integral_constant<u32, 8>{});
            else if (a.burst >= 4) stage(std::integral_constant<u32, 4>{});
            else stage(std::integral_constant<u32, 2>{});
        } else {
            for (u32 i0 = 0; i0 < a.hub_real; i0 += 8 * LZX_SPMV_BLOCK) {
                double t[8];
#pragma unroll
                for (u32 u = 0; u < 8; ++u) {
                    const u32 i = i0 + tid + u * LZX_SPMV_BLOCK;
                    const u32 j = i < a.hub_real ? i : a.hub_real - 1;
                    t[u] = a.x[(j % a.world) * a.xs0 + j / a.world];
                }
#pragma unroll
                for (u32 u = 0; u < 8; ++u) {
                    const u32 i = i0 + tid + u * LZX_SPMV_BLOCK;
                    hubv[i < a.hub_real ? i : a.hub_real - 1] = t[u];
                }
            }
        }
        for (u32 i = a.hub_real + tid; i < a.hub; i += LZX_SPMV_BLOCK) hubv[i] = 0.0;
        __syncthreads();
    }

    for (u32 i = block * LZX_SPMV_BLOCK + tid; i < a.n_zero; i += nblocks * LZX_SPMV_BLOCK) a.v[i] = 0.0;

    const u32 waves = nblocks * (LZX_SPMV_BLOCK / 64);
    const u32 w0 = block * (LZX_SPMV_BLOCK / 64) + wv;   // scalar: wv came through readfirstlane

    // Units (split-row items, then 64-row slices) are dealt to wavefronts round-robin, unit i of a wavefront being
    // w0 + i * waves: neighbouring wavefronts stream neighbouring memory, and because widths fall monotonically
    // (and are capped by the split-row threshold) every wavefront gets the same work.
    // Both loops are software pipelines built so that hipcc can wait with exact vmcnt(N) counts; after the hub
    // split leaves most units only a few packets long the kernel is otherwise bound by memory round trips, not by
    // bandwidth (measured 0.25 ms for 0.63 GB of indices on the 10 M-vertex graph).  Three rules:
    //   * descriptors: one load fetches the descriptors of the wavefront's next 64 units, one per lane; each unit
    //     then reads its own with v_readlane, so no unit waits for a descriptor round trip;
    //   * packets: the first packets of unit i+1 are in flight while unit i is summed, in two register sets used
    //     alternately (a rotation by register moves would have to wait for the loads it moves);
    //   * every pipelined load is issued unconditionally from a clamped, always valid address: loads under `if`s
    //     make the number in flight unknown to the compiler, which then drains the queue (vmcnt(0)) at every use.

    // ---- split rows: one wavefront sums one item of <= LZX_ITEM entries, lanes striding 16-byte index
    //      packets; the item totals are combined in row order by k_long_finish.
    {
        const u32 mine = a.n_items > w0 ? (a.n_items - w0 + waves - 1) / waves : 0;
        for (u32 base = 0; base < mine; base += 64) {
            const u32 cnt = mine - base < 64 ? mine - base : 64;
            const u32 di = w0 + (base + (lane < cnt ? lane : 0)) * waves;
            const u64 d_beg = a.item_beg[di];
            const u32 d_pk = a.item_len[di] >> 2;
            // IPF packets of an item (per lane) are in flight ahead of its summation, in two register sets (2 -> 4 -> 8:
            // the split-row part of the 10 M-vertex graph 33 -> 28 -> .. us: it is bound by bytes in flight per wavefront)
            constexpr int IPF = HUB == 2 ? 8 : 4;
            auto issue = [&](u32 j, PK (&kk)[IPF]) {
                const PK *p = reinterpret_cast<const PK *>(long_cols + lane_u64(d_beg, j));
                const u32 packets = lane_u32(d_pk, j);
#pragma unroll
                for (int u = 0; u < IPF; ++u) kk[u] = load_raw<NT>(p + (lane + 64u * u < packets ? lane + 64u * u : 0));
            };
            auto add4 = [&](const PK &raw, double &acc) {
                const uint4 k = codes_of(raw);
                const double x0 = gather<HUB>(k.x, a.x, hubv, a.hub);
                const double x1 = gather<HUB>(k.y, a.x, hubv, a.hub);
                const double x2 = gather<HUB>(k.z, a.x, hubv, a.hub);
                const double x3 = gather<HUB>(k.w, a.x, hubv, a.hub);
                acc += x0; acc += x1; acc += x2; acc += x3;
            };
            auto consume = [&](u32 j, const PK (&kk)[IPF]) {
                const PK *p = reinterpret_cast<const PK *>(long_cols + lane_u64(d_beg, j));
                const u32 packets = lane_u32(d_pk, j);
                double acc = 0.0;
#pragma unroll
                for (int u = 0; u < IPF; ++u)
                    if (lane + 64u * u < packets) add4(kk[u], acc);
                u32 q = lane + 64u * IPF;
                for (; q + 64u * 3 < packets; q += 64u * 4) {
                    PK c[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) c[u] = load_raw<NT>(p + q + 64u * u);
#pragma unroll
                    for (int u = 0; u < 4; ++u) add4(c[u], acc);
                }
                for (; q < packets; q += 64) {
                    const PK c = load_raw<NT>(p + q);
                    add4(c, acc);
                }
                acc = wave_sum(acc);
                if (lane == 0) a.long_partial[w0 + (base + j) * waves] = acc;
            };
            PK ka[IPF], kb[IPF];
            issue(0, ka);
            for (u32 j = 0; j < cnt; j += 2) {
                issue(j + 1 < cnt ? j + 1 : j, kb);
                consume(j, ka);
                if (j + 1 >= cnt) break;
                issue(j + 2 < cnt ? j + 2 : j + 1, ka);
                consume(j + 1, kb);
            }
        }
    }

    // ---- sliced-ELL body: one wavefront per slice of 64 rows, lane = row.  Each lane adds its row's
    //      entries one at a time in the caller's column order: the same left-to-right sum as the
    //      reference's spMV (serial/lib/SPMV.cc:24-27), so these rows come out bit-identical to it.
    double dot = 0.0;
    {
        const u32 mine = a.n_slices > w0 ? (a.n_slices - w0 + waves - 1) / waves : 0;
        for (u32 base = 0; base < mine; base += 64) {
            const u32 cnt = mine - base < 64 ? mine - base : 64;
            const u32 di = w0 + (base + (lane < cnt ? lane : 0)) * waves;
            const u32 d_sid = a.slice_perm[di];    // which 64 rows
            const u64 d_off = a.slice_off[d_sid];
            const u32 d_st = a.slice_w[d_sid] >> 2;   // packets per lane
            // PF packets of a slice are fetched ahead of its summation; beyond them a wide slice streams GP packets at a time
            constexpr int PF = 4;
            constexpr int GP = HUB == 2 ? 8 : 4;
            auto issue = [&](u32 j, PK (&f)[PF], double &qrow) {
                const PK *p = reinterpret_cast<const PK *>(sell_cols + lane_u64(d_off, j)) + lane;
                const u32 st = lane_u32(d_st, j);
#pragma unroll
                for (int u = 0; u < PF; ++u) f[u] = load_raw<NT>(p + (size_t)((u32)u < st ? u : 0) * 64);
                qrow = a.q_loc[a.row0 + lane_u32(d_sid, j) * 64 + lane];
            };
            auto add4 = [&](const PK &raw, double &acc) {   // left to right: the reference's order
                const uint4 k = codes_of(raw);
                const double x0 = gather<HUB>(k.x, a.x, hubv, a.hub);
                const double x1 = gather<HUB>(k.y, a.x, hubv, a.hub);
                const double x2 = gather<HUB>(k.z, a.x, hubv, a.hub);
                const double x3 = gather<HUB>(k.w, a.x, hubv, a.hub);
                acc += x0; acc += x1; acc += x2; acc += x3;
            };
            auto consume = [&](u32 j, const PK (&f)[PF], double qrow) {
                const PK *p = reinterpret_cast<const PK *>(sell_cols + lane_u64(d_off, j)) + lane;
                const u32 st = lane_u32(d_st, j);
                double acc = 0.0;
#pragma unroll
                for (int u = 0; u < PF; ++u)
                    if ((u32)u < st) add4(f[u], acc);
                u32 i = PF;
                for (; i + GP <= st; i += GP) {
                    PK c[GP];
#pragma unroll
                    for (int u = 0; u < GP; ++u) c[u] = load_raw<NT>(p + (size_t)(i + u) * 64);
#pragma unroll
                    for (int u = 0; u < GP; ++u) add4(c[u], acc);
                }
                if (i < st) {   // up to GP - 1 more packets: wave-uniform count, all in flight together
                    PK c[GP - 1];
#pragma unroll
                    for (int u = 0; u < GP - 1; ++u) c[u] = load_raw<NT>(p + (size_t)(i + u < st ? i + u : i) * 64);
#pragma unroll
                    for (int u = 0; u < GP - 1; ++u)
                        if (i + u < st) add4(c[u], acc);
                }
                a.v[a.row0 + lane_u32(d_sid, j) * 64 + lane] = acc;
                dot += acc * qrow;
            };
            if (HUB == 2 && a.deep) {
                // experiment (debug knob spmv_deep): the packets of the next THREE slices in flight while one is summed --
                // four register sets used in rotation, every load from a clamped, valid address
                PK f0[PF], f1[PF], f2[PF], f3[PF];
                double q0, q1, q2, q3;
                const u32 lastj = cnt - 1;
                auto at = [&](u32 j) { return j < cnt ? j : lastj; };
                issue(0, f0, q0);
                issue(at(1), f1, q1);
                issue(at(2), f2, q2);
                for (u32 j = 0; j < cnt; j += 4) {
                    issue(at(j + 3), f3, q3);
                    consume(j, f0, q0);
                    if (j + 1 >= cnt) break;
                    issue(at(j + 4), f0, q0);
                    consume(j + 1, f1, q1);
                    if (j + 2 >= cnt) break;
                    issue(at(j + 5), f1, q1);
                    consume(j + 2, f2, q2);
                    if (j + 3 >= cnt) break;
                    issue(at(j + 6), f2, q2);
                    consume(j + 3, f3, q3);
                }
            } else {
                PK fa[PF], fb[PF];
                double qa, qb;
                issue(0, fa, qa);
                for (u32 j = 0; j < cnt; j += 2) {
                    issue(j + 1 < cnt ? j + 1 : j, fb, qb);
                    consume(j, fa, qa);
                    if (j + 1 >= cnt) break;
                    issue(j + 2 < cnt ? j + 2 : j + 1, fa, qa);
                    consume(j + 1, fb, qb);
                }
            }
        }
    }

    // ---- blocked mode: the narrow staged-only slices.  Two thirds of the slices of a large R-MAT graph hold 0 or 4 codes per
    //      row (low-degree rows have one or two staged neighbours, a fifth of the rows none): through the general loop
    //      each cost its ~150 instructions of descriptor, pipeline and tail handling plus four clamped loads -- three
    //      quarters of that loop's instruction stream for a seventh of its entries.  Here: four slices per step, their
    //      packets and q values all in flight together, no pipeline state across steps.
    if (HUB == 2) {
        auto narrow = [&](u32 first, u32 count, auto np_tag) {
            constexpr int NP = decltype(np_tag)::value;
            const u32 mine = count > w0 ? (count - w0 + waves - 1) / waves : 0;
            for (u32 base = 0; base < mine; base += 64) {
                const u32 cnt = mine - base < 64 ? mine - base : 64;
                const u32 d_sid = a.slice_perm[first + w0 + (base + (lane < cnt ? lane : 0)) * waves];
                const u64 d_off = NP ? a.slice_off[d_sid] : 0;
                for (u32 j = 0; j < cnt; j += 4) {
                    PK f[4][NP ? NP : 1];
                    double qr[4];
                    u32 row[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const u32 jj = j + u < cnt ? j + u : cnt - 1;   // clamped: unconditional loads
                        row[u] = a.row0 + lane_u32(d_sid, jj) * 64 + lane;
                        if (NP) {
                            const PK *pp = reinterpret_cast<const PK *>(sell_cols + lane_u64(d_off, jj)) + lane;
#pragma unroll
                            for (int e = 0; e < NP; ++e) f[u][e] = load_raw<NT>(pp + (size_t)e * 64);
                            qr[u] = a.q_loc[row[u]];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (j + u < cnt) {   // wave-uniform
                            double acc = 0.0;
#pragma unroll
                            for (int e = 0; e < NP; ++e) {
                                const uint4 k = codes_of(f[u][e]);
                                const double x0 = hubv[k.x], x1 = hubv[k.y], x2 = hubv[k.z], x3 = hubv[k.w];
                                acc += x0; acc += x1; acc += x2; acc += x3;
                            }
                            a.v[row[u]] = acc;
                            if (NP) dot += acc * qr[u];
                        }
                    }
                }
            }
        };
        narrow(a.n_slices, a.ns_w8, std::integral_constant<int, 2>{});
        narrow(a.n_slices + a.ns_w8, a.ns_w4, std::integral_constant<int, 1>{});
        narrow(a.n_slices + a.ns_w8 + a.ns_w4, a.ns_w0, std::integral_constant<int, 0>{});
    }

    dot = wave_sum(dot);
    if (lane == 0) wsum[wv] = dot;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (u32 i = 0; i < LZX_SPMV_BLOCK / 64; ++i) s += wsum[i];
        a.partials[block] = s;
    }
}

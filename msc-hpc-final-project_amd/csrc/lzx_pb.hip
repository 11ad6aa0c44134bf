// lzx_pb.hip -- propagation-blocked SpMV for the entries whose column is NOT staged in LDS by k_spmv.
//
// Why: past the 4 MiB per-XCD L2 a random 8-byte gather of x costs a whole 128-byte fabric transaction
// (tools/gather_bench.hip: 55 Ggather/s = 7 TB/s of traffic), and even L2-resident gathers top out near
// 200 Ggather/s, so the plain CSR gather moves 8x the algorithmic bytes (profiles/r1_c3_baseline_pmc.json: 15 GB per
// SpMV on the 10 M-vertex graph).  Here the same work is two streaming passes around LDS.
//
// Bands and runs.  Column bands = 16 Ki positions of the exchange layout (one 128 KiB LDS tile of x).  Row bands =
// consecutive local rows, 1024 of them (fewer at the top of the degree order, see pb_prepare_impl).  The entries of one
// (row band, column band) pair are a RUN; the gather order of everything is (row band, column band, row, column).
//
//   scatter (k_pb_scatter): work ordered by COLUMN band; a workgroup stages the band's x values in LDS once and walks
//     REDUCED runs (>= LZX_PBR_MIN_RUN entries: the runs of the high-degree rows and, in every row band, those of the
//       popular columns -- a row has many entries per column band there, so PARTIAL ROW SUMS cross the passes, not
//       x values): cut into steps of 512 entries; lane l of a wavefront takes 8 consecutive entries (one 16-byte
//       load of 15-bit columns-in-band, bit 15 = last entry of a PIECE = of its row within the step), adds them up
//       from LDS, rows that span lanes being summed through wave-private LDS carry slots, and emits one value per
//       piece.  Pieces that close at entry e of their lane form plane e of the step and are written lane-compacted
//       (ballot + mbcnt): every store instruction writes one contiguous stretch, no prefix scan, and the matching
//       row slots are static.  Several pieces of one row are simply added by the gather pass.
//     PLAIN runs (shorter): padded to 8 entries; a lane takes a QUAD of four entries (one 8-byte load of their
//       columns-in-band, one 4-byte load of the quad's value slot), looks the four values up and writes them with
//       two 16-byte stores (padding reads a zero kept behind the staged band).
//     Value slots are in gather order, so a run is one contiguous, 64-byte aligned stretch of writes.
//   gather (k_pb_gather): a row band's values are cut into items of up to ~256 Ki values; one WORKGROUP per item, its
//     eight wavefronts taking the item's 128-value blocks round-robin (the workgroup streams 8 consecutive KiB of
//     values + 2-byte LDS slots at a time) and adding each value into a wave-private LDS y tile with ds_add_f64.  The
//     slot of a value is row * rep + replica, the replica chosen when the graph is reshaped so that the 64 lanes of one
//     instruction (almost) never share a slot: no shuffles, no serialised conflicts; only its own wavefront touches a
//     tile, so additions happen in program order.  The eight tiles are then folded in wavefront order (rep replicas
//     per row) and added to v (the band's only item) or left as per-row totals that k_pb_finish adds in item order; the
//     workgroup forms its share of alpha = v . q.
// All tables are static (built once per graph by lzx_pb_prepare: radix sorts and a few scans, on the device).
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdlib>
#include <queue>
#include <utility>

#include "lzx_internal.h"
#include "lzx_spmv_body.h"

namespace {

// 64-bit sort key: row band (24 bits) | column band (16) | row in band (10) | column in band (14)
__device__ __forceinline__ u64 pack_key(u32 rband, u32 cband, u32 lrow, u32 lcol)
{
    return ((u64)rband << 40) | ((u64)cband << 24) | ((u64)lrow << 14) | (u64)lcol;
}

// one wavefront (64-thread block) per local row, rows strided over the grid (a launch holds at most 2^32 work-items):
// keep the entries whose code is not a hub slot
__global__ void __launch_bounds__(64)
k_pb_emit(const u64 *row_ptr, const u32 *col_idx, const u32 *code_of_old, const u32 *old_of_local,
          const u32 *deg_local, const u32 *nh_off, u32 n_rows, u32 hub, const u32 *band_row0, u32 nr, u32 cb, u64 *keys)
{
    const u32 lane = threadIdx.x;
    for (u32 l = blockIdx.x; l < n_rows; l += gridDim.x) {
        const u32 d = deg_local[l];
        // row band of local row l: last band whose first row is <= l
        u32 lo = 0, hi = nr;
        while (hi - lo > 1) {
            const u32 mid = (lo + hi) >> 1;
            if (band_row0[mid] <= l) lo = mid; else hi = mid;
        }
        const u32 R = lo, lrow = l - band_row0[lo];
        const u64 base = row_ptr[old_of_local[l]];
        u32 out = nh_off[l];
        for (u32 k0 = 0; k0 < d; k0 += 64) {
            const u32 k = k0 + lane;
            u32 cde = 0;
            bool keep = false;
            if (k < d) {
                cde = code_of_old[col_idx[base + k]];
                keep = cde >= hub;
            }
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const u32 p = cde - hub;
                const u32 pre = __popcll(m & ((1ull << lane) - 1ull));
                keys[out + pre] = pack_key(R, p / cb, lrow, p % cb);
            }
            out += __popcll(m);
        }
    }
}

// head[i] = 1 where a (row band, column band) run starts
__global__ void k_pb_heads(const u64 *keys, u64 count, u32 *head)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    head[i] = (i == 0 || (keys[i] >> 24) != (keys[i - 1] >> 24)) ? 1u : 0u;
}

// runstart[r] = first entry of run r (runid = inclusive scan of head, minus 1)
__global__ void k_pb_runstarts(const u32 *head, const u32 *runid_incl, u64 count, u32 *runstart)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    if (head[i]) runstart[runid_incl[i] - 1] = (u32)i;
}

// Per run: its format (1 = reduced: its entries cross the passes as partial row sums) and, for reduced runs, the
// length padded to whole steps.
__global__ void k_pb_run_format(const u32 *runstart, u32 nruns, u64 count, u32 min_run, uint8_t *fmt, u32 *epad)
{
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nruns) return;
    const u32 len = ((r + 1 < nruns) ? runstart[r + 1] : (u32)count) - runstart[r];
    const bool red = len >= min_run;
    fmt[r] = red ? 1 : 0;
    epad[r] = red ? ((len + LZX_PBR_STEP - 1) & ~(LZX_PBR_STEP - 1)) : 0u;
}

// Values a run hands to the gather pass, padded to `align` (so every run starts on a 64-byte boundary): the pieces of
// a reduced run (step_excl = exclusive scan of the pieces per step), the entries of a plain one.
__global__ void k_pb_run_values(const u32 *runstart, u32 nruns, u64 count, const uint8_t *fmt, const u32 *estart,
                                const u32 *step_excl, u32 align, u32 *vcount)
{
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nruns) return;
    u32 n;
    if (fmt[r]) n = step_excl[estart[r + 1] / LZX_PBR_STEP] - step_excl[estart[r] / LZX_PBR_STEP];
    else n = ((r + 1 < nruns) ? runstart[r + 1] : (u32)count) - runstart[r];
    vcount[r] = (n + align - 1u) & ~(align - 1u);
}

// plain runs: place every entry at its value position: row / column within band, and per quad its column band
__global__ void k_pb_place(const u64 *keys, const u32 *runid_incl, const u32 *runstart, const u32 *vpos, const uint8_t *fmt,
                           u64 count, uint16_t *prow, uint16_t *plcol, uint16_t *quad_cband)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const u32 r = runid_incl[i] - 1;
    if (fmt[r]) return;
    const u32 pos = vpos[r] + ((u32)i - runstart[r]);
    const u64 k = keys[i];
    prow[pos] = (uint16_t)((k >> 14) & 0x3ffu);
    plcol[pos] = (uint16_t)(k & 0x3fffu);
    if ((pos & 3u) == 0) quad_cband[pos >> 2] = (uint16_t)((k >> 24) & 0xffffu);
}

__global__ void k_pb_fill16(uint16_t *a, u64 count, uint16_t v)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) a[i] = v;
}

__global__ void k_pb_iota_widen(const uint16_t *in, u64 count, u32 *key, u32 *idx)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        key[i] = in[i];
        idx[i] = (u32)i;
    }
}

// scatter order: quad j of the order is padded quad qsorted[j]
__global__ void k_pb_quads(const u32 *qsorted, const uint16_t *plcol, u64 nquads, uint2 *q_lcol, u32 *q_dst)
{
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nquads) return;
    const u32 q = qsorted[j];
    q_lcol[j] = *reinterpret_cast<const uint2 *>(plcol + (size_t)q * 4);
    q_dst[j] = q * 4u;
}

__global__ void k_pb_bounds_u32(const u32 *keys, u64 count, u32 targets, u32 *out)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > targets) return;
    u64 lo = 0, hi = count;
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (keys[mid] < t) lo = mid + 1; else hi = mid;
    }
    out[t] = (u32)lo;
}

// first UNPADDED entry of each row band (keys sorted by row band first)
__global__ void k_pb_bounds_u64(const u64 *keys, u64 count, u32 shift, u32 targets, u32 *out)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > targets) return;
    u64 lo = 0, hi = count;
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if ((keys[mid] >> shift) < t) lo = mid + 1; else hi = mid;
    }
    out[t] = (u32)lo;
}

// first value position of every row band (a band starts a run); bands without entries get their successor's
__global__ void k_pb_band_pos(const u32 *rstart, const u32 *runid_incl, const u32 *vpos, u32 nr, u64 count, u32 len, u32 *band_pos)
{
    const u32 R = blockIdx.x * blockDim.x + threadIdx.x;
    if (R > nr) return;
    const u32 i = rstart[R];
    band_pos[R] = i < count ? vpos[runid_incl[i] - 1] : len;
}

// Which 64 values meet in one ds_add_f64 instruction of k_pb_gather (counted from the start of the row band): in every
// whole block of 128 values lane l owns values 2 l and 2 l + 1 (one 16-byte load of values, one 4-byte load of slots),
// so the block's two instructions add the even and the odd values; the band's tail (< 128 values) goes 64 consecutive
// values at a time.
__device__ __forceinline__ u32 pb_group_position(u32 band_beg, u32 band_end, u32 g, u32 lane)
{
    const u32 whole = (band_end - band_beg) / 128u;
    if (g < whole * 2u) return band_beg + (g >> 1) * 128u + lane * 2u + (g & 1u);
    return band_beg + whole * 128u + (g - whole * 2u) * 64u + lane;
}

// occ[p] = how many earlier lanes of the same instruction carry the same row; band_rep[R] = max over the band of occ + 1.
__global__ void __launch_bounds__(64)
k_pb_occurrence(const uint16_t *prow, const u32 *rstart_pad, const u32 *band_step0, u32 nr, uint8_t *occ, u32 *band_rep)
{
    const u32 step = blockIdx.x, lane = threadIdx.x;
    u32 lo = 0, hi = nr;   // band of this step: last band whose first step is <= step
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (band_step0[mid] <= step) lo = mid; else hi = mid;
    }
    const u32 R = lo;
    const u32 i = pb_group_position(rstart_pad[R], rstart_pad[R + 1], step - band_step0[R], lane);
    const u32 r = i < rstart_pad[R + 1] ? (u32)prow[i] : 0xffffu;
    const bool live = r != 0xffffu;   // padding carries 0xffff
    u32 k = 0;
    unsigned long long todo = __ballot(live);
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const u32 r0 = __shfl(r, leader, 64);
        const unsigned long long same = __ballot(live && r == r0);
        if (live && r == r0) k = __popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    if (live) occ[i] = (uint8_t)k;
    u32 m = live ? k + 1 : 0;
    for (int o = 32; o > 0; o >>= 1) m = max(m, (u32)__shfl_xor((int)m, o, 64));
    if (lane == 0 && m > 1) atomicMax(&band_rep[R], m);
}

// slot of padded position p in its wave-private y tile: row * rep + (occurrence mod rep); padding -> the tile's
// spare slot (index LZX_PB_RB)
__global__ void k_pb_slots(const uint16_t *prow, const uint8_t *occ, const u32 *rstart_pad, const u32 *band_rep, u32 nr,
                           u64 count, uint16_t *lslot)
{
    const u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= count) return;
    const u32 r = prow[p];
    if (r == 0xffffu) {
        lslot[p] = (uint16_t)LZX_PB_RB;
        return;
    }
    u32 lo = 0, hi = nr;   // band of p
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (rstart_pad[mid] <= p) lo = mid; else hi = mid;
    }
    const u32 rep = band_rep[lo];
    lslot[p] = (uint16_t)(r * rep + (occ[p] % rep));
}


// ---- reduced bands: build ------------------------------------------------------------------------------------
// reduced runs: entry i of the (row band, column band, row, column)-sorted keys -> its place in the step-padded code
// table: code = column in band | 0x8000 on the last entry of a piece (same row, same step), row in band, and per
// step its column band and run
__global__ void k_pbr_place(const u64 *keys, const u32 *runid_incl, const u32 *runstart, const u32 *estart, const uint8_t *fmt,
                            u64 count, uint16_t *rcode, uint16_t *rrow, uint16_t *step_cband, u32 *step_run)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const u32 r = runid_incl[i] - 1;
    if (!fmt[r]) return;
    const u32 off = (u32)i - runstart[r];
    const u32 pos = estart[r] + off;
    const u64 k = keys[i];
    const u32 lrow = (u32)((k >> 14) & 0x3ffu);
    const bool last = (off & (LZX_PBR_STEP - 1)) == LZX_PBR_STEP - 1 || i + 1 == count || runid_incl[i + 1] - 1 != r ||
                      (u32)((keys[i + 1] >> 14) & 0x3ffu) != lrow;
    rcode[pos] = (uint16_t)((k & 0x3fffu) | (last ? 0x8000u : 0u));
    rrow[pos] = (uint16_t)lrow;
    if ((off & (LZX_PBR_STEP - 1)) == 0) {
        step_cband[pos / LZX_PBR_STEP] = (uint16_t)((k >> 24) & 0xffffu);
        step_run[pos / LZX_PBR_STEP] = r;
    }
}

// first value slot of every step: its run's first slot + the pieces of the run's earlier steps
__global__ void k_pbr_step_base(const u32 *step_run, const u32 *estart, const u32 *step_excl, const u32 *vpos, u32 nsteps, u32 *step_base)
{
    const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsteps) return;
    const u32 r = step_run[s];
    step_base[s] = vpos[r] + step_excl[s] - step_excl[estart[r] / LZX_PBR_STEP];
}

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

// pieces per step
__global__ void __launch_bounds__(64) k_pbr_count(const uint4 *rcode, u32 *cnt)
{
    const uint4 c = rcode[(size_t)blockIdx.x * 64 + threadIdx.x];
    u32 n = __popc(c.x & 0x80008000u) + __popc(c.y & 0x80008000u) + __popc(c.z & 0x80008000u) + __popc(c.w & 0x80008000u);
    for (int o = 32; o > 0; o >>= 1) n += (u32)__shfl_xor((int)n, o, 64);
    if (threadIdx.x == 0) cnt[blockIdx.x] = n;
}

// row (in band) of every piece, at the position k_pb_scatter writes the piece to
__global__ void __launch_bounds__(64) k_pbr_rows(const uint4 *rcode, const uint4 *rrow, const u32 *step_base, uint16_t *prow)
{
    const uint4 c = rcode[(size_t)blockIdx.x * 64 + threadIdx.x];
    const uint4 r = rrow[(size_t)blockIdx.x * 64 + threadIdx.x];
    u32 base = step_base[blockIdx.x];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bool f = pbr_flag(c, e) != 0;
        const unsigned long long m = __ballot(f);
        if (f) prow[base + lanes_below(m)] = (uint16_t)pbr_half(r, e);
        base += (u32)__popcll(m);
    }
}

// scatter order of the steps: step j of the order is step ssorted[j] of the gather (row band major) order
__global__ void __launch_bounds__(64) k_pbr_steps(const u32 *ssorted, const uint4 *rcode, const u32 *step_base, uint4 *scode, u32 *sbase)
{
    const u32 s = ssorted[blockIdx.x];
    scode[(size_t)blockIdx.x * 64 + threadIdx.x] = rcode[(size_t)s * 64 + threadIdx.x];
    if (threadIdx.x == 0) sbase[blockIdx.x] = step_base[s];
}

// ---- the per-iteration kernels --------------------------------------------------------------------------------
// Scatter pass.  unit = {column band, first step, last step, first quad, last quad}: the workgroup stages the band's
// CB x values (plus a zero for padding) in LDS once and then walks its share of both tables.  CB = 16 Ki (128 KiB, one
// workgroup per CU) or, for graphs whose x sits in the L2s, 8 Ki (two workgroups per CU).
//   reduced part: wavefront w takes steps w, w+16, ... of the unit, four steps' loads in flight; lane = 8 consecutive
//       entries, pieces written plane by plane, lane-compacted;
//   plain part: each wavefront walks its own contiguous share of the quads 64 at a time (lane = consecutive quad):
//       contiguous loads, and 32-byte-per-lane stores that are contiguous inside a run; 4 quads per lane in flight.
// DBG: the LZX_ABLATE experiment switches behind DESIGN.md's ablation numbers are compiled in (slower even when 0).
template <u32 CB, bool DBG>
__device__ __forceinline__ void
pb_scatter_body(const u32 *unit, const uint4 *scode, const u32 *sbase, const uint2 *q_lcol, const u32 *q_dst,
                const double *__restrict__ x, u64 xlen, double *val, int ablate_arg, const u32 ublock)
{
    const int ablate = DBG ? ablate_arg : 0;
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
        {   // up to three more steps: requested together (wave-uniform predicates), not one round trip each
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
        {   // up to three more quads per lane (the last one partly filled): requested together
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

template <u32 CB, bool DBG>
__global__ void __launch_bounds__(1024)
k_pb_scatter(const u32 *unit, const uint4 *scode, const u32 *sbase, const uint2 *q_lcol, const u32 *q_dst,
             const double *__restrict__ x, u64 xlen, double *val, int ablate_arg)
{
    pb_scatter_body<CB, DBG>(unit, scode, sbase, q_lcol, q_dst, x, xlen, val, ablate_arg, blockIdx.x);
}

// Scatter pass and staged-columns kernel in ONE launch (single GPU, 16 Ki bands): the persistent workgroups of the
// staged-columns kernel (lzx_spmv_body.h) and the scatter units share a grid.  Both only read x and write different things
// (v / values), one workgroup of either kind fits a CU, and workgroups are dispatched in index order.  spmv_first (the
// default): the 256 staged-columns workgroups start on every CU at once and the scatter units follow as CUs come free, so
// the launch ends with the small tapered units instead of a second ramp and drain; the other order (debug knob
// fuse_staged = 1: staged columns behind the units, filling the scatter pass's tail while the write-back of its values
// drains in their shadow) measures the same.  C3: SpMV 0.616 -> 0.601 ms against separate launches.
template <u32 CB>
__global__ void __launch_bounds__(1024)
k_pb_scatter_spmv(const u32 *unit, u32 n_units, const uint4 *scode, const u32 *sbase, const uint2 *q_lcol, const u32 *q_dst,
                  const double *__restrict__ x, u64 xlen, double *val, const SpmvArgs a, const u32 spmv_first)
{
    const u32 spmv_blocks = gridDim.x - n_units;
    if (spmv_first) {
        if (blockIdx.x < spmv_blocks) spmv_body<2, false>(a, blockIdx.x, spmv_blocks);
        else pb_scatter_body<CB, false>(unit, scode, sbase, q_lcol, q_dst, x, xlen, val, 0, blockIdx.x - spmv_blocks);
        return;
    }
    if (blockIdx.x < n_units) pb_scatter_body<CB, false>(unit, scode, sbase, q_lcol, q_dst, x, xlen, val, 0, blockIdx.x);
    else spmv_body<2, false>(a, blockIdx.x - n_units, spmv_blocks);
}

__device__ __forceinline__ double wave_sum_pb(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// item table entry: {row band, begin, end, slot} in gather positions; slot == 0xffffffff: the item is its band's
// only one and adds straight into v; otherwise it is one of several items of its band and leaves its per-row totals
// in part[slot + row in band] for k_pb_finish.
// One WORKGROUP per item: its eight wavefronts take the item's 128-value blocks round-robin (so the workgroup reads 8
// consecutive KiB at a time -- a few hundred sequential streams chip-wide instead of four thousand), each adding into
// its own y tile; the tiles are folded in wavefront order.
// STAMP (debug library, option pb_stamps): wavefront 0 of every workgroup keeps 100 MHz time stamps per section --
// stamps[8 * workgroup ..]: start, end, ticks zeroing tiles + reading item records, ticks streaming, ticks in barriers
// before the fold, ticks folding, items, values.
template <bool STAMP>
__global__ void __launch_bounds__(LZX_PB_GATHER_BLOCK)
k_pb_gather(const uint4 *items, u32 n_items, const u32 *band_row0, const u32 *band_rep, const u32 *band_beg,
            const uint16_t *lslot, const double *val, double *v, const double *__restrict__ q_loc, double *part,
            double *partials, unsigned long long *stamps)
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
    double *wsum = lds + (size_t)WAVES * TILE;           // [WAVES]
    double dot = 0.0;
    for (u32 it = blockIdx.x; it < n_items; it += gridDim.x) {
        const uint4 item = items[it];
        if (item.w == LZX_PB_ITEM_NONE) continue;   // filler of the balanced schedule
        if (item.w == LZX_PB_ITEM_GROUP) {
            // a group of up to eight SMALL consecutive bands, one per wavefront: each wavefront streams its own band into
            // its own tile and folds it itself -- no workgroup barrier, no cross-wavefront fold, eight bands' round trips
            // in flight per workgroup (the low-degree end of the row order is thousands of bands of a few thousand values)
            __syncthreads();                 // the previous item's fold (it reads every wavefront's tile) is done
            if (wv < item.y) {
                const u32 R = item.x + wv;
                const u32 row0 = band_row0[R], rows = band_row0[R + 1] - row0, rep = band_rep[R];
                const u32 beg = band_beg[R], end = band_beg[R + 1];
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
            }
            continue;
        }
        const u32 R = item.x, beg = item.y, end = item.z;
        if (STAMP) { ++n_it; n_vals += end - beg; }
        const u32 row0 = band_row0[R], rows = band_row0[R + 1] - row0;
        if (rows == 1) {
            // one heavy row: plain strided sum (padding holds zeros), fixed reduction order
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
            __syncthreads();                 // wsum free
            if (lane == 0) wsum[wv] = acc;
            __syncthreads();
            if (tid == 0) {
                double t = 0.0;
                for (u32 w = 0; w < WAVES; ++w) t += wsum[w];
                if (item.w == 0xffffffffu) {
                    v[row0] += t;
                    dot += t * q_loc[row0];
                } else {
                    part[item.w] = t;
                }
            }
            continue;
        }
        const u32 rep = band_rep[R];
        const u32 slots = rows * rep;
        __syncthreads();                     // the previous item's fold is done with the tiles
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
        for (; kb < blocks; kb += WAVES) {
            const u32 p = beg + kb * 128u + lane * 2;
            const double2 a = *reinterpret_cast<const double2 *>(val + p);
            const u32 s = *reinterpret_cast<const u32 *>(lslot + p);
            atomicAdd(&ytile[s & 0xffffu], a.x);
            atomicAdd(&ytile[s >> 16], a.y);
        }
        // the band's tail (< 128 values): 64 consecutive values per instruction, wavefront 0
        if (wv == 0)
            for (u32 i = beg + blocks * 128u + lane; i < end; i += 64) atomicAdd(&ytile[lslot[i]], val[i]);
        GSTAMP(t_stream);
        __syncthreads();
        GSTAMP(t_bar);
        // fold: wavefront tiles in order, replicas in order; every thread a few rows, loads before stores
        if (item.w == 0xffffffffu) {
            for (u32 j0 = tid; j0 < rows; j0 += 2 * LZX_PB_GATHER_BLOCK) {
                double vv[2], qq[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const u32 j = j0 + u * LZX_PB_GATHER_BLOCK;
                    vv[u] = j < rows ? v[row0 + j] : 0.0;
                    qq[u] = j < rows ? q_loc[row0 + j] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const u32 j = j0 + u * LZX_PB_GATHER_BLOCK;
                    if (j < rows) {
                        double y = 0.0;
                        for (u32 w = 0; w < WAVES; ++w)
                            for (u32 t = 0; t < rep; ++t) y += lds[(size_t)w * TILE + j * rep + t];
                        v[row0 + j] = vv[u] + y;
                        dot += y * qq[u];
                    }
                }
            }
        } else {
            for (u32 j = tid; j < rows; j += LZX_PB_GATHER_BLOCK) {
                double y = 0.0;
                for (u32 w = 0; w < WAVES; ++w)
                    for (u32 t = 0; t < rep; ++t) y += lds[(size_t)w * TILE + j * rep + t];
                part[item.w + j] = y;
            }
        }
        GSTAMP(t_fold);
    }
    if (STAMP && tid == 0) {
        unsigned long long *o = stamps + 8 * (size_t)blockIdx.x;
        o[0] = t_start; o[1] = wall_clock64(); o[2] = t_zero; o[3] = t_stream; o[4] = t_bar; o[5] = t_fold; o[6] = n_it; o[7] = n_vals;
    }
#undef GSTAMP
    dot = wave_sum_pb(dot);
    __syncthreads();
    if (lane == 0) wsum[wv] = dot;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (u32 i = 0; i < WAVES; ++i) s += wsum[i];
        partials[blockIdx.x] = s;
    }
}

// ---- gather pass, product form (round 3) ------------------------------------------------------------------------
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
enum : u32 { LZX_G3_NORMAL = 0, LZX_G3_ONE_ROW = 1, LZX_G3_GROUP = 2, LZX_G3_IDLE = 3 };

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

// v[row] += totals that were left for it, in their fixed order; alpha partials for those rows.  Threads [0, n_multi):
// rows of bands cut into several gather items (their per-item totals); threads behind them: the split rows of the
// staged-column kernel (their item totals, k_long_finish's job in plain mode).
__global__ void __launch_bounds__(LZX_VEC_BLOCK)
k_pb_finish(const uint4 *multi /*[n]: row, first slot, items, slot stride*/, u32 n_multi, const double *part,
            const u32 *item_first, const double *long_partial, const uint8_t *long_is_multi, u32 n_long, double *v,
            const double *q_loc, double *partials, const double *item_dot, u32 n_item_dot)
{
    __shared__ double sh[4];
    const u32 t = blockIdx.x * LZX_VEC_BLOCK + threadIdx.x;
    double dot = 0.0;
    // the gather pass's per-item alpha partials (k_pb_gather3), 256 per block in item order: nothing depends on which
    // workgroup drew which item
    const double idot = t < n_item_dot ? item_dot[t] : 0.0;
    if (t < n_multi) {
        const uint4 m = multi[t];
        const u32 row = m.x;
        double s = 0.0;
        for (u32 k = 0; k < m.z; ++k) s += part[m.y + (size_t)k * m.w];
        if (row < n_long)   // a split row of a multi-item band: one thread owns the row's update
            for (u32 it = item_first[row]; it < item_first[row + 1]; ++it) s += long_partial[it];
        v[row] += s;
        dot = s * q_loc[row];
    } else if (t - n_multi < n_long && !long_is_multi[t - n_multi]) {
        const u32 r = t - n_multi;
        double s = 0.0;
        for (u32 it = item_first[r]; it < item_first[r + 1]; ++it) s += long_partial[it];
        v[r] += s;
        dot = s * q_loc[r];
    }
    dot += idot;
    dot = wave_sum_pb(dot);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
}


#ifdef LZX_DEBUG_KNOBS
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

#endif  // LZX_DEBUG_KNOBS

template <typename T>
int pb_alloc(T **p, u64 count)
{
    *p = nullptr;
    LZX_HIP(hipMalloc(reinterpret_cast<void **>(p), (count ? count : 1) * sizeof(T)));
    return LZX_OK;
}
template <typename T>
void pb_free(T *&p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}
}  // namespace

void lzx_pb_release(lzx_ctx *c)
{
    pb_free(c->d_pb_lcol);
    pb_free(c->d_pb_dst);
    pb_free(c->d_pb_lrow);
    pb_free(c->d_pb_val);
    pb_free(c->d_pb_unit);
    pb_free(c->d_pb_row0);
    pb_free(c->d_pb_rep);
    pb_free(c->d_pb_beg);
    pb_free(c->d_pb_items);
    pb_free(c->d_pb_items2);
    pb_free(c->d_pb_wg_begin);
    pb_free(c->d_pb_seg);
    pb_free(c->d_pb_seg_begin);
    pb_free(c->d_pb_stamps);
    pb_free(c->d_pb_gstamps);
    pb_free(c->d_pb_grec);
    pb_free(c->d_pb_item_dot);
    pb_free(c->d_pb_gqueue);
    c->pb_g3 = false;
    c->pb_gq_base = 0;
    pb_free(c->d_pb_queue);
    pb_free(c->d_pb_multi);
    pb_free(c->d_pb_part);
    pb_free(c->d_pb_long_multi);
    pb_free(c->d_pbr_code);
    pb_free(c->d_pbr_base);
    c->pb = false;
    c->pb_entries = c->pb_values = c->pbr_entries = 0;
    c->pb_units = c->pb_units0 = c->pb_nr = c->pb_gather_grid = c->pb_n_items = c->pb_n_multi = c->pb_finish_grid = 0;
    c->pbr_steps = 0;
}

// block partials of alpha the blocked passes leave: the ticketed gather pass hands its per-item partials to k_pb_finish
u32 lzx_pb_partials(const lzx_ctx *c) { return c->pb ? (c->pb_g3 ? 0u : c->pb_gather_grid) + c->pb_finish_grid : 0; }

namespace {
// ---- build helpers: every one leaves its temporaries to the caller's arena, which frees them on any exit ----------
struct Arena {
    std::vector<void *> ptrs;
    template <typename T> int get(T **p, u64 count)
    {
        LZX_TRY(pb_alloc(p, count));
        ptrs.push_back(*p);
        return LZX_OK;
    }
    template <typename T> void drop(T *&p)
    {
        for (auto &q : ptrs)
            if (q == p) q = nullptr;
        pb_free(p);
    }
    ~Arena()
    {
        for (void *q : ptrs)
            if (q) (void)hipFree(q);
    }
};
#define GRID(n) dim3((u32)(((u64)(n) + 255) / 256)), dim3(256), 0, st

int pb_sort_keys(hipStream_t st, u64 *in, u64 *out, u64 count)
{
    size_t tb = 0;
    void *tmp = nullptr;
    LZX_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, in, out, count, 0, 64, st));
    LZX_HIP(hipMalloc(&tmp, tb ? tb : 16));
    hipError_t e = hipcub::DeviceRadixSort::SortKeys(tmp, tb, in, out, count, 0, 64, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    LZX_HIP(e);
    return LZX_OK;
}
int pb_sort_pairs16(hipStream_t st, u32 *kin, u32 *kout, u32 *vin, u32 *vout, u64 count)
{
    size_t tb = 0;
    void *tmp = nullptr;
    LZX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, kin, kout, vin, vout, count, 0, 16, st));
    LZX_HIP(hipMalloc(&tmp, tb ? tb : 16));
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(tmp, tb, kin, kout, vin, vout, count, 0, 16, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    LZX_HIP(e);
    return LZX_OK;
}
int pb_scan(hipStream_t st, bool inclusive, u32 *in, u32 *out, u64 count)
{
    size_t tb = 0;
    void *tmp = nullptr;
    if (inclusive) LZX_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tb, in, out, count, st));
    else LZX_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, in, out, count, st));
    LZX_HIP(hipMalloc(&tmp, tb ? tb : 16));
    hipError_t e = inclusive ? hipcub::DeviceScan::InclusiveSum(tmp, tb, in, out, count, st)
                             : hipcub::DeviceScan::ExclusiveSum(tmp, tb, in, out, count, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    LZX_HIP(e);
    return LZX_OK;
}

// Scatter work units {column band, first step, last step, first quad, last quad}: a band's steps (reduced part) and
// quads (plain part) are cut into as many units as its entries need at `cap` entries per unit, each unit taking the
// same share of both.  units0 = how many of them (they are sorted by band) lie wholly inside chunk 0 of the exchange.
int pb_units(lzx_ctx *c, hipStream_t st, const std::vector<u32> &sstart, const std::vector<u32> &qstart, u32 nb, u32 cap)
{
    std::vector<u32> units;
    u64 all = 0, done = 0;
    for (u32 b = 0; b < nb; ++b)
        all += (u64)((sstart.empty() ? 0 : sstart[b + 1] - sstart[b])) * LZX_PBR_STEP + (u64)((qstart.empty() ? 0 : qstart[b + 1] - qstart[b])) * 4;
    for (u32 b = 0; b < nb; ++b) {
        const u32 s0 = sstart.empty() ? 0 : sstart[b], s1 = sstart.empty() ? 0 : sstart[b + 1];
        const u32 q0 = qstart.empty() ? 0 : qstart[b], q1 = qstart.empty() ? 0 : qstart[b + 1];
        const u64 entries = (u64)(s1 - s0) * LZX_PBR_STEP + (u64)(q1 - q0) * 4;
        if (entries == 0) continue;
        // tapered: units are dispatched in this order, so the first 70 % of the entries go in units of twice the size (x is
        // restaged half as often: every restaged band is bytes through the CU's memory pipeline, the scatter pass's
        // bound), the last 10 % in units of half the size, which even out the tail (C3: 0.338 -> 0.322 ms, neutral on
        // rank shares and on C2; three schedules tried, all alike; option pb_taper = 0 switches it off)
        u64 capb = cap;
        if (c->pb_taper_opt != 0) capb = done * 10 < all * 7 ? 2ull * cap : done * 10 < all * 9 ? cap : std::max<u64>(16384, cap / 2);
        done += entries;
        const u32 parts = (u32)((entries + capb - 1) / capb);
        for (u32 i = 0; i < parts; ++i) {
            units.push_back(b);
            units.push_back(s0 + (u32)((u64)(s1 - s0) * i / parts));
            units.push_back(s0 + (u32)((u64)(s1 - s0) * (i + 1) / parts));
            units.push_back(q0 + (u32)((u64)(q1 - q0) * i / parts));
            units.push_back(q0 + (u32)((u64)(q1 - q0) * (i + 1) / parts));
        }
    }
    c->pb_units = (u32)(units.size() / 5);
    c->pb_units0 = c->pb_units;
    if (c->overlap) {
        const u64 chunk0_end = (u64)c->world * c->xs0;
        u32 u0 = 0;
        while (u0 < c->pb_units && ((u64)units[5 * u0] + 1) * c->pb_cb <= chunk0_end) ++u0;
        c->pb_units0 = u0;
    }
    LZX_TRY(pb_alloc(&c->d_pb_unit, units.size()));
    if (!units.empty())
        LZX_HIP(hipMemcpyAsync(c->d_pb_unit, units.data(), sizeof(u32) * units.size(), hipMemcpyHostToDevice, st));
    LZX_HIP(hipStreamSynchronize(st));
    return LZX_OK;
}

#ifdef LZX_DEBUG_KNOBS
// Static scatter schedule: the scatter order (band by band: a band's steps, then its quads) is cut into `groups`
// stretches of equal cost -- bytes read + written: per step its 1 KiB of codes + 8 B per piece, per quad 12 B + 32 B --
// one per workgroup; a stretch is stored as segments {band, steps, quads}, one per band it touches.  Two schedules:
// the bands of chunk 0 of the exchange (all of them without the two-chunk exchange) and the rest.
int pb_segments(lzx_ctx *c, hipStream_t st, const std::vector<u32> &sstart, const std::vector<u32> &qstart, u32 nb, u32 nsteps)
{
    std::vector<u32> cnt(nsteps, 0u);
    if (nsteps) {
        u32 *d_cnt = nullptr;
        LZX_TRY(pb_alloc(&d_cnt, nsteps));
        hipLaunchKernelGGL(k_pbr_count, dim3(nsteps), dim3(64), 0, st, c->d_pbr_code, d_cnt);
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

#endif

int pb_download(hipStream_t st, const u32 *d, size_t count, std::vector<u32> &h)
{
    h.resize(count);
    LZX_HIP(hipMemcpyAsync(h.data(), d, sizeof(u32) * count, hipMemcpyDeviceToHost, st));
    LZX_HIP(hipStreamSynchronize(st));
    return LZX_OK;
}

int pb_prepare_impl(lzx_ctx *c, const u32 *d_code, const u32 *d_old_of_local, const u32 *d_deg_local,
                    const u32 *d_nh_off, const std::vector<u32> &h_nh, u64 total)
{
    hipStream_t st = c->stream;
    if (total >= LZX_PB_SLOT_LIMIT) LZX_FAIL(LZX_ERR_LIMIT, "propagation blocking: %llu entries do not fit 32-bit slots", (unsigned long long)total);
    // column band = LDS tile of x: 16 Ki values; 8 Ki (two scatter workgroups per CU, more and shorter units) when x sits
    // in the L2s anyway (C2: scatter 0.033 -> 0.024 ms; on C3 the doubled number of (row, band) pairs loses: 0.75 -> 0.87 ms)
    // (that was for a scatter pass with a launch of its own: sharing one with the staged-columns kernel, whose 128 KiB tile
    //  leaves room for one workgroup per CU either way, 16 Ki bands win there too -- C2 SpMV 0.076 -> 0.068 ms; 8 Ki bands
    //  remain for the two-chunk exchange on several ranks, where the passes are launched separately)
    c->pb_cb = (c->pb_cb_opt == 8192 || c->pb_cb_opt == 16384) ? (u32)c->pb_cb_opt
               : (c->xlen * sizeof(double) <= (16u << 20) && (c->overlap || c->fuse_opt == 0) ? 8192u : LZX_PB_CB);
    const u32 nb = (u32)((c->xlen + c->pb_cb - 1) / c->pb_cb);
    if (nb >= (1u << 16)) LZX_FAIL(LZX_ERR_LIMIT, "propagation blocking: %u column bands (limit 65535)", nb);
    // values per gather item / entries per plain band: every wavefront slot of the gather pass (2 workgroups of 8 per
    // CU) should get a few items, and an item should not be shorter than its fold is worth
    u32 target = LZX_PB_TARGET;
    {
        // a workgroup (8 wavefronts) per item, two workgroups per CU, a few items each
        const u64 want = total / ((u64)c->cu_count * 8);
        target = (u32)std::min<u64>(8 * LZX_PB_TARGET, std::max<u64>(8192, (want + 1023) & ~1023ull));
    }
    if (c->pb_target_opt > 0) target = (u32)c->pb_target_opt;
    // entries per scatter unit: every unit restages its column band while its CU does nothing else, and a workgroup
    // only leaves its CU to the next one when its last wavefront is done, so big units are cheaper; small ones even out
    // the tail.  About four units per CU, between 64 Ki and 128 Ki entries (measured, tools/perf_probe.py @unit and
    // tools/rank_probe.py: C3 on one GPU 128 Ki 0.317 vs 64 Ki 0.344 vs 256 Ki 0.326 ms; its 1/8 share 64 Ki best);
    // down to 8 Ki only to give every CU about four units on graphs whose x sits in the L2s anyway
    u32 unit_cap = (u32)std::min<u64>(131072, std::max<u64>(65536, (total / ((u64)c->cu_count * 4) + 511) & ~511ull));
    if (c->xlen * sizeof(double) <= (16u << 20))
        unit_cap = (u32)std::min<u64>(65536, std::max<u64>(8192, (total / ((u64)c->cu_count * 4) + 511) & ~511ull));
    if (c->pb_unit_opt > 0) unit_cap = (u32)c->pb_unit_opt;

    // ---- row bands: consecutive local rows (they are in descending degree order): as many rows as the wave-private
    //      y tile can give enough replica slots -- a row with many entries per column band sends many pieces in a row
    //      to the same slot -- so 16 rows at the very top, LZX_PB_RB rows from a few thousand entries per row down.
    //      (Option pb_reduce = 0, everything plain: closed at ~target entries instead, one gather item per band.)
    const bool reduce = c->pb_reduce_opt != 0;
    const u32 min_run = !reduce ? 0xffffffffu : c->pb_reduce_opt > 1 ? (u32)c->pb_reduce_opt : LZX_PBR_MIN_RUN;
    // bands cover the rows that can have entries: rows without an edge (the tail behind rows_live) belong to no band --
    // the gather pass must not touch them (the lazy loop keeps neither v nor the basis columns there)
    const u32 band_rows = std::min(c->n_loc_real, c->rows_live);
    std::vector<u32> row0;
    row0.push_back(0);
    if (reduce) {
        u32 l = 0;
        while (l < band_rows) {
            // replicas for the band's heaviest row (rows are ranked by degree or by staged-column count: the blocked
            // count falls only roughly along the order)
            u32 rep = 1;
            for (;;) {
                const u32 e = (u32)std::min<u64>((u64)l + LZX_PB_RB / rep, band_rows);
                u32 heaviest = 0;
                for (u32 j = l; j < e; ++j) heaviest = std::max(heaviest, h_nh[j]);
                u32 need = rep;
                while (need < 64 && (u64)need * 2 * nb < heaviest) need <<= 1;
                if (need == rep) break;
                rep = need;
            }
            l = (u32)std::min<u64>((u64)l + LZX_PB_RB / rep, band_rows);
            row0.push_back(l);
        }
    } else {
        u32 rows = 0;
        u64 cnt = 0;
        for (u32 l = 0; l < band_rows; ++l) {
            const u32 nh = h_nh[l];
            if (rows > 0 && (cnt + nh > target || rows == LZX_PB_RB)) {
                row0.push_back(l);
                rows = 0;
                cnt = 0;
            }
            ++rows;
            cnt += nh;
        }
        if (rows > 0) row0.push_back(band_rows);
    }
    if (row0.size() == 1) row0.push_back(band_rows);
    const u32 nr = (u32)row0.size() - 1;
    if (nr >= (1u << 24)) LZX_FAIL(LZX_ERR_LIMIT, "propagation blocking: %u row bands (limit 2^24)", nr);
    LZX_TRY(pb_alloc(&c->d_pb_row0, (u64)nr + 1));
    LZX_HIP(hipMemcpyAsync(c->d_pb_row0, row0.data(), sizeof(u32) * ((size_t)nr + 1), hipMemcpyHostToDevice, st));

    Arena ar;
    // 1. emit + sort by (row band, column band, row, column): the gather order
    u64 *d_keys = nullptr, *d_sorted = nullptr;
    LZX_TRY(ar.get(&d_keys, total)); LZX_TRY(ar.get(&d_sorted, total));
    if (c->n_loc_real)
        hipLaunchKernelGGL(k_pb_emit, dim3(std::min<u32>(c->n_loc_real, 1u << 22)), dim3(64), 0, st, c->d_row_ptr, c->d_col_idx, d_code,
                           d_old_of_local, d_deg_local, d_nh_off, c->n_loc_real, c->hub_real, c->d_pb_row0, nr, c->pb_cb, d_keys);
    LZX_TRY(pb_sort_keys(st, d_keys, d_sorted, total));
    ar.drop(d_keys);

    // 2. runs = maximal stretches of one (row band, column band).  A run of at least min_run entries is REDUCED (cut
    //    into steps, its rows' partial sums cross the passes), a shorter one PLAIN (its x values do).  On R-MAT graphs
    //    the long runs are those of the high-degree rows and, for every row band, those of the most popular columns.
    u32 *d_head = nullptr, *d_runid = nullptr, *d_runstart = nullptr, *d_epad = nullptr, *d_estart = nullptr;
    uint8_t *d_fmt = nullptr;
    u32 nruns = 0, epad_total = 0;
    LZX_TRY(ar.get(&d_head, total)); LZX_TRY(ar.get(&d_runid, total));
    hipLaunchKernelGGL(k_pb_heads, GRID(total), d_sorted, total, d_head);
    LZX_TRY(pb_scan(st, true, d_head, d_runid, total));
    LZX_HIP(hipMemcpy(&nruns, d_runid + (total - 1), sizeof(u32), hipMemcpyDeviceToHost));
    LZX_TRY(ar.get(&d_runstart, (u64)nruns + 1)); LZX_TRY(ar.get(&d_epad, (u64)nruns + 1)); LZX_TRY(ar.get(&d_estart, (u64)nruns + 1));
    LZX_TRY(ar.get(&d_fmt, (u64)nruns + 1));
    hipLaunchKernelGGL(k_pb_runstarts, GRID(total), d_head, d_runid, total, d_runstart);
    ar.drop(d_head);
    LZX_HIP(hipMemsetAsync(d_epad + nruns, 0, sizeof(u32), st));
    hipLaunchKernelGGL(k_pb_run_format, GRID(nruns), d_runstart, nruns, total, min_run, d_fmt, d_epad);
    {   // 64-bit check before the 32-bit scan
        std::vector<u32> h;
        LZX_TRY(pb_download(st, d_epad, nruns, h));
        u64 sum = 0;
        for (u32 v : h) sum += v;
        if (sum >= LZX_PB_SLOT_LIMIT) LZX_FAIL(LZX_ERR_LIMIT, "propagation blocking: %llu padded entries do not fit 32-bit slots", (unsigned long long)sum);
        epad_total = (u32)sum;
    }
    LZX_TRY(pb_scan(st, false, d_epad, d_estart, (u64)nruns + 1));
    ar.drop(d_epad);
    const u32 nsteps = epad_total / LZX_PBR_STEP;

    // 3. reduced runs: code table in gather (row band major) order, pieces per step
    uint16_t *d_rcode = nullptr, *d_rrow = nullptr, *d_step_cband = nullptr;
    u32 *d_step_run = nullptr, *d_step_cnt = nullptr, *d_step_excl = nullptr;
    LZX_TRY(ar.get(&d_rcode, (u64)epad_total + 8)); LZX_TRY(ar.get(&d_rrow, (u64)epad_total + 8));
    LZX_TRY(ar.get(&d_step_cband, (u64)nsteps + 1)); LZX_TRY(ar.get(&d_step_run, (u64)nsteps + 1));
    LZX_TRY(ar.get(&d_step_cnt, (u64)nsteps + 1)); LZX_TRY(ar.get(&d_step_excl, (u64)nsteps + 1));
    if (nsteps) {
        hipLaunchKernelGGL(k_pb_fill16, GRID(epad_total), d_rcode, epad_total, (uint16_t)c->pb_cb);   // padding: the zero slot, no flag
        hipLaunchKernelGGL(k_pb_fill16, GRID(epad_total), d_rrow, epad_total, (uint16_t)0xffffu);
        hipLaunchKernelGGL(k_pbr_place, GRID(total), d_sorted, d_runid, d_runstart, d_estart, d_fmt, total, d_rcode, d_rrow,
                           d_step_cband, d_step_run);
        hipLaunchKernelGGL(k_pbr_count, dim3(nsteps), dim3(64), 0, st, reinterpret_cast<const uint4 *>(d_rcode), d_step_cnt);
    }
    LZX_HIP(hipMemsetAsync(d_step_cnt + nsteps, 0, sizeof(u32), st));
    LZX_TRY(pb_scan(st, false, d_step_cnt, d_step_excl, (u64)nsteps + 1));
    ar.drop(d_step_cnt);

    // 4. value positions: every run gets its values (pieces or entries) padded to whole 64-byte lines, in gather order
    const u32 run_align = (c->pb_align_opt == 8 || c->pb_align_opt == 16) ? (u32)c->pb_align_opt : LZX_PB_ALIGN;
    u32 *d_vcount = nullptr, *d_vpos = nullptr;
    LZX_TRY(ar.get(&d_vcount, (u64)nruns + 1)); LZX_TRY(ar.get(&d_vpos, (u64)nruns + 1));
    LZX_HIP(hipMemsetAsync(d_vcount + nruns, 0, sizeof(u32), st));
    hipLaunchKernelGGL(k_pb_run_values, GRID(nruns), d_runstart, nruns, total, d_fmt, d_estart, d_step_excl, run_align, d_vcount);
    u64 len = 0, red_entries = 0;
    {
        std::vector<u32> h;
        LZX_TRY(pb_download(st, d_vcount, nruns, h));
        for (u32 v : h) len += v;
        if (len >= LZX_PB_SLOT_LIMIT) LZX_FAIL(LZX_ERR_LIMIT, "propagation blocking: %llu values do not fit 32-bit slots", (unsigned long long)len);
    }
    LZX_TRY(pb_scan(st, false, d_vcount, d_vpos, (u64)nruns + 1));
    ar.drop(d_vcount);
    const u64 nquads = len / 4;
    uint16_t *d_prow = nullptr, *d_plcol = nullptr, *d_qcband = nullptr;
    u32 *d_step_base = nullptr;
    LZX_TRY(ar.get(&d_prow, len + 8)); LZX_TRY(ar.get(&d_plcol, len + 8)); LZX_TRY(ar.get(&d_qcband, nquads + 2));
    LZX_TRY(ar.get(&d_step_base, (u64)nsteps + 1));
    hipLaunchKernelGGL(k_pb_fill16, GRID(len + 8), d_prow, len + 8, (uint16_t)0xffffu);        // padding: no row
    hipLaunchKernelGGL(k_pb_fill16, GRID(len + 8), d_plcol, len + 8, (uint16_t)c->pb_cb);     // padding: the zero behind the staged band
    hipLaunchKernelGGL(k_pb_fill16, GRID(nquads + 2), d_qcband, nquads + 2, (uint16_t)0xffffu);   // quads outside plain runs: no band
    if (nsteps) {
        hipLaunchKernelGGL(k_pbr_step_base, GRID(nsteps), d_step_run, d_estart, d_step_excl, d_vpos, nsteps, d_step_base);
        hipLaunchKernelGGL(k_pbr_rows, dim3(nsteps), dim3(64), 0, st, reinterpret_cast<const uint4 *>(d_rcode),
                           reinterpret_cast<const uint4 *>(d_rrow), d_step_base, d_prow);
    }
    hipLaunchKernelGGL(k_pb_place, GRID(total), d_sorted, d_runid, d_runstart, d_vpos, d_fmt, total, d_prow, d_plcol, d_qcband);
    // first position of every band
    std::vector<u32> rstart;
    {
        u32 *d_rstart = nullptr, *d_band_pos = nullptr;
        LZX_TRY(ar.get(&d_rstart, (u64)nr + 1)); LZX_TRY(ar.get(&d_band_pos, (u64)nr + 1));
        hipLaunchKernelGGL(k_pb_bounds_u64, GRID(nr + 1), d_sorted, total, 40u, nr, d_rstart);
        hipLaunchKernelGGL(k_pb_band_pos, GRID(nr + 1), d_rstart, d_runid, d_vpos, nr, total, (u32)len, d_band_pos);
        LZX_TRY(pb_download(st, d_band_pos, (size_t)nr + 1, rstart));
        ar.drop(d_rstart); ar.drop(d_band_pos);
    }
#ifdef LZX_DEBUG_KNOBS
    if (getenv("LZX_PB_STATS")) {   // run-length histogram (entries per class), for DESIGN.md
        std::vector<u32> rs;
        LZX_TRY(pb_download(st, d_runstart, nruns, rs));
        u64 cls[14] = {0}, cnt[14] = {0};
        for (u32 r = 0; r < nruns; ++r) {
            const u32 rl = (r + 1 < nruns ? rs[r + 1] : (u32)total) - rs[r];
            u32 b = 0;
            while ((2u << b) <= rl && b < 13) ++b;
            cls[b] += rl;
            ++cnt[b];
        }
        for (u32 b = 0; b < 14; ++b)
            fprintf(stderr, "[lzx pb stats] runs of %u..%u entries: %llu runs, %llu entries (%.1f %%)\n", 1u << b, (2u << b) - 1,
                    (unsigned long long)cnt[b], (unsigned long long)cls[b], 100.0 * cls[b] / (double)total);
    }
#endif
    {   // entries of the reduced runs (reporting only)
        std::vector<u32> rs;
        std::vector<uint8_t> fm(nruns);
        LZX_TRY(pb_download(st, d_runstart, nruns, rs));
        LZX_HIP(hipMemcpy(fm.data(), d_fmt, nruns, hipMemcpyDeviceToHost));
        for (u32 r = 0; r < nruns; ++r)
            if (fm[r]) red_entries += (r + 1 < nruns ? rs[r + 1] : (u32)total) - rs[r];
    }
#ifdef LZX_DEBUG_KNOBS
    if (getenv("LZX_PB_STATS")) {   // where the bytes of the tables go
        std::vector<u32> rs;
        std::vector<uint8_t> fm(nruns);
        LZX_TRY(pb_download(st, d_runstart, nruns, rs));
        LZX_HIP(hipMemcpy(fm.data(), d_fmt, nruns, hipMemcpyDeviceToHost));
        u64 nred = 0, npl = 0, pl_entries = 0, pl_pad8 = 0;
        for (u32 r = 0; r < nruns; ++r) {
            const u32 rl = (r + 1 < nruns ? rs[r + 1] : (u32)total) - rs[r];
            if (fm[r]) ++nred; else { ++npl; pl_entries += rl; pl_pad8 += (rl + 7u) & ~7u; }
        }
        std::vector<u32> se;
        LZX_TRY(pb_download(st, d_step_excl, (size_t)nsteps + 1, se));
        fprintf(stderr, "[lzx pb stats] entries %llu: reduced %llu in %llu runs -> %u steps = %llu padded entries (%.1f %% padding), %u pieces; "
                "plain %llu in %llu runs -> %llu padded to 8; values incl. run alignment %llu (quads %llu); row bands %u, column bands %u, runs %u\n",
                (unsigned long long)total, (unsigned long long)red_entries, (unsigned long long)nred, nsteps, (unsigned long long)epad_total,
                100.0 * ((double)epad_total - (double)red_entries) / std::max<double>(1.0, (double)epad_total), nsteps ? se[nsteps] : 0u,
                (unsigned long long)pl_entries, (unsigned long long)npl, (unsigned long long)pl_pad8, (unsigned long long)len,
                (unsigned long long)nquads, nr, nb, nruns);
    }
#endif
    LZX_HIP(hipStreamSynchronize(st));
    ar.drop(d_sorted); ar.drop(d_runid); ar.drop(d_runstart); ar.drop(d_estart); ar.drop(d_fmt); ar.drop(d_vpos);
    ar.drop(d_rrow); ar.drop(d_step_run); ar.drop(d_step_excl);

    // 5. scatter order: steps and quads sorted (stably) by column band
    std::vector<u32> sstart, qstart;
    if (nsteps) {
        u32 *skey = nullptr, *skey_s = nullptr, *sidx = nullptr, *ssorted = nullptr, *bstart = nullptr;
        LZX_TRY(ar.get(&skey, nsteps)); LZX_TRY(ar.get(&skey_s, nsteps)); LZX_TRY(ar.get(&sidx, nsteps)); LZX_TRY(ar.get(&ssorted, nsteps));
        hipLaunchKernelGGL(k_pb_iota_widen, GRID(nsteps), d_step_cband, nsteps, skey, sidx);
        LZX_TRY(pb_sort_pairs16(st, skey, skey_s, sidx, ssorted, nsteps));
        LZX_TRY(pb_alloc(&c->d_pbr_code, (u64)nsteps * 64 + 1));
        LZX_TRY(pb_alloc(&c->d_pbr_base, (u64)nsteps + 1));
        hipLaunchKernelGGL(k_pbr_steps, dim3(nsteps), dim3(64), 0, st, ssorted, reinterpret_cast<const uint4 *>(d_rcode), d_step_base,
                           c->d_pbr_code, c->d_pbr_base);
        LZX_TRY(ar.get(&bstart, (u64)nb + 1));
        hipLaunchKernelGGL(k_pb_bounds_u32, GRID(nb + 1), skey_s, nsteps, nb, bstart);
        LZX_TRY(pb_download(st, bstart, (size_t)nb + 1, sstart));
        ar.drop(skey); ar.drop(skey_s); ar.drop(sidx); ar.drop(ssorted); ar.drop(bstart);
    }
    ar.drop(d_rcode); ar.drop(d_step_cband); ar.drop(d_step_base);
    c->pbr_steps = nsteps;
    if (nquads) {
        u32 *qkey = nullptr, *qkey_s = nullptr, *qidx = nullptr, *qsorted = nullptr, *bstart = nullptr;
        LZX_TRY(ar.get(&qkey, nquads)); LZX_TRY(ar.get(&qkey_s, nquads)); LZX_TRY(ar.get(&qidx, nquads)); LZX_TRY(ar.get(&qsorted, nquads));
        hipLaunchKernelGGL(k_pb_iota_widen, GRID(nquads), d_qcband, nquads, qkey, qidx);
        LZX_TRY(pb_sort_pairs16(st, qkey, qkey_s, qidx, qsorted, nquads));
        ar.drop(qkey); ar.drop(qidx);
        // quads that belong to no plain run carry key 0xffff and sort behind every band
        LZX_TRY(ar.get(&bstart, (u64)nb + 1));
        hipLaunchKernelGGL(k_pb_bounds_u32, GRID(nb + 1), qkey_s, nquads, nb, bstart);
        LZX_TRY(pb_download(st, bstart, (size_t)nb + 1, qstart));
        const u64 live = qstart[nb];
        {
            uint2 *q_lcol = nullptr;
            LZX_TRY(pb_alloc(&q_lcol, live + 1));
            c->d_pb_lcol = reinterpret_cast<uint16_t *>(q_lcol);
        }
        LZX_TRY(pb_alloc(&c->d_pb_dst, live + 1));
        if (live)
            hipLaunchKernelGGL(k_pb_quads, GRID(live), qsorted, d_plcol, live, reinterpret_cast<uint2 *>(c->d_pb_lcol), c->d_pb_dst);
        LZX_HIP(hipStreamSynchronize(st));
        ar.drop(qkey_s); ar.drop(qsorted); ar.drop(bstart);
    }
    ar.drop(d_plcol); ar.drop(d_qcband);
    LZX_TRY(pb_units(c, st, sstart, qstart, nb, unit_cap));
#ifdef LZX_DEBUG_KNOBS
    if (c->pb_persist_opt == 1) LZX_TRY(pb_segments(c, st, sstart, qstart, nb, nsteps));   // static scatter schedule (experiment)
#endif
    LZX_HIP(hipGetLastError());

    // 6. conflict-free LDS slots for the gather pass
    u32 *d_rstart_pad = nullptr, *d_step0 = nullptr;
    uint8_t *d_occ = nullptr;
    LZX_TRY(ar.get(&d_rstart_pad, (u64)nr + 1));
    LZX_HIP(hipMemcpyAsync(d_rstart_pad, rstart.data(), sizeof(u32) * ((size_t)nr + 1), hipMemcpyHostToDevice, st));
    LZX_TRY(pb_alloc(&c->d_pb_lrow, len + 8)); LZX_TRY(pb_alloc(&c->d_pb_rep, (u64)nr));
    std::vector<u32> rep((size_t)nr, 1u);
    {
        std::vector<u32> step0((size_t)nr + 1);
        u64 steps = 0;
        for (u32 R = 0; R < nr; ++R) {
            step0[R] = (u32)steps;
            steps += (rstart[R + 1] - rstart[R] + 63u) / 64u;
        }
        step0[nr] = (u32)steps;
        LZX_TRY(ar.get(&d_step0, (u64)nr + 1)); LZX_TRY(ar.get(&d_occ, len + 8));
        LZX_HIP(hipMemcpyAsync(d_step0, step0.data(), sizeof(u32) * ((size_t)nr + 1), hipMemcpyHostToDevice, st));
        LZX_HIP(hipMemcpyAsync(c->d_pb_rep, rep.data(), sizeof(u32) * nr, hipMemcpyHostToDevice, st));
        LZX_HIP(hipMemsetAsync(d_occ, 0, len + 8, st));
        if (steps)
            hipLaunchKernelGGL(k_pb_occurrence, dim3((u32)steps), dim3(64), 0, st, d_prow, d_rstart_pad, d_step0, nr, d_occ,
                               c->d_pb_rep);
        LZX_HIP(hipMemcpyAsync(rep.data(), c->d_pb_rep, sizeof(u32) * nr, hipMemcpyDeviceToHost, st));
        LZX_HIP(hipStreamSynchronize(st));
        // replicas per row: enough for the worst step of the band, but the tile holds LZX_PB_RB slots
        for (u32 R = 0; R < nr; ++R) {
            const u32 rows = row0[R + 1] - row0[R];
            const u32 room = std::max(1u, LZX_PB_RB / std::max(rows, 1u));
            rep[R] = std::max(1u, std::min(std::min(rep[R], room), 64u));
        }
        LZX_HIP(hipMemcpyAsync(c->d_pb_rep, rep.data(), sizeof(u32) * nr, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_pb_slots, GRID(len), d_prow, d_occ, d_rstart_pad, c->d_pb_rep, nr, len, c->d_pb_lrow);
        LZX_HIP(hipStreamSynchronize(st));
    }

    // 7. gather items: one per band; a band above 2 targets of values is cut into about target-sized items whose
    //    per-row totals k_pb_finish adds in item order
    std::vector<u32> items, multi;
    u64 slots = 0;
    for (u32 R = 0; R < nr; ++R) {
        const u32 beg = rstart[R], end = rstart[R + 1];
        if (beg == end) continue;
        const u32 rows = row0[R + 1] - row0[R];
        if (end - beg > 2 * target) {
            const u32 cnt = (end - beg + target - 1) / target;
            const u32 piece = ((end - beg + cnt - 1) / cnt + 127u) & ~127u;   // whole 128-value blocks of the band
            u32 made = 0;
            for (u32 s = beg; s < end; s += piece, ++made) {
                items.push_back(R); items.push_back(s); items.push_back(std::min(end, s + piece));
                items.push_back((u32)(slots + (u64)made * rows));
            }
            for (u32 j = 0; j < rows; ++j) {
                multi.push_back(row0[R] + j); multi.push_back((u32)(slots + j)); multi.push_back(made); multi.push_back(rows);
            }
            slots += (u64)made * rows;
            if (slots >= 0xffffffffull) LZX_FAIL(LZX_ERR_LIMIT, "propagation blocking: item totals overflow 32-bit slots");
        } else {
            items.push_back(R); items.push_back(beg); items.push_back(end); items.push_back(0xffffffffu);
        }
    }
    c->pb_n_items = (u32)(items.size() / 4);
    c->pb_n_multi = (u32)(multi.size() / 4);
    // one item per workgroup at a time, two workgroups per CU
    // 16 wavefronts per CU (their private y tiles fill the LDS): two workgroups of eight, or four of four
    c->pb_gather_block = c->pb_gwaves_opt == 4 ? 256u : 512u;
    c->pb_gather_grid = std::min<u32>((u32)c->cu_count * (1024u / c->pb_gather_block), std::max(1u, c->pb_n_items));
    LZX_TRY(pb_alloc(&c->d_pb_beg, (u64)nr + 1));
    LZX_HIP(hipMemcpyAsync(c->d_pb_beg, rstart.data(), sizeof(u32) * ((size_t)nr + 1), hipMemcpyHostToDevice, st));
    const u32 group_cap = c->pb_group_opt >= 0 ? (u32)c->pb_group_opt : LZX_PB_GROUP;
    // product form of the gather pass: tickets over fat records (k_pb_gather3); the static longest-first lists
    // (k_pb_gather) remain in the debug library behind the knob pb_gather_tickets = 0
    bool g3 = c->pb_gather_block == 512u && c->pb_persist_opt <= 0;
#ifdef LZX_DEBUG_KNOBS
    if (c->pb_g3_opt == 0) g3 = false;
#endif
    c->pb_g3 = g3;
    std::vector<u32> g3_out;      // items after grouping {R | first band, begin | bands, end | values, slot | marker}
    std::vector<u64> g3_cost;
    if (group_cap > 0 && c->pb_gather_block == 512u && c->pb_persist_opt <= 0) {
        // Small bands (the low-degree end of the row order: thousands of bands of a few thousand values, each of which
        // cost a workgroup three dependent round trips and two barriers) are gathered by ONE wavefront each, up to eight
        // consecutive ones per item.  Then the items are dealt to the workgroups longest first, each to the workgroup with
        // the least work so far, and stored round by round, so that the kernel's strided loop
        // (item = workgroup + round * grid) walks that schedule: which workgroup adds which share of alpha stays fixed.
        std::vector<u32> out;
        auto small = [&](size_t i) {
            const u32 R = items[4 * i];
            return items[4 * i + 3] == 0xffffffffu && row0[R + 1] - row0[R] > 1 && items[4 * i + 2] - items[4 * i + 1] <= group_cap;
        };
        std::vector<u64> cost;
        const size_t ni = items.size() / 4;
        // as many bands per group as it takes to give every workgroup ONE round of small bands (1 M-vertex graph: about a
        // thousand small bands, groups of two: gather 0.031 -> 0.024 ms; groups of eight there leave three quarters of the
        // workgroups without work: 0.036 ms); up to 512 small bands: a band per workgroup, as before
        size_t n_small = 0;
        for (size_t i = 0; i < ni; ++i) n_small += small(i) ? 1 : 0;
        const size_t per_group = c->pb_group_force_opt > 0 ? std::min<size_t>(8, std::max<size_t>(2, (size_t)c->pb_group_force_opt)) : std::min<size_t>(8, (n_small + c->pb_gather_grid - 1) / std::max<u32>(1u, c->pb_gather_grid));
        for (size_t i = 0; i < ni;) {
            if (per_group < 2 || !small(i)) {
                const u32 R = items[4 * i];
                out.insert(out.end(), items.begin() + 4 * i, items.begin() + 4 * i + 4);
                cost.push_back(10ull * (items[4 * i + 2] - items[4 * i + 1]) + 24ull * (row0[R + 1] - row0[R]) + 40000ull);
                ++i;
                continue;
            }
            size_t j = i;
            u64 vals = 0, rows = 0, widest = 0;
            while (j < ni && j - i < per_group && small(j) && items[4 * j] == items[4 * i] + (u32)(j - i)) {
                const u32 R = items[4 * j];
                vals += items[4 * j + 2] - items[4 * j + 1];
                widest = std::max<u64>(widest, items[4 * j + 2] - items[4 * j + 1]);
                rows += row0[R + 1] - row0[R];
                ++j;
            }
            out.push_back(items[4 * i]); out.push_back((u32)(j - i)); out.push_back((u32)vals); out.push_back(LZX_PB_ITEM_GROUP);
            // a group lasts as long as its widest band's wavefront: eight serial batches per Ki values
            cost.push_back(std::max<u64>(10ull * vals + 24ull * rows, 80ull * widest) + 40000ull);
            i = j;
        }
        if (g3) {
            g3_out = out;
            g3_cost = cost;
        }
        const u32 G = c->pb_gather_grid;
        const size_t no = cost.size();
        std::vector<u32> order(no);
        for (size_t i = 0; i < no; ++i) order[i] = (u32)i;
        std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b2) { return cost[a] > cost[b2]; });
        std::vector<std::vector<u32>> lists(G);
        std::priority_queue<std::pair<u64, u32>, std::vector<std::pair<u64, u32>>, std::greater<std::pair<u64, u32>>> heap;
        for (u32 w = 0; w < G; ++w) heap.push({0ull, w});
        size_t rounds = 0;
        for (u32 i : order) {
            auto [load, w] = heap.top();
            heap.pop();
            lists[w].push_back(i);
            rounds = std::max(rounds, lists[w].size());
            heap.push({load + cost[i], w});
        }
        items.assign(rounds * G * 4, 0u);
        for (size_t r = 0; r < rounds; ++r)
            for (u32 w = 0; w < G; ++w) {
                u32 *o = &items[4 * (r * G + w)];
                if (r < lists[w].size()) {
                    const u32 *it = &out[4 * (size_t)lists[w][r]];
                    o[0] = it[0]; o[1] = it[1]; o[2] = it[2]; o[3] = it[3];
                } else {
                    o[3] = LZX_PB_ITEM_NONE;
                }
            }
        c->pb_n_items = (u32)(items.size() / 4);
    }
#ifdef LZX_DEBUG_KNOBS
    if (c->pb_persist_opt > 0)
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
    }
#endif
    if (g3) {
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
    }
#ifdef LZX_DEBUG_KNOBS
    if (c->pb_stamps_opt > 0 && c->pb_persist_opt <= 0) {   // per-workgroup section stamps of the product gather pass
        LZX_TRY(pb_alloc(&c->d_pb_gstamps, 8 * (u64)c->pb_gather_grid));
        LZX_HIP(hipMemsetAsync(c->d_pb_gstamps, 0, sizeof(unsigned long long) * 8 * c->pb_gather_grid, st));
    }
#endif
    {   // split rows (the first n_long64 local rows) that are also rows of a multi-item band
        std::vector<uint8_t> flag((size_t)c->n_long64 + 1, 0);
        for (size_t i = 0; i < multi.size(); i += 4)
            if (multi[i] < c->n_long64) flag[multi[i]] = 1;
        LZX_TRY(pb_alloc(&c->d_pb_long_multi, flag.size()));
        LZX_HIP(hipMemcpyAsync(c->d_pb_long_multi, flag.data(), flag.size(), hipMemcpyHostToDevice, st));
        LZX_HIP(hipStreamSynchronize(st));
    }
    LZX_TRY(pb_alloc(&c->d_pb_items, items.size())); LZX_TRY(pb_alloc(&c->d_pb_multi, multi.size())); LZX_TRY(pb_alloc(&c->d_pb_part, slots));
    if (!items.empty())
        LZX_HIP(hipMemcpyAsync(c->d_pb_items, items.data(), sizeof(u32) * items.size(), hipMemcpyHostToDevice, st));
    if (!multi.empty())
        LZX_HIP(hipMemcpyAsync(c->d_pb_multi, multi.data(), sizeof(u32) * multi.size(), hipMemcpyHostToDevice, st));

    LZX_TRY(pb_alloc(&c->d_pb_val, len + 8));
    LZX_HIP(hipMemsetAsync(c->d_pb_val, 0, sizeof(double) * (len + 8), st));
    LZX_HIP(hipStreamSynchronize(st));
    LZX_HIP(hipGetLastError());

    c->pb = true;
    c->pb_entries = total;
    c->pbr_entries = red_entries;
    c->pb_values = len;
    c->pb_nr = nr;
    constexpr u32 waves_per_wg = LZX_PB_GATHER_BLOCK / 64;
    (void)waves_per_wg;
    c->pb_finish_grid = (c->pb_n_multi + c->n_long64 + LZX_VEC_BLOCK - 1) / LZX_VEC_BLOCK;   // + the split rows of k_spmv
    if (c->pb_g3) c->pb_finish_grid = std::max<u32>(c->pb_finish_grid, (c->pb_g3_items + LZX_VEC_BLOCK - 1) / LZX_VEC_BLOCK);   // + the gather pass's item partials
    return LZX_OK;
}
#undef GRID
}  // namespace

int lzx_pb_prepare(lzx_ctx *c, const u32 *d_code, const u32 *d_old_of_local, const u32 *d_deg_local,
                   const u32 *d_nh_off, const std::vector<u32> &h_nh, u64 total)
{
    const int rc = pb_prepare_impl(c, d_code, d_old_of_local, d_deg_local, d_nh_off, h_nh, total);
    if (rc != LZX_OK) {
        (void)hipStreamSynchronize(c->stream);
        lzx_pb_release(c);
    }
    return rc;
}

// can the staged-columns workgroups share the scatter pass's launch?  (not with the experimental forms of the pass)
bool lzx_pb_can_fuse(const lzx_ctx *c)
{
    if (!c->pb || !(c->pb_cb == LZX_PB_CB || c->pb_cb == 8192)) return false;
#ifdef LZX_DEBUG_KNOBS
    if (c->pb_persist_opt > 0 || getenv("LZX_ABLATE") || (c->phase_mask_opt & (4 | 8))) return false;
#endif
    return true;
}

int lzx_pb_launch(lzx_ctx *c, const double *x, const double *q_loc, double *v, double *partials, hipEvent_t chunk1_ready,
                  hipEvent_t v_ready, int phases, const SpmvArgs *fuse, u32 fuse_blocks, bool *fused)
{
    if (fused) *fused = false;
    const bool do_scatter = phases & 1, do_gather = phases & 2;
    if (!c->pb) {
        // no blocked tables on this rank: still order the stream behind the second chunk of the exchange, so that the next
        // collective on the main stream never starts while that all-gather is in flight on the exchange stream
        if (chunk1_ready && do_scatter) LZX_HIP(hipStreamWaitEvent(c->stream, chunk1_ready, 0));
        return LZX_OK;
    }
    const size_t lds1 = ((size_t)c->pb_cb + 2 + 16 * 66) * sizeof(double);
#ifdef LZX_DEBUG_KNOBS
    static const int ablate = getenv("LZX_ABLATE") ? atoi(getenv("LZX_ABLATE")) : 0;
    auto kern = c->pb_cb == 8192 ? (ablate ? k_pb_scatter<8192, true> : k_pb_scatter<8192, false>)
                                 : (ablate ? k_pb_scatter<LZX_PB_CB, true> : k_pb_scatter<LZX_PB_CB, false>);
#else
    const int ablate = 0;
    auto kern = c->pb_cb == 8192 ? k_pb_scatter<8192, false> : k_pb_scatter<LZX_PB_CB, false>;
#endif
    if (c->pb_units)
        LZX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    // both libraries run one workgroup per unit / static item lists (k_pb_scatter, k_pb_gather); the persistent forms are
    // round-2 experiments kept in the debug library (knob pb_persistent: 1 static schedule, 2 tickets): none was faster
#ifdef LZX_DEBUG_KNOBS
    const bool persistent = c->pb_persist_opt > 0 && !ablate;
    const size_t lds1p = lds1 + 16;
    auto kern2 = c->pb_cb == 8192 ? k_pb_scatter2<8192> : k_pb_scatter2<LZX_PB_CB>;
    if (c->pb_units && persistent)
        LZX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1p));
    auto kern3 = c->pb_cb == 8192 ? k_pb_scatter3<8192> : k_pb_scatter3<LZX_PB_CB>;
    const bool fixed = persistent && c->pb_persist_opt != 2;   // 2 = the ticket-driven form
    if (c->pb_units && fixed)
        LZX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1p));
#endif
    auto scatter = [&](u32 u0, u32 u1, bool may_fuse) {
        if (u1 <= u0 && !(may_fuse && fuse && fused)) return;
#ifdef LZX_DEBUG_KNOBS
        if (fixed) {
            // schedule 0 = the bands of chunk 0 (units [0, pb_units0)), schedule 1 = the rest; a call for all units runs both
            for (int part = 0; part < 2; ++part) {
                const bool wanted = part == 0 ? u0 == 0 : u1 == c->pb_units && c->pb_units0 < c->pb_units;
                const u32 grid = c->pb_seg_groups[part];
                if (!wanted || !grid) continue;
                hipLaunchKernelGGL(kern3, dim3(grid), dim3(1024), lds1p, c->stream, c->d_pb_seg, c->d_pb_seg_begin + c->pb_seg_first[part],
                                   c->d_pbr_code, c->d_pbr_base, reinterpret_cast<const uint2 *>(c->d_pb_lcol), c->d_pb_dst, x, c->xlen,
                                   c->d_pb_val, c->d_pb_stamps ? c->d_pb_stamps + 4096 * part : nullptr);
            }
            return;
        }
        if (persistent) {
            const u32 q = u0 == 0 ? 0u : 1u, n = u1 - u0;
            const u32 grid = std::min<u32>(n, (u32)c->cu_count * (c->pb_cb == 8192 ? 2u : 1u));
            hipLaunchKernelGGL(kern2, dim3(grid), dim3(1024), lds1p, c->stream, c->d_pb_unit + 5 * (size_t)u0, n, c->d_pb_queue + q,
                               c->pb_qbase[q], c->d_pbr_code, c->d_pbr_base, reinterpret_cast<const uint2 *>(c->d_pb_lcol),
                               c->d_pb_dst, x, c->xlen, c->d_pb_val, c->d_pb_stamps ? c->d_pb_stamps + 4096 * q : nullptr);
            c->pb_qbase[q] += n + grid;
            return;
        }
#endif
        if (may_fuse && fuse && fused && !ablate && (c->pb_cb == LZX_PB_CB || c->pb_cb == 8192) && u0 == 0) {
            // the staged-columns workgroups share the launch of the scatter units (k_pb_scatter_spmv)
            auto kf = c->pb_cb == 8192 ? k_pb_scatter_spmv<8192> : k_pb_scatter_spmv<LZX_PB_CB>;
            const size_t ldsf = std::max(lds1, c->spmv_lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf);
            hipLaunchKernelGGL(kf, dim3(u1 + fuse_blocks), dim3(1024), ldsf, c->stream, c->d_pb_unit, u1, c->d_pbr_code, c->d_pbr_base,
                               reinterpret_cast<const uint2 *>(c->d_pb_lcol), c->d_pb_dst, x, c->xlen, c->d_pb_val, *fuse, c->fuse_opt == 1 ? 0u : 1u);
            *fused = true;
            return;
        }
        hipLaunchKernelGGL(kern, dim3(u1 - u0), dim3(1024), lds1, c->stream, c->d_pb_unit + 5 * (size_t)u0, c->d_pbr_code,
                           c->d_pbr_base, reinterpret_cast<const uint2 *>(c->d_pb_lcol), c->d_pb_dst, x, c->xlen, c->d_pb_val, ablate);
    };
    // column bands of chunk 0 first; the rest once the second chunk of the exchange has arrived
    if ((c->phase_mask_opt & 4) || !do_scatter) {
        // experiment: gather pass alone (reads values a previous SpMV left); or the scatter pass was launched earlier
    } else if (chunk1_ready) {
        scatter(0, c->pb_units0, true);    // with the staged-columns workgroups ahead of the chunk-0 units when the caller asks
        LZX_HIP(hipStreamWaitEvent(c->stream, chunk1_ready, 0));
        scatter(c->pb_units0, c->pb_units, false);
    } else {
        scatter(0, c->pb_units, true);
    }
    if (c->trace && do_scatter) LZX_HIP(hipEventRecord(c->trace_ev[3], c->stream));
    if (!do_gather) {
        LZX_HIP(hipGetLastError());
        return LZX_OK;
    }
    if (v_ready) LZX_HIP(hipStreamWaitEvent(c->stream, v_ready, 0));   // the staged-columns kernel wrote the v this pass adds into
    const size_t lds2 = ((size_t)(LZX_PB_GATHER_BLOCK / 64) * (LZX_PB_RB + 8) + LZX_PB_GATHER_BLOCK / 64) * sizeof(double);
    const size_t lds3 = lds2 + ((size_t)LZX_PB_GATHER_BLOCK + 2) * sizeof(double);   // + fold scratch and the two ticket words
    LZX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pb_gather<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    if (c->phase_mask_opt & 8) {
        // experiment: scatter pass alone
#ifdef LZX_DEBUG_KNOBS
    } else if (persistent) {
        const u32 block = c->pb_gather_block;
        const size_t lds2p = ((size_t)(block / 64) * (LZX_PB_RB + 8) + block / 64) * sizeof(double) + 16;
        auto gk = block == 256 ? k_pb_gather2<256> : k_pb_gather2<512>;
        LZX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2p));
        hipLaunchKernelGGL(gk, dim3(c->pb_gather_grid), dim3(block), lds2p, c->stream,
                           reinterpret_cast<const uint4 *>(c->d_pb_items2), c->d_pb_wg_begin, c->d_pb_lrow, c->d_pb_val, v, q_loc,
                           c->d_pb_part, partials, c->d_pb_stamps ? c->d_pb_stamps + 8192 : nullptr);
    } else if (c->pb_g3 && c->pb_stamps_opt > 0 && c->d_pb_gstamps) {
        LZX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pb_gather3<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
        hipLaunchKernelGGL(k_pb_gather3<true>, dim3(c->pb_gather_grid), dim3(LZX_PB_GATHER_BLOCK), lds3, c->stream,
                           reinterpret_cast<const uint4 *>(c->d_pb_grec), c->pb_g3_items, c->d_pb_gqueue, c->pb_gq_base, c->d_pb_lrow,
                           c->d_pb_val, v, q_loc, c->d_pb_part, c->d_pb_item_dot, c->d_pb_gstamps);
        c->pb_gq_base += c->pb_g3_items - c->pb_gather_grid;
    } else if (c->pb_stamps_opt > 0 && c->d_pb_gstamps) {
        LZX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pb_gather<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        hipLaunchKernelGGL(k_pb_gather<true>, dim3(c->pb_gather_grid), dim3(LZX_PB_GATHER_BLOCK), lds2, c->stream,
                           reinterpret_cast<const uint4 *>(c->d_pb_items), c->pb_n_items, c->d_pb_row0, c->d_pb_rep, c->d_pb_beg, c->d_pb_lrow,
                           c->d_pb_val, v, q_loc, c->d_pb_part, partials, c->d_pb_gstamps);
#endif
    } else if (c->pb_g3) {
        LZX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pb_gather3<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
        hipLaunchKernelGGL(k_pb_gather3<false>, dim3(c->pb_gather_grid), dim3(LZX_PB_GATHER_BLOCK), lds3, c->stream,
                           reinterpret_cast<const uint4 *>(c->d_pb_grec), c->pb_g3_items, c->d_pb_gqueue, c->pb_gq_base, c->d_pb_lrow,
                           c->d_pb_val, v, q_loc, c->d_pb_part, c->d_pb_item_dot, nullptr);
        c->pb_gq_base += c->pb_g3_items - c->pb_gather_grid;   // what the launch advances the ticket counter by (below)
    } else
    hipLaunchKernelGGL(k_pb_gather<false>, dim3(c->pb_gather_grid), dim3(LZX_PB_GATHER_BLOCK), lds2, c->stream,
                       reinterpret_cast<const uint4 *>(c->d_pb_items), c->pb_n_items, c->d_pb_row0, c->d_pb_rep, c->d_pb_beg, c->d_pb_lrow,
                       c->d_pb_val, v, q_loc, c->d_pb_part, partials, nullptr);
    if (c->pb_finish_grid)
        hipLaunchKernelGGL(k_pb_finish, dim3(c->pb_finish_grid), dim3(LZX_VEC_BLOCK), 0, c->stream,
                           reinterpret_cast<const uint4 *>(c->d_pb_multi), c->pb_n_multi, c->d_pb_part, c->d_item_first,
                           c->d_long_partial, c->d_pb_long_multi, c->n_long64, v, q_loc, partials + (c->pb_g3 ? 0u : c->pb_gather_grid),
                           c->pb_g3 ? c->d_pb_item_dot : nullptr, c->pb_g3 ? c->pb_g3_items : 0u);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

// lzx_pb.hip -- propagation-blocked SpMV for the entries whose column is NOT staged in LDS by k_spmv.
//
// Why: past the 4 MiB per-XCD L2 a random 8-byte gather of x costs a whole 128-byte fabric transaction
// (tools/gather_bench.hip: 55 Ggather/s = 7 TB/s of traffic), and even L2-resident gathers top out near
// 200 Ggather/s, so the plain CSR gather moves 8x the algorithmic bytes (profiles/archive/r1_c3_baseline_pmc.json: 15 GB per
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
//     PLAIN runs (shorter): padded to whole quads (LZX_PB_ALIGN = 4 entries); a lane takes a QUAD of four entries (one 8-byte load of their
//       columns-in-band, one 4-byte load of the quad's value slot), looks the four values up and writes them with
//       two 16-byte stores (padding reads a zero kept behind the staged band).
//     Value slots are in gather order, so a run is one contiguous, 32-byte aligned stretch of writes.
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
#include "lzx_pb_shared.h"

namespace {

// 64-bit sort key: row band (23 bits) | column band (16) | row in band (10) | column in band (15: bands of up to 32 Ki columns;
// the reduced codes keep their bit 15 for the end-of-piece flag)
constexpr u32 KEY_LCOL_BITS = 15, KEY_LROW_SHIFT = KEY_LCOL_BITS, KEY_CBAND_SHIFT = KEY_LROW_SHIFT + 10, KEY_RBAND_SHIFT = KEY_CBAND_SHIFT + 16;
constexpr u32 KEY_LCOL_MASK = (1u << KEY_LCOL_BITS) - 1u;
__device__ __forceinline__ u64 pack_key(u32 rband, u32 cband, u32 lrow, u32 lcol)
{
    return ((u64)rband << KEY_RBAND_SHIFT) | ((u64)cband << KEY_CBAND_SHIFT) | ((u64)lrow << KEY_LROW_SHIFT) | (u64)lcol;
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
    head[i] = (i == 0 || (keys[i] >> KEY_CBAND_SHIFT) != (keys[i - 1] >> KEY_CBAND_SHIFT)) ? 1u : 0u;
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

// Values a run hands to the gather pass, padded to `align` (so every run starts on a quad's 32-byte boundary): the pieces of
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
    prow[pos] = (uint16_t)((k >> KEY_LROW_SHIFT) & 0x3ffu);
    plcol[pos] = (uint16_t)(k & KEY_LCOL_MASK);
    if ((pos & 3u) == 0) quad_cband[pos >> 2] = (uint16_t)((k >> KEY_CBAND_SHIFT) & 0xffffu);
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

// slot word of padded position p: slot | round << 10.  slot = row * rep + (occurrence mod rep) in the wave-private y tile, where
// `occurrence` counts the earlier lanes of the SAME ds_add_f64 instruction that carry the same row (k_pb_occurrence); round =
// occurrence / rep.  The gather pass adds round 0 with one instruction and every later round with an instruction of its own
// (tile_add2), so no two lanes of one instruction ever add into the same address: the order of every sum is fixed by the
// program, not by how the LDS unit resolves same-address lanes (round 4: bit-reproducibility by construction; until then
// the replica was occurrence mod rep and the rare overflow shared a slot inside one instruction).  Padding values are
// zeros (d_pb_val is cleared once and padding is never written with anything else): they go to slot 0, round 0, where
// adding them changes nothing in whatever order.
__global__ void k_pb_slots(const uint16_t *prow, const uint8_t *occ, const u32 *rstart_pad, const u32 *band_rep, u32 nr,
                           u64 count, uint16_t *lslot, const u32 no_rounds /* debug A/B only: same-slot lanes left to the LDS unit, as in rounds 1 - 3 */)
{
    const u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= count) return;
    const u32 r = prow[p];
    if (r == 0xffffu) {
        lslot[p] = 0;
        return;
    }
    u32 lo = 0, hi = nr;   // band of p
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (rstart_pad[mid] <= p) lo = mid; else hi = mid;
    }
    const u32 rep = band_rep[lo];
    lslot[p] = (uint16_t)((r * rep + (occ[p] % rep)) | (no_rounds ? 0u : (occ[p] / rep) << 10));   // rows * rep <= 1024, occurrence < 64
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
    const u32 lrow = (u32)((k >> KEY_LROW_SHIFT) & 0x3ffu);
    const bool last = (off & (LZX_PBR_STEP - 1)) == LZX_PBR_STEP - 1 || i + 1 == count || runid_incl[i + 1] - 1 != r ||
                      (u32)((keys[i + 1] >> KEY_LROW_SHIFT) & 0x3ffu) != lrow;
    rcode[pos] = (uint16_t)((k & KEY_LCOL_MASK) | (last ? 0x8000u : 0u));
    rrow[pos] = (uint16_t)lrow;
    if ((off & (LZX_PBR_STEP - 1)) == 0) {
        step_cband[pos / LZX_PBR_STEP] = (uint16_t)((k >> KEY_CBAND_SHIFT) & 0xffffu);
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
// (The LZX_ABLATE switches behind DESIGN.md's ablation numbers live in a copy of this body in lzx_pb_dbg.hip.)
// Cross-lane moves of the reduced step's fixed-order carry (below): DPP controls of the GFX9 family -- row_shr:n moves inside a
// row of 16 lanes, row_bcast:15 / row_bcast:31 hand the last lane of a row / of the first half on to the rows named by the
// row mask, wave_shr:1 shifts the whole wavefront by one lane.  A lane without a source keeps the `old` operand: 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u32 dpp_u32(u32 v)
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
// What holder lane L of a reduced step starts its running sum from: the tails (what follows a lane's last piece end; its whole
// sum if it has none) of the lanes from the previous holder up to L - 1 -- a SEGMENTED inclusive scan over the lanes with the
// holders as segment heads, shifted by one lane.  Six DPP steps (1, 2, 4, 8 inside the rows of 16, then across them), every
// lane adding the same operands in the same order in every run: the sum's order is fixed by the program.
__device__ __forceinline__ double pbr_carry_scan(double tail, bool has)
{
    double P = tail;
    u32 F = has ? 1u : 0u;
    // (the addend is selected, not the sum: a lane whose segment is closed adds + 0.0, which changes nothing)
#define LZX_SCAN_STEP(CTRL, RM)                          \
    {                                                    \
        const double ps = dpp_f64<CTRL, RM>(P);          \
        const u32 fs = dpp_u32<CTRL, RM>(F);             \
        P += F ? 0.0 : ps;                               \
        F |= fs;                                         \
    }
    // Round 5: only the steps some segment needs.  A step of distance d adds something only to a lane whose d predecessors
    // (itself included) hold no piece end; the holders are known as a lane mask, so that is a few scalar operations and a
    // wave-uniform branch per step -- scalar work, where the step itself is seven vector instructions.  In a typical step of
    // the 10 M-vertex graph (93 pieces in 64 lanes) nearly every lane holds a piece end and the scan shrinks to its last line.
    // A skipped step would have added + 0.0 everywhere: the sums are the same.  (The tests against rows of 16 are left
    // out -- row_shr does not cross them -- which can only run a step that was not needed.)
    const unsigned long long nh = ~__ballot(has);           // lanes without a piece end
    if (nh) {
        LZX_SCAN_STEP(0x111, 0xf)   // row_shr:1
        const unsigned long long q2 = nh & (nh << 1);       // lane i and lane i - 1
        if (q2) {
            LZX_SCAN_STEP(0x112, 0xf)   // row_shr:2
            const unsigned long long q4 = q2 & (q2 << 2);   // lanes i .. i - 3
            if (q4) {
                LZX_SCAN_STEP(0x114, 0xf)   // row_shr:4
                const unsigned long long q8 = q4 & (q4 << 4);
                if (q8) LZX_SCAN_STEP(0x118, 0xf)   // row_shr:8
            }
        }
        // across the rows of 16: a row's first lane without a piece end continues the row before it
        if (nh & ((1ull << 16) | (1ull << 48))) LZX_SCAN_STEP(0x142, 0xa)   // row_bcast:15 into rows 1 and 3
        if (nh & ((1ull << 32) | (1ull << 48))) LZX_SCAN_STEP(0x143, 0xc)   // row_bcast:31 into rows 2 and 3
    }
#undef LZX_SCAN_STEP
    return dpp_f64<0x138, 0xf>(P);   // wave_shr:1: lane L takes lane L - 1's prefix (lane 0: nothing before it)
}

// the scatter pass's table loads (codes, value slots): each is read exactly once per SpMV.  NTC: as non-temporal loads, so that
// they do not take Infinity-Cache room from the values the pass is writing (which the gather pass then finds there)
template <bool NTC> __device__ __forceinline__ uint4 sld(const uint4 *p)
{
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    if (!NTC) return *p;
    const u4v t = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(p));
    return make_uint4(t.x, t.y, t.z, t.w);
}
template <bool NTC> __device__ __forceinline__ uint2 sld(const uint2 *p)
{
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    if (!NTC) return *p;
    const u2v t = __builtin_nontemporal_load(reinterpret_cast<const u2v *>(p));
    return make_uint2(t.x, t.y);
}
template <bool NTC> __device__ __forceinline__ u32 sld(const u32 *p) { return NTC ? __builtin_nontemporal_load(p) : *p; }

// SCAN: a row that spans lanes is summed across them by pbr_carry_scan (registers only, order fixed by the program);
// otherwise through wave-private LDS carry slots (the round-1 .. 3 form: one ds_add_f64 per lane, whose same-slot lanes the
// LDS unit orders).
template <u32 CB, bool SCAN, bool NTC>
__device__ __forceinline__ void
pb_scatter_body(const u32 *unit, const uint4 *scode, const u32 *sbase, const uint2 *q_lcol, const u32 *q_dst,
                const double *__restrict__ x, u64 xlen, double *val, const u32 ublock)
{
    extern __shared__ __attribute__((aligned(16))) double tile[];   // CB staged values + a zero for padding
    const u32 band = unit[5 * ublock];
    const u64 base = (u64)band * CB;
    // staging is dead time for this CU (the tile leaves room for one workgroup): all eight 16-byte loads of a
    // thread are issued before the first LDS write, so it costs one memory round trip
    if (base + CB <= xlen) {
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
    if (!SCAN)
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
        // Round 5 form of the step (SCAN): one pass forms every piece's LOCAL sum (val[e], kept in registers) and, as what is left
        // in the running sum behind the lane's last piece end, the tail the carry scan needs; the stores follow the scan, and the
        // lane's FIRST piece -- the one that continues a row from the lanes before -- takes the carried sum on its way out.
        // Against the round-4 form (a masked tail pass, then the running sum started from the carry: kept below for the LDS-carry
        // shape) that is a third fewer vector instructions per step: no `last`, no masked second pass over the eight values, the
        // piece-end flags tested where they lie in the code words (an odd entry's flag is its word's sign bit) and held as lane
        // masks.  The sums are the same up to the place of one addition: (x0 + x1 + ..) + carried instead of ((carried + x0) + x1) ..
        auto body_scan = [&](const uint4 &c, u32 pos) {
            const u32 w[4] = {c.x, c.y, c.z, c.w};
            double xv[8];
            bool f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xv[e] = tile[pbr_half(c, e) & 0x7fffu];
                f[e] = (e & 1) ? ((int)w[e >> 1] < 0) : ((w[e >> 1] & 0x8000u) != 0u);
            }
            const bool has = ((c.x | c.y | c.z | c.w) & 0x80008000u) != 0u;
            double pv[8];
            double run = 0.0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                run += xv[e];
                pv[e] = run;
                run = f[e] ? 0.0 : run;
            }
            double carry = pbr_carry_scan(run, has);   // (only read by lanes that hold a piece end)
            double *out = val + pos;   // wave-uniform: the step's first value slot
            u32 done = 0;              // pieces of the planes before this one
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned long long m = __ballot(f[e]);
                if (m) {               // scalar branch: steps of few long rows have mostly empty planes
                    if (f[e]) {
                        out[done + lanes_below(m)] = pv[e] + carry;
                        carry = 0.0;
                    }
                    done += (u32)__popcll(m);
                }
            }
        };
        auto body_lds = [&](const uint4 &c, u32 pos) {
            double xv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[e] = tile[pbr_half(c, e) & 0x7fffu];
            u32 ends = 0;   // bit e: entry e closes a piece
#pragma unroll
            for (int e = 0; e < 8; ++e) ends |= pbr_flag(c, e) << e;
            const bool has = ends != 0;
            const int last = has ? 31 - __clz((int)ends) : -1;
            double tail = 0.0;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (e > last) tail += xv[e];
            double s;
            if (SCAN) {
                const double carried = pbr_carry_scan(tail, has);
                s = has ? carried : 0.0;
            } else {
                const unsigned long long holders = __ballot(has);
                const unsigned long long before = holders & ((1ull << lane) - 1ull);
                const u32 from = before ? 64u - (u32)__clzll((long long)before) : 0u;   // 1 + last holder before this lane
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
                        out[done + lanes_below(m)] = s;
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
                c[u] = sld<NTC>(scode + (size_t)(s + u * W) * 64 + lane);
                b[u] = (u32)__builtin_amdgcn_readfirstlane((int)sbase[s + u * W]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (SCAN) body_scan(c[u], b[u]);
                else body_lds(c[u], b[u]);
            }
        }
        if (s < end) {   // up to three more steps: requested together (wave-uniform predicates), not one round trip each
            uint4 c[3];
            u32 b[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (s + u * W < end) {
                    c[u] = sld<NTC>(scode + (size_t)(s + u * W) * 64 + lane);
                    b[u] = (u32)__builtin_amdgcn_readfirstlane((int)sbase[s + u * W]);
                }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (s + u * W < end) {
                    if (SCAN) body_scan(c[u], b[u]);
                    else body_lds(c[u], b[u]);
                }
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
                c[u] = sld<NTC>(q_lcol + j + u * 64);
                d[u] = sld<NTC>(q_dst + j + u * 64);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                double2 lo, hi;
                lo.x = tile[c[u].x & 0xffffu];
                lo.y = tile[c[u].x >> 16];
                hi.x = tile[c[u].y & 0xffffu];
                hi.y = tile[c[u].y >> 16];
                double2 *out = reinterpret_cast<double2 *>(val + d[u]);   // 32-byte aligned: slots of a quad
                out[0] = lo;
                out[1] = hi;
            }
        }
        if (wend < blk + 256u) {   // a partly filled block: up to three more quads per lane, requested together
            uint2 c[3];
            u32 d[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const u32 jj = j + u * 64 < wend ? j + u * 64 : blk;     // clamped: unconditional loads
                c[u] = sld<NTC>(q_lcol + jj);
                d[u] = sld<NTC>(q_dst + jj);
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

template <u32 CB, bool SCAN, bool NTC>
__global__ void __launch_bounds__(1024)
k_pb_scatter(const u32 *unit, const uint4 *scode, const u32 *sbase, const uint2 *q_lcol, const u32 *q_dst,
             const double *__restrict__ x, u64 xlen, double *val)
{
    pb_scatter_body<CB, SCAN, NTC>(unit, scode, sbase, q_lcol, q_dst, x, xlen, val, blockIdx.x);
}

// Scatter pass and staged-columns kernel in ONE launch (single GPU, 16 Ki bands): the persistent workgroups of the
// staged-columns kernel (lzx_spmv_body.h) and the scatter units share a grid.  Both only read x and write different things
// (v / values), one workgroup of either kind fits a CU, and workgroups are dispatched in index order.  spmv_first (the
// default): the 256 staged-columns workgroups start on every CU at once and the scatter units follow as CUs come free, so
// the launch ends with the small tapered units instead of a second ramp and drain; the other order (debug knob
// fuse_staged = 1: staged columns behind the units, filling the scatter pass's tail while the write-back of its values
// drains in their shadow) measures the same.  C3: SpMV 0.616 -> 0.601 ms against separate launches.
template <u32 CB, bool SCAN, bool NTC>
__global__ void __launch_bounds__(1024)
k_pb_scatter_spmv(const u32 *unit, u32 n_units, const uint4 *scode, const u32 *sbase, const uint2 *q_lcol, const u32 *q_dst,
                  const double *__restrict__ x, u64 xlen, double *val, const SpmvArgs a, const u32 spmv_first)
{
    const u32 spmv_blocks = gridDim.x - n_units;
    if (spmv_first) {
        if (blockIdx.x < spmv_blocks) spmv_body<2, false>(a, blockIdx.x, spmv_blocks);
        else pb_scatter_body<CB, SCAN, NTC>(unit, scode, sbase, q_lcol, q_dst, x, xlen, val, blockIdx.x - spmv_blocks);
        return;
    }
    if (blockIdx.x < n_units) pb_scatter_body<CB, SCAN, NTC>(unit, scode, sbase, q_lcol, q_dst, x, xlen, val, blockIdx.x);
    else spmv_body<2, false>(a, blockIdx.x - n_units, spmv_blocks);
}

// item table entry: {row band, begin, end, slot} in gather positions; slot == 0xffffffff: the item is its band's
// only one and adds straight into v; otherwise it is one of several items of its band and leaves its per-row totals
// in part[slot + row in band] for k_pb_finish.
// One WORKGROUP per item: its eight wavefronts take the item's 128-value blocks round-robin (so the workgroup reads 8
// consecutive KiB at a time -- a few hundred sequential streams chip-wide instead of four thousand), each adding into
// its own y tile; the tiles are folded in wavefront order.
// STAMP (debug library, option pb_stamps): wavefront 0 of every workgroup keeps 100 MHz time stamps per section --
// stamps[8 * workgroup ..]: start, end, ticks zeroing tiles + reading item records, ticks streaming, ticks in barriers
// before the fold, ticks folding, items | s_memtime cycles << 16, values.
// the gather pass's stream loads: every value and slot is read exactly once.  NT: as non-temporal loads -- they do not
// displace what the scatter pass has just written from the Infinity Cache, so part of the value stream is read back from
// there instead of from HBM (10 M-vertex graph: SpMV 0.630 -> 0.575 ms, six of six alternating processes,
// profiles/r3_nt_ab.txt); chosen by the size of the stream (lzx_pb_launch): a stream that fits the caches whole (1 M-vertex
// graph, 33 MB) is read 5 % faster through them.
template <bool NT>
__device__ __forceinline__ double2 gld_val2(const double *p)
{
    typedef double d2v __attribute__((ext_vector_type(2)));
    const d2v t = NT ? __builtin_nontemporal_load(reinterpret_cast<const d2v *>(p)) : *reinterpret_cast<const d2v *>(p);
    return make_double2(t.x, t.y);
}
template <bool NT>
__device__ __forceinline__ u32 gld_slot2(const uint16_t *p)
{
    return NT ? __builtin_nontemporal_load(reinterpret_cast<const u32 *>(p)) : *reinterpret_cast<const u32 *>(p);
}

// The gather pass's LDS adds.  A slot word is slot | round << 10 (k_pb_slots): values of one ds_add_f64 instruction that share a
// slot carry different rounds, so that no two lanes of one instruction ever add into the same address and every slot receives
// its addends in program order (LDS operations of one wavefront execute in issue order).
//   tile_add2: one pair of instructions -- the even and the odd values of a 128-value block, lane l holding values 2 l and
//     2 l + 1.  Round 0 is branch-free and costs one instruction per value more than a plain add: a word of a later round is
//     >= 1024 and is clamped to the lane's own spare slot behind the tile, where its value lands harmlessly (the spare slots
//     are never read).  Later rounds -- a second value of a row inside one instruction is common, consecutive runs of a row
//     band cover the same rows: 20 - 90 % of the blocks hold one -- follow behind ONE wave-uniform branch per pair, each round an
//     instruction of its own.  (Measured on the 1 M-vertex graph, whose 16 us pass is latency-bound: exec-masked round-0 adds
//     + 10 us, a batch-wide slow path of sixteen masked adds per round + 7 us, rounds 0 and 1 both branch-free + 7 us.)
__device__ __forceinline__ void tile_add2(double *ytile, u32 lane, u32 sv, double ax, double ay)
{
    const u32 s0 = sv & 0xffffu, s1 = sv >> 16;
    const u32 spare = LZX_PB_RB + lane;
    atomicAdd(&ytile[min(s0, spare)], ax);
    atomicAdd(&ytile[min(s1, spare)], ay);
    if (__ballot((s0 | s1) >= 1024u)) {
        u32 r = 1;
        unsigned long long more;
        do {
            if ((s0 >> 10) == r) atomicAdd(&ytile[s0 & 1023u], ax);
            if ((s1 >> 10) == r) atomicAdd(&ytile[s1 & 1023u], ay);
            more = __ballot((s0 >> 10) > r || (s1 >> 10) > r);
            ++r;
        } while (more);
    }
}
// ... and one instruction of 64 consecutive values (the tail of a band)
__device__ __forceinline__ void tile_add1(double *ytile, u32 lane, u32 s0, double a)
{
    atomicAdd(&ytile[min(s0, LZX_PB_RB + lane)], a);
    if (__ballot(s0 >= 1024u)) {
        u32 r = 1;
        unsigned long long more;
        do {
            if ((s0 >> 10) == r) atomicAdd(&ytile[s0 & 1023u], a);
            more = __ballot((s0 >> 10) > r);
            ++r;
        } while (more);
    }
}

// NOSLOT (debug library, LZX_ABLATE_SLOTS=1: an ABLATION, wrong sums): the slot stream is not read at all -- what the pass
// would cost if its 2 bytes per value came for free (lever i)
template <bool STAMP, bool NT, bool NOSLOT = false>
__global__ void __launch_bounds__(LZX_PB_GATHER_BLOCK, 4)   // four wavefronts per SIMD = two workgroups per CU: at most 128 VGPRs (the stamped build took 129 and ran one per CU)
k_pb_gather(const uint4 *items, u32 n_static, u32 n_dyn, u32 *counter, double *item_dot, const u32 *band_row0, const u32 *band_rep,
            const u32 *band_beg, const uint16_t *lslot, const double *val, double *v, const double *__restrict__ q_loc, double *part,
            double *partials, unsigned long long *stamps, const u32 probe_l2g, const u32 *item_first, const double *long_partial,
            const uint8_t *long_mode, const u32 n_long)
{
    // Round 5: the SPLIT ROWS of the staged-columns kernel (local rows [0, n_long): their item totals, long_partial) are closed
    // HERE, by the fold that writes the row anyway, when the row's band is a single gather item (long_mode 0) -- on graphs without
    // multi-item bands (the 1 M-vertex benchmark graph) k_pb_finish then has nothing left to do and is not launched: three launches
    // per SpMV become two.  Same operands in the same order as k_pb_finish added them: v = (v + y) + s -- in a short loop BEHIND the
    // fold (inside it the extra live values cost the 128-VGPR build six spills).
    // first..first + stride * m: the rows THIS thread has just folded (it re-reads its own stores); almost every band lies beyond
    // n_long, where the wave-uniform test ends the matter.  mode 0: rows of a single-item band -- v and alpha; mode 1: rows of a
    // multi-item band, called by the band's FIRST item -- alpha only (their v is closed by k_pb_finish or, deferred, by
    // k_lazy_update, which add the same totals)
    auto close_split_rows = [&](u32 row0, u32 rows, u32 first, u32 stride, double &dot, u32 mode) {
        if (row0 >= n_long) return;
        const u32 lim = min(rows, n_long - row0);
        for (u32 j = first; j < lim; j += stride) {
            const u32 row = row0 + j;
            if (long_mode[row] != mode) continue;
            double sl = 0.0;
            for (u32 it = item_first[row]; it < item_first[row + 1]; ++it) sl += long_partial[it];
            if (mode == 0) v[row] += sl;
            dot += sl * q_loc[row];
        }
    };
    unsigned long long t_start = 0, t_mark = 0, t_zero = 0, t_stream = 0, t_bar = 0, t_fold = 0, n_vals = 0;
    u32 n_it = 0;
#define GSTAMP(acc) do { if (STAMP) { const unsigned long long t_ = wall_clock64(); acc += t_ - t_mark; t_mark = t_; } } while (0)
    unsigned long long c_start = 0;
    if (STAMP) { t_start = t_mark = wall_clock64(); c_start = __builtin_amdgcn_s_memtime(); }
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr u32 WAVES = LZX_PB_GATHER_BLOCK / 64;
    constexpr u32 TILE = LZX_PB_RB + 64;                 // + one spare slot per lane (tile_add2)
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 wv = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    double *ytile = lds + (size_t)wv * TILE;             // private to this wavefront
    double *wsum = lds + (size_t)WAVES * TILE;           // [WAVES]
    double dot = 0.0;
    // The workgroup's item list is static (item = workgroup + round * grid), so everything its items need is fetched ONCE,
    // for up to 16 rounds at a time: thread (round, wavefront) reads the item and -- dependent on it -- the band tables and
    // leaves one record per (round, wavefront) in LDS.  Before, every item began with those two dependent round trips
    // (14 us of a workgroup's 165 on the 10 M-vertex graph, profiles/r3_gather_stamps.txt).
    // record: beg, end, row0, rows, rep, item.w (slot / marker), live (0: this wavefront has nothing to do in a group item)
    //
    // Behind the static lists may come a DYNAMIC TAIL (n_dyn > 0: test shape pb_dyn_share; off by default): the cheapest items
    // (a fixed share of the pass's work, items[n_static .. n_static + n_dyn), dearest first) are not dealt by the host but drawn
    // from a counter by whichever workgroup has finished its list.  Equal bytes do not take workgroups equal time (the static
    // schedule ends between 122 and 192 us on the 10 M-vertex graph); the tail evens that out (170 .. 194 us) -- and the pass
    // ends when it did before, because its bound is the aggregate streaming rate: a workgroup that finishes early leaves its
    // bandwidth to the others (profiles/r3_gather_balance.txt).  Kept as a measured alternative, exercised by the parity tests.
    // A drawn item leaves its share of alpha in item_dot[ticket] (closed in ticket order by k_pb_finish), so alpha does not
    // depend on who drew what.  Every workgroup draws exactly one ticket >= n_dyn (its exit), so the counter ends at
    // n_dyn + grid: the workgroup that drew the last ticket puts it back to 0 for the next launch.
    constexpr u32 MAXR = 16;
    u32 *lrec = reinterpret_cast<u32 *>(wsum + WAVES);   // [MAXR][WAVES][8]
    u32 *ltick = lrec + MAXR * WAVES * 8;                // the ticket wavefront 0 drew
    auto uni = [](u32 x) { return (u32)__builtin_amdgcn_readfirstlane((int)x); };
    u32 it0 = blockIdx.x;
    bool dyn = false;
    for (;;) {
    if (!dyn && it0 >= n_static) dyn = true;
    if (dyn && n_dyn == 0) break;
    __syncthreads();                     // the previous batch's records (and the previous item's tiles) are no longer read
    u32 base = it0, stride = gridDim.x, limit = n_static, nrounds = MAXR, ticket = 0;
    if (dyn) {
        if (tid == 0) *ltick = atomicAdd(counter, 1u);
        __syncthreads();
        ticket = uni(*ltick);
        if (ticket >= n_dyn) {
            if (tid == 0 && ticket == n_dyn + gridDim.x - 1) *counter = 0u;   // the last draw of this launch
            break;
        }
        base = n_static + ticket; stride = 0; limit = base + 1; nrounds = 1;
        __syncthreads();                 // everybody has read the ticket before thread 0 may draw the next one
    }
    for (u32 t = tid; t < nrounds * WAVES; t += LZX_PB_GATHER_BLOCK) {
        const u32 r = t / WAVES, w = t % WAVES;
        const u32 it = base + r * stride;
        u32 *o = lrec + (size_t)t * 8;
        uint4 item = make_uint4(0u, 0u, 0u, LZX_PB_ITEM_NONE);
        if (it < limit) item = items[it];
        o[5] = item.w;
        o[6] = 0u;
        o[7] = 1u;
        if (item.w == LZX_PB_ITEM_GROUP) {
            // a group of item.y <= 8 small bands.  With fewer than five of them (small graphs, rank shares: one round of small
            // bands per workgroup) the idle wavefronts help: every band is streamed by `split` wavefronts, each into its own
            // tile -- wavefront w takes part w / y of band w % y -- and folded by its first one
            const u32 y = item.y, split = y <= 1u ? 8u : y <= 2u ? 4u : y <= 4u ? 2u : 1u;
            o[7] = split;
            if (w < y * split) {
                const u32 R = item.x + w % y;
                const u32 r0 = band_row0[R];
                o[0] = band_beg[R]; o[1] = band_beg[R + 1]; o[2] = r0; o[3] = band_row0[R + 1] - r0; o[4] = band_rep[R]; o[6] = 1u + (w / y) + (y << 8);
            }
        } else if (item.w != LZX_PB_ITEM_NONE) {
            const u32 R = item.x;
            const u32 r0 = band_row0[R];
            o[0] = item.y; o[1] = item.z; o[2] = r0; o[3] = band_row0[R + 1] - r0; o[4] = band_rep[R];
            o[6] = 1u + (item.y == band_beg[R] ? 2u : 0u);   // + 2: the band's first item (it forms the alpha share of the band's split rows)
        }
    }
    __syncthreads();
    const double dot_static = dot;       // a drawn item's share of alpha is kept apart (item_dot)
    if (dyn) dot = 0.0;
    for (u32 rr = 0; rr < nrounds && base + rr * stride < limit; ++rr) {
        const u32 *rec = lrec + ((size_t)rr * WAVES + wv) * 8;
        const u32 r_beg = uni(rec[0]), r_end = uni(rec[1]), r_row0 = uni(rec[2]), r_rows = uni(rec[3]), r_rep = uni(rec[4]);
        const u32 item_w = uni(rec[5]), r_live = uni(rec[6]), r_split = uni(rec[7]);
        if (item_w == LZX_PB_ITEM_NONE) continue;   // filler of the balanced schedule
        if (item_w == LZX_PB_ITEM_GROUP) {
            // a group of up to eight SMALL consecutive bands: each wavefront streams its band -- or, in a group of few bands, its
            // part of one (r_split wavefronts per band, blocks dealt round-robin) -- into its own tile; the band's first wavefront
            // folds.  Eight bands: no workgroup barrier, no cross-wavefront fold, eight bands' round trips in flight per workgroup
            // (the low-degree end of the row order is thousands of bands of a few thousand values); fewer: one barrier before the
            // fold, and a band's serial chain of batches is r_split times shorter (1 M-vertex graph: the pass's critical path).
            __syncthreads();                 // the previous item's fold (it reads every wavefront's tile) is done
            const u32 part = r_live ? (r_live & 0xffu) - 1u : 0u, gy = r_live >> 8;
            if (r_live) {
                const u32 rows = r_rows, rep = r_rep;
                const u32 beg = r_beg, end = r_end;
                const u32 slots = rows * rep;
                for (u32 j = lane; j < slots; j += 64) ytile[j] = 0.0;
                __builtin_amdgcn_wave_barrier();
                if (STAMP && part == 0) { ++n_it; n_vals += end - beg; }
                GSTAMP(t_zero);
                const u32 blocks = (end - beg) / 128u;
                u32 kb = part;
                for (; kb + 7 * r_split < blocks; kb += 8 * r_split) {
                    double2 av[8];
                    u32 sv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const u32 p = beg + (kb + u * r_split) * 128u + lane * 2;
                        av[u] = gld_val2<NT>(val + p);
                        sv[u] = NOSLOT ? (lane * 2u) | ((lane * 2u + 1u) << 16) : gld_slot2<NT>(lslot + p);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) tile_add2(ytile, lane, sv[u], av[u].x, av[u].y);
                }
                {   // up to seven more blocks and (the band's first wavefront) the band's tail (< 128 values): all fetched before the first add
                    double2 av[7];
                    u32 sv[7];
                    double tv[2] = {0.0, 0.0};
                    u32 ts[2] = {0u, 0u};
#pragma unroll
                    for (int u = 0; u < 7; ++u) {
                        if (kb + u * r_split < blocks) {           // wave-uniform
                            const u32 p = beg + (kb + u * r_split) * 128u + lane * 2;
                            av[u] = gld_val2<NT>(val + p);
                            sv[u] = NOSLOT ? (lane * 2u) | ((lane * 2u + 1u) << 16) : gld_slot2<NT>(lslot + p);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const u32 i = beg + blocks * 128u + lane + u * 64;
                        if (part == 0 && i < end) {
                            tv[u] = val[i];
                            ts[u] = lslot[i];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 7; ++u) {
                        if (kb + u * r_split < blocks) tile_add2(ytile, lane, sv[u], av[u].x, av[u].y);   // (wave-uniform)
                    }
                    tile_add1(ytile, lane, ts[0], tv[0]);
                    __builtin_amdgcn_wave_barrier();   // the tail goes 64 consecutive values per instruction, in order
                    tile_add1(ytile, lane, ts[1], tv[1]);
                }
#ifdef LZX_DEBUG_KNOBS
                if (probe_l2g) {
                    // PROBE (debug library, LZX_PROBE_L2G = entries per small band; wrong sums, timing only): what would it cost the
                    // pass to fetch some x values itself -- 8-byte gathers from a 1 MB window of x that stays in the L2s -- instead of
                    // receiving them through the value stream?  probe_l2g look-ups per band, eight in flight per lane.
                    const double *xt = q_loc + 16384;
                    for (u32 i = lane; i < probe_l2g; i += 64 * 8) {
                        double t8[8];
                        u32 h8[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            h8[u] = ((i + u * 64) * 2654435761u + r_row0 * 40503u) >> 7;
                            t8[u] = xt[h8[u] & 0x1ffffu];
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (i + u * 64 < probe_l2g) tile_add1(ytile, lane, (h8[u] >> 17) % slots, t8[u]);
                    }
                }
#endif
                __builtin_amdgcn_wave_barrier();
                GSTAMP(t_stream);
            }
            if (r_split > 1u) __syncthreads();   // (wave-uniform per item: every wavefront of the workgroup passes here) the partner tiles are complete
            if (r_live && part == 0u) {
                const u32 row0 = r_row0, rows = r_rows, rep = r_rep;
                // fold: the band's tiles in wavefront order, replicas in order; four rows per lane at a time, loads before stores
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
                            for (u32 pt = 0; pt < r_split; ++pt)
                                for (u32 t = 0; t < rep; ++t) y += ytile[(size_t)pt * gy * TILE + j * rep + t];
                            v[row0 + j] = vv[u] + y;
                            dot += y * qq[u];
                        }
                    }
                }
                close_split_rows(row0, rows, lane, 64u, dot, 0u);
                GSTAMP(t_fold);
            }
            continue;
        }
        const u32 beg = r_beg, end = r_end;
        if (STAMP) { ++n_it; n_vals += end - beg; }
        const u32 row0 = r_row0, rows = r_rows;
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
                if (item_w == 0xffffffffu) {
                    double vr = v[row0] + t;
                    dot += t * q_loc[row0];
                    if (row0 < n_long && long_mode[row0] == 0) {
                        double sl = 0.0;
                        for (u32 it = item_first[row0]; it < item_first[row0 + 1]; ++it) sl += long_partial[it];
                        vr += sl;
                        dot += sl * q_loc[row0];
                    }
                    v[row0] = vr;
                } else {
                    // one of several items of its band: the total is added to v later (k_pb_finish / k_lazy_update); its share of
                    // alpha is formed here, by the workgroup the static schedule gave the item to
                    part[item_w] = t;
                    dot += t * q_loc[row0];
                    if ((r_live & 2u) && row0 < n_long && long_mode[row0] == 1) {
                        double sl = 0.0;
                        for (u32 it = item_first[row0]; it < item_first[row0 + 1]; ++it) sl += long_partial[it];
                        dot += sl * q_loc[row0];
                    }
                }
            }
            continue;
        }
        const u32 rep = r_rep;
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
                av[u] = gld_val2<NT>(val + p);
                sv[u] = NOSLOT ? (lane * 2u) | ((lane * 2u + 1u) << 16) : gld_slot2<NT>(lslot + p);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) tile_add2(ytile, lane, sv[u], av[u].x, av[u].y);
        }
        for (; kb < blocks; kb += WAVES) {
            const u32 p = beg + kb * 128u + lane * 2;
            const double2 a = gld_val2<NT>(val + p);
            const u32 s = NOSLOT ? (lane * 2u) | ((lane * 2u + 1u) << 16) : gld_slot2<NT>(lslot + p);
            tile_add2(ytile, lane, s, a.x, a.y);
        }
        // the band's tail (< 128 values): 64 consecutive values per instruction, wavefront 0
        if (wv == 0)
            for (u32 i0 = beg + blocks * 128u; i0 < end; i0 += 64) {   // (every lane takes part in tile_add1's ballots)
                const u32 i = i0 + lane;
                tile_add1(ytile, lane, i < end ? (u32)lslot[i] : 0u, i < end ? val[i] : 0.0);
            }
        GSTAMP(t_stream);
        __syncthreads();
        GSTAMP(t_bar);
        // fold: wavefront tiles in order, replicas in order; every thread a few rows, loads before stores
        if (item_w == 0xffffffffu) {
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
            close_split_rows(row0, rows, tid, LZX_PB_GATHER_BLOCK, dot, 0u);
        } else {
            for (u32 j = tid; j < rows; j += LZX_PB_GATHER_BLOCK) {
                double y = 0.0;
                for (u32 w = 0; w < WAVES; ++w)
                    for (u32 t = 0; t < rep; ++t) y += lds[(size_t)w * TILE + j * rep + t];
                part[item_w + j] = y;
                dot += y * q_loc[row0 + j];   // (see the single-row item above)
            }
            if (r_live & 2u) close_split_rows(row0, rows, tid, LZX_PB_GATHER_BLOCK, dot, 1u);
        }
        GSTAMP(t_fold);
    }
    if (dyn) {
        const double idot = wave_sum_pb(dot);
        dot = dot_static;
        __syncthreads();                 // wsum free (the single-row path uses it too)
        if (lane == 0) wsum[wv] = idot;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (u32 w = 0; w < WAVES; ++w) t += wsum[w];
            item_dot[ticket] = t;
        }
    } else {
        it0 += MAXR * gridDim.x;
    }
    }   // batches: up to MAXR static rounds, or one drawn item
    if (STAMP && tid == 0) {
        unsigned long long *o = stamps + 8 * (size_t)blockIdx.x;
        o[0] = t_start; o[1] = wall_clock64(); o[2] = t_zero; o[3] = t_stream; o[4] = t_bar; o[5] = t_fold;
        o[6] = (unsigned long long)n_it | ((__builtin_amdgcn_s_memtime() - c_start) << 16);   // items | shader-clock cycles: the clock the kernel ran at
        o[7] = n_vals;
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

// v[row] += totals that were left for it, in their fixed order; alpha partials for those rows.  Threads [0, n_multi):
// rows of bands cut into several gather items (their per-item totals); threads behind them: the split rows of the
// staged-column kernel (their item totals, k_long_finish's job in plain mode) that no gather item covers (long_mode 2: a row
// without a blocked entry; mode 0 rows were closed by the gather pass's fold, mode 1 rows by their multi thread here).
// Launched only when there is such work (pb_prepare_impl: pb_finish_grid).
__global__ void __launch_bounds__(LZX_VEC_BLOCK)
k_pb_finish(const uint4 *multi /*[n]: row, first slot, items, slot stride*/, u32 n_multi, const double *part,
            const u32 *item_first, const double *long_partial, const uint8_t *long_mode, u32 n_long, double *v,
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
        v[row] += s;        // (the row's share of alpha was formed by the gather pass's items, round 5: this launch only completes v,
                            //  which is why k_lazy_update can do it instead -- lzx_ctx::pb_deferring)
    } else if (t - n_multi < n_long && long_mode[t - n_multi] == 2) {
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



}  // namespace

// Where the driver places the value stream decides part of the SpMV's speed, for the life of the allocation: the same
// tables with d_pb_val in five fresh allocations read 1.05 / 1.22 / 1.23 / 1.23 / 1.23 ms on the Erdos-Renyi benchmark graph
// and 0.520-0.531 ms on the R-MAT one, while moving any other array changes nothing (debug library, LZX_RELOC_SHOP;
// profiles/r4_placement.txt) -- the "process states" of rounds 2-4.  User space cannot choose physical memory, but it can
// ask again: the buffer is allocated up to `placement_trials` more times (the best so far and the last loser stay allocated
// meanwhile, so every candidate is different memory), the SpMV is timed with each (3 runs of x = 0: its time does not depend on
// the values) and the fastest is kept.  The buffer is scratch that only needs its zero padding, so a trial costs an allocation, a
// clear and three SpMVs.  Results are bit-identical whichever candidate wins.
int lzx_pb_place_values(lzx_ctx *c)
{
    c->place_tried = c->place_kept = 0;
    const size_t bytes = sizeof(double) * (c->pb_values + 8);
    // default: two more candidates (round 5, ADVICE r4: a library that may sit inside a host framework does not hold eight copies
    // of a stream at its hand-over by default; bench.py asks for seven and reports what they read); never more than 16 GB among
    // the candidates tried (the driver clears what it hands out: a 10 GB allocation takes most of a second)
    const u32 trials = (u32)std::min<u64>(c->place_opt >= 0 ? (u64)std::min<int64_t>(c->place_opt, 7) : 2ull,
                                          c->place_opt >= 0 ? 7ull : std::max<u64>(1, (16ull << 30) / std::max<size_t>(bytes, 1)));
    if (!c->pb || !c->d_pb_val || trials == 0 || bytes < LZX_PB_NT_BYTES) return LZX_OK;   // a stream the caches hold: nothing to choose
    LZX_HIP(hipSetDevice(c->device));
    const bool multi = lzx_exchanges(c);
    SpmvLaunch l{multi ? c->d_xbuf : c->d_ybuf, c->d_ybuf + (size_t)c->rank * c->n_loc_pad, c->d_v, c->d_partials};
    auto time_spmv = [&](float *best) -> int {
        *best = 1e30f;
        for (int r = 0; r < 4; ++r) {   // the first one warms the tables up
            LZX_HIP(hipEventRecord(c->ev_a, c->stream));
            LZX_TRY(lzx_launch_spmv(c, l));
            LZX_HIP(hipEventRecord(c->ev_b, c->stream));
            LZX_HIP(hipEventSynchronize(c->ev_b));
            float ms = 0.f;
            LZX_HIP(hipEventElapsedTime(&ms, c->ev_a, c->ev_b));
            if (r > 0 && ms < *best) *best = ms;
        }
        return LZX_OK;
    };
    // At most three candidates are alive at any time: the best so far, the one being timed, and the last loser -- which is freed
    // only AFTER the next allocation, so that the driver cannot hand the same memory out again as a "new" candidate.
    double *best_buf = c->d_pb_val, *loser = nullptr;
    u32 kept = 0;
    int rc = time_spmv(&c->place_ms[0]);
    for (u32 t = 1; rc == LZX_OK && t <= trials; ++t) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < 2 * bytes + (4ull << 30)) break;   // never at the price of the caller's memory
        double *fresh = nullptr;
        // (other memory types are no way out: hipDeviceMallocContiguous and hipDeviceMallocUncached candidates read 1.76 ms on the
        //  Erdos-Renyi graph and 0.70 ms on the R-MAT one against 1.05-1.24 / 0.53-0.55; fine-grained ones draw from the same lottery)
        if (hipMalloc(reinterpret_cast<void **>(&fresh), bytes) != hipSuccess) { (void)hipGetLastError(); break; }
        if (loser) { (void)hipFree(loser); loser = nullptr; }
        if (hipMemsetAsync(fresh, 0, bytes, c->stream) != hipSuccess) { (void)hipGetLastError(); loser = fresh; break; }
        c->d_pb_val = fresh;
        rc = time_spmv(&c->place_ms[t]);
        if (rc != LZX_OK) { loser = fresh; break; }
        c->place_tried = t + 1;
        if (c->place_ms[t] < c->place_ms[kept]) {
            kept = t;
            loser = best_buf;
            best_buf = fresh;
        } else {
            loser = fresh;
        }
    }
    if (c->place_tried == 0) c->place_tried = 1;
    (void)hipStreamSynchronize(c->stream);
    if (loser) (void)hipFree(loser);
    c->d_pb_val = best_buf;
    c->place_kept = kept;
    // the SpMV wrote v and the partials: leave them as the hand-over does
    LZX_HIP(hipMemsetAsync(c->d_v, 0, sizeof(double) * c->ldq, c->stream));
    LZX_HIP(hipMemsetAsync(c->d_pb_val, 0, bytes, c->stream));
    LZX_HIP(hipStreamSynchronize(c->stream));
    return rc;
}

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
    pb_free(c->d_pb_gstamps);
    pb_free(c->d_pb_item_dot);
    pb_free(c->d_pb_gcounter);
    pb_free(c->d_pb_multi);
    pb_free(c->d_pb_part);
    pb_free(c->d_pb_long_multi);
    pb_free(c->d_pb_mrow);
    c->pb_multi_limit = 0;
    c->pb_defer_ok = c->pb_deferring = false;
    pb_free(c->d_pbr_code);
    pb_free(c->d_pbr_base);
    c->pb = false;
    c->pb_entries = c->pb_values = c->pbr_entries = 0;
    c->pb_units = c->pb_units0 = c->pb_nr = c->pb_gather_grid = c->pb_n_items = c->pb_n_static = c->pb_n_dyn = c->pb_n_multi = c->pb_finish_grid = 0;
    c->pbr_steps = 0;
}

// block partials of alpha the blocked passes leave
u32 lzx_pb_partials(const lzx_ctx *c) { return c->pb ? c->pb_gather_grid + c->pb_finish_grid : 0; }

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
    // 18 Ki (144 KiB of the CU's 160 KiB): 4 % fewer (row, band) pairs -- values that cross the passes -- for 12 % more staging
    // per unit.  Test shape pb_column_band = 16384 | 18432.
    // Measured, every configuration in two consecutive processes (profiles/r4_wide_band.txt): 10 M-vertex R-MAT graph - 0.5 %
    // (- 1.9 % with 18 Ki staged values as well), 1 M-vertex graph - 2.8 %, Erdos-Renyi 10 M - 2 .. 4 %: the default on one rank.
    // Rank shares gain as well (rank 0 of 2 / 4 / 8 on the 10 M-vertex graph: local SpMV - 3.8 / - 1.6 / - 3.0 %, profiles/r4_wide_band.txt);
    // the exchange layout rounds chunk 0 to this width (lzx_graph.hip), and a band that straddles the chunk boundary -- 8 Ki bands
    // chosen below for a small x -- simply waits for the second chunk.
    c->pb_cb = (c->pb_cb_opt == 8192 || c->pb_cb_opt == 16384 || c->pb_cb_opt == LZX_PB_CB_WIDE) ? (u32)c->pb_cb_opt
               : (c->xlen * sizeof(double) <= (16u << 20) && (c->overlap || c->fuse_opt == 0) ? 8192u : LZX_PB_CB_WIDE);
    const u32 nb = (u32)((c->xlen + c->pb_cb - 1) / c->pb_cb);
    if (nb >= (1u << 16)) LZX_FAIL(LZX_ERR_LIMIT, "propagation blocking: %u column bands (limit 65535)", nb);
    // values per gather item / entries per plain band: every wavefront slot of the gather pass (2 workgroups of 8 per
    // CU) should get a few items, and an item should not be shorter than its fold is worth
    u32 target = LZX_PB_TARGET;
    {
        // a workgroup (8 wavefronts) per item, two workgroups per CU, a few items each
        const u64 want = total / ((u64)c->cu_count * 8);
        // ... nor so short that a band of a small problem is cut into several items (each cut band costs its rows a trip through
        // the per-item totals and the SpMV a k_pb_finish launch): from 32 Ki values per item (round 5, tools/target_sweep.py,
        // profiles/r5_target_sweep.txt: 1 M-vertex graph 0.0626 -> 0.0594 ms per SpMV, rank 0 of 8 of the 10 M-vertex graph
        // 0.111 -> 0.102 ms, rank 0 of 4 and of 2 unchanged; the floor was 8 Ki)
        target = (u32)std::min<u64>(8 * LZX_PB_TARGET, std::max<u64>(LZX_PB_TARGET, (want + 1023) & ~1023ull));
    }
    if (c->pb_target_opt > 0) target = (u32)c->pb_target_opt;
    // entries per scatter unit: every unit restages its column band while its CU does nothing else, and a workgroup
    // only leaves its CU to the next one when its last wavefront is done, so big units are cheaper; small ones even out
    // the tail.  About four units per CU, between 64 Ki and 128 Ki entries (measured, tools/perf_probe.py @unit and
    // tools/rank_probe.py: C3 on one GPU 128 Ki 0.317 vs 64 Ki 0.344 vs 256 Ki 0.326 ms; its 1/8 share: 32 Ki 0.1055,
    // 64 Ki 0.109, 128 Ki 0.110 ms -- one round of big units leaves the launch as long as its slowest unit);
    // down to 8 Ki only to give every CU about two units (of tapered size) on graphs whose x sits in the L2s anyway
    u32 unit_cap = (u32)std::min<u64>(131072, std::max<u64>(32768, (total / ((u64)c->cu_count * 4) + 511) & ~511ull));
    if (c->xlen * sizeof(double) <= (16u << 20))
        unit_cap = (u32)std::min<u64>(65536, std::max<u64>(8192, (total / ((u64)c->cu_count * 2) + 511) & ~511ull));   // (1 M-vertex graph: 32 Ki 0.0650, 16 Ki 0.0680, 48 Ki 0.0675, 64 Ki 0.0735 ms per SpMV)
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

    // 4. value positions: every run gets its values (pieces or entries) padded to whole quads (32 bytes), in gather order
    const u32 run_align = (c->pb_align_opt == 4 || c->pb_align_opt == 8 || c->pb_align_opt == 16) ? (u32)c->pb_align_opt : LZX_PB_ALIGN;
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
        hipLaunchKernelGGL(k_pb_bounds_u64, GRID(nr + 1), d_sorted, total, KEY_RBAND_SHIFT, nr, d_rstart);
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
        u32 no_rounds = 0;
#ifdef LZX_DEBUG_KNOBS
        if (getenv("LZX_PB_NO_ROUNDS")) no_rounds = 1;   // A/B of what the rounds cost
#endif
        hipLaunchKernelGGL(k_pb_slots, GRID(len), d_prow, d_occ, d_rstart_pad, c->d_pb_rep, nr, len, c->d_pb_lrow, no_rounds);
        LZX_HIP(hipStreamSynchronize(st));
#ifdef LZX_DEBUG_KNOBS
        if (getenv("LZX_PB_STATS")) {   // how often does the gather pass need a later round? (values whose slot an earlier lane of the same instruction holds)
            std::vector<uint16_t> hs(len);
            LZX_HIP(hipMemcpy(hs.data(), c->d_pb_lrow, sizeof(uint16_t) * len, hipMemcpyDeviceToHost));
            u64 late = 0, blocks_late = 0, maxr = 0;
            for (u64 b0 = 0; b0 < len; b0 += 128) {
                bool any = false;
                for (u64 i = b0; i < std::min<u64>(len, b0 + 128); ++i) {
                    const u32 r = hs[i] >> 10;
                    if (r) { ++late; any = true; maxr = std::max<u64>(maxr, r); }
                }
                blocks_late += any ? 1 : 0;
            }
            {   // the first few late values with their neighbourhood: position, slot words around it
                std::vector<uint16_t> hp(len);
                LZX_HIP(hipMemcpy(hp.data(), d_prow, sizeof(uint16_t) * len, hipMemcpyDeviceToHost));
                int shown = 0;
                for (u64 i = 0; i < len && shown < 6; ++i)
                    if (hs[i] >> 10) {
                        u32 R = 0;
                        while (R + 1 < nr && rstart[R + 1] <= i) ++R;
                        fprintf(stderr, "[lzx pb stats]   late value at %llu (band %u begins at %u, offset in band %llu, block offset %llu, rep %u): rows",
                                (unsigned long long)i, R, rstart[R], (unsigned long long)(i - rstart[R]), (unsigned long long)((i - rstart[R]) % 128), rep[R]);
                        for (u64 j = (i >= 6 ? i - 6 : 0); j < std::min<u64>(len, i + 3); ++j) fprintf(stderr, " %s%u/%u", j == i ? "*" : "", (unsigned)hp[j], (unsigned)(hs[j] >> 10));
                        fprintf(stderr, "\n");
                        ++shown;
                    }
            }
            u64 rep1 = 0;
            for (u32 R = 0; R < nr; ++R) rep1 += rep[R] == 1 ? 1 : 0;
            fprintf(stderr, "[lzx pb stats] gather slots: %llu values, %llu of them (%.3f %%) in a round > 0 (highest round %llu), %llu of %llu 128-value blocks (%.2f %%) "
                    "take the slow path; %llu of %u row bands have one slot per row\n", (unsigned long long)len, (unsigned long long)late, 100.0 * late / std::max<double>(1.0, (double)len),
                    (unsigned long long)maxr, (unsigned long long)blocks_late, (unsigned long long)((len + 127) / 128), 100.0 * blocks_late / std::max<double>(1.0, (double)((len + 127) / 128)),
                    (unsigned long long)rep1, nr);
        }
#endif
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
    c->pb_n_items = c->pb_n_static = (u32)(items.size() / 4);
    c->pb_n_dyn = 0;
    c->pb_n_multi = (u32)(multi.size() / 4);
    // one item per workgroup at a time, two workgroups per CU
    // 16 wavefronts per CU (their private y tiles fill the LDS): two workgroups of eight, or four of four
    c->pb_gather_block = c->pb_gwaves_opt == 4 ? 256u : 512u;
    c->pb_gather_grid = std::min<u32>((u32)c->cu_count * (1024u / c->pb_gather_block), std::max(1u, c->pb_n_items));
    if (c->pb_grid_cap_opt > 0) c->pb_gather_grid = std::min<u32>(c->pb_gather_grid, (u32)c->pb_grid_cap_opt);   // test shape: few workgroups, many rounds
    LZX_TRY(pb_alloc(&c->d_pb_beg, (u64)nr + 1));
    LZX_HIP(hipMemcpyAsync(c->d_pb_beg, rstart.data(), sizeof(u32) * ((size_t)nr + 1), hipMemcpyHostToDevice, st));
    const u32 group_cap = c->pb_group_opt >= 0 ? (u32)c->pb_group_opt : LZX_PB_GROUP;
    // The gather pass walks static longest-first lists (k_pb_gather).  (A ticketed form over fat records was built and measured
    // in round 3 -- it balances perfectly and still loses, C3 0.644 vs 0.630 ms per SpMV: profiles/r3_gather_ab.txt -- and was
    // removed with the other experiments in round 4.)
    if (group_cap > 0 && c->pb_gather_block == 512u) {
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
            // a group lasts as long as its widest band's wavefront: eight serial batches per Ki values (a group of few bands
            // gives every band several wavefronts: k_pb_gather)
            const u64 nbands = j - i, split = nbands <= 1 ? 8 : nbands <= 2 ? 4 : nbands <= 4 ? 2 : 1;
            cost.push_back(std::max<u64>(10ull * vals + 24ull * rows, 80ull * widest / split) + 40000ull);
            i = j;
        }
        const u32 G = c->pb_gather_grid;
        const size_t no = cost.size();
        std::vector<u32> order(no);
        for (size_t i = 0; i < no; ++i) order[i] = (u32)i;
        std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b2) { return cost[a] > cost[b2]; });
        // the dynamic tail (k_pb_gather): the cheapest items, `share` per cent of the pass's estimated cost, are drawn from a
        // counter at run time instead of being dealt here; only when every workgroup has more than one item to its name
        size_t n_tail = 0;
        {
            const u64 share = c->pb_dyn_opt >= 0 ? (u64)std::min<int64_t>(90, c->pb_dyn_opt) : LZX_PB_DYN_SHARE;
            u64 all = 0, tail = 0;
            for (u64 x : cost) all += x;
            if (share > 0 && no > (c->pb_dyn_opt > 0 ? 1 : 2) * (size_t)G)
                while (n_tail + G < no && (tail + cost[order[no - 1 - n_tail]]) * 100 <= all * share) tail += cost[order[no - 1 - n_tail++]];
        }
        const size_t n_dealt = no - n_tail;
        std::vector<std::vector<u32>> lists(G);
        std::priority_queue<std::pair<u64, u32>, std::vector<std::pair<u64, u32>>, std::greater<std::pair<u64, u32>>> heap;
        for (u32 w = 0; w < G; ++w) heap.push({0ull, w});
        size_t rounds = 0;
        for (size_t oi = 0; oi < n_dealt; ++oi) {
            const u32 i = order[oi];
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
        c->pb_n_static = (u32)(items.size() / 4);
        for (size_t oi = n_dealt; oi < no; ++oi) {   // the tail behind the static rounds, dearest first
            const u32 *it = &out[4 * (size_t)order[oi]];
            items.insert(items.end(), it, it + 4);
        }
        c->pb_n_dyn = (u32)n_tail;
        c->pb_n_items = (u32)(items.size() / 4);
    }
#ifdef LZX_DEBUG_KNOBS
    if (c->pb_stamps_opt > 0) {   // per-workgroup section stamps of the product gather pass
        LZX_TRY(pb_alloc(&c->d_pb_gstamps, 8 * (u64)c->pb_gather_grid));
        LZX_HIP(hipMemsetAsync(c->d_pb_gstamps, 0, sizeof(unsigned long long) * 8 * c->pb_gather_grid, st));
    }
#endif
    u32 n_uncovered = 0;
    {   // split rows (the first n_long64 local rows): who closes them?  0 = the gather pass's fold (the row's band is one gather
        // item), 1 = the row's multi thread of k_pb_finish (its band was cut into several items), 2 = k_pb_finish's split-row
        // thread (no gather item covers the row: it has no blocked entry)
        std::vector<uint8_t> flag((size_t)c->n_long64 + 1, 2);
        for (u32 R = 0; R < nr; ++R)
            if (rstart[R] != rstart[R + 1])
                for (u32 r = row0[R]; r < std::min<u32>(row0[R + 1], c->n_long64); ++r) flag[r] = 0;
        for (size_t i = 0; i < multi.size(); i += 4)
            if (multi[i] < c->n_long64) flag[multi[i]] = 1;
        for (u32 r = 0; r < c->n_long64; ++r) n_uncovered += flag[r] == 2 ? 1 : 0;
        LZX_TRY(pb_alloc(&c->d_pb_long_multi, flag.size()));
        LZX_HIP(hipMemcpyAsync(c->d_pb_long_multi, flag.data(), flag.size(), hipMemcpyHostToDevice, st));
        LZX_HIP(hipStreamSynchronize(st));
    }
    LZX_TRY(pb_alloc(&c->d_pb_items, items.size())); LZX_TRY(pb_alloc(&c->d_pb_multi, multi.size())); LZX_TRY(pb_alloc(&c->d_pb_part, slots));
    {   // the multi rows by ROW (k_lazy_update's look-up when it stands in for k_pb_finish): {row, first slot, items, slot stride}
        u32 limit = 0;
        for (size_t i = 0; i < multi.size(); i += 4) limit = std::max(limit, multi[i] + 1);
        limit = (limit + 1u) & ~1u;   // (k_lazy_update takes rows in pairs)
        std::vector<u32> mrow((size_t)limit * 4, 0u);
        for (size_t i = 0; i < multi.size(); i += 4)
            for (int u = 0; u < 4; ++u) mrow[(size_t)multi[i] * 4 + u] = multi[i + u];
        c->pb_multi_limit = limit;
        LZX_TRY(pb_alloc(&c->d_pb_mrow, (u64)std::max<u32>(limit, 1u)));
        if (limit) LZX_HIP(hipMemcpyAsync(c->d_pb_mrow, mrow.data(), sizeof(u32) * mrow.size(), hipMemcpyHostToDevice, st));
        LZX_HIP(hipStreamSynchronize(st));
    }
    if (!items.empty())
        LZX_HIP(hipMemcpyAsync(c->d_pb_items, items.data(), sizeof(u32) * items.size(), hipMemcpyHostToDevice, st));
    if (!multi.empty())
        LZX_HIP(hipMemcpyAsync(c->d_pb_multi, multi.data(), sizeof(u32) * multi.size(), hipMemcpyHostToDevice, st));

    LZX_TRY(pb_alloc(&c->d_pb_gcounter, 4));
    LZX_HIP(hipMemsetAsync(c->d_pb_gcounter, 0, sizeof(u32) * 4, st));
    if (c->pb_n_dyn) {
        LZX_TRY(pb_alloc(&c->d_pb_item_dot, c->pb_n_dyn));
        LZX_HIP(hipMemsetAsync(c->d_pb_item_dot, 0, sizeof(double) * c->pb_n_dyn, st));
    }
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
    // k_pb_finish: the rows of multi-item bands, then (only when some split row is covered by no gather item) a thread per split row
    c->pb_finish_grid = (c->pb_n_multi + (n_uncovered ? c->n_long64 : 0u) + LZX_VEC_BLOCK - 1) / LZX_VEC_BLOCK;
    // ... and the lazy loop may leave that launch out (k_lazy_update adds the totals where it reads v) when multi-item bands are all
    // it serves: no uncovered split row, no drawn items whose alpha shares it closes
    c->pb_defer_ok = c->pb_n_multi > 0 && n_uncovered == 0 && c->pb_n_dyn == 0;
    if (c->pb_n_dyn) c->pb_finish_grid = std::max<u32>(c->pb_finish_grid, (c->pb_n_dyn + LZX_VEC_BLOCK - 1) / LZX_VEC_BLOCK);   // + the drawn items' alpha partials
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
    if (!c->pb || !(c->pb_cb == LZX_PB_CB || c->pb_cb == 8192 || c->pb_cb == LZX_PB_CB_WIDE)) return false;
#ifdef LZX_DEBUG_KNOBS
    if (c->phase_mask_opt & (4 | 8)) return false;
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
    // a row that spans lanes of a reduced step: summed by a cross-lane scan in registers (test shape pb_carry_scan = 1) or
    // through LDS carry slots (0)
    const bool scan = c->pb_scan_opt >= 0 ? c->pb_scan_opt > 0 : LZX_PB_CARRY_SCAN;
    const size_t lds1 = ((size_t)c->pb_cb + 2 + (scan ? 0 : 16 * 66)) * sizeof(double);
    // the pass's tables as non-temporal loads (test shape pb_scatter_nt).  Default: when the value stream is larger than the caches
    // and most entries sit in reduced runs -- there the tables are the larger stream and streaming them past the Infinity Cache
    // leaves more of the freshly written values for the gather (10 M-vertex R-MAT -1.2 %, 4 M -4.3 %); where the values are the
    // larger stream (Erdos-Renyi: almost no reduced runs) it measured +1.3 % (profiles/r4_knob_sweeps.txt)
    const bool ntc = c->pb_scatter_nt_opt >= 0 ? c->pb_scatter_nt_opt > 0
                                               : (LZX_PB_SCATTER_NT && 10ull * c->pb_values > LZX_PB_NT_BYTES && 2 * c->pbr_entries > c->pb_entries);
    using scatter_fn = void (*)(const u32 *, const uint4 *, const u32 *, const uint2 *, const u32 *, const double *, u64, double *);
    auto pick_scatter = [&]() -> scatter_fn {
        if (c->pb_cb == 8192) {
            if (scan) return ntc ? k_pb_scatter<8192, true, true> : k_pb_scatter<8192, true, false>;
            return ntc ? k_pb_scatter<8192, false, true> : k_pb_scatter<8192, false, false>;
        }
        if (c->pb_cb == LZX_PB_CB_WIDE) return ntc ? k_pb_scatter<LZX_PB_CB_WIDE, true, true> : k_pb_scatter<LZX_PB_CB_WIDE, true, false>;
        if (scan) return ntc ? k_pb_scatter<LZX_PB_CB, true, true> : k_pb_scatter<LZX_PB_CB, true, false>;
        return ntc ? k_pb_scatter<LZX_PB_CB, false, true> : k_pb_scatter<LZX_PB_CB, false, false>;
    };
    scatter_fn kern = pick_scatter();
    if (c->pb_units)
        LZX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    // one workgroup per unit / static item lists (k_pb_scatter, k_pb_gather); persistent and ticketed forms were measured in
    // rounds 2 and 3 (none was faster: profiles/NOTES.md) and removed in round 4
    auto scatter = [&](u32 u0, u32 u1, bool may_fuse) {
        if (u1 <= u0 && !(may_fuse && fuse && fused)) return;
        if (may_fuse && fuse && fused && (c->pb_cb == LZX_PB_CB || c->pb_cb == 8192 || c->pb_cb == LZX_PB_CB_WIDE) && u0 == 0) {
            // the staged-columns workgroups share the launch of the scatter units (k_pb_scatter_spmv)
            using fused_fn = void (*)(const u32 *, u32, const uint4 *, const u32 *, const uint2 *, const u32 *, const double *, u64, double *, const SpmvArgs, const u32);
            fused_fn kf;
            if (c->pb_cb == 8192) {
                if (scan) kf = ntc ? k_pb_scatter_spmv<8192, true, true> : k_pb_scatter_spmv<8192, true, false>;
                else kf = ntc ? k_pb_scatter_spmv<8192, false, true> : k_pb_scatter_spmv<8192, false, false>;
            } else if (c->pb_cb == LZX_PB_CB_WIDE) {
                kf = ntc ? k_pb_scatter_spmv<LZX_PB_CB_WIDE, true, true> : k_pb_scatter_spmv<LZX_PB_CB_WIDE, true, false>;
            } else {
                if (scan) kf = ntc ? k_pb_scatter_spmv<LZX_PB_CB, true, true> : k_pb_scatter_spmv<LZX_PB_CB, true, false>;
                else kf = ntc ? k_pb_scatter_spmv<LZX_PB_CB, false, true> : k_pb_scatter_spmv<LZX_PB_CB, false, false>;
            }
            const size_t ldsf = std::max(lds1, c->spmv_lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf);
            hipLaunchKernelGGL(kf, dim3(u1 + fuse_blocks), dim3(1024), ldsf, c->stream, c->d_pb_unit, u1, c->d_pbr_code, c->d_pbr_base,
                               reinterpret_cast<const uint2 *>(c->d_pb_lcol), c->d_pb_dst, x, c->xlen, c->d_pb_val, *fuse, c->fuse_opt == 1 ? 0u : 1u);
            *fused = true;
            return;
        }
        hipLaunchKernelGGL(kern, dim3(u1 - u0), dim3(1024), lds1, c->stream, c->d_pb_unit + 5 * (size_t)u0, c->d_pbr_code,
                           c->d_pbr_base, reinterpret_cast<const uint2 *>(c->d_pb_lcol), c->d_pb_dst, x, c->xlen, c->d_pb_val);
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
    const size_t lds2 = ((size_t)(LZX_PB_GATHER_BLOCK / 64) * (LZX_PB_RB + 64) + LZX_PB_GATHER_BLOCK / 64) * sizeof(double) +
                        16 * (LZX_PB_GATHER_BLOCK / 64) * 8 * sizeof(u32) + 16;   // tiles, wavefront sums, the preloaded item records, the ticket
    // stream loads of the pass as non-temporal loads when the value stream is larger than the caches can hold anyway (k_pb_gather)
    const bool nt = c->pb_gather_nt_opt >= 0 ? c->pb_gather_nt_opt > 0 : 10ull * c->pb_values > LZX_PB_NT_BYTES;
    u32 probe_l2g = 0;
#ifdef LZX_DEBUG_KNOBS
    if (const char *pe = getenv("LZX_PROBE_L2G")) probe_l2g = (u32)atoi(pe);   // probe: extra L2-resident gathers per small band (k_pb_gather)
#endif
    auto gather = [&](auto kern, unsigned long long *stamps) -> int {
        LZX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        hipLaunchKernelGGL(kern, dim3(c->pb_gather_grid), dim3(LZX_PB_GATHER_BLOCK), lds2, c->stream,
                           reinterpret_cast<const uint4 *>(c->d_pb_items), c->pb_n_static, c->pb_n_dyn, c->d_pb_gcounter, c->d_pb_item_dot, c->d_pb_row0,
                           c->d_pb_rep, c->d_pb_beg, c->d_pb_lrow, c->d_pb_val, v, q_loc, c->d_pb_part, partials, stamps, probe_l2g,
                           c->d_item_first, c->d_long_partial, c->d_pb_long_multi, c->n_long64);
        return LZX_OK;
    };
    bool gathered = false;
#ifdef LZX_DEBUG_KNOBS
    if (!(c->phase_mask_opt & 8)) {
        if (getenv("LZX_ABLATE_SLOTS")) {   // ablation: the pass without its slot stream (wrong sums; timing only)
            LZX_TRY(nt ? gather(k_pb_gather<false, true, true>, nullptr) : gather(k_pb_gather<false, false, true>, nullptr));
            gathered = true;
        } else
        if (c->pb_stamps_opt > 0 && c->d_pb_gstamps) {   // the product kernel with its section stamps
            LZX_TRY(nt ? gather(k_pb_gather<true, true>, c->d_pb_gstamps) : gather(k_pb_gather<true, false>, c->d_pb_gstamps));
            gathered = true;
        }
    }
#endif
    if (c->phase_mask_opt & 8) {
        // experiment: scatter pass alone
    } else if (!gathered)
        LZX_TRY(nt ? gather(k_pb_gather<false, true>, nullptr) : gather(k_pb_gather<false, false>, nullptr));
    if (c->pb_finish_grid && !(c->pb_deferring && c->pb_defer_ok))   // (deferred: k_lazy_update completes v, lzx_ctx::pb_deferring)
        hipLaunchKernelGGL(k_pb_finish, dim3(c->pb_finish_grid), dim3(LZX_VEC_BLOCK), 0, c->stream,
                           reinterpret_cast<const uint4 *>(c->d_pb_multi), c->pb_n_multi, c->d_pb_part, c->d_item_first,
                           c->d_long_partial, c->d_pb_long_multi, c->n_long64, v, q_loc, partials + c->pb_gather_grid,
                           c->pb_n_dyn ? c->d_pb_item_dot : nullptr, c->pb_n_dyn);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

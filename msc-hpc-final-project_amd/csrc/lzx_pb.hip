// lzx_pb.hip -- propagation-blocked SpMV for the entries whose column is NOT staged in LDS by k_spmv.
//
// Why: past the 4 MiB per-XCD L2 a random 8-byte gather of x costs a whole 128-byte fabric transaction
// (tools/gather_bench.hip: 55 Ggather/s = 7 TB/s of traffic), and even L2-resident gathers top out near
// 200 Ggather/s, so the plain CSR gather moves 8x the algorithmic bytes (profiles/r1_c3_pmc.json: 15 GB per SpMV
// on the 10 M-vertex graph).  Here the same work is two streaming passes around LDS:
//   scatter (k_pb_scatter): entries ordered by COLUMN band; the band's 16 Ki x values are staged in LDS; a lane
//       takes a QUAD of four entries (one 8-byte load of their columns-in-band, one 4-byte load of the quad's
//       slot), looks the four values up in LDS and writes them with two 16-byte stores.  Slots are ordered by ROW
//       band, so a (row band, column band) run is one contiguous stretch of writes; runs are padded to whole quads
//       (padding reads a zero kept behind the staged band).
//   gather (k_pb_gather): row bands hold about LZX_PB_TARGET entries each (1 .. 1024 consecutive rows, so heavy
//       rows get bands of their own and every wavefront gets the same amount of work); one wavefront streams a
//       band's values + 2-byte LDS slot and adds each value into a wave-private LDS y tile with ds_add_f64.  The
//       slot of an entry is row * rep + replica, the replica chosen when the graph is reshaped so that the 64
//       lanes of one step (almost) never share a slot: no shuffles, no serialised conflicts; only this wavefront
//       touches the tile, so additions happen in program order.  The tile is then folded (rep replicas per row),
//       added to v, and the wave forms its share of alpha = v . q.  A single row with more than 2 targets of
//       entries is cut into items whose totals k_pb_finish adds in order.
// All tables are static (built once per graph by lzx_pb_prepare: two radix sorts and a few scans).
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "lzx_internal.h"

namespace {

// 64-bit sort key: row band (24 bits) | column band (16) | row in band (10) | column in band (14)
__device__ __forceinline__ u64 pack_key(u32 rband, u32 cband, u32 lrow, u32 lcol)
{
    return ((u64)rband << 40) | ((u64)cband << 24) | ((u64)lrow << 14) | (u64)lcol;
}

// one wavefront (64-thread block) per local row: keep the entries whose code is not a hub slot
__global__ void __launch_bounds__(64)
k_pb_emit(const u64 *row_ptr, const u32 *col_idx, const u32 *code_of_old, const u32 *old_of_local,
          const u32 *deg_local, const u32 *nh_off, u32 hub, const u32 *band_row0, u32 nr, u64 *keys)
{
    const u32 l = blockIdx.x, lane = threadIdx.x;
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
            keys[out + pre] = pack_key(R, p / LZX_PB_CB, lrow, p % LZX_PB_CB);
        }
        out += __popcll(m);
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

// padded run length: a multiple of `align` entries (4 = whole quads; 16 = whole 128-byte lines of values)
__global__ void k_pb_padlen(const u32 *runstart, u32 nruns, u64 count, u32 align, u32 *padlen)
{
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nruns) return;
    const u32 end = (r + 1 < nruns) ? runstart[r + 1] : (u32)count;
    padlen[r] = (end - runstart[r] + align - 1u) & ~(align - 1u);
}

// place every entry at its padded position: row / column within band, and per quad its column band
__global__ void k_pb_place(const u64 *keys, const u32 *runid_incl, const u32 *runstart, const u32 *pstart, u64 count,
                           uint16_t *prow, uint16_t *plcol, uint16_t *quad_cband, u32 *pos_of_entry)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const u32 r = runid_incl[i] - 1;
    const u32 pos = pstart[r] + ((u32)i - runstart[r]);
    const u64 k = keys[i];
    prow[pos] = (uint16_t)((k >> 14) & 0x3ffu);
    plcol[pos] = (uint16_t)(k & 0x3fffu);
    if ((pos & 3u) == 0) quad_cband[pos >> 2] = (uint16_t)((k >> 24) & 0xffffu);
    pos_of_entry[i] = pos;
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

// ... and the same boundaries in padded positions
__global__ void k_pb_rstart_pad(const u32 *rstart, const u32 *pos_of_entry, u32 nr, u64 count, u32 total_pad, u32 *rstart_pad)
{
    const u32 R = blockIdx.x * blockDim.x + threadIdx.x;
    if (R > nr) return;
    rstart_pad[R] = rstart[R] < count ? pos_of_entry[rstart[R]] : total_pad;
}

// Gather order (padded), step = 64 consecutive positions counted from the start of the row band.  occ[p] = how many
// earlier lanes of the same step carry the same row; band_rep[R] = max over the band of occ + 1.
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
    const u32 i = rstart_pad[R] + (step - band_step0[R]) * 64 + lane;
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

// ---- the per-iteration kernels --------------------------------------------------------------------------------
// unit = {column band, first quad, last quad} in scatter order.  Each wavefront walks its own contiguous share of the
// unit 64 quads at a time (lane = consecutive quad): contiguous loads, and 32-byte-per-lane stores that are
// contiguous inside a run.  4 quads per lane in flight.
__global__ void __launch_bounds__(1024)
k_pb_scatter(const u32 *unit, const uint2 *q_lcol, const u32 *q_dst, const double *__restrict__ x, u64 xlen, double *val)
{
    extern __shared__ __attribute__((aligned(16))) double tile[];   // LZX_PB_CB staged values + a zero for padding
    const u32 band = unit[3 * blockIdx.x], beg = unit[3 * blockIdx.x + 1], end = unit[3 * blockIdx.x + 2];
    const u64 base = (u64)band * LZX_PB_CB;
    for (u32 j = threadIdx.x; j < LZX_PB_CB + 2; j += 1024) {
        const u64 p = base + j;
        tile[j] = (j < LZX_PB_CB && p < xlen) ? x[p] : 0.0;
    }
    __syncthreads();
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u32 span = (((end - beg) + 15u) / 16u + 63u) & ~63u;   // quads per wavefront, multiple of 64
    const u32 wbeg = beg + wv * span;
    const u32 wend = wbeg + span < end ? wbeg + span : end;
    u32 j = wbeg + lane;
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
            double2 *out = reinterpret_cast<double2 *>(val + d[u]);   // 32-byte aligned: slots of a quad
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

__device__ __forceinline__ double wave_sum_pb(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// item table entry: {row band, begin, end, slot} in padded gather positions; slot == 0xffffffff: the item is its
// band's only one and adds straight into v; otherwise it is one of several items of a single-row band and leaves
// its total in part[slot].
__global__ void __launch_bounds__(LZX_PB_GATHER_BLOCK)
k_pb_gather(const uint4 *items, u32 n_items, const u32 *band_row0, const u32 *band_rep, const uint16_t *lslot,
            const double *val, double *v, const double *__restrict__ q_loc, double *part, double *partials)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr u32 WAVES = LZX_PB_GATHER_BLOCK / 64;
    constexpr u32 TILE = LZX_PB_RB + 8;                  // + spare slot for padding entries
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double *ytile = lds + (size_t)wv * TILE;             // private to this wavefront
    double *wsum = lds + (size_t)WAVES * TILE;
    const u32 waves = gridDim.x * WAVES;
    double dot = 0.0;
    for (u32 it = blockIdx.x * WAVES + wv; it < n_items; it += waves) {
        const uint4 item = items[it];
        const u32 R = item.x, beg = item.y, end = item.z;
        const u32 row0 = band_row0[R], rows = band_row0[R + 1] - row0;
        if (rows == 1) {
            // one heavy row: plain strided sum (padding holds zeros), fixed butterfly
            double acc = 0.0;
            u32 i = beg + lane;
            for (; i + 7 * 64 < end; i += 8 * 64) {
                double a[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] = val[i + u * 64];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += a[u];
            }
            for (; i < end; i += 64) acc += val[i];
            acc = wave_sum_pb(acc);
            if (lane == 0) {
                if (item.w == 0xffffffffu) {
                    v[row0] += acc;
                    dot += acc * q_loc[row0];
                } else {
                    part[item.w] = acc;
                }
            }
            continue;
        }
        const u32 rep = band_rep[R];
        const u32 slots = rows * rep;
        for (u32 j = lane; j < slots; j += 64) ytile[j] = 0.0;
        __builtin_amdgcn_wave_barrier();
        u32 i = beg + lane;
        for (; i + 7 * 64 < end; i += 8 * 64) {
            double av[8];
            u32 sv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                av[u] = val[i + u * 64];
                sv[u] = lslot[i + u * 64];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) atomicAdd(&ytile[sv[u]], av[u]);
        }
        for (; i < end; i += 64) atomicAdd(&ytile[lslot[i]], val[i]);
        __builtin_amdgcn_wave_barrier();
        for (u32 j = lane; j < rows; j += 64) {
            double y = 0.0;
            for (u32 t = 0; t < rep; ++t) y += ytile[j * rep + t];
            v[row0 + j] += y;
            dot += y * q_loc[row0 + j];
        }
        __builtin_amdgcn_wave_barrier();
    }
    dot = wave_sum_pb(dot);
    if (lane == 0) wsum[wv] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (u32 i = 0; i < WAVES; ++i) s += wsum[i];
        partials[blockIdx.x] = s;
    }
}

// rows cut into several items: v[row] += item totals in item order; alpha partials for those rows
__global__ void __launch_bounds__(LZX_VEC_BLOCK)
k_pb_finish(const u32 *multi /*[n][3]: row, first slot, slots*/, u32 n_multi, const double *part, double *v,
            const double *q_loc, double *partials)
{
    __shared__ double sh[4];
    const u32 t = blockIdx.x * LZX_VEC_BLOCK + threadIdx.x;
    double dot = 0.0;
    if (t < n_multi) {
        const u32 row = multi[3 * t], first = multi[3 * t + 1], cnt = multi[3 * t + 2];
        double s = 0.0;
        for (u32 k = 0; k < cnt; ++k) s += part[first + k];
        v[row] += s;
        dot = s * q_loc[row];
    }
    dot = wave_sum_pb(dot);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

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
    pb_free(c->d_pb_items);
    pb_free(c->d_pb_multi);
    pb_free(c->d_pb_part);
    c->pb = false;
    c->pb_entries = 0;
    c->pb_units = c->pb_nr = c->pb_gather_grid = c->pb_n_items = c->pb_n_multi = c->pb_finish_grid = 0;
}

u32 lzx_pb_partials(const lzx_ctx *c) { return c->pb ? c->pb_gather_grid + c->pb_finish_grid : 0; }

int lzx_pb_prepare(lzx_ctx *c, const u32 *d_code, const u32 *d_old_of_local, const u32 *d_deg_local,
                   const u32 *d_nh_off, const std::vector<u32> &h_nh, u64 total)
{
    hipStream_t st = c->stream;
    if (total >= (1ull << 31)) LZX_FAIL(LZX_ERR_LIMIT, "propagation blocking: %llu entries do not fit 32-bit slots", (unsigned long long)total);
    const u32 nb = (u32)((c->xlen + LZX_PB_CB - 1) / LZX_PB_CB);
    if (nb >= (1u << 16)) LZX_FAIL(LZX_ERR_LIMIT, "propagation blocking: %u column bands (limit 65535)", nb);

    // ---- row bands: consecutive local rows, closed at ~target entries or LZX_PB_RB rows; a row heavier than the
    //      target is a band of its own.  Every wavefront of the gather pass gets one band (or one item of a very
    //      heavy row), so work per wavefront is even although degrees are not.
    const u32 target = c->pb_target_opt > 0 ? (u32)c->pb_target_opt : LZX_PB_TARGET;
    std::vector<u32> row0;
    row0.push_back(0);
    {
        u32 rows = 0;
        u64 cnt = 0;
        for (u32 l = 0; l < c->n_loc_real; ++l) {
            const u32 nh = h_nh[l];
            if (rows > 0 && (cnt + nh > target || rows == LZX_PB_RB)) {
                row0.push_back(l);
                rows = 0;
                cnt = 0;
            }
            ++rows;
            cnt += nh;
        }
        row0.push_back(c->n_loc_real);
    }
    const u32 nr = (u32)row0.size() - 1;
    if (nr >= (1u << 24)) LZX_FAIL(LZX_ERR_LIMIT, "propagation blocking: %u row bands (limit 2^24)", nr);

    u64 *d_keys = nullptr, *d_sorted = nullptr;
    u32 *d_head = nullptr, *d_runid = nullptr, *d_runstart = nullptr, *d_padlen = nullptr, *d_pstart = nullptr;
    u32 *d_pos = nullptr, *d_rstart = nullptr, *d_rstart_pad = nullptr, *d_step0 = nullptr;
    u32 *d_qkey = nullptr, *d_qkey_s = nullptr, *d_qidx = nullptr, *d_qsorted = nullptr, *d_bstart = nullptr;
    uint16_t *d_prow = nullptr, *d_plcol = nullptr, *d_qcband = nullptr;
    uint8_t *d_occ = nullptr;
    void *d_tmp = nullptr;
    int rc = LZX_OK;
    auto free_tmp = [&]() { if (d_tmp) (void)hipFree(d_tmp); d_tmp = nullptr; };
    auto cleanup = [&]() {
        pb_free(d_keys); pb_free(d_sorted); pb_free(d_head); pb_free(d_runid); pb_free(d_runstart); pb_free(d_padlen);
        pb_free(d_pstart); pb_free(d_pos); pb_free(d_rstart); pb_free(d_rstart_pad); pb_free(d_step0); pb_free(d_qkey);
        pb_free(d_qkey_s); pb_free(d_qidx); pb_free(d_qsorted); pb_free(d_bstart); pb_free(d_prow); pb_free(d_plcol);
        pb_free(d_qcband); pb_free(d_occ);
        free_tmp();
    };
#define PB(call) do { rc = (call); if (rc != LZX_OK) { cleanup(); lzx_pb_release(c); return rc; } } while (0)
#define PB_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
        lzx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); cleanup(); lzx_pb_release(c); \
        return e_ == hipErrorOutOfMemory ? LZX_ERR_NOMEM : LZX_ERR_HIP; } } while (0)
#define GRID(n) dim3((u32)(((u64)(n) + 255) / 256)), dim3(256), 0, st

    PB(pb_alloc(&c->d_pb_row0, (u64)nr + 1));
    PB_HIP(hipMemcpyAsync(c->d_pb_row0, row0.data(), sizeof(u32) * ((size_t)nr + 1), hipMemcpyHostToDevice, st));

    // 1. emit + sort by (row band, column band, row, column): the (unpadded) GATHER order
    PB(pb_alloc(&d_keys, total)); PB(pb_alloc(&d_sorted, total));
    if (c->n_loc_real)
        hipLaunchKernelGGL(k_pb_emit, dim3(c->n_loc_real), dim3(64), 0, st, c->d_row_ptr, c->d_col_idx, d_code,
                           d_old_of_local, d_deg_local, d_nh_off, c->hub_real, c->d_pb_row0, nr, d_keys);
    size_t tb = 0;
    PB_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, d_keys, d_sorted, (u64)total, 0, 64, st));
    PB_HIP(hipMalloc(&d_tmp, tb ? tb : 16));
    PB_HIP(hipcub::DeviceRadixSort::SortKeys(d_tmp, tb, d_keys, d_sorted, (u64)total, 0, 64, st));
    PB_HIP(hipStreamSynchronize(st));
    free_tmp();
    pb_free(d_keys);

    // 2. runs = maximal stretches of one (row band, column band); padded to whole quads
    PB(pb_alloc(&d_head, total)); PB(pb_alloc(&d_runid, total));
    hipLaunchKernelGGL(k_pb_heads, GRID(total), d_sorted, total, d_head);
    tb = 0;
    PB_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tb, d_head, d_runid, (u64)total, st));
    PB_HIP(hipMalloc(&d_tmp, tb ? tb : 16));
    PB_HIP(hipcub::DeviceScan::InclusiveSum(d_tmp, tb, d_head, d_runid, (u64)total, st));
    u32 nruns = 0;
    PB_HIP(hipMemcpyAsync(&nruns, d_runid + (total - 1), sizeof(u32), hipMemcpyDeviceToHost, st));
    PB_HIP(hipStreamSynchronize(st));
    free_tmp();
    PB(pb_alloc(&d_runstart, (u64)nruns + 1)); PB(pb_alloc(&d_padlen, (u64)nruns + 1)); PB(pb_alloc(&d_pstart, (u64)nruns + 1));
    hipLaunchKernelGGL(k_pb_runstarts, GRID(total), d_head, d_runid, total, d_runstart);
    PB_HIP(hipMemsetAsync(d_padlen + nruns, 0, sizeof(u32), st));
    const u32 run_align = (c->pb_align_opt == 8 || c->pb_align_opt == 16 || c->pb_align_opt == 4) ? (u32)c->pb_align_opt : LZX_PB_ALIGN;
    hipLaunchKernelGGL(k_pb_padlen, GRID(nruns), d_runstart, nruns, total, run_align, d_padlen);
    tb = 0;
    PB_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, d_padlen, d_pstart, (u64)nruns + 1, st));
    PB_HIP(hipMalloc(&d_tmp, tb ? tb : 16));
    PB_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp, tb, d_padlen, d_pstart, (u64)nruns + 1, st));
    u32 total_pad = 0;
    PB_HIP(hipMemcpyAsync(&total_pad, d_pstart + nruns, sizeof(u32), hipMemcpyDeviceToHost, st));
    PB_HIP(hipStreamSynchronize(st));
    free_tmp();
    pb_free(d_head); pb_free(d_padlen);
    if (total_pad >= (1u << 31) || total_pad < total) { cleanup(); lzx_pb_release(c); LZX_FAIL(LZX_ERR_LIMIT, "propagation blocking: padded entry count overflows"); }
    const u64 nquads = total_pad / 4;

    // 3. padded positions: row / column in band per position (0xffff / column "CB" = the zero behind the staged
    //    band for padding), column band per quad
    PB(pb_alloc(&d_prow, (u64)total_pad + 8)); PB(pb_alloc(&d_plcol, (u64)total_pad + 8)); PB(pb_alloc(&d_qcband, nquads + 2));
    PB(pb_alloc(&d_pos, total));
    hipLaunchKernelGGL(k_pb_fill16, GRID(total_pad), d_prow, total_pad, (uint16_t)0xffffu);
    hipLaunchKernelGGL(k_pb_fill16, GRID(total_pad), d_plcol, total_pad, (uint16_t)LZX_PB_CB);
    hipLaunchKernelGGL(k_pb_fill16, GRID(nquads + 2), d_qcband, nquads + 2, (uint16_t)0xffffu);   // all-padding quads: no band
    hipLaunchKernelGGL(k_pb_place, GRID(total), d_sorted, d_runid, d_runstart, d_pstart, total, d_prow, d_plcol, d_qcband, d_pos);
    PB(pb_alloc(&d_rstart, (u64)nr + 1)); PB(pb_alloc(&d_rstart_pad, (u64)nr + 1));
    hipLaunchKernelGGL(k_pb_bounds_u64, GRID(nr + 1), d_sorted, total, 40u, nr, d_rstart);
    hipLaunchKernelGGL(k_pb_rstart_pad, GRID(nr + 1), d_rstart, d_pos, nr, total, total_pad, d_rstart_pad);
    std::vector<u32> rstart((size_t)nr + 1);
    PB_HIP(hipMemcpyAsync(rstart.data(), d_rstart_pad, sizeof(u32) * ((size_t)nr + 1), hipMemcpyDeviceToHost, st));
    PB_HIP(hipStreamSynchronize(st));
    pb_free(d_sorted); pb_free(d_runid); pb_free(d_runstart); pb_free(d_pstart); pb_free(d_pos); pb_free(d_rstart);

    // 4. conflict-free LDS slots for the gather pass
    PB(pb_alloc(&c->d_pb_lrow, (u64)total_pad + 8)); PB(pb_alloc(&c->d_pb_rep, (u64)nr));
    {
        std::vector<u32> step0((size_t)nr + 1), rep((size_t)nr, 1u);
        u64 steps = 0;
        for (u32 R = 0; R < nr; ++R) {
            step0[R] = (u32)steps;
            steps += (rstart[R + 1] - rstart[R] + 63u) / 64u;
        }
        step0[nr] = (u32)steps;
        PB(pb_alloc(&d_step0, (u64)nr + 1)); PB(pb_alloc(&d_occ, (u64)total_pad + 8));
        PB_HIP(hipMemcpyAsync(d_step0, step0.data(), sizeof(u32) * ((size_t)nr + 1), hipMemcpyHostToDevice, st));
        PB_HIP(hipMemcpyAsync(c->d_pb_rep, rep.data(), sizeof(u32) * nr, hipMemcpyHostToDevice, st));
        PB_HIP(hipMemsetAsync(d_occ, 0, (u64)total_pad + 8, st));
        if (steps)
            hipLaunchKernelGGL(k_pb_occurrence, dim3((u32)steps), dim3(64), 0, st, d_prow, d_rstart_pad, d_step0, nr, d_occ,
                               c->d_pb_rep);
        PB_HIP(hipMemcpyAsync(rep.data(), c->d_pb_rep, sizeof(u32) * nr, hipMemcpyDeviceToHost, st));
        PB_HIP(hipStreamSynchronize(st));
        // replicas per row: enough for the worst step of the band, but the tile holds LZX_PB_RB slots
        for (u32 R = 0; R < nr; ++R) {
            const u32 rows = row0[R + 1] - row0[R];
            const u32 room = std::max(1u, LZX_PB_RB / std::max(rows, 1u));
            rep[R] = std::max(1u, std::min(std::min(rep[R], room), 64u));
        }
        PB_HIP(hipMemcpyAsync(c->d_pb_rep, rep.data(), sizeof(u32) * nr, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_pb_slots, GRID(total_pad), d_prow, d_occ, d_rstart_pad, c->d_pb_rep, nr, total_pad, c->d_pb_lrow);
        PB_HIP(hipStreamSynchronize(st));
        pb_free(d_step0); pb_free(d_occ); pb_free(d_prow); pb_free(d_rstart_pad);
    }

    // 5. scatter order: quads sorted (stably) by column band
    PB(pb_alloc(&d_qkey, nquads)); PB(pb_alloc(&d_qkey_s, nquads)); PB(pb_alloc(&d_qidx, nquads)); PB(pb_alloc(&d_qsorted, nquads));
    hipLaunchKernelGGL(k_pb_iota_widen, GRID(nquads), d_qcband, nquads, d_qkey, d_qidx);
    tb = 0;
    PB_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, d_qkey, d_qkey_s, d_qidx, d_qsorted, (u64)nquads, 0, 16, st));
    PB_HIP(hipMalloc(&d_tmp, tb ? tb : 16));
    PB_HIP(hipcub::DeviceRadixSort::SortPairs(d_tmp, tb, d_qkey, d_qkey_s, d_qidx, d_qsorted, (u64)nquads, 0, 16, st));
    PB_HIP(hipStreamSynchronize(st));
    free_tmp();
    pb_free(d_qkey); pb_free(d_qidx); pb_free(d_qcband);
    {
        uint2 *q_lcol = nullptr;
        PB(pb_alloc(&q_lcol, nquads + 1));
        c->d_pb_lcol = reinterpret_cast<uint16_t *>(q_lcol);
    }
    PB(pb_alloc(&c->d_pb_dst, nquads + 1));
    hipLaunchKernelGGL(k_pb_quads, GRID(nquads), d_qsorted, d_plcol, nquads, reinterpret_cast<uint2 *>(c->d_pb_lcol), c->d_pb_dst);
    PB(pb_alloc(&d_bstart, (u64)nb + 1));
    hipLaunchKernelGGL(k_pb_bounds_u32, GRID(nb + 1), d_qkey_s, nquads, nb, d_bstart);
    std::vector<u32> bstart((size_t)nb + 1);
    PB_HIP(hipMemcpyAsync(bstart.data(), d_bstart, sizeof(u32) * ((size_t)nb + 1), hipMemcpyDeviceToHost, st));
    PB_HIP(hipStreamSynchronize(st));
    pb_free(d_qkey_s); pb_free(d_qsorted); pb_free(d_plcol); pb_free(d_bstart);

    // scatter work units: (column band, first quad, last quad), at most LZX_PB_UNIT entries each
    std::vector<u32> units;
    const u32 unit_quads = LZX_PB_UNIT / 4;
    for (u32 b = 0; b < nb; ++b)
        for (u32 s = bstart[b]; s < bstart[b + 1]; s += unit_quads) {
            units.push_back(b);
            units.push_back(s);
            units.push_back(std::min(bstart[b + 1], s + unit_quads));
        }
    c->pb_units = (u32)(units.size() / 3);
    // units are sorted by column band: those whose band ends inside chunk 0 of the exchange layout come first
    c->pb_units0 = c->pb_units;
    if (c->overlap) {
        const u64 chunk0_end = (u64)c->world * c->xs0;
        u32 u0 = 0;
        while (u0 < c->pb_units && ((u64)units[3 * u0] + 1) * LZX_PB_CB <= chunk0_end) ++u0;
        c->pb_units0 = u0;
    }
    PB(pb_alloc(&c->d_pb_unit, units.size()));
    if (!units.empty())
        PB_HIP(hipMemcpyAsync(c->d_pb_unit, units.data(), sizeof(u32) * units.size(), hipMemcpyHostToDevice, st));

    // 6. gather items (padded positions): one per band; a single-row band above 2 targets is cut into target-sized items
    std::vector<u32> items, multi;
    u32 slots = 0;
    for (u32 R = 0; R < nr; ++R) {
        const u32 beg = rstart[R], end = rstart[R + 1];
        if (beg == end) continue;
        const u32 rows = row0[R + 1] - row0[R];
        if (rows == 1 && end - beg > 2 * target) {
            multi.push_back(row0[R]);
            multi.push_back(slots);
            u32 cnt = 0;
            for (u32 s = beg; s < end; s += target, ++cnt) {
                items.push_back(R); items.push_back(s); items.push_back(std::min(end, s + target));
                items.push_back(slots + cnt);
            }
            multi.push_back(cnt);
            slots += cnt;
        } else {
            items.push_back(R); items.push_back(beg); items.push_back(end); items.push_back(0xffffffffu);
        }
    }
    c->pb_n_items = (u32)(items.size() / 4);
    c->pb_n_multi = (u32)(multi.size() / 3);
    PB(pb_alloc(&c->d_pb_items, items.size())); PB(pb_alloc(&c->d_pb_multi, multi.size())); PB(pb_alloc(&c->d_pb_part, slots));
    if (!items.empty())
        PB_HIP(hipMemcpyAsync(c->d_pb_items, items.data(), sizeof(u32) * items.size(), hipMemcpyHostToDevice, st));
    if (!multi.empty())
        PB_HIP(hipMemcpyAsync(c->d_pb_multi, multi.data(), sizeof(u32) * multi.size(), hipMemcpyHostToDevice, st));

    PB(pb_alloc(&c->d_pb_val, (u64)total_pad + 8));
    PB_HIP(hipMemsetAsync(c->d_pb_val, 0, sizeof(double) * ((u64)total_pad + 8), st));
    PB_HIP(hipStreamSynchronize(st));
    PB_HIP(hipGetLastError());

    c->pb = true;
    c->pb_entries = total;
    c->pb_nr = nr;
    constexpr u32 waves_per_wg = LZX_PB_GATHER_BLOCK / 64;
    c->pb_gather_grid = std::min<u32>((u32)c->cu_count * 2, std::max(1u, (c->pb_n_items + waves_per_wg - 1) / waves_per_wg));
    c->pb_finish_grid = (c->pb_n_multi + LZX_VEC_BLOCK - 1) / LZX_VEC_BLOCK;
    cleanup();
#undef PB
#undef PB_HIP
#undef GRID
    return LZX_OK;
}

int lzx_pb_launch(lzx_ctx *c, const double *x, const double *q_loc, double *v, double *partials, hipEvent_t chunk1_ready)
{
    if (!c->pb) return LZX_OK;
    if (c->pb_units) {
        const size_t lds1 = ((size_t)LZX_PB_CB + 2) * sizeof(double);
        LZX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pb_scatter),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        // column bands of chunk 0 first; the rest once the second chunk of the exchange has arrived
        const u32 first = chunk1_ready ? c->pb_units0 : c->pb_units;
        if (first)
            hipLaunchKernelGGL(k_pb_scatter, dim3(first), dim3(1024), lds1, c->stream, c->d_pb_unit,
                               reinterpret_cast<const uint2 *>(c->d_pb_lcol), c->d_pb_dst, x, c->xlen, c->d_pb_val);
        if (chunk1_ready) {
            LZX_HIP(hipStreamWaitEvent(c->stream, chunk1_ready, 0));
            if (c->pb_units > first)
                hipLaunchKernelGGL(k_pb_scatter, dim3(c->pb_units - first), dim3(1024), lds1, c->stream, c->d_pb_unit + 3 * (size_t)first,
                                   reinterpret_cast<const uint2 *>(c->d_pb_lcol), c->d_pb_dst, x, c->xlen, c->d_pb_val);
        }
    }
    if (c->trace) LZX_HIP(hipEventRecord(c->trace_ev[3], c->stream));
    const size_t lds2 = ((size_t)(LZX_PB_GATHER_BLOCK / 64) * (LZX_PB_RB + 8) + LZX_PB_GATHER_BLOCK / 64) * sizeof(double);
    LZX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pb_gather),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    hipLaunchKernelGGL(k_pb_gather, dim3(c->pb_gather_grid), dim3(LZX_PB_GATHER_BLOCK), lds2, c->stream,
                       reinterpret_cast<const uint4 *>(c->d_pb_items), c->pb_n_items, c->d_pb_row0, c->d_pb_rep, c->d_pb_lrow,
                       c->d_pb_val, v, q_loc, c->d_pb_part, partials);
    if (c->pb_finish_grid)
        hipLaunchKernelGGL(k_pb_finish, dim3(c->pb_finish_grid), dim3(LZX_VEC_BLOCK), 0, c->stream, c->d_pb_multi,
                           c->pb_n_multi, c->d_pb_part, v, q_loc, partials + c->pb_gather_grid);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

// lzx_pb.hip -- propagation-blocked SpMV for the entries whose column is NOT staged in LDS by k_spmv.
//
// Why: past the 4 MiB per-XCD L2 a random 8-byte gather of x costs a whole 128-byte fabric transaction
// (tools/gather_bench.hip: 55 Ggather/s = 7 TB/s of traffic), so the plain CSR gather moves 8x the
// algorithmic bytes (profiles/r1_c3_baseline_pmc.json: 14.5 GB per SpMV on the 10 M-vertex graph).  Here the
// same work is two perfectly streaming passes, 24 bytes per entry:
//   scatter (k_pb_scatter): entries ordered by COLUMN band; the band's 16 Ki x values are staged in LDS,
//       every entry reads its 2-byte column-in-band, looks the value up in LDS and writes it to its slot in
//       the value array -- slots are ordered by ROW band, so a (row band, column band) segment is one
//       contiguous run of writes;
//   gather (k_pb_gather): row bands hold about LZX_PB_TARGET entries each (1 .. 1024 consecutive rows, so
//       heavy rows get bands of their own and every wavefront gets the same amount of work); one wavefront
//       streams a band's values + 2-byte row-in-band, pre-sums equal-row runs with a segmented shuffle scan
//       and adds the run totals into a wave-private LDS y tile (distinct addresses per step: no atomics, fixed
//       order), then adds the tile to v and forms its share of alpha = v . q.  A single row with more than
//       2 * LZX_PB_TARGET entries is cut into items whose totals k_pb_finish adds in order.
// All tables are static (built once per graph by lzx_pb_prepare with two radix sorts).
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

__global__ void k_pb_split_keys(const u64 *keys, u64 count, uint16_t *lrow, u32 *cband, u32 *idx)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const u64 k = keys[i];
    lrow[i] = (uint16_t)((k >> 14) & 0x3ffu);
    cband[i] = (u32)((k >> 24) & 0xffffu);
    idx[i] = (u32)i;
}

// padding entries of the scatter order: column 0 of the band, slot = the spare one behind the value array
__global__ void k_pb_fill_pad(uint16_t *lcol, u32 *dst, u64 count, u32 spare_slot)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        lcol[i] = 0;
        dst[i] = spare_slot;
    }
}

// entry i of the (unpadded) scatter order goes to padded position bstart_pad[b] + (i - bstart[b])
__global__ void k_pb_place(const u64 *keys, const u32 *dst_raw, const u32 *cband_sorted, const u32 *bstart,
                           const u32 *bstart_pad, u64 count, uint16_t *lcol, u32 *dst)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const u32 b = cband_sorted[i];
    const u64 pos = (u64)bstart_pad[b] + (i - bstart[b]);
    const u32 slot = dst_raw[i];
    dst[pos] = slot;
    lcol[pos] = (uint16_t)(keys[slot] & 0x3fffu);
}

// first index whose (key >> shift) & mask >= target, for target = 0..count_targets
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

// ---- the two per-iteration kernels ---------------------------------------------------------------------------
// (Scatter-order arrays are padded so that every column band starts on a multiple of 4 entries; padding entries
// point at the spare slot behind the value array.)
__global__ void __launch_bounds__(1024)
k_pb_scatter(const u32 *unit, const uint16_t *lcol, const u32 *dst, const double *__restrict__ x, u64 xlen,
             double *val, int dbg)
{
    extern __shared__ __attribute__((aligned(16))) double tile[];
    const u32 band = unit[3 * blockIdx.x], beg = unit[3 * blockIdx.x + 1], end = unit[3 * blockIdx.x + 2];
    const u64 base = (u64)band * LZX_PB_CB;
    for (u32 j = threadIdx.x; j < LZX_PB_CB; j += 1024) {
        const u64 p = base + j;
        tile[j] = p < xlen ? x[p] : 0.0;
    }
    __syncthreads();
    // Each wavefront walks its own contiguous share of the unit 64 entries at a time (lane = consecutive entry):
    // loads are contiguous, and so are the stores inside a (row band, column band) run.  8 steps in flight.
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u32 span = (((end - beg) + 15u) / 16u + 63u) & ~63u;   // entries per wavefront, multiple of 64
    const u32 wbeg = beg + wv * span;
    const u32 wend = wbeg + span < end ? wbeg + span : end;
    u32 i = wbeg + lane;
    for (; i + 7 * 64 < wend; i += 8 * 64) {
        u32 cc[8], dd[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            cc[u] = lcol[i + u * 64];
            dd[u] = dst[i + u * 64];
        }
        if (dbg == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) val[dd[u]] = tile[cc[u]];
        } else if (dbg == 5) {          // non-temporal stores
#pragma unroll
            for (int u = 0; u < 8; ++u) __builtin_nontemporal_store(tile[cc[u]], &val[dd[u]]);
        } else if (dbg == 1) {          // no stores: LDS gathers summed
            double s = 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) s += tile[cc[u]] + (double)dd[u];
            if (s == 1.234e-300) val[0] = s;
        } else if (dbg == 2) {          // no LDS: scattered stores of a constant
#pragma unroll
            for (int u = 0; u < 8; ++u) val[dd[u]] = (double)cc[u];
        } else if (dbg == 3) {          // sequential stores (position in scatter order), LDS kept
#pragma unroll
            for (int u = 0; u < 8; ++u) val[i + u * 64] = tile[cc[u]] + (double)dd[u];
        } else {                        // loads only
            u32 s = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) s += cc[u] ^ dd[u];
            if (s == 0x12345u) val[0] = 1.0;
        }
    }
    for (; i < wend; i += 64) val[dst[i]] = tile[lcol[i]];
}

__device__ __forceinline__ double wave_sum_pb(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// item table entry: {row band, begin, end, slot}; slot == 0xffffffff: the item is its band's only one and adds
// straight into v; otherwise it is one of several items of a single-row band and leaves its total in part[slot].
__global__ void __launch_bounds__(LZX_PB_GATHER_BLOCK)
k_pb_gather(const uint4 *items, u32 n_items, const u32 *band_row0, const uint16_t *lrow, const double *val,
            double *v, const double *__restrict__ q_loc, double *part, double *partials, int dbg)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr u32 WAVES = LZX_PB_GATHER_BLOCK / 64;
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double *ytile = lds + (size_t)wv * LZX_PB_RB;   // private to this wavefront
    double *wsum = lds + (size_t)WAVES * LZX_PB_RB;
    const u32 waves = gridDim.x * WAVES;
    double dot = 0.0;
    for (u32 it = blockIdx.x * WAVES + wv; it < n_items; it += waves) {
        const uint4 item = items[it];
        const u32 R = item.x, beg = item.y, end = item.z;
        const u32 row0 = band_row0[R], rows = band_row0[R + 1] - row0;
        if (rows == 1) {
            // one heavy row: plain strided sum, fixed butterfly
            double acc = 0.0;
            u32 i = beg + lane;
            for (; i + 3 * 64 < end; i += 4 * 64) {
                const double a0 = val[i], a1 = val[i + 64], a2 = val[i + 128], a3 = val[i + 192];
                acc += a0; acc += a1; acc += a2; acc += a3;
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
        for (u32 j = lane; j < rows; j += 64) ytile[j] = 0.0;
        __builtin_amdgcn_wave_barrier();
        // chunks of 4 steps; the next chunk's loads are issued before the current one is reduced
        double av[4], nv[4];
        u32 rv[4], nr_[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 i = beg + u * 64 + lane;
            const bool live = i < end;
            av[u] = live ? val[i] : 0.0;
            rv[u] = live ? (u32)lrow[i] : 0xffffu;
        }
        for (u32 i0 = beg; i0 < end; i0 += 4 * 64) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u32 i = i0 + (4 + u) * 64 + lane;
                const bool live = i < end;
                nv[u] = live ? val[i] : 0.0;
                nr_[u] = live ? (u32)lrow[i] : 0xffffu;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                double a = av[u];
                const u32 r = rv[u];
                const bool live = r != 0xffffu;
                // entries are sorted by (column band, row): equal rows form runs of adjacent lanes.  Segmented
                // inclusive scan over the runs (head flags), then the last lane of each run adds the run total.
                // The same row can end two runs of one step only across a column-band boundary; ds_add_f64
                // resolves that.  No lane continues a run (the usual case in bands of light rows): skip the scan.
                const u32 rprev = __shfl_up(r, 1, 64);
                const bool head = (lane == 0) || (rprev != r);
                if (dbg < 2 && __ballot(!head) != 0ull) {
                    bool f = head;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const double ua = __shfl_up(a, o, 64);
                        const int uf = __shfl_up((int)f, o, 64);
                        if ((int)lane >= o && !f) {
                            a += ua;
                            f = uf != 0;
                        }
                    }
                }
                const int next_head = __shfl_down((int)head, 1, 64);
                if (dbg == 3) { if (live) atomicAdd(&ytile[r], av[u]); }
                else if (dbg == 1) { if (live) ytile[(r + lane) & 1023] = a; }
                else if (live && (lane == 63 || next_head)) atomicAdd(&ytile[r], a);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                av[u] = nv[u];
                rv[u] = nr_[u];
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (u32 j = lane; j < rows; j += 64) {
            const double y = ytile[j];
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
    if (total >= (1ull << 32) - 8) LZX_FAIL(LZX_ERR_LIMIT, "propagation blocking: %llu entries do not fit 32-bit slots", (unsigned long long)total);
    const u32 nb = (u32)((c->xlen + LZX_PB_CB - 1) / LZX_PB_CB);
    if (nb >= (1u << 16)) LZX_FAIL(LZX_ERR_LIMIT, "propagation blocking: %u column bands (limit 65535)", nb);

    // ---- row bands: consecutive local rows, closed at ~LZX_PB_TARGET entries or LZX_PB_RB rows; a row heavier
    //      than the target is a band of its own.  Every wavefront of the gather pass gets one band (or one
    //      item of a very heavy row), so work per wavefront is even although degrees are not.
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
    u32 *d_cband = nullptr, *d_cband_s = nullptr, *d_idx = nullptr, *d_bstart = nullptr, *d_rstart = nullptr;
    u32 *d_dst_raw = nullptr, *d_bstart_pad = nullptr;
    void *d_tmp = nullptr;
    int rc = LZX_OK;
    auto cleanup = [&]() {
        pb_free(d_keys); pb_free(d_sorted); pb_free(d_cband); pb_free(d_cband_s); pb_free(d_idx); pb_free(d_bstart);
        pb_free(d_rstart); pb_free(d_dst_raw); pb_free(d_bstart_pad);
        if (d_tmp) (void)hipFree(d_tmp);
        d_tmp = nullptr;
    };
#define PB(call) do { rc = (call); if (rc != LZX_OK) { cleanup(); lzx_pb_release(c); return rc; } } while (0)
#define PB_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
        lzx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); cleanup(); lzx_pb_release(c); \
        return e_ == hipErrorOutOfMemory ? LZX_ERR_NOMEM : LZX_ERR_HIP; } } while (0)

    PB(pb_alloc(&c->d_pb_row0, (u64)nr + 1));
    PB_HIP(hipMemcpyAsync(c->d_pb_row0, row0.data(), sizeof(u32) * ((size_t)nr + 1), hipMemcpyHostToDevice, st));

    // 1. emit + sort by (row band, column band, row, column): this is the GATHER order
    PB(pb_alloc(&d_keys, total)); PB(pb_alloc(&d_sorted, total));
    if (c->n_loc_real)
        hipLaunchKernelGGL(k_pb_emit, dim3(c->n_loc_real), dim3(64), 0, st, c->d_row_ptr, c->d_col_idx, d_code,
                           d_old_of_local, d_deg_local, d_nh_off, c->hub_real, c->d_pb_row0, nr, d_keys);
    size_t tb = 0;
    PB_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, d_keys, d_sorted, (u64)total, 0, 64, st));
    PB_HIP(hipMalloc(&d_tmp, tb ? tb : 16));
    PB_HIP(hipcub::DeviceRadixSort::SortKeys(d_tmp, tb, d_keys, d_sorted, (u64)total, 0, 64, st));
    PB_HIP(hipStreamSynchronize(st));
    (void)hipFree(d_tmp); d_tmp = nullptr;
    pb_free(d_keys);

    // 2. per-entry row-in-band (gather order), and a stable sort by column band: the SCATTER order
    PB(pb_alloc(&c->d_pb_lrow, total));
    PB(pb_alloc(&d_cband, total)); PB(pb_alloc(&d_cband_s, total)); PB(pb_alloc(&d_idx, total));
    PB(pb_alloc(&d_dst_raw, total));
    const u32 g = (u32)((total + 255) / 256);
    hipLaunchKernelGGL(k_pb_split_keys, dim3(g), dim3(256), 0, st, d_sorted, total, c->d_pb_lrow, d_cband, d_idx);
    tb = 0;
    PB_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, d_cband, d_cband_s, d_idx, d_dst_raw, (u64)total, 0, 16, st));
    PB_HIP(hipMalloc(&d_tmp, tb ? tb : 16));
    PB_HIP(hipcub::DeviceRadixSort::SortPairs(d_tmp, tb, d_cband, d_cband_s, d_idx, d_dst_raw, (u64)total, 0, 16, st));
    PB_HIP(hipStreamSynchronize(st));
    (void)hipFree(d_tmp); d_tmp = nullptr;
    pb_free(d_cband); pb_free(d_idx);

    // 3. band boundaries in both orders
    PB(pb_alloc(&d_rstart, (u64)nr + 1)); PB(pb_alloc(&d_bstart, (u64)nb + 1));
    hipLaunchKernelGGL(k_pb_bounds_u32, dim3((nb + 256) / 256), dim3(256), 0, st, d_cband_s, total, nb, d_bstart);
    hipLaunchKernelGGL(k_pb_bounds_u64, dim3((nr + 256) / 256), dim3(256), 0, st, d_sorted, total, 40u, nr, d_rstart);
    std::vector<u32> bstart((size_t)nb + 1), rstart((size_t)nr + 1);
    PB_HIP(hipMemcpyAsync(bstart.data(), d_bstart, sizeof(u32) * ((size_t)nb + 1), hipMemcpyDeviceToHost, st));
    PB_HIP(hipMemcpyAsync(rstart.data(), d_rstart, sizeof(u32) * ((size_t)nr + 1), hipMemcpyDeviceToHost, st));
    PB_HIP(hipStreamSynchronize(st));

    // 4. scatter order, padded so that every column band starts on a multiple of 4 entries; work units:
    //    (column band, begin, end) in padded positions, at most LZX_PB_UNIT entries each
    std::vector<u32> bstart_pad((size_t)nb + 1);
    {
        u64 pos = 0;
        for (u32 b = 0; b < nb; ++b) {
            bstart_pad[b] = (u32)pos;
            pos += (bstart[b + 1] - bstart[b] + 3u) & ~3u;
        }
        bstart_pad[nb] = (u32)pos;
        if (pos >= (1ull << 32) - 8) { cleanup(); lzx_pb_release(c); LZX_FAIL(LZX_ERR_LIMIT, "propagation blocking: padded entry count overflows"); }
    }
    const u64 total_pad = bstart_pad[nb];
    PB(pb_alloc(&d_bstart_pad, (u64)nb + 1));
    PB_HIP(hipMemcpyAsync(d_bstart_pad, bstart_pad.data(), sizeof(u32) * ((size_t)nb + 1), hipMemcpyHostToDevice, st));
    PB(pb_alloc(&c->d_pb_lcol, total_pad + 8)); PB(pb_alloc(&c->d_pb_dst, total_pad + 8));
    if (total_pad) {
        hipLaunchKernelGGL(k_pb_fill_pad, dim3((u32)((total_pad + 255) / 256)), dim3(256), 0, st, c->d_pb_lcol, c->d_pb_dst,
                           total_pad, (u32)total);
        hipLaunchKernelGGL(k_pb_place, dim3(g), dim3(256), 0, st, d_sorted, d_dst_raw, d_cband_s, d_bstart, d_bstart_pad, total,
                           c->d_pb_lcol, c->d_pb_dst);
    }
    std::vector<u32> units;
    for (u32 b = 0; b < nb; ++b)
        for (u32 s = bstart_pad[b]; s < bstart_pad[b + 1]; s += LZX_PB_UNIT) {
            units.push_back(b);
            units.push_back(s);
            units.push_back(std::min(bstart_pad[b + 1], s + LZX_PB_UNIT));
        }
    c->pb_units = (u32)(units.size() / 3);
    PB(pb_alloc(&c->d_pb_unit, units.size()));
    if (!units.empty())
        PB_HIP(hipMemcpyAsync(c->d_pb_unit, units.data(), sizeof(u32) * units.size(), hipMemcpyHostToDevice, st));

    // 5. gather items: one per band; a single-row band above 2 targets is cut into target-sized items
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

    PB(pb_alloc(&c->d_pb_val, total + 8));   // + the spare slot padding entries write to
    PB_HIP(hipMemsetAsync(c->d_pb_val, 0, sizeof(double) * (total + 8), st));
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
    return LZX_OK;
}

int lzx_pb_launch(lzx_ctx *c, const double *x, const double *q_loc, double *v, double *partials)
{
    if (!c->pb) return LZX_OK;
    if (c->pb_units) {
        const size_t lds1 = (size_t)LZX_PB_CB * sizeof(double);
        LZX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pb_scatter),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        hipLaunchKernelGGL(k_pb_scatter, dim3(c->pb_units), dim3(1024), lds1, c->stream, c->d_pb_unit, c->d_pb_lcol,
                           c->d_pb_dst, x, c->xlen, c->d_pb_val, (int)(c->pb_debug & 15));
    }
    if (c->trace) LZX_HIP(hipEventRecord(c->trace_ev[3], c->stream));
    const size_t lds2 = ((size_t)(LZX_PB_GATHER_BLOCK / 64) * LZX_PB_RB + LZX_PB_GATHER_BLOCK / 64) * sizeof(double);
    LZX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pb_gather),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    hipLaunchKernelGGL(k_pb_gather, dim3(c->pb_gather_grid), dim3(LZX_PB_GATHER_BLOCK), lds2, c->stream,
                       reinterpret_cast<const uint4 *>(c->d_pb_items), c->pb_n_items, c->d_pb_row0, c->d_pb_lrow,
                       c->d_pb_val, v, q_loc, c->d_pb_part, partials, (int)(c->pb_debug >> 4));
    if (c->pb_finish_grid)
        hipLaunchKernelGGL(k_pb_finish, dim3(c->pb_finish_grid), dim3(LZX_VEC_BLOCK), 0, c->stream, c->d_pb_multi,
                           c->pb_n_multi, c->d_pb_part, v, q_loc, partials + c->pb_gather_grid);
    LZX_HIP(hipGetLastError());
    return LZX_OK;
}

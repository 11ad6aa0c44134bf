// tools/tile_bench.hip -- feasibility of a 2-D tiled SpMV pass (x band AND y tile of a row group in LDS, no value stream):
// the inner loop on synthetic packets.  A packet = 8 half-words; bit 15 set = "switch to this row of the group",
// else a column of the band; the first half-word of every packet is a row marker.  Lane = packet: running sum over
// its columns, flushed into the y tile with ds_add_f64 at every marker and at the end.  Parameters: entries per row
// segment D (1, 2, 3, 7, 50), band 8 Ki columns (64 KiB), y tile 8 Ki rows (64 KiB), 16 wavefronts, 1 workgroup/CU.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/tile_bench.hip -o tools/_tile_bench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32;
typedef uint64_t u64;
extern __shared__ __attribute__((aligned(16))) double lds[];
constexpr u32 CB = 8192, RB = 8192;

template <int MODE /*0 full, 1 no y adds, 2 no x look-ups*/>
__global__ void __launch_bounds__(1024) k_tile(const uint4 *pk, u64 npackets, const double *x, double *y)
{
    double *xt = lds, *yt = lds + CB + 2;
    for (u32 j = threadIdx.x; j < CB; j += 1024) xt[j] = x[j];
    if (threadIdx.x < 2) xt[CB + threadIdx.x] = 0.0;
    for (u32 j = threadIdx.x; j < RB + 8; j += 1024) yt[j] = 0.0;
    __syncthreads();
    const u64 per = npackets / gridDim.x, p0 = per * blockIdx.x;
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    auto body = [&](const uint4 &c) {
        const u32 w[4] = {c.x, c.y, c.z, c.w};
        u32 cur = w[0] & 0x7fffu;   // half-word 0: row marker
        double s = 0.0;
#pragma unroll
        for (int e = 1; e < 8; ++e) {
            const u32 h = (e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu);
            const bool mk = (e & 1) ? ((int)w[e >> 1] < 0) : ((w[e >> 1] & 0x8000u) != 0u);
            const double xv = MODE == 2 ? (double)(h & 0x1fffu) : xt[mk ? CB : (h & 0x7fffu)];
            if (__ballot(mk)) {
                if (mk) {
                    if (MODE != 1) atomicAdd(&yt[cur], s);
                    cur = h & 0x7fffu;
                    s = 0.0;
                }
            }
            s += xv;
        }
        if (MODE != 1) atomicAdd(&yt[cur], s);
        else if (s == 1.2345e-300) yt[cur] = s;
    };
    u64 p = wv * 64 + lane;
    for (; p + 3 * 1024 < per; p += 4 * 1024) {
        uint4 c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = pk[p0 + p + u * 1024];
#pragma unroll
        for (int u = 0; u < 4; ++u) body(c[u]);
    }
    __syncthreads();
    for (u32 j = threadIdx.x; j < RB; j += 1024) y[(u64)blockIdx.x * RB + j] = yt[j];
}

int main()
{
    const u64 npk = 32ull << 20;   // 32 Mi packets = 512 MiB
    uint4 *pk; double *x, *y;
    CHECK(hipMalloc(&pk, npk * 16)); CHECK(hipMalloc(&x, 1 << 20)); CHECK(hipMalloc(&y, 256ull * RB * 8));
    CHECK(hipMemset(x, 0, 1 << 20));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const size_t ldsb = (CB + 2 + RB + 8) * 8;
    CHECK(hipFuncSetAttribute((const void *)k_tile<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    CHECK(hipFuncSetAttribute((const void *)k_tile<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    CHECK(hipFuncSetAttribute((const void *)k_tile<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    std::vector<uint16_t> h(npk * 8);
    printf("entries per row segment | entries per packet | ms full / no y adds / no x look-ups | full: TB/s of packets, G entries/s\n");
    for (u32 D : {1u, 2u, 3u, 7u, 50u}) {
        u32 q = 99u + D, row = 0, left = D;
        u64 entries = 0;
        for (u64 p = 0; p < npk; ++p) {
            uint16_t *o = &h[p * 8];
            o[0] = (uint16_t)(0x8000u | (row & 0x1fffu));
            for (int e = 1; e < 8; ++e) {
                if (left == 0) {   // next row: rows ascend, so that neighbouring lanes hit neighbouring y slots (as sorted data would)
                    row = (row + 1) & 0x1fffu;
                    left = D;
                    o[e] = (uint16_t)(0x8000u | row);
                } else {
                    q = q * 1664525u + 1013904223u;
                    o[e] = (uint16_t)((q >> 10) & 0x1fffu);
                    --left;
                    ++entries;
                }
            }
        }
        CHECK(hipMemcpy(pk, h.data(), npk * 16, hipMemcpyHostToDevice));
        float t[3];
        auto run = [&](int mode) {
            float best = 1e30f;
            for (int r = 0; r < 4; ++r) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) k_tile<0><<<256, 1024, ldsb>>>(pk, npk, x, y);
                if (mode == 1) k_tile<1><<<256, 1024, ldsb>>>(pk, npk, x, y);
                if (mode == 2) k_tile<2><<<256, 1024, ldsb>>>(pk, npk, x, y);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (r > 0 && ms < best) best = ms;
            }
            CHECK(hipGetLastError());
            return best;
        };
        for (int m = 0; m < 3; ++m) t[m] = run(m);
        printf("%3u | %.2f | %.3f / %.3f / %.3f | %.2f TB/s, %.0f G entries/s\n", D, entries / (double)npk, t[0], t[1], t[2],
               npk * 16.0 / t[0] / 1e9, entries / t[0] / 1e6);
        fflush(stdout);
    }
    return 0;
}

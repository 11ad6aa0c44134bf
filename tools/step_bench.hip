// tools/step_bench.hip -- the reduced step of the scatter pass (csrc/lzx_pb.hip: pbr_step) in isolation, as a function
// of piece density: synthetic steps whose pieces are D entries long (D = 1 ... 512), random columns, one workgroup of
// 16 wavefronts per CU with the 128 KiB band in LDS, every workgroup streaming its own contiguous share of the steps.
// Variants: full; no value stores; no LDS look-ups; planes replaced by ONE dense store per lane (what the pass would
// cost if pieces were free); values compacted across planes in registers (ds_permute) and stored 64 at a time.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/step_bench.hip -o tools/_step_bench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32;
typedef uint64_t u64;

extern __shared__ __attribute__((aligned(16))) double tile[];

__device__ __forceinline__ u32 lanes_below(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}

// MODE 0 full, 1 no stores, 2 no LDS look-ups, 3 dense store instead of planes, 4 register compaction
template <int MODE>
__device__ __forceinline__ void step(const uint4 &c, u32 pos, const double *tl, double *carry, u32 lane, double *val)
{
    const u32 w[4] = {c.x, c.y, c.z, c.w};
    double xv[8];
    bool f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const u32 h = (e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu);
        xv[e] = MODE == 2 ? (double)(h & 0x7fffu) : tl[h & 0x7fffu];
        f[e] = (e & 1) ? ((int)w[e >> 1] < 0) : ((w[e >> 1] & 0x8000u) != 0u);
    }
    if (MODE == 3) {
        double t = 0.0;
#pragma unroll
        for (int e = 0; e < 8; ++e) t += xv[e];
        val[pos + lane] = t;
        return;
    }
    const bool has = f[0] | f[1] | f[2] | f[3] | f[4] | f[5] | f[6] | f[7];
    double t = 0.0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        t += xv[e];
        t = f[e] ? 0.0 : t;
    }
    const unsigned long long holders = __ballot(has);
    const unsigned long long before = holders & ((1ull << lane) - 1ull);
    const u32 from = before ? 64u - (u32)__clzll((long long)before) : 0u;
    atomicAdd(&carry[has ? lane + 1 : from], t);
    __builtin_amdgcn_wave_barrier();
    double s = has ? carry[from] : 0.0;
    __builtin_amdgcn_wave_barrier();
    if (has) carry[from] = 0.0;
    double *out = val + pos;
    if (MODE == 4) {
        // pieces of all planes compacted into a rolling 64-lane register and stored 64 at a time
        double acc = 0.0;
        u32 fill = 0;   // wave-uniform
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s += xv[e];
            const unsigned long long m = __ballot(f[e]);
            if (m) {
                const u32 P = (u32)__popcll(m);
                const u32 r = lanes_below(m);
                // push (ds_permute): the lane holding the r-th piece of the plane sends it to lane fill + r; every lane
                // must send somewhere, so the lanes without a piece take the 64 - P destinations nobody reads
                const u32 dst = (f[e] ? fill + r : fill + P + (lane - r)) & 63u;
                const int lo = __builtin_amdgcn_ds_permute((int)(dst << 2), __double2loint(s));
                const int hi = __builtin_amdgcn_ds_permute((int)(dst << 2), __double2hiint(s));
                const double got = __hiloint2double(hi, lo);
                const u32 endp = fill + P;
                const bool cur = lane >= fill && lane < (endp < 64u ? endp : 64u);
                acc = cur ? got : acc;
                if (endp >= 64u) {
                    out[lane] = acc;
                    out += 64;
                    acc = lane < endp - 64u ? got : 0.0;
                }
                fill = endp & 63u;
                s = f[e] ? 0.0 : s;
            }
        }
        if (fill && lane < fill) out[lane] = acc;
        return;
    }
    u32 done = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        s += xv[e];
        const unsigned long long m = __ballot(f[e]);
        if (m) {
            if (f[e] && MODE != 1) out[done + lanes_below(m)] = s;
            s = f[e] ? 0.0 : s;
            done += (u32)__popcll(m);
        }
    }
    if (MODE == 1 && s == 1.2345e-300) out[0] = s;
}

template <int MODE>
__global__ void __launch_bounds__(1024) k_steps(const uint4 *code, const u32 *sbase, u64 nsteps, const double *x, double *val)
{
    for (u32 j = threadIdx.x; j < 16384; j += 1024) tile[j] = x[j];
    if (threadIdx.x < 2) tile[16384 + threadIdx.x] = 0.0;
    for (u32 j = threadIdx.x; j < 16 * 66; j += 1024) tile[16384 + 2 + j] = 0.0;
    __syncthreads();
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double *carry = tile + 16384 + 2 + wv * 66;
    const u64 per = nsteps / gridDim.x, s0 = per * blockIdx.x;
    u64 s = wv;
    for (; s + 3 * 16 < per; s += 4 * 16) {
        uint4 c[4];
        u32 b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            c[u] = code[(s0 + s + u * 16) * 64 + lane];
            b[u] = (u32)__builtin_amdgcn_readfirstlane((int)sbase[s0 + s + u * 16]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) step<MODE>(c[u], b[u], tile, carry, lane, val);
    }
}

static hipEvent_t e0, e1;
template <typename F>
static float timeit(F f)
{
    f();
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        CHECK(hipEventRecord(e0));
        f();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    CHECK(hipGetLastError());
    return best;
}

int main()
{
    const u64 nsteps = 256 * 2048;   // 512 Ki steps = 512 MiB of codes, 268 M entries: the blocked part of C3
    uint4 *code; u32 *sbase; double *x, *val;
    CHECK(hipMalloc(&code, nsteps * 1024)); CHECK(hipMalloc(&sbase, nsteps * 4)); CHECK(hipMalloc(&x, 1 << 20));
    CHECK(hipMalloc(&val, nsteps * 512 * 8 + 4096));
    CHECK(hipMemset(x, 0, 1 << 20));
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<uint16_t> h(nsteps * 512);
    std::vector<u32> hb(nsteps);
    const size_t lds = (16384 + 2 + 16 * 66) * 8;
#define SETLDS(M) CHECK(hipFuncSetAttribute((const void *)k_steps<M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
    SETLDS(0); SETLDS(1); SETLDS(2); SETLDS(3); SETLDS(4);
    printf("piece length D | pieces/step | ms: full  no-stores  no-LDS  dense-store  reg-compaction | full: clk per step per CU, TB/s read+written\n");
    for (u32 D : {1u, 2u, 3u, 5u, 8u, 16u, 64u, 512u}) {
        u32 q = 12345u + D;
        u64 pos = 0;
        for (u64 s = 0; s < nsteps; ++s) {
            hb[s] = (u32)pos;
            u32 pieces = 0;
            for (u32 i = 0; i < 512; ++i) {
                q = q * 1664525u + 1013904223u;
                uint16_t cdv = (uint16_t)((q >> 9) & 0x3fffu);
                if ((i + 1) % D == 0) { cdv |= 0x8000u; ++pieces; }
                h[s * 512 + i] = cdv;
            }
            pos += pieces;
            pos = (pos + 7) & ~7ull;
        }
        CHECK(hipMemcpy(code, h.data(), nsteps * 1024, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(sbase, hb.data(), nsteps * 4, hipMemcpyHostToDevice));
        float t[5];
        t[0] = timeit([&] { k_steps<0><<<256, 1024, lds>>>(code, sbase, nsteps, x, val); });
        t[1] = timeit([&] { k_steps<1><<<256, 1024, lds>>>(code, sbase, nsteps, x, val); });
        t[2] = timeit([&] { k_steps<2><<<256, 1024, lds>>>(code, sbase, nsteps, x, val); });
        t[3] = timeit([&] { k_steps<3><<<256, 1024, lds>>>(code, sbase, nsteps, x, val); });
        t[4] = timeit([&] { k_steps<4><<<256, 1024, lds>>>(code, sbase, nsteps, x, val); });
        const double pieces = 512.0 / D;
        printf("%5u | %6.1f | %7.3f %7.3f %7.3f %7.3f %7.3f | %6.0f clk  %5.2f TB/s\n", D, pieces, t[0], t[1], t[2], t[3], t[4],
               t[0] * 1e-3 * 2.4e9 / (nsteps / 256.0), (nsteps * 1024.0 + nsteps * pieces * 8.0) / t[0] / 1e9);
        fflush(stdout);
    }
    return 0;
}

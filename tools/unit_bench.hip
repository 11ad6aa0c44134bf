// tools/unit_bench.hip -- what the building blocks of the blocked SpMV can reach on this chip, one at a time:
//   rd / wr     streaming 16-byte loads / stores at a given occupancy (LDS bytes per workgroup set the workgroups
//               per CU), loads in flight per lane, and stream shape (grid-interleaved or one contiguous chunk per WG)
//   ldsadd      ds_add_f64 (no return) throughput, conflict-free and with 2 / 4 lanes per slot
//   ldsrd       random ds_read_b64 from a 128 KiB tile
//   gath        the gather pass's inner loop alone: 16 B of values + 4 B of slots per lane, two LDS adds
//               (atomic, or plain read-modify-write for comparison)
//   scat        the scatter pass's inner loop alone: 16 B of codes per lane, 8 LDS look-ups, one 8-byte store per
//               lane (dense) or one per 4 lanes (compacted)
// Build: hipcc --offload-arch=gfx950 -O3 tools/unit_bench.hip -o tools/_unit_bench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32;
typedef uint64_t u64;

extern __shared__ __attribute__((aligned(16))) double lds[];

template <int U, bool CHUNK>
__global__ void __launch_bounds__(1024) k_rd(const uint4 *p, u64 n16, double *out)
{
    const u32 T = blockDim.x;
    u64 acc = 0;
    if (threadIdx.x == 0) lds[0] = 0.0;
    if (CHUNK) {
        const u64 per = n16 / gridDim.x;
        const uint4 *q = p + per * blockIdx.x;
        for (u64 i = threadIdx.x; i + (U - 1) * T < per; i += (u64)U * T) {
            uint4 c[U];
#pragma unroll
            for (int u = 0; u < U; ++u) c[u] = q[i + (u64)u * T];
#pragma unroll
            for (int u = 0; u < U; ++u) acc += c[u].x ^ c[u].y ^ c[u].z ^ c[u].w;
        }
    } else {
        const u64 nt = (u64)gridDim.x * T;
        for (u64 i = (u64)blockIdx.x * T + threadIdx.x; i + (U - 1) * nt < n16; i += U * nt) {
            uint4 c[U];
#pragma unroll
            for (int u = 0; u < U; ++u) c[u] = p[i + u * nt];
#pragma unroll
            for (int u = 0; u < U; ++u) acc += c[u].x ^ c[u].y ^ c[u].z ^ c[u].w;
        }
    }
    if (acc == 0x123456789abcull) out[0] = (double)acc;
}

template <bool CHUNK>
__global__ void __launch_bounds__(1024) k_wr(uint4 *p, u64 n16, u32 v)
{
    const u32 T = blockDim.x;
    if (threadIdx.x == 0) lds[0] = 0.0;
    if (CHUNK) {
        const u64 per = n16 / gridDim.x;
        uint4 *q = p + per * blockIdx.x;
        for (u64 i = threadIdx.x; i < per; i += T) q[i] = make_uint4(v, v, v, (u32)i);
    } else {
        const u64 nt = (u64)gridDim.x * T;
        for (u64 i = (u64)blockIdx.x * T + threadIdx.x; i < n16; i += nt) p[i] = make_uint4(v, v, v, (u32)i);
    }
}

// ds_add_f64: every lane adds into slot perm(lane, it) of a wave-private 1024-slot tile; SHARE lanes share a slot
template <int SHARE>
__global__ void __launch_bounds__(1024) k_ldsadd(u32 iters, double *out)
{
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double *tile = lds + wv * 1032;
    for (u32 j = lane; j < 1032; j += 64) tile[j] = 0.0;
    __syncthreads();
    u32 s = lane / SHARE;   // consecutive slots: distinct banks; every lane moves by the same odd stride
    for (u32 it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            atomicAdd(&tile[s & 1023u], 1.0);
            s += 131u;
        }
    }
    __syncthreads();
    if (tile[lane] == 1.2345e-300) out[0] = tile[lane];
}

// random ds_read_b64 from a CB-double tile
__global__ void __launch_bounds__(1024) k_ldsrd(u32 iters, u32 cb, double *out)
{
    for (u32 j = threadIdx.x; j < cb; j += 1024) lds[j] = (double)j;
    __syncthreads();
    u32 s = threadIdx.x * 2654435761u + blockIdx.x;
    double acc = 0.0;
    for (u32 it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            acc += lds[(s >> 8) & (cb - 1u)];
        }
    }
    if (acc == 1.2345e-300) out[0] = acc;
}

// gather-pass inner loop: wave-private 1032-slot tiles, 8 waves per workgroup, blocks of 128 values round-robin
template <int MODE /*0 atomic, 1 plain rmw, 2 no lds*/, int U>
__global__ void __launch_bounds__(512) k_gath(const double *val, const uint16_t *slot, u64 nblocks, double *out)
{
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double *tile = lds + wv * 1032;
    for (u32 j = lane; j < 1032; j += 64) tile[j] = 0.0;
    __syncthreads();
    const u64 per = nblocks / gridDim.x;
    const u64 b0 = per * blockIdx.x;
    double acc = 0.0;
    for (u64 kb = wv; kb + (U - 1) * 8 < per; kb += U * 8) {
        double2 av[U];
        u32 sv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u64 p = (b0 + kb + u * 8) * 128u + lane * 2;
            av[u] = *reinterpret_cast<const double2 *>(val + p);
            sv[u] = *reinterpret_cast<const u32 *>(slot + p);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE == 0) {
                atomicAdd(&tile[sv[u] & 0x3ffu], av[u].x);
                atomicAdd(&tile[(sv[u] >> 16) & 0x3ffu], av[u].y);
            } else if (MODE == 1) {
                tile[sv[u] & 0x3ffu] += av[u].x;
                tile[(sv[u] >> 16) & 0x3ffu] += av[u].y;
            } else {
                acc += av[u].x + av[u].y + (double)sv[u];
            }
        }
    }
    __syncthreads();
    if (tile[lane] + acc == 1.2345e-300) out[0] = tile[lane];
}

// scatter-pass inner loop: 128 KiB tile, 16 waves, steps of 64 x 16 B codes; PIECES lanes per store (1 = dense)
template <int U, int SPARSE>
__global__ void __launch_bounds__(1024) k_scat(const uint4 *code, u64 nsteps, const double *x, double *val)
{
    for (u32 j = threadIdx.x; j < 16384; j += 1024) lds[j] = x[j];
    __syncthreads();
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u64 per = nsteps / gridDim.x;
    const u64 s0 = per * blockIdx.x;
    for (u64 s = wv; s + (U - 1) * 16 < per; s += U * 16) {
        uint4 c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) c[u] = code[(s0 + s + u * 16) * 64 + lane];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double t = 0.0;
            t += lds[c[u].x & 0x3fffu];
            t += lds[(c[u].x >> 16) & 0x3fffu];
            t += lds[c[u].y & 0x3fffu];
            t += lds[(c[u].y >> 16) & 0x3fffu];
            t += lds[c[u].z & 0x3fffu];
            t += lds[(c[u].z >> 16) & 0x3fffu];
            t += lds[c[u].w & 0x3fffu];
            t += lds[(c[u].w >> 16) & 0x3fffu];
            if (SPARSE == 1) val[(s0 + s + u * 16) * 64 + lane] = t;
            else if ((lane % SPARSE) == 0) val[((s0 + s + u * 16) * 64 + lane) / SPARSE] = t;
        }
    }
}

static hipEvent_t e0, e1;
template <typename F>
static float timeit(F f, int reps = 5)
{
    f();
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
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

int main(int argc, char **argv)
{
    const u64 S = 2ull << 30;
    uint4 *buf; uint4 *buf2; double *out; double *x;
    CHECK(hipMalloc(&buf, S)); CHECK(hipMalloc(&buf2, S)); CHECK(hipMalloc(&out, 4096)); CHECK(hipMalloc(&x, 1 << 20));
    CHECK(hipMemset(buf, 0x11, S)); CHECK(hipMemset(buf2, 0, S)); CHECK(hipMemset(x, 0, 1 << 20));
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const u64 n16 = S / 16;
#define MAXLDS(k) CHECK(hipFuncSetAttribute((const void *)(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
    struct Occ { int threads, lds_kb, wgs; const char *name; };
    const Occ occ[] = {{1024, 128, 1, "1x1024thr/128K (16 waves)"}, {1024, 64, 2, "2x1024thr/64K (32 waves)"}, {512, 64, 2, "2x512thr/64K (16 waves)"},
                       {256, 16, 8, "8x256thr/16K (32 waves)"}, {256, 36, 4, "4x256thr/36K (16 waves)"}, {256, 72, 2, "2x256thr/72K (8 waves)"}};
    printf("== streaming reads of 2 GiB, TB/s (rows: occupancy; columns: 16-B loads in flight per lane U=1,2,4,8; grid-interleaved | chunk per WG)\n");
    for (const Occ &o : occ) {
        printf("%-28s", o.name);
        const int grid = 256 * o.wgs;
        const size_t sh = (size_t)o.lds_kb * 1024;
#define RD(U, C) { MAXLDS((k_rd<U, C>)); float ms = timeit([&] { k_rd<U, C><<<grid, o.threads, sh>>>(buf, n16, out); }); printf(" %5.2f", S / ms / 1e9); }
        RD(1, false) RD(2, false) RD(4, false) RD(8, false) printf("  |");
        RD(1, true) RD(2, true) RD(4, true) RD(8, true) printf("\n");
        fflush(stdout);
    }
    printf("== same, 4x the workgroups (grid = 4 waves of workgroups), U=4: ");
    for (const Occ &o : occ) { const int grid = 256 * o.wgs * 4; const size_t sh = (size_t)o.lds_kb * 1024; float ms = timeit([&] { k_rd<4, true><<<grid, o.threads, sh>>>(buf, n16, out); }); printf(" %5.2f", S / ms / 1e9); }
    printf("\n== streaming stores of 2 GiB, TB/s (interleaved | chunk)\n");
    for (const Occ &o : occ) {
        const int grid = 256 * o.wgs; const size_t sh = (size_t)o.lds_kb * 1024;
        MAXLDS((k_wr<false>)); MAXLDS((k_wr<true>));
        float a = timeit([&] { k_wr<false><<<grid, o.threads, sh>>>(buf2, n16, 7); });
        float b = timeit([&] { k_wr<true><<<grid, o.threads, sh>>>(buf2, n16, 7); });
        printf("%-28s %5.2f | %5.2f\n", o.name, S / a / 1e9, S / b / 1e9);
    }
    fflush(stdout);
    {
        const u32 iters = 2000;
        const double instr = 256.0 * 16 * iters * 8;   // wave-instructions chip-wide
        MAXLDS((k_ldsadd<1>)); MAXLDS((k_ldsadd<2>)); MAXLDS((k_ldsadd<4>));
        float a = timeit([&] { k_ldsadd<1><<<256, 1024, 16 * 1032 * 8>>>(iters, out); });
        float b = timeit([&] { k_ldsadd<2><<<256, 1024, 16 * 1032 * 8>>>(iters, out); });
        float c = timeit([&] { k_ldsadd<4><<<256, 1024, 16 * 1032 * 8>>>(iters, out); });
        printf("== ds_add_f64, 16 waves/CU: conflict-free %.1f clk/wave-instr/CU (%.0f G adds/s), 2 lanes/slot %.1f, 4 lanes/slot %.1f  [clk at 2.4 GHz]\n",
               a * 1e-3 * 2.4e9 / (instr / 256), instr * 64 / a / 1e6, b * 1e-3 * 2.4e9 / (instr / 256), c * 1e-3 * 2.4e9 / (instr / 256));
        MAXLDS(k_ldsrd);
        for (u32 cb : {1024u, 8192u, 16384u}) {
            float d = timeit([&] { k_ldsrd<<<256, 1024, 128 * 1024>>>(iters, cb, out); });
            printf("== random ds_read_b64 from %u doubles, 16 waves/CU: %.1f clk/wave-instr/CU (%.0f G reads/s)\n", cb, d * 1e-3 * 2.4e9 / (instr / 256), instr * 64 / d / 1e6);
        }
    }
    fflush(stdout);
    {
        // gather loop: 64 Mi values (512 MiB) + slots (128 MiB); slots = conflict-free pattern within a block
        const u64 nval = 64ull << 20, nblocks = nval / 128;
        double *val = reinterpret_cast<double *>(buf);
        uint16_t *slot = reinterpret_cast<uint16_t *>(buf2);
        uint16_t *h = (uint16_t *)malloc(nval * 2);
        u32 r = 12345;
        for (u64 b = 0; b < nblocks; ++b) {
            r = r * 1664525u + 1013904223u;
            const u32 base = r >> 12;
            for (u32 i = 0; i < 128; ++i) h[b * 128 + i] = (uint16_t)((base + (i >> 1) * 5 + (i & 1) * 517) & 1023u);
        }
        CHECK(hipMemcpy(slot, h, nval * 2, hipMemcpyHostToDevice));
        free(h);
        const double bytes = nval * 10.0;
        for (int pattern : {0, 1})
        for (int wgs : {2, 1}) {
            if (pattern == 1 && wgs == 2) {   // random slots: bank conflicts and a few shared slots per instruction
                uint16_t *h2 = (uint16_t *)malloc(nval * 2);
                u32 q = 777;
                for (u64 i = 0; i < nval; ++i) { q = q * 1664525u + 1013904223u; h2[i] = (uint16_t)((q >> 10) & 1023u); }
                CHECK(hipMemcpy(slot, h2, nval * 2, hipMemcpyHostToDevice));
                free(h2);
                printf("   (random slots from here)\n");
            }
            const int grid = 256 * wgs;
            const size_t sh = (wgs == 2 ? 70 : 140) * 1024;
            MAXLDS((k_gath<0, 8>)); MAXLDS((k_gath<1, 8>)); MAXLDS((k_gath<2, 8>)); MAXLDS((k_gath<0, 4>)); MAXLDS((k_gath<0, 16>));
            float a = timeit([&] { k_gath<0, 8><<<grid, 512, sh>>>(val, slot, nblocks, out); });
            float b = timeit([&] { k_gath<1, 8><<<grid, 512, sh>>>(val, slot, nblocks, out); });
            float c = timeit([&] { k_gath<2, 8><<<grid, 512, sh>>>(val, slot, nblocks, out); });
            float d = timeit([&] { k_gath<0, 4><<<grid, 512, sh>>>(val, slot, nblocks, out); });
            float e = timeit([&] { k_gath<0, 16><<<grid, 512, sh>>>(val, slot, nblocks, out); });
            printf("== gather loop (10 B/value), %d WG/CU x 8 waves: atomic U=8 %.2f TB/s, plain rmw %.2f, no LDS %.2f, atomic U=4 %.2f, U=16 %.2f\n", wgs,
                   bytes / a / 1e9, bytes / b / 1e9, bytes / c / 1e9, bytes / d / 1e9, bytes / e / 1e9);
        }
        // 4x the workgroups (short items)
        {
            float a = timeit([&] { k_gath<0, 8><<<256 * 8, 512, 70 * 1024>>>(val, slot, nblocks, out); });
            float b = timeit([&] { k_gath<0, 8><<<256 * 32, 512, 70 * 1024>>>(val, slot, nblocks, out); });
            printf("== gather loop, 2 WG/CU resident, grid 2048 / 8192 workgroups: %.2f / %.2f TB/s\n", bytes / a / 1e9, bytes / b / 1e9);
        }
    }
    fflush(stdout);
    {
        // scatter loop: 1 Mi steps of 1 KiB codes (1 GiB), values 8 B per lane
        const u64 nsteps = 1ull << 20;
        {   // random 14-bit codes (bank conflicts as in the real pass): one 64 MiB block repeated
            const size_t hb = 64u << 20;
            uint16_t *h = (uint16_t *)malloc(hb);
            u32 q = 4242;
            for (size_t i = 0; i < hb / 2; ++i) { q = q * 1664525u + 1013904223u; h[i] = (uint16_t)((q >> 9) & 0x3fffu); }
            for (size_t off = 0; off < (size_t)nsteps * 1024; off += hb) CHECK(hipMemcpy((char *)buf + off, h, hb, hipMemcpyHostToDevice));
            free(h);
        }
        double *val = reinterpret_cast<double *>(buf2);
        MAXLDS((k_scat<4, 1>)); MAXLDS((k_scat<4, 4>)); MAXLDS((k_scat<2, 1>)); MAXLDS((k_scat<8, 1>)); MAXLDS((k_scat<4, 64>));
        const size_t sh = 132 * 1024;
        float a = timeit([&] { k_scat<4, 1><<<256, 1024, sh>>>(buf, nsteps, x, val); });
        float b = timeit([&] { k_scat<4, 4><<<256, 1024, sh>>>(buf, nsteps, x, val); });
        float c = timeit([&] { k_scat<2, 1><<<256, 1024, sh>>>(buf, nsteps, x, val); });
        float d = timeit([&] { k_scat<8, 1><<<256, 1024, sh>>>(buf, nsteps, x, val); });
        float e = timeit([&] { k_scat<4, 64><<<256, 1024, sh>>>(buf, nsteps, x, val); });
        const double rb = nsteps * 1024.0;
        printf("== scatter loop (16 B codes -> 8 LDS look-ups), 1 WG/CU x 16 waves: dense stores U=4 %.2f TB/s read + %.2f written; 1 store per 4 lanes %.2f + %.2f; "
               "dense U=2 %.2f, U=8 %.2f; 1 store per wave %.2f\n",
               rb / a / 1e9, rb / 2 / a / 1e9, rb / b / 1e9, rb / 8 / b / 1e9, rb / c / 1e9, rb / d / 1e9, rb / e / 1e9);
        float f = timeit([&] { k_scat<4, 1><<<256 * 8, 1024, sh>>>(buf, nsteps, x, val); });
        printf("== scatter loop, grid 2048 workgroups (x tile restaged 8x as often): %.2f TB/s read\n", rb / f / 1e9);
    }
    return 0;
}

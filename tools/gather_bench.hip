// tools/gather_bench.hip -- microbenchmark behind DESIGN.md's gather-rate table: how many random 8-byte
// gathers per second does an MI355X sustain when a wave streams 16-byte index packets (as the SpMV does)
// and the gathered table sits in L2 / Infinity Cache / HBM?  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void fill_idx(uint32_t *idx, uint64_t count, uint32_t table, int skew)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint64_t r = mix((i + 1) * 0x9E3779B97F4A7C15ull);
    uint32_t v = (uint32_t)(((r >> 32) * table) >> 32);
    if (skew) {  // R-MAT-like popularity: AND of two uniforms biases towards few set bits
        uint64_t r2 = mix(r + 12345);
        v &= (uint32_t)(((r2 >> 32) * table) >> 32) | (uint32_t)(r2 & (r2 >> 13));
        if (v >= table) v %= table;
    }
    idx[i] = v;
}

// window-sorted indices: inside every window of W consecutive indices the values ascend over the whole table (roughly uniformly) --
// the access order of a row-band-major SpMV whose band entries are sorted by column (the uniform family)
__global__ void fill_idx_sorted(uint32_t *idx, uint64_t count, uint32_t table, uint32_t W)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t j = (uint32_t)(i % W);
    const double frac = (double)(mix((i + 1) * 0x9E3779B97F4A7C15ull) >> 11) * (1.0 / 9007199254740992.0);
    uint64_t v = (uint64_t)(((double)j + frac) * (double)table / (double)W);
    idx[i] = (uint32_t)(v < table ? v : table - 1);
}

template <int MODE> __device__ __forceinline__ double ld(const double *p)
{
    if (MODE == 1) return __builtin_nontemporal_load(p);
    if (MODE == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

template <int UNROLL, int MODE>
__global__ void __launch_bounds__(1024) gather_k(const uint32_t *idx, uint64_t packets, const double *tab, double *out)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    const uint4 *p = reinterpret_cast<const uint4 *>(idx);
    double acc = 0.0;
    uint64_t q = tid;
    for (; q + (UNROLL - 1) * nthreads < packets; q += UNROLL * nthreads) {
        uint4 c[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) c[u] = p[q + u * nthreads];
        double x[UNROLL * 4];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            x[4 * u + 0] = ld<MODE>(tab + c[u].x); x[4 * u + 1] = ld<MODE>(tab + c[u].y);
            x[4 * u + 2] = ld<MODE>(tab + c[u].z); x[4 * u + 3] = ld<MODE>(tab + c[u].w);
        }
#pragma unroll
        for (int u = 0; u < UNROLL * 4; ++u) acc += x[u];
    }
    if (acc == 1.2345e-300) out[tid] = acc;  // keep the loads alive
}

// one workgroup per window (a row band with its y tile in LDS): workgroup b sweeps windows b, b + grid, ... one after the other, its 1024
// threads reading consecutive packets; all workgroups start together, so on a uniform graph they move over x at the same pace
__global__ void __launch_bounds__(1024) gather_bands_k(const uint32_t *idx, uint64_t n_windows, uint32_t W, const double *tab, double *out)
{
    double acc = 0.0;
    for (uint64_t w = blockIdx.x; w < n_windows; w += gridDim.x) {
        const uint4 *p = reinterpret_cast<const uint4 *>(idx + w * W);
        const uint32_t packets = W / 4;
        uint32_t q = threadIdx.x;
        for (; q + 1024 < packets; q += 2048) {
            const uint4 c0 = p[q], c1 = p[q + 1024];
            const double x0 = tab[c0.x], x1 = tab[c0.y], x2 = tab[c0.z], x3 = tab[c0.w];
            const double x4 = tab[c1.x], x5 = tab[c1.y], x6 = tab[c1.z], x7 = tab[c1.w];
            acc += ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
        }
    }
    if (acc == 1.2345e-300) out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

__global__ void stream_k(const uint4 *p, uint64_t packets, uint32_t *out)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (uint64_t q = tid; q < packets; q += nthreads) { uint4 c = p[q]; acc += c.x ^ c.y ^ c.z ^ c.w; }
    if (acc == 0x12345678u) out[tid] = acc;
}

int main(int argc, char **argv)
{
    const uint64_t count = 256ull << 20;  // 256 Mi indices = 1 GiB of index stream
    uint32_t *idx; double *tab, *out;
    CHECK(hipMalloc(&idx, count * 4));
    CHECK(hipMalloc(&out, 8ull << 20));
    const uint64_t max_tab = 256ull << 20;  // doubles (2 GiB)
    CHECK(hipMalloc(&tab, max_tab * 8));
    CHECK(hipMemset(tab, 0, max_tab * 8));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int grid = 256 * 2;

    {   // pure index stream
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) {
            CHECK(hipEventRecord(a)); stream_k<<<grid * 4, 256>>>((const uint4 *)idx, count / 4, (uint32_t *)out);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        printf("index stream only: %.3f ms  %.2f TB/s\n", best, count * 4 / best / 1e9);
    }
    if (argc > 1 && !strcmp(argv[1], "er")) {
        // Erdos-Renyi benchmark graph: x = 10 M doubles (80 MB: beyond the L2s, inside the Infinity Cache), 200 M look-ups per SpMV
        const uint32_t table = 10000000u;
        const uint32_t windows[] = {0u, 20480u, 327680u, 2621440u, 20971520u};   // entries of a band of 1 Ki / 16 Ki / 128 Ki / 1 Mi rows at 20 per row
        for (uint32_t W : windows) {
            if (W) fill_idx_sorted<<<(unsigned)((count + 255) / 256), 256>>>(idx, count, table, W);
            else fill_idx<<<(unsigned)((count + 255) / 256), 256>>>(idx, count, table, 0);
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 3; ++r) {
                CHECK(hipEventRecord(a));
                gather_k<2, 0><<<grid, 1024>>>(idx, count / 4, tab, out);
                CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
                float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
            }
            printf("x = 80 MB, look-ups %s%u: %7.1f G look-ups/s -> 200 M look-ups (one SpMV of the ER graph) take %.2f ms, index stream and sums not counted\n",
                   W ? "ascending inside windows of " : "in random order, window ", W, count / best / 1e6, 200e6 / (count / best / 1e6) * 1e-6);
            fflush(stdout);
            if (W && W <= 2621440u) {   // the same windows, ONE WORKGROUP EACH, 256 / 512 of them sweeping side by side
                for (int g : {256, 512}) {
                    const uint64_t n_windows = count / W;
                    float bb = 1e30f;
                    for (int r = 0; r < 3; ++r) {
                        CHECK(hipEventRecord(a));
                        gather_bands_k<<<g, 1024>>>(idx, n_windows, W, tab, out);
                        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
                        float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < bb) bb = ms;
                    }
                    const double done = (double)n_windows * W;
                    printf("    one workgroup per window, %d workgroups side by side: %7.1f G look-ups/s -> %.2f ms per 200 M\n", g, done / bb / 1e6, 200e6 / (done / bb / 1e6) * 1e-6);
                }
                fflush(stdout);
            }
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "sym")) {
        // Round 5, VERDICT r4 item 1 (A = A^T): a scatter unit that also adds the transposed contribution y_B[j] += x_i needs x of
        // the ROWS that stream past it -- per (row, column band) pair one x_i, the unit's rows being a sorted, sparse subset of the
        // vertex order.  tools/symmetric_stats.py counts them on C3: 41.7 M fetches per SpMV touching 13.2 M lines (3.16 doubles per
        // 128-byte line; the M-L block 3.14, M-M 12.5, L-L 1.24).  What do such fetches cost?  x = the 5.9 M vertices with an edge.
        const uint32_t table = 5903948u;
        const double per_line[] = {1.24, 3.14, 6.0, 12.5, 16.0};
        for (double d : per_line) {
            const uint32_t W = (uint32_t)(d * table / 16.0) & ~3u;   // a window = one unit's rows: W ascending fetches spread over all of x
            fill_idx_sorted<<<(unsigned)((count + 255) / 256), 256>>>(idx, count, table, W);
            CHECK(hipDeviceSynchronize());
            const uint64_t n_windows = count / W;
            float bb = 1e30f;
            for (int r = 0; r < 3; ++r) {
                CHECK(hipEventRecord(a));
                gather_bands_k<<<256, 1024>>>(idx, n_windows, W, tab, out);
                CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
                float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < bb) bb = ms;
            }
            const double done = (double)n_windows * W, rate = done / bb / 1e6;   // G fetches / s
            printf("x = 47 MB, %5.2f needed doubles per 128-byte line, one workgroup per unit, 256 side by side: %7.1f G fetches/s = %5.2f TB/s of lines touched"
                   " -> the 41.7 M x_i of one C3 SpMV take %.3f ms (4-byte index stream included)\n", d, rate, rate * 1e9 / d * 128.0 / 1e12, 41.7e6 / (rate * 1e9) * 1e3);
            fflush(stdout);
        }
        return 0;
    }
    const uint64_t tables[] = {4096, 1ull << 17, 1ull << 19, 1ull << 20, 1ull << 22, 10ull << 20, 1ull << 25, 1ull << 27, 1ull << 28};
    for (int skew = 0; skew < 2; ++skew)
        for (uint64_t t : tables) {
            fill_idx<<<(unsigned)((count + 255) / 256), 256>>>(idx, count, (uint32_t)t, skew);
            CHECK(hipDeviceSynchronize());
            float bestm[3];
            for (int mode = 0; mode < 3; ++mode) {
                float best = 1e30f;
                for (int r = 0; r < 3; ++r) {
                    CHECK(hipEventRecord(a));
                    if (mode == 0) gather_k<2, 0><<<grid, 1024>>>(idx, count / 4, tab, out);
                    if (mode == 1) gather_k<2, 1><<<grid, 1024>>>(idx, count / 4, tab, out);
                    if (mode == 2) gather_k<2, 2><<<grid, 1024>>>(idx, count / 4, tab, out);
                    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
                    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
                }
                bestm[mode] = best;
            }
            const float best = bestm[0];
            printf("skew=%d table %9.2f MB: %8.3f ms  %7.1f Ggather/s  (index stream %.2f TB/s)  nt %7.1f  sc1 %7.1f Ggather/s\n", skew, t * 8 / 1e6,
                   best, count / best / 1e6, count * 4 / best / 1e9, count / bestm[1] / 1e6, count / bestm[2] / 1e6);
            fflush(stdout);
        }
    return 0;
}

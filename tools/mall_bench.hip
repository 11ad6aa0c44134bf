// tools/mall_bench.hip -- does a buffer that is written by one kernel and read by the next stay in the 256 MiB
// Infinity Cache?  Times write(S) / read(S) pairs over a ring of S bytes for several S, optionally with an unrelated
// stream of `noise` bytes read between the two (the index stream the real kernels read next to the values).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mall_bench.hip -o /tmp/mall_bench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(1024) k_write(double2 *p, uint64_t n16, double v)
{
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += nthreads) p[i] = make_double2(v, v + i);
}
__global__ void __launch_bounds__(1024) k_read(const double2 *p, uint64_t n16, double *out)
{
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += nthreads) { double2 c = p[i]; acc += c.x + c.y; }
    if (acc == 1.2345e-300) out[0] = acc;
}

int main()
{
    const uint64_t cap = 4ull << 30;
    double2 *buf, *noise; double *out;
    CHECK(hipMalloc(&buf, cap)); CHECK(hipMalloc(&noise, cap)); CHECK(hipMalloc(&out, 4096));
    CHECK(hipMemset(buf, 0, cap)); CHECK(hipMemset(noise, 0, cap));
    hipEvent_t ev[4]; for (auto &x : ev) CHECK(hipEventCreate(&x));
    const int grid = 256 * 2;
    for (uint64_t noise_mb : {0ull, 64ull})
        for (uint64_t mb : {16ull, 32ull, 64ull, 96ull, 128ull, 192ull, 256ull, 512ull, 1024ull, 2048ull}) {
            const uint64_t S = mb << 20, n16 = S / 16, N = (noise_mb << 20) / 16;
            const int reps = (int)((8ull << 30) / S) < 4 ? 4 : (int)((8ull << 30) / S);
            float tw = 0, tr = 0, tn = 0;
            uint64_t noff = 0;
            for (int r = -2; r < reps; ++r) {
                CHECK(hipEventRecord(ev[0]));
                k_write<<<grid, 1024>>>(buf, n16, (double)r);
                CHECK(hipEventRecord(ev[1]));
                if (N) { k_read<<<grid, 1024>>>(noise + noff, N, out); noff = (noff + N) % ((cap / 16) - N); }
                CHECK(hipEventRecord(ev[2]));
                k_read<<<grid, 1024>>>(buf, n16, out);
                CHECK(hipEventRecord(ev[3]));
                CHECK(hipEventSynchronize(ev[3]));
                float a, b, c;
                CHECK(hipEventElapsedTime(&a, ev[0], ev[1])); CHECK(hipEventElapsedTime(&b, ev[1], ev[2])); CHECK(hipEventElapsedTime(&c, ev[2], ev[3]));
                if (r >= 0) { tw += a; tn += b; tr += c; }
            }
            printf("ring %5llu MB, noise %3llu MB: write %6.2f TB/s  read %6.2f TB/s  (noise read %6.2f TB/s)\n", (unsigned long long)mb,
                   (unsigned long long)noise_mb, S * (double)reps / tw / 1e9, S * (double)reps / tr / 1e9, N ? N * 16.0 * reps / tn / 1e9 : 0.0);
            fflush(stdout);
        }
    return 0;
}

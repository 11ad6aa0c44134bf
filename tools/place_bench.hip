// tools/place_bench.hip -- does the speed of WRITES into a large allocation depend on which allocation it is, and on which part of
// it?  (profiles/r4_placement.txt: the blocked SpMV's scatter pass is 30 % slower or faster by where the driver put its 0.6-1.7 GB
// value stream.)  Several allocations are kept alive together; each is timed chunk by chunk (64 MiB) with three patterns:
//   fill     streaming 16-byte stores, consecutive lanes consecutive addresses;
//   sectors  every 32-byte sector of the chunk written once, in a pseudo-random order (a multiplicative permutation): the partial-
//            line traffic of the scatter pass's plain quads;
//   read     streaming 16-byte loads.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/place_bench tools/place_bench.hip && /tmp/place_bench [MiB per allocation] [allocations]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(1024) k_fill(double2 *p, size_t n16, double v)
{
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 1024) p[i] = make_double2(v, v);
}

__global__ void __launch_bounds__(1024) k_sectors(double2 *p, unsigned n32, unsigned mult, double v)   // n32: power of two
{
    for (unsigned i = blockIdx.x * 1024 + threadIdx.x; i < n32; i += gridDim.x * 1024) {
        const unsigned s = (i * mult) & (n32 - 1);   // odd multiplier: a permutation of the sectors
        p[2 * (size_t)s] = make_double2(v, v);
        p[2 * (size_t)s + 1] = make_double2(v, v);
    }
}

__global__ void __launch_bounds__(1024) k_read(const double2 *p, size_t n16, double *out)
{
    double a = 0;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 1024) { const double2 t = p[i]; a += t.x + t.y; }
    if (a == 1.2345e-300) *out = a;
}

int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 1600;
    const int n_alloc = argc > 2 ? atoi(argv[2]) : 6;
    const size_t chunk = 64ull << 20;
    const size_t chunks = (mib << 20) / chunk;
    std::vector<char *> bufs;
    double *d_out = nullptr;
    CK(hipMalloc(&d_out, 8));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int t = 0; t < n_alloc; ++t) {
        char *p = nullptr;
        CK(hipMalloc(&p, chunks * chunk));
        CK(hipMemset(p, 0, chunks * chunk));
        bufs.push_back(p);
    }
    CK(hipDeviceSynchronize());
    auto timed = [&](auto launch) {
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(a, 0));
            launch();
            CK(hipEventRecord(b, 0));
            CK(hipEventSynchronize(b));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        return best;
    };
    for (int t = 0; t < n_alloc; ++t) {
        char *p = bufs[t];
        // whole-buffer figures first, then chunk by chunk
        const float wf = timed([&] { hipLaunchKernelGGL(k_fill, dim3(2048), dim3(1024), 0, 0, (double2 *)p, chunks * chunk / 16, 1.0); });
        const float wr = timed([&] { hipLaunchKernelGGL(k_read, dim3(2048), dim3(1024), 0, 0, (const double2 *)p, chunks * chunk / 16, d_out); });
        printf("alloc %d at %p: whole fill %.0f GB/s, read %.0f GB/s | per 64 MiB chunk, fill / sectors GB/s:", t, (void *)p, chunks * chunk / wf * 1e-6,
               chunks * chunk / wr * 1e-6);
        float smin = 1e30f, smax = 0, ssum = 0;
        for (size_t c = 0; c < chunks; ++c) {
            char *q = p + c * chunk;
            const float f = timed([&] { hipLaunchKernelGGL(k_fill, dim3(1024), dim3(1024), 0, 0, (double2 *)q, chunk / 16, 2.0); });
            const float s = timed([&] { hipLaunchKernelGGL(k_sectors, dim3(1024), dim3(1024), 0, 0, (double2 *)q, (unsigned)(chunk / 32), 2654435761u, 3.0); });
            const float gs = chunk / s * 1e-6f;
            printf(" %.0f/%.0f", chunk / f * 1e-6, gs);
            smin = gs < smin ? gs : smin;
            smax = gs > smax ? gs : smax;
            ssum += gs;
        }
        // the scatter pattern over the WHOLE buffer (beyond the caches, like the value stream)
        unsigned n32 = 1;
        while ((size_t)n32 * 2 * 32 <= chunks * chunk) n32 *= 2;
        const float ws = timed([&] { hipLaunchKernelGGL(k_sectors, dim3(2048), dim3(1024), 0, 0, (double2 *)p, n32, 2654435761u, 4.0); });
        printf(" | sectors per chunk min %.0f max %.0f mean %.0f; sectors over the first %zu MiB at once: %.0f GB/s\n", smin, smax, ssum / chunks,
               (size_t)n32 * 32 >> 20, (double)n32 * 32 / ws * 1e-6);
        fflush(stdout);
    }
    for (char *p : bufs) CK(hipFree(p));
    return 0;
}

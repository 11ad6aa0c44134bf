// tools/width_bench.hip -- does the width of a streaming load matter?  1 GiB read with 4-, 8- and 16-byte loads per lane,
// U loads in flight per lane, one 1024-thread workgroup per CU holding 128 KiB of LDS (the staged-columns kernel's shape)
// and 8 x 256-thread workgroups per CU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/width_bench.hip -o tools/_width_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint64_t u64;
extern __shared__ double lds[];
template <typename T> __device__ __forceinline__ u64 fold(const T &v);
template <> __device__ __forceinline__ u64 fold<uint32_t>(const uint32_t &v) { return v; }
template <> __device__ __forceinline__ u64 fold<uint2>(const uint2 &v) { return v.x ^ v.y; }
template <> __device__ __forceinline__ u64 fold<uint4>(const uint4 &v) { return v.x ^ v.y ^ v.z ^ v.w; }
template <typename T, int U>
__global__ void __launch_bounds__(1024) k_rd(const T *p, u64 n, double *out)
{
    if (threadIdx.x == 0) lds[0] = 0.0;
    const u64 per = n / gridDim.x;
    const T *q = p + per * blockIdx.x;
    u64 acc = 0;
    for (u64 i = threadIdx.x; i + (U - 1) * (u64)blockDim.x < per; i += (u64)U * blockDim.x) {
        T c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) c[u] = q[i + (u64)u * blockDim.x];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += fold<T>(c[u]);
    }
    if (acc == 0x123456789abcull) out[0] = (double)acc;
}
template <typename T, int U>
static void run(const void *buf, u64 bytes, double *out, int threads, int wgs, size_t ldsb, const char *name)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute((const void *)k_rd<T, U>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0));
        k_rd<T, U><<<256 * wgs, threads, ldsb>>>((const T *)buf, bytes / sizeof(T), out);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    printf("%-34s U=%d  %5.2f TB/s\n", name, U, bytes / best / 1e9);
}
int main()
{
    const u64 S = 1ull << 30;
    void *buf; double *out;
    CHECK(hipMalloc(&buf, S)); CHECK(hipMalloc(&out, 64)); CHECK(hipMemset(buf, 0x11, S));
    run<uint4, 2>(buf, S, out, 1024, 1, 128 * 1024, "16 B/lane, 1x1024thr/128K");
    run<uint4, 4>(buf, S, out, 1024, 1, 128 * 1024, "16 B/lane, 1x1024thr/128K");
    run<uint2, 2>(buf, S, out, 1024, 1, 128 * 1024, " 8 B/lane, 1x1024thr/128K");
    run<uint2, 4>(buf, S, out, 1024, 1, 128 * 1024, " 8 B/lane, 1x1024thr/128K");
    run<uint2, 8>(buf, S, out, 1024, 1, 128 * 1024, " 8 B/lane, 1x1024thr/128K");
    run<uint2, 16>(buf, S, out, 1024, 1, 128 * 1024, " 8 B/lane, 1x1024thr/128K");
    run<uint32_t, 4>(buf, S, out, 1024, 1, 128 * 1024, " 4 B/lane, 1x1024thr/128K");
    run<uint32_t, 16>(buf, S, out, 1024, 1, 128 * 1024, " 4 B/lane, 1x1024thr/128K");
    run<uint4, 2>(buf, S, out, 256, 8, 16 * 1024, "16 B/lane, 8x256thr/16K");
    run<uint2, 2>(buf, S, out, 256, 8, 16 * 1024, " 8 B/lane, 8x256thr/16K");
    run<uint2, 8>(buf, S, out, 256, 8, 16 * 1024, " 8 B/lane, 8x256thr/16K");
    run<uint32_t, 8>(buf, S, out, 256, 8, 16 * 1024, " 4 B/lane, 8x256thr/16K");
    return 0;
}

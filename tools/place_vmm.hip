// tools/place_vmm.hip -- can user space BUILD a large buffer whose writes run at the fast rate every time?  The buffer's virtual range is
// reserved once and backed by many small physical allocations (HIP's virtual-memory API), in allocation order or shuffled; beside plain
// hipMalloc.  Figure of merit: streaming fill of the whole buffer (tools/place_bench.hip: 4.8-5.8 TB/s by allocation; the blocked SpMV's
// scatter pass follows it).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/place_vmm tools/place_vmm.hip && /tmp/place_vmm [MiB] [instances] [chunk MiB]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(1024) k_fill(double2 *p, size_t n16, double v)
{
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 1024) p[i] = make_double2(v, v);
}

static hipEvent_t ea, eb;
static double fill_gbs(void *p, size_t bytes)
{
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(ea, 0));
        hipLaunchKernelGGL(k_fill, dim3(2048), dim3(1024), 0, 0, (double2 *)p, bytes / 16, 1.0 + r);
        CK(hipEventRecord(eb, 0));
        CK(hipEventSynchronize(eb));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, ea, eb));
        if (r > 0 && ms < best) best = ms;
    }
    return bytes / best * 1e-6;
}

int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 1600;
    const int inst = argc > 2 ? atoi(argv[2]) : 4;
    size_t chunk_mib = argc > 3 ? (size_t)atol(argv[3]) : 0;
    CK(hipEventCreate(&ea));
    CK(hipEventCreate(&eb));
    int dev = 0;
    CK(hipGetDevice(&dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran_min = 0, gran_rec = 0;
    CK(hipMemGetAllocationGranularity(&gran_min, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended));
    const size_t chunk = chunk_mib ? chunk_mib << 20 : gran_rec;
    printf("granularity: minimum %zu, recommended %zu bytes; chunk %zu bytes\n", gran_min, gran_rec, chunk);
    const size_t bytes = ((mib << 20) + chunk - 1) / chunk * chunk;
    const size_t n_chunks = bytes / chunk;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::mt19937 rng(12345);
    std::vector<void *> plain;
    for (int i = 0; i < inst; ++i) {
        // plain hipMalloc, kept alive
        void *p = nullptr;
        CK(hipMalloc(&p, bytes));
        plain.push_back(p);
        printf("instance %d: hipMalloc %.0f GB/s", i, fill_gbs(p, bytes));
        for (int shuffled = 0; shuffled < 2; ++shuffled) {
            // twice as many physical chunks as needed when shuffling: a random half of them, in random order
            const size_t n_phys = shuffled ? 2 * n_chunks : n_chunks;
            std::vector<hipMemGenericAllocationHandle_t> h(n_phys);
            for (size_t c = 0; c < n_phys; ++c) CK(hipMemCreate(&h[c], chunk, &prop, 0));
            std::vector<size_t> order(n_phys);
            for (size_t c = 0; c < n_phys; ++c) order[c] = c;
            if (shuffled) std::shuffle(order.begin(), order.end(), rng);
            void *va = nullptr;
            CK(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
            for (size_t c = 0; c < n_chunks; ++c) CK(hipMemMap((char *)va + c * chunk, chunk, 0, h[order[c]], 0));
            CK(hipMemSetAccess(va, bytes, &acc, 1));
            printf(" | virtual-memory API, %s: %.0f GB/s", shuffled ? "random half of 2x chunks, shuffled" : "chunks in order", fill_gbs(va, bytes));
            CK(hipMemUnmap(va, bytes));
            CK(hipMemAddressFree(va, bytes));
            for (size_t c = 0; c < n_phys; ++c) CK(hipMemRelease(h[c]));
        }
        printf("\n");
        fflush(stdout);
    }
    for (void *p : plain) CK(hipFree(p));
    return 0;
}

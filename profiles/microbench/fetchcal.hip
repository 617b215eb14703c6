// What FETCH_SIZE counts for the access widths of the overlap SpGEMM (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate
// on a known byte count in your own access pattern"): a coalesced stream of 4 GiB read as 4 / 8 / 16 bytes per lane (the row-entry stream of
// k_spgemm_direct is 8 bytes per lane), one launch each; run under `rocprofv3 --pmc FETCH_SIZE` and compare with 4 GiB.
// hipcc -O3 --offload-arch=gfx950 fetchcal.hip -o fetchcal
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <typename T>
__global__ __launch_bounds__(256) void k_stream(const T *a, uint64_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const T v = a[i];
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&v);
        for (unsigned q = 0; q < sizeof(T) / 4; ++q) acc ^= w[q];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <typename T>
void run(const char *name, const void *arr, size_t bytes, uint32_t *sink)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_stream<T>, dim3(256 * 16), dim3(256), 0, 0, (const T *)arr, (uint64_t)(bytes / sizeof(T)), sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s %zu bytes in %.3f ms = %.2f TB/s\n", name, bytes, ms, bytes / (ms * 1e-3) / 1e12);
}
int main()
{
    const size_t bytes = 4ull << 30;
    void *arr; uint32_t *sink;
    if (hipMalloc(&arr, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(arr, 1, bytes); hipMemset(sink, 0, 64); hipDeviceSynchronize();
    run<uint32_t>("k_stream<4 B per lane>", arr, bytes, sink);
    run<uint2>("k_stream<8 B per lane>", arr, bytes, sink);
    run<uint4>("k_stream<16 B per lane>", arr, bytes, sink);
    hipDeviceSynchronize();
    return 0;
}

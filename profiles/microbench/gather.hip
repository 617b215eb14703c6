// random-gather ceiling: how many random 64-byte lines per second does one MI355X deliver when nothing else limits the kernel?
// Every lane loads LB bytes (4, 8 or 16) at a random 4-byte-aligned place of an array of S bytes; the places come from a coalesced index
// stream (4 bytes per gather); U independent gathers per lane are in flight before the first is used.  Reported: gathers/s and the
// line traffic they stand for (64 B per gather when S is far beyond the caches).  Build: hipcc -O3 --offload-arch=gfx950 gather.hip -o gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
struct __attribute__((packed, aligned(4))) Quad { uint32_t a, b, c, d; };
template <int LB, int U>
__global__ __launch_bounds__(256) void k_gather(const uint32_t *idx, const uint32_t *arr, uint64_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * U;
    for (uint64_t base = ((uint64_t)blockIdx.x * blockDim.x) * U + threadIdx.x; base < n; base += stride) {
        uint32_t ix[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const uint64_t g = base + (uint64_t)u * blockDim.x; ix[u] = g < n ? idx[g] : 0u; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (LB == 4) acc += arr[ix[u]];
            else if (LB == 8) { acc += arr[ix[u]] ^ arr[ix[u] + 1]; }
            else { const Quad q = *reinterpret_cast<const Quad *>(arr + ix[u]); acc += q.a ^ q.b ^ q.c ^ q.d; }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <int LB, int U>
void run(size_t S, uint64_t n, int wg_per_cu)
{
    uint32_t *idx, *arr, *sink;
    hipMalloc(&idx, n * 4); hipMalloc(&arr, S + 64); hipMalloc(&sink, 4);
    hipMemset(arr, 1, S + 64);
    std::vector<uint32_t> h(n);
    uint64_t x = 88172645463325252ull;
    const uint64_t words = S / 4;
    for (uint64_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)(x % (words - 4)); }
    hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * wg_per_cu;
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((k_gather<LB, U>), dim3(grid), dim3(256), 0, 0, idx, arr, n, sink);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
    }
    printf("array %7.0f MB  load %2d B  %d in flight/lane  %2d WG/CU : %7.3f ms  %6.2f G gathers/s  = %5.2f TB/s of 64-B lines (+ index stream %4.2f TB/s)\n", S / 1e6, LB, U, wg_per_cu, best,
           n / (best * 1e-3) / 1e9, n * 64.0 / (best * 1e-3) / 1e12, n * 4.0 / (best * 1e-3) / 1e12);
    hipFree(idx); hipFree(arr); hipFree(sink);
}
int main()
{
    const uint64_t n = 64ull << 20;            // 67 M gathers per launch
    for (size_t S : {(size_t)57 << 20, (size_t)113 << 20, (size_t)1 << 30, (size_t)8 << 30}) {
        run<16, 4>(S, n, 8);
        run<16, 8>(S, n, 8);
        run<4, 8>(S, n, 8);
        run<16, 8>(S, n, 4);
    }
    return 0;
}

// Random aligned SEGMENTS of 64 / 128 / 256 bytes (SEG / 16 adjacent lanes x 16 bytes, one global_load_dwordx4 per lane) — does a 128-byte
// segment cost the memory system one request or two?  (dense path of spgemm_direct.hpp: a k-mer column of <= 32 partner ids is 128 bytes)
//   ALIGN == SEG: segments start at multiples of their size; ALIGN == 64 with SEG == 128: two adjacent 64-byte lines, any parity.
//   W4: the same bytes read as 4-byte words by SEG / 4 lanes (the dense path's present shape).
// hipcc -O3 --offload-arch=gfx950 gather128.hip -o gather128
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
template <int SEG, bool W4>
__global__ __launch_bounds__(256) void k_seg(const uint32_t *idx, const char *arr, uint64_t nseg, uint32_t align, uint32_t *sink)
{
    constexpr int LPS = W4 ? SEG / 4 : SEG / 16;      // lanes per segment
    uint32_t acc = 0;
    const uint32_t sub = threadIdx.x & (LPS - 1);
    const uint64_t per_block = 256 / LPS;
    for (uint64_t l0 = (uint64_t)blockIdx.x * per_block; l0 < nseg; l0 += (uint64_t)gridDim.x * per_block) {
        const uint64_t l = l0 + threadIdx.x / LPS;
        if (l >= nseg) continue;
        const char *q = arr + (uint64_t)idx[l] * align;
        if (W4) acc += reinterpret_cast<const uint32_t *>(q)[sub];
        else { const uint4 v = reinterpret_cast<const uint4 *>(q)[sub]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <int SEG, bool W4>
void run(const char *name, size_t S, uint64_t n, uint32_t align, int wg_per_cu)
{
    uint32_t *idx, *sink; char *arr;
    hipMalloc(&idx, n * 4); hipMalloc(&arr, S + 512); hipMalloc(&sink, 4);
    hipMemset(arr, 1, S + 512);
    std::vector<uint32_t> h(n);
    uint64_t x = 88172645463325252ull;
    const uint64_t places = (S - SEG) / align;
    for (uint64_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)(x % places); }
    hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((k_seg<SEG, W4>), dim3(256 * wg_per_cu), dim3(256), 0, 0, idx, arr, n, align, sink);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
    }
    printf("array %6.0f MB  %-18s align %3u  %2d WG/CU : %7.3f ms  %6.2f G segments/s  %5.2f TB/s\n", S / 1e6, name, align, wg_per_cu, best, n / (best * 1e-3) / 1e9, n * (double)SEG / (best * 1e-3) / 1e12);
    fflush(stdout);
    hipFree(idx); hipFree(arr); hipFree(sink);
}
int main()
{
    const uint64_t n = 32ull << 20;
    for (size_t S : {(size_t)3 << 30, (size_t)12 << 30}) {
        run<64, false>("64 B, 4 x 16", S, n, 64, 8);
        run<128, false>("128 B, 8 x 16", S, n, 128, 8);
        run<128, false>("128 B, 8 x 16", S, n, 64, 8);
        run<256, false>("256 B, 16 x 16", S, n, 256, 8);
        run<64, true>("64 B, 16 x 4", S, n, 64, 8);
        run<128, true>("128 B, 32 x 4", S, n, 128, 8);
        run<128, true>("128 B, 32 x 4", S, n, 32, 8);
        run<128, false>("128 B, 8 x 16", S, n, 128, 16);
    }
    return 0;
}

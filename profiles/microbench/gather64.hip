// How should a kernel read random, aligned 64-byte lines (the padded k-mer columns of the plan-free SpGEMM, spgemm_direct.hpp)?
//   L4x16 : four adjacent lanes read one line, 16 bytes each (one global_load_dwordx4 per lane; 16 lines per wave-instruction)
//   L2x32 : two adjacent lanes read one line, 32 bytes each (two loads per lane)
//   L1x64 : one lane reads the whole line (four loads per lane; 64 lines per wave-instruction)
//   L1x32 : one lane reads the first 32 bytes of the line (two loads per lane)
// Line indices come from a coalesced stream (4 bytes per line).  Reported: lines/s.  hipcc -O3 --offload-arch=gfx950 gather64.hip -o gather64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
template <int LPL, int BPL>      // lanes per line, bytes per lane
__global__ __launch_bounds__(256) void k_lines(const uint32_t *idx, const uint4 *arr, uint64_t nlines, uint32_t *sink)
{
    uint32_t acc = 0;
    const uint32_t sub = threadIdx.x & (LPL - 1);
    const uint64_t lines_per_block = 256 / LPL;
    for (uint64_t l0 = (uint64_t)blockIdx.x * lines_per_block; l0 < nlines; l0 += (uint64_t)gridDim.x * lines_per_block) {
        const uint64_t l = l0 + threadIdx.x / LPL;
        if (l >= nlines) continue;
        const uint4 *q = arr + (uint64_t)idx[l] * 4 + sub * (BPL / 16);
#pragma unroll
        for (int u = 0; u < BPL / 16; ++u) { const uint4 v = q[u]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <int LPL, int BPL>
void run(const char *name, size_t S, uint64_t n, int wg_per_cu)
{
    uint32_t *idx, *sink; uint4 *arr;
    hipMalloc(&idx, n * 4); hipMalloc(&arr, S + 64); hipMalloc(&sink, 4);
    hipMemset(arr, 1, S + 64);
    std::vector<uint32_t> h(n);
    uint64_t x = 88172645463325252ull;
    const uint64_t lines = S / 64;
    for (uint64_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)(x % lines); }
    hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((k_lines<LPL, BPL>), dim3(256 * wg_per_cu), dim3(256), 0, 0, idx, arr, n, sink);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
    }
    printf("array %7.0f MB  %s  %2d WG/CU : %7.3f ms  %6.2f G lines/s  (%5.2f TB/s if every line is 64 B from memory)\n", S / 1e6, name, wg_per_cu, best, n / (best * 1e-3) / 1e9, n * 64.0 / (best * 1e-3) / 1e12);
    hipFree(idx); hipFree(arr); hipFree(sink);
}
int main()
{
    const uint64_t n = 32ull << 20;
    for (size_t S : {(size_t)350 << 20, (size_t)12 << 30}) {
        run<4, 16>("L4x16", S, n, 8);
        run<2, 32>("L2x32", S, n, 8);
        run<1, 64>("L1x64", S, n, 8);
        run<1, 32>("L1x32", S, n, 8);
        run<4, 16>("L4x16", S, n, 16);
        run<1, 64>("L1x64", S, n, 16);
    }
    return 0;
}

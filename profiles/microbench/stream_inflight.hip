// Streaming READ of a 16 GB array by persistent workgroups that hold a bounded number of bytes in flight per CU — what the bucket count (k_msd_count: one
// 1024-lane workgroup per CU, 96 KB requested a bucket ahead) and the partition scatters (a 128 KB tile per CU) can reach at all.
//   PRE = false: a workgroup requests its chunk (KPT x 8 bytes per lane), waits, takes the next one (what two workgroups without prefetch do);
//   PRE = true : the NEXT chunk is requested before this one is consumed (register double buffer: k_msd_count's scheme).
// chunks are dealt round-robin (chunk c + i * grid): all workgroups together stream the array front to back.
// hipcc -O3 --offload-arch=gfx950 stream_inflight.hip -o stream_inflight
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int THREADS, int KPT, bool PRE, bool V4>
__global__ __launch_bounds__(THREADS) void k_stream(const uint64_t *arr, uint64_t nchunks, uint64_t *sink)
{
    constexpr uint64_t CH = (uint64_t)THREADS * KPT;      // words per chunk
    uint64_t acc = 0;
    uint64_t cur[KPT], nxt[KPT];
    auto load = [&](uint64_t c, uint64_t (&r)[KPT]) {
        const uint64_t *p = arr + c * CH;
        if (V4) {
#pragma unroll
            for (int u = 0; u < KPT; u += 2) { const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(p)[(uint64_t)(u >> 1) * THREADS + threadIdx.x]; r[u] = v.x; r[u + 1] = v.y; }
        } else {
#pragma unroll
            for (int u = 0; u < KPT; ++u) r[u] = p[(uint64_t)u * THREADS + threadIdx.x];
        }
    };
    uint64_t c = blockIdx.x;
    if (PRE && c < nchunks) load(c, cur);
    for (; c < nchunks; c += gridDim.x) {
        if (PRE) { if (c + gridDim.x < nchunks) load(c + gridDim.x, nxt); }
        else load(c, cur);
#pragma unroll
        for (int u = 0; u < KPT; ++u) acc ^= cur[u];
        __syncthreads();      // (the kernels this stands for meet at a barrier per chunk)
        if (PRE) {
#pragma unroll
            for (int u = 0; u < KPT; ++u) cur[u] = nxt[u];
        }
    }
    if (acc == 0x123456789ull) sink[0] = acc;
}
template <int THREADS, int KPT, bool PRE, bool V4>
void run(const uint64_t *arr, uint64_t words, int wg_per_cu, uint64_t *sink)
{
    const uint64_t CH = (uint64_t)THREADS * KPT, nchunks = words / CH;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_stream<THREADS, KPT, PRE, V4>), dim3(256 * wg_per_cu), dim3(THREADS), 0, 0, arr, nchunks, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    const double inflight = (double)CH * 8 * wg_per_cu / 1024.0;
    printf("%4d lanes x %2d words%s, %d workgroup(s) per CU, %s: %6.0f KB requested per CU at a time  %7.3f ms  %6.2f TB/s\n", THREADS, KPT, V4 ? " (16-byte loads)" : "", wg_per_cu,
           PRE ? "next chunk requested before this one is consumed" : "request, wait, consume", inflight, best, (double)nchunks * CH * 8 / best / 1e9);
}
int main()
{
    const uint64_t words = 2ull << 30;      // 16 GB
    uint64_t *arr, *sink;
    if (hipMalloc(&arr, words * 8) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(arr, 1, words * 8);
    hipDeviceSynchronize();
    run<1024, 4, false, false>(arr, words, 1, sink);
    run<1024, 8, false, false>(arr, words, 1, sink);
    run<1024, 12, false, false>(arr, words, 1, sink);
    run<1024, 16, false, false>(arr, words, 1, sink);
    run<1024, 8, false, false>(arr, words, 2, sink);
    run<1024, 16, false, false>(arr, words, 2, sink);
    run<1024, 16, false, true>(arr, words, 2, sink);
    run<1024, 12, true, false>(arr, words, 1, sink);
    run<1024, 12, true, true>(arr, words, 1, sink);
    run<1024, 16, true, false>(arr, words, 1, sink);
    run<1024, 8, true, false>(arr, words, 2, sink);
    run<512, 16, false, false>(arr, words, 2, sink);
    run<512, 16, false, false>(arr, words, 4, sink);
    run<256, 16, false, false>(arr, words, 8, sink);
    run<256, 32, false, true>(arr, words, 8, sink);
    return 0;
}

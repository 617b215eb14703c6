// residency probe: how many 128-thread workgroups with L bytes of dynamic LDS does a CU hold at once?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int NV>
__global__ __launch_bounds__(128) void k(unsigned long long *t0, unsigned *hw, float *sink, int spin)
{
    extern __shared__ unsigned sm[];
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = threadIdx.x * 0.5f + i;
    unsigned long long t = wall_clock64();
    if (threadIdx.x == 0) { t0[blockIdx.x] = t; unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id)); hw[blockIdx.x] = id; }
    sm[threadIdx.x] = threadIdx.x;
    while (wall_clock64() - t < (unsigned long long)spin) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = v[i] * 1.0001f + 0.5f;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i];
    if (s == 123.456f) sink[0] = s + sm[(threadIdx.x + 1) & 127];
}
template <int NV>
void run(int lds, int wgs_per_cu)
{
    int cus = 256, n = cus * wgs_per_cu;
    unsigned long long *t0; unsigned *hw; float *sink;
    hipMalloc(&t0, n * 8); hipMalloc(&hw, n * 4); hipMalloc(&sink, 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<NV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k<NV>, dim3(n), dim3(128), lds, 0, t0, hw, sink, 3000);   // 30 us at 100 MHz
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(n); hipMemcpy(h.data(), t0, n * 8, hipMemcpyDeviceToHost);
    unsigned long long mn = *std::min_element(h.begin(), h.end());
    int early = 0; for (auto x : h) early += (x - mn) < 1500;    // started within 15 us of the first
    hipFuncAttributes a; hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k<NV>));
    printf("NV=%d regs=%d lds=%6d launched %2d/CU -> resident at once %.2f/CU\n", NV, a.numRegs, lds, wgs_per_cu, early / 256.0);
    hipFree(t0); hipFree(hw); hipFree(sink);
}
int main()
{
    for (int lds : {1024, 5376, 10368, 10752, 20608}) run<8>(lds, 16);
    for (int lds : {1024, 10368}) run<60>(lds, 16);
    return 0;
}

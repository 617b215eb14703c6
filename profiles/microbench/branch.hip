// what does an instruction cost a wave, and what does a taken branch cost?  bodies of 8 / 64 independent VALU instructions per loop trip,
// and a body with extra taken forward branches
#include <hip/hip_runtime.h>
#include <cstdio>
#define V8 "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n v_add_u32 %4, %4, 1\n v_add_u32 %5, %5, 1\n v_add_u32 %6, %6, 1\n v_add_u32 %7, %7, 1\n"
#define OPS : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long *out, int iters)
{
    unsigned v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, v4 = 4, v5 = 5, v6 = 6, v7 = 7;
    unsigned long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile(V8 OPS);
        if (MODE == 1) asm volatile(V8 V8 V8 V8 V8 V8 V8 V8 OPS);
        if (MODE == 2)     // 64 instructions with 8 taken forward branches (each skips one instruction)
            asm volatile(V8 "s_branch 1f\n v_add_u32 %0, %0, 1\n1:\n" V8 "s_branch 2f\n v_add_u32 %0, %0, 1\n2:\n" V8 "s_branch 3f\n v_add_u32 %0, %0, 1\n3:\n" V8 "s_branch 4f\n v_add_u32 %0, %0, 1\n4:\n"
                         V8 "s_branch 5f\n v_add_u32 %0, %0, 1\n5:\n" V8 "s_branch 6f\n v_add_u32 %0, %0, 1\n6:\n" V8 "s_branch 7f\n v_add_u32 %0, %0, 1\n7:\n" V8 "s_branch 8f\n v_add_u32 %0, %0, 1\n8:\n" OPS);
        if (MODE == 3)     // 64 instructions with 8 NOT-taken conditional branches (execz with exec != 0)
            asm volatile(V8 "s_cbranch_execz 1f\n" V8 "s_cbranch_execz 1f\n" V8 "s_cbranch_execz 1f\n" V8 "s_cbranch_execz 1f\n" V8 "s_cbranch_execz 1f\n" V8 "s_cbranch_execz 1f\n" V8 "s_cbranch_execz 1f\n" V8 "s_cbranch_execz 1f\n1:\n" OPS);
    }
    unsigned long long c1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = c1 - c0;
    if (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 == 0xdeadbeef) out[1] = 1;
}
template <int MODE>
void run(const char *name)
{
    unsigned long long *d, h[2];
    (void)hipMalloc(&d, 16);
    for (int w : {1, 2, 4, 8}) {
        const int iters = 5000;
        hipLaunchKernelGGL(k<MODE>, dim3(256 * w), dim3(256), 0, 0, d, iters);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%-44s %d waves/SIMD: %.1f cycles per loop trip\n", name, w, (double)h[0] / iters);
    }
    (void)hipFree(d);
}
int main()
{
    run<0>("8 VALU + loop branch");
    run<1>("64 VALU + loop branch");
    run<2>("64 VALU + 8 taken branches + loop branch");
    run<3>("64 VALU + 8 untaken branches + loop branch");
    return 0;
}

// instruction issue-rate probe: wave-instructions per cycle per SIMD for VALU / SALU / mixed / branchy code at 1..8 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long *out, int iters)
{
    unsigned v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, v4 = 4, v5 = 5, v6 = 6, v7 = 7;
    unsigned s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3, s4 = 4, s5 = 5, s6 = 6, s7 = 7;
    unsigned long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0 || MODE == 2) {
            asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n v_add_u32 %4, %4, 1\n v_add_u32 %5, %5, 1\n v_add_u32 %6, %6, 1\n v_add_u32 %7, %7, 1"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
        }
        if (MODE == 1 || MODE == 2) {
            asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1"
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) :: "scc");
        }
        if (MODE == 3) {   // dependent chain: v -> v -> v
            asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1" : "+v"(v0));
        }
        if (MODE == 4) {   // 64-bit adds
            unsigned long long a = ((unsigned long long)v1 << 32) | v0;
            asm volatile("v_lshl_add_u64 %0, %0, 0, 1\n v_lshl_add_u64 %0, %0, 0, 1\n v_lshl_add_u64 %0, %0, 0, 1\n v_lshl_add_u64 %0, %0, 0, 1\n v_lshl_add_u64 %0, %0, 0, 1\n v_lshl_add_u64 %0, %0, 0, 1\n v_lshl_add_u64 %0, %0, 0, 1\n v_lshl_add_u64 %0, %0, 0, 1" : "+v"(a));
            v0 = (unsigned)a; v1 = (unsigned)(a >> 32);
        }
    }
    unsigned long long c1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = c1 - c0;
    if (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 == 0xdeadbeef) out[1] = 1;
}
template <int MODE>
void run(const char *name, int per_iter)
{
    unsigned long long *d, h[2];
    hipMalloc(&d, 16);
    for (int wg_per_cu : {1, 2, 4, 8}) {      // 256-thread WGs = 1 wave per SIMD each
        const int iters = 20000;
        hipLaunchKernelGGL(k<MODE>, dim3(256 * wg_per_cu), dim3(256), 0, 0, d, iters);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        double cyc_per_instr_wave = (double)h[0] / ((double)iters * per_iter);
        printf("%-22s %d waves/SIMD: %.2f cycles per instruction per wave -> %.2f instr/cycle/SIMD\n", name, wg_per_cu, cyc_per_instr_wave, wg_per_cu / cyc_per_instr_wave);
    }
    hipFree(d);
}
int main()
{
    run<0>("VALU independent", 8);
    run<1>("SALU independent", 8);
    run<2>("VALU+SALU mixed", 16);
    run<3>("VALU dependent chain", 8);
    run<4>("VALU 64-bit add chain", 8);
    return 0;
}

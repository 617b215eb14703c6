#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
__global__ void k(unsigned long long *out, int spin_wall)
{
    unsigned long long w0 = wall_clock64(), c0 = clock64();
    float v = threadIdx.x;
    while (wall_clock64() - w0 < (unsigned long long)spin_wall) v = v * 1.0001f + 0.5f;
    unsigned long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = w1 - w0; out[1] = c1 - c0; out[2] = (unsigned long long)v; }
}
int main()
{
    unsigned long long *d, h[3];
    hipMalloc(&d, 24);
    int spins[] = {2000, 20000, 200000, 2000000};   // 20 us, 200 us, 2 ms, 20 ms of wall clock at 100 MHz
    for (int rep = 0; rep < 2; ++rep)
    for (int s : spins) {
        usleep(200000);
        hipLaunchKernelGGL(k, dim3(256 * 8), dim3(128), 0, 0, d, s);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("after idle: spin %8d wall ticks -> shader/wall = %.2f -> sclk ~ %.0f MHz\n", s, (double)h[1] / h[0], 100.0 * h[1] / h[0]);
    }
    // back-to-back short kernels (like the bench loop): 300 launches of 150 us
    for (int i = 0; i < 300; ++i) hipLaunchKernelGGL(k, dim3(256 * 8), dim3(128), 0, 0, d, 15000);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("after 300 back-to-back 150-us kernels: sclk ~ %.0f MHz\n", 100.0 * h[1] / h[0]);
    return 0;
}

// Issue cadence of dependent / independent fp64 multiplies and adds on one wavefront (alone on its SIMD): how far apart must
// two dependent v_mul_f64 / v_add_f64 be for the vector unit to stay busy?  hipcc --offload-arch=gfx950 -O2 -o /tmp/dp dp_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH> __global__ void chains(double *out, long long *cyc, double a, double b)
{
    double x[CH];
    for (int c = 0; c < CH; ++c) x[c] = a + c + threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 4096; ++it) {
        #pragma unroll
        for (int u = 0; u < 8; ++u) {
            #pragma unroll
            for (int c = 0; c < CH; ++c) { x[c] = x[c] * b; asm volatile("" : "+v"(x[c])); x[c] = x[c] + a; asm volatile("" : "+v"(x[c])); }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int c = 0; c < CH; ++c) s += x[c];
    out[threadIdx.x + blockIdx.x * 64] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CH> void run(int waves_per_simd, double *out, long long *cyc)
{
    // one block of 64 * 4 * waves threads on one CU -> `waves` wavefronts per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(chains<CH>, dim3(1), dim3(64 * 4 * waves_per_simd), 0, 0, out, cyc, 1e-9, 1.0000001);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(chains<CH>, dim3(1), dim3(64 * 4 * waves_per_simd), 0, 0, out, cyc, 1e-9, 1.0000001);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = 4096.0 * 8 * CH * 2;
    printf("chains %d waves/SIMD %d: %.2f clocks per fp64 instruction of one wavefront (%.0f instr; the launch %.1f us = %.2f ns per instruction)\n", CH, waves_per_simd, (double)c / n, n, ms * 1e3, ms * 1e6 / n);
}
int main()
{
    double *out; long long *cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 1024);
    for (int w = 1; w <= 4; ++w) { run<1>(w, out, cyc); run<2>(w, out, cyc); run<3>(w, out, cyc); run<4>(w, out, cyc); run<8>(w, out, cyc); }
    return 0;
}

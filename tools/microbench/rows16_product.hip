// rows16_product() of mcall.hip (the reduce-scatter over 16 rows of per-lane factors: v_permlane32_swap, v_permlane16_swap, DPP moves with a
// bank mask) checked by itself: row r, lane l carries the factor 1 + (r * 64 + l) / 4096 with exponent r * 64 + l; lane l must end up
// with the product / sum over the 64 lanes of row l >> 2.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/rows16 tools/microbench/rows16_product.hip && /tmp/rows16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
namespace bcfgpu {
__device__ __forceinline__ double frexp_mant(double x) { return __builtin_amdgcn_frexp_mant(x); }
__device__ __forceinline__ int frexp_exp(double x) { return __builtin_amdgcn_frexp_exp(x); }
template <int CTRL> __device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const uint32_t lo = (uint32_t)dpp_i32<CTRL>((int)(uint32_t)b), hi = (uint32_t)dpp_i32<CTRL>((int)(uint32_t)(b >> 32));
    return __builtin_bit_cast(double, (unsigned long long)hi << 32 | lo);
}
#include "../../bcftools_amd/csrc/mcall_rows16.h"
}
__global__ void k(double *om, int *oe)
{
    double m[16]; int e[16];
    for (int r = 0; r < 16; ++r) { m[r] = (1.0 + (r * 64 + threadIdx.x) / 4096.0) * 0.5; e[r] = r * 64 + threadIdx.x; }
    bcfgpu::rows16_product(m, e);
    om[threadIdx.x] = m[0]; oe[threadIdx.x] = e[0];
}
int main()
{
    double *dm; int *de; CK(hipMalloc(&dm, 64 * 8)); CK(hipMalloc(&de, 64 * 4));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dm, de);
    double hm[64]; int he[64];
    CK(hipMemcpy(hm, dm, sizeof hm, hipMemcpyDeviceToHost)); CK(hipMemcpy(he, de, sizeof he, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int r = l >> 2;
        double lg = 0; long es = 0;
        for (int j = 0; j < 64; ++j) { lg += log2((1.0 + (r * 64 + j) / 4096.0) * 0.5); es += r * 64 + j; }
        const double got = log2(hm[l]) + he[l];
        if (fabs(got - (lg + es)) > 1e-9) { if (bad < 8) printf("lane %d row %d: got %.12f want %.12f\n", l, r, got, lg + es); ++bad; }
    }
    printf(bad ? "rows16_product: %d lanes WRONG\n" : "rows16_product: ok\n", bad);
    return bad != 0;
}

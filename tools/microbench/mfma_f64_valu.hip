// Do v_mfma_f64_16x16x4_f64 and the vector ALU's own instructions run beside one another on a SIMD of gfx950, or do they take turns?
// (The fp64 matrix peak of the part equals its vector fp64 FMA peak, which suggests one set of fp64 units: mcall_kernel's subset scan
// is matrix instructions + vector fp64 / integer instructions)
// One workgroup of eight wavefronts per CU: wavefronts 0-3 (one per SIMD) run instruction stream X, wavefronts 4-7 stream Y.
//   X = Y = matrix        -> t_mm   (two matrix streams a SIMD)
//   X = Y = vector        -> t_vv
//   X = matrix, Y = vector -> t_mv : max(t_mm, t_vv) / 2 if they run side by side, (t_mm + t_vv) / 2 if they take turns
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_valu_bin mfma_f64_valu.hip && ./mfma_f64_valu_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double d4_t __attribute__((ext_vector_type(4)));

// 8 matrix instructions on four independent accumulators (no instruction waits for the one before it)
__device__ __forceinline__ void matrix8(d4_t &c0, d4_t &c1, d4_t &c2, d4_t &c3, double a, double b)
{
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
}

// VK: 0 = v_mul_f64, 1 = v_fma_f64, 2 = v_add_u32, 3 = v_cndmask_b32 (64 independent instructions a round)
template <int VK>
__device__ __forceinline__ void vector64(double (&d)[8], unsigned (&u)[8], double dc, unsigned uc)
{
    #pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (VK == 0) asm volatile("v_mul_f64 %0, %0, %8\nv_mul_f64 %1, %1, %8\nv_mul_f64 %2, %2, %8\nv_mul_f64 %3, %3, %8\nv_mul_f64 %4, %4, %8\nv_mul_f64 %5, %5, %8\nv_mul_f64 %6, %6, %8\nv_mul_f64 %7, %7, %8\n"
                                  : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(dc));
        if (VK == 1) asm volatile("v_fma_f64 %0, %0, %8, %8\nv_fma_f64 %1, %1, %8, %8\nv_fma_f64 %2, %2, %8, %8\nv_fma_f64 %3, %3, %8, %8\nv_fma_f64 %4, %4, %8, %8\nv_fma_f64 %5, %5, %8, %8\nv_fma_f64 %6, %6, %8, %8\nv_fma_f64 %7, %7, %8, %8\n"
                                  : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(dc));
        if (VK == 2) asm volatile("v_add_u32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_add_u32 %2, %2, %8\nv_add_u32 %3, %3, %8\nv_add_u32 %4, %4, %8\nv_add_u32 %5, %5, %8\nv_add_u32 %6, %6, %8\nv_add_u32 %7, %7, %8\n"
                                  : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(uc));
        if (VK == 3) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n"
                                  : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(uc) : "vcc");
    }
}

// MODE: 0 = every wavefront the matrix stream, 1 = every wavefront the vector stream, 2 = wavefronts 0-3 matrix, 4-7 vector
template <int MODE, int VK>
__global__ __launch_bounds__(512) void k(double *out, int reps, double seed)
{
    const int wave = threadIdx.x >> 6;
    const bool matrix = MODE == 0 || (MODE == 2 && wave < 4);
    d4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double d[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { d[i] = seed + i + threadIdx.x; u[i] = (unsigned)(i * 7 + threadIdx.x); }
    const double a = seed * 1e-3, b = 1.0 + seed * 1e-9;
    if (matrix) for (int r = 0; r < reps; ++r) matrix8(c0, c1, c2, c3, a, b);
    else for (int r = 0; r < reps; ++r) vector64<VK>(d, u, b, 3u);
    double s = c0[0] + c1[1] + c2[2] + c3[3];
    for (int i = 0; i < 8; ++i) s += d[i] + (double)u[i];
    if (s == 123.456) out[threadIdx.x] = s;                 // keeps the work alive
}

template <int MODE, int VK>
static double run(double *out, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE, VK>), dim3(256), dim3(512), 0, 0, out, reps / 8, 1.5);      // warm-up
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, VK>), dim3(256), dim3(512), 0, 0, out, reps, 1.5);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

template <int VK>
static void row(const char *name, double *out, int reps, double t_mm)
{
    const double t_vv = run<1, VK>(out, reps), t_mv = run<2, VK>(out, reps);
    printf("%-14s vector alone %.3f ms, matrix + vector %.3f ms; side by side would be %.3f, taking turns %.3f\n", name, t_vv, t_mv,
           (t_mm > t_vv ? t_mm : t_vv) / 2, (t_mm + t_vv) / 2);
}

int main()
{
    double *out; CK(hipMalloc(&out, 4096));
    const int reps = 20000;                                   // x 8 matrix instructions or x 64 vector instructions a wavefront
    const double t_mm = run<0, 0>(out, reps);
    printf("one workgroup of 8 wavefronts on each of 256 CUs (two wavefronts a SIMD), %d rounds of 8 v_mfma_f64_16x16x4_f64 or 64 vector instructions\n", reps);
    printf("matrix alone %.3f ms = %.1f cycles a matrix instruction and SIMD at 2.4 GHz\n", t_mm, t_mm * 1e-3 * 2.4e9 / (2.0 * reps * 8));
    row<0>("v_mul_f64", out, reps, t_mm);
    row<1>("v_fma_f64", out, reps, t_mm);
    row<2>("v_add_u32", out, reps, t_mm);
    row<3>("v_cndmask_b32", out, reps, t_mm);
    return 0;
}

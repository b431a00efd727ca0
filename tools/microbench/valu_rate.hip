// Issue rate of plain vector instructions on one SIMD as a function of the wavefronts resident on it: what bounds a kernel whose
// cost is its instruction count (glfgen_kernel).  Every wavefront runs REPS x 64 independent instructions of one kind
// on eight registers; the grid puts `w` wavefronts on every SIMD of the chip.  Prints cycles per wave-instruction and SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define OP8_32(INS) \
    asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" \
                 INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c))
#define OP8_3(INS) \
    asm volatile(INS " %0, %0, %8, %8\n" INS " %1, %1, %8, %8\n" INS " %2, %2, %8, %8\n" INS " %3, %3, %8, %8\n" \
                 INS " %4, %4, %8, %8\n" INS " %5, %5, %8, %8\n" INS " %6, %6, %8, %8\n" INS " %7, %7, %8, %8\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c))
#define OP8_RAW(PRE, INS, POST) \
    asm volatile(PRE INS " %0, %0, %8" POST "\n" PRE INS " %1, %1, %8" POST "\n" PRE INS " %2, %2, %8" POST "\n" PRE INS " %3, %3, %8" POST "\n" \
                 PRE INS " %4, %4, %8" POST "\n" PRE INS " %5, %5, %8" POST "\n" PRE INS " %6, %6, %8" POST "\n" PRE INS " %7, %7, %8" POST "\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s10", "s11")
#define OP8_64C(INS) \
    asm volatile(INS " %0, %8, %0\n" INS " %1, %8, %1\n" INS " %2, %8, %2\n" INS " %3, %8, %3\n" \
                 INS " %4, %8, %4\n" INS " %5, %8, %5\n" INS " %6, %8, %6\n" INS " %7, %8, %7\n" \
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(c))
#define OP8_FMT(F0, F1, F2, F3, F4, F5, F6, F7) \
    asm volatile(F0 "\n" F1 "\n" F2 "\n" F3 "\n" F4 "\n" F5 "\n" F6 "\n" F7 "\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) \
                 : "vcc", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "s25")
#define OP8_64(INS) \
    asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" \
                 INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n" \
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dc))

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(unsigned *out, int reps, unsigned seed)
{
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, c = seed | 1;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7, dc = 1.0 + 1e-9 * seed;
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (KIND == 0) OP8_32("v_add_u32");
            if (KIND == 1) OP8_32("v_xor_b32");
            if (KIND == 2) OP8_32("v_mul_lo_u32");
            if (KIND == 3) OP8_64("v_add_f64");
            if (KIND == 4) OP8_64("v_mul_f64");
            if (KIND == 5) OP8_32("v_lshlrev_b32");
            if (KIND == 6) OP8_32("v_bcnt_u32_b32");
            if (KIND == 7) OP8_32("v_max_u32");
            if (KIND == 8) OP8_32("v_add_f32");
            if (KIND == 10) OP8_32("v_mul_u32_u24");
            if (KIND == 11) OP8_32("v_and_b32");
            if (KIND == 12) OP8_32("v_or_b32");
            if (KIND == 13) OP8_32("v_sub_u32");
            if (KIND == 14) OP8_32("v_lshrrev_b32");
            if (KIND == 15) OP8_32("v_min_u32");
            if (KIND == 16) OP8_RAW("", "v_cndmask_b32", ", vcc");
            if (KIND == 17) OP8_RAW("v_cmp_lt_u32 vcc, %0, %8\n", "v_cndmask_b32", ", vcc");          // two instructions
            if (KIND == 41) OP8_RAW("", "v_cndmask_b32_e64", ", s[10:11]");
            if (KIND == 42) OP8_RAW("v_cmp_lt_u32_e64 s[10:11], %0, %8\n", "v_cndmask_b32_e64", ", s[10:11]");
            if (KIND == 43) OP8_RAW("v_cmp_lt_u32_e32 vcc, %0, %8\n", "v_add_u32", "");                          // compare + add
            if (KIND == 45) OP8_RAW("", "v_add_u32", "\n s_nop 0");                                           // add + s_nop
            if (KIND == 46) OP8_RAW("s_mov_b64 vcc, exec\n", "v_cndmask_b32", ", vcc");
            if (KIND == 50) OP8_FMT("v_cmp_lt_u32_e32 vcc, %0, %8", "v_cmp_lt_u32_e32 vcc, %1, %8", "v_cmp_lt_u32_e32 vcc, %2, %8", "v_cmp_lt_u32_e32 vcc, %3, %8",
                                    "v_cmp_lt_u32_e32 vcc, %4, %8", "v_cmp_lt_u32_e32 vcc, %5, %8", "v_cmp_lt_u32_e32 vcc, %6, %8", "v_cmp_lt_u32_e32 vcc, %7, %8");
            if (KIND == 51) OP8_FMT("v_cmp_lt_u32_e64 s[10:11], %0, %8", "v_cmp_lt_u32_e64 s[10:11], %1, %8", "v_cmp_lt_u32_e64 s[10:11], %2, %8", "v_cmp_lt_u32_e64 s[10:11], %3, %8",
                                    "v_cmp_lt_u32_e64 s[10:11], %4, %8", "v_cmp_lt_u32_e64 s[10:11], %5, %8", "v_cmp_lt_u32_e64 s[10:11], %6, %8", "v_cmp_lt_u32_e64 s[10:11], %7, %8");
            if (KIND == 52) OP8_FMT("v_cmp_lt_u32_e64 s[10:11], %0, %8", "v_cmp_lt_u32_e64 s[12:13], %1, %8", "v_cmp_lt_u32_e64 s[14:15], %2, %8", "v_cmp_lt_u32_e64 s[16:17], %3, %8",
                                    "v_cmp_lt_u32_e64 s[18:19], %4, %8", "v_cmp_lt_u32_e64 s[20:21], %5, %8", "v_cmp_lt_u32_e64 s[22:23], %6, %8", "v_cmp_lt_u32_e64 s[24:25], %7, %8");
            if (KIND == 53) OP8_FMT("v_add_co_u32_e32 %0, vcc, %0, %8", "v_add_co_u32_e32 %1, vcc, %1, %8", "v_add_co_u32_e32 %2, vcc, %2, %8", "v_add_co_u32_e32 %3, vcc, %3, %8",
                                    "v_add_co_u32_e32 %4, vcc, %4, %8", "v_add_co_u32_e32 %5, vcc, %5, %8", "v_add_co_u32_e32 %6, vcc, %6, %8", "v_add_co_u32_e32 %7, vcc, %7, %8");
            if (KIND == 54) OP8_FMT("v_add_co_u32_e32 %0, vcc, %0, %8", "v_addc_co_u32_e32 %1, vcc, %1, %8, vcc", "v_add_co_u32_e32 %2, vcc, %2, %8", "v_addc_co_u32_e32 %3, vcc, %3, %8, vcc",
                                    "v_add_co_u32_e32 %4, vcc, %4, %8", "v_addc_co_u32_e32 %5, vcc, %5, %8, vcc", "v_add_co_u32_e32 %6, vcc, %6, %8", "v_addc_co_u32_e32 %7, vcc, %7, %8, vcc");
            if (KIND == 55) OP8_FMT("v_readfirstlane_b32 s10, %0", "v_readfirstlane_b32 s11, %1", "v_readfirstlane_b32 s12, %2", "v_readfirstlane_b32 s13, %3",
                                    "v_readfirstlane_b32 s14, %4", "v_readfirstlane_b32 s15, %5", "v_readfirstlane_b32 s16, %6", "v_readfirstlane_b32 s17, %7");
            if (KIND == 56) OP8_FMT("v_cmp_lt_f32_e32 vcc, %0, %8", "v_cmp_lt_f32_e32 vcc, %1, %8", "v_cmp_lt_f32_e32 vcc, %2, %8", "v_cmp_lt_f32_e32 vcc, %3, %8",
                                    "v_cmp_lt_f32_e32 vcc, %4, %8", "v_cmp_lt_f32_e32 vcc, %5, %8", "v_cmp_lt_f32_e32 vcc, %6, %8", "v_cmp_lt_f32_e32 vcc, %7, %8");
            if (KIND == 57) OP8_FMT("v_cmp_lt_u32_e64 s[10:11], %0, %8", "v_cndmask_b32_e64 %1, %1, %8, s[24:25]", "v_cmp_lt_u32_e64 s[14:15], %2, %8", "v_cndmask_b32_e64 %3, %3, %8, s[24:25]",
                                    "v_cmp_lt_u32_e64 s[18:19], %4, %8", "v_cndmask_b32_e64 %5, %5, %8, s[24:25]", "v_cmp_lt_u32_e64 s[22:23], %6, %8", "v_cndmask_b32_e64 %7, %7, %8, s[24:25]");
            if (KIND == 58) OP8_FMT("v_cmp_lt_u32_e64 s[10:11], %0, %8", "v_add_u32 %1, %1, %8", "v_add_u32 %2, %2, %8", "v_add_u32 %3, %3, %8",
                                    "v_add_u32 %4, %4, %8", "v_add_u32 %5, %5, %8", "v_add_u32 %6, %6, %8", "v_add_u32 %7, %7, %8");
            if (KIND == 59) OP8_FMT("v_cmp_lt_u32_e64 s[10:11], %0, %8", "v_mul_u32_u24 %1, %1, %8", "v_mul_u32_u24 %2, %2, %8", "v_mul_u32_u24 %3, %3, %8",
                                    "v_mul_u32_u24 %4, %4, %8", "v_mul_u32_u24 %5, %5, %8", "v_mul_u32_u24 %6, %6, %8", "v_mul_u32_u24 %7, %7, %8");
            if (KIND == 60) OP8_FMT("v_add_u32 %0, %0, %8", "v_add_u32 %0, %0, %8", "v_add_u32 %0, %0, %8", "v_add_u32 %0, %0, %8",
                                    "v_add_u32 %0, %0, %8", "v_add_u32 %0, %0, %8", "v_add_u32 %0, %0, %8", "v_add_u32 %0, %0, %8");
            if (KIND == 61) OP8_FMT("v_mul_u32_u24 %0, %0, %8", "v_mul_u32_u24 %0, %0, %8", "v_mul_u32_u24 %0, %0, %8", "v_mul_u32_u24 %0, %0, %8",
                                    "v_mul_u32_u24 %0, %0, %8", "v_mul_u32_u24 %0, %0, %8", "v_mul_u32_u24 %0, %0, %8", "v_mul_u32_u24 %0, %0, %8");
            if (KIND == 62) OP8_FMT("v_cmp_lt_u32_e32 vcc, %0, %8", "v_cndmask_b32_e32 %1, %1, %8, vcc", "v_cmp_lt_u32_e32 vcc, %2, %8", "v_cndmask_b32_e32 %3, %3, %8, vcc",
                                    "v_cmp_lt_u32_e32 vcc, %4, %8", "v_cndmask_b32_e32 %5, %5, %8, vcc", "v_cmp_lt_u32_e32 vcc, %6, %8", "v_cndmask_b32_e32 %7, %7, %8, vcc");
            if (KIND == 63) OP8_FMT("v_cmp_lt_u32_e32 vcc, %0, %8", "v_add_u32 %1, %1, %8", "v_cmp_lt_u32_e32 vcc, %2, %8", "v_add_u32 %3, %3, %8",
                                    "v_cmp_lt_u32_e32 vcc, %4, %8", "v_add_u32 %5, %5, %8", "v_cmp_lt_u32_e32 vcc, %6, %8", "v_add_u32 %7, %7, %8");
            if (KIND == 65) OP8_FMT("v_mul_u32_u24 %1, %0, %8", "v_add_u32 %0, %0, %8", "v_mul_u32_u24 %3, %2, %8", "v_add_u32 %2, %2, %8",
                                    "v_mul_u32_u24 %5, %4, %8", "v_add_u32 %4, %4, %8", "v_mul_u32_u24 %7, %6, %8", "v_add_u32 %6, %6, %8");
            if (KIND == 66) OP8_FMT("v_cmp_lt_u32_e64 s[10:11], %0, %8", "v_add_u32 %1, %1, %8", "v_add_u32 %2, %2, %8", "v_add_u32 %3, %3, %8",
                                    "v_cndmask_b32_e64 %4, %4, %8, s[10:11]", "v_add_u32 %5, %5, %8", "v_add_u32 %6, %6, %8", "v_add_u32 %7, %7, %8");
            if (KIND == 67) OP8_FMT("v_mul_u32_u24 %1, %0, %8", "v_mul_u32_u24 %0, %1, %8", "v_mul_u32_u24 %3, %2, %8", "v_mul_u32_u24 %2, %3, %8",
                                    "v_mul_u32_u24 %5, %4, %8", "v_mul_u32_u24 %4, %5, %8", "v_mul_u32_u24 %7, %6, %8", "v_mul_u32_u24 %6, %7, %8");
            if (KIND == 68) OP8_FMT("v_mul_u32_u24 %0, %0, %8", "v_add_u32 %1, %1, %8", "v_mul_u32_u24 %2, %2, %8", "v_add_u32 %3, %3, %8",
                                    "v_mul_u32_u24 %4, %4, %8", "v_add_u32 %5, %5, %8", "v_mul_u32_u24 %6, %6, %8", "v_add_u32 %7, %7, %8");
            if (KIND == 70) OP8_FMT("s_and_b64 s[10:11], exec, exec", "v_cndmask_b32_e64 %1, %1, %8, s[10:11]", "s_and_b64 s[12:13], exec, exec", "v_cndmask_b32_e64 %3, %3, %8, s[12:13]",
                                    "s_and_b64 s[14:15], exec, exec", "v_cndmask_b32_e64 %5, %5, %8, s[14:15]", "s_and_b64 s[16:17], exec, exec", "v_cndmask_b32_e64 %7, %7, %8, s[16:17]");
            if (KIND == 72) OP8_FMT("v_cmp_lt_u32_e32 vcc, %0, %8", "v_cndmask_b32_e32 %1, %1, %8, vcc", "v_cndmask_b32_e32 %2, %2, %8, vcc", "v_cndmask_b32_e32 %3, %3, %8, vcc",
                                    "v_cndmask_b32_e32 %4, %4, %8, vcc", "v_cndmask_b32_e32 %5, %5, %8, vcc", "v_cndmask_b32_e32 %6, %6, %8, vcc", "v_cndmask_b32_e32 %7, %7, %8, vcc");
            if (KIND == 73) OP8_FMT("v_cndmask_b32_e64 %0, %0, %8, vcc", "v_cndmask_b32_e64 %1, %1, %8, vcc", "v_cndmask_b32_e64 %2, %2, %8, vcc", "v_cndmask_b32_e64 %3, %3, %8, vcc",
                                    "v_cndmask_b32_e64 %4, %4, %8, vcc", "v_cndmask_b32_e64 %5, %5, %8, vcc", "v_cndmask_b32_e64 %6, %6, %8, vcc", "v_cndmask_b32_e64 %7, %7, %8, vcc");
            if (KIND == 74) OP8_FMT("v_cndmask_b32_e32 %0, %0, %8, vcc", "v_add_u32 %1, %1, %8", "v_add_u32 %2, %2, %8", "v_add_u32 %3, %3, %8",
                                    "v_add_u32 %4, %4, %8", "v_add_u32 %5, %5, %8", "v_add_u32 %6, %6, %8", "v_add_u32 %7, %7, %8");
            if (KIND == 75) OP8_FMT("s_and_b64 s[10:11], exec, exec", "v_add_u32 %1, %1, %8", "v_add_u32 %2, %2, %8", "v_add_u32 %3, %3, %8",
                                    "v_cndmask_b32_e64 %4, %4, %8, s[10:11]", "v_add_u32 %5, %5, %8", "v_add_u32 %6, %6, %8", "v_add_u32 %7, %7, %8");
            if (KIND == 76) OP8_FMT("v_cndmask_b32_e32 %0, %0, %8, vcc", "v_cndmask_b32_e32 %1, %8, %1, vcc", "v_cndmask_b32_e32 %2, %2, %8, vcc", "v_cndmask_b32_e32 %3, %8, %3, vcc",
                                    "v_cndmask_b32_e32 %4, %4, %8, vcc", "v_cndmask_b32_e32 %5, %8, %5, vcc", "v_cndmask_b32_e32 %6, %6, %8, vcc", "v_cndmask_b32_e32 %7, %8, %7, vcc");
            if (KIND == 80) OP8_64C("v_lshlrev_b64");
            if (KIND == 81) OP8_64C("v_lshrrev_b64");
            if (KIND == 82) OP8_FMT("v_ffbh_u32 %0, %0", "v_ffbh_u32 %1, %1", "v_ffbh_u32 %2, %2", "v_ffbh_u32 %3, %3", "v_ffbh_u32 %4, %4", "v_ffbh_u32 %5, %5", "v_ffbh_u32 %6, %6", "v_ffbh_u32 %7, %7");
            if (KIND == 83) OP8_FMT("v_ffbl_b32 %0, %0", "v_ffbl_b32 %1, %1", "v_ffbl_b32 %2, %2", "v_ffbl_b32 %3, %3", "v_ffbl_b32 %4, %4", "v_ffbl_b32 %5, %5", "v_ffbl_b32 %6, %6", "v_ffbl_b32 %7, %7");
            if (KIND == 84) OP8_FMT("v_readlane_b32 s10, %0, 3", "v_readlane_b32 s11, %1, 3", "v_readlane_b32 s12, %2, 3", "v_readlane_b32 s13, %3, 3",
                                    "v_readlane_b32 s14, %4, 3", "v_readlane_b32 s15, %5, 3", "v_readlane_b32 s16, %6, 3", "v_readlane_b32 s17, %7, 3");
            if (KIND == 85) OP8_FMT("v_writelane_b32 %0, s10, 3", "v_writelane_b32 %1, s10, 3", "v_writelane_b32 %2, s10, 3", "v_writelane_b32 %3, s10, 3",
                                    "v_writelane_b32 %4, s10, 3", "v_writelane_b32 %5, s10, 3", "v_writelane_b32 %6, s10, 3", "v_writelane_b32 %7, s10, 3");
            if (KIND == 86) OP8_64("v_fma_f64 %0, %0, %8, %8\n;");
            if (KIND == 100) OP8_64("v_mul_f64");
            if (KIND == 101) OP8_64("v_add_f64");
            if (KIND == 102) asm volatile("v_mul_f64 %0, %0, %1\nv_mul_f64 %0, %0, %1\nv_mul_f64 %0, %0, %1\nv_mul_f64 %0, %0, %1\n"
                                          "v_mul_f64 %0, %0, %1\nv_mul_f64 %0, %0, %1\nv_mul_f64 %0, %0, %1\nv_mul_f64 %0, %0, %1\n" : "+v"(d0) : "v"(dc));   // one dependent chain
            if (KIND == 103) asm volatile("v_add_f64 %0, %0, %1\nv_add_f64 %0, %0, %1\nv_add_f64 %0, %0, %1\nv_add_f64 %0, %0, %1\n"
                                          "v_add_f64 %0, %0, %1\nv_add_f64 %0, %0, %1\nv_add_f64 %0, %0, %1\nv_add_f64 %0, %0, %1\n" : "+v"(d0) : "v"(dc));
            if (KIND == 104) asm volatile("v_mul_f64 %0, %0, %2\nv_mul_f64 %1, %1, %2\nv_mul_f64 %0, %0, %2\nv_mul_f64 %1, %1, %2\n"
                                          "v_mul_f64 %0, %0, %2\nv_mul_f64 %1, %1, %2\nv_mul_f64 %0, %0, %2\nv_mul_f64 %1, %1, %2\n" : "+v"(d0), "+v"(d1) : "v"(dc));   // two chains
            if (KIND == 105) asm volatile("v_mul_f64 %0, %0, %1\nv_add_f64 %0, %0, %1\nv_mul_f64 %0, %0, %1\nv_add_f64 %0, %0, %1\n"
                                          "v_mul_f64 %0, %0, %1\nv_add_f64 %0, %0, %1\nv_mul_f64 %0, %0, %1\nv_add_f64 %0, %0, %1\n" : "+v"(d0) : "v"(dc));   // mul -> add chain (the D recurrence)
            if (KIND == 106) asm volatile("v_cmp_gt_f64 vcc, %0, %1\nv_cndmask_b32 %2, %2, %3, vcc\nv_cmp_gt_f64 vcc, %1, %0\nv_cndmask_b32 %2, %2, %3, vcc\n"
                                          "v_cmp_gt_f64 vcc, %0, %1\nv_cndmask_b32 %2, %2, %3, vcc\nv_cmp_gt_f64 vcc, %1, %0\nv_cndmask_b32 %2, %2, %3, vcc\n" : "+v"(d0), "+v"(d1), "+v"(a0) : "v"(c) : "vcc");
            if (KIND == 107) OP8_FMT("v_cmp_lt_u32_e32 vcc, %0, %8", "s_nop 4", "v_cndmask_b32_e32 %2, %2, %8, vcc", "v_cndmask_b32_e32 %3, %3, %8, vcc",
                                     "v_cndmask_b32_e32 %4, %4, %8, vcc", "v_cndmask_b32_e32 %5, %5, %8, vcc", "v_cndmask_b32_e32 %6, %6, %8, vcc", "v_cndmask_b32_e32 %7, %7, %8, vcc");
            if (KIND == 108) OP8_FMT("v_cmp_lt_u32_e32 vcc, %0, %8", "v_add_u32 %1, %1, %8", "v_cndmask_b32_e32 %2, %2, %8, vcc", "v_add_u32 %3, %3, %8",
                                     "v_cndmask_b32_e32 %4, %4, %8, vcc", "v_add_u32 %5, %5, %8", "v_cndmask_b32_e32 %6, %6, %8, vcc", "v_add_u32 %7, %7, %8");
            if (KIND == 109) OP8_FMT("v_cmp_lt_u32_e64 s[10:11], %0, %8", "v_cndmask_b32_e64 %1, %1, %8, s[10:11]", "v_cndmask_b32_e64 %2, %2, %8, s[10:11]", "v_cndmask_b32_e64 %3, %3, %8, s[10:11]",
                                     "v_cndmask_b32_e64 %4, %4, %8, s[10:11]", "v_cndmask_b32_e64 %5, %5, %8, s[10:11]", "v_cndmask_b32_e64 %6, %6, %8, s[10:11]", "v_cndmask_b32_e64 %7, %7, %8, s[10:11]");
            if (KIND == 111) OP8_FMT("v_cmp_lt_u32_e32 vcc, %0, %8", "v_cndmask_b32_e32 %1, %1, %8, vcc", "v_cndmask_b32_e32 %2, %2, %8, vcc", "v_add_u32 %3, %3, %8",
                                     "v_add_u32 %4, %4, %8", "v_add_u32 %5, %5, %8", "v_add_u32 %6, %6, %8", "v_add_u32 %7, %7, %8");     // the (lo, hi) pair of a double's select
            if (KIND == 112) OP8_FMT("v_cmp_lt_u32_e32 vcc, %0, %8", "v_cndmask_b32_e32 %1, %1, %8, vcc", "v_cndmask_b32_e64 %2, %2, %8, vcc", "v_add_u32 %3, %3, %8",
                                     "v_add_u32 %4, %4, %8", "v_add_u32 %5, %5, %8", "v_add_u32 %6, %6, %8", "v_add_u32 %7, %7, %8");     // the same, second one VOP3
            if (KIND == 113) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %5\nv_cndmask_b32_e32 %1, %1, %5, vcc\nv_mul_f64 %3, %3, %6\nv_cndmask_b32_e32 %2, %2, %5, vcc\n"
                                          "v_mul_f64 %4, %4, %6\nv_cndmask_b32_e32 %0, %0, %5, vcc\nv_mul_f64 %3, %3, %6\nv_cndmask_b32_e32 %1, %1, %5, vcc\n"
                                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(d0), "+v"(d1) : "v"(c), "v"(dc) : "vcc");      // selects with fp64 work between them
            if (KIND == 110) OP8_FMT("s_and_saveexec_b64 s[10:11], vcc", "v_add_u32 %1, %1, %8", "s_or_b64 exec, exec, s[10:11]", "v_add_u32 %3, %3, %8",
                                     "s_and_saveexec_b64 s[12:13], vcc", "v_add_u32 %5, %5, %8", "s_or_b64 exec, exec, s[12:13]", "v_add_u32 %7, %7, %8");      // exec-mask toggles around single VALU ops
            if (KIND == 88) OP8_32("v_fmac_f32");
            if (KIND == 89) OP8_3("v_dot4_u32_u8");
            if (KIND == 90) OP8_32("v_pk_add_u16 %0, %0, %8\n;");
            if (KIND == 91) OP8_3("v_pk_mad_u16");
            if (KIND == 92) OP8_32("v_mul_hi_u32");
            if (KIND == 93) OP8_FMT("v_rcp_f32 %0, %0", "v_rcp_f32 %1, %1", "v_rcp_f32 %2, %2", "v_rcp_f32 %3, %3", "v_rcp_f32 %4, %4", "v_rcp_f32 %5, %5", "v_rcp_f32 %6, %6", "v_rcp_f32 %7, %7");
            if (KIND == 94) OP8_FMT("v_mbcnt_lo_u32_b32 %0, %8, %0", "v_mbcnt_lo_u32_b32 %1, %8, %1", "v_mbcnt_lo_u32_b32 %2, %8, %2", "v_mbcnt_lo_u32_b32 %3, %8, %3",
                                    "v_mbcnt_lo_u32_b32 %4, %8, %4", "v_mbcnt_lo_u32_b32 %5, %8, %5", "v_mbcnt_lo_u32_b32 %6, %8, %6", "v_mbcnt_lo_u32_b32 %7, %8, %7");
            if (KIND == 95) OP8_FMT("v_mov_b32_dpp %0, %0 row_shr:1", "v_mov_b32_dpp %1, %1 row_shr:1", "v_mov_b32_dpp %2, %2 row_shr:1", "v_mov_b32_dpp %3, %3 row_shr:1",
                                    "v_mov_b32_dpp %4, %4 row_shr:1", "v_mov_b32_dpp %5, %5 row_shr:1", "v_mov_b32_dpp %6, %6 row_shr:1", "v_mov_b32_dpp %7, %7 row_shr:1");
            if (KIND == 18) OP8_3("v_and_or_b32");
            if (KIND == 19) OP8_3("v_lshl_add_u32");
            if (KIND == 20) OP8_3("v_add3_u32");
            if (KIND == 21) OP8_3("v_bfe_u32");
            if (KIND == 22) OP8_3("v_perm_b32");
            if (KIND == 23) OP8_3("v_mad_u32_u24");
            if (KIND == 24) OP8_3("v_alignbit_b32");
            if (KIND == 25) OP8_3("v_xad_u32");
            if (KIND == 26) OP8_3("v_or3_b32");
            if (KIND == 27) OP8_3("v_lshl_or_b32");
            if (KIND == 28) OP8_3("v_dot2_u32_u16");
            if (KIND == 29) OP8_3("v_fma_f32");
            if (KIND == 30) OP8_32("v_mul_f32");
            if (KIND == 31) OP8_RAW("", "v_add_u32_sdwa", " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD");
            if (KIND == 33) OP8_32("v_mov_b32 %0, %8\n;");
            if (KIND == 34) OP8_32("v_cvt_f32_u32 %0, %8\n;");
            if (KIND == 36) OP8_32("v_ashrrev_i32");
            if (KIND == 37) OP8_3("v_med3_u32");
            if (KIND == 38) OP8_3("v_max3_u32");
            if (KIND == 39) OP8_3("v_sad_u32");
            if (KIND == 40) OP8_3("v_bfi_b32");
        }
    }
    unsigned s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    double t = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
    if (s == 0x12345678u && t == 3.0) out[0] = s;      // keeps the chains alive
}

#include <cstring>
static const char *g_only = nullptr;      // argv[1]: run only the rows whose name contains it
template <int KIND>
static void run(const char *name, unsigned *d_out, int n_cu, double ghz)
{
    if (g_only && !strstr(name, g_only)) return;
    const int reps = 4096;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%-20s", name);
    for (int w = 1; w <= 8; w++) {
        rate_kernel<KIND><<<n_cu * w, 256>>>(d_out, 16, 1);            // warm
        CK(hipEventRecord(e0));
        rate_kernel<KIND><<<n_cu * w, 256>>>(d_out, reps, 1);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double inst_per_simd = (double)reps * 64 * w;                 // one workgroup of 4 wavefronts per CU and unit of w: one per SIMD
        printf("  w=%d %5.2f", w, ms * 1e-3 * ghz * 1e9 / inst_per_simd);
    }
    printf("\n");
}

int main(int argc, char **argv)
{
    if (argc > 1) g_only = argv[1];
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    double ghz = pr.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz nominal; cycles per wave-instruction per SIMD (at the nominal clock) by wavefronts per SIMD\n", pr.name, pr.multiProcessorCount, ghz);
    unsigned *d_out; CK(hipMalloc(&d_out, 64));
    int n = pr.multiProcessorCount;
    run<0>("v_add_u32", d_out, n, ghz);
    run<1>("v_xor_b32", d_out, n, ghz);
    run<7>("v_max_u32", d_out, n, ghz);
    run<5>("v_lshlrev_b32", d_out, n, ghz);
    run<6>("v_bcnt_u32_b32", d_out, n, ghz);
    run<10>("v_mul_u32_u24", d_out, n, ghz);
    run<2>("v_mul_lo_u32", d_out, n, ghz);
    run<8>("v_add_f32", d_out, n, ghz);
    run<11>("v_and_b32", d_out, n, ghz);
    run<12>("v_or_b32", d_out, n, ghz);
    run<13>("v_sub_u32", d_out, n, ghz);
    run<14>("v_lshrrev_b32", d_out, n, ghz);
    run<36>("v_ashrrev_i32", d_out, n, ghz);
    run<15>("v_min_u32", d_out, n, ghz);
    run<16>("v_cndmask_b32", d_out, n, ghz);
    run<17>("cmp+cndmask /2", d_out, n, ghz);
    run<41>("cndmask sgpr", d_out, n, ghz);
    run<42>("cmp+cnd sgpr /2", d_out, n, ghz);
    run<43>("cmp+add /2", d_out, n, ghz);
    run<45>("add+s_nop /2", d_out, n, ghz);
    run<46>("smov+cnd /2", d_out, n, ghz);
    run<50>("v_cmp vcc", d_out, n, ghz);
    run<51>("v_cmp sgpr same", d_out, n, ghz);
    run<52>("v_cmp sgpr 8", d_out, n, ghz);
    run<56>("v_cmp_f32 vcc", d_out, n, ghz);
    run<53>("v_add_co vcc", d_out, n, ghz);
    run<54>("add_co+addc_co", d_out, n, ghz);
    run<55>("v_readfirstlane", d_out, n, ghz);
    run<57>("cmp,cnd indep", d_out, n, ghz);
    run<58>("1 cmp + 7 add", d_out, n, ghz);
    run<59>("1 cmp + 7 mul24", d_out, n, ghz);
    run<60>("add chain", d_out, n, ghz);
    run<61>("mul24 chain", d_out, n, ghz);
    run<67>("mul24 chain of 2", d_out, n, ghz);
    run<62>("cmp>cnd vcc noWAR", d_out, n, ghz);
    run<63>("cmp,add noWAR", d_out, n, ghz);
    run<66>("cmp,3add,cnd,3add", d_out, n, ghz);
    run<70>("s_and>cnd sgpr", d_out, n, ghz);
    run<72>("cmp, 7 cnd vcc", d_out, n, ghz);
    run<73>("cnd_e64 vcc", d_out, n, ghz);
    run<74>("cnd vcc, 7 add", d_out, n, ghz);
    run<75>("s_and,3add,cnd,3add", d_out, n, ghz);
    run<76>("cnd vcc swapped", d_out, n, ghz);
    run<33>("v_mov_b32", d_out, n, ghz);
    run<18>("v_and_or_b32", d_out, n, ghz);
    run<26>("v_or3_b32", d_out, n, ghz);
    run<40>("v_bfi_b32", d_out, n, ghz);
    run<19>("v_lshl_add_u32", d_out, n, ghz);
    run<27>("v_lshl_or_b32", d_out, n, ghz);
    run<20>("v_add3_u32", d_out, n, ghz);
    run<25>("v_xad_u32", d_out, n, ghz);
    run<21>("v_bfe_u32", d_out, n, ghz);
    run<22>("v_perm_b32", d_out, n, ghz);
    run<24>("v_alignbit_b32", d_out, n, ghz);
    run<23>("v_mad_u32_u24", d_out, n, ghz);
    run<39>("v_sad_u32", d_out, n, ghz);
    run<37>("v_med3_u32", d_out, n, ghz);
    run<38>("v_max3_u32", d_out, n, ghz);
    run<28>("v_dot2_u32_u16", d_out, n, ghz);
    run<31>("v_add_u32_sdwa", d_out, n, ghz);
    run<29>("v_fma_f32", d_out, n, ghz);
    run<30>("v_mul_f32", d_out, n, ghz);
    run<34>("v_cvt_f32_u32", d_out, n, ghz);
    run<88>("v_fmac_f32", d_out, n, ghz);
    run<89>("v_dot4_u32_u8", d_out, n, ghz);
    run<90>("v_pk_add_u16", d_out, n, ghz);
    run<91>("v_pk_mad_u16", d_out, n, ghz);
    run<92>("v_mul_hi_u32", d_out, n, ghz);
    run<93>("v_rcp_f32", d_out, n, ghz);
    run<94>("v_mbcnt_lo", d_out, n, ghz);
    run<95>("v_mov_b32_dpp", d_out, n, ghz);
    run<82>("v_ffbh_u32", d_out, n, ghz);
    run<83>("v_ffbl_b32", d_out, n, ghz);
    run<84>("v_readlane_b32", d_out, n, ghz);
    run<85>("v_writelane_b32", d_out, n, ghz);
    run<80>("v_lshlrev_b64", d_out, n, ghz);
    run<81>("v_lshrrev_b64", d_out, n, ghz);
    run<86>("v_fma_f64", d_out, n, ghz);
    run<100>("v_mul_f64", d_out, n, ghz);
    run<101>("v_add_f64", d_out, n, ghz);
    run<102>("mul_f64 chain", d_out, n, ghz);
    run<103>("add_f64 chain", d_out, n, ghz);
    run<104>("mul_f64 2 chains", d_out, n, ghz);
    run<105>("mul>add f64 chain", d_out, n, ghz);
    run<106>("cmp_f64>cnd", d_out, n, ghz);
    run<107>("cmp,nop,6 cnd vcc", d_out, n, ghz);
    run<108>("alt: cmp,add,cnd..", d_out, n, ghz);
    run<109>("cmp, 7 cnd sgpr", d_out, n, ghz);
    run<111>("pair: cmp,2 cnd,5 add", d_out, n, ghz);
    run<112>("pair: cnd e32+e64", d_out, n, ghz);
    run<113>("pair: cnd,mul_f64 alt", d_out, n, ghz);
    run<110>("saveexec toggles", d_out, n, ghz);
    run<3>("v_add_f64", d_out, n, ghz);
    run<4>("v_mul_f64", d_out, n, ghz);
    return 0;
}

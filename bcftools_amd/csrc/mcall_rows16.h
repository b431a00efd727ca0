// mcall_rows16.h -- included inside namespace bcfgpu by mcall.hip (and by tools/microbench/rows16_product.hip, which checks it alone):
// the product over a wavefront's 64 lanes of sixteen rows of (mantissa, exponent) factors at once, as a reduce-scatter.
// Needs dpp_i32<>, dpp_f64<>, frexp_mant(), frexp_exp() from the including file.
typedef unsigned u2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double mk_f64(uint32_t lo, uint32_t hi) { return __builtin_bit_cast(double, (unsigned long long)hi << 32 | lo); }
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    return mk_f64((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, lane), (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), lane));
}
// Two rows of per-lane (mantissa, exponent) factors become one register pair: afterwards the lanes whose bit (5, 4, 3 or 2) is
// clear hold the products of row A's two halves, the others row B's -- a reduce-scatter step; v_permlane32/16_swap exchange
// the halves of two registers in one instruction, the steps inside a row of 16 lanes are moves with a bank mask.
__device__ __forceinline__ void merge_swap32(double &am, int &ae, const double bm, const int be)
{
    const unsigned long long a = __builtin_bit_cast(unsigned long long, am), b = __builtin_bit_cast(unsigned long long, bm);
    const u2_t lo = __builtin_amdgcn_permlane32_swap((uint32_t)a, (uint32_t)b, false, false);
    const u2_t hi = __builtin_amdgcn_permlane32_swap((uint32_t)(a >> 32), (uint32_t)(b >> 32), false, false);
    const u2_t ee = __builtin_amdgcn_permlane32_swap((uint32_t)ae, (uint32_t)be, false, false);
    am = mk_f64(lo.x, hi.x) * mk_f64(lo.y, hi.y);
    ae = (int)ee.x + (int)ee.y;
}
__device__ __forceinline__ void merge_swap16(double &am, int &ae, const double bm, const int be)
{
    const unsigned long long a = __builtin_bit_cast(unsigned long long, am), b = __builtin_bit_cast(unsigned long long, bm);
    const u2_t lo = __builtin_amdgcn_permlane16_swap((uint32_t)a, (uint32_t)b, false, false);
    const u2_t hi = __builtin_amdgcn_permlane16_swap((uint32_t)(a >> 32), (uint32_t)(b >> 32), false, false);
    const u2_t ee = __builtin_amdgcn_permlane16_swap((uint32_t)ae, (uint32_t)be, false, false);
    am = mk_f64(lo.x, hi.x) * mk_f64(lo.y, hi.y);
    ae = (int)ee.x + (int)ee.y;
}
// CTRL pairs every lane with one whose owner bit differs (row_ror:8 for bit 3, row_half_mirror for bit 2); BM = the banks (four
// lanes each) of a row that own row B
template <int CTRL, int BM> __device__ __forceinline__ int merge_dpp_word(const int a, const int b, int &recv)
{
    recv = __builtin_amdgcn_update_dpp(0, a, CTRL, 0xf, 0xf, false);          // the partner's A ...
    recv = __builtin_amdgcn_update_dpp(recv, b, CTRL, 0xf, BM, false);        // ... its B where this lane owns row B
    return __builtin_amdgcn_update_dpp(a, b, 0xE4, 0xf, BM, false);           // this lane's own value of the row it owns
}
template <int CTRL, int BM> __device__ __forceinline__ void merge_dpp(double &am, int &ae, const double bm, const int be)
{
    const unsigned long long a = __builtin_bit_cast(unsigned long long, am), b = __builtin_bit_cast(unsigned long long, bm);
    int rlo, rhi, re;
    const int klo = merge_dpp_word<CTRL, BM>((int)(uint32_t)a, (int)(uint32_t)b, rlo);
    const int khi = merge_dpp_word<CTRL, BM>((int)(uint32_t)(a >> 32), (int)(uint32_t)(b >> 32), rhi);
    const int ke = merge_dpp_word<CTRL, BM>(ae, be, re);
    am = mk_f64((uint32_t)klo, (uint32_t)khi) * mk_f64((uint32_t)rlo, (uint32_t)rhi);
    ae = ke + re;
}
// sixteen rows of per-lane factors (mantissas in [0.5, 1] -- 64 of them multiply to 2^-64 at least) -> lane l holds the product
// over all 64 lanes of row l >> 2
__device__ __forceinline__ void rows16_product(double (&m)[16], int (&e)[16])
{
    #pragma unroll
    for (int r = 0; r < 8; ++r) merge_swap32(m[r], e[r], m[r + 8], e[r + 8]);
    #pragma unroll
    for (int r = 0; r < 4; ++r) merge_swap16(m[r], e[r], m[r + 4], e[r + 4]);
    #pragma unroll
    for (int r = 0; r < 2; ++r) merge_dpp<0x128, 0xC>(m[r], e[r], m[r + 2], e[r + 2]);
    merge_dpp<0x141, 0xA>(m[0], e[0], m[1], e[1]);
    m[0] *= dpp_f64<0xB1>(m[0]); e[0] += dpp_i32<0xB1>(e[0]);
    m[0] *= dpp_f64<0x4E>(m[0]); e[0] += dpp_i32<0x4E>(e[0]);
    e[0] += frexp_exp(m[0]); m[0] = frexp_mant(m[0]);
}


// mcall.hip -- the multiallelic caller `bcftools call -m`, one wavefront (64 lanes) per site.
//
// Replaces mcall() (mcall.c:1430-1684) and what it calls: set_pdg (:451-544),
// mcall_find_best_alleles (:591-710), mcall_set_ref_genotypes (:713-743),
// mcall_call_genotypes (:745-886), init_allele_trimming_maps (:547-570),
// mcall_trim_and_update_PLs (:1158-1194), plus the QS/-G/-F frequency set-up (:1453-1535)
// and the record-loop prologue of vcfcall.c:1112-1115.
//
// The work per site is a chain of short, dependent steps, so throughput comes from having thousands of sites in
// flight: one 64-lane workgroup per site, no workgroup barriers that matter.  The allele-subset scan of
// find_best_alleles is a (subsets x genotypes) * (genotypes x samples) product with a sparse left factor: the FAST
// instantiations (u8 PLs of the fused pipeline; ploidy arrays and -G groups as template parameters) scan a lane per sample
// with the subsets' coefficients in scalar registers (sparse_scan below; the 25-subset instantiation alone still uses dense
// tiles on the f64 matrix cores); the general ones (missing PLs, int32 PLs of a VCF) keep a sample's P(D|G) in the lane's
// LDS column and index it with run-time genotype indices.  Genotype calling runs lanes over samples in both.
//
// Integer results (GT, AC/AN, trimmed PL, GQ) are exact; log-likelihood sums are accumulated as running products
// (mantissa, exponent), so QUAL agrees with the sequential CPU sum of logs to ~1e-13 relative, far inside the 1e-4
// contract.  Order-sensitive float32 pieces (group qsum from AD, -F prior, normalisation) are replayed sequentially
// by single lanes exactly as the reference does.
#include <hip/hip_runtime.h>
#include <math.h>
#include "kernels.h"
#include "kfunc_dev.h"

namespace bcfgpu {

#define WGS 64
#define MISSING BCFGPU_INT32_MISSING
#define VEND    BCFGPU_INT32_VECTOR_END

__device__ __forceinline__ int a2gt(int a, int b) { return a > b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a; }
__device__ __attribute__((noinline)) double lse2(double a, double b)
{
    if (a > b) return log(1 + exp(b - a)) + a;
    else       return log(1 + exp(a - b)) + b;
}
__device__ __forceinline__ double wsum(double v)
{
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wor(int v)
{
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double frexp_mant(double x) { return __builtin_amdgcn_frexp_mant(x); }   // in [0.5,1)
__device__ __forceinline__ int frexp_exp(double x) { return __builtin_amdgcn_frexp_exp(x); }
__device__ __forceinline__ int wsumi(int v)
{
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// A value of the lane's partner inside its row of 16 lanes, by data-parallel-primitive moves (no trip through the LDS
// crossbar as __shfl_xor makes): the four pairings quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror join
// lanes, quads, halves -- after them every lane of the row holds the reduction over its 16 lanes.
template <int CTRL> __device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const uint32_t lo = (uint32_t)dpp_i32<CTRL>((int)(uint32_t)b), hi = (uint32_t)dpp_i32<CTRL>((int)(uint32_t)(b >> 32));
    return __builtin_bit_cast(double, (unsigned long long)hi << 32 | lo);
}
// product of the 16 lanes' (mantissa, exponent) pairs, renormalised at every step, and the OR of their flags
template <int CTRL> __device__ __forceinline__ void row16_step(double &m, int &e, int &f)
{
    const double mm = m * dpp_f64<CTRL>(m);
    e += dpp_i32<CTRL>(e) + __builtin_amdgcn_frexp_exp(mm);
    m = __builtin_amdgcn_frexp_mant(mm);
    f |= dpp_i32<CTRL>(f);
}
__device__ __forceinline__ void row16_product(double &m, int &e, int &f)
{
    row16_step<0xB1>(m, e, f); row16_step<0x4E>(m, e, f); row16_step<0x141>(m, e, f); row16_step<0x140>(m, e, f);
}

// one allele subset of mcall_find_best_alleles: genotype indices and frequency products
struct Subset {
    int ia, ib, ic;             // ib/ic = -1 when absent
    int iaa, ibb, icc, iab, iac, ibc;
    double fa, fb, fc, fa2, fb2, fc2, fab, fac, fbc;
};

template <int NSUB>
struct CallShared {
    float qsum[5];              // current group's allele frequencies
    int nsub;                   // subsets visited for the current group
    Subset sub[NSUB];               // (the kernel's LDS decides how many sites a CU holds: 15 subsets leave room for 16)
    double red[32];             // reduced log-likelihood sums, indexed like sub[] (FAST: + the sum row at [nsub])
    int rede[32];               // FAST: binary exponents of the row products before the logarithm
    int redset;
    int als_new, nals_new, is_variant, early;
    int als_map[5]; int pl_map[15];
    int ac[5];
    double max_qual, ref_lk, lk_sum, ref_cur;
    double red_dip;             // FAST: like red[nsub] but over the samples of ploidy 1 or 2 only (pairs and triples)
    int prior_fail;
};

template <int NG>
__device__ __forceinline__ void load_pl(const McallParams &P, int is, int s, int ngts, int (&pl)[NG])
{
    const size_t S = P.n_smpl;
    if (P.pl_is_u8) {
        const uint8_t *src = reinterpret_cast<const uint8_t*>(P.pl) + (size_t)is * BCFGPU_MAX_PL * S + s;
        #pragma unroll
        for (int j = 0; j < NG; ++j) pl[j] = j < ngts ? (int)src[(size_t)j * S] : VEND;
    } else {
        const int32_t *src = reinterpret_cast<const int32_t*>(P.pl) + (size_t)is * P.n_gt_max * S + s;
        #pragma unroll
        for (int j = 0; j < NG; ++j) pl[j] = j < ngts ? src[(size_t)j * S] : VEND;
    }
}

__device__ __attribute__((noinline)) double npow10(double x) { return pow(10., x); }
__device__ __forceinline__ double pl2prob(const double *pl2p, int v) { return v < 256 ? pl2p[v] : npow10(-v / 10.); }

// set_pdg for one sample (mcall.c:460-543) without its final normalisation: pdg[] receives the raw 10^(-PL/10)
// values and the function returns their sum, or 0 when the sample is "all missing" (pdg must then read as zeros).
// The caller divides (exactly as `pdg[j] /= sum` does) where bit-exact values are needed.  `scr` is this lane's
// LDS column (stride WGS) used only by the rare partially-missing path, which needs run-time indexing.
template <int NG>
__device__ __forceinline__ double set_pdg_one(const double *pl2p, int (&pl)[NG], double (&pdg)[NG], int n_gt, int nals, int unseen, int *scr)
{
    double sum = 0;
    int j = n_gt;               // index of the first missing value, or n_gt
    #pragma unroll
    for (int k = 0; k < NG; ++k) {
        if (k < n_gt && j == n_gt) {
            if (pl[k] == VEND) j = 0;
            else if (pl[k] == MISSING) j = k;
            else { pdg[k] = pl2prob(pl2p, pl[k]); sum += pdg[k]; }
        }
    }
    if (j == 0) {
        j = n_gt; sum = n_gt;
    } else if (j < n_gt && unseen < 0) {
        sum = 0;
        #pragma unroll
        for (int k = 0; k < NG; ++k)
            if (k < n_gt) {
                if (pl[k] == MISSING) pl[k] = 255;
                pdg[k] = pl2prob(pl2p, pl[k]); sum += pdg[k];
            }
        j = n_gt;
    }
    if (j < n_gt) {
        // fill missing values from the unseen-allele likelihoods (mcall.c:495-527)
        #pragma unroll
        for (int k = 0; k < NG; ++k) scr[k * WGS] = pl[k];
        int jj = 0;
        sum = 0;
        for (int ia = 0; ia < nals; ia++)
            for (int ib = 0; ib <= ia; ib++) {
                if (scr[jj * WGS] == MISSING) {
                    int k = a2gt(ia, unseen);
                    if (scr[k * WGS] == MISSING) k = a2gt(ib, unseen);
                    if (scr[k * WGS] == MISSING) k = a2gt(unseen, unseen);
                    if (scr[k * WGS] == MISSING) scr[jj * WGS] = 255;
                    else scr[jj * WGS] = scr[k * WGS];
                }
                jj++;
            }
        #pragma unroll
        for (int k = 0; k < NG; ++k)
            if (k < n_gt) { pl[k] = scr[k * WGS]; pdg[k] = pl2p[pl[k] & 255]; sum += pdg[k]; }
    }
    if (sum == (double)n_gt) {
        #pragma unroll
        for (int k = 0; k < NG; ++k) pdg[k] = 0;
        return 0.0;
    }
    #pragma unroll
    for (int k = 0; k < NG; ++k) if (k >= n_gt) pdg[k] = 0;
    return sum;
}

// a group id outside [0, n_grp) is reported by grp_check_kernel (BCFGPU_E_RANGE); clamped here so that it cannot index past the tables
#define GRP_OF(s_) (min(max(P.grp[s_], 0), ngrp - 1))
__device__ __forceinline__ void write_skipped(bcfgpu_call_site *cs, int ret)
{
    cs->ret = ret; cs->nals_new = 0; cs->als_new = 0; cs->an = 0; cs->qual = 0; cs->qual_missing = 0; cs->pl_dropped = 0;
    cs->has_i16 = 0; cs->mq = 0; cs->pv4_tested = 0;
    for (int i = 0; i < 4; ++i) { cs->dp4[i] = 0; cs->pv4[i] = 0.f; }
    for (int i = 0; i < 5; ++i) { cs->als_map[i] = -1; cs->ac[i] = 0; }
}

typedef double d4_t __attribute__((ext_vector_type(4)));

// ---- the subset scan of the sites with at most fifteen subsets: a lane per sample, the vector ALU only ----
// The scan is (subset coefficients) x (genotypes x samples), but the coefficient matrix is sparse -- a single allele has one
// non-zero, a pair three, a triple six: 47 of the 256 entries of a 4-allele site's 16 x 16 tile -- and on this chip an fp64
// matrix instruction does its 1024 multiply-adds in the 64 cycles the vector ALU needs for as many (profiles/r4_mfma_f64_valu.txt),
// and holds the SIMD meanwhile.  So the sites of up to four alleles with a frequency (all but the 25-subset instantiation)
// are scanned a lane per sample: the sample's P(D|G) from the table, 42 multiply-adds for its six pairs and four triples with
// the coefficients in scalar registers, one multiplication into each subset's running product; no padding of rows or of the
// inner dimension.  The alleles are visited in a permuted order -- those with a frequency first -- by permuting which PL plane
// a genotype slot is loaded from (wavefront-uniform), so that the subsets are compile-time register indices.
#include "mcall_rows16.h"

// One group's scan.  NZ = alleles with a frequency (1..4; a group without any runs as NZ = 1: it has no pair).  `permw`: nibble x =
// the allele visited as x, those with a frequency first, both parts in the reference's order, so that pair i / triple i of the
// permuted enumeration is the i-th pair / triple mcall_find_best_alleles visits (mcall.c:617-698).  Slots of the result: 0-4 the
// single alleles in permuted order, 5.. the pairs, then the triples, then the row of the samples' normalisation sums; lane l
// returns slot l >> 2.  HAP: haploid samples take f_a P(aa) + f_b P(bb) (+ f_c P(cc)) (mcall.c:643, 688), samples of ploidy 0
// enter the single-allele rows only, and (dip_m, dip_e) is the normalisation product over the samples of ploidy 1 or 2.
#ifndef MCALL_SCAN_INLINE
#define MCALL_SCAN_INLINE __forceinline__
#endif
template <int NZ, bool HAP, int SPL>
__device__ MCALL_SCAN_INLINE void sparse_scan(const McallParams &P, const uint8_t *plb, const int S, const int nals, const int g, const int ngrp,
                                            const int s_first, const int s_last, const float *qf, const uint32_t permw, const double *s_p2, uint8_t *s_nz,
                                            const bool note_nz, double &out_m, int &out_e, double &dip_m, int &dip_e, bool &f_single, bool &f_pt)
{
    constexpr int NP = NZ * (NZ - 1) / 2, NT = NZ * (NZ - 1) * (NZ - 2) / 6, SUMSLOT = 5 + NP + NT;
    constexpr int PX[6] = {1, 2, 2, 3, 3, 3}, PY[6] = {0, 0, 1, 0, 1, 2};
    constexpr int TX[4] = {2, 3, 3, 3}, TY[4] = {1, 1, 2, 2}, TW[4] = {0, 0, 0, 1};
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));        // (what is derived from the lane number is formed here, per group: hoisted out of the loop over the groups it holds registers through all of it)
    const size_t Ss = (size_t)S;
    // the coefficients: lane i < 6 forms pair i's, lane 6 + i triple i's (float quotients widened, mcall.c:629-630, 671-673)
    double cfa = 0., cfb = 0., cfc = 0.;
    if (NP > 0) {
        const int i = tid < 6 ? tid : (tid - 6) & 3;
        const int x = tid < 6 ? (0x333221 >> (4 * i)) & 15 : (0x3332 >> (4 * i)) & 15;
        const int y = tid < 6 ? (0x210100 >> (4 * i)) & 15 : (0x2211 >> (4 * i)) & 15;
        const int w = tid < 6 ? -1 : (0x1000 >> (4 * i)) & 15;
        const float qa = qf[(permw >> (4 * x)) & 7], qb = qf[(permw >> (4 * y)) & 7], qc = w >= 0 ? qf[(permw >> (4 * w)) & 7] : 0.f;
        const float den = w >= 0 ? qa + qb + qc : qa + qb;
        cfa = (double)(qa / den); cfb = (double)(qb / den); cfc = (double)(qc / den);
    }
    // diploid samples: pairs as f_a^2 P(aa) + f_b^2 P(bb) + 2 f_a f_b P(ab) with the three products in scalar registers; triples (and,
    // with a ploidy array, the pairs too) from the frequencies alone -- f_a (f_a P(aa) + f_b 2P(ab) + f_c 2P(ac)) + f_b (f_b P(bb) +
    // f_c 2P(bc)) + f_c f_c P(cc): twelve scalar values instead of twenty-four, and the haploid sum falls out of its first terms
    constexpr bool PDIRECT = !HAP;
    #ifndef MCALL_TRIPLE_DIRECT
    #define MCALL_TRIPLE_DIRECT 0
    #endif
    constexpr bool TDIRECT = !HAP && MCALL_TRIPLE_DIRECT;      // (24 more scalar registers: measured, profiles/r5_mcall_sparse.txt)
    double k_a2[NP > 0 ? NP : 1], k_b2[NP > 0 ? NP : 1], k_ab[NP > 0 ? NP : 1], k_pa[NP > 0 ? NP : 1], k_pb[NP > 0 ? NP : 1];
    double k_ta[NT > 0 ? NT : 1], k_tb[NT > 0 ? NT : 1], k_tc[NT > 0 ? NT : 1];
    double k_t2[NT > 0 ? NT : 1][6];                             // TDIRECT: f_a^2, f_b^2, f_c^2, 2 f_a f_b, 2 f_a f_c, 2 f_b f_c
    {
        const double a2 = cfa * cfa, b2 = cfb * cfb, ab = 2 * cfa * cfb;
        #pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (PDIRECT) { k_a2[i] = readlane_f64(a2, i); k_b2[i] = readlane_f64(b2, i); k_ab[i] = readlane_f64(ab, i); }
            else { k_pa[i] = readlane_f64(cfa, i); k_pb[i] = readlane_f64(cfb, i); }
        }
        #pragma unroll
        for (int i = 0; i < NT; ++i) {
            if (TDIRECT) {
                k_t2[i][0] = readlane_f64(a2, 6 + i); k_t2[i][1] = readlane_f64(b2, 6 + i); k_t2[i][2] = readlane_f64(cfc * cfc, 6 + i);
                k_t2[i][3] = readlane_f64(ab, 6 + i); k_t2[i][4] = readlane_f64(2 * cfa * cfc, 6 + i); k_t2[i][5] = readlane_f64(2 * cfb * cfc, 6 + i);
            } else { k_ta[i] = readlane_f64(cfa, 6 + i); k_tb[i] = readlane_f64(cfb, 6 + i); k_tc[i] = readlane_f64(cfc, 6 + i); }
        }
    }
    double man[16]; int ex[16];
    #pragma unroll
    for (int r = 0; r < 16; ++r) { man[r] = 1.0; ex[r] = 0; }
    double sdm = 1.0; int sde = 0;
    bool fs = false, fp = false;
    // the PL plane of permuted genotype (x >= y)
    uint32_t offv = 0;                                                       // lane c: byte offset of the plane of permuted genotype c = (x >= y)
    if (tid < 15) {
        const int x = (int)((0x444443333222110ull >> (4 * tid)) & 15), y = (int)((0x432103210210100ull >> (4 * tid)) & 15);
        offv = (uint32_t)a2gt((int)((permw >> (4 * x)) & 7), (int)((permw >> (4 * y)) & 7)) * (uint32_t)S;
    }
    // A lane takes SPL = four consecutive samples, one 4-byte word per plane: 256 samples a trip; SPL = 1 (a byte per plane) for a range of
    // at most 128 samples -- a small group: 38 samples of 1000 in 26 populations would keep ten lanes busy at four a lane.
    for (int s0 = s_first; s0 < (BCFGPU_ABL(P, 16) ? 0 : s_last); s0 += SPL * WGS) {
        const int sb = s0 + SPL * tid, rem = S - sb;
        const bool live = rem > 0 && sb < s_last;
        // permuted genotype slot c is loaded from the plane lane c of `offv` names (one v_readlane per plane and trip)
        uint32_t w[15];
        if (SPL == 1 || __all(!live || rem >= 4)) {
            const uint32_t at = live ? (uint32_t)sb : 0u;            // (a lane past the range reads a word it does not use)
            #pragma unroll
            for (int x = 0; x < 5; ++x) {
                #pragma unroll
                for (int y = 0; y <= x; ++y) {
                    const int c = x * (x + 1) / 2 + y;
                    uint32_t v = 0;
                    if (x < NZ || x < nals) { if (SPL == 4) __builtin_memcpy(&v, plb + ((uint32_t)__builtin_amdgcn_readlane((int)offv, c) + at), 4); else v = plb[(uint32_t)__builtin_amdgcn_readlane((int)offv, c) + at]; }
                    w[c] = v;
                }
            }
        } else {                                                     // the trip that holds the last samples of a count not divisible by four
            #pragma unroll
            for (int c = 0; c < 15; ++c) {
                uint32_t v = 0;
                if (live && c < nals * (nals + 1) / 2) {
                    const uint8_t *src = plb + ((uint32_t)__builtin_amdgcn_readlane((int)offv, c) + (uint32_t)sb);
                    if (rem >= 4) __builtin_memcpy(&v, src, 4);
                    else { v = src[0]; if (rem > 1) v |= (uint32_t)src[1] << 8; if (rem > 2) v |= (uint32_t)src[2] << 16; }
                }
                w[c] = v;
            }
        }
        uint32_t pw = 0x02020202u, gm = live ? 0xfu : 0u;
        if (live && (HAP || ngrp > 1)) {
            if (SPL == 1) {
                if (HAP) pw = P.ploidy[sb];
                if (ngrp > 1) gm = P.grp[sb] == g ? 1u : 0u;
            } else if (rem >= 4) {
                if (HAP) __builtin_memcpy(&pw, P.ploidy + sb, 4);
                if (ngrp > 1) {
                    int gv[4];
                    __builtin_memcpy(gv, P.grp + sb, 16);
                    gm = (gv[0] == g ? 1u : 0u) | (gv[1] == g ? 2u : 0u) | (gv[2] == g ? 4u : 0u) | (gv[3] == g ? 8u : 0u);
                }
            } else {
                #pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < rem) {
                        if (HAP) pw = (pw & ~(0xffu << (8 * j))) | (uint32_t)P.ploidy[sb + j] << (8 * j);
                        if (ngrp > 1 && GRP_OF(sb + j) != g) gm &= ~(1u << j);
                    }
            }
        }
        uint32_t any = 0;
        #pragma unroll
        for (int c = 0; c < 15; ++c) any |= w[c];
        // which samples carry data, for the genotypes of a site that stays REF-only (set_pdg: all PLs 0 = no data, mcall.c:529-535)
        if (SPL == 4 && note_nz && live)
            s_nz[sb >> 2] = (uint8_t)(((any & 0xffu) ? 1u : 0u) | ((any & 0xff00u) ? 2u : 0u) | ((any & 0xff0000u) ? 4u : 0u) | ((any & 0xff000000u) ? 8u : 0u));
        #pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const bool has = ((any >> (8 * j)) & 0xff) != 0 && ((gm >> j) & 1);
            const int pd = HAP ? (int)((pw >> (8 * j)) & 0xff) : 2;
            if (has) {
                fs = true;
                double p[15];
                double sum = 0.;
                #pragma unroll
                for (int x = 0; x < 5; ++x) {
                    if (x < NZ || x < nals) {
                        #pragma unroll
                        for (int y = 0; y <= x; ++y) {
                            const int c = x * (x + 1) / 2 + y;
                            p[c] = s_p2[(w[c] >> (8 * j)) & 0xff];
                            sum += p[c];
                        }
                        man[x] *= p[x * (x + 3) / 2];
                    }
                }
                man[SUMSLOT] *= sum;
                if (!HAP || pd == 1 || pd == 2) {
                    fp = true;
                    if (HAP) sdm *= sum;
                    const bool dipl = !HAP || pd == 2;
                    // twice the heterozygous values (exact), for the forms built from the frequencies alone
                    double h2[10];
                    if (!(PDIRECT && TDIRECT))
                    #pragma unroll
                    for (int x = 1; x < NZ; ++x)
                        #pragma unroll
                        for (int y = 0; y < x; ++y) h2[x * (x - 1) / 2 + y] = p[x * (x + 1) / 2 + y] + p[x * (x + 1) / 2 + y];
                    #pragma unroll
                    for (int i = 0; i < NP; ++i) {
                        const int x = PX[i], y = PY[i];
                        const double paa = p[x * (x + 3) / 2], pbb = p[y * (y + 3) / 2];
                        double v;
                        if (PDIRECT) v = __builtin_fma(k_ab[i], p[x * (x + 1) / 2 + y], __builtin_fma(k_b2[i], pbb, k_a2[i] * paa));
                        else {
                            const double ta = k_pa[i] * paa, tb = k_pb[i] * pbb;
                            const double v2 = __builtin_fma(k_pa[i], __builtin_fma(k_pb[i], h2[x * (x - 1) / 2 + y], ta), k_pb[i] * tb);
                            v = dipl ? v2 : ta + tb;
                        }
                        man[5 + i] *= v;
                    }
                    #pragma unroll
                    for (int i = 0; i < NT; ++i) {
                        const int x = TX[i], y = TY[i], z = TW[i];
                        if (TDIRECT) {
                            man[5 + NP + i] *= __builtin_fma(k_t2[i][5], p[y * (y + 1) / 2 + z], __builtin_fma(k_t2[i][4], p[x * (x + 1) / 2 + z],
                                               __builtin_fma(k_t2[i][3], p[x * (x + 1) / 2 + y], __builtin_fma(k_t2[i][2], p[z * (z + 3) / 2],
                                               __builtin_fma(k_t2[i][1], p[y * (y + 3) / 2], k_t2[i][0] * p[x * (x + 3) / 2])))));
                            continue;
                        }
                        const double ta = k_ta[i] * p[x * (x + 3) / 2], tb = k_tb[i] * p[y * (y + 3) / 2], tc = k_tc[i] * p[z * (z + 3) / 2];
                        const double ua = __builtin_fma(k_tc[i], h2[x * (x - 1) / 2 + z], __builtin_fma(k_tb[i], h2[x * (x - 1) / 2 + y], ta));
                        const double ub = __builtin_fma(k_tc[i], h2[y * (y - 1) / 2 + z], tb);
                        const double v2 = __builtin_fma(k_ta[i], ua, __builtin_fma(k_tb[i], ub, k_tc[i] * tc));
                        man[5 + NP + i] *= (HAP && !dipl) ? ta + tb + tc : v2;
                    }
                }
            }
        }
        // (renormalised once per trip, at most four samples: four factors are at least 1e-150 together; splitting off a power of two is exact)
        #pragma unroll
        for (int r = 0; r < 16; ++r) { ex[r] += frexp_exp(man[r]); man[r] = frexp_mant(man[r]); }
        if (HAP) { sde += frexp_exp(sdm); sdm = frexp_mant(sdm); }
    }
    f_single = __any(fs); f_pt = __any(fp);
    rows16_product(man, ex);
    out_m = man[0]; out_e = ex[0];
    if (HAP) {
        int f_ = 0;
        row16_product(sdm, sde, f_);
        #pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const double mm = sdm * __shfl_xor(sdm, o);
            sde += __shfl_xor(sde, o) + frexp_exp(mm);
            sdm = frexp_mant(mm);
        }
        dip_m = sdm; dip_e = sde;
    }
}

// FAST: the u8-PL case of the fused pipeline (HAP: with a ploidy array; sample groups in both).  The subset scan of find_best_alleles is the
// product (subset coefficients [rows] x genotypes [k]) * (genotypes [k] x samples [cols]); a row for each sample's
// normalisation sum comes with it, whose product is divided out at the end.  NSUB <= 15: sparse_scan() above, a lane per sample;
// NSUB = 25 (five alleles with a frequency): dense tiles on the f64 matrix cores (v_mfma_f64_16x16x4_f64), 16 samples per issue.
#ifndef MCALL_WAVES
#define MCALL_WAVES 4        // wavefronts per SIMD of the all-diploid FAST instantiations
#endif
template <int MAXA, int NSUB, bool FAST, bool HAP, bool GRP>
#ifndef MCALL_WAVES_GRP
#define MCALL_WAVES_GRP 3    // ... of the instantiations with sample groups
#endif
#ifndef MCALL_WAVES_HAP
#define MCALL_WAVES_HAP 3    // ... with a ploidy array (two coefficient matrices)
#endif
__global__ __launch_bounds__(WGS) __attribute__((amdgpu_waves_per_eu(!FAST ? 1 : (HAP || GRP) && NSUB > 15 ? 3 : HAP ? MCALL_WAVES_HAP : GRP ? MCALL_WAVES_GRP : MCALL_WAVES, !FAST ? 8 : (HAP || GRP) && NSUB > 15 ? 3 : HAP ? MCALL_WAVES_HAP : GRP ? MCALL_WAVES_GRP : MCALL_WAVES))) void mcall_kernel(const McallParams P)
{
    constexpr int NG = MAXA * (MAXA + 1) / 2;
    constexpr int TILES = NSUB >= 16 ? 2 : 1;     // 16-row tiles of the coefficient matrix (subsets + the sum row)
    extern __shared__ __align__(16) unsigned char dsm[];
    float *s_gq = reinterpret_cast<float*>(dsm);              // [n_grp][5] group qsum, then [n_grp][2] best allele sets
    __shared__ CallShared<(FAST && GRP) ? 1 : NSUB> sh;        // (the batched groups of FAST && GRP keep their subsets in s_sid)
    __shared__ double s_pdg[FAST ? 1 : NG * WGS];   // the lane's current sample: raw P(D|G) (not yet divided by its sum)
    int *s_fill = reinterpret_cast<int*>(s_pdg);   // scratch of set_pdg's rare missing-value path (used before s_pdg is written)
    // pass 1: per-lane running products of the subset likelihoods, kept as mantissa (f64) and exponent (i32):
    //         sum_s log(val_s) = log(prod_s val_s), so each sample costs a multiply + frexp instead of a log()
    // pass 2: the lane's current sample: PLs after set_pdg's in-place fills, genotype posteriors
    // FAST: pass 1 = the coefficient matrix; pass 2 = the lane's PL bytes (u8) and genotype posteriors (f32)
    constexpr int U1 = FAST ? TILES * 16 * 16 * 8 * (HAP ? 2 : 1) : NSUB * WGS * 12, U2 = FAST ? NG * WGS * 5 : NG * WGS * 8,
                  UB = U1 > U2 ? U1 : U2;
    __shared__ __align__(8) unsigned char s_union[UB];
    double *s_man = reinterpret_cast<double*>(s_union);
    int    *s_exp = reinterpret_cast<int*>(s_union + NSUB * WGS * 8);
    int    *s_plc = reinterpret_cast<int*>(s_union);
    float  *s_gps = reinterpret_cast<float*>(s_union + (FAST ? 0 : NG * WGS * 4));
    uint8_t *s_plb = s_union + NG * WGS * 4;               // FAST only

    const int tid = threadIdx.x;
    const int is = blockIdx.x;
    const int S = P.n_smpl;
    const size_t Ss = (size_t)S;
    const int ngrp = GRP ? (P.n_grp > 1 ? P.n_grp : 1) : 1;         // GRP = false: launched only for a single group
    bcfgpu_call_site *cs = &P.out.site[is];

    int nals, unseen;
    if (P.msite) {
        nals = P.msite[is].n_alleles;
        unseen = P.msite[is].unseen > 0 ? P.msite[is].unseen : 0;     // vcfcall.c:1102-1111
        if (P.msite[is].ret < 0) nals = -1;                           // mpileup wrote no record here (an indel column without an ALT allele, bam2bcf.c:611)
    } else { nals = P.nals[is]; unseen = P.unseen[is]; }
    // (one site per workgroup: the same in every lane -- said so, these decide scalar branches instead of lane masks)
    nals = __builtin_amdgcn_readfirstlane(nals); unseen = __builtin_amdgcn_readfirstlane(unseen);
    if (P.msite && nals == -1) {                                      // nothing to call: the site's call record says "skipped", no error
        if ((MAXA == 3 || (P.small_too && NSUB == 15)) && tid == 0) write_skipped(cs, 0);
        return;
    }
    const int ngts = nals * (nals + 1) / 2;
    // A record outside what the planes can hold (mcall() itself takes up to 32 alleles, mcall.c:1539; B2B_MAX_ALLELES = 5 is
    // what mpileup writes): refused record by record, ret = -2, and the call as a whole reports BCFGPU_E_RANGE at the next sync.
    if (nals < 1 || nals > BCFGPU_MAX_ALLELES || ngts > P.n_gt_max || (P.ad && nals > P.n_al_max) || unseen < 0 || unseen >= nals) {
        if ((MAXA == 3 || (P.small_too && NSUB == 15)) && tid == 0) { write_skipped(cs, -2); atomicExch(P.err, BCFGPU_E_RANGE); }
        return;
    }
    // the instantiations share the grid: sites with <=3 alleles run in the small one, the rest in the general ones -- or, with
    // many samples (launch_mcall), in mcall_kernel<5, 15, ...> as well, and the small one is not launched
    if (P.small_too ? (nals <= 3 && NSUB == 25) : (nals <= 3) != (MAXA == 3)) return;
    if (BCFGPU_ABL(P, 8)) return;

    // record-loop prologue of vcfcall.c:1112-1115: with -v a REF-only record never reaches mcall()
    if ((P.call_flag & BCFGPU_CALL_VARONLY) && (nals == 1 || (nals == 2 && unseen > 0))) {
        if (tid == 0) write_skipped(cs, 0);
        return;
    }
    const double *s_pl2p = P.pl2p;
    __shared__ double s_p2[FAST ? 264 : 1];                   // 10^(-PL/10), PL = 0..255; [256] = 0: what a sample without data (or a slot past the genotypes) looks up; [257] = 1
    if constexpr (FAST) {
        // (the lane's four entries requested together: as a loop this was four round trips, one after the other)
        const double t0 = P.pl2p[tid], t1 = P.pl2p[tid + WGS], t2 = P.pl2p[tid + 2 * WGS], t3 = P.pl2p[tid + 3 * WGS];
        s_p2[tid] = t0; s_p2[tid + WGS] = t1; s_p2[tid + 2 * WGS] = t2; s_p2[tid + 3 * WGS] = t3;
        if (tid < 8) s_p2[256 + tid] = tid == 1 ? 1.0 : 0.0;
    }   // [257] = 1: the free slot of a sample without data (subset scan)
    // the subset scan notes which samples carry data, four of them a byte (bit j: sample 4i+j), for the genotypes of a site that
    // stays REF-only (below); 0x80: no group's scan came by (the planes are read then)
    constexpr int NZ_MAX_S = 4096;
    __shared__ __align__(4) uint8_t s_nz[FAST ? NZ_MAX_S / 4 : 4];
    if constexpr (FAST) { if (S <= NZ_MAX_S) for (int i = tid; i < (S + 15) / 16; i += WGS) reinterpret_cast<uint32_t*>(s_nz)[i] = 0x80808080u; }

    // The 5-allele instantiations split the sites by the number of subsets to visit (LDS for the running products): more
    // than 15 only when all five alleles have a non-zero frequency (5 + 10 + 10 subsets).  With sample groups the
    // frequencies are sequential float32 sums over all samples (below): the split is then by the number of alleles alone,
    // before those sums are formed -- the 25-subset instantiation takes every 5-allele site.
    if (MAXA == 5 && ngrp > 1 && (nals == 5) != (NSUB == 25)) return;
    // ---- allele-frequency set-up (mcall.c:1453-1535), sequential float32 ----
    if (tid == 0) {
        sh.als_new = 0; sh.early = 0; sh.prior_fail = 0;
        sh.max_qual = -HUGE_VAL; sh.ref_lk = -HUGE_VAL; sh.lk_sum = -HUGE_VAL;
        for (int i = 0; i < 5; ++i) { sh.ac[i] = 0; sh.als_map[i] = -1; }
    }
    if (ngrp == 1) {
        if (tid < 5) {
            float v = 0;
            if (tid < nals) v = P.msite ? P.msite[is].qsum[tid] : P.qs[(size_t)is * 5 + tid];
            s_gq[tid] = v;
        }
    } else if constexpr (GRP) {
        // the groups' allele-frequency sums from FORMAT/AD (or QS) were formed by grp_qsum_kernel (below): a kernel of its own --
        // chains of dependent float additions and the memory round trips that feed them -- at twice this kernel's occupancy
        for (int i = tid; i < ngrp * 5; i += WGS) s_gq[i] = P.grp_q[(size_t)is * ngrp * 5 + i];
    }
    __syncthreads();
    // -F AN,AC prior and normalisation: lane g handles group g (mcall.c:1506-1535)
    for (int g = tid; g < ngrp; g += WGS) {
        float *q = s_gq + g * 5;
        if (P.prior_an && P.prior_ac && P.prior_an[is] != MISSING) {
            const int an = P.prior_an[is];
            const int32_t *acv = P.prior_ac + (size_t)is * 4;
            int nac = 0;
            while (nac < 4 && acv[nac] != VEND) nac++;
            if (an > 0 && nac == nals - 1) {
                int gn = 0;
                if (ngrp == 1) gn = S; else for (int s = 0; s < S; ++s) gn += (GRP_OF(s) == g);
                int ac0 = an;
                for (int i = 0; i < nals - 1; i++) {
                    if (acv[i] == VEND) break;
                    if (acv[i] == MISSING) continue;
                    ac0 -= acv[i];
                    q[i + 1] = (float)(((double)q[i + 1] + 0.5 * acv[i]) / (gn + 0.5 * an));
                }
                if (ac0 < 0) sh.prior_fail = 1;
                q[0] = (float)(((double)q[0] + 0.5 * ac0) / (gn + 0.5 * an));
            }
        }
        float sum = 0;
        for (int i = 0; i < nals; i++) sum += q[i];
        if (sum != 0.f) for (int i = 0; i < nals; i++) q[i] /= sum;
    }
    __syncthreads();
    if (sh.prior_fail) {                  // error("Incorrect AN,AC values") in the reference
        if (tid == 0) write_skipped(cs, -1);
        return;
    }
    int *grp_als_tab = reinterpret_cast<int*>(s_gq + ngrp * 5);
    // The 5-allele instantiations split the sites by the number of subsets to visit (LDS for the running products): more
    // than 15 only when all five alleles have a non-zero frequency (5 + 10 + 10 subsets), in any group -- decided once
    // per site, before the groups are walked, so that exactly one instantiation takes the site.
    if (MAXA == 5 && ngrp == 1) {
        bool five = false;
        if (nals == 5)
            for (int g = 0; g < ngrp; ++g) {
                const float *q = s_gq + g * 5;
                five = five || (q[0] != 0.f && q[1] != 0.f && q[2] != 0.f && q[3] != 0.f && q[4] != 0.f);
            }
        if (five != (NSUB == 25)) return;
    }

    // ---- per group: mcall_find_best_alleles ----
    // BATCH (the matrix-core scan with sample groups): what a group costs beside its scan -- the enumeration of its subsets, the
    // logarithms of its row products, the maximum / log-sum-exp over its subsets -- is done for several groups at once, CPG
    // lanes a group (a 64-lane wavefront holds GB = 64 / CPG groups), and the coefficient matrix of a group is formed in
    // registers from the subset list instead of being staged through LDS: no workgroup barrier inside a batch but the two
    // around its last step.  (One group after the other, each with its own set-up and reductions, was 1 450 of a site's
    // instructions per group: profiles/r4_mcall_grp_probe.txt.)
    constexpr bool BATCH = FAST && GRP;
    const int CPG = ngrp == 1 ? 64 : (NSUB > 15 ? 32 : 16), GB = 64 / CPG;
    __shared__ uint32_t s_sid[BATCH ? 64 : 1];            // [GB][CPG] a group's subsets: ia | ib << 8 | ic << 16 (0xff: absent)
    __shared__ int s_nsub[BATCH ? 4 : 1], s_set[BATCH ? 4 : 1], s_rede2[BATCH ? 64 : 1], s_dipe[BATCH ? 4 : 1], s_rals[BATCH ? 4 : 1];
    __shared__ double s_red2[BATCH ? 64 : 1], s_dipm[BATCH ? 4 : 1], s_rmax[BATCH ? 4 : 1], s_rsum[BATCH ? 4 : 1], s_rref[BATCH ? 4 : 1], s_rqual[BATCH ? 4 : 1];
    for (int g = 0; g < ngrp; ++g) {
        if constexpr (BATCH) {
            if (g % GB == 0) {
                // the subsets of the batch's groups: lane (group, candidate of the canonical enumeration), compacted per group
                __syncthreads();
                const int gl = tid / CPG, c = tid % CPG, gg = g + gl;
                const float *qf = s_gq + (gg < ngrp ? gg : ngrp - 1) * 5;
                const int npair = nals > 1 ? nals * (nals - 1) / 2 : 0;
                const int ntrip = nals > 2 ? nals * (nals - 1) * (nals - 2) / 6 : 0;
                int ia = 0xff, ib = 0xff, ic = 0xff;
                bool valid = false;
                if (c < nals) { ia = c; valid = true; }
                else if (c < nals + npair) {
                    const int cc = c - nals;
                    ia = 1; while ((ia + 1) * ia / 2 <= cc) ia++;
                    ib = cc - ia * (ia - 1) / 2;
                    valid = qf[ia] != 0 && qf[ib] != 0;
                } else if (c < nals + npair + ntrip) {
                    int cc = c - nals - npair;
                    ia = 2; while ((ia + 1) * ia * (ia - 1) / 6 <= cc) ia++;
                    cc -= ia * (ia - 1) * (ia - 2) / 6;
                    ib = 1; while ((ib + 1) * ib / 2 <= cc) ib++;
                    ic = cc - ib * (ib - 1) / 2;
                    valid = qf[ia] != 0 && qf[ib] != 0 && qf[ic] != 0;
                }
                valid = valid && gg < ngrp;
                const unsigned long long bal = __ballot(valid);
                const unsigned long long seg = CPG == 64 ? bal : (bal >> (gl * CPG)) & ((1ull << CPG) - 1);
                if (valid) s_sid[gl * CPG + __popcll(seg & ((1ull << c) - 1))] = (uint32_t)ia | (uint32_t)ib << 8 | (uint32_t)ic << 16;
                if (c == 0) s_nsub[gl] = __popcll(seg);
                __syncthreads();
            }
        } else {
        __syncthreads();
        if (tid < 5) sh.qsum[tid] = s_gq[g * 5 + tid];
        __syncthreads();
        // the subsets that mcall.c:600-698 visits, in its order (singles, pairs, triples; alleles of zero frequency are
        // skipped), with their frequency products: candidate c of the canonical enumeration is decoded by lane c and
        // the surviving ones are compacted in order
        {
            const float *qf = sh.qsum;
            const int npair = nals > 1 ? nals * (nals - 1) / 2 : 0;
            const int ntrip = nals > 2 ? nals * (nals - 1) * (nals - 2) / 6 : 0;
            int ia = -1, ib = -1, ic = -1;
            bool valid = false;
            if (tid < nals) { ia = tid; valid = true; }
            else if (tid < nals + npair) {
                const int c = tid - nals;
                ia = 1; while ((ia + 1) * ia / 2 <= c) ia++;
                ib = c - ia * (ia - 1) / 2;
                valid = qf[ia] != 0 && qf[ib] != 0;
            } else if (tid < nals + npair + ntrip) {
                int c = tid - nals - npair;
                ia = 2; while ((ia + 1) * ia * (ia - 1) / 6 <= c) ia++;
                c -= ia * (ia - 1) * (ia - 2) / 6;
                ib = 1; while ((ib + 1) * ib / 2 <= c) ib++;
                ic = c - ib * (ib - 1) / 2;
                valid = qf[ia] != 0 && qf[ib] != 0 && qf[ic] != 0;
            }
            const unsigned long long bal = __ballot(valid);
            if (valid) {
                Subset &t = sh.sub[__popcll(bal & ((1ull << tid) - 1))];
                t.ia = ia; t.ib = ib; t.ic = ic;
                t.iaa = (ia + 1) * (ia + 2) / 2 - 1;
                t.ibb = t.icc = t.iab = t.iac = t.ibc = 0;
                t.fa = t.fa2 = 1; t.fb = t.fc = t.fb2 = t.fc2 = t.fab = t.fac = t.fbc = 0;
                if (ib >= 0 && ic < 0) {
                    t.ibb = (ib + 1) * (ib + 2) / 2 - 1; t.iab = t.iaa - ia + ib;
                    t.fa = (double)(qf[ia] / (qf[ia] + qf[ib])); t.fb = (double)(qf[ib] / (qf[ia] + qf[ib]));
                    t.fa2 = t.fa * t.fa; t.fb2 = t.fb * t.fb; t.fab = 2 * t.fa * t.fb;
                } else if (ic >= 0) {
                    t.ibb = (ib + 1) * (ib + 2) / 2 - 1; t.icc = (ic + 1) * (ic + 2) / 2 - 1;
                    t.iab = t.iaa - ia + ib; t.iac = t.iaa - ia + ic; t.ibc = t.ibb - ib + ic;
                    const float den = qf[ia] + qf[ib] + qf[ic];
                    t.fa = (double)(qf[ia] / den); t.fb = (double)(qf[ib] / den); t.fc = (double)(qf[ic] / den);
                    t.fa2 = t.fa * t.fa; t.fb2 = t.fb * t.fb; t.fc2 = t.fc * t.fc;
                    t.fab = 2 * t.fa * t.fb; t.fac = 2 * t.fa * t.fc; t.fbc = 2 * t.fb * t.fc;
                }
            }
            if (tid == 0) sh.nsub = __popcll(bal);
        }
        __syncthreads();
        }
        const int nsub = BATCH ? s_nsub[g % GB] : sh.nsub;
        int setbits = 0;
        if constexpr (FAST && NSUB <= 15) {
            // ---- subset scan, a lane per sample (sparse_scan above): every site but those of the 25-subset instantiation ----
            const float *qf = s_gq + g * 5;
            uint32_t nzm = 0;
            #pragma unroll
            for (int a = 0; a < 5; ++a) if (a < nals && qf[a] != 0.f) nzm |= 1u << a;
            nzm = __builtin_amdgcn_readfirstlane(nzm);
            const int nz = __popc(nzm);
            uint32_t permw = 0;
            {
                int n = 0;
                #pragma unroll
                for (int a = 0; a < 5; ++a) if (a < nals && ((nzm >> a) & 1)) { permw |= (uint32_t)a << (4 * n); ++n; }
                #pragma unroll
                for (int a = 0; a < 5; ++a) if (a < nals && !((nzm >> a) & 1)) { permw |= (uint32_t)a << (4 * n); ++n; }
            }
            const uint8_t *plb = reinterpret_cast<const uint8_t*>(P.pl) + (size_t)is * BCFGPU_MAX_PL * Ss;
            // a group's samples lie in [first, last] (grp_range_kernel): consecutive for populations listed one after the
            // other, and then the groups' scans together read every sample once
            const int s_last = (ngrp > 1 && P.grp_rng) ? P.grp_rng[3 * g + 1] : S;
            const int s_first = (ngrp > 1 && P.grp_rng) ? (min(P.grp_rng[3 * g], s_last) & ~3) : 0;
            double rm = 1.0, dm = 1.0; int re = 0, de = 0;
            bool f_single = false, f_pt = false;
            const bool note = S <= NZ_MAX_S;
            // (a range of at most 128 samples -- a small group, few samples -- is scanned a sample a lane)
            const bool narrow = s_last - s_first <= 2 * WGS;
            #define SCAN(NZ_) do { if (narrow) sparse_scan<NZ_, HAP, 1>(P, plb, S, nals, g, ngrp, s_first, s_last, qf, permw, s_p2, s_nz, note, rm, re, dm, de, f_single, f_pt); \
                                   else sparse_scan<NZ_, HAP, 4>(P, plb, S, nals, g, ngrp, s_first, s_last, qf, permw, s_p2, s_nz, note, rm, re, dm, de, f_single, f_pt); } while (0)
            switch (nz) {
            case 4:  SCAN(4); break;
            case 3:  SCAN(3); break;
            case 2:  SCAN(2); break;
            default: SCAN(1); break;
            }
            #undef SCAN
            // slot -> the row of the subset list: single allele x -> its own index, pairs and triples in visiting order, the sums last
            const int np = nz * (nz - 1) / 2, nt = nz * (nz - 1) * (nz - 2) / 6;
            const int slot = tid >> 2;
            int row = -1;
            if (slot < 5) { if (slot < nals) row = (int)((permw >> (4 * slot)) & 7); }
            else if (slot < 5 + np + nt) row = nals + slot - 5;
            else if (slot == 5 + np + nt) row = nsub;
            if ((tid & 3) == 0 && row >= 0) {
                if constexpr (BATCH) { s_red2[(g % GB) * CPG + row] = rm; s_rede2[(g % GB) * CPG + row] = re; }
                else { sh.red[row] = rm; sh.rede[row] = re; }      // log taken below, one row per lane
            }
            if (HAP && tid == 0) {
                if constexpr (BATCH) { s_dipm[g % GB] = dm; s_dipe[g % GB] = de; }
                else { sh.red[31] = dm; sh.rede[31] = de; }        // (at most 16 rows are in use)
            }
            if constexpr (!BATCH) {
            __syncthreads();
            if (tid <= nsub || (HAP && tid == 31)) sh.red[tid] = log(sh.red[tid]) + (double)sh.rede[tid] * 0.693147180559945309417232121458;
            __syncthreads();
            if (HAP && tid == 0) sh.red_dip = sh.red[31];
            }
            // the rows some sample entered: the single alleles and the sums with any sample that has data, pairs and triples with
            // one of ploidy 1 or 2 among them
            const int single_rows = ((1 << nals) - 1) | (1 << nsub);
            setbits = (f_single ? single_rows : 0) | (f_pt ? ((1 << nsub) - 1) & ~((1 << nals) - 1) : 0);
        } else if constexpr (FAST) {
            // ---- subset scan on the matrix cores ----
            // Rows = subsets (+ an all-ones row for the samples' normalisation sums), columns = samples, k = genotypes.
            // Haploid samples use a second coefficient matrix (f_a P(aa) + f_b P(bb) [+ f_c P(cc)], mcall.c:643,688);
            // samples of other groups, of ploidy 0 (pairs/triples only) or without data contribute nothing.
            const int col = tid & 15, kq = tid >> 4;
            // Slot k = 4 kk + kq of the products' inner dimension -> genotype: the homozygous genotypes first, then the heterozygous
            // ones in index order, nothing past ngts.  The haploid matrix has coefficients on the homozygous genotypes only, so its
            // product ends after the first nals slots, and the diploid one after ngts of the 16: a 4-allele site (10 genotypes)
            // takes 3 + 1 matrix instructions per 16 samples instead of 4 + 4.
            int gpk[4];
            #pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int sl = 4 * kk + kq;
                gpk[kk] = sl < nals ? (sl + 1) * (sl + 2) / 2 - 1 : sl < ngts ? (int)((0xDCBA876431ull >> (4 * (sl - nals))) & 15) : -1;
            }
            // Without a ploidy array, slot ngts -- the first one past the genotypes, always inside the last block of four the products
            // visit (ngts is 1, 3, 6, 10 or 15) -- carries the coefficient 1 in every row and the value 1 for a sample that is not
            // scanned (no data, another group), 0 for one that is: every product of such a sample is exactly 1, the products of the
            // others are what they were (+ 0.0), and the running products take them all without a test.
            uint32_t nodata[4];
            #pragma unroll
            for (int kk = 0; kk < 4; ++kk) nodata[kk] = (!HAP && 4 * kk + kq == ngts) ? 257u : 256u;
            double a[TILES][4], ah[HAP ? TILES : 1][4];
            if constexpr (BATCH) {
                // the lane's elements of the coefficient matrices, rows col (+ 16 t), genotypes 4 kk + kq, straight from the subset
                // list: genotype k = (x <= y) has f_x f_y (twice that when x != y) if both alleles are in the subset; the
                // frequencies inside the subset are float quotients, widened (mcall.c:629-630, 671-673); the haploid matrix has
                // f_x on the homozygous genotypes (mcall.c:643, 688); row nsub is the all-ones row of the normalisation sums
                const float *qf = s_gq + g * 5;
                #pragma unroll
                for (int t = 0; t < TILES; ++t) {
                    const int row = t * 16 + col;
                    const uint32_t sid = row < nsub ? s_sid[(g % GB) * CPG + row] : 0xffffffu;
                    const int ia = (int)(sid & 0xff), ib = (int)((sid >> 8) & 0xff), ic = (int)((sid >> 16) & 0xff);
                    double fa = 1.0, fb = 0.0, fc = 0.0;
                    if (row < nsub && ib != 0xff) {
                        if (ic == 0xff) { const float den = qf[ia] + qf[ib]; fa = (double)(qf[ia] / den); fb = (double)(qf[ib] / den); }
                        else { const float den = qf[ia] + qf[ib] + qf[ic]; fa = (double)(qf[ia] / den); fb = (double)(qf[ib] / den); fc = (double)(qf[ic] / den); }
                    }
                    #pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const int k = gpk[kk] >= 0 ? gpk[kk] : 15;          // (genotype 15 = (0,5): an allele no subset holds)
                        const int y = (int)((0x5444443333222110ull >> (4 * k)) & 15), x = k - y * (y + 1) / 2;
                        const double fx = x == ia ? fa : x == ib ? fb : x == ic ? fc : 0.0, fy = y == ia ? fa : y == ib ? fb : y == ic ? fc : 0.0;
                        double av = 0.0, hv = 0.0;
                        if (row < nsub) { av = x == y ? fx * fx : 2 * fy * fx; hv = x == y ? fx : 0.0; }
                        else if (row == nsub) av = gpk[kk] >= 0 ? 1.0 : 0.0;
                        a[t][kk] = nodata[kk] == 257u ? 1.0 : av;
                        if (HAP) ah[t][kk] = hv;
                    }
                }
            } else {
            double *s_coef = reinterpret_cast<double*>(s_union);            // [TILES*16 rows][16 genotypes]
            double *s_coefh = s_coef + TILES * 256;                         // HAP: the haploid rows
            for (int i = tid; i < TILES * 256 * (HAP ? 2 : 1); i += WGS) s_coef[i] = 0.0;
            __syncthreads();
            if (tid < nsub) {
                const Subset &u = sh.sub[tid];
                double *row = s_coef + tid * 16;
                if (u.ib < 0) row[u.iaa] = 1.0;
                else {
                    row[u.iaa] = u.fa2; row[u.ibb] = u.fb2; row[u.iab] = u.fab;
                    if (u.ic >= 0) { row[u.icc] = u.fc2; row[u.iac] = u.fac; row[u.ibc] = u.fbc; }
                }
                if (HAP) {
                    double *rh = s_coefh + tid * 16;
                    rh[u.iaa] = u.fa;
                    if (u.ib >= 0) rh[u.ibb] = u.fb;
                    if (u.ic >= 0) rh[u.icc] = u.fc;
                }
            }
            if (tid < ngts) s_coef[nsub * 16 + tid] = 1.0;                  // the sum row
            __syncthreads();
            #pragma unroll
            for (int t = 0; t < TILES; ++t)
                #pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    a[t][kk] = gpk[kk] >= 0 ? s_coef[(t * 16 + col) * 16 + gpk[kk]] : nodata[kk] == 257u ? 1.0 : 0.0;
                    if (HAP) ah[t][kk] = gpk[kk] >= 0 ? s_coefh[(t * 16 + col) * 16 + gpk[kk]] : 0.0;
                }
            }
            // this lane accumulates rows kq + 4r (+16t) over the sample columns col, col+16, ...
            double man[TILES][4]; int ex[TILES][4];
            #pragma unroll
            for (int t = 0; t < TILES; ++t)
                #pragma unroll
                for (int r = 0; r < 4; ++r) { man[t][r] = 1.0; ex[t][r] = 0; }
            double sdm = 1.0; int sde = 0;                                   // HAP: the sum row over ploidy-1/2 samples
            int singles = 0;                                                // bit t*4+r: a row without ploidy term
            #pragma unroll
            for (int t = 0; t < TILES; ++t)
                #pragma unroll
                for (int r = 0; r < 4; ++r) { const int row = t * 16 + kq + 4 * r; if (row < nals || row == nsub) singles |= 1 << (t * 4 + r); }
            const uint8_t *plb = reinterpret_cast<const uint8_t*>(P.pl) + (size_t)is * BCFGPU_MAX_PL * Ss;
            // A lane takes 16 consecutive samples of a 256-sample block: the sixteen 4-byte loads of its four plane groups (and the
            // ploidy / group words) are all issued before the first use, so a block costs one memory round trip -- with
            // four wavefronts per SIMD the scan is bound by such round trips.  Word q of the block holds the samples
            // sb+4q .. sb+4q+3; one matrix product covers the 16 lanes' samples sb+4q+j.
            // (few samples: 4*nq consecutive samples per lane with nq = 2 or 1, so that the 16 lanes' columns stay filled)
            // a group's samples lie in [first, last] (grp_range_kernel): consecutive for populations listed one after the
            // other, and then the groups' scans together read every sample once
            const int s_last = (ngrp > 1 && P.grp_rng) ? P.grp_rng[3 * g + 1] : S;
            const int s_first = (ngrp > 1 && P.grp_rng) ? (min(P.grp_rng[3 * g], s_last) & ~3) : 0;
            const int span = s_last - s_first;
            const int nq = span > 128 ? 4 : span > 64 ? 2 : 1;
            for (int s0 = s_first; s0 < (BCFGPU_ABL(P, 16) ? 0 : s_last); s0 += 64 * nq) {
                const int sb0 = s0 + 4 * nq * col;
                uint32_t wq[4][4], pwq[4], gmq[4];
                #pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q >= nq) break;
                    const int sb = sb0 + 4 * q, rem = S - sb;
                    uint32_t pw = 0x02020202u, gmask = 0xf;                 // ploidy bytes and group membership of the 4 samples
                    if (rem <= 0) gmask = 0;
                    else if (HAP || ngrp > 1) {
                        if (rem >= 4) {
                            if (HAP) __builtin_memcpy(&pw, P.ploidy + sb, 4);
                            if (ngrp > 1) {
                                int gv[4];
                                __builtin_memcpy(gv, P.grp + sb, 16);
                                gmask = (gv[0] == g ? 1u : 0u) | (gv[1] == g ? 2u : 0u) | (gv[2] == g ? 4u : 0u) | (gv[3] == g ? 8u : 0u);
                            }
                        } else {
                            #pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (j < rem) {
                                    if (HAP) pw = (pw & ~(0xffu << (8 * j))) | (uint32_t)P.ploidy[sb + j] << (8 * j);
                                    if (ngrp > 1 && GRP_OF(sb + j) != g) gmask &= ~(1u << j);
                                }
                        }
                    }
                    pwq[q] = pw; gmq[q] = gmask;
                    // the plane of slot 4kk+kq, samples sb .. sb+3 as one word; samples past the end read as "no data"
                    #pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const int k = gpk[kk];
                        uint32_t v = 0;
                        if (k >= 0 && rem > 0) {
                            const uint8_t *src = plb + (size_t)k * Ss + sb;
                            if (rem >= 4) __builtin_memcpy(&v, src, 4);
                            else { v = src[0]; if (rem > 1) v |= (uint32_t)src[1] << 8; if (rem > 2) v |= (uint32_t)src[2] << 16; }
                        }
                        wq[q][kk] = v;
                    }
                }
                #pragma unroll
                for (int q = 0; q < 4; ++q) {
                const uint32_t pw = pwq[q], gmask = gmq[q];
                const uint32_t (&w)[4] = wq[q];
                if (q >= nq) break;
                if (ngrp > 1 && !__any(gmask != 0)) continue;              // no sample of this group in this word of the 16 lanes
                // set_pdg: a sample whose PLs are all 0 (sum == n_gt) carries no data and is skipped (mcall.c:529-535)
                uint32_t any = w[0] | w[1] | w[2] | w[3];
                any |= __shfl_xor(any, 16);
                any |= __shfl_xor(any, 32);
                if (kq == 0 && S <= NZ_MAX_S && sb0 + 4 * q < S)
                    s_nz[(sb0 + 4 * q) >> 2] = (uint8_t)(((any & 0xffu) ? 1u : 0u) | ((any & 0xff00u) ? 2u : 0u) | ((any & 0xff0000u) ? 4u : 0u) | ((any & 0xff000000u) ? 8u : 0u));
                #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool has = ((any >> (8 * j)) & 0xff) != 0 && ((gmask >> j) & 1);
                    const int pd = (int)((pw >> (8 * j)) & 0xff);
                    if (!HAP && has) setbits = (1 << (TILES * 4)) - 1;       // (a scanned sample's products are all positive)
                    double b[4];
                    #pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        b[kk] = s_p2[has ? (gpk[kk] >= 0 ? (w[kk] >> (8 * j)) & 0xff : 256u) : nodata[kk]];    // (an unconditional look-up: no branch around the read; the next sample's look-ups requested a sample ahead: slower)
                    #pragma unroll
                    for (int t = 0; t < TILES; ++t) {
                        d4_t d = {0., 0., 0., 0.}, dh = {0., 0., 0., 0.};
                        #pragma unroll
                        for (int kk = 0; kk < 4; ++kk) if (4 * kk < ngts) d = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t][kk], b[kk], d, 0, 0, 0);
                        if (HAP) {
                            #pragma unroll
                            for (int kk = 0; kk < 2; ++kk) if (4 * kk < nals) dh = __builtin_amdgcn_mfma_f64_16x16x4f64(ah[t][kk], b[kk], dh, 0, 0, 0);
                        }
                        #pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            double v = d[r];
                            if (HAP && !((singles >> (t * 4 + r)) & 1)) v = pd == 2 ? d[r] : pd == 1 ? dh[r] : 0.0;
                            if constexpr (!HAP) man[t][r] *= v;
                            else if (v != 0.0) { man[t][r] *= v; setbits |= 1 << (t * 4 + r); }
                            if (HAP && t * 16 + kq + 4 * r == nsub && (pd == 1 || pd == 2) && d[r] != 0.0) sdm *= d[r];
                        }
                    }
                    // (renormalised once per word of four samples: four factors are at least 1e-102 together, far from the range's end, and
                    // splitting off a power of two is exact whenever it is done; every other sample was 1 % slower, every eighth no faster)
                    if (j == 3) {
                        #pragma unroll
                        for (int t = 0; t < TILES; ++t)
                            #pragma unroll
                            for (int r = 0; r < 4; ++r) { ex[t][r] += frexp_exp(man[t][r]); man[t][r] = frexp_mant(man[t][r]); }
                        if (HAP) { sde += frexp_exp(sdm); sdm = frexp_mant(sdm); }
                    }
                }
                }
            }
            // product over the 16 sample columns of each row, then rows -> sh.red[]
            int rowbits = 0;
            #pragma unroll
            for (int t = 0; t < TILES; ++t)
                #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double m = man[t][r]; int e = ex[t][r]; int f = (setbits >> (t * 4 + r)) & 1;
                    row16_product(m, e, f);
                    const int row = t * 16 + kq + 4 * r;
                    if (col == 0 && row <= nsub) {
                        if constexpr (BATCH) { s_red2[(g % GB) * CPG + row] = m; s_rede2[(g % GB) * CPG + row] = e; }
                        else { sh.red[row] = m; sh.rede[row] = e; }  // log taken below, one row per lane
                        if (f) rowbits |= 1 << row;
                    }
                }
            if (HAP) {
                double m = sdm; int e = sde, f_ = 0;
                row16_product(m, e, f_);
                if (col == 0 && kq == (nsub & 3)) {
                    if constexpr (BATCH) { s_dipm[g % GB] = m; s_dipe[g % GB] = e; }
                    else { sh.red[31] = m; sh.rede[31] = e; }      // (at most 26 rows are in use)
                }
            }
            if constexpr (!BATCH) {
            __syncthreads();
            if (tid <= nsub || (HAP && tid == 31)) sh.red[tid] = log(sh.red[tid]) + (double)sh.rede[tid] * 0.693147180559945309417232121458;
            __syncthreads();
            if (HAP && tid == 0) sh.red_dip = sh.red[31];
            }
            setbits = rowbits;
        } else {
        for (int t = 0; t < nsub; ++t) { s_man[t * WGS + tid] = 1.0; s_exp[t * WGS + tid] = 0; }
        // (a group no sample belongs to keeps {INT_MAX, 0, 0}: an empty range, not a start that overflows when tid is added)
        const int g_last = (ngrp > 1 && P.grp_rng) ? P.grp_rng[3 * g + 1] : S;
        const int g_first = (ngrp > 1 && P.grp_rng) ? min(P.grp_rng[3 * g], g_last) : 0;
        for (int s = g_first + tid; s < (BCFGPU_ABL(P, 16) ? 0 : g_last); s += WGS) {
            if (ngrp > 1 && GRP_OF(s) != g) continue;
            int pl[NG]; double pdg[NG];
            load_pl<NG>(P, is, s, ngts, pl);
            const double psum = set_pdg_one<NG>(s_pl2p, pl, pdg, ngts, nals, unseen, s_fill + tid);
            // the subset likelihoods only feed log-sums (QUAL, 1e-4 contract): one reciprocal instead of n_gt divisions
            const double rsum = psum != 0.0 ? 1.0 / psum : 0.0;
            #pragma unroll
            for (int k = 0; k < NG; ++k) s_pdg[k * WGS + tid] = pdg[k];
            const int ploidy = P.ploidy ? P.ploidy[s] : 2;
            for (int t = 0; t < nsub; ++t) {
                const Subset &u = sh.sub[t];
                double val;
                const double paa = s_pdg[u.iaa * WGS + tid];
                if (u.ib < 0) val = paa;                                   // single allele: no ploidy term (mcall.c:607-611)
                else {
                    const double pbb = s_pdg[u.ibb * WGS + tid];
                    if (u.ic < 0) {
                        if (ploidy == 2) val = u.fa2 * paa + u.fb2 * pbb + u.fab * s_pdg[u.iab * WGS + tid];
                        else if (ploidy == 1) val = u.fa * paa + u.fb * pbb;
                        else val = 0;
                    } else {
                        const double pcc = s_pdg[u.icc * WGS + tid];
                        if (ploidy == 2)
                            val = u.fa2 * paa + u.fb2 * pbb + u.fc2 * pcc + u.fab * s_pdg[u.iab * WGS + tid]
                                + u.fac * s_pdg[u.iac * WGS + tid] + u.fbc * s_pdg[u.ibc * WGS + tid];
                        else if (ploidy == 1) val = u.fa * paa + u.fb * pbb + u.fc * pcc;
                        else val = 0;
                    }
                }
                if (val != 0.0) {
                    const double m = s_man[t * WGS + tid] * (val * rsum);
                    s_man[t * WGS + tid] = frexp_mant(m);
                    s_exp[t * WGS + tid] += frexp_exp(m);
                    setbits |= 1 << t;
                }
            }
        }
        for (int t = 0; t < nsub; ++t) {
            // wave product of the mantissas (renormalised at every step) and sum of the exponents
            double m = s_man[t * WGS + tid]; int e = s_exp[t * WGS + tid];
            #pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double mm = m * __shfl_xor(m, o);
                e += __shfl_xor(e, o) + frexp_exp(mm);
                m = frexp_mant(mm);
            }
            if (tid == 0) sh.red[t] = log(m) + (double)e * 0.693147180559945309417232121458;
        }
        }
        setbits = wor(setbits);
        if constexpr (BATCH) {
            if (tid == 0) s_set[g % GB] = setbits;
            if (g % GB != GB - 1 && g != ngrp - 1) continue;             // the batch goes on: its groups are finished together
            // ---- the batch's groups side by side, CPG lanes each: logarithms of the row products, UPDATE_MAX_LKs over the
            // visited subsets (mcall.c:582-585, 600-698; the first maximum in visiting order wins), one log-sum-exp for lk_sum ----
            __syncthreads();
            const int g0 = g - g % GB, gl = tid / CPG, c = tid % CPG;
            const bool live = g0 + gl <= g;
            const int ns = live ? s_nsub[gl] : 0;
            const double LN2 = 0.693147180559945309417232121458;
            double lg = 0.0, dip = 0.0;
            if (live && c <= ns) lg = log(s_red2[gl * CPG + c]) + (double)s_rede2[gl * CPG + c] * LN2;
            if (HAP && live) dip = log(s_dipm[gl]) + (double)s_dipe[gl] * LN2;
            const double rsum = __shfl(lg, gl * CPG + ns);               // the row of the normalisation sums
            const int set = live ? s_set[gl] : 0;
            const double theta = P.theta;
            double lk_tot = -HUGE_VAL, lk_add = -HUGE_VAL, v = 0.0;
            int als = 0;
            if (live && c < ns) {
                const uint32_t sid = s_sid[gl * CPG + c];
                const int ia = (int)(sid & 0xff), ib = (int)((sid >> 8) & 0xff), ic = (int)((sid >> 16) & 0xff);
                // divide out the product of the normalisation sums of the samples that contributed to this row
                v = lg - ((HAP && ib != 0xff) ? dip : rsum);
                als = 1 << ia;
                if (ia != 0) v += theta;
                if (ib != 0xff) { als |= 1 << ib; if (ib != 0) v += theta; }
                if (ic != 0xff) { als |= 1 << ic; if (ic != 0) v += theta; }
                if ((set >> c) & 1) { lk_tot = v; if (als != 1) lk_add = v; }
            }
            const double refc = __shfl(v, gl * CPG);                     // subset 0 is the REF-only one
            double m = lk_tot; int mi = c;
            double ma = lk_add;
            for (int o = CPG >> 1; o > 0; o >>= 1) {
                const double om = __shfl_xor(m, o); const int oi = __shfl_xor(mi, o);
                if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
                const double oa = __shfl_xor(ma, o);
                ma = oa > ma ? oa : ma;
            }
            int max_als_b = __shfl(als, gl * CPG + mi);
            if (m == -HUGE_VAL) max_als_b = 0;
            double e = lk_add != -HUGE_VAL ? exp(lk_add - ma) : 0.0;
            for (int o = CPG >> 1; o > 0; o >>= 1) e += __shfl_xor(e, o);
            const double lk_sum_b = ma != -HUGE_VAL ? ma + log(e) : -HUGE_VAL;
            if (live && c == 0) {
                s_rmax[gl] = m; s_rsum[gl] = lk_sum_b; s_rref[gl] = refc; s_rals[gl] = max_als_b;
                s_rqual[gl] = m != -HUGE_VAL ? -4.343 * (refc - lse2(lk_sum_b, refc)) : 0.0;
            }
            __syncthreads();
            if (tid == 0) {                                              // in group order, as the reference walks them (mcall.c:1546-1560)
                for (int b = 0; b <= g - g0; ++b) {
                    const int max_als2 = s_rals[b];
                    int n = 0;
                    for (int i = 0; i < nals; i++) if (max_als2 & 1 << i) n++;
                    sh.als_new |= max_als2;
                    if (s_rmax[b] != -HUGE_VAL && sh.max_qual < s_rqual[b]) { sh.max_qual = s_rqual[b]; sh.lk_sum = s_rsum[b]; sh.ref_lk = s_rref[b]; }
                    grp_als_tab[(g0 + b) * 2] = max_als2;
                    grp_als_tab[(g0 + b) * 2 + 1] = n;
                }
            }
            continue;
        }
        if (tid == 0) sh.redset = setbits;
        __syncthreads();
        // UPDATE_MAX_LKs over the visited subsets (mcall.c:582-585, 600-698), lane t = subset t: the first maximum in
        // visiting order wins, and lk_sum (which only feeds QUAL) is one log-sum-exp instead of a chain of pairwise ones
        int max_als = 0;
        double ref_lk, max_lk, lk_sum;
        {
            const int set = sh.redset;
            const double theta = P.theta;
            double lk_tot = -HUGE_VAL, lk_add = -HUGE_VAL;
            int als = 0;
            if (tid < nsub) {
                const Subset &u = sh.sub[tid];
                double v = sh.red[tid];
                // divide out the product of the normalisation sums of the samples that contributed to this row
                if (FAST) v -= (HAP && u.ib >= 0) ? sh.red_dip : sh.red[nsub];
                als = 1 << u.ia;
                if (u.ib < 0) { if (u.ia != 0) v += theta; }
                else {
                    als |= 1 << u.ib;
                    if (u.ia != 0) v += theta;
                    if (u.ib != 0) v += theta;
                    if (u.ic >= 0) { als |= 1 << u.ic; if (u.ic != 0) v += theta; }
                }
                if (tid == 0) sh.ref_cur = v;               // subset 0 is the REF-only one
                if ((set >> tid) & 1) { lk_tot = v; if (als != 1) lk_add = v; }
            }
            double m = lk_tot; int mi = tid;
            double ma = lk_add;
            #pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double om = __shfl_xor(m, o); const int oi = __shfl_xor(mi, o);
                if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
                const double oa = __shfl_xor(ma, o);
                ma = oa > ma ? oa : ma;
            }
            max_lk = m;
            max_als = __shfl(als, mi);
            if (m == -HUGE_VAL) max_als = 0;
            double e = lk_add != -HUGE_VAL ? exp(lk_add - ma) : 0.0;
            e = wsum(e);
            lk_sum = ma != -HUGE_VAL ? ma + log(e) : -HUGE_VAL;
            __syncthreads();
            ref_lk = sh.ref_cur;
        }
        if (tid == 0) {
            int n = 0;
            for (int i = 0; i < nals; i++) if (max_als & 1 << i) n++;
            sh.als_new |= max_als;
            if (max_lk != -HUGE_VAL) {
                const double qual = -4.343 * (ref_lk - lse2(lk_sum, ref_lk));
                if (sh.max_qual < qual) { sh.max_qual = qual; sh.lk_sum = lk_sum; sh.ref_lk = ref_lk; }
            }
            grp_als_tab[g * 2] = max_als;
            grp_als_tab[g * 2 + 1] = n;
        }
    }
    __syncthreads();

    // ---- allele trimming, output set-up (mcall.c:1563-1577) ----
    if (tid == 0) {
        if (!(sh.als_new & 1)) sh.als_new |= 1;
        sh.is_variant = sh.als_new == 1 ? 0 : 1;
        if ((P.call_flag & BCFGPU_CALL_VARONLY) && !sh.is_variant) sh.early = 1;
        int nn = 0;
        for (int i = 0; i < nals; i++) {
            if (i > 0 && i == unseen) continue;
            if (P.call_flag & BCFGPU_CALL_KEEPALT) sh.als_new |= 1 << i;
            if (sh.als_new & (1 << i)) nn++;
        }
        sh.nals_new = nn;
        int nout = 0;
        for (int i = 0; i < nals; i++) sh.als_map[i] = (sh.als_new & (1 << i)) ? nout++ : -1;
        int k = 0, l = 0;
        for (int i = 0; i < nals; i++)
            for (int j = 0; j <= i; j++) {
                if ((sh.als_new & (1 << i)) && (sh.als_new & (1 << j))) sh.pl_map[k++] = l;
                l++;
            }
    }
    __syncthreads();
    if (sh.early) {
        if (tid == 0) write_skipped(cs, 0);
        return;
    }
    const int als_new = sh.als_new, nals_new = sh.nals_new, is_variant = sh.is_variant;
    const int ngts_new = nals_new * (nals_new + 1) / 2;
    const bool ref_only = (als_new == 1);
    const bool want_gqgp = (P.output_tags & (BCFGPU_CALL_FMT_GQ | BCFGPU_CALL_FMT_GP)) != 0;

    // ---- genotypes (mcall_set_ref_genotypes / mcall_call_genotypes) + PL trimming ----
    int ac_loc[5] = {0, 0, 0, 0, 0};
    const int ogt = P.out_n_gt_max;
    {
    const uint8_t *plb2 = reinterpret_cast<const uint8_t*>(P.pl) + (size_t)is * BCFGPU_MAX_PL * Ss;
    // FAST: the PL bytes of the lane's next sample are requested before the current one is worked on (the loop is a
    // chain of memory round trips otherwise: 16 of them for 1000 samples)
    uint32_t pl_nx[FAST ? NG : 1];
    auto fetch_pl = [&](int s) {
        if constexpr (FAST) {
            #pragma unroll
            for (int k = 0; k < NG; ++k) pl_nx[k] = (k < ngts && s < S) ? (uint32_t)plb2[(size_t)k * Ss + s] : 0u;
        }
    };
    // A site that stays REF-only (most of them) needs of its samples only "any data at all?" (mcall_set_ref_genotypes,
    // mcall.c:529-541): four samples a lane and a 4-byte store per GT plane, from the subset scan's notes or (sample groups)
    // from a 4-byte load per PL plane -- four trips to memory for 1000 samples where the general loop below makes sixteen
    bool gt_done = BCFGPU_ABL(P, 32);
    if constexpr (FAST) {
        int8_t *gt0 = P.out.gt + (size_t)is * 2 * Ss;
        if (!gt_done && !is_variant && ref_only && !(S & 3) && !(((uintptr_t)plb2 | (uintptr_t)gt0) & 3)) {
            for (int s4 = tid * 4; s4 < S; s4 += WGS * 4) {
                uint32_t any = 0;
                const uint32_t nib = S <= NZ_MAX_S ? s_nz[s4 >> 2] : 0x80u;
                if (!(nib & 0x80u)) {                            // the scan's note: no PL is read again
                    any = (nib & 1u) | (nib & 2u) << 7 | (nib & 4u) << 14 | (nib & 8u) << 21;
                } else {
                    #pragma unroll
                    for (int k = 0; k < NG; ++k)
                        if (k < ngts) any |= *reinterpret_cast<const uint32_t*>(plb2 + (size_t)k * Ss + s4);
                }
                uint32_t w = 0, w1 = 0, pw = 0x02020202u;
                if (HAP && P.ploidy) __builtin_memcpy(&pw, P.ploidy + s4, 4);
                #pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint32_t pd = (pw >> (8 * b)) & 0xFFu;
                    const bool called = ((any >> (8 * b)) & 0xFFu) && pd;
                    if (called) ac_loc[0] += (int)pd;
                    const uint32_t g0 = called ? 0u : (uint32_t)(uint8_t)BCFGPU_GT_MISSING;
                    w |= g0 << (8 * b);
                    w1 |= (pd == 2 ? g0 : (uint32_t)(uint8_t)BCFGPU_GT_VECTOR_END) << (8 * b);
                }
                *reinterpret_cast<uint32_t*>(gt0 + s4) = w;
                *reinterpret_cast<uint32_t*>(gt0 + Ss + s4) = w1;
            }
            gt_done = true;
        }
    }
    if (!gt_done) fetch_pl(tid);
    for (int s = tid; s < (gt_done ? 0 : S); s += WGS) {
        const int ploidy = (FAST && !HAP) ? 2 : (P.ploidy ? P.ploidy[s] : 2);    // FAST without HAP is launched only without a ploidy array
        // P(D|G) = raw/psum is formed lazily below, with the same division the reference performs (bit-exact genotypes)
        double psum = 0;
        bool allzero = true;
        if constexpr (FAST) {
            // u8 PLs of the fused pipeline (no missing values): set_pdg is a table lookup per byte, so only the PL
            // bytes are staged (run-time genotype indices)
            uint32_t anynz = 0;
            #pragma unroll
            for (int k = 0; k < NG; ++k) {
                if (k < ngts) {
                    const uint32_t v = pl_nx[k];
                    s_plb[k * WGS + tid] = (uint8_t)v;
                    psum += s_p2[v];
                    anynz |= v;
                }
                if (want_gqgp) s_gps[k * WGS + tid] = 0.f;
            }
            allzero = anynz == 0;                            // sum == n_gt: no data (mcall.c:529-535)
            fetch_pl(s + WGS);
        } else {
            int pl[NG]; double pdg[NG];
            load_pl<NG>(P, is, s, ngts, pl);
            psum = set_pdg_one<NG>(s_pl2p, pl, pdg, ngts, nals, unseen, s_fill + tid);
            #pragma unroll
            for (int k = 0; k < NG; ++k) {
                if (k < ngts && pdg[k] != 0.0) allzero = false;
                s_pdg[k * WGS + tid] = pdg[k]; s_plc[k * WGS + tid] = pl[k];
                if (want_gqgp) s_gps[k * WGS + tid] = 0.f;
            }
            if (psum == 0.0) allzero = true;
        }
        auto pdg_at = [&](int i) -> double { if constexpr (FAST) return s_p2[s_plb[i * WGS + tid]]; else return s_pdg[i * WGS + tid]; };
        auto plc_at = [&](int i) -> int { if constexpr (FAST) return (int)s_plb[i * WGS + tid]; else return s_plc[i * WGS + tid]; };
        int g0, g1, gq = 0, gnals = 0;
        if (!is_variant) {
            if (allzero || !ploidy) { g0 = BCFGPU_GT_MISSING; g1 = ploidy == 2 ? BCFGPU_GT_MISSING : BCFGPU_GT_VECTOR_END; }
            else { g0 = 0; g1 = ploidy == 2 ? 0 : BCFGPU_GT_VECTOR_END; ac_loc[0] += ploidy; }
        } else {
            const int g = ngrp > 1 ? GRP_OF(s) : 0;
            const int gals = grp_als_tab[g * 2];
            gnals = grp_als_tab[g * 2 + 1];
            const float *gq5 = s_gq + g * 5;
            if (!ploidy) { g0 = BCFGPU_GT_MISSING; g1 = BCFGPU_GT_VECTOR_END; s_gps[tid] = -1; }
            else if (allzero) { g0 = BCFGPU_GT_MISSING; g1 = ploidy == 2 ? BCFGPU_GT_MISSING : BCFGPU_GT_VECTOR_END; s_gps[tid] = -1; }
            else {
                g0 = 0; g1 = ploidy == 2 ? 0 : BCFGPU_GT_VECTOR_END;
                double best_lk = 0;
                for (int ia = 0; ia < nals; ++ia) {
                    if (!(gals & 1 << ia)) continue;
                    const int iaa = (ia + 1) * (ia + 2) / 2 - 1;
                    const double p = pdg_at(iaa) / psum;
                    const double lk = ploidy == 2 ? p * gq5[ia] * gq5[ia] : p * gq5[ia];
                    const int am = sh.als_map[ia];
                    const int igt = ploidy == 2 ? a2gt(am, am) : am;
                    s_gps[igt * WGS + tid] = (float)lk;
                    if (best_lk < lk) { best_lk = lk; g0 = am; }
                }
                if (ploidy == 2) {
                    g1 = g0;
                    for (int ia = 1; ia < nals; ++ia) {
                        if (!(gals & 1 << ia)) continue;
                        const int iaa = (ia + 1) * (ia + 2) / 2 - 1;
                        for (int ib = 0; ib < ia; ++ib) {
                            if (!(gals & 1 << ib)) continue;
                            const int iab = iaa - ia + ib;
                            const double lk = 2 * (pdg_at(iab) / psum) * gq5[ia] * gq5[ib];
                            const int igt = a2gt(sh.als_map[ia], sh.als_map[ib]);
                            s_gps[igt * WGS + tid] = (float)lk;
                            if (best_lk < lk) { best_lk = lk; g0 = sh.als_map[ib]; g1 = sh.als_map[ia]; }
                        }
                    }
                } else g1 = BCFGPU_GT_VECTOR_END;
                #pragma unroll
                for (int k = 0; k < 5; ++k) { if (g0 == k) ac_loc[k]++; if (g1 == k) ac_loc[k]++; }
            }
            if (want_gqgp) {
                // mcall.c:842-885
                int nmax;
                if (P.ploidy) nmax = ploidy == 2 ? ngts_new : (ploidy == 1 ? gnals : 0);
                else nmax = ngts_new;
                double mx = s_gps[tid];
                if (mx < 0 || nmax == 0) {
                    if (P.output_tags & BCFGPU_CALL_FMT_GP) {
                        for (int k = 0; k < nmax; ++k) s_gps[k * WGS + tid] = 0;
                        if (nmax == 0) { s_gps[tid] = __uint_as_float(0x7F800001u); nmax = 1; }
                        if (nmax < ngts_new) s_gps[nmax * WGS + tid] = __uint_as_float(0x7F800002u);
                    }
                    gq = 0;
                } else {
                    double sum = mx;
                    for (int k = 1; k < nmax; ++k) { const double v = s_gps[k * WGS + tid]; if (mx < v) mx = v; sum += v; }
                    mx = -4.34294 * log(1 - mx / sum);
                    gq = mx <= 127 ? (int)mx : 127;
                    if (P.output_tags & BCFGPU_CALL_FMT_GP) {
                        for (int k = 0; k < nmax; ++k) s_gps[k * WGS + tid] = (float)(s_gps[k * WGS + tid] / sum);
                        for (int k = nmax; k < ngts_new; ++k) s_gps[k * WGS + tid] = __uint_as_float(0x7F800002u);
                    }
                }
            }
        }
        P.out.gt[((size_t)is * 2 + 0) * Ss + s] = (int8_t)g0;
        P.out.gt[((size_t)is * 2 + 1) * Ss + s] = (int8_t)g1;
        if (is_variant && want_gqgp) {
            if ((P.output_tags & BCFGPU_CALL_FMT_GQ) && P.out.gq) P.out.gq[(size_t)is * Ss + s] = gq;
            if ((P.output_tags & BCFGPU_CALL_FMT_GP) && P.out.gp)
                for (int k = 0; k < ngts_new; ++k) P.out.gp[((size_t)is * ogt + k) * Ss + s] = s_gps[k * WGS + tid];
        }
        // trimmed PLs (mcall.c:1158-1194)
        if (!ref_only && P.out.pl) {
            int32_t *dst = P.out.pl + (size_t)is * ogt * Ss + s;
            if (ploidy == 2) {
                for (int k = 0; k < ngts_new; ++k) dst[(size_t)k * Ss] = plc_at(sh.pl_map[k]);
            } else if (ploidy == 1) {
                int k;
                for (k = 0; k < nals_new; ++k) dst[(size_t)k * Ss] = plc_at(sh.pl_map[(k + 1) * (k + 2) / 2 - 1]);
                if (k < ngts_new) dst[(size_t)k * Ss] = VEND;
            } else {
                dst[0] = MISSING;
                dst[Ss] = VEND;
            }
        }
    }
    }
    // AC totals
    #pragma unroll
    for (int k = 0; k < 5; ++k) { const int v = wsumi(ac_loc[k]); if (tid == 0) sh.ac[k] = v; }
    __syncthreads();

    if (tid == 0) {
        int nAC = 0;
        if (is_variant) for (int i = 1; i < nals_new; i++) nAC += sh.ac[i];
        if (is_variant && !nAC && (P.call_flag & BCFGPU_CALL_VARONLY)) { write_skipped(cs, 0); return; }
        float qual = 0; int qmiss = 0;
        if (nAC) qual = (float)sh.max_qual;
        else {
            if (sh.lk_sum != -HUGE_VAL) qual = (float)(-4.343 * (sh.lk_sum - lse2(sh.lk_sum, sh.ref_lk)));
            else if (sh.ac[0]) qual = P.theta != 0 ? (float)(-4.343 * P.theta) : 0.f;
            else qmiss = 1;
        }
        cs->qual = qual; cs->qual_missing = qmiss;
        for (int i = 0; i < 5; ++i) cs->ac[i] = i < nals_new ? sh.ac[i] : 0;
        cs->an = nAC + sh.ac[0];
        cs->nals_new = nals_new; cs->als_new = als_new;
        for (int i = 0; i < 5; ++i) cs->als_map[i] = i < nals ? sh.als_map[i] : -1;
        cs->ret = nals_new;
        cs->pl_dropped = ref_only ? 1 : 0;
        cs->has_i16 = 0; cs->mq = 0; cs->pv4_tested = 0;
        for (int i = 0; i < 4; ++i) { cs->dp4[i] = 0; cs->pv4[i] = 0.f; }
    }
}

// (A workgroup of four wavefronts per site -- all 256 lanes normalising a round of 1024 samples, the groups dealt to the
// wavefronts -- was measured at 0.37 ms per 32 768-site tile against 0.32 ms for this one-wavefront form: the kernel is bound
// by its trips to memory and the start of its workgroups, not by the additions.)
// The groups' allele-frequency sums of `call -G` (mcall.c:1474-1503): qsum[grp][j] += AD[j] / sum(AD) over the group's samples in
// sample order -- sequential float32 sums, the reference's rounding -- from FORMAT/AD (or QS), for every site that mcall_kernel
// will work on.  One wavefront per site; out[site][grp][5].  Its registers are few (the calling kernel's matrix-core scan is
// what needs many), so that eight wavefronts share a SIMD and cover one another's round trips and chains.
__global__ __launch_bounds__(WGS) __attribute__((amdgpu_waves_per_eu(4, 8))) void grp_qsum_kernel(const McallParams P)
{
    extern __shared__ __align__(16) unsigned char dsm[];
    float *s_gq = reinterpret_cast<float*>(dsm);              // [n_grp][5]
    constexpr int SB = 4 * WGS;                                 // samples per staging round: four consecutive ones per lane
    constexpr int UB = (5 * SB) * (int)sizeof(float) + SB * (int)sizeof(int);
    __shared__ __align__(16) unsigned char s_union[UB];
    const int tid = threadIdx.x;
    const int is = blockIdx.x;
    const int S = P.n_smpl;
    const size_t Ss = (size_t)S;
    const int ngrp = P.n_grp;
    int nals, unseen;
    if (P.msite) { nals = P.msite[is].ret < 0 ? 0 : P.msite[is].n_alleles; unseen = P.msite[is].unseen > 0 ? P.msite[is].unseen : 0; }
    else { nals = P.nals[is]; unseen = P.unseen[is]; }
    const int ngts = nals * (nals + 1) / 2;
    // the records mcall_kernel refuses or never reaches (vcfcall.c:1112-1115) need no sums
    if (nals < 1 || nals > BCFGPU_MAX_ALLELES || ngts > P.n_gt_max || (P.ad && nals > P.n_al_max) || unseen < 0 || unseen >= nals) return;
    if ((P.call_flag & BCFGPU_CALL_VARONLY) && (nals == 1 || (nals == 2 && unseen > 0))) return;
    {
        // group qsum from FORMAT/AD (or QS): qsum[grp][j] += AD[j]/sum in sample order (mcall.c:1478-1503).
        // 64 samples at a time: every lane normalises one sample (coalesced plane reads), then lane j < 5 adds allele
        // j's fractions in sample order -- the sequential float32 sum of the reference -- keeping the running sum of
        // the current group in a register (samples of a group are usually consecutive).
        for (int i = tid; i < ngrp * 5; i += WGS) s_gq[i] = 0;
        constexpr int SB = 4 * WGS;                                 // samples per staging round: four consecutive ones per lane
        float *s_fr = reinterpret_cast<float*>(s_union);            // [5][SB] fractions of the staged samples
        int *s_gg = reinterpret_cast<int*>(s_fr + 5 * SB);          // [SB] their groups
        static_assert(UB >= (int)((5 * SB) * sizeof(float) + SB * sizeof(int)), "staging fits in s_union");
        const int nad = P.ad ? P.n_al_max : nals;
        int cur = -1;
        float acc = 0.f;
        // the next round's counts and groups are requested before the current ones are normalised and added (the loop is a
        // chain of memory round trips otherwise); a lane's four samples of a byte plane are one 4-byte load
        int xn[5][4], gnx[4];
        auto fetch_at = [&](const int s) __attribute__((always_inline)) {   // the lane's four samples s .. s+3
            const int rem = S - s;
            #pragma unroll
            for (int k = 0; k < 5; ++k) {
                #pragma unroll
                for (int j = 0; j < 4; ++j) xn[k][j] = VEND;
                if (rem <= 0 || k >= nad) continue;
                if (P.ad) {
                    #pragma unroll
                    for (int j = 0; j < 4; ++j) if (j < rem) xn[k][j] = P.ad[((size_t)is * P.n_al_max + k) * Ss + s + j];
                } else if (k < nals) {
                    if (P.qs_i32) {
                        #pragma unroll
                        for (int j = 0; j < 4; ++j) if (j < rem) xn[k][j] = (int)P.qs_i32[((size_t)is * 5 + k) * Ss + s + j];
                    } else {
                        // FORMAT/AD = ADF + ADR (bam2bcf.c:892-896): a lane's four samples of a 16-bit plane are one 8-byte load
                        const uint16_t *pa = P.ad_u16 + ((size_t)is * 5 + k) * Ss + s, *pb = P.ad_u16b + ((size_t)is * 5 + k) * Ss + s;
                        uint16_t wa[4] = {0, 0, 0, 0}, wb[4] = {0, 0, 0, 0};
                        if (rem >= 4 && ((reinterpret_cast<uintptr_t>(pa) | reinterpret_cast<uintptr_t>(pb)) & 7) == 0) { __builtin_memcpy(wa, pa, 8); __builtin_memcpy(wb, pb, 8); }
                        else for (int j = 0; j < 4 && j < rem; ++j) { wa[j] = pa[j]; wb[j] = pb[j]; }
                        #pragma unroll
                        for (int j = 0; j < 4; ++j) if (j < rem) xn[k][j] = (int)wa[j] + (int)wb[j];
                    }
                }
            }
            #pragma unroll
            for (int j = 0; j < 4; ++j) gnx[j] = j < rem ? GRP_OF(s + j) : 0;
        };
        auto fetch_ad = [&](int base) __attribute__((always_inline)) { fetch_at(base + 4 * tid); };
        // Groups that are runs of consecutive samples (the usual -G file): lane (group, allele) keeps its group's running sum and
        // adds, round by round, the fractions of its group's samples of that round from LDS -- the reference's order for every
        // group (mcall.c:1485-1500), without a group test per sample and without the five chains over all samples of the general
        // path below.  (Round 3 left the fractions in a global scratch row per allele and let every chain read its stretch
        // back: sixteen dependent trips to L2 per chain, a quarter of the kernel at four groups: profiles/r4_mcall_ablations.txt.)
        bool side_by_side = P.grp_rng != nullptr && ngrp * 5 <= WGS;
        if (side_by_side) {
            bool ok = true;
            if (tid < ngrp) { const int f = P.grp_rng[3 * tid], l = P.grp_rng[3 * tid + 1], n = P.grp_rng[3 * tid + 2]; ok = n == 0 || l - f == n; }
            side_by_side = __all(ok);
        }
        if (side_by_side) {
            // A round stages, for EVERY group, the next CH of its samples (LPG = 64 / n_grp lanes a group, four consecutive samples a
            // lane, from the group's first sample rounded down to a multiple of four: aligned loads; samples outside the group's
            // range count as +0), so all the chains work in every round: n / CH rounds of CH additions per chain instead of S / 256
            // rounds in which one group's chains add 256 values while the others wait.
            const int cg = tid / 5, ca = tid % 5;
            const bool chain = tid < ngrp * 5 && ca < nals;
            const int LPG = WGS / ngrp, CH = 4 * LPG;
            const int lg = tid / LPG, li = tid % LPG;
            int lf = 0, ll = 0;                                           // the range of the lane's group
            if (lg < ngrp && P.grp_rng[3 * lg + 2]) { lf = P.grp_rng[3 * lg]; ll = P.grp_rng[3 * lg + 1]; }
            const int la0 = lf & ~3;
            int rounds = (ll - la0 + CH - 1) / CH;
            #pragma unroll
            for (int o = 32; o > 0; o >>= 1) rounds = max(rounds, __shfl_xor(rounds, o));
            auto lane_s = [&](int r) { const int s = la0 + r * CH + 4 * li; return (lg < ngrp && s < ll) ? s : S; };
            int cspan = 0;                                                // the chain's group: samples from its aligned start
            if (chain && P.grp_rng[3 * cg + 2]) cspan = P.grp_rng[3 * cg + 1] - (P.grp_rng[3 * cg] & ~3);
            float gacc = 0.f;
            fetch_at(lane_s(0));
            for (int r = 0; r < (BCFGPU_ABL(P, 128) ? 0 : rounds); ++r) {
                __syncthreads();                                         // (the chains of the round before are through with s_fr)
                int xc[5][4];
                #pragma unroll
                for (int k = 0; k < 5; ++k)
                    #pragma unroll
                    for (int j = 0; j < 4; ++j) xc[k][j] = xn[k][j];
                const int s0 = lane_s(r);
                fetch_at(lane_s(r + 1));
                if (lg < ngrp) {
                    float fr[5][4];
                    #pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int v[5];
                        float sum = 0;
                        int nvalid = 0;                                   // values before the first vector_end
                        #pragma unroll
                        for (int k = 0; k < 5; ++k) {
                            v[k] = VEND;
                            if (k < nad && nvalid == k) {
                                const int x = xc[k][j];
                                if (x != VEND) { v[k] = x; nvalid = k + 1; if (x != MISSING) sum += (float)x; }
                            }
                        }
                        const bool mine = s0 + j >= lf && s0 + j < ll;   // (s0 = S past the group's end: nothing was loaded)
                        #pragma unroll
                        for (int k = 0; k < 5; ++k)                       // +0 where the reference adds nothing (and outside the group)
                            fr[k][j] = (mine && sum != 0.f && k < nvalid && v[k] != MISSING) ? (float)v[k] / sum : 0.f;
                    }
                    #pragma unroll
                    for (int k = 0; k < 5; ++k)
                        if (k < nals) *reinterpret_cast<float4*>(s_fr + k * SB + lg * CH + 4 * li) = make_float4(fr[k][0], fr[k][1], fr[k][2], fr[k][3]);
                }
                __syncthreads();
                if (chain && r * CH < cspan) {
                    const float4 *q4 = reinterpret_cast<const float4*>(s_fr + ca * SB + cg * CH);
                    const int nb = LPG;                                    // float4s of the round; four of them in flight
                    float4 c0 = make_float4(0, 0, 0, 0), c1 = c0, c2 = c0, c3 = c0;
                    if (nb > 0) c0 = q4[0];
                    if (nb > 1) c1 = q4[1];
                    if (nb > 2) c2 = q4[2];
                    if (nb > 3) c3 = q4[3];
                    for (int b4 = 0; b4 < nb; b4 += 4) {
                        const float4 a0 = c0, a1 = c1, a2 = c2, a3 = c3;
                        if (b4 + 4 < nb) c0 = q4[b4 + 4];
                        if (b4 + 5 < nb) c1 = q4[b4 + 5];
                        if (b4 + 6 < nb) c2 = q4[b4 + 6];
                        if (b4 + 7 < nb) c3 = q4[b4 + 7];
                        gacc += a0.x; gacc += a0.y; gacc += a0.z; gacc += a0.w;
                        if (b4 + 1 < nb) { gacc += a1.x; gacc += a1.y; gacc += a1.z; gacc += a1.w; }
                        if (b4 + 2 < nb) { gacc += a2.x; gacc += a2.y; gacc += a2.z; gacc += a2.w; }
                        if (b4 + 3 < nb) { gacc += a3.x; gacc += a3.y; gacc += a3.z; gacc += a3.w; }
                    }
                }
            }
            __syncthreads();
            if (chain) s_gq[cg * 5 + ca] = gacc;
            cur = -1;                                                     // (nothing left in the running-group register)
        } else {
        fetch_ad(0);
        for (int base = 0; base < (BCFGPU_ABL(P, 128) ? 0 : S); base += SB) {
            const int cn = min(SB, S - base);
            __syncthreads();
            int xc[5][4], gcur[4];
            #pragma unroll
            for (int k = 0; k < 5; ++k)
                #pragma unroll
                for (int j = 0; j < 4; ++j) xc[k][j] = xn[k][j];
            #pragma unroll
            for (int j = 0; j < 4; ++j) gcur[j] = gnx[j];
            fetch_ad(base + SB);
            #pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ls = 4 * tid + j;                        // the sample's place in the round
                if (ls < cn) {
                    int v[5];
                    float sum = 0;
                    int nvalid = 0;                                   // values before the first vector_end
                    #pragma unroll
                    for (int k = 0; k < 5; ++k) {
                        v[k] = VEND;
                        if (k < nad && nvalid == k) {
                            const int x = xc[k][j];
                            if (x != VEND) { v[k] = x; nvalid = k + 1; if (x != MISSING) sum += (float)x; }
                        }
                    }
                    #pragma unroll
                    for (int k = 0; k < 5; ++k)                       // +0 where the reference adds nothing
                        s_fr[k * SB + ls] = (sum != 0.f && k < nvalid && v[k] != MISSING) ? (float)v[k] / sum : 0.f;
                    s_gg[ls] = gcur[j];
                }
            }
            __syncthreads();
            if (tid < 5 && tid < nals) {
                // sixteen samples per trip: their four LDS read pairs are in flight together (one read pair per trip leaves the
                // lane waiting out an LDS round trip for every four additions)
                const float4 *fr4 = reinterpret_cast<const float4*>(s_fr + tid * SB);
                const int4 *gg4 = reinterpret_cast<const int4*>(s_gg);
                const int nb = (cn + 3) >> 2;                      // the tail of the last block holds +0 / stale groups: see below
                for (int b0 = 0; b0 < nb; b0 += 4) {
                    float4 fq[4]; int4 gq4[4];
                    #pragma unroll
                    for (int u = 0; u < 4; ++u) if (b0 + u < nb) { fq[u] = fr4[b0 + u]; gq4[u] = gg4[b0 + u]; }
                    #pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int b4 = b0 + u;
                        if (b4 >= nb) break;
                        const float4 f = fq[u]; const int4 gv = gq4[u];
                        const float fv[4] = {f.x, f.y, f.z, f.w};
                        const int gs4[4] = {gv.x, gv.y, gv.z, gv.w};
                        // four samples of the running group (groups are usually runs of consecutive samples): just the four adds
                        if (4 * b4 + 3 < cn && ((gv.x ^ cur) | (gv.y ^ cur) | (gv.z ^ cur) | (gv.w ^ cur)) == 0) {
                            acc += f.x; acc += f.y; acc += f.z; acc += f.w;
                            continue;
                        }
                        #pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (4 * b4 + k < cn) {
                                const int g = gs4[k];
                                if (g != cur) {
                                    if (cur >= 0) s_gq[cur * 5 + tid] = acc;
                                    cur = g; acc = s_gq[g * 5 + tid];
                                }
                                acc += fv[k];
                            }
                        }
                    }
                }
            }
        }
        }
        if (tid < 5 && tid < nals && cur >= 0) s_gq[cur * 5 + tid] = acc;
    }
    __syncthreads();
    for (int i = tid; i < ngrp * 5; i += WGS) P.grp_q[(size_t)is * ngrp * 5 + i] = s_gq[i];
}

// DP4, MQ and PV4 from I16 (mcall.c:1659-1679; test16 of ccall.c:103-138): per-site scalar work with loops of its own
// (Fisher's exact test, the continued fraction of kf_betai), kept out of mcall_kernel so that its registers stay what the
// calling needs.  Four lanes per site, one per PV4 test; runs after the calling kernels of the same launch sequence.
__global__ __launch_bounds__(64) void i16_kernel(const McallParams P)
{
    const int is = blockIdx.x * 16 + (threadIdx.x >> 2), k = threadIdx.x & 3;
    if (is >= P.n_sites) return;
    bcfgpu_call_site *cs = &P.out.site[is];
    if (cs->ret <= 0 && cs->nals_new == 0) return;               // a skipped record (write_skipped) carries nothing
    // the reference reads I16 back from the record as floats
    float a[16];
    #pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = P.msite ? (float)P.msite[is].anno[i] : P.i16[(size_t)is * 16 + i];
    const float depth = a[0] + a[1] + a[2] + a[3];
    if (k == 0) {
        cs->has_i16 = 1;
        for (int i = 0; i < 4; ++i) cs->dp4[i] = (int32_t)a[i];
        // float division truncated to int32; at depth 0 the reference converts a NaN (x86: INT32_MIN = missing)
        cs->mq = depth != 0.f ? (int32_t)((a[8] + a[10]) / depth) : BCFGPU_INT32_MISSING;
    }
    if (P.output_tags & BCFGPU_CALL_FMT_PV4) {
        const bool tested = depth != 0.f && a[0] + a[1] > 0 && a[2] + a[3] > 0;
        if (tested) cs->pv4[k] = (float)dev_test16_one(a, k);
        if (k == 0) cs->pv4_tested = tested ? 1 : 0;
    }
}

__global__ void grp_range_init_kernel(int32_t *rng, int n_grp)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g < n_grp) { rng[3 * g] = 0x7fffffff; rng[3 * g + 1] = 0; rng[3 * g + 2] = 0; }
}
// group ids are checked (BCFGPU_E_RANGE) and every group's sample range [first, last + 1) is taken
__global__ void grp_check_kernel(const int32_t *grp, int n_smpl, int n_grp, int *err, int32_t *rng)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_smpl) return;
    const int g = grp[s];
    if (g < 0 || g >= n_grp) { atomicExch(err, BCFGPU_E_RANGE); return; }
    if (!rng) return;
    // the samples of a group tend to follow one another: a wavefront whose lanes all have one group makes three atomics, not 192
    const int g0 = __builtin_amdgcn_readfirstlane(g);
    if (__all(g == g0)) {
        const unsigned long long m = __ballot(1);
        if ((int)(threadIdx.x & 63) == __builtin_ctzll(m)) { atomicMin(&rng[3 * g], s); atomicMax(&rng[3 * g + 1], s + (63 - __builtin_clzll(m) - __builtin_ctzll(m)) + 1); atomicAdd(&rng[3 * g + 2], (int)__popcll(m)); }
    } else { atomicMin(&rng[3 * g], s); atomicMax(&rng[3 * g + 1], s + 1); atomicAdd(&rng[3 * g + 2], 1); }
}

void launch_mcall(const McallParams &p_in, hipStream_t s)
{
    if (p_in.n_sites == 0) return;
    #ifndef MCALL_SMALL_TOO_SAMPLES
    #define MCALL_SMALL_TOO_SAMPLES 256
    #endif
    McallParams p = p_in;
    p.small_too = (p.pl_is_u8 && p.n_smpl >= MCALL_SMALL_TOO_SAMPLES) ? 1 : 0;        // (below; the fused pipeline's launches only)
    const int ngrp = p.n_grp > 1 ? p.n_grp : 1;
    const size_t lds = (size_t)ngrp * 5 * sizeof(float) + (size_t)ngrp * 2 * sizeof(int);
    if (p.grp && ngrp > 1) {
        if (p.grp_rng) hipLaunchKernelGGL(grp_range_init_kernel, dim3((ngrp + 255) / 256), dim3(256), 0, s, p.grp_rng, ngrp);
        hipLaunchKernelGGL(grp_check_kernel, dim3((p.n_smpl + 255) / 256), dim3(256), 0, s, p.grp, p.n_smpl, ngrp, p.err, p.grp_rng);
    }
    if (p.grp && ngrp > 1) hipLaunchKernelGGL(grp_qsum_kernel, dim3(p.n_sites), dim3(WGS), (size_t)ngrp * 5 * sizeof(float), s, p);
    // Each instantiation is launched over all sites and a workgroup leaves at once when the site is another's: 25-45 us a launch of
    // 32 768 workgroups.  With many samples nearly every site shows four or five alleles (sequencing errors alone), so the small
    // instantiation is not launched then and the few sites it would take run in the 15-subset one (same arithmetic, longer loops).
    #define MCALL_LAUNCH3(FAST_, HAP_, GRP_) do { \
        if (!p.small_too) hipLaunchKernelGGL((mcall_kernel<3, 7, FAST_, HAP_, GRP_>), dim3(p.n_sites), dim3(WGS), lds, s, p); \
        hipLaunchKernelGGL((mcall_kernel<5, 15, FAST_, HAP_, GRP_>), dim3(p.n_sites), dim3(WGS), lds, s, p); \
        hipLaunchKernelGGL((mcall_kernel<5, 25, FAST_, HAP_, GRP_>), dim3(p.n_sites), dim3(WGS), lds, s, p); } while (0)
    if (p.pl_is_u8 && !BCFGPU_ABL(p, 64)) {
        // u8 PLs (the fused pipeline): the lane-per-sample subset scan (matrix cores for 25 subsets); the haploid forms only with a ploidy
        // array, the group handling only with more than one group
        if (p.ploidy) MCALL_LAUNCH3(true, true, true);
        else if (ngrp > 1) MCALL_LAUNCH3(true, false, true);
        else MCALL_LAUNCH3(true, false, false);
    } else {
        MCALL_LAUNCH3(false, false, true);
    }
    #undef MCALL_LAUNCH3
    if (p.msite || p.i16) hipLaunchKernelGGL(i16_kernel, dim3((p.n_sites + 15) / 16), dim3(64), 0, s, p);
}

}  // namespace bcfgpu

// mcall.hip -- the multiallelic caller `bcftools call -m`, one 256-thread workgroup per site.
//
// Replaces mcall() (mcall.c:1430-1684) and what it calls: set_pdg (:451-544),
// mcall_find_best_alleles (:591-710), mcall_set_ref_genotypes (:713-743),
// mcall_call_genotypes (:745-886), init_allele_trimming_maps (:547-570),
// mcall_trim_and_update_PLs (:1158-1194), plus the QS/-G/-F frequency set-up (:1453-1535).
//
// Lanes run over samples.  Per-sample P(D|G) lives in registers (static indices only).  The
// subset scan of find_best_alleles keeps one register accumulator per 1-, 2- and 3-allele
// subset (25 for five alleles), reduced across the workgroup in fp64.  Integer results (GT,
// AC/AN, trimmed PL) are exact; log-likelihood sums are tree-reduced, so QUAL agrees with
// the sequential CPU sum to ~1e-13 relative, far inside the 1e-4 contract.
// Order-sensitive float32 pieces (group qsum from AD, -F prior, normalisation) are replayed
// sequentially by single lanes exactly as the reference does.
#include <hip/hip_runtime.h>
#include <math.h>
#include "kernels.h"

namespace bcfgpu {

#define WG 256
#define MISSING BCFGPU_INT32_MISSING
#define VEND    BCFGPU_INT32_VECTOR_END

__device__ __forceinline__ int a2gt(int a, int b) { return a > b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a; }
__device__ __forceinline__ double lse2(double a, double b)
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

struct CallShared {
    int nals, unseen, ngts;
    float qsum[5];              // current group's allele frequencies
    double red[25][4];          // per-wave partial sums of the subset log-likelihoods
    int    redset[4];           // per-wave lk_tot_set bits
    int als_new, nals_new, is_variant, early;
    int grp_als, grp_nals;      // current group's best allele set
    int als_map[5]; int pl_map[15];
    int ac[5];
    double max_qual, ref_lk, lk_sum;
    float prior_fail;
};

// subset numbering: singles 0..4, pairs 5..14 (ia>ib), triples 15..24 (ia>ib>ic), in the
// reference's enumeration order
__device__ __forceinline__ int pair_id(int ia, int ib) { return 5 + ia * (ia - 1) / 2 + ib; }
__device__ __forceinline__ int trip_id(int ia, int ib, int ic)
{   // order: (2,1,0),(3,1,0),(3,2,0),(3,2,1),(4,1,0),(4,2,0),(4,2,1),(4,3,0),(4,3,1),(4,3,2)
    const int base[5] = {0, 0, 0, 1, 4};
    return 15 + base[ia] + ib * (ib - 1) / 2 + ic;
}

// load the PLs of one sample into registers; returns them as int (sentinels preserved)
__device__ __forceinline__ void load_pl(const McallParams &P, int is, int s, int ngts, int (&pl)[15])
{
    const size_t S = P.n_smpl;
    if (P.pl_is_u8) {
        const uint8_t *src = reinterpret_cast<const uint8_t*>(P.pl) + (size_t)is * BCFGPU_MAX_PL * S + s;
        #pragma unroll
        for (int j = 0; j < 15; ++j) pl[j] = j < ngts ? (int)src[(size_t)j * S] : VEND;
    } else {
        const int32_t *src = reinterpret_cast<const int32_t*>(P.pl) + (size_t)is * P.n_gt_max * S + s;
        #pragma unroll
        for (int j = 0; j < 15; ++j) pl[j] = j < ngts ? src[(size_t)j * S] : VEND;
    }
}

__device__ __forceinline__ double pl2prob(const double *pl2p, int v) { return v < 256 ? pl2p[v] : pow(10., -v / 10.); }

// set_pdg for one sample (mcall.c:460-543).  `scr` is this lane's LDS column (stride WG) used
// only by the rare partially-missing path, which needs run-time indexing.
__device__ void set_pdg_one(const double *pl2p, int (&pl)[15], double (&pdg)[15], int n_gt, int nals, int unseen, int *scr)
{
    double sum = 0;
    int j = n_gt;               // index of the first missing value, or n_gt
    bool vend0 = false;
    #pragma unroll
    for (int k = 0; k < 15; ++k) {
        if (k < n_gt && j == n_gt) {
            if (pl[k] == VEND) { j = 0; vend0 = true; }
            else if (pl[k] == MISSING) j = k;
            else { pdg[k] = pl2prob(pl2p, pl[k]); sum += pdg[k]; }
        }
    }
    (void)vend0;
    if (j == 0) {
        j = n_gt; sum = n_gt;
    } else if (j < n_gt && unseen < 0) {
        sum = 0;
        #pragma unroll
        for (int k = 0; k < 15; ++k)
            if (k < n_gt) {
                if (pl[k] == MISSING) pl[k] = 255;
                pdg[k] = pl2prob(pl2p, pl[k]); sum += pdg[k];
            }
        j = n_gt;
    }
    if (j < n_gt) {
        // fill missing values from the unseen-allele likelihoods (mcall.c:495-527)
        #pragma unroll
        for (int k = 0; k < 15; ++k) scr[k * WG] = pl[k];
        int jj = 0;
        sum = 0;
        for (int ia = 0; ia < nals; ia++)
            for (int ib = 0; ib <= ia; ib++) {
                if (scr[jj * WG] == MISSING) {
                    int k = a2gt(ia, unseen);
                    if (scr[k * WG] == MISSING) k = a2gt(ib, unseen);
                    if (scr[k * WG] == MISSING) k = a2gt(unseen, unseen);
                    if (scr[k * WG] == MISSING) scr[jj * WG] = 255;
                    else scr[jj * WG] = scr[k * WG];
                }
                jj++;
            }
        #pragma unroll
        for (int k = 0; k < 15; ++k)
            if (k < n_gt) { pl[k] = scr[k * WG]; pdg[k] = pl2p[pl[k] & 255]; sum += pdg[k]; }
    }
    if (sum == (double)n_gt) {
        #pragma unroll
        for (int k = 0; k < 15; ++k) pdg[k] = 0;
    } else {
        #pragma unroll
        for (int k = 0; k < 15; ++k) if (k < n_gt) pdg[k] /= sum; else pdg[k] = 0;
    }
}

__global__ __launch_bounds__(WG) void mcall_kernel(const McallParams P)
{
    extern __shared__ __align__(16) unsigned char dsm[];
    float *s_gq = reinterpret_cast<float*>(dsm);              // [n_grp][5] group qsum (when n_grp>1)
    __shared__ CallShared sh;
    __shared__ int s_scr[15 * WG];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int is = blockIdx.x;
    const int S = P.n_smpl;
    const size_t Ss = (size_t)S;
    const int ngrp = P.n_grp > 1 ? P.n_grp : 1;
    bcfgpu_call_site *cs = &P.out.site[is];

    int nals, unseen;
    if (P.msite) {
        nals = P.msite[is].n_alleles;
        unseen = P.msite[is].unseen > 0 ? P.msite[is].unseen : 0;     // vcfcall.c:1102-1111
    } else { nals = P.nals[is]; unseen = P.unseen[is]; }
    const int ngts = nals * (nals + 1) / 2;

    // record-loop prologue of vcfcall.c:1112-1115: with -v a REF-only record never reaches mcall()
    if ((P.call_flag & BCFGPU_CALL_VARONLY) && (nals == 1 || (nals == 2 && unseen > 0))) {
        if (tid == 0) { cs->ret = 0; cs->nals_new = 0; cs->als_new = 0; cs->an = 0; cs->qual = 0; cs->qual_missing = 0; cs->pl_dropped = 0;
                        for (int i = 0; i < 5; ++i) { cs->als_map[i] = -1; cs->ac[i] = 0; } }
        return;
    }

    // ---- allele-frequency set-up (mcall.c:1453-1535), sequential float32 ----
    if (tid == 0) {
        sh.nals = nals; sh.unseen = unseen; sh.ngts = ngts;
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
    } else {
        // group qsum from FORMAT/AD (or QS): qsum[j] += AD[j]/sum in sample order (mcall.c:1478-1503)
        for (int i = tid; i < ngrp * 5; i += WG) s_gq[i] = 0;
        __syncthreads();
        if (tid < 5 && tid < nals) {
            const int j = tid;
            const int nad = P.ad ? P.n_al_max : nals;
            for (int s = 0; s < S; ++s) {
                float sum = 0; int myv = MISSING; bool mine_valid = false;
                for (int k = 0; k < nad; ++k) {
                    int v;
                    if (P.ad) v = P.ad[((size_t)is * P.n_al_max + k) * Ss + s];
                    else if (P.qs_u16) v = k < nals ? (int)P.qs_u16[((size_t)is * 5 + k) * Ss + s] : VEND;
                    else v = k < nals ? (int)P.ad_u8[((size_t)is * 5 + k) * Ss + s] + (int)P.ad_u8b[((size_t)is * 5 + k) * Ss + s] : VEND;
                    if (v == VEND) break;
                    if (v != MISSING) sum += (float)v;
                    if (k == j) { myv = v; mine_valid = true; }
                }
                if (sum != 0.f && mine_valid && myv != MISSING) s_gq[P.grp[s] * 5 + j] += (float)myv / sum;
            }
        }
    }
    __syncthreads();
    // -F AN,AC prior and normalisation: lane g handles group g (mcall.c:1506-1535)
    for (int g = tid; g < ngrp; g += WG) {
        float *q = s_gq + g * 5;
        if (P.prior_an && P.prior_ac && P.prior_an[is] != MISSING) {
            const int an = P.prior_an[is];
            const int32_t *acv = P.prior_ac + (size_t)is * 4;
            int nac = 0;
            while (nac < 4 && acv[nac] != VEND) nac++;
            if (an > 0 && nac == nals - 1) {
                int gn = 0;
                if (ngrp == 1) gn = S; else for (int s = 0; s < S; ++s) gn += (P.grp[s] == g);
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
    if (sh.prior_fail != 0.f) {           // error("Incorrect AN,AC values") in the reference
        if (tid == 0) { cs->ret = -1; cs->nals_new = 0; cs->als_new = 0; cs->an = 0; cs->qual = 0; cs->qual_missing = 0; cs->pl_dropped = 0;
                        for (int i = 0; i < 5; ++i) { cs->als_map[i] = -1; cs->ac[i] = 0; } }
        return;
    }

    // ---- per group: mcall_find_best_alleles ----
    for (int g = 0; g < ngrp; ++g) {
        __syncthreads();
        if (tid < 5) sh.qsum[tid] = s_gq[g * 5 + tid];
        __syncthreads();
        float qf[5];
        #pragma unroll
        for (int i = 0; i < 5; ++i) qf[i] = sh.qsum[i];

        double acc[25];
        #pragma unroll
        for (int i = 0; i < 25; ++i) acc[i] = 0;
        int setbits = 0;
        for (int s = tid; s < S; s += WG) {
            if (ngrp > 1 && P.grp[s] != g) continue;
            int pl[15]; double pdg[15];
            load_pl(P, is, s, ngts, pl);
            set_pdg_one(P.pl2p, pl, pdg, ngts, nals, unseen, s_scr + tid);
            const int ploidy = P.ploidy ? P.ploidy[s] : 2;
            #pragma unroll
            for (int ia = 0; ia < 5; ++ia) {
                if (ia >= nals) break;
                const int iaa = (ia + 1) * (ia + 2) / 2 - 1;
                if (pdg[iaa] != 0.0) { acc[ia] += log(pdg[iaa]); setbits |= 1 << ia; }
            }
            #pragma unroll
            for (int ia = 1; ia < 5; ++ia) {
                if (ia >= nals) break;
                if (qf[ia] == 0.f) continue;
                const int iaa = (ia + 1) * (ia + 2) / 2 - 1;
                #pragma unroll
                for (int ib = 0; ib < ia; ++ib) {
                    if (qf[ib] == 0.f) continue;
                    const double fa = (double)(qf[ia] / (qf[ia] + qf[ib]));
                    const double fb = (double)(qf[ib] / (qf[ia] + qf[ib]));
                    const double fa2 = fa * fa, fb2 = fb * fb, fab = 2 * fa * fb;
                    const int ibb = (ib + 1) * (ib + 2) / 2 - 1, iab = iaa - ia + ib;
                    double val = 0;
                    if (ploidy == 2) val = fa2 * pdg[iaa] + fb2 * pdg[ibb] + fab * pdg[iab];
                    else if (ploidy == 1) val = fa * pdg[iaa] + fb * pdg[ibb];
                    const int id = pair_id(ia, ib);
                    if (val != 0.0) { acc[id] += log(val); setbits |= 1 << id; }
                }
            }
            #pragma unroll
            for (int ia = 2; ia < 5; ++ia) {
                if (ia >= nals) break;
                if (qf[ia] == 0.f) continue;
                const int iaa = (ia + 1) * (ia + 2) / 2 - 1;
                #pragma unroll
                for (int ib = 1; ib < ia; ++ib) {
                    if (qf[ib] == 0.f) continue;
                    const int ibb = (ib + 1) * (ib + 2) / 2 - 1, iab = iaa - ia + ib;
                    #pragma unroll
                    for (int ic = 0; ic < ib; ++ic) {
                        if (qf[ic] == 0.f) continue;
                        const float den = qf[ia] + qf[ib] + qf[ic];
                        const double fa = (double)(qf[ia] / den), fb = (double)(qf[ib] / den), fc = (double)(qf[ic] / den);
                        const double fa2 = fa * fa, fb2 = fb * fb, fc2 = fc * fc;
                        const double fab = 2 * fa * fb, fac = 2 * fa * fc, fbc = 2 * fb * fc;
                        const int icc = (ic + 1) * (ic + 2) / 2 - 1, iac = iaa - ia + ic, ibc = ibb - ib + ic;
                        double val = 0;
                        if (ploidy == 2)
                            val = fa2 * pdg[iaa] + fb2 * pdg[ibb] + fc2 * pdg[icc] + fab * pdg[iab] + fac * pdg[iac] + fbc * pdg[ibc];
                        else if (ploidy == 1)
                            val = fa * pdg[iaa] + fb * pdg[ibb] + fc * pdg[icc];
                        const int id = trip_id(ia, ib, ic);
                        if (val != 0.0) { acc[id] += log(val); setbits |= 1 << id; }
                    }
                }
            }
        }
        // workgroup reduction
        #pragma unroll
        for (int i = 0; i < 25; ++i) { const double v = wsum(acc[i]); if (lane == 0) sh.red[i][wave] = v; }
        setbits = wor(setbits);
        if (lane == 0) sh.redset[wave] = setbits;
        __syncthreads();
        if (tid == 0) {
            // replay of the subset enumeration (mcall.c:600-698) on the reduced sums
            const int set = sh.redset[0] | sh.redset[1] | sh.redset[2] | sh.redset[3];
            int max_als = 0;
            double ref_lk = -HUGE_VAL, max_lk = -HUGE_VAL, lk_sum = -HUGE_VAL;
            const double theta = P.theta;
            for (int ia = 0; ia < nals; ia++) {
                double lk_tot = sh.red[ia][0] + sh.red[ia][1] + sh.red[ia][2] + sh.red[ia][3];
                const int lk_tot_set = (set >> ia) & 1;
                if (ia == 0) ref_lk = lk_tot; else lk_tot += theta;
                if (max_lk < lk_tot && lk_tot_set) { max_lk = lk_tot; max_als = 1 << ia; }
                if (ia > 0 && lk_tot_set) lk_sum = lse2(lk_tot, lk_sum);
            }
            if (nals > 1)
                for (int ia = 0; ia < nals; ia++) {
                    if (sh.qsum[ia] == 0) continue;
                    for (int ib = 0; ib < ia; ib++) {
                        if (sh.qsum[ib] == 0) continue;
                        const int id = pair_id(ia, ib);
                        double lk_tot = sh.red[id][0] + sh.red[id][1] + sh.red[id][2] + sh.red[id][3];
                        const int lk_tot_set = (set >> id) & 1;
                        if (ia != 0) lk_tot += theta;
                        if (ib != 0) lk_tot += theta;
                        if (max_lk < lk_tot && lk_tot_set) { max_lk = lk_tot; max_als = 1 << ia | 1 << ib; }
                        if (lk_tot_set) lk_sum = lse2(lk_tot, lk_sum);
                    }
                }
            if (nals > 2)
                for (int ia = 0; ia < nals; ia++) {
                    if (sh.qsum[ia] == 0) continue;
                    for (int ib = 0; ib < ia; ib++) {
                        if (sh.qsum[ib] == 0) continue;
                        for (int ic = 0; ic < ib; ic++) {
                            if (sh.qsum[ic] == 0) continue;
                            const int id = trip_id(ia, ib, ic);
                            double lk_tot = sh.red[id][0] + sh.red[id][1] + sh.red[id][2] + sh.red[id][3];
                            const int lk_tot_set = (set >> id) & 1;
                            if (ia != 0) lk_tot += theta;
                            if (ib != 0) lk_tot += theta;
                            if (ic != 0) lk_tot += theta;
                            if (max_lk < lk_tot && lk_tot_set) { max_lk = lk_tot; max_als = 1 << ia | 1 << ib | 1 << ic; }
                            if (lk_tot_set) lk_sum = lse2(lk_tot, lk_sum);
                        }
                    }
                }
            // group result is needed again by call_genotypes: keep als per group in s_gq's spare float slot?
            // -> stored in the als array below
            sh.grp_als = max_als;
            int n = 0;
            for (int i = 0; i < nals; i++) if (max_als & 1 << i) n++;
            sh.grp_nals = n;
            sh.als_new |= max_als;
            if (max_lk != -HUGE_VAL) {
                const double qual = -4.343 * (ref_lk - lse2(lk_sum, ref_lk));
                if (sh.max_qual < qual) { sh.max_qual = qual; sh.lk_sum = lk_sum; sh.ref_lk = ref_lk; }
            }
            // remember the group's allele set (float bit pattern of an int, 5th..: use a separate int view)
            reinterpret_cast<int*>(s_gq + ngrp * 5)[g * 2] = max_als;
            reinterpret_cast<int*>(s_gq + ngrp * 5)[g * 2 + 1] = n;
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
        if (tid == 0) { cs->ret = 0; cs->nals_new = 0; cs->als_new = 0; cs->an = 0; cs->qual = 0; cs->qual_missing = 0; cs->pl_dropped = 0;
                        for (int i = 0; i < 5; ++i) { cs->als_map[i] = -1; cs->ac[i] = 0; } }
        return;
    }
    const int als_new = sh.als_new, nals_new = sh.nals_new, is_variant = sh.is_variant;
    const int ngts_new = nals_new * (nals_new + 1) / 2;
    const bool ref_only = (als_new == 1);
    const bool want_gqgp = (P.output_tags & (BCFGPU_CALL_FMT_GQ | BCFGPU_CALL_FMT_GP)) != 0;
    const int *grp_als_tab = reinterpret_cast<int*>(s_gq + ngrp * 5);
    int amap[5];
    #pragma unroll
    for (int i = 0; i < 5; ++i) amap[i] = sh.als_map[i];

    // ---- genotypes (mcall_set_ref_genotypes / mcall_call_genotypes) + PL trimming ----
    int ac_loc[5] = {0, 0, 0, 0, 0};
    const int ogt = P.out_n_gt_max;
    for (int s = tid; s < S; s += WG) {
        int pl[15]; double pdg[15];
        load_pl(P, is, s, ngts, pl);
        set_pdg_one(P.pl2p, pl, pdg, ngts, nals, unseen, s_scr + tid);
        const int ploidy = P.ploidy ? P.ploidy[s] : 2;
        bool allzero = true;
        #pragma unroll
        for (int k = 0; k < 15; ++k) if (k < ngts && pdg[k] != 0.0) allzero = false;
        int g0, g1;
        float gps[15];
        #pragma unroll
        for (int k = 0; k < 15; ++k) gps[k] = 0.f;
        int gq = 0;
        int gnals = 0;
        if (!is_variant) {
            if (allzero || !ploidy) { g0 = BCFGPU_GT_MISSING; g1 = ploidy == 2 ? BCFGPU_GT_MISSING : BCFGPU_GT_VECTOR_END; }
            else { g0 = 0; g1 = ploidy == 2 ? 0 : BCFGPU_GT_VECTOR_END; ac_loc[0] += ploidy; }
        } else {
            const int g = ngrp > 1 ? P.grp[s] : 0;
            const int gals = grp_als_tab[g * 2];
            gnals = grp_als_tab[g * 2 + 1];
            const float *gq5 = s_gq + g * 5;
            if (!ploidy) { g0 = BCFGPU_GT_MISSING; g1 = BCFGPU_GT_VECTOR_END; gps[0] = -1; }
            else if (allzero) { g0 = BCFGPU_GT_MISSING; g1 = ploidy == 2 ? BCFGPU_GT_MISSING : BCFGPU_GT_VECTOR_END; gps[0] = -1; }
            else {
                g0 = 0; g1 = ploidy == 2 ? 0 : BCFGPU_GT_VECTOR_END;
                double best_lk = 0;
                #pragma unroll
                for (int ia = 0; ia < 5; ++ia) {
                    if (ia >= nals) break;
                    if (!(gals & 1 << ia)) continue;
                    const int iaa = (ia + 1) * (ia + 2) / 2 - 1;
                    const double lk = ploidy == 2 ? pdg[iaa] * gq5[ia] * gq5[ia] : pdg[iaa] * gq5[ia];
                    const int igt = ploidy == 2 ? a2gt(amap[ia], amap[ia]) : amap[ia];
                    #pragma unroll
                    for (int k = 0; k < 15; ++k) if (k == igt) gps[k] = (float)lk;
                    if (best_lk < lk) { best_lk = lk; g0 = amap[ia]; }
                }
                if (ploidy == 2) {
                    g1 = g0;
                    #pragma unroll
                    for (int ia = 1; ia < 5; ++ia) {
                        if (ia >= nals) break;
                        if (!(gals & 1 << ia)) continue;
                        const int iaa = (ia + 1) * (ia + 2) / 2 - 1;
                        #pragma unroll
                        for (int ib = 0; ib < ia; ++ib) {
                            if (!(gals & 1 << ib)) continue;
                            const int iab = iaa - ia + ib;
                            const double lk = 2 * pdg[iab] * gq5[ia] * gq5[ib];
                            const int igt = a2gt(amap[ia], amap[ib]);
                            #pragma unroll
                            for (int k = 0; k < 15; ++k) if (k == igt) gps[k] = (float)lk;
                            if (best_lk < lk) { best_lk = lk; g0 = amap[ib]; g1 = amap[ia]; }
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
                double mx = gps[0];
                if (mx < 0 || nmax == 0) {
                    if (P.output_tags & BCFGPU_CALL_FMT_GP) {
                        #pragma unroll
                        for (int k = 0; k < 15; ++k) if (k < nmax) gps[k] = 0;
                        if (nmax == 0) { gps[0] = __uint_as_float(0x7F800001u); nmax = 1; }
                        #pragma unroll
                        for (int k = 0; k < 15; ++k) if (k == nmax && nmax < ngts_new) gps[k] = __uint_as_float(0x7F800002u);
                    }
                    gq = 0;
                } else {
                    double sum = gps[0];
                    #pragma unroll
                    for (int k = 1; k < 15; ++k) if (k < nmax) { if (mx < gps[k]) mx = gps[k]; sum += gps[k]; }
                    mx = -4.34294 * log(1 - mx / sum);
                    gq = mx <= 127 ? (int)mx : 127;
                    if (P.output_tags & BCFGPU_CALL_FMT_GP) {
                        #pragma unroll
                        for (int k = 0; k < 15; ++k) {
                            if (k < nmax) gps[k] = (float)(gps[k] / sum);
                            else if (k < ngts_new) gps[k] = __uint_as_float(0x7F800002u);
                        }
                    }
                }
            }
        }
        P.out.gt[((size_t)is * 2 + 0) * Ss + s] = (int8_t)g0;
        P.out.gt[((size_t)is * 2 + 1) * Ss + s] = (int8_t)g1;
        if (is_variant && want_gqgp) {
            if ((P.output_tags & BCFGPU_CALL_FMT_GQ) && P.out.gq) P.out.gq[(size_t)is * Ss + s] = gq;
            if ((P.output_tags & BCFGPU_CALL_FMT_GP) && P.out.gp) {
                #pragma unroll
                for (int k = 0; k < 15; ++k) if (k < ngts_new) P.out.gp[((size_t)is * ogt + k) * Ss + s] = gps[k];
            }
        }
        // trimmed PLs (mcall.c:1158-1194)
        if (!ref_only && P.out.pl) {
            int32_t *dst = P.out.pl + (size_t)is * ogt * Ss + s;
            #pragma unroll
            for (int k = 0; k < 15; ++k) s_scr[k * WG + tid] = pl[k];
            if (ploidy == 2) {
                for (int k = 0; k < ngts_new; ++k) dst[(size_t)k * Ss] = s_scr[sh.pl_map[k] * WG + tid];
            } else if (ploidy == 1) {
                int k;
                for (k = 0; k < nals_new; ++k) dst[(size_t)k * Ss] = s_scr[sh.pl_map[(k + 1) * (k + 2) / 2 - 1] * WG + tid];
                if (k < ngts_new) dst[(size_t)k * Ss] = VEND;
            } else {
                dst[0] = MISSING;
                dst[Ss] = VEND;
            }
        }
    }
    // AC totals
    #pragma unroll
    for (int k = 0; k < 5; ++k) {
        int v = ac_loc[k];
        #pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0 && v) atomicAdd(&sh.ac[k], v);
    }
    __syncthreads();

    if (tid == 0) {
        int nAC = 0;
        if (is_variant) for (int i = 1; i < nals_new; i++) nAC += sh.ac[i];
        if (is_variant && !nAC && (P.call_flag & BCFGPU_CALL_VARONLY)) {
            cs->ret = 0; cs->nals_new = 0; cs->als_new = 0; cs->an = 0; cs->qual = 0; cs->qual_missing = 0; cs->pl_dropped = 0;
            for (int i = 0; i < 5; ++i) { cs->als_map[i] = -1; cs->ac[i] = 0; }
            return;
        }
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
    }
}

void launch_mcall(const McallParams &p, hipStream_t s)
{
    if (p.n_sites == 0) return;
    const int ngrp = p.n_grp > 1 ? p.n_grp : 1;
    const size_t lds = (size_t)ngrp * 5 * sizeof(float) + (size_t)ngrp * 2 * sizeof(int);
    hipLaunchKernelGGL(mcall_kernel, dim3(p.n_sites), dim3(WG), lds, s, p);
}

}  // namespace bcfgpu

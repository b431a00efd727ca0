/*  mcall_oracle.c -- ORACLE (test infrastructure only): CPU restatement of the
 *  multiallelic caller `bcftools call -m`, one site at a time, in the
 *  reference's own order of operations and floating-point types.
 *
 *  Follows (paths in the bcftools source tree):
 *     pl2p table                 mcall.c:56-61     call_init_pl2p
 *     prior                      mcall.c:396-416   mcall_init (Watterson factor)
 *     set_pdg()                  mcall.c:451-544
 *     trimming_maps()            mcall.c:547-570   init_allele_trimming_maps
 *     find_best_alleles()        mcall.c:591-710   mcall_find_best_alleles
 *     set_ref_genotypes()        mcall.c:713-743
 *     call_genotypes()           mcall.c:745-886
 *     trim_pls()                 mcall.c:1158-1194 mcall_trim_and_update_PLs
 *     mcall_site()               mcall.c:1430-1684 mcall
 *  The record-level plumbing (bcf_get_xxx / bcf_update_xxx) is replaced by the SoA
 *  planes of include/bcfgpu.h.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <assert.h>
#include "bcforacle.h"

#define GT_MISSING  BCFGPU_GT_MISSING
#define GT_VEND     BCFGPU_GT_VECTOR_END

typedef struct {
    double ref_lk, max_lk, lk_sum;
    float qsum[8];
    int nsmpl; const int *smpl;
    uint32_t nals, als;
} grp_t;

typedef struct {
    int nsmpl;
    const uint8_t *ploidy;      /* may be NULL */
    double pl2p[256];
    double theta;               /* log-scaled prior or 0 */
    uint32_t flag, output_tags;
    int unseen;
    /* per-site scratch */
    int32_t *PLs; double *pdg; int32_t *gts; float *GPs; int32_t *GQs;
    int ac[8], als_map[8], pl_map[64];
    int als_new, nals_new;
    uint32_t grp_nals;          /* nals of the group passed to call_genotypes */
} call_t;

static inline int alleles2gt(int a, int b) { return a > b ? a*(a+1)/2 + b : b*(b+1)/2 + a; }
static inline void gt2alleles(int igt, int *a, int *b)
{
    int k = 0, dk = 1;
    while (k < igt) { dk++; k += dk; }
    *b = dk - 1; *a = igt - k + *b;
}
static inline void float_set_missing(float *f)    { uint32_t u = 0x7F800001; memcpy(f, &u, 4); }
static inline void float_set_vector_end(float *f) { uint32_t u = 0x7F800002; memcpy(f, &u, 4); }

/* mcall.c:451-544 */
static void set_pdg(double *pl2p, int *PLs, double *pdg, int n_smpl, int n_gt, int unseen)
{
    int i, j, nals;
    gt2alleles(n_gt-1, &i, &nals);
    assert(i == nals);
    nals++;
    for (i = 0; i < n_smpl; i++) {
        double sum = 0;
        for (j = 0; j < n_gt; j++) {
            if (PLs[j] == BCFGPU_INT32_VECTOR_END) { j = 0; break; }
            if (PLs[j] == BCFGPU_INT32_MISSING) break;
            pdg[j] = PLs[j] < 256 ? pl2p[PLs[j]] : pow(10., -PLs[j]/10.);
            sum += pdg[j];
        }
        if (j == 0) {
            j = sum = n_gt;
        } else if (j < n_gt && unseen < 0) {
            sum = 0;
            for (j = 0; j < n_gt; j++) {
                assert(PLs[j] != BCFGPU_INT32_VECTOR_END);
                if (PLs[j] == BCFGPU_INT32_MISSING) PLs[j] = 255;
                pdg[j] = PLs[j] < 256 ? pl2p[PLs[j]] : pow(10., -PLs[j]/10.);
                sum += pdg[j];
            }
        }
        if (j < n_gt) {
            int ia, ib, k;
            j = 0;
            sum = 0;
            for (ia = 0; ia < nals; ia++) {
                for (ib = 0; ib <= ia; ib++) {
                    if (PLs[j] == BCFGPU_INT32_MISSING) {
                        k = alleles2gt(ia, unseen);
                        if (PLs[k] == BCFGPU_INT32_MISSING) k = alleles2gt(ib, unseen);
                        if (PLs[k] == BCFGPU_INT32_MISSING) k = alleles2gt(unseen, unseen);
                        if (PLs[k] == BCFGPU_INT32_MISSING) PLs[j] = 255;
                        else PLs[j] = PLs[k];
                    }
                    pdg[j] = pl2p[PLs[j]];
                    sum += pdg[j];
                    j++;
                }
            }
        }
        if (sum == n_gt) { for (j = 0; j < n_gt; j++) pdg[j] = 0; }
        else for (j = 0; j < n_gt; j++) pdg[j] /= sum;
        PLs += n_gt;
        pdg += n_gt;
    }
}

/* mcall.c:547-570 */
static void trimming_maps(call_t *call, int nals_ori, int als_out)
{
    int i, j, nout = 0;
    for (i = 0; i < nals_ori; i++) {
        if (als_out & (1<<i)) call->als_map[i] = nout++;
        else call->als_map[i] = -1;
    }
    int k = 0, l = 0;
    for (i = 0; i < nals_ori; i++)
        for (j = 0; j <= i; j++) {
            if ((als_out & (1<<i)) && (als_out & (1<<j))) call->pl_map[k++] = l;
            l++;
        }
}

static inline double logsumexp2(double a, double b)
{
    if (a > b) return log(1 + exp(b-a)) + a;
    else       return log(1 + exp(a-b)) + b;
}

#define UPDATE_MAX_LKs(als,sum) { \
     if ( max_lk<lk_tot && lk_tot_set ) { max_lk = lk_tot; max_als = (als); } \
     if ( sum ) lk_sum = logsumexp2(lk_tot,lk_sum); \
}

/* mcall.c:591-710 */
static int find_best_alleles(call_t *call, int nals, grp_t *grp)
{
    int ia, ib, ic;
    int max_als = 0;
    double ref_lk = -HUGE_VAL, max_lk = -HUGE_VAL;
    double lk_sum = -HUGE_VAL;
    int nsmpl = grp->nsmpl;
    int ngts  = nals*(nals+1)/2;

    for (ia = 0; ia < nals; ia++) {
        double lk_tot = 0;
        int lk_tot_set = 0;
        int iaa = (ia+1)*(ia+2)/2-1;
        int ismpl;
        for (ismpl = 0; ismpl < nsmpl; ismpl++) {
            double *pdg = call->pdg + grp->smpl[ismpl]*ngts + iaa;
            if (*pdg) { lk_tot += log(*pdg); lk_tot_set = 1; }
        }
        if (ia == 0) ref_lk = lk_tot;
        else lk_tot += call->theta;
        UPDATE_MAX_LKs(1<<ia, ia>0 && lk_tot_set);
    }
    if (nals > 1) {
        for (ia = 0; ia < nals; ia++) {
            if (grp->qsum[ia] == 0) continue;
            int iaa = (ia+1)*(ia+2)/2-1;
            for (ib = 0; ib < ia; ib++) {
                if (grp->qsum[ib] == 0) continue;
                double lk_tot = 0;
                int lk_tot_set = 0;
                double fa  = grp->qsum[ia]/(grp->qsum[ia] + grp->qsum[ib]);
                double fb  = grp->qsum[ib]/(grp->qsum[ia] + grp->qsum[ib]);
                double fa2 = fa*fa;
                double fb2 = fb*fb;
                double fab = 2*fa*fb;
                int is, ibb = (ib+1)*(ib+2)/2-1, iab = iaa - ia + ib;
                for (is = 0; is < nsmpl; is++) {
                    int ismpl = grp->smpl[is];
                    double *pdg = call->pdg + ismpl*ngts;
                    double val = 0;
                    if (!call->ploidy || call->ploidy[ismpl] == 2)
                        val = fa2*pdg[iaa] + fb2*pdg[ibb] + fab*pdg[iab];
                    else if (call->ploidy && call->ploidy[ismpl] == 1)
                        val = fa*pdg[iaa] + fb*pdg[ibb];
                    if (val) { lk_tot += log(val); lk_tot_set = 1; }
                }
                if (ia != 0) lk_tot += call->theta;
                if (ib != 0) lk_tot += call->theta;
                UPDATE_MAX_LKs(1<<ia|1<<ib, lk_tot_set);
            }
        }
    }
    if (nals > 2) {
        for (ia = 0; ia < nals; ia++) {
            if (grp->qsum[ia] == 0) continue;
            int iaa = (ia+1)*(ia+2)/2-1;
            for (ib = 0; ib < ia; ib++) {
                if (grp->qsum[ib] == 0) continue;
                int ibb = (ib+1)*(ib+2)/2-1;
                int iab = iaa - ia + ib;
                for (ic = 0; ic < ib; ic++) {
                    if (grp->qsum[ic] == 0) continue;
                    double lk_tot = 0;
                    int lk_tot_set = 0;
                    double fa  = grp->qsum[ia]/(grp->qsum[ia] + grp->qsum[ib] + grp->qsum[ic]);
                    double fb  = grp->qsum[ib]/(grp->qsum[ia] + grp->qsum[ib] + grp->qsum[ic]);
                    double fc  = grp->qsum[ic]/(grp->qsum[ia] + grp->qsum[ib] + grp->qsum[ic]);
                    double fa2 = fa*fa;
                    double fb2 = fb*fb;
                    double fc2 = fc*fc;
                    double fab = 2*fa*fb, fac = 2*fa*fc, fbc = 2*fb*fc;
                    int is, icc = (ic+1)*(ic+2)/2-1;
                    int iac = iaa - ia + ic, ibc = ibb - ib + ic;
                    for (is = 0; is < nsmpl; is++) {
                        int ismpl = grp->smpl[is];
                        double *pdg = call->pdg + ismpl*ngts;
                        double val = 0;
                        if (!call->ploidy || call->ploidy[ismpl] == 2)
                            val = fa2*pdg[iaa] + fb2*pdg[ibb] + fc2*pdg[icc] + fab*pdg[iab] + fac*pdg[iac] + fbc*pdg[ibc];
                        else if (call->ploidy && call->ploidy[ismpl] == 1)
                            val = fa*pdg[iaa] + fb*pdg[ibb] + fc*pdg[icc];
                        if (val) { lk_tot += log(val); lk_tot_set = 1; }
                    }
                    if (ia != 0) lk_tot += call->theta;
                    if (ib != 0) lk_tot += call->theta;
                    if (ic != 0) lk_tot += call->theta;
                    UPDATE_MAX_LKs(1<<ia|1<<ib|1<<ic, lk_tot_set);
                }
            }
        }
    }
    int i, n = 0;
    for (i = 0; i < nals; i++) if (max_als & 1<<i) n++;
    grp->max_lk = max_lk;
    grp->ref_lk = ref_lk;
    grp->lk_sum = lk_sum;
    grp->als  = max_als;
    grp->nals = n;
    return n;
}

/* mcall.c:713-743 */
static void set_ref_genotypes(call_t *call, int nals_ori)
{
    int i;
    int ngts  = nals_ori*(nals_ori+1)/2;
    int nsmpl = call->nsmpl;
    for (i = 0; i < nals_ori; i++) call->ac[i] = 0;
    int *gts    = call->gts;
    double *pdg = call->pdg;
    int isample;
    for (isample = 0; isample < nsmpl; isample++) {
        int ploidy = call->ploidy ? call->ploidy[isample] : 2;
        for (i = 0; i < ngts; i++) if (pdg[i] != 0.0) break;
        if (i == ngts || !ploidy) {
            gts[0] = GT_MISSING;
            gts[1] = ploidy == 2 ? GT_MISSING : GT_VEND;
        } else {
            gts[0] = 0;
            gts[1] = ploidy == 2 ? 0 : GT_VEND;
            call->ac[0] += ploidy;
        }
        gts += 2;
        pdg += ngts;
    }
}

/* mcall.c:745-886 */
static void call_genotypes(call_t *call, int nals_ori, grp_t *grp)
{
    int ia, ib, i;
    int ngts_ori = nals_ori*(nals_ori+1)/2;
    int ngts_new = call->nals_new*(call->nals_new+1)/2;
    int nsmpl = grp->nsmpl;
    int is;
    for (is = 0; is < nsmpl; is++) {
        int ismpl   = grp->smpl[is];
        double *pdg = call->pdg + ismpl*ngts_ori;
        float *gps  = call->GPs + ismpl*ngts_new;
        int *gts    = call->gts + ismpl*2;
        int ploidy = call->ploidy ? call->ploidy[ismpl] : 2;
        assert(ploidy >= 0 && ploidy <= 2);
        if (!ploidy) {
            gts[0] = GT_MISSING;
            gts[1] = GT_VEND;
            gps[0] = -1;
            continue;
        }
        for (i = 0; i < ngts_ori; i++) if (pdg[i] != 0.0) break;
        if (i == ngts_ori) {
            gts[0] = GT_MISSING;
            gts[1] = ploidy == 2 ? GT_MISSING : GT_VEND;
            gps[0] = -1;
            continue;
        }
        gts[0] = 0;
        gts[1] = ploidy == 2 ? 0 : GT_VEND;
        double best_lk = 0;
        for (ia = 0; ia < nals_ori; ia++) {
            if (!(grp->als & 1<<ia)) continue;
            int iaa = (ia+1)*(ia+2)/2-1;
            double lk = ploidy == 2 ? pdg[iaa]*grp->qsum[ia]*grp->qsum[ia] : pdg[iaa]*grp->qsum[ia];
            int igt  = ploidy == 2 ? alleles2gt(call->als_map[ia], call->als_map[ia]) : call->als_map[ia];
            gps[igt] = lk;
            if (best_lk < lk) {
                best_lk = lk;
                gts[0] = call->als_map[ia];
            }
        }
        if (ploidy == 2) {
            gts[1] = gts[0];
            for (ia = 0; ia < nals_ori; ia++) {
                if (!(grp->als & 1<<ia)) continue;
                int iaa = (ia+1)*(ia+2)/2-1;
                for (ib = 0; ib < ia; ib++) {
                    if (!(grp->als & 1<<ib)) continue;
                    int iab = iaa - ia + ib;
                    double lk = 2*pdg[iab]*grp->qsum[ia]*grp->qsum[ib];
                    int igt  = alleles2gt(call->als_map[ia], call->als_map[ib]);
                    gps[igt] = lk;
                    if (best_lk < lk) {
                        best_lk = lk;
                        gts[0] = call->als_map[ib];
                        gts[1] = call->als_map[ia];
                    }
                }
            }
        } else
            gts[1] = GT_VEND;
        call->ac[gts[0]]++;
        if (gts[1] != GT_VEND) call->ac[gts[1]]++;
    }
    if (!(call->output_tags & (BCFGPU_CALL_FMT_GQ|BCFGPU_CALL_FMT_GP))) return;
    double max, sum;
    for (is = 0; is < nsmpl; is++) {
        int ismpl  = grp->smpl[is];
        float *gps = call->GPs + ismpl*ngts_new;
        int nmax;
        if (call->ploidy) {
            if (call->ploidy[ismpl] == 2) nmax = ngts_new;
            else if (call->ploidy[ismpl] == 1) nmax = grp->nals;
            else nmax = 0;
        } else nmax = ngts_new;
        max = gps[0];
        if (max < 0 || nmax == 0) {
            if (call->output_tags & BCFGPU_CALL_FMT_GP) {
                for (i = 0; i < nmax; i++) gps[i] = 0;
                if (nmax == 0) { float_set_missing(&gps[i]); nmax++; }
                if (nmax < ngts_new) float_set_vector_end(&gps[nmax]);
            }
            call->GQs[ismpl] = 0;
            continue;
        }
        sum = gps[0];
        for (i = 1; i < nmax; i++) {
            if (max < gps[i]) max = gps[i];
            sum += gps[i];
        }
        max = -4.34294*log(1 - max/sum);
        call->GQs[ismpl] = max <= INT8_MAX ? max : INT8_MAX;
        if (call->output_tags & BCFGPU_CALL_FMT_GP) {
            for (i = 0; i < nmax; i++) gps[i] = gps[i]/sum;
            for (; i < ngts_new; i++) float_set_vector_end(&gps[i]);
        }
    }
}

/* mcall.c:1158-1194 (all_diploid is never set by vcfcall.c, so the copy always runs) */
static void trim_pls(call_t *call, int nals_ori, int nals_new)
{
    int npls_src = nals_ori*(nals_ori+1)/2;
    int npls_dst = nals_new*(nals_new+1)/2;
    int *pls_src = call->PLs, *pls_dst = call->PLs;
    int nsmpl = call->nsmpl;
    int isample, ia;
    for (isample = 0; isample < nsmpl; isample++) {
        int ploidy = call->ploidy ? call->ploidy[isample] : 2;
        if (ploidy == 2) {
            for (ia = 0; ia < npls_dst; ia++)
                pls_dst[ia] = pls_src[call->pl_map[ia]];
        } else if (ploidy == 1) {
            for (ia = 0; ia < nals_new; ia++) {
                int isrc = (ia+1)*(ia+2)/2-1;
                pls_dst[ia] = pls_src[call->pl_map[isrc]];
            }
            if (ia < npls_dst) pls_dst[ia] = BCFGPU_INT32_VECTOR_END;
        } else {
            pls_dst[0] = BCFGPU_INT32_MISSING;
            pls_dst[1] = BCFGPU_INT32_VECTOR_END;
        }
        pls_src += npls_src;
        pls_dst += npls_dst;
    }
}

/* mcall(), mcall.c:1430-1684, for site `is` of the tile */
static int mcall_site(call_t *call, const bcfgpu_cfg *cfg, const bcfgpu_call_in *in, const bcfgpu_call_out *out,
                      int is, grp_t *grps, int ngrp)
{
    const int nsmpl = call->nsmpl;
    const size_t S = nsmpl;
    int i, j;
    int unseen = in->unseen[is];
    int nals_ori = in->nals[is];
    int ngts_ori = nals_ori*(nals_ori+1)/2;
    bcfgpu_call_site *cs = &out->site[is];
    memset(cs, 0, sizeof(*cs));
    for (i = 0; i < 5; i++) cs->als_map[i] = -1;

    /* FORMAT/PL -> sample-major copy */
    for (i = 0; i < nsmpl; i++)
        for (j = 0; j < ngts_ori; j++)
            call->PLs[i*ngts_ori + j] = in->pl[((size_t)is*in->n_gt_max + j)*S + i];
    set_pdg(call->pl2p, call->PLs, call->pdg, nsmpl, ngts_ori, unseen);

    /* allele frequencies (mcall.c:1453-1535) */
    if (ngrp == 1) {
        for (i = 0; i < nals_ori; i++) grps[0].qsum[i] = i < 5 ? in->qs[(size_t)is*5 + i] : 0;
    } else {
        int nad = in->n_al_max;
        for (i = 0; i < ngrp; i++) {
            grp_t *grp = &grps[i];
            int k;
            for (j = 0; j < nals_ori; j++) grp->qsum[j] = 0;
            for (k = 0; k < grp->nsmpl; k++) {
                int ismpl = grp->smpl[k];
                float sum = 0;
                for (j = 0; j < nad; j++) {
                    int32_t v = in->ad[((size_t)is*nad + j)*S + ismpl];
                    if (v == BCFGPU_INT32_VECTOR_END) break;
                    if (v != BCFGPU_INT32_MISSING) sum += v;
                }
                if (sum) {
                    for (j = 0; j < nad; j++) {
                        int32_t v = in->ad[((size_t)is*nad + j)*S + ismpl];
                        if (v == BCFGPU_INT32_VECTOR_END) break;
                        if (v != BCFGPU_INT32_MISSING) grp->qsum[j] += v/sum;
                    }
                }
            }
        }
    }
    /* reference-panel prior, -F AN,AC (mcall.c:1506-1527) */
    if (in->prior_an && in->prior_ac && in->prior_an[is] != BCFGPU_INT32_MISSING) {
        int an = in->prior_an[is];
        const int32_t *acv = in->prior_ac + (size_t)is*4;
        int nac = 0;
        while (nac < 4 && acv[nac] != BCFGPU_INT32_VECTOR_END) nac++;
        if (an > 0 && nac == nals_ori-1) {
            int ac0 = an;
            for (i = 0; i < nals_ori-1; i++) {
                if (acv[i] == BCFGPU_INT32_VECTOR_END) break;
                if (acv[i] == BCFGPU_INT32_MISSING) continue;
                ac0 -= acv[i];
                for (j = 0; j < ngrp; j++)
                    grps[j].qsum[i+1] = (grps[j].qsum[i+1] + 0.5*acv[i]) / (grps[j].nsmpl + 0.5*an);
            }
            if (ac0 < 0) { cs->ret = -1; return -1; }
            for (j = 0; j < ngrp; j++)
                grps[j].qsum[0] = (grps[j].qsum[0] + 0.5*ac0) / (grps[j].nsmpl + 0.5*an);
        }
    }
    for (j = 0; j < ngrp; j++) {
        float sum = 0;
        for (i = 0; i < nals_ori; i++) sum += grps[j].qsum[i];
        if (sum) for (i = 0; i < nals_ori; i++) grps[j].qsum[i] /= sum;
    }

    call->als_new = 0;
    double ref_lk = -HUGE_VAL, lk_sum = -HUGE_VAL, max_qual = -HUGE_VAL;
    for (j = 0; j < ngrp; j++) {
        grp_t *grp = &grps[j];
        find_best_alleles(call, nals_ori, grp);
        call->als_new |= grp->als;
        if (grp->max_lk == -HUGE_VAL) continue;
        double qual = -4.343*(grp->ref_lk - logsumexp2(grp->lk_sum, grp->ref_lk));
        if (max_qual < qual) {
            max_qual = qual;
            lk_sum = grp->lk_sum;
            ref_lk = grp->ref_lk;
        }
    }
    if (!(call->als_new & 1)) call->als_new |= 1;
    int is_variant = call->als_new == 1 ? 0 : 1;
    if ((call->flag & BCFGPU_CALL_VARONLY) && !is_variant) { cs->ret = 0; return 0; }

    call->nals_new = 0;
    for (i = 0; i < nals_ori; i++) {
        if (i > 0 && i == unseen) continue;
        if (call->flag & BCFGPU_CALL_KEEPALT) call->als_new |= 1<<i;
        if (call->als_new & (1<<i)) call->nals_new++;
    }
    trimming_maps(call, nals_ori, call->als_new);

    int nAC = 0;
    int ngts_new = call->nals_new*(call->nals_new+1)/2;
    if (call->als_new == 1) {
        set_ref_genotypes(call, nals_ori);
        cs->pl_dropped = 1;
    } else if (!is_variant) {
        set_ref_genotypes(call, nals_ori);
        trim_pls(call, nals_ori, call->nals_new);
    } else {
        for (i = 0; i < call->nals_new; i++) call->ac[i] = 0;
        if (call->output_tags & (BCFGPU_CALL_FMT_GQ|BCFGPU_CALL_FMT_GP)) {
            memset(call->GPs, 0, sizeof(float)*nsmpl*ngts_new);
            memset(call->GQs, 0, sizeof(int32_t)*nsmpl);
        }
        for (i = 0; i < ngrp; i++) call_genotypes(call, nals_ori, &grps[i]);
        for (i = 1; i < call->nals_new; i++) nAC += call->ac[i];
        if (!nAC && (call->flag & BCFGPU_CALL_VARONLY)) { cs->ret = 0; return 0; }
        if ((call->output_tags & BCFGPU_CALL_FMT_GP) && out->gp)
            for (i = 0; i < nsmpl; i++)
                for (j = 0; j < ngts_new; j++)
                    out->gp[((size_t)is*in->n_gt_max + j)*S + i] = call->GPs[i*ngts_new + j];
        if ((call->output_tags & BCFGPU_CALL_FMT_GQ) && out->gq)
            for (i = 0; i < nsmpl; i++) out->gq[(size_t)is*S + i] = call->GQs[i];
        trim_pls(call, nals_ori, call->nals_new);
    }

    /* QUAL (mcall.c:1630-1645) */
    float qual = 0;
    if (nAC) qual = max_qual;
    else {
        if (lk_sum != -HUGE_VAL) qual = -4.343*(lk_sum - logsumexp2(lk_sum, ref_lk));
        else if (call->ac[0]) qual = call->theta ? -4.343*call->theta : 0;
        else cs->qual_missing = 1;
    }
    cs->qual = qual;
    for (i = 0; i < 5; i++) cs->ac[i] = i < call->nals_new ? call->ac[i] : 0;
    nAC += call->ac[0];
    cs->an = nAC;
    cs->nals_new = call->nals_new;
    cs->als_new = call->als_new;
    for (i = 0; i < nals_ori && i < 5; i++) cs->als_map[i] = call->als_map[i];
    cs->ret = call->nals_new;

    /* DP4, MQ and PV4 from I16 (mcall.c:1659-1679) */
    if (in->i16) {
        const float *a16 = in->i16 + (size_t)is*16;
        cs->has_i16 = 1;
        for (i = 0; i < 4; i++) cs->dp4[i] = (int32_t) a16[i];
        {   /* float division truncated to int32; at depth 0 the reference converts a NaN, which x86 turns into
             * INT32_MIN = bcf_int32_missing: MQ is then printed as '.' */
            const float dsum = a16[0] + a16[1] + a16[2] + a16[3];
            cs->mq = dsum != 0 ? (int32_t)((a16[8] + a16[10]) / dsum) : BCFGPU_INT32_MISSING;
        }
        if (call->output_tags & BCFGPU_CALL_FMT_PV4) {
            double p4[4]; int tested;
            if (orc_test16(a16, p4, &tested) >= 0 && tested) {
                cs->pv4_tested = 1;
                for (i = 0; i < 4; i++) cs->pv4[i] = (float) p4[i];
            }
        }
    }

    for (i = 0; i < nsmpl; i++) {
        out->gt[((size_t)is*2 + 0)*S + i] = (int8_t) call->gts[2*i];
        out->gt[((size_t)is*2 + 1)*S + i] = (int8_t) call->gts[2*i+1];
    }
    if (!cs->pl_dropped && out->pl)
        for (i = 0; i < nsmpl; i++)
            for (j = 0; j < ngts_new; j++)
                out->pl[((size_t)is*in->n_gt_max + j)*S + i] = call->PLs[i*ngts_new + j];
    return call->nals_new;
}

int orc_mcall(const bcfgpu_cfg *cfg, const bcfgpu_call_in *in, const bcfgpu_call_out *out)
{
    call_t call;
    memset(&call, 0, sizeof(call));
    const int S = cfg->n_smpl;
    int i;
    call.nsmpl = S;
    call.ploidy = in->ploidy;
    call.flag = cfg->call_flag;
    call.output_tags = cfg->output_tags;
    for (i = 0; i < 256; i++) call.pl2p[i] = pow(10., -i/10.);
    /* the prior, mcall.c:396-416.  vcfcall.c:654-655 initialises every sample's
     * ploidy to ploidy_max before mcall_init, hence n = ploidy_max * nsamples */
    call.theta = cfg->call_theta;
    if (call.theta > 0) {
        int pm = cfg->ploidy_max > 0 ? cfg->ploidy_max : 2;
        int n = pm * S;
        double aM = 1;
        for (i = 2; i < n; i++) aM += 1./i;
        call.theta *= aM;
        if (call.theta >= 1) call.theta = 0.99;
        call.theta = log(call.theta);
    }
    int ngmax = in->n_gt_max > 0 ? in->n_gt_max : BCFGPU_MAX_PL;
    call.PLs = (int32_t*) malloc(sizeof(int32_t)*S*ngmax);
    call.pdg = (double*) malloc(sizeof(double)*S*ngmax);
    call.gts = (int32_t*) calloc(S*2, sizeof(int32_t));
    call.GPs = (float*) calloc((size_t)S*ngmax, sizeof(float));
    call.GQs = (int32_t*) calloc(S, sizeof(int32_t));

    /* sample groups (mcall.c:250-349) */
    int ngrp = 1;
    if (in->grp && cfg->n_grp > 1) ngrp = cfg->n_grp;
    grp_t *grps = (grp_t*) calloc(ngrp, sizeof(grp_t));
    int *smpl = (int*) malloc(sizeof(int)*S);
    if (ngrp == 1) {
        for (i = 0; i < S; i++) smpl[i] = i;
        grps[0].smpl = smpl; grps[0].nsmpl = S;
    } else {
        int g, k = 0;
        for (g = 0; g < ngrp; g++) {
            grps[g].smpl = smpl + k;
            for (i = 0; i < S; i++) if (in->grp[i] == g) { smpl[k++] = i; grps[g].nsmpl++; }
        }
    }
    int is, ret = 0;
    for (is = 0; is < in->n_sites; is++) {
        if (in->nals[is] > 5 || in->nals[is]*(in->nals[is]+1)/2 > ngmax) { ret = BCFGPU_E_ARG; break; }
        /* record-loop prologue, vcfcall.c:1112-1115: with -v a REF-only record never reaches mcall() */
        if ((cfg->call_flag & BCFGPU_CALL_VARONLY) && (in->nals[is] == 1 || (in->nals[is] == 2 && in->unseen[is] > 0))) {
            int k;
            memset(&out->site[is], 0, sizeof(out->site[is]));
            for (k = 0; k < 5; k++) out->site[is].als_map[k] = -1;
            continue;
        }
        mcall_site(&call, cfg, in, out, is, grps, ngrp);
    }
    free(grps); free(smpl);
    free(call.PLs); free(call.pdg); free(call.gts); free(call.GPs); free(call.GQs);
    return ret;
}

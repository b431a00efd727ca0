/*  bam2bcf_oracle.c -- ORACLE (test infrastructure only): CPU restatement of the
 *  mpileup side of the hot path, in the reference's own loop order
 *  (per site -> per sample -> per read), double precision where the reference
 *  uses double and float where it uses float.
 *
 *  Follows (paths in the bcftools source tree):
 *     glfgen()          bam2bcf.c:147-258   bcf_call_glfgen
 *     orc_calc_vdb()    bam2bcf.c:281-342   calc_vdb
 *     mann_whitney_*    bam2bcf.c:369-385 + mw.h (the table is the same recursion, printed %.17f)
 *     orc_calc_mwu_bias bam2bcf.c:440-484   calc_mwu_bias
 *     calc_segbias()    bam2bcf.c:494-530   calc_SegBias
 *     combine()         bam2bcf.c:558-754   bcf_call_combine
 *     orc_mpileup()     mpileup.c:343-347   per-site order of operations
 *  Input is the SoA tile of include/bcfgpu.h instead of bam_pileup1_t[].
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include <assert.h>
#include "bcforacle.h"

#define DEF_MAPQ 20     /* bam2bcf.c:39 */
#define CAP_DIST 25     /* bam2bcf.c:41 */
#define CALL_DEFTHETA 0.83

/* htslib seq_nt16_int: 4-bit IUPAC code -> 2-bit base, 4 for anything ambiguous */
static const int nt16_int[16] = { 4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4 };

typedef struct {
    int ref_pos[BCFGPU_NPOS], alt_pos[BCFGPU_NPOS];
    int ref_mq[BCFGPU_NQUAL], alt_mq[BCFGPU_NQUAL];
    int ref_bq[BCFGPU_NQUAL], alt_bq[BCFGPU_NQUAL];
    int fwd_mqs[BCFGPU_NQUAL], rev_mqs[BCFGPU_NQUAL];
} site_hist;

/* bcf_call_glfgen, bam2bcf.c:147-258 */
static int glfgen(const bcfgpu_cfg *cfg, const orc_errmod *em, site_hist *h,
                  int _n, const uint32_t *rd, const uint8_t *epos_arr, const uint32_t *aux_arr,
                  int ref_base, orc_callret *r, uint16_t **bases_p, int *max_bases, int *err)
{
    int i, n, ref4, is_indel, ori_depth = 0;
    memset(r, 0, sizeof(*r));
    if (ref_base >= 0) { ref4 = nt16_int[ref_base]; is_indel = 0; }
    else ref4 = 4, is_indel = 1;
    if (_n == 0) { r->n = -1; return -1; }
    if (*max_bases < _n) {                  /* "enlarge the bases array if necessary", bam2bcf.c:164-168 */
        *max_bases = _n;
        *bases_p = (uint16_t*) realloc(*bases_p, 2 * (size_t)*max_bases);
    }
    uint16_t *bases = *bases_p;
    for (i = n = 0; i < _n; ++i) {
        uint32_t w = rd[i];
        int q, b, mapQ, baseQ, is_diff, min_dist, seqQ;
        int is_rev = (w & BCFGPU_RD_REV) ? 1 : 0;
        int nt16 = (w>>16) & 0xf;
        if (w & BCFGPU_RD_SKIP) continue;
        if ((w & BCFGPU_RD_DEL) && !is_indel) continue;
        ++ori_depth;
        if (is_indel) {
            uint32_t aux = aux_arr[i];
            b = aux>>16 & 0x3f;
            baseQ = q = aux & 0xff;
            if (q < cfg->min_baseQ) b = 0, q = (int)(w & 0xff);
            seqQ = aux>>8 & 0xff;
            is_diff = (b != 0);
        } else {
            b = nt16;
            b = nt16_int[b ? b : ref_base];
            baseQ = q = (int)(w & 0xff);
            if (q < cfg->min_baseQ) continue;
            seqQ = 99;
            is_diff = (ref4 < 4 && b == ref4) ? 0 : 1;
        }
        mapQ = (int)((w>>8) & 0xff);
        mapQ = mapQ < 255 ? mapQ : DEF_MAPQ;
        if (!mapQ) r->mq0++;
        if (q > seqQ) q = seqQ;
        mapQ = mapQ < cfg->capQ ? mapQ : cfg->capQ;
        if (q > mapQ) q = mapQ;
        if (q > 63) q = 63;
        if (q < 4) q = 4;
        bases[n++] = q<<5 | is_rev<<4 | b;
        if ((cfg->fmt_flag & (BCFGPU_INFO_SCR|BCFGPU_FMT_SCR)) && (w & BCFGPU_RD_SCLIP)) r->SCR++;
        if (b < 4) {
            r->QS[b] += q;
            if (is_rev) r->ADR[b]++; else r->ADF[b]++;
        }
        ++r->anno[0<<2|is_diff<<1|is_rev];
        min_dist = (int)(w>>24);            /* min(qpos, l_qseq-1-qpos), saturated at 255 by the packer */
        if (min_dist > CAP_DIST) min_dist = CAP_DIST;
        r->anno[1<<2|is_diff<<1|0] += baseQ;
        r->anno[1<<2|is_diff<<1|1] += baseQ * baseQ;
        r->anno[2<<2|is_diff<<1|0] += mapQ;
        r->anno[2<<2|is_diff<<1|1] += mapQ * mapQ;
        r->anno[3<<2|is_diff<<1|0] += min_dist;
        r->anno[3<<2|is_diff<<1|1] += min_dist * min_dist;

        /* bias tests */
        if (baseQ > 59) baseQ = 59;
        if (mapQ > 59) mapQ = 59;
        int epos = 0;
        if (cfg->fmt_flag & (BCFGPU_INFO_RPB|BCFGPU_INFO_VDB)) epos = epos_arr[i];
        int ibq = baseQ/60. * BCFGPU_NQUAL;
        int imq = mapQ/60. * BCFGPU_NQUAL;
        if (is_rev) h->rev_mqs[imq]++; else h->fwd_mqs[imq]++;
        if (nt16 == ref_base) { h->ref_pos[epos]++; h->ref_bq[ibq]++; h->ref_mq[imq]++; }
        else                  { h->alt_pos[epos]++; h->alt_bq[ibq]++; h->alt_mq[imq]++; }
    }
    r->ori_depth = ori_depth;
    int e = orc_errmod_cal(em, n, 5, bases, r->p);
    if (e) { *err = e; return -1; }
    r->n = n;
    return n;
}

/* calc_vdb, bam2bcf.c:281-342 */
double orc_calc_vdb(const int *pos, int npos)
{
    const int readlen = 100;
    assert(npos == readlen);
    #define nparam 15
    const float param[nparam][3] = { {3,0.079,18}, {4,0.09,19.8}, {5,0.1,20.5}, {6,0.11,21.5},
        {7,0.125,21.6}, {8,0.135,22}, {9,0.14,22.2}, {10,0.153,22.3}, {15,0.19,22.8},
        {20,0.22,23.2}, {30,0.26,23.4}, {40,0.29,23.5}, {50,0.35,23.65}, {100,0.5,23.7},
        {200,0.7,23.7} };
    int i, dp = 0;
    float mean_pos = 0, mean_diff = 0;
    for (i = 0; i < npos; i++) {
        if (!pos[i]) continue;
        dp += pos[i];
        mean_pos += pos[i]*i;
    }
    if (dp < 2) return HUGE_VAL;
    mean_pos /= dp;
    for (i = 0; i < npos; i++) {
        if (!pos[i]) continue;
        mean_diff += pos[i] * fabs(i - mean_pos);
    }
    mean_diff /= dp;
    int ipos = mean_diff;
    if (dp == 2)
        return (2*readlen-2*(ipos+1)-1)*(ipos+1)/(readlen-1)/(readlen*0.5);
    if (dp >= 200) i = nparam;
    else {
        for (i = 0; i < nparam; i++)
            if (param[i][0] >= dp) break;
    }
    float pshift, pscale;
    if (i == nparam) { pscale = param[nparam-1][1]; pshift = param[nparam-1][2]; }
    else if (i > 0 && param[i][0] != dp) {
        pscale = (param[i-1][1] + param[i][1])*0.5;
        pshift = (param[i-1][2] + param[i][2])*0.5;
    } else { pscale = param[i][1]; pshift = param[i][2]; }
    return 0.5*orc_kf_erfc(-(mean_diff-pshift)*pscale);
    #undef nparam
}

/* bam2bcf.c:369-385; mw.h holds exactly these values for n,m in 2..7, U<50 */
static double mann_whitney_1947_(int n, int m, int U)
{
    if (U < 0) return 0;
    if (n == 0 || m == 0) return U == 0 ? 1 : 0;
    return (double)n/(n+m)*mann_whitney_1947_(n-1,m,U-m) + (double)m/(n+m)*mann_whitney_1947_(n,m-1,U);
}
static double mw_tab[6][6][50];
static int mw_tab_ready = 0;
static double mann_whitney_1947(int n, int m, int U)
{
    if (!mw_tab_ready) {
        int i, j, k;
        for (i = 2; i < 8; i++) for (j = 2; j < 8; j++) for (k = 0; k < 50; k++)
            mw_tab[i-2][j-2][k] = mann_whitney_1947_(i, j, k);
        mw_tab_ready = 1;
    }
    assert(n >= 2 && m >= 2);
    return (n < 8 && m < 8 && U < 50) ? mw_tab[n-2][m-2][U] : mann_whitney_1947_(n,m,U);
}

/* calc_mwu_bias, bam2bcf.c:440-484 */
double orc_calc_mwu_bias(const int *a, const int *b, int n)
{
    int na = 0, nb = 0, i;
    double U = 0;
    for (i = 0; i < n; i++) {
        if (!a[i]) {
            if (!b[i]) continue;
            nb += b[i];
        } else if (!b[i]) {
            na += a[i];
            U  += a[i] * nb;
        } else {
            na += a[i];
            U  += a[i] * (nb + b[i]*0.5);
            nb += b[i];
        }
    }
    if (!na || !nb) return HUGE_VAL;
    if (na == 1 || nb == 1) return 1.0;
    double mean = ((double)na*nb)*0.5;
    if (na == 2 || nb == 2) return U > mean ? (2.0*mean-U)/mean : U/mean;
    double var2 = ((double)na*nb)*(na+nb+1)/12.0;
    if (na >= 8 || nb >= 8) return exp(-0.5*(U-mean)*(U-mean)/var2);
    return mann_whitney_1947(na, nb, U) * sqrt(2*M_PI*var2);
}

static inline double logsumexp2(double a, double b)
{
    if (a > b) return log(1 + exp(b-a)) + a;
    else       return log(1 + exp(a-b)) + b;
}

/* calc_SegBias, bam2bcf.c:494-530 */
static float calc_segbias(const orc_callret *bcr, int n_smpl, const double *anno)
{
    int nr = anno[2] + anno[3];
    if (!nr) return HUGE_VAL;
    int avg_dp = (anno[0] + anno[1] + nr) / n_smpl;
    double M = floor((double)nr / avg_dp + 0.5);
    if (M > n_smpl) M = n_smpl;
    else if (M == 0) M = 1;
    double f = M / 2. / n_smpl;
    double p = (double) nr / n_smpl;
    double q = (double) nr / M;
    double sum = 0;
    const double log2 = log(2.0);
    int i;
    for (i = 0; i < n_smpl; i++) {
        int oi = bcr[i].anno[2] + bcr[i].anno[3];
        double tmp;
        if (oi) {
            tmp = logsumexp2(log(2*(1-f)), log(f) + oi*log2 - q);
            tmp += log(f) + oi*log(q/p) - q + p;
        } else
            tmp = log(2*f*(1-f)*exp(-q) + f*f*exp(-2*q) + (1-f)*(1-f)) + p;
        sum += tmp;
    }
    return sum;
}

/* bcf_call_combine, bam2bcf.c:558-754.  Writes the site struct and the
 * per-sample planes of site `is`. */
static int combine(const bcfgpu_cfg *cfg, int n, const orc_callret *calls, const site_hist *h,
                   int ref_base, int is, const bcfgpu_mplp_out *out)
{
    bcfgpu_site *call = &out->site[is];
    int ref4, i, j;
    float qsum[5] = {0,0,0,0,0};
    memset(call, 0, sizeof(*call));
    if (ref_base >= 0) {
        call->ori_ref = ref4 = nt16_int[ref_base];
        if (ref4 > 4) ref4 = 4;
    } else call->ori_ref = -1, ref4 = 0;

    for (i = 0; i < n; ++i) {
        float sum = 0;
        for (j = 0; j < 4; ++j) sum += calls[i].QS[j];
        if (sum)
            for (j = 0; j < 4; j++) qsum[j] += (float)calls[i].QS[j] / sum;
    }
    float *ptr[5], *tmp;
    for (i = 0; i < 5; i++) ptr[i] = &qsum[i];
    for (i = 1; i < 4; i++)
        for (j = i; j > 0 && *ptr[j] < *ptr[j-1]; j--)
            tmp = ptr[j], ptr[j] = ptr[j-1], ptr[j-1] = tmp;

    for (i = 0; i < 5; i++) call->a[i] = -1;
    for (i = 0; i < 5; i++) call->qsum[i] = 0;
    call->unseen = -1;
    call->a[0] = ref4;
    for (i = 3, j = 1; i >= 0; i--) {
        int ipos = ptr[i] - qsum;
        if (ipos == ref4) call->qsum[0] = qsum[ipos];
        else {
            if (!qsum[ipos]) break;
            call->qsum[j] = qsum[ipos];
            call->a[j++]  = ipos;
        }
    }
    if (ref_base >= 0) {
        if (((ref4 < 4 && j < 4) || (ref4 == 4 && j < 5)) && i >= 0)
            call->unseen = j, call->a[j++] = ptr[i] - qsum;
        call->n_alleles = j;
    } else {
        call->n_alleles = j;
        if (call->n_alleles == 1) { call->ret = -1; return -1; }
    }
    const size_t S = (size_t) n;
    {
        int x, g[15], z;
        double sum_min = 0.;
        x = call->n_alleles * (call->n_alleles + 1) / 2;
        for (i = z = 0; i < call->n_alleles; ++i)
            for (j = 0; j <= i; ++j)
                g[z++] = call->a[j] * 5 + call->a[i];
        uint8_t *PL = out->pl + (size_t)is * BCFGPU_MAX_PL * S;
        for (i = 0; i < n; ++i) {
            const orc_callret *r = calls + i;
            float min = FLT_MAX;
            for (j = 0; j < x; ++j)
                if (min > r->p[g[j]]) min = r->p[g[j]];
            sum_min += min;
            for (j = 0; j < x; ++j) {
                int y = (int)(r->p[g[j]] - min + .499);
                if (y > 255) y = 255;
                PL[(size_t)j*S + i] = (uint8_t) y;
            }
        }
        uint16_t *DP4 = out->dp4 + (size_t)is * 4 * S;
        for (i = 0; i < n; i++) {
            DP4[0*S+i] = (uint16_t)(int) calls[i].anno[0];
            DP4[1*S+i] = (uint16_t)(int) calls[i].anno[1];
            DP4[2*S+i] = (uint16_t)(int) calls[i].anno[2];
            DP4[3*S+i] = (uint16_t)(int) calls[i].anno[3];
        }
        /* FMT/SP: what bcf_call2bcf derives from DP4 (bam2bcf.c:867-885) */
        if (out->sp && (cfg->fmt_flag & BCFGPU_FMT_SP))
            for (i = 0; i < n; i++)
                out->sp[(size_t)is*S + i] = (uint8_t) orc_format_sp(DP4[0*S+i], DP4[1*S+i], DP4[2*S+i], DP4[3*S+i]);
        for (i = 0; i < n; i++) {
            call->scr_tot += calls[i].SCR;
            if (out->scr) out->scr[(size_t)is*S + i] = (uint16_t) calls[i].SCR;
        }
        /* ADF/ADR reordered to allele order, with site totals (bam2bcf.c:668-697) */
        for (i = 0; i < n; i++) {
            for (j = 0; j < call->n_alleles; j++) {
                int aj = call->a[j];
                int vr = aj < 4 ? calls[i].ADR[aj] : 0;
                int vf = aj < 4 ? calls[i].ADF[aj] : 0;
                call->adr_tot[j] += vr;
                call->adf_tot[j] += vf;
                if (out->adr) out->adr[((size_t)is*5 + j)*S + i] = (uint16_t) vr;
                if (out->adf) out->adf[((size_t)is*5 + j)*S + i] = (uint16_t) vf;
            }
        }
        /* FMT/QS reordered (bam2bcf.c:698-712) */
        if (out->qs)
            for (i = 0; i < n; i++)
                for (j = 0; j < call->n_alleles; j++) {
                    int aj = call->a[j];
                    out->qs[((size_t)is*5 + j)*S + i] = (int32_t)(aj < 4 ? calls[i].QS[aj] : 0);
                }
        call->shift = (int)(sum_min + .499);
    }
    call->ori_depth = 0; call->depth = 0; call->mq0 = 0;
    for (i = 0; i < n; ++i) {
        call->depth += calls[i].anno[0] + calls[i].anno[1] + calls[i].anno[2] + calls[i].anno[3];
        call->ori_depth += calls[i].ori_depth;
        call->mq0 += calls[i].mq0;
        for (j = 0; j < 16; ++j) call->anno[j] += calls[i].anno[j];
    }
    call->seg_bias = calc_segbias(calls, n, call->anno);
    call->mwu_pos = HUGE_VAL;   /* bcf_call_t is zero-initialised in mpileup.c; the tag is only set with RPB */
    if (cfg->fmt_flag & BCFGPU_INFO_RPB)
        call->mwu_pos = orc_calc_mwu_bias(h->ref_pos, h->alt_pos, BCFGPU_NPOS);
    call->mwu_mq  = orc_calc_mwu_bias(h->ref_mq,  h->alt_mq,  BCFGPU_NQUAL);
    call->mwu_bq  = orc_calc_mwu_bias(h->ref_bq,  h->alt_bq,  BCFGPU_NQUAL);
    call->mwu_mqs = orc_calc_mwu_bias(h->fwd_mqs, h->rev_mqs, BCFGPU_NQUAL);
    call->vdb = HUGE_VAL;
    if (cfg->fmt_flag & BCFGPU_INFO_VDB)
        call->vdb = orc_calc_vdb(h->alt_pos, BCFGPU_NPOS);
    return 0;
}

int orc_mpileup(const bcfgpu_cfg *cfg_in, const bcfgpu_tile *tile, const bcfgpu_mplp_out *out, orc_callret *ret_dbg)
{
    bcfgpu_cfg cfg = *cfg_in;
    if (cfg.capQ <= 0) cfg.capQ = 60;
    double theta = cfg.errmod_theta <= 0. ? CALL_DEFTHETA : cfg.errmod_theta;
    static orc_errmod *em = NULL;
    static double em_theta = -1;
    if (!em || em_theta != theta) {
        orc_errmod_destroy(em);
        em = orc_errmod_init(1. - theta);
        em_theta = theta;
    }
    const int S = cfg.n_smpl;
    orc_callret *bcr = (orc_callret*) malloc(sizeof(orc_callret) * S);
    uint16_t *bases = NULL;
    int max_bases = 0;
    site_hist h;
    int is, s, err = 0;
    for (is = 0; is < tile->n_sites && !err; is++) {
        int ref_base = tile->is_indel ? -1 : tile->ref16[is];
        memset(&h, 0, sizeof(h));                         /* bcf_callaux_clean */
        for (s = 0; s < S; s++) {
            size_t k = (size_t)is*S + s;
            uint32_t beg = tile->plp_off[k], end = tile->plp_off[k+1];
            glfgen(&cfg, em, &h, (int)(end-beg), tile->rd + beg, tile->epos + beg,
                   tile->aux ? tile->aux + beg : NULL, ref_base, &bcr[s], &bases, &max_bases, &err);
            if (err) break;
            if (ret_dbg) ret_dbg[k] = bcr[s];
        }
        if (err) break;
        combine(&cfg, S, bcr, &h, ref_base, is, out);
    }
    free(bcr); free(bases);
    return err;
}

/*  errmod.c -- ORACLE (test infrastructure only): restatement of htslib's error
 *  model, errmod_init()/errmod_cal().
 *
 *  htslib is an external, un-vendored dependency of the reference (Makefile:96
 *  HTSDIR=../htslib, CI uses develop HEAD ~1.11-1.12), so the
 *  algorithm is restated from its published form (Li 2011, "revised MAQ error
 *  model") and anchored on the reference's call sites bam2bcf.c:51 (init with
 *  depcorr = 1-0.83) and bam2bcf.c:256 (errmod_cal(bca->e, n, 5, bases, r->p)),
 *  and on the golden test/mpileup/mpileup.3.out.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "bcforacle.h"

struct orc_errmod {
    double depcorr;
    double fk[256];
    double *beta;   /* [q<<16 | n<<8 | k], q<64 */
    double *lhet;   /* [n<<8 | k] */
};

orc_errmod *orc_errmod_init(double depcorr)
{
    const double eta = 0.03;
    orc_errmod *em = (orc_errmod*) calloc(1, sizeof(*em));
    int n, k, q;
    em->depcorr = depcorr;
    em->beta = (double*) calloc(256*256*64, sizeof(double));
    em->lhet = (double*) calloc(256*256, sizeof(double));
    /* fk: dependency decay of the k-th error on one strand */
    em->fk[0] = 1.0;
    for (n = 1; n < 256; ++n)
        em->fk[n] = pow(1. - depcorr, n) * (1.0 - eta) + eta;
    /* log binomial coefficients */
    double *lC = (double*) calloc(256*256, sizeof(double));
    for (n = 1; n != 256; ++n)
        for (k = 1; k <= n; ++k)
            lC[n<<8|k] = lgamma(n+1) - lgamma(k+1) - lgamma(n-k+1);
    /* beta: -10log10 of the ratio of successive binomial upper tails */
    for (q = 1; q != 64; ++q) {
        double e = pow(10.0, -q/10.0);
        double le = log(e);
        double le1 = log(1.0 - e);
        for (n = 1; n <= 255; ++n) {
            double *beta = em->beta + (q<<16|n<<8);
            long double sum, sum1;
            sum1 = sum = 0.0;
            for (k = n; k >= 0; --k, sum1 = sum) {
                sum = sum1 + expl(lC[n<<8|k] + k*le + (n-k)*le1);
                beta[k] = -10. / M_LN10 * logl(sum1 / sum);
            }
        }
    }
    /* lhet: log P(k of n | het) */
    for (n = 0; n < 256; ++n)
        for (k = 0; k < 256; ++k)
            em->lhet[n<<8|k] = lC[n<<8|k] - M_LN2 * n;
    free(lC);
    return em;
}

void orc_errmod_destroy(orc_errmod *em)
{
    if (!em) return;
    free(em->beta); free(em->lhet); free(em);
}

const double *orc_errmod_fk(const orc_errmod *em)   { return em->fk; }
const double *orc_errmod_beta(const orc_errmod *em) { return em->beta; }
const double *orc_errmod_lhet(const orc_errmod *em) { return em->lhet; }

/* ascending sort of the 16-bit codes (htslib uses ks_introsort(uint16_t)); any
 * correct sort gives the same sequence since equal codes are indistinguishable */
static void sort_u16(int n, uint16_t *a)
{
    int i, j;
    if (n > 24) {
        /* one level of 3-way counting on the top bits would also do; a plain
         * shell/insertion mix is ample for n<=255 */
        int gap;
        for (gap = n/2; gap > 4; gap = gap*5/11) {
            for (i = gap; i < n; ++i) {
                uint16_t t = a[i];
                for (j = i; j >= gap && a[j-gap] > t; j -= gap) a[j] = a[j-gap];
                a[j] = t;
            }
        }
    }
    for (i = 1; i < n; ++i) {
        uint16_t t = a[i];
        for (j = i; j > 0 && a[j-1] > t; --j) a[j] = a[j-1];
        a[j] = t;
    }
}

/* ---- errmod_cal's subsampling of cells deeper than 255 reads.  PARITY UNPINNED: htslib's source is not in the reference
 * tree and no golden of the reference reaches 256 reads in a cell; restated from htslib's published code:
 *   hts_drand48 (hts_os.c / os/rand.c, NetBSD's rand48): X <- (0x5DEECE66D * X + 0xB) mod 2^48 from the default seed
 *     0x1234ABCD330E (mpileup never calls hts_srand48), returning X / 2^48 AFTER the step;
 *   ks_shuffle (ksort.h): for (i = n; i > 1; --i) { j = (int)(hts_drand48() * i); swap(a[j], a[i-1]); }
 * The generator is process-wide: a cell's draw depends on every deeper-than-255 cell the run visited before it.
 * orc_errmod_deep_rule(1) replaces the draw by "the first 255 reads in pileup order", the rule the device library uses when it
 * is not told the generator's position (bcfgpu_truncated_cells). */
static uint64_t rand48_x = 0x1234ABCD330EULL;
static int deep_rule = 0;
double orc_drand48(void)
{
    rand48_x = (0x5DEECE66DULL * rand48_x + 0xBULL) & 0xFFFFFFFFFFFFULL;
    return ldexp((double)(rand48_x & 0xffff), -48) + ldexp((double)((rand48_x >> 16) & 0xffff), -32) + ldexp((double)(rand48_x >> 32), -16);
}
void orc_srand48_reset(void) { rand48_x = 0x1234ABCD330EULL; }
uint64_t orc_rand48_state(void) { return rand48_x; }
void orc_errmod_deep_rule(int rule) { deep_rule = rule; }
static void shuffle_u16(int n, uint16_t *a)
{
    int i, j;
    for (i = n; i > 1; --i) {
        uint16_t tmp;
        j = (int)(orc_drand48() * i);
        tmp = a[j]; a[j] = a[i-1]; a[i-1] = tmp;
    }
}

int orc_errmod_cal(const orc_errmod *em, int n, int m, uint16_t *bases, float *q)
{
    double fsum[16], bsum[16];
    uint32_t c[16];
    int i, j, k, w[32];

    memset(q, 0, m * m * sizeof(float));
    if (n == 0) return 0;
    if (n > 255) {                          /* htslib errmod.c: "then sample 255 bases": ks_shuffle(uint16_t, n, bases); n = 255 */
        if (deep_rule == 0) shuffle_u16(n, bases);
        n = 255;                            /* rule 1: the first 255 in pileup order */
    }
    sort_u16(n, bases);
    memset(w, 0, sizeof(w));
    memset(fsum, 0, sizeof(fsum)); memset(bsum, 0, sizeof(bsum)); memset(c, 0, sizeof(c));
    for (j = n - 1; j >= 0; --j) {          /* from the best quality downwards */
        uint16_t b = bases[j];
        int qual = b>>5 < 4? 4 : b>>5;
        if (qual > 63) qual = 63;
        int basestrand = b & 0x1f;
        int base = b & 0xf;
        fsum[base] += em->fk[w[basestrand]];
        bsum[base] += em->fk[w[basestrand]] * em->beta[qual<<16 | n<<8 | c[base]];
        ++c[base];
        ++w[basestrand];
    }
    for (j = 0; j < m; ++j) {
        float tmp1, tmp3;
        int tmp2;
        /* homozygous */
        for (k = 0, tmp1 = tmp3 = 0.0, tmp2 = 0; k < m; ++k) {
            if (k == j) continue;
            tmp1 += bsum[k]; tmp2 += c[k]; tmp3 += fsum[k];
        }
        if (tmp2) q[j*m+j] = tmp1;
        /* heterozygous */
        for (k = j + 1; k < m; ++k) {
            int cjk = c[j] + c[k];
            for (i = 0, tmp2 = 0, tmp1 = tmp3 = 0.0; i < m; ++i) {
                if (i == j || i == k) continue;
                tmp1 += bsum[i]; tmp2 += c[i]; tmp3 += fsum[i];
            }
            if (tmp2) q[j*m+k] = q[k*m+j] = -4.343 * em->lhet[cjk<<8|c[k]] + tmp1;
            else      q[j*m+k] = q[k*m+j] = -4.343 * em->lhet[cjk<<8|c[k]];
        }
        for (k = 0; k < m; ++k) if (q[j*m+k] < 0.0) q[j*m+k] = 0.0;
    }
    return 0;
}

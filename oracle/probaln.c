/*  probaln.c -- ORACLE (test infrastructure only): restatement of htslib's banded glocal pair-HMM
 *  probaln_glocal() (probaln.c) and of the BAQ wrapper sam_prob_realn() (realn.c).
 *
 *  htslib is an external, un-vendored dependency of the reference; the algorithm is restated
 *  from its published form (Li 2011, "Improving SNP discovery by base alignment quality", and the
 *  kpa_glocal implementation notes) and anchored on the reference's call sites:
 *     bam2bcf_indel.c:346,352   probaln_glocal(ref2+tbeg-left, tend-tbeg+|type|, query, qend-qbeg, qq, &apf, 0, 0)
 *     mpileup.c:234             sam_prob_realn(b, ref, ref_len, 3)   (apply + extend)
 *  Parity status: pinned only through the BAQ-dependent goldens (tests/test_oracle_golden_baq.py).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>
#include "bcforacle.h"

#define EI .25
#define EM .33333333333

#define set_u(u, b, i, k) { int x_=(i)-(b); x_=x_>0?x_:0; (u)=((k)-x_+1)*3; }

/* ref/query: 0..3 = ACGT, >3 = N.  iqual: phred per query base (NULL -> 30).  state/q may both be NULL
 * (forward score only, as bam2bcf_indel.c uses it).  Returns the phred-scaled alignment score. */
int orc_probaln_glocal(const uint8_t *ref, int l_ref, const uint8_t *query, int l_query, const uint8_t *iqual,
                       double d, double e_, int cbw, int *state, uint8_t *q)
{
    double **f, **b = NULL, *s, m[9], sI, sM, bI, bM;
    float *qual;
    int bw, bw2, i, k, is_backward, Pr;
    if (l_ref <= 0 || l_query <= 0) return 0;

    is_backward = state && q ? 1 : 0;
    bw = l_ref > l_query ? l_ref : l_query;
    if (bw > cbw) bw = cbw;
    if (bw < abs(l_ref - l_query)) bw = abs(l_ref - l_query);
    bw2 = bw * 2 + 1;
    f = (double**) calloc(l_query + 1, sizeof(double*));
    if (is_backward) b = (double**) calloc(l_query + 1, sizeof(double*));
    for (i = 0; i <= l_query; ++i) {
        f[i] = (double*) calloc(bw2 * 3 + 6, sizeof(double));
        if (is_backward) b[i] = (double*) calloc(bw2 * 3 + 6, sizeof(double));
    }
    s = (double*) calloc(l_query + 2, sizeof(double));
    qual = (float*) calloc(l_query, sizeof(float));
    for (i = 0; i < l_query; ++i)
        qual[i] = pow(10, -(iqual ? iqual[i] : 30) / 10.);
    sM = sI = 1. / (2 * l_query + 2);
    m[0*3+0] = (1 - d - d) * (1 - sM); m[0*3+1] = m[0*3+2] = d * (1 - sM);
    m[1*3+0] = (1 - e_) * (1 - sI); m[1*3+1] = e_ * (1 - sI); m[1*3+2] = 0.;
    m[2*3+0] = 1 - e_; m[2*3+1] = 0.; m[2*3+2] = e_;
    bM = (1 - d) / l_ref; bI = d / l_ref;
    /*** forward ***/
    set_u(k, bw, 0, 0);
    f[0][k] = s[0] = 1.;
    {   /* f[1] */
        double *fi = f[1], sum;
        int beg = 1, end = l_ref < bw + 1 ? l_ref : bw + 1, _beg, _end;
        for (k = beg, sum = 0.; k <= end; ++k) {
            int u;
            double e = (ref[k - 1] > 3 || query[0] > 3) ? 1. : ref[k - 1] == query[0] ? 1. - qual[0] : qual[0] * EM;
            set_u(u, bw, 1, k);
            fi[u+0] = e * bM; fi[u+1] = EI * bI;
            sum += fi[u] + fi[u+1];
        }
        s[1] = sum;
        set_u(_beg, bw, 1, beg); set_u(_end, bw, 1, end); _end += 2;
        for (k = _beg; k <= _end; ++k) fi[k] /= sum;
    }
    for (i = 2; i <= l_query; ++i) {
        double *fi = f[i], *fi1 = f[i-1], sum, qli = qual[i-1];
        int beg = 1, end = l_ref, x, _beg, _end;
        uint8_t qyi = query[i - 1];
        x = i - bw; beg = beg > x ? beg : x;
        x = i + bw; end = end < x ? end : x;
        for (k = beg, sum = 0.; k <= end; ++k) {
            int u, v11, v01, v10;
            double e;
            e = (ref[k - 1] > 3 || qyi > 3) ? 1. : ref[k - 1] == qyi ? 1. - qli : qli * EM;
            set_u(u, bw, i, k); set_u(v11, bw, i-1, k-1); set_u(v10, bw, i-1, k); set_u(v01, bw, i, k-1);
            fi[u+0] = e * (m[0] * fi1[v11+0] + m[3] * fi1[v11+1] + m[6] * fi1[v11+2]);
            fi[u+1] = EI * (m[1] * fi1[v10+0] + m[4] * fi1[v10+1]);
            fi[u+2] = m[2] * fi[v01+0] + m[8] * fi[v01+2];
            sum += fi[u] + fi[u+1] + fi[u+2];
        }
        s[i] = sum;
        set_u(_beg, bw, i, beg); set_u(_end, bw, i, end); _end += 2;
        for (k = _beg, sum = 1./sum; k <= _end; ++k) fi[k] *= sum;
    }
    {   /* f[l_query+1] */
        double sum;
        for (k = 1, sum = 0.; k <= l_ref; ++k) {
            int u;
            set_u(u, bw, l_query, k);
            if (u < 3 || u >= bw2*3+3) continue;
            sum += f[l_query][u+0] * sM + f[l_query][u+1] * sI;
        }
        s[l_query+1] = sum;
    }
    {   /* likelihood */
        double p = 1., Pr1 = 0.;
        for (i = 0; i <= l_query + 1; ++i) {
            p *= s[i];
            if (p < 1e-100) Pr1 += -4.343 * log(p), p = 1.;
        }
        Pr1 += -4.343 * log(p * l_ref * l_query);
        Pr = (int)(Pr1 + .499);
    }
    if (is_backward) {
        double pb;
        /*** backward ***/
        for (k = 1; k <= l_ref; ++k) {
            int u;
            double *bi = b[l_query];
            set_u(u, bw, l_query, k);
            if (u < 3 || u >= bw2*3+3) continue;
            bi[u+0] = sM / s[l_query] / s[l_query+1]; bi[u+1] = sI / s[l_query] / s[l_query+1];
        }
        for (i = l_query - 1; i >= 1; --i) {
            int beg = 1, end = l_ref, x, _beg, _end;
            double *bi = b[i], *bi1 = b[i+1], y = (i > 1), qli1 = qual[i];
            uint8_t qyi1 = query[i];
            x = i - bw; beg = beg > x ? beg : x;
            x = i + bw; end = end < x ? end : x;
            for (k = end; k >= beg; --k) {
                int u, v11, v01, v10;
                double e;
                set_u(u, bw, i, k); set_u(v11, bw, i+1, k+1); set_u(v10, bw, i+1, k); set_u(v01, bw, i, k+1);
                e = (k >= l_ref ? 0 : (ref[k] > 3 || qyi1 > 3) ? 1. : ref[k] == qyi1 ? 1. - qli1 : qli1 * EM) * bi1[v11];
                bi[u+0] = e * m[0] + EI * m[1] * bi1[v10+1] + m[2] * bi[v01+2];
                bi[u+1] = e * m[3] + EI * m[4] * bi1[v10+1];
                bi[u+2] = (e * m[6] + m[8] * bi[v01+2]) * y;
            }
            set_u(_beg, bw, i, beg); set_u(_end, bw, i, end); _end += 2;
            for (k = _beg, y = 1./s[i]; k <= _end; ++k) bi[k] *= y;
        }
        {
            int beg = 1, end = l_ref < bw + 1 ? l_ref : bw + 1;
            double sum = 0.;
            for (k = end; k >= beg; --k) {
                int u;
                double e = (ref[k - 1] > 3 || query[0] > 3) ? 1. : ref[k - 1] == query[0] ? 1. - qual[0] : qual[0] * EM;
                set_u(u, bw, 1, k);
                if (u < 3 || u >= bw2*3+3) continue;
                sum += e * b[1][u+0] * bM + EI * b[1][u+1] * bI;
            }
            set_u(k, bw, 0, 0);
            pb = b[0][k] = sum / s[0];
            (void)pb;
        }
        /*** MAP ***/
        for (i = 1; i <= l_query; ++i) {
            double sum = 0., *fi = f[i], *bi = b[i], max = 0.;
            int beg = 1, end = l_ref, x, max_k = -1;
            x = i - bw; beg = beg > x ? beg : x;
            x = i + bw; end = end < x ? end : x;
            for (k = beg; k <= end; ++k) {
                int u;
                double z;
                set_u(u, bw, i, k);
                z = fi[u+0] * bi[u+0]; if (z > max) max = z, max_k = (k-1)<<2 | 0; sum += z;
                z = fi[u+1] * bi[u+1]; if (z > max) max = z, max_k = (k-1)<<2 | 1; sum += z;
            }
            max /= sum; sum *= s[i];
            if (state) state[i-1] = max_k;
            if (q) k = (int)(-4.343 * log(1. - max) + .499), q[i-1] = k > 100 ? 99 : k;
        }
    }
    for (i = 0; i <= l_query; ++i) { free(f[i]); if (b) free(b[i]); }
    free(f); free(b); free(s); free(qual);
    return Pr;
}

/* sam_prob_realn (BAQ) on plain arrays.  seq4: 2-bit/4=N codes of the read, qual: modified in place when
 * apply (flag&1); cigar in BAM encoding; ref: ASCII reference of the contig; zq (l_qseq bytes) receives the
 * "ZQ" tag bytes (64-based offsets) when non-NULL.  Returns 0 when applied, <0 when the read is left alone. */
int orc_sam_prob_realn(int pos, int l_qseq, const uint8_t *seq4, uint8_t *qual, const uint32_t *cigar, int n_cigar,
                       const char *ref, int ref_len, int flag, uint8_t *zq)
{
    static const uint8_t nt4[256] = {
#define N4 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4
        N4,N4,N4,N4,
        4,0,4,1,4,4,4,2,4,4,4,4,4,4,4,4, 4,4,4,4,3,4,4,4,4,4,4,4,4,4,4,4,
        4,0,4,1,4,4,4,2,4,4,4,4,4,4,4,4, 4,4,4,4,3,4,4,4,4,4,4,4,4,4,4,4,
        N4,N4,N4,N4,N4,N4,N4,N4
#undef N4
    };
    int k, i, bw, x, y, yb, ye, xb, xe, apply_baq = flag & 1, extend_baq = flag >> 1 & 1;
    if (l_qseq == 0 || qual[0] == 0xff) return -1;
    x = pos; y = 0; yb = ye = xb = xe = -1;
    for (k = 0; k < n_cigar; ++k) {
        int op = cigar[k] & 0xf, l = cigar[k] >> 4;
        if (op == 0 || op == 7 || op == 8) {
            if (yb < 0) yb = y;
            if (xb < 0) xb = x;
            ye = y + l; xe = x + l;
            x += l; y += l;
        } else if (op == 4 || op == 1) y += l;
        else if (op == 2) x += l;
        else if (op == 3) return -1;
    }
    if (xb == -1) return -1;
    bw = 7;
    if (abs((xe - xb) - (ye - yb)) > bw) bw = abs((xe - xb) - (ye - yb)) + 3;
    xb -= yb + bw/2; if (xb < 0) xb = 0;
    xe += l_qseq - ye + bw/2;
    if (xe - xb - l_qseq > bw)
        xb += (xe - xb - l_qseq - bw) / 2, xe -= (xe - xb - l_qseq - bw) / 2;   /* second term sees the new xb */
    {
        int lref = xe > xb ? xe - xb : 1;
        uint8_t *tref = (uint8_t*) calloc(lref + 1, 1);
        uint8_t *q = (uint8_t*) calloc(l_qseq, 1);
        uint8_t *bq = (uint8_t*) malloc(l_qseq);
        int *state = (int*) calloc(l_qseq, sizeof(int));
        memcpy(bq, qual, l_qseq);
        for (k = xb; k < xe && k < ref_len && ref[k]; ++k) tref[k - xb] = nt4[(uint8_t) ref[k]];
        xe = k;
        orc_probaln_glocal(tref, xe - xb, seq4, l_qseq, qual, 0.001, 0.1, bw, state, q);
        if (!extend_baq) {
            for (k = 0, x = pos, y = 0; k < n_cigar; ++k) {
                int op = cigar[k] & 0xf, l = cigar[k] >> 4;
                if (op == 0 || op == 7 || op == 8) {
                    for (i = y; i < y + l; ++i) {
                        if ((state[i]&3) != 0 || state[i]>>2 != x - xb + (i - y)) bq[i] = 0;
                        else bq[i] = bq[i] < q[i] ? bq[i] : q[i];
                    }
                    x += l; y += l;
                } else if (op == 4 || op == 1) y += l;
                else if (op == 2) x += l;
            }
            for (i = 0; i < l_qseq; ++i) bq[i] = qual[i] - bq[i] + 64;
        } else {
            uint8_t *left = (uint8_t*) calloc(l_qseq, 1), *rght = (uint8_t*) calloc(l_qseq, 1);
            for (k = 0, x = pos, y = 0; k < n_cigar; ++k) {
                int op = cigar[k] & 0xf, l = cigar[k] >> 4;
                if (op == 0 || op == 7 || op == 8) {
                    for (i = y; i < y + l; ++i)
                        bq[i] = ((state[i]&3) != 0 || state[i]>>2 != x - xb + (i - y)) ? 0 : q[i];
                    for (left[y] = bq[y], i = y + 1; i < y + l; ++i)
                        left[i] = bq[i] > left[i-1] ? bq[i] : left[i-1];
                    for (rght[y+l-1] = bq[y+l-1], i = y + l - 2; i >= y; --i)
                        rght[i] = bq[i] > rght[i+1] ? bq[i] : rght[i+1];
                    for (i = y; i < y + l; ++i)
                        bq[i] = left[i] < rght[i] ? left[i] : rght[i];
                    x += l; y += l;
                } else if (op == 4 || op == 1) y += l;
                else if (op == 2) x += l;
            }
            for (i = 0; i < l_qseq; ++i) bq[i] = 64 + (qual[i] <= bq[i] ? 0 : qual[i] - bq[i]);
            free(left); free(rght);
        }
        if (apply_baq)
            for (i = 0; i < l_qseq; ++i) qual[i] -= bq[i] - 64;
        if (zq) memcpy(zq, bq, l_qseq);
        free(tref); free(q); free(bq); free(state);
    }
    return 0;
}

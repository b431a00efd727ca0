/*  indel_oracle.c -- ORACLE (test infrastructure only): CPU restatement of bcf_call_gap_prep()
 *  (bam2bcf_indel.c:99-470) with its helpers tpos2qpos (:40-66), est_seqQ (:69-75), est_indelreg (:77-88),
 *  over a flat read model instead of bam_pileup1_t / bam1_t.
 *
 *  Reads:   r_pos[], r_lq[], r_flag[], r_ncig[], r_cig_off[] -> cig[] (BAM encoding), r_seq_off[] -> seq16[] (one
 *           4-bit nt16 code per byte), qual[] (same offsets; the qualities the pileup sees, i.e. after BAQ and the
 *           mate-overlap tweak), zq[] (same offsets; "ZQ" bytes) with r_has_zq[].
 *  Pileup:  for sample s the entries smpl_off[s]..smpl_off[s+1]-1 of p_read[] / p_qpos[] / p_indel[].
 *  Output:  p_aux[] (= p->aux after the call), indel_types[4], inscns[4*maxins], maxins, indelreg, max_support,
 *           max_frac; return value as the reference (0 or -1).
 */
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <assert.h>
#include "bcforacle.h"

#define MINUS_CONST 0x10000000
#define INDEL_WINDOW_SIZE 50
#define B2B_INDEL_NULL 10000

static const uint8_t nt16_table[256] = {
    15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15,
    15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15,
    15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15,
     1, 2, 4, 8, 15,15,15,15, 15,15,15,15, 15, 0 /*=*/,15,15,
    15, 1,14, 2, 13,15,15, 4, 11,15,15,12, 15, 3,15,15,
    15,15, 5, 6,  8,15, 7, 9, 15,10,15,15, 15,15,15,15,
    15, 1,14, 2, 13,15,15, 4, 11,15,15,12, 15, 3,15,15,
    15,15, 5, 6,  8,15, 7, 9, 15,10,15,15, 15,15,15,15,
    15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15,
    15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15,
    15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15,
    15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15
};
static const int nt16_int[16] = { 4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4 };

static int cmp_u32(const void *a, const void *b) { uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b; return x < y ? -1 : x > y; }

static int tpos2qpos(int cpos, int n_cigar, const uint32_t *cigar, int32_t tpos, int is_left, int32_t *_tpos)
{
    int k, x = cpos, y = 0, last_y = 0;
    *_tpos = cpos;
    for (k = 0; k < n_cigar; ++k) {
        int op = cigar[k] & 0xf;
        int l = cigar[k] >> 4;
        if (op == 0 || op == 7 || op == 8) {
            if (cpos > tpos) return y;
            if (x + l > tpos) { *_tpos = tpos; return y + (tpos - x); }
            x += l; y += l;
            last_y = y;
        } else if (op == 1 || op == 4) y += l;
        else if (op == 2 || op == 3) {
            if (x + l > tpos) { *_tpos = is_left ? x : x + l; return y; }
            x += l;
        }
    }
    *_tpos = x;
    return last_y;
}

static inline int est_seqQ(int openQ, int extQ, int tandemQ, int l, int l_run)
{
    int q, qh;
    q = openQ + extQ * (abs(l) - 1);
    qh = l_run >= 3 ? (int)(tandemQ * (double)abs(l) / l_run + .499) : 1000;
    return q < qh ? q : qh;
}

static inline int est_indelreg(int pos, const char *ref, int l, char *ins4)
{
    int i, j, max = 0, max_i = pos, score = 0;
    l = abs(l);
    for (i = pos + 1, j = 0; ref[i]; ++i, ++j) {
        if (ins4) score += (toupper(ref[i]) != "ACGTN"[(int)ins4[j%l]]) ? -10 : 1;
        else score += (toupper(ref[i]) != toupper(ref[pos+1+j%l])) ? -10 : 1;
        if (score < 0) break;
        if (max < score) max = score, max_i = i;
    }
    return max_i - pos;
}

static int cigar2qlen(int n, const uint32_t *cig)
{
    int k, l = 0;
    for (k = 0; k < n; ++k) { int op = cig[k] & 0xf; if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) l += cig[k] >> 4; }
    return l;
}

int orc_gap_prep(int n, const int *smpl_off, const int *p_read, const int *p_qpos, const int *p_indel,
                 const int *r_pos, const int *r_lq, const int *r_flag, const int *r_ncig, const int *r_cig_off,
                 const uint32_t *cig, const int *r_seq_off, const uint8_t *seq16, const uint8_t *qualp,
                 const uint8_t *zqp, const uint8_t *r_has_zq,
                 int pos, const char *ref, int openQ, int extQ, int tandemQ, int min_support, double min_frac,
                 int per_sample_flt,
                 uint32_t *p_aux, int *indel_types, char *inscns_out, int inscns_cap, int *maxins_out, int *indelreg_out,
                 int *max_support_out, float *max_frac_out)
{
    int i, s, j, k, t, n_types, *types, max_rd_len, left, right, max_ins, *score1, *score2, max_ref2;
    int N, K, l_run, ref_type, n_alt;
    char *inscns = 0, *ref2, *query, **ref_sample;
    uint32_t max_support = 0; float max_frac = 0;
    if (ref == 0) return -1;
    #define NPLP(s) (smpl_off[(s)+1] - smpl_off[s])
    #define PE(s,i) (smpl_off[s] + (i))

    for (s = N = 0; s < n; ++s) {
        for (i = 0; i < NPLP(s); ++i)
            if (p_indel[PE(s,i)] != 0) break;
        if (i < NPLP(s)) break;
    }
    if (s == n) return -1;
    for (s = N = 0; s < n; ++s) N += NPLP(s);
    {
        int m, n_alt = 0, n_tot = 0, indel_support_ok = 0;
        uint32_t *aux = (uint32_t*) calloc(N + 1, 4);
        m = max_rd_len = 0;
        aux[m++] = MINUS_CONST;
        for (s = 0; s < n; ++s) {
            int na = 0, nt = 0;
            for (i = 0; i < NPLP(s); ++i) {
                int e = PE(s,i), r = p_read[e];
                ++nt;
                if (p_indel[e] != 0) { ++na; aux[m++] = MINUS_CONST + p_indel[e]; }
                j = cigar2qlen(r_ncig[r], cig + r_cig_off[r]);
                if (j > max_rd_len) max_rd_len = j;
            }
            double frac = (double)na/nt;
            if (!indel_support_ok && na >= min_support && frac >= min_frac) indel_support_ok = 1;
            if (na > (int)max_support && frac > 0) max_support = na, max_frac = frac;
            n_alt += na;
            n_tot += nt;
        }
        *max_support_out = max_support; *max_frac_out = max_frac;
        int nN = 0; for (i = pos; i-pos < max_rd_len && ref[i]; i++) if (ref[i] == 'N') nN++;
        if (nN*2 > (i-pos)) { free(aux); return -1; }
        qsort(aux, m, 4, cmp_u32);
        for (i = 1, n_types = 1; i < m; ++i)
            if (aux[i] != aux[i-1]) ++n_types;
        if (!per_sample_flt)
            indel_support_ok = ((double)n_alt / n_tot < min_frac || n_alt < min_support) ? 0 : 1;
        if (n_types == 1 || !indel_support_ok) { free(aux); return -1; }
        if (n_types >= 64) { free(aux); return -1; }
        types = (int*) calloc(n_types, sizeof(int));
        t = 0;
        types[t++] = aux[0] - MINUS_CONST;
        for (i = 1; i < m; ++i)
            if (aux[i] != aux[i-1]) types[t++] = aux[i] - MINUS_CONST;
        free(aux);
        for (t = 0; t < n_types; ++t)
            if (types[t] == 0) break;
        ref_type = t;
    }
    {
        left = pos > INDEL_WINDOW_SIZE ? pos - INDEL_WINDOW_SIZE : 0;
        right = pos + INDEL_WINDOW_SIZE;
        if (types[0] < 0) right -= types[0];
        for (i = pos; i < right; ++i)
            if (ref[i] == 0) break;
        right = i;
    }
    {
        int L = right - left + 1, max_i, max2_i;
        uint32_t *cns, max, max2;
        char *ref0, *r;
        ref_sample = (char**) calloc(n, sizeof(char*));
        cns = (uint32_t*) calloc(L, 4);
        ref0 = (char*) calloc(L, 1);
        for (i = 0; i < right - left; ++i) ref0[i] = nt16_table[(uint8_t)ref[i+left]];
        for (s = 0; s < n; ++s) {
            r = ref_sample[s] = (char*) calloc(L, 1);
            memset(cns, 0, sizeof(int) * L);
            for (i = 0; i < NPLP(s); ++i) {
                int rd = p_read[PE(s,i)];
                const uint32_t *cigar = cig + r_cig_off[rd];
                const uint8_t *seq = seq16 + r_seq_off[rd];
                int x = r_pos[rd], y = 0;
                for (k = 0; k < r_ncig[rd]; ++k) {
                    int op = cigar[k]&0xf;
                    int j, l = cigar[k]>>4;
                    if (op == 0 || op == 7 || op == 8) {
                        for (j = 0; j < l; ++j)
                            if (x + j >= left && x + j < right)
                                cns[x+j-left] += (seq[y+j] == ref0[x+j-left]) ? 1 : 0x10000;
                        x += l; y += l;
                    } else if (op == 2 || op == 3) x += l;
                    else if (op == 1 || op == 4) y += l;
                }
            }
            for (i = 0; i < right - left; ++i) r[i] = ref0[i];
            max = max2 = 0; max_i = max2_i = -1;
            for (i = 0; i < right - left; ++i) {
                if (cns[i]>>16 >= max>>16) max2 = max, max2_i = max_i, max = cns[i], max_i = i;
                else if (cns[i]>>16 >= max2>>16) max2 = cns[i], max2_i = i;
            }
            if ((double)(max&0xffff) / ((max&0xffff) + (max>>16)) >= 0.7) max_i = -1;
            if ((double)(max2&0xffff) / ((max2&0xffff) + (max2>>16)) >= 0.7) max2_i = -1;
            if (max_i >= 0) r[max_i] = 15;
            if (max2_i >= 0) r[max2_i] = 15;
        }
        free(ref0); free(cns);
    }
    {
        int c = nt16_table[(uint8_t)ref[pos + 1]];
        if (c == 15) l_run = 1;
        else {
            for (i = pos + 2; ref[i]; ++i)
                if (nt16_table[(uint8_t)ref[i]] != c) break;
            l_run = i;
            for (i = pos; i >= 0; --i)
                if (nt16_table[(uint8_t)ref[i]] != c) break;
            l_run -= i + 1;
        }
    }
    max_ins = types[n_types - 1];
    if (max_ins > 0) {
        int *inscns_aux = (int*) calloc(5 * n_types * max_ins, sizeof(int));
        for (t = 0; t < n_types; ++t) {
            if (types[t] > 0) {
                for (s = 0; s < n; ++s) {
                    for (i = 0; i < NPLP(s); ++i) {
                        int e = PE(s,i), rd = p_read[e];
                        if (p_indel[e] == types[t]) {
                            const uint8_t *seq = seq16 + r_seq_off[rd];
                            for (k = 1; k <= p_indel[e]; ++k) {
                                int c = nt16_int[seq[p_qpos[e] + k]];
                                assert(c < 5);
                                ++inscns_aux[(t*max_ins+(k-1))*5 + c];
                            }
                        }
                    }
                }
            }
        }
        inscns = (char*) calloc(n_types * max_ins, 1);
        for (t = 0; t < n_types; ++t) {
            for (j = 0; j < types[t]; ++j) {
                int max = 0, max_k = -1, *ia = &inscns_aux[(t*max_ins+j)*5];
                for (k = 0; k < 5; ++k)
                    if (ia[k] > max) max = ia[k], max_k = k;
                inscns[t*max_ins + j] = max ? max_k : 4;
                if (max_k == 4) { types[t] = 0; break; }
            }
        }
        free(inscns_aux);
    }
    max_ref2 = right - left + 2 + 2 * (max_ins > -types[0] ? max_ins : -types[0]);
    ref2  = (char*) calloc(max_ref2, 1);
    query = (char*) calloc(right - left + max_rd_len + max_ins + 2, 1);
    score1 = (int*) calloc(N * n_types, sizeof(int));
    score2 = (int*) calloc(N * n_types, sizeof(int));
    int indelreg = 0;
    for (t = 0; t < n_types; ++t) {
        int l, ir;
        const int bw = abs(types[t]) + 3;
        if (types[t] == 0) ir = 0;
        else if (types[t] > 0) ir = est_indelreg(pos, ref, types[t], &inscns[t*max_ins]);
        else ir = est_indelreg(pos, ref, -types[t], 0);
        if (ir > indelreg) indelreg = ir;
        for (s = K = 0; s < n; ++s) {
            for (k = 0, j = left; j <= pos; ++j)
                ref2[k++] = nt16_int[(int)ref_sample[s][j-left]];
            if (types[t] <= 0) j += -types[t];
            else for (l = 0; l < types[t]; ++l)
                     ref2[k++] = inscns[t*max_ins + l];
            for (; j < right && ref[j]; ++j)
                ref2[k++] = nt16_int[(int)ref_sample[s][j-left]];
            for (; k < max_ref2; ++k) ref2[k] = 4;
            if (j < right) right = j;
            for (i = 0; i < NPLP(s); ++i, ++K) {
                int e = PE(s,i), rd = p_read[e];
                int qbeg, qend, tbeg, tend, sc, kk;
                const uint8_t *seq = seq16 + r_seq_off[rd];
                const uint32_t *cigar = cig + r_cig_off[rd];
                if (r_flag[rd] & 4) continue;
                for (kk = 0; kk < r_ncig[rd]; ++kk)
                    if ((cigar[kk] & 0xf) == 3) break;
                if (kk < r_ncig[rd]) continue;
                qbeg = tpos2qpos(r_pos[rd], r_ncig[rd], cigar, left,  0, &tbeg);
                qend = tpos2qpos(r_pos[rd], r_ncig[rd], cigar, right, 1, &tend);
                if (types[t] < 0) {
                    int l = -types[t];
                    tbeg = tbeg - l > left ? tbeg - l : left;
                }
                for (l = qbeg; l < qend; ++l)
                    query[l - qbeg] = nt16_int[seq[l]];
                {
                    const uint8_t *qual = qualp + r_seq_off[rd], *bq = r_has_zq && r_has_zq[rd] ? zqp + r_seq_off[rd] : NULL;
                    uint8_t *qq = (uint8_t*) calloc(qend - qbeg + 1, 1);
                    for (l = qbeg; l < qend; ++l) {
                        qq[l - qbeg] = bq ? qual[l] + (bq[l] - 64) : qual[l];
                        if (qq[l - qbeg] > 30) qq[l - qbeg] = 30;
                        if (qq[l - qbeg] < 7) qq[l - qbeg] = 7;
                    }
                    sc = orc_probaln_glocal((uint8_t*)ref2 + tbeg - left, tend - tbeg + abs(types[t]),
                                            (uint8_t*)query, qend - qbeg, qq, 1e-4, 1e-2, bw, 0, 0);
                    l = (int)(100. * sc / (qend - qbeg) + .499);
                    if (l > 255) l = 255;
                    score1[K*n_types + t] = score2[K*n_types + t] = sc<<8 | l;
                    if (sc > 5) {
                        sc = orc_probaln_glocal((uint8_t*)ref2 + tbeg - left, tend - tbeg + abs(types[t]),
                                                (uint8_t*)query, qend - qbeg, qq, 1e-6, 1e-3, bw, 0, 0);
                        l = (int)(100. * sc / (qend - qbeg) + .499);
                        if (l > 255) l = 255;
                        score2[K*n_types + t] = sc<<8 | l;
                    }
                    free(qq);
                }
            }
        }
    }
    free(ref2); free(query);
    {
        int sc_a[16], sumq_a[16];
        int tmp, *sc = sc_a, *sumq = sumq_a;
        if (n_types > 16) {
            sc   = (int *)malloc(n_types * sizeof(int));
            sumq = (int *)malloc(n_types * sizeof(int));
        }
        memset(sumq, 0, n_types * sizeof(int));
        for (s = K = 0; s < n; ++s) {
            for (i = 0; i < NPLP(s); ++i, ++K) {
                int e = PE(s,i);
                int *sct = &score1[K*n_types], indelQ1, indelQ2, seqQ, indelQ;
                for (t = 0; t < n_types; ++t) sc[t] = sct[t]<<6 | t;
                for (t = 1; t < n_types; ++t)
                    for (j = t; j > 0 && sc[j] < sc[j-1]; --j)
                        tmp = sc[j], sc[j] = sc[j-1], sc[j-1] = tmp;
                if ((sc[0]&0x3f) == ref_type) {
                    indelQ1 = (sc[1]>>14) - (sc[0]>>14);
                    seqQ = est_seqQ(openQ, extQ, tandemQ, types[sc[1]&0x3f], l_run);
                } else {
                    for (t = 0; t < n_types; ++t)
                        if ((sc[t]&0x3f) == ref_type) break;
                    indelQ1 = (sc[t]>>14) - (sc[0]>>14);
                    seqQ = est_seqQ(openQ, extQ, tandemQ, types[sc[0]&0x3f], l_run);
                }
                tmp = sc[0]>>6 & 0xff;
                indelQ1 = tmp > 111 ? 0 : (int)((1. - tmp/111.) * indelQ1 + .499);
                sct = &score2[K*n_types];
                for (t = 0; t < n_types; ++t) sc[t] = sct[t]<<6 | t;
                for (t = 1; t < n_types; ++t)
                    for (j = t; j > 0 && sc[j] < sc[j-1]; --j)
                        tmp = sc[j], sc[j] = sc[j-1], sc[j-1] = tmp;
                if ((sc[0]&0x3f) == ref_type) {
                    indelQ2 = (sc[1]>>14) - (sc[0]>>14);
                } else {
                    for (t = 0; t < n_types; ++t)
                        if ((sc[t]&0x3f) == ref_type) break;
                    indelQ2 = (sc[t]>>14) - (sc[0]>>14);
                }
                tmp = sc[0]>>6 & 0xff;
                indelQ2 = tmp > 111 ? 0 : (int)((1. - tmp/111.) * indelQ2 + .499);
                indelQ = indelQ1 < indelQ2 ? indelQ1 : indelQ2;
                if (indelQ > 255) indelQ = 255;
                if (seqQ > 255) seqQ = 255;
                p_aux[e] = (sc[0]&0x3f)<<16 | seqQ<<8 | indelQ;
                sumq[sc[0]&0x3f] += indelQ < seqQ ? indelQ : seqQ;
            }
        }
        *maxins_out = max_ins;
        for (t = 0; t < n_types; ++t) sumq[t] = sumq[t]<<6 | t;
        for (t = 1; t < n_types; ++t)
            for (j = t; j > 0 && sumq[j] > sumq[j-1]; --j)
                tmp = sumq[j], sumq[j] = sumq[j-1], sumq[j-1] = tmp;
        for (t = 0; t < n_types; ++t)
            if ((sumq[t]&0x3f) == ref_type) break;
        if (t) {
            tmp = sumq[t];
            for (; t > 0; --t) sumq[t] = sumq[t-1];
            sumq[0] = tmp;
        }
        for (t = 0; t < 4; ++t) indel_types[t] = B2B_INDEL_NULL;
        for (t = 0; t < 4 && t < n_types; ++t) {
            indel_types[t] = types[sumq[t]&0x3f];
            if (max_ins > 0 && (t + 1) * max_ins <= inscns_cap)
                memcpy(&inscns_out[t * max_ins], &inscns[(sumq[t]&0x3f) * max_ins], max_ins);
        }
        for (s = n_alt = 0; s < n; ++s) {
            for (i = 0; i < NPLP(s); ++i) {
                int e = PE(s,i);
                int x = types[p_aux[e]>>16&0x3f];
                for (j = 0; j < 4; ++j)
                    if (x == indel_types[j]) break;
                p_aux[e] = j<<16 | (j == 4 ? 0 : (p_aux[e]&0xffff));
                if ((p_aux[e]>>16&0x3f) > 0) ++n_alt;
            }
        }
        if (sc   != sc_a)   free(sc);
        if (sumq != sumq_a) free(sumq);
    }
    *indelreg_out = indelreg;
    free(score1); free(score2);
    for (i = 0; i < n; ++i) free(ref_sample[i]);
    free(ref_sample);
    free(types); free(inscns);
    return n_alt > 0 ? 0 : -1;
}

/*  overlap.c -- ORACLE (test infrastructure only): the mate-overlap quality tweak of the pileup engine.
 *
 *  `bcftools mpileup` enables it with bam_mplp_init_overlaps() (mpileup.c:640); the arithmetic lives in htslib sam.c
 *  (overlap_push / tweak_overlap_quality), which is not part of /root/reference.  Restated from its
 *  published behaviour: for a read pair (a arrived first, b second) every reference position that both reads cover with
 *  an aligned (M/=/X) base is visited in order; equal bases: qual_a = min(200, qual_a + qual_b), qual_b = 0; different
 *  bases: the higher quality (a on ties) is scaled by 0.8 and truncated, the other set to 0.
 *  Parity pins: the reads of test/mpileup/mpileup.{1,2,4}.sam contain overlapping mates; the goldens
 *  test/mpileup/mpileup.{1,2,4,5}.out only come out right with this tweak (tests/test_oracle_golden_baq.py).
 */
#include <stdint.h>
#include "bcforacle.h"

int orc_overlap_tweak(const bcfgpu_reads *rd, int32_t n_pairs, const int32_t *pair_a, const int32_t *pair_b, uint8_t *qual)
{
    for (int p = 0; p < n_pairs; ++p) {
        const int ra = pair_a[p], rb = pair_b[p];
        const uint32_t *ca = rd->cig + rd->r_cig_off[ra], *cb = rd->cig + rd->r_cig_off[rb];
        const int na = rd->r_ncig[ra], nb = rd->r_ncig[rb];
        const uint8_t *sa = rd->seq16 + rd->r_seq_off[ra], *sb = rd->seq16 + rd->r_seq_off[rb];
        uint8_t *qa = qual + rd->r_seq_off[ra], *qb = qual + rd->r_seq_off[rb];
        /* current aligned block of each read: reference start x, query start y, length l (l == 0: fetch the next) */
        int ka = 0, kb = 0, xa = rd->r_pos[ra], ya = 0, la = 0, xb = rd->r_pos[rb], yb = 0, lb = 0;
        for (;;) {
            while (la == 0 && ka < na) {
                const int op = ca[ka] & 0xf, l = (int)(ca[ka] >> 4); ++ka;
                if (op == 0 || op == 7 || op == 8) la = l;
                else if (op == 2 || op == 3) xa += l;
                else if (op == 1 || op == 4) ya += l;
            }
            while (lb == 0 && kb < nb) {
                const int op = cb[kb] & 0xf, l = (int)(cb[kb] >> 4); ++kb;
                if (op == 0 || op == 7 || op == 8) lb = l;
                else if (op == 2 || op == 3) xb += l;
                else if (op == 1 || op == 4) yb += l;
            }
            if (la == 0 || lb == 0) break;
            /* bring the blocks to a common reference position */
            if (xa < xb) { int d = xb - xa; if (d > la) d = la; xa += d; ya += d; la -= d; continue; }
            if (xb < xa) { int d = xa - xb; if (d > lb) d = lb; xb += d; yb += d; lb -= d; continue; }
            int m = la < lb ? la : lb;
            for (int i = 0; i < m; ++i) {
                uint8_t *pa = qa + ya + i, *pb = qb + yb + i;
                if (sa[ya + i] == sb[yb + i]) {
                    const int q = (int)*pa + (int)*pb;
                    *pa = (uint8_t)(q > 200 ? 200 : q);
                    *pb = 0;
                } else if (*pa >= *pb) { *pa = (uint8_t)(0.8 * *pa); *pb = 0; }
                else { *pb = (uint8_t)(0.8 * *pb); *pa = 0; }
            }
            xa += m; ya += m; la -= m; xb += m; yb += m; lb -= m;
        }
    }
    return 0;
}

/*  ORACLE (test infrastructure only): sam_cap_mapq() of htslib sam.c, what `bcftools mpileup -C INT` applies to every read
 *  (mpileup.c:235-239).  htslib is an external, un-vendored dependency of the reference; restated from its
 *  published source.  PARITY UNPINNED: no golden of the reference's tests runs `mpileup -C`, so this restatement is checked
 *  only for internal consistency (tests/test_oracle_capmapq.py) and serves as the device kernel's differential partner.
 */
#include <math.h>
#include <stdint.h>

static int nt16_of(char c)
{
    switch (c) {
        case 'A': case 'a': return 1;  case 'C': case 'c': return 2;  case 'G': case 'g': return 4;  case 'T': case 't': return 8;
        case '=': return 0;
        case 'M': case 'm': return 3;  case 'R': case 'r': return 5;  case 'S': case 's': return 6;  case 'V': case 'v': return 7;
        case 'W': case 'w': return 9;  case 'Y': case 'y': return 10; case 'H': case 'h': return 11; case 'K': case 'k': return 12;
        case 'D': case 'd': return 13; case 'B': case 'b': return 14;
        case 'U': case 'u': return 8;  case '0': return 1; case '1': return 2; case '2': return 4; case '3': return 8;   /* the rest of htslib's seq_nt16_table */
        default: return 15;
    }
}

/* one read: pos, CIGAR (BAM encoding), one 4-bit code per byte in seq16, qualities; returns the cap or -1 */
int orc_cap_mapq(int pos, int n_cigar, const uint32_t *cigar, const uint8_t *seq16, const uint8_t *qual,
                 const char *ref, int ref_len, int thres)
{
    int i, y, mm, q, len, clip_l, clip_q;
    long x;
    double t;
    if (thres < 0) thres = 40;
    mm = q = len = clip_l = clip_q = 0;
    for (i = y = 0, x = pos; i < n_cigar; ++i) {
        int j, l = cigar[i] >> 4, op = cigar[i] & 0xf;
        if (op == 0 || op == 7 || op == 8) {
            for (j = 0; j < l; ++j) {
                int c1, c2, z = y + j;
                if (x + j >= ref_len || ref[x + j] == '\0') break;
                c1 = seq16[z] & 15; c2 = nt16_of(ref[x + j]);
                if (c2 != 15 && c1 != 15 && qual[z] >= 13) {
                    ++len;
                    if (c1 && c1 != c2 && qual[z] >= 13) { ++mm; q += qual[z] > 33 ? 33 : qual[z]; }
                }
            }
            if (j < l) break;
            x += l; y += l; len += l;
        } else if (op == 2) {
            for (j = 0; j < l; ++j) if (x + j >= ref_len || ref[x + j] == '\0') break;
            if (j < l) break;
            x += l;
        } else if (op == 4) {
            for (j = 0; j < l; ++j) clip_q += qual[y + j];
            clip_l += l; y += l;
        } else if (op == 5) { clip_q += 13 * l; clip_l += l; }
        else if (op == 1) y += l;
        else if (op == 3) x += l;
    }
    for (i = 0, t = 1; i < mm; ++i) t *= (double)len / (i + 1);
    t = q - 4.343 * log(t) + clip_q / 5.;
    if (t > thres) return -1;
    if (t < 0) t = 0;
    t = sqrt((thres - t) / thres) * thres;
    return (int)(t + .499);
}

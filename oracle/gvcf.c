/*  ORACLE (test infrastructure only): gvcf_write() of the reference (gvcf.c:88-226) as the sequential state machine it is,
 *  over records held in arrays instead of bcf1_t.  The product path never links this.
 *  Pinned by the reference's golden test/mpileup/mpileup.6.out (mpileup -a DP,DV --gvcf 0,2,5; test.pl:645-646) in
 *  tests/test_oracle_golden_gvcf.py.
 */
#include <string.h>
#include "bcforacle.h"

typedef struct {
    int prev_range, rid, start, end, min_dp, first_site, last_site, nb;
} gv_state;

/* the flush of gvcf.c:134-166: `next_pos`/`next_rid` describe the record that triggers it (next_rid = -2: none) */
static void gv_flush(gv_state *g, int next_rid, int next_pos, bcfgpu_gvcf_block *block)
{
    if (next_rid == g->rid && next_pos == g->end) g->end--;          /* gvcf.c:139 */
    g->end++;                                                        /* 0-based -> 1-based, gvcf.c:141 */
    bcfgpu_gvcf_block *b = &block[g->nb++];
    b->first_site = g->first_site; b->last_site = g->last_site;
    b->start_pos = g->start; b->end1 = g->end; b->min_dp = g->min_dp; b->range = g->prev_range;
    g->prev_range = 0; g->rid = -1;
}

int orc_gvcf_blocks(int n, int S, const int32_t *pos, const int32_t *rid, const uint8_t *brk, const uint8_t *is_ref,
                    const int32_t *dp, const int32_t *pl, const int32_t *dp_range, int n_range,
                    int32_t *blk, int32_t *min_dp_out, bcfgpu_gvcf_block *block, int32_t *dp_out, int32_t *pl_out)
{
    gv_state g; memset(&g, 0, sizeof g); g.rid = -1;
    for (int i = 0; i < n; ++i) {
        const int r = rid ? rid[i] : 0;
        if (brk && (brk[i] & 2)) { blk[i] = -1; min_dp_out[i] = 0; continue; }     /* no record for this site (a column without reads) */
        const int32_t *d = dp + (size_t)i * S, *p = pl + (size_t)i * 3 * S;
        int can = is_ref[i] ? 1 : 0, range = 0, min_dp = 0, needs_flush = can ? 0 : 1;
        if (can) {                                                   /* gvcf.c:106-128 */
            min_dp = d[0];
            for (int s = 1; s < S; ++s) if (min_dp > d[s]) min_dp = d[s];
            int k;
            for (k = 0; k < n_range; ++k) if (min_dp < dp_range[k]) break;
            range = k;
            if (!range) { needs_flush = 1; can = 0; }
        }
        min_dp_out[i] = min_dp;
        if (g.prev_range && g.prev_range != range) needs_flush = 1;
        if (g.rid != r || pos[i] > g.end + 1) needs_flush = 1;
        if (g.prev_range && needs_flush) gv_flush(&g, r, pos[i], block);
        blk[i] = -1;
        if (can) {
            int32_t *bd = dp_out + (size_t)g.nb * S, *bp = pl_out + (size_t)g.nb * 3 * S;
            if (!g.prev_range) {                                     /* gvcf.c:171-189 */
                memcpy(bd, d, (size_t)S * 4); memcpy(bp, p, (size_t)3 * S * 4);
                g.rid = r; g.start = pos[i]; g.min_dp = min_dp; g.first_site = i;
            } else {                                                 /* gvcf.c:190-213 */
                if (g.min_dp > min_dp) g.min_dp = min_dp;
                for (int s = 0; s < S; ++s) {
                    if (bd[s] > d[s]) bd[s] = d[s];
                    if (bp[S + s] > p[S + s]) { bp[S + s] = p[S + s]; bp[2 * S + s] = p[2 * S + s]; }
                    else if (bp[S + s] == p[S + s] && bp[2 * S + s] > p[2 * S + s]) bp[2 * S + s] = p[2 * S + s];
                }
            }
            g.prev_range = range; g.end = pos[i]; g.last_site = i;
            blk[i] = g.nb;
        }
        if (brk && (brk[i] & 1) && g.prev_range) gv_flush(&g, r, pos[i], block);   /* a record that cannot join, same position */
    }
    if (g.prev_range) gv_flush(&g, -2, 0, block);                    /* gvcf_write(.., NULL, 0) at the end, mpileup.c:303-307 */
    return g.nb;
}

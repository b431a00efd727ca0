/*  bcfgpu_host.c -- a plain-C host program over the C-ABI of include/bcfgpu.h.
 *
 *  It plays the part of mpileup_reg()'s column loop (mpileup.c:320-367) followed by the record loop of
 *  `bcftools call -m` (vcfcall.c:1089-1148) for one tile: pileup columns are packed with bcfgpu_pack_read() into the
 *  site x sample x read SoA, uploaded, run through bcfgpu_pipeline() (glfgen+errmod -> combine -> call -m with
 *  PL/QS kept in HBM), and the per-site call records are downloaded and printed one line per site, VCF-like:
 *
 *      site  REF  ALT-list  QUAL  AN  AC-list  DP  first-sample GT
 *
 *  The pileup itself is synthetic (a small deterministic generator stands in for the BAM readers, which are outside
 *  the path): usage  bcfgpu_host <n_sites> <n_smpl> <depth> <seed> [-v]
 *  The same generator is restated in tests/test_c_host.py, which checks the printed records against the oracle.
 *
 *  Build:  gcc -std=c99 -O2 -Iinclude host/bcfgpu_host.c -Lbcftools_amd -lbcfgpu -Wl,-rpath,$PWD/bcftools_amd
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "bcfgpu.h"

#define READ_LEN 100

/* xorshift64*: the only state is one 64-bit word, easy to restate elsewhere */
static uint64_t rng_state;
static uint32_t rnd32(void)
{
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (uint32_t)((rng_state * 2685821657736338717ULL) >> 32);
}
static uint32_t rnd_below(uint32_t n) { return (uint32_t)(((uint64_t)rnd32() * n) >> 32); }

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s: %s (%d)\n", #call, bcfgpu_last_error(), rc_); exit(1); } } while (0)

static void *dev_alloc(bcfgpu_ctx *ctx, size_t bytes)
{
    void *p = NULL;
    CHECK(bcfgpu_malloc(ctx, bytes ? bytes : 16, &p));
    CHECK(bcfgpu_memset(ctx, p, 0, bytes ? bytes : 16));
    return p;
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s n_sites n_smpl depth seed [-v]\n", argv[0]); return 2; }
    const int n_sites = atoi(argv[1]), n_smpl = atoi(argv[2]), depth = atoi(argv[3]);
    rng_state = strtoull(argv[4], NULL, 10) * 2 + 1;
    const int varonly = argc > 5 && !strcmp(argv[5], "-v");
    static const uint8_t bq_values[4] = { 11, 25, 37, 40 };
    static const char nt[] = "ACGTN";

    /* ---- the column loop: one pileup column per site, depth..2*depth-1 reads per sample ---- */
    const size_t ncell = (size_t)n_sites * n_smpl;
    const size_t max_reads = ncell * (size_t)(2 * depth);
    int8_t   *ref16 = malloc(n_sites);
    uint32_t *off   = malloc((ncell + 1) * sizeof *off);
    uint32_t *rd    = malloc((max_reads + 4) * sizeof *rd);
    uint8_t  *epos  = malloc(max_reads + 16);
    if (!ref16 || !off || !rd || !epos) { fprintf(stderr, "out of memory\n"); return 1; }
    size_t nr = 0;
    off[0] = 0;
    for (int k = 0; k < n_sites; k++) {
        const int ref2 = (int)rnd_below(4), alt2 = (ref2 + 1 + (int)rnd_below(3)) & 3;
        const int is_var = rnd_below(4) == 0;
        ref16[k] = (int8_t)(1 << ref2);
        for (int s = 0; s < n_smpl; s++) {
            const int nalt = is_var ? (int)rnd_below(3) : 0;          /* 0, 1 or 2 ALT copies in this sample */
            const int n = depth + (int)rnd_below((uint32_t)depth);
            for (int j = 0; j < n; j++) {
                const int bq = bq_values[rnd_below(4)];
                int base = (nalt == 2 || (nalt == 1 && (rnd32() & 1))) ? alt2 : ref2;
                if (rnd_below(1000) < (bq < 20 ? 80u : 3u)) base = (base + 1 + (int)rnd_below(3)) & 3;   /* sequencing error */
                const int mapq = rnd_below(10) ? 60 : (int)rnd_below(60);
                const int qpos = (int)rnd_below(READ_LEN);
                const uint32_t one_match = (uint32_t)READ_LEN << 4;      /* CIGAR 100M */
                bcfgpu_pack_read(1 << base, bq, mapq, (int)(rnd32() & 1), 0, 0, 0, qpos, READ_LEN, &one_match, 1, 1,
                                 &rd[nr], &epos[nr]);
                nr++;
            }
            off[(size_t)k * n_smpl + s + 1] = (uint32_t)nr;
        }
    }

    /* ---- context, upload ---- */
    bcfgpu_cfg cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = 0; cfg.n_smpl = n_smpl; cfg.max_sites = n_sites; cfg.max_reads = nr;
    cfg.min_baseQ = 13; cfg.capQ = 60; cfg.errmod_theta = 0.; cfg.fmt_flag = BCFGPU_INFO_VDB | BCFGPU_INFO_RPB;
    cfg.call_theta = 1.1e-3; cfg.call_flag = varonly ? BCFGPU_CALL_VARONLY : 0; cfg.n_grp = 1; cfg.ploidy_max = 2;
    bcfgpu_ctx *ctx = NULL;
    CHECK(bcfgpu_create(&cfg, &ctx));

    int8_t *d_ref = dev_alloc(ctx, n_sites);
    uint32_t *d_off = dev_alloc(ctx, (ncell + 1) * 4), *d_rd = dev_alloc(ctx, (nr + 4) * 4);
    uint8_t *d_ep = dev_alloc(ctx, nr + 16);
    CHECK(bcfgpu_memcpy_h2d(ctx, d_ref, ref16, n_sites));
    CHECK(bcfgpu_memcpy_h2d(ctx, d_off, off, (ncell + 1) * 4));
    CHECK(bcfgpu_memcpy_h2d(ctx, d_rd, rd, nr * 4));
    CHECK(bcfgpu_memcpy_h2d(ctx, d_ep, epos, nr));

    bcfgpu_mplp_out mo;
    memset(&mo, 0, sizeof mo);
    mo.site = dev_alloc(ctx, (size_t)n_sites * sizeof(bcfgpu_site));
    mo.pl   = dev_alloc(ctx, ncell * BCFGPU_MAX_PL);
    mo.dp4  = dev_alloc(ctx, ncell * 4 * sizeof(uint16_t));
    bcfgpu_call_out co;
    memset(&co, 0, sizeof co);
    co.site = dev_alloc(ctx, (size_t)n_sites * sizeof(bcfgpu_call_site));
    co.gt   = dev_alloc(ctx, ncell * 2);
    co.pl   = dev_alloc(ctx, ncell * BCFGPU_MAX_PL * sizeof(int32_t));

    /* ---- the hot path ---- */
    bcfgpu_tile tile;
    memset(&tile, 0, sizeof tile);
    tile.n_sites = n_sites; tile.is_indel = 0; tile.n_reads = nr;
    tile.ref16 = d_ref; tile.plp_off = d_off; tile.rd = d_rd; tile.epos = d_ep;
    CHECK(bcfgpu_pipeline(ctx, &tile, NULL, NULL, &mo, &co));
    CHECK(bcfgpu_sync(ctx));

    /* ---- the record loop: what bcf_call2bcf + mcall leave in the record ---- */
    bcfgpu_site *ms = malloc((size_t)n_sites * sizeof *ms);
    bcfgpu_call_site *cs = malloc((size_t)n_sites * sizeof *cs);
    int8_t *gt = malloc(ncell * 2);
    CHECK(bcfgpu_memcpy_d2h(ctx, ms, mo.site, (size_t)n_sites * sizeof *ms));
    CHECK(bcfgpu_memcpy_d2h(ctx, cs, co.site, (size_t)n_sites * sizeof *cs));
    CHECK(bcfgpu_memcpy_d2h(ctx, gt, co.gt, ncell * 2));
    for (int k = 0; k < n_sites; k++) {
        if (cs[k].ret <= 0) continue;                         /* skipped by -v or not callable (vcfcall.c:1140-1144) */
        printf("%d\t%c\t", k + 1, nt[ms[k].a[0] < 0 ? 4 : ms[k].a[0]]);
        int first = 1;
        for (int i = 1; i < ms[k].n_alleles; i++) {
            if (cs[k].als_map[i] <= 0) continue;              /* allele trimmed away (mcall.c:547-570) */
            printf("%s%c", first ? "" : ",", i == ms[k].unseen ? '*' : nt[ms[k].a[i]]);
            first = 0;
        }
        if (first) printf(".");
        if (cs[k].qual_missing) printf("\t."); else printf("\t%.4g", cs[k].qual);
        printf("\t%d\t", cs[k].an);
        for (int i = 1; i < cs[k].nals_new; i++) printf("%s%d", i > 1 ? "," : "", cs[k].ac[i]);
        if (cs[k].nals_new < 2) printf(".");
        const int g0 = gt[((size_t)k * 2 + 0) * n_smpl], g1 = gt[((size_t)k * 2 + 1) * n_smpl];
        printf("\t%u\t", ms[k].depth);
        if (g0 < 0) printf("./.\n"); else printf("%d/%d\n", g0, g1);
    }
    fprintf(stderr, "%d sites, %d samples, %zu reads\n", n_sites, n_smpl, nr);

    bcfgpu_free(ctx, d_ref); bcfgpu_free(ctx, d_off); bcfgpu_free(ctx, d_rd); bcfgpu_free(ctx, d_ep);
    bcfgpu_free(ctx, mo.site); bcfgpu_free(ctx, mo.pl); bcfgpu_free(ctx, mo.dp4);
    bcfgpu_free(ctx, co.site); bcfgpu_free(ctx, co.gt); bcfgpu_free(ctx, co.pl);
    bcfgpu_destroy(ctx);
    free(ref16); free(off); free(rd); free(epos); free(ms); free(cs); free(gt);
    return 0;
}

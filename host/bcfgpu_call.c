/*  bcfgpu_call.c -- `bcftools call -m [-v]` over a VCF from `bcftools mpileup`, in plain C over the C-ABI of
 *  include/bcfgpu.h: the record loop of main_vcfcall (vcfcall.c:1089-1148) with mcall() on the device.
 *
 *      bcfgpu_call [-v] [-S samples.txt] [--ploidy-file file] [-G -|groups.txt [--group-samples-tag TAG]]
 *                  [-F AN_TAG,AC_TAG] [-a GQ,GP] <in.vcf>
 *          -S: the samples to keep, in that order: NAME [PLOIDY|SEX] per line, or a PED file (vcfcall.c:202-344)
 *          --ploidy-file: CHROM FROM TO SEX PLOIDY lines, '*' = default for the sex (ploidy.c)
 *          -G: sample groups with their own allele frequencies, '-' = every sample alone, or NAME GROUP lines
 *              (mcall.c:258-345); the frequencies come from FORMAT/QS or FORMAT/AD (--group-samples-tag)
 *          -F: INFO tags holding AN and AC of a prior population (mcall.c:1499-1520)
 *          -a: FORMAT/GQ and FORMAT/GP on called variant records (mcall.c:1618-1623)
 *
 *  Host: VCF text in, what mcall() reads from a record (alleles, FORMAT/PL, INFO/QS, INFO/I16) packed into the planes of
 *  bcfgpu_call_in, one bcfgpu_mcall over all records, then what mcall.c:1627-1681 does to the record: alleles trimmed with
 *  als_map, GT in front of the FORMAT fields, PL trimmed (or dropped), QUAL, INFO/AC, AN, DP4, MQ appended, I16 and QS
 *  removed.  Prints the data lines of the output VCF; tests/test_c_host.py compares them, byte for byte, with the
 *  reference's goldens of `call -m` (test.pl:276-308: mpileup.{1,3,4,5}, mpileup.X{,.2}, mpileup.hwe.*, call-G.*,
 *  call.af-fixation.*).
 *  Number=R tags of INFO and FORMAT follow the alleles (mcall_trim_and_update_numberR, mcall.c:1196-1265).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "bcfgpu.h"
#include "vcfio.h"

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s: %s (%d)\n", #call, bcfgpu_last_error(), rc_); exit(1); } } while (0)
#define DIE(...) do { fprintf(stderr, __VA_ARGS__); exit(1); } while (0)

static FILE *LN; static char *ln_buf; static size_t ln_len;      /* the record being written: a memory stream, framed by vcfio */

typedef struct { char *line; char **fld; int nfld; char **als; int nals, unseen, pl_idx, ad_idx; uint8_t *ploidy; } rec_t;

typedef struct { char chrom[256]; int from, to, ploidy; char sex[64]; } preg_t;

static char **split(char *s, char sep, int *n)
{
    int cap = 8; char **v = malloc((size_t)cap * sizeof *v); *n = 0;
    for (;;) {
        if (*n == cap) { cap *= 2; v = realloc(v, (size_t)cap * sizeof *v); }
        v[(*n)++] = s;
        s = strchr(s, sep);
        if (!s) break;
        *s++ = 0;
    }
    return v;
}

/* the Number=R tags the header declares (they follow the alleles when some are dropped) */
static char infoR[64][64], fmtR[64][64];
static int n_infoR = 0, n_fmtR = 0, has_fmt_qs = 0, has_fmt_ad = 0;

static void header_line(const char *ln)
{
    const int is_info = !strncmp(ln, "##INFO=<ID=", 11), is_fmt = !strncmp(ln, "##FORMAT=<ID=", 13);
    if (!is_info && !is_fmt) return;
    const char *id = ln + (is_info ? 11 : 13), *e = strchr(id, ',');
    if (is_fmt && e && e - id == 2) { has_fmt_qs |= !strncmp(id, "QS", 2); has_fmt_ad |= !strncmp(id, "AD", 2); }
    if (!e || !strstr(e, "Number=R") || e - id > 63) return;
    char (*tab)[64] = is_info ? infoR : fmtR; int *cnt = is_info ? &n_infoR : &n_fmtR;
    if (*cnt == 64) return;
    memcpy(tab[*cnt], id, (size_t)(e - id)); tab[*cnt][e - id] = 0; ++*cnt;
}

static int is_numberR(char (*tab)[64], int cnt, const char *key, size_t klen)
{
    for (int i = 0; i < cnt; ++i) if (strlen(tab[i]) == klen && !strncmp(tab[i], key, klen)) return 1;
    return 0;
}

/* a comma-separated Number=R value list with the kept alleles' values in their new places */
static void print_numberR(const char *vals, const int32_t *als_map, int nals, int nn)
{
    char *c = strdup(vals); int nv; char **v = split(c, ',', &nv);
    if (nv != nals) fputs(vals, LN);                        /* '.', or not one value per allele: left alone */
    else if (nn == 1) fputs(v[0], LN);
    else {
        const char *o[5] = { ".", ".", ".", ".", "." };
        for (int i = 0; i < nals; ++i) if (als_map[i] >= 0) o[als_map[i]] = v[i];
        for (int i = 0; i < nn; ++i) fprintf(LN, "%s%s", i ? "," : "", o[i]);
    }
    free(v); free(c);
}

static void *dev_upload(bcfgpu_ctx *ctx, const void *src, size_t bytes)
{
    void *d = NULL;
    CHECK(bcfgpu_malloc(ctx, bytes ? bytes : 16, &d));
    if (bytes) CHECK(bcfgpu_memcpy_h2d(ctx, d, src, bytes));
    return d;
}

int main(int argc, char **argv)
{
    int varonly = 0, out_tags = 0;
    const char *smpl_file = NULL, *ploidy_file = NULL, *grp_arg = NULL, *grp_tag = NULL;
    char prior_an_tag[64] = "", prior_ac_tag[64] = "";
    char out_mode = 'v'; const char *out_path = "-";
    while (argc > 2 && argv[1][0] == '-') {
        if (!strcmp(argv[1], "-v")) { varonly = 1; ++argv; --argc; }
        else if (!strcmp(argv[1], "-O") && argc > 3) { out_mode = argv[2][0]; argv += 2; argc -= 2; }      /* version.c:67-82 */
        else if (!strncmp(argv[1], "-O", 2) && argv[1][2]) { out_mode = argv[1][2]; ++argv; --argc; }
        else if (!strcmp(argv[1], "-o") && argc > 3) { out_path = argv[2]; argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "-G") && argc > 3) { grp_arg = argv[2]; argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "--group-samples-tag") && argc > 3) { grp_tag = argv[2]; argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "-F") && argc > 3) {
            const char *c = strchr(argv[2], ',');
            if (!c || c == argv[2] || !c[1] || c - argv[2] > 63 || strlen(c + 1) > 63) DIE("-F: expected AN_TAG,AC_TAG\n");
            memcpy(prior_an_tag, argv[2], (size_t)(c - argv[2])); prior_an_tag[c - argv[2]] = 0; strcpy(prior_ac_tag, c + 1);
            argv += 2; argc -= 2;
        }
        else if (!strcmp(argv[1], "-a") && argc > 3) {
            char *c = strdup(argv[2]); int nt; char **t = split(c, ',', &nt);
            for (int i = 0; i < nt; ++i)
                if (!strcmp(t[i], "GQ")) out_tags |= BCFGPU_CALL_FMT_GQ;
                else if (!strcmp(t[i], "GP")) out_tags |= BCFGPU_CALL_FMT_GP;
                else DIE("-a: unknown tag %s\n", t[i]);
            free(t); free(c); argv += 2; argc -= 2;
        }
        else if (!strcmp(argv[1], "-S") && argc > 3) { smpl_file = argv[2]; argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "--ploidy-file") && argc > 3) { ploidy_file = argv[2]; argv += 2; argc -= 2; }
        else break;
    }
    if (argc != 2) { fprintf(stderr, "usage: bcfgpu_call [-v] [-S samples.txt] [--ploidy-file file] [-G -|groups.txt [--group-samples-tag TAG]] [-F AN,AC] [-a GQ,GP] [-O v|z|u|b] [-o out] in.vcf|in.bcf\n"); return 2; }
    /* ploidy definition (ploidy.c): regions per sex, '*' lines = the sex's default; the last sex named is the default sex */
    preg_t *preg = NULL; int npreg = 0; char last_sex[64] = "";
    if (ploidy_file) {
        FILE *pf = fopen(ploidy_file, "r");
        if (!pf) DIE("cannot open %s\n", ploidy_file);
        char ln[1024], c[256], a[64], b[64], sx[64]; int pl;
        while (fgets(ln, sizeof ln, pf))
            if (sscanf(ln, "%255s %63s %63s %63s %d", c, a, b, sx, &pl) == 5) {
                preg = realloc(preg, (size_t)(npreg + 1) * sizeof *preg);
                preg_t *q = &preg[npreg++];
                strcpy(q->chrom, c); strcpy(q->sex, sx); q->ploidy = pl;
                q->from = !strcmp(a, "*") ? -1 : atoi(a); q->to = !strcmp(b, "*") ? -1 : atoi(b);
                strcpy(last_sex, sx);
            }
        fclose(pf);
    }
    vio_file *fin = vio_open_read(argv[1]);                  /* VCF, bgzipped VCF or BCF (hts_open of vcfcall.c) */
    if (!fin) DIE("%s\n", vio_error());
    vio_hdr *hdr = vio_read_hdr(fin);
    if (!hdr) DIE("%s\n", vio_error());
    char *buf = NULL; size_t bufcap = 0;
    rec_t *recs = NULL; int n = 0, cap = 0, S = -1, ngmax = 1, S_in = -1;
    int *col = NULL;                              /* output sample s = input column col[s] (bcf_subset with -S) */
    char **names = NULL;                          /* names of the input columns */
    char (*spec)[64] = NULL;                      /* its ploidy ("0", "1", "2") or sex name */
    for (int i = 0; i < vio_hdr_nlines(hdr); ++i) header_line(vio_hdr_line(hdr, i));
    {
        S_in = S = vio_hdr_nsamples(hdr);
        names = malloc((size_t)(S > 0 ? S : 1) * sizeof *names);
        for (int s = 0; s < S; ++s) names[s] = strdup(vio_hdr_sample(hdr, s));
        col = malloc((size_t)(S > 0 ? S : 1) * sizeof *col);
        spec = malloc((size_t)(S > 0 ? S : 1) * sizeof *spec);
        for (int s = 0; s < S; ++s) { col[s] = s; strcpy(spec[s], ploidy_file ? last_sex : "2"); }   /* vcfcall.c:645-650 */
        if (smpl_file) {
            FILE *sf = fopen(smpl_file, "r");
            if (!sf) DIE("cannot open %s\n", smpl_file);
            char ln[1024]; int m = 0;
            while (fgets(ln, sizeof ln, sf)) {
                char w[6][256]; const int nw = sscanf(ln, "%255s %255s %255s %255s %255s %255s", w[0], w[1], w[2], w[3], w[4], w[5]);
                if (nw < 1 || w[0][0] == '#') continue;
                const char *name = nw >= 5 ? w[1] : w[0];                   /* PED: family, sample, father, mother, sex */
                const char *sp = nw >= 5 ? (!strcmp(w[4], "1") ? "M" : "F") : nw >= 2 ? w[1] : "2";
                int i;
                for (i = 0; i < S_in; ++i) if (!strcmp(names[i], name)) break;
                if (i == S_in) continue;                                    /* not in the VCF: ignored */
                if (m == S_in) DIE("too many samples in %s\n", smpl_file);
                col[m] = i; strcpy(spec[m], sp); ++m;
            }
            fclose(sf);
            S = m;
        }
    }
    int rrc;
    while ((rrc = vio_read_line(fin, hdr, &buf, &bufcap)) > 0) {
        size_t l = strlen(buf);
        if (!l) continue;
        if (n == cap) { cap = cap ? 2 * cap : 1024; recs = realloc(recs, (size_t)cap * sizeof *recs); }
        rec_t *r = &recs[n++];
        r->line = strdup(buf);
        r->fld = split(r->line, '\t', &r->nfld);
        if (S_in < 0 || r->nfld != 9 + S_in) DIE("malformed VCF\n");
        /* alleles; the unseen allele as vcfcall.c:1102-1111 finds it */
        int nalt = 0; char *alt = strdup(r->fld[4]), **alts = split(alt, ',', &nalt);
        if (!strcmp(r->fld[4], ".")) nalt = 0;
        r->nals = 1 + nalt; r->als = malloc((size_t)r->nals * sizeof *r->als); r->als[0] = r->fld[3]; r->unseen = 0;
        for (int i = 0; i < nalt; ++i) {
            r->als[1 + i] = alts[i];
            const char *a = alts[i];
            if (!r->unseen && (a[0] == 'X' || (a[0] == '<' && (a[1] == 'X' || a[1] == '*') && a[2] == '>'))) r->unseen = 1 + i;
        }
        free(alts);
        if (r->nals > 5) DIE("more than 5 alleles at %s:%s\n", r->fld[0], r->fld[1]);
        const int ng = r->nals * (r->nals + 1) / 2;
        if (ng > ngmax) ngmax = ng;
        /* the ploidy of every sample at this record (set_ploidy, vcfcall.c:807-825) */
        r->ploidy = malloc((size_t)S);
        const int pos1 = atoi(r->fld[1]);
        for (int s = 0; s < S; ++s) {
            int pl = 2;
            if (!strcmp(spec[s], "0") || !strcmp(spec[s], "1") || !strcmp(spec[s], "2")) pl = atoi(spec[s]);
            else {
                int found = 0;
                for (int i = 0; i < npreg && !found; ++i)
                    if (preg[i].from >= 0 && !strcmp(preg[i].chrom, r->fld[0]) && !strcmp(preg[i].sex, spec[s]) && preg[i].from <= pos1 && pos1 <= preg[i].to) { pl = preg[i].ploidy; found = 1; }
                for (int i = 0; i < npreg && !found; ++i)
                    if (preg[i].from < 0 && !strcmp(preg[i].sex, spec[s])) { pl = preg[i].ploidy; found = 1; }
            }
            r->ploidy[s] = (uint8_t)pl;
        }
    }
    if (rrc < 0) DIE("%s\n", vio_error());
    vio_close(fin);
    if (S <= 0) DIE("no samples\n");

    /* ---- -G: the group of every sample; ids in the order the groups first appear in the file (mcall.c:308-330) ---- */
    int32_t *grp = NULL; int ngrp = 1;
    if (grp_arg && !strcmp(grp_arg, "-")) {
        grp = malloc((size_t)S * 4); ngrp = S;
        for (int s = 0; s < S; ++s) grp[s] = s;
    } else if (grp_arg) {
        FILE *gf = fopen(grp_arg, "r");
        if (!gf) DIE("cannot open %s\n", grp_arg);
        grp = malloc((size_t)S * 4);
        for (int s = 0; s < S; ++s) grp[s] = -1;
        char (*gname)[256] = NULL; char ln[1024], w0[256], w1[256]; ngrp = 0;
        while (fgets(ln, sizeof ln, gf)) {
            if (sscanf(ln, "%255s %255s", w0, w1) != 2 || w0[0] == '#') continue;
            int s, g;
            for (s = 0; s < S; ++s) if (!strcmp(names[col[s]], w0)) break;
            if (s == S) continue;                                           /* not among the samples called */
            for (g = 0; g < ngrp; ++g) if (!strcmp(gname[g], w1)) break;
            if (g == ngrp) { gname = realloc(gname, (size_t)(ngrp + 1) * sizeof *gname); strcpy(gname[ngrp++], w1); }
            grp[s] = g;
        }
        fclose(gf); free(gname);
        for (int s = 0; s < S; ++s) if (grp[s] < 0) DIE("sample %s is in no group of %s\n", names[col[s]], grp_arg);
    }
    if (ngrp > 1 && !grp_tag) grp_tag = has_fmt_qs ? "QS" : has_fmt_ad ? "AD" : NULL;       /* mcall.c:272-281 */
    if (ngrp > 1 && !grp_tag) DIE("-G needs FORMAT/QS or FORMAT/AD\n");
    int namax = 1;
    for (int k = 0; k < n; ++k) if (recs[k].nals > namax) namax = recs[k].nals;

    /* ---- what mcall() reads from the records: PL planes (missing / vector_end kept), QS, I16 ---- */
    int32_t *nals = malloc((size_t)n * 4), *unseen = malloc((size_t)n * 4);
    int32_t *pl = malloc((size_t)n * ngmax * S * 4);
    float *qs = calloc((size_t)n * 5, 4), *i16 = calloc((size_t)n * 16, 4);
    int32_t *ad = ngrp > 1 ? malloc((size_t)n * namax * S * 4) : NULL;
    int32_t *pan = prior_an_tag[0] ? malloc((size_t)n * 4) : NULL, *pac = prior_an_tag[0] ? malloc((size_t)n * 4 * 4) : NULL;
    const size_t l_pan = strlen(prior_an_tag), l_pac = strlen(prior_ac_tag);
    for (int k = 0; k < n; ++k) {
        rec_t *r = &recs[k];
        nals[k] = r->nals; unseen[k] = r->unseen;
        for (size_t i = 0; i < (size_t)ngmax * S; ++i) pl[(size_t)k * ngmax * S + i] = BCFGPU_INT32_VECTOR_END;
        if (ad) for (size_t i = 0; i < (size_t)namax * S; ++i) ad[(size_t)k * namax * S + i] = BCFGPU_INT32_VECTOR_END;
        if (pan) { pan[k] = BCFGPU_INT32_MISSING; for (int i = 0; i < 4; ++i) pac[(size_t)k * 4 + i] = BCFGPU_INT32_VECTOR_END; }
        /* FORMAT/PL */
        int nk; char *fmt = strdup(r->fld[8]), **keys = split(fmt, ':', &nk);
        r->pl_idx = r->ad_idx = -1;
        for (int i = 0; i < nk; ++i) {
            if (!strcmp(keys[i], "PL")) r->pl_idx = i;
            if (ad && !strcmp(keys[i], grp_tag)) r->ad_idx = i;
        }
        free(keys); free(fmt);
        if (r->pl_idx < 0) DIE("no FORMAT/PL at %s:%s\n", r->fld[0], r->fld[1]);
        if (ad && r->ad_idx < 0) DIE("FORMAT/%s is required with -G (%s:%s)\n", grp_tag, r->fld[0], r->fld[1]);     /* mcall.c:1476 */
        for (int s = 0; s < S; ++s) {
            char *smp = strdup(r->fld[9 + col[s]]); int nv; char **vals = split(smp, ':', &nv);
            if (r->pl_idx < nv) {
                int np; char **pv = split(vals[r->pl_idx], ',', &np);
                for (int j = 0; j < np && j < ngmax; ++j)
                    pl[((size_t)k * ngmax + j) * S + s] = !strcmp(pv[j], ".") ? BCFGPU_INT32_MISSING : atoi(pv[j]);
                free(pv);
            } else pl[((size_t)k * ngmax) * S + s] = BCFGPU_INT32_MISSING;
            if (ad && r->ad_idx < nv) {
                int na; char **av = split(vals[r->ad_idx], ',', &na);
                for (int j = 0; j < na && j < namax; ++j)
                    ad[((size_t)k * namax + j) * S + s] = !strcmp(av[j], ".") ? BCFGPU_INT32_MISSING : atoi(av[j]);
                free(av);
            } else if (ad) ad[((size_t)k * namax) * S + s] = BCFGPU_INT32_MISSING;
            free(vals); free(smp);
        }
        /* INFO/QS, INFO/I16 */
        char *info = strdup(r->fld[7]); int ni; char **iv = split(info, ';', &ni);
        for (int i = 0; i < ni; ++i) {
            if (pan && !strncmp(iv[i], prior_an_tag, l_pan) && iv[i][l_pan] == '=') {        /* mcall.c:1499-1520 */
                if (!strchr(iv[i], ',')) pan[k] = atoi(iv[i] + l_pan + 1);
                continue;
            }
            if (pan && !strncmp(iv[i], prior_ac_tag, l_pac) && iv[i][l_pac] == '=') {
                char *c = strdup(iv[i] + l_pac + 1); int nv; char **v = split(c, ',', &nv);
                for (int j = 0; j < nv && j < 4; ++j) pac[(size_t)k * 4 + j] = !strcmp(v[j], ".") ? BCFGPU_INT32_MISSING : atoi(v[j]);
                free(v); free(c);
                continue;
            }
            float *dst = !strncmp(iv[i], "QS=", 3) ? qs + (size_t)k * 5 : !strncmp(iv[i], "I16=", 4) ? i16 + (size_t)k * 16 : NULL;
            if (!dst) continue;
            const int lim = dst == qs + (size_t)k * 5 ? 5 : 16;
            int nv; char **v = split(strchr(iv[i], '=') + 1, ',', &nv);
            for (int j = 0; j < nv && j < lim; ++j) dst[j] = (float)atof(v[j]);
            free(v);
        }
        free(iv); free(info);
    }

    /* ---- the device ---- */
    bcfgpu_cfg cfg; memset(&cfg, 0, sizeof cfg);
    cfg.device = 0; cfg.n_smpl = S; cfg.max_sites = n; cfg.max_reads = 64;
    cfg.min_baseQ = 13; cfg.capQ = 60; cfg.call_theta = 1.1e-3; cfg.call_flag = varonly ? BCFGPU_CALL_VARONLY : 0; cfg.n_grp = ngrp; cfg.ploidy_max = 2;
    cfg.output_tags = out_tags;
    bcfgpu_ctx *ctx = NULL;
    CHECK(bcfgpu_create(&cfg, &ctx));
    /* everything goes up once; the records are called in runs of equal ploidy vectors (the ploidy is per call:
     * vcfcall.c:807-825 re-initialises it when it changes) -- the planes are [record][...]: a run is a slice */
    int32_t *d_nals = dev_upload(ctx, nals, (size_t)n * 4), *d_unseen = dev_upload(ctx, unseen, (size_t)n * 4);
    int32_t *d_plin = dev_upload(ctx, pl, (size_t)n * ngmax * S * 4);
    float *d_qs = dev_upload(ctx, qs, (size_t)n * 5 * 4), *d_i16 = dev_upload(ctx, i16, (size_t)n * 16 * 4);
    int32_t *d_ad = ad ? dev_upload(ctx, ad, (size_t)n * namax * S * 4) : NULL, *d_grp = grp ? dev_upload(ctx, grp, (size_t)S * 4) : NULL;
    int32_t *d_pan = pan ? dev_upload(ctx, pan, (size_t)n * 4) : NULL, *d_pac = pan ? dev_upload(ctx, pac, (size_t)n * 16) : NULL;
    void *d_site, *d_gt, *d_pl, *d_ploidy, *d_gq = NULL, *d_gp = NULL;
    if (out_tags & BCFGPU_CALL_FMT_GQ) CHECK(bcfgpu_malloc(ctx, (size_t)n * S * 4, &d_gq));
    if (out_tags & BCFGPU_CALL_FMT_GP) CHECK(bcfgpu_malloc(ctx, (size_t)n * ngmax * S * 4, &d_gp));
    CHECK(bcfgpu_malloc(ctx, (size_t)n * sizeof(bcfgpu_call_site), &d_site)); CHECK(bcfgpu_malloc(ctx, (size_t)n * 2 * S, &d_gt));
    CHECK(bcfgpu_malloc(ctx, (size_t)n * ngmax * S * 4, &d_pl)); CHECK(bcfgpu_malloc(ctx, (size_t)S + 16, &d_ploidy));
    for (int i = 0; i < n; ) {
        int j = i + 1, all2 = 1;
        while (j < n && !memcmp(recs[j].ploidy, recs[i].ploidy, (size_t)S)) ++j;
        for (int s = 0; s < S; ++s) all2 &= recs[i].ploidy[s] == 2;
        bcfgpu_call_in in; memset(&in, 0, sizeof in);
        in.n_sites = j - i; in.n_gt_max = ngmax; in.n_al_max = 0;
        in.nals = d_nals + i; in.unseen = d_unseen + i; in.pl = d_plin + (size_t)i * ngmax * S; in.qs = d_qs + (size_t)i * 5;
        in.i16 = d_i16 + (size_t)i * 16;
        if (d_ad) { in.ad = d_ad + (size_t)i * namax * S; in.n_al_max = namax; in.grp = d_grp; }
        if (d_pan) { in.prior_an = d_pan + i; in.prior_ac = d_pac + (size_t)i * 4; }
        if (!all2) { CHECK(bcfgpu_memcpy_h2d(ctx, d_ploidy, recs[i].ploidy, (size_t)S)); in.ploidy = d_ploidy; }
        bcfgpu_call_out out; memset(&out, 0, sizeof out);
        out.site = (bcfgpu_call_site*)d_site + i; out.gt = (int8_t*)d_gt + (size_t)i * 2 * S; out.pl = (int32_t*)d_pl + (size_t)i * ngmax * S;
        if (d_gq) out.gq = (int32_t*)d_gq + (size_t)i * S;
        if (d_gp) out.gp = (float*)d_gp + (size_t)i * ngmax * S;
        CHECK(bcfgpu_mcall(ctx, &in, &out));
        CHECK(bcfgpu_sync(ctx));                               /* (d_ploidy is reused by the next run) */
        i = j;
    }
    bcfgpu_call_site *cs = malloc((size_t)n * sizeof *cs);
    int8_t *gt = malloc((size_t)n * 2 * S); int32_t *opl = malloc((size_t)n * ngmax * S * 4);
    CHECK(bcfgpu_memcpy_d2h(ctx, cs, d_site, (size_t)n * sizeof *cs)); CHECK(bcfgpu_memcpy_d2h(ctx, gt, d_gt, (size_t)n * 2 * S));
    CHECK(bcfgpu_memcpy_d2h(ctx, opl, d_pl, (size_t)n * ngmax * S * 4));
    int32_t *gq = d_gq ? malloc((size_t)n * S * 4) : NULL; float *gp = d_gp ? malloc((size_t)n * ngmax * S * 4) : NULL;
    if (gq) CHECK(bcfgpu_memcpy_d2h(ctx, gq, d_gq, (size_t)n * S * 4));
    if (gp) CHECK(bcfgpu_memcpy_d2h(ctx, gp, d_gp, (size_t)n * ngmax * S * 4));
    CHECK(bcfgpu_sync(ctx));

    /* ---- the output header: the input's, for the samples kept, without the calling-only tags, plus what mcall_init
     * declares (vcfcall.c:670,703-704; mcall.c:382-394) ---- */
    if (smpl_file && vio_hdr_subset(hdr, S, col)) DIE("%s\n", vio_error());
    vio_hdr_remove(hdr, "INFO", "QS");
    vio_hdr_remove(hdr, "INFO", "I16");
    vio_hdr_append(hdr, "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">");
    if (out_tags & BCFGPU_CALL_FMT_GQ) vio_hdr_append(hdr, "##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Phred-scaled Genotype Quality\">");
    if (out_tags & BCFGPU_CALL_FMT_GP) vio_hdr_append(hdr, "##FORMAT=<ID=GP,Number=G,Type=Float,Description=\"Genotype posterior probabilities in the range 0 to 1\">");
    vio_hdr_append(hdr, "##INFO=<ID=AC,Number=A,Type=Integer,Description=\"Allele count in genotypes for each ALT allele, in the same order as listed\">");
    vio_hdr_append(hdr, "##INFO=<ID=AN,Number=1,Type=Integer,Description=\"Total number of alleles in called genotypes\">");
    vio_hdr_append(hdr, "##INFO=<ID=DP4,Number=4,Type=Integer,Description=\"Number of high-quality ref-forward , ref-reverse, alt-forward and alt-reverse bases\">");
    vio_hdr_append(hdr, "##INFO=<ID=MQ,Number=1,Type=Integer,Description=\"Average mapping quality\">");
    vio_file *fout = vio_open_write(out_path, out_mode);
    if (!fout || vio_write_hdr(fout, hdr)) DIE("%s\n", vio_error());
    LN = open_memstream(&ln_buf, &ln_len);
    if (!LN) DIE("open_memstream failed\n");
    /* ---- the record loop (vcfcall.c:1137-1147, mcall.c:1627-1681) ---- */
    for (int k = 0; k < n; ++k) {
        const rec_t *r = &recs[k];
        const bcfgpu_call_site *c = &cs[k];
        if (c->ret == -2 || (varonly && c->ret == 0) || c->ret < 0) continue;
        const int nn = c->nals_new, ngn = nn * (nn + 1) / 2;
        fprintf(LN, "%s\t%s\t%s\t%s\t", r->fld[0], r->fld[1], r->fld[2], r->fld[3]);
        {   /* ALT: the kept alleles in their new order */
            const char *al[5] = { 0, 0, 0, 0, 0 };
            for (int i = 0; i < r->nals; ++i) if (c->als_map[i] >= 0) al[c->als_map[i]] = r->als[i];
            if (nn < 2) fputc('.', LN);
            for (int i = 1; i < nn; ++i) fprintf(LN, "%s%s", i > 1 ? "," : "", al[i]);
        }
        if (c->qual_missing) fputs("\t.", LN); else fprintf(LN, "\t%g", (double)c->qual);
        fprintf(LN, "\t%s\t", r->fld[6]);
        {   /* INFO: I16 and QS go, AC / AN / DP4 / MQ come */
            char *info = strdup(r->fld[7]); int ni, first = 1; char **iv = split(info, ';', &ni);
            for (int i = 0; i < ni; ++i) {
                if (!strncmp(iv[i], "I16=", 4) || !strncmp(iv[i], "QS=", 3) || !strcmp(iv[i], ".")) continue;
                const char *eq = strchr(iv[i], '=');
                if (eq && nn != r->nals && is_numberR(infoR, n_infoR, iv[i], (size_t)(eq - iv[i]))) {
                    fprintf(LN, "%s%.*s=", first ? "" : ";", (int)(eq - iv[i]), iv[i]);
                    print_numberR(eq + 1, c->als_map, r->nals, nn);
                } else fprintf(LN, "%s%s", first ? "" : ";", iv[i]);
                first = 0;
            }
            free(iv); free(info);
            if (nn > 1) { fprintf(LN, "%sAC=", first ? "" : ";"); first = 0; for (int i = 1; i < nn; ++i) fprintf(LN, "%s%d", i > 1 ? "," : "", c->ac[i]); }
            fprintf(LN, "%sAN=%d", first ? "" : ";", c->an);
            if (c->has_i16) {
                fprintf(LN, ";DP4=%d,%d,%d,%d", c->dp4[0], c->dp4[1], c->dp4[2], c->dp4[3]);
                if (c->mq == BCFGPU_INT32_MISSING) fputs(";MQ=.", LN); else fprintf(LN, ";MQ=%d", c->mq);
            }
        }
        /* FORMAT: GT first, PL trimmed or dropped, the rest as it came */
        int nk; char *fmt = strdup(r->fld[8]), **keys = split(fmt, ':', &nk);
        fputs("\tGT", LN);
        for (int i = 0; i < nk; ++i) if (i != r->pl_idx || !c->pl_dropped) fprintf(LN, ":%s", keys[i]);
        const int called = nn > 1 && c->ret > 0;               /* mcall_call_genotypes ran: GP and GQ exist (mcall.c:1618-1623) */
        if (called && gp) fputs(":GP", LN);
        if (called && gq) fputs(":GQ", LN);
        for (int s = 0; s < S; ++s) {
            const int g0 = gt[((size_t)k * 2 + 0) * S + s], g1 = gt[((size_t)k * 2 + 1) * S + s];
            fputc('\t', LN);
            if (g0 == BCFGPU_GT_MISSING) fputc('.', LN); else fprintf(LN, "%d", g0);
            if (g1 != BCFGPU_GT_VECTOR_END) { fputc('/', LN); if (g1 == BCFGPU_GT_MISSING) fputc('.', LN); else fprintf(LN, "%d", g1); }
            char *smp = strdup(r->fld[9 + col[s]]); int nv; char **vals = split(smp, ':', &nv);
            for (int i = 0; i < nk; ++i) {
                if (i == r->pl_idx) {
                    if (c->pl_dropped) continue;
                    fputc(':', LN);
                    int printed = 0;
                    for (int j = 0; j < ngn; ++j) {
                        const int32_t v = opl[((size_t)k * ngmax + j) * S + s];
                        if (v == BCFGPU_INT32_VECTOR_END) break;
                        if (printed++) fputc(',', LN);
                        if (v == BCFGPU_INT32_MISSING) fputc('.', LN); else fprintf(LN, "%d", v);
                    }
                    if (!printed) fputc('.', LN);
                } else if (i < nv && nn != r->nals && is_numberR(fmtR, n_fmtR, keys[i], strlen(keys[i]))) {
                    fputc(':', LN);
                    print_numberR(vals[i], c->als_map, r->nals, nn);
                } else fprintf(LN, ":%s", i < nv ? vals[i] : ".");
            }
            if (called && gp) {
                fputc(':', LN);
                int printed = 0;
                for (int j = 0; j < ngn; ++j) {
                    uint32_t bits; memcpy(&bits, &gp[((size_t)k * ngmax + j) * S + s], 4);
                    if (bits == 0x7F800002u) break;
                    if (printed++) fputc(',', LN);
                    if (bits == 0x7F800001u) fputc('.', LN); else fprintf(LN, "%g", (double)gp[((size_t)k * ngmax + j) * S + s]);
                }
                if (!printed) fputc('.', LN);
            }
            if (called && gq) {
                const int32_t v = gq[(size_t)k * S + s];
                if (v == BCFGPU_INT32_MISSING) fputs(":.", LN); else fprintf(LN, ":%d", v);
            }
            free(vals); free(smp);
        }
        free(keys); free(fmt);
        fputc(0, LN); fflush(LN);                              /* the record, NUL-terminated, then the stream starts over */
        if (vio_write_line(fout, hdr, ln_buf)) DIE("%s\n", vio_error());
        rewind(LN);
    }
    if (vio_close(fout)) DIE("%s\n", vio_error());
    bcfgpu_destroy(ctx);
    return 0;
}

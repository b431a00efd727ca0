/*  bcfgpu_call.c -- `bcftools call -m [-v]` over a VCF from `bcftools mpileup`, in plain C over the C-ABI of
 *  include/bcfgpu.h: the record loop of main_vcfcall (vcfcall.c:1089-1148) with mcall() on the device.
 *
 *      bcfgpu_call [-v] [-S samples.txt | -s NAME,...] [--ploidy-file file | --ploidy GRCh37|GRCh38|X|Y|1] [-G -|groups.txt [--group-samples-tag TAG]]
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
#include <strings.h>
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

/* ---- call -C alleles -T targets [-i]: the record is re-expressed in the alleles of the target file before mcall() sees it
 * (mcall_constrain_alleles, mcall.c:1271-1421), the target line is chosen as next_line() does (vcfcall.c:501-605) with the
 * allele comparison of vcmp.c:55-119, and -i writes a line for every target that met no record (tgt_flush, vcfcall.c:408-455).
 * Host logic on the record text; the call itself is the device's, with -A. ---- */
typedef struct { char *chrom; int pos; char **als; int nals, used, order; } tgt_t;
static tgt_t *tgt = NULL; static int n_tgt = 0; static int *tgt_sorted = NULL;
static int cals = 0, insert_missed = 0;

static void tgt_parse(const char *path)
{
    FILE *f = fopen(path, "r");
    if (!f) DIE("cannot open %s\n", path);
    char ln[1 << 16], c[256], a[1 << 15]; int pos;
    while (fgets(ln, sizeof ln, f)) {
        if (sscanf(ln, "%255s %d %32767s", c, &pos, a) != 3) continue;
        tgt = realloc(tgt, (size_t)(n_tgt + 1) * sizeof *tgt);
        tgt_t *t = &tgt[n_tgt];
        t->chrom = strdup(c); t->pos = pos; t->used = 0; t->order = n_tgt;
        char *al = strdup(a); t->als = split(al, ',', &t->nals);
        ++n_tgt;
    }
    fclose(f);
    /* the regions of a sequence sorted by start (regidx), sequences in the order they first appear */
    tgt_sorted = malloc((size_t)(n_tgt ? n_tgt : 1) * sizeof *tgt_sorted);
    int m = 0;
    for (int i = 0; i < n_tgt; ++i) {
        int seen = 0;
        for (int j = 0; j < i; ++j) if (!strcmp(tgt[j].chrom, tgt[i].chrom)) { seen = 1; break; }
        if (seen) continue;
        const int m0 = m;
        for (int j = i; j < n_tgt; ++j) if (!strcmp(tgt[j].chrom, tgt[i].chrom)) tgt_sorted[m++] = j;
        for (int x = m0 + 1; x < m; ++x)                        /* stable insertion sort by position */
            for (int y = x; y > m0 && tgt[tgt_sorted[y]].pos < tgt[tgt_sorted[y - 1]].pos; --y) { const int t = tgt_sorted[y]; tgt_sorted[y] = tgt_sorted[y - 1]; tgt_sorted[y - 1] = t; }
    }
}
static size_t common_prefix_ci(const char *a, const char *b)
{
    size_t i = 0;
    while (a[i] && b[i] && (a[i] & ~32) == (b[i] & ~32) && ((a[i] | 32) >= 'a' && (a[i] | 32) <= 'z' ? 1 : a[i] == b[i])) ++i;
    return i;
}
static int ci_equal(const char *a, const char *b) { while (*a && *b) { char x = *a, y = *b; if (x >= 'a' && x <= 'z') x -= 32; if (y >= 'a' && y <= 'z') y -= 32; if (x != y) return 0; ++a; ++b; } return !*a && !*b; }
/* vcmp_set_ref / vcmp_find_allele: the difference of the two REF strings, then allele matching modulo that suffix */
typedef struct { int ndref; const char *dref; } vcmp_t;
static int vcmp_set_ref(vcmp_t *v, const char *r1, const char *r2)
{
    v->ndref = 0; v->dref = "";
    const size_t i = common_prefix_ci(r1, r2), l1 = strlen(r1), l2 = strlen(r2);
    if (i == l1 && i == l2) return 0;
    if (i < l1 && i < l2) return -1;
    if (i < l1) { v->dref = r1 + i; v->ndref = (int)(l1 - i); } else { v->dref = r2 + i; v->ndref = -(int)(l2 - i); }
    return 0;
}
static int vcmp_find_allele(const vcmp_t *v, char **als1, int n1, const char *al2)
{
    for (int i = 0; i < n1; ++i) {
        const char *a = als1[i];
        const size_t k = common_prefix_ci(a, al2), la = strlen(a), lb = strlen(al2);
        if (k < la && k < lb) continue;
        if (!v->ndref) { if (k == la && k == lb) return i; continue; }
        if (k < la) { if (v->ndref < 0 || !ci_equal(a + k, v->dref)) continue; return i; }
        if (v->ndref > 0 || !ci_equal(al2 + k, v->dref)) continue;
        return i;
    }
    return -1;
}
static int als_is_indel(char **als, int n)                     /* vcfcall.c:456-470 */
{
    if (n > 1 && als[1][0] == '<') return 0;
    for (int i = 0; i < n; ++i) if (als[i][0] != '<' && strlen(als[i]) > 1) return 1;
    return 0;
}
static int gt_index(int a, int b) { return a > b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a; }
static void gt_alleles(int igt, int *a, int *b) { int k = 0; while ((k + 1) * (k + 2) / 2 <= igt) ++k; *b = k; *a = igt - k * (k + 1) / 2; }

/* the -i lines of the targets in [beg0, end0] of `chrom` that met no record, appended to the event list */
typedef struct { int is_missed; int tgt; } event_t;             /* is_missed: print target `tgt`; else: record (index in recs) */
static event_t *events = NULL; static int n_events = 0;
static void push_event(int is_missed, int idx) { events = realloc(events, (size_t)(n_events + 1) * sizeof *events); events[n_events].is_missed = is_missed; events[n_events++].tgt = idx; }
static void flush_region(const char *chrom, long beg0, long end0)
{
    for (int x = 0; x < n_tgt; ++x) {
        tgt_t *t = &tgt[tgt_sorted[x]];
        if (strcmp(t->chrom, chrom) || t->pos - 1 < beg0 || t->pos - 1 > end0 || t->used) continue;
        t->used = 1;
        push_event(1, tgt_sorted[x]);
    }
}

/* The record `line` (S_in sample columns) in the alleles of target t: a new malloc'ed line, the same line when nothing
 * changes, or NULL when mcall() would return -2 (the site is skipped).  *unseen: in/out. */
static char *constrain_line(const char *line, const tgt_t *t, int S_in, int *unseen_io)
{
    if (t->nals > 5) DIE("Maximum accepted number of alleles is 5\n");
    char *c = strdup(line); int nf; char **f = split(c, '\t', &nf);
    int nalt = 0; char *altc = strdup(f[4]), **alts = split(altc, ',', &nalt);
    if (!strcmp(f[4], ".")) nalt = 0;
    const int nori = 1 + nalt, unseen = *unseen_io;
    vcmp_t vc;
    if (vcmp_set_ref(&vc, f[3], t->als[0]) < 0) DIE("The reference alleles are not compatible at %s:%s\n", f[0], f[1]);
    int amap[8], nals = 1, has_new = 0; const char *als[8];
    amap[0] = 0; als[0] = t->als[0];
    for (int i = 1; i < t->nals; ++i) {
        const int j = vcmp_find_allele(&vc, alts, nalt, t->als[i]);
        if (j + 1 == unseen) { free(alts); free(altc); free(f); free(c); return NULL; }   /* mcall.c:1294-1303 */
        if (j >= 0) amap[nals] = j + 1; else { amap[nals] = unseen >= 0 ? unseen : nori - 1; has_new = 1; }
        als[nals++] = t->als[i];
    }
    char *unseen_al = NULL;
    if (unseen) { amap[nals] = unseen; unseen_al = strdup(unseen == 0 ? f[3] : alts[unseen - 1]); als[nals++] = unseen_al; }
    if (!has_new && nals == nori) { free(unseen_al); free(alts); free(altc); free(f); free(c); return strdup(line); }
    int pl_map[64], npl = 0;
    for (int i = 0; i < nals; ++i) for (int j = 0; j <= i; ++j) pl_map[npl++] = gt_index(amap[i], amap[j]);
    /* FORMAT keys */
    int nk; char *fmt = strdup(f[8]), **keys = split(fmt, ':', &nk);
    int ipl = -1;
    for (int i = 0; i < nk; ++i) if (!strcmp(keys[i], "PL")) ipl = i;
    if (ipl < 0) DIE("no FORMAT/PL at %s:%s\n", f[0], f[1]);
    /* the widest PL vector of the record is the stride of bcf_get_format_int32 */
    int width = 1;
    char ***sv = malloc((size_t)S_in * sizeof *sv); int *snv = malloc((size_t)S_in * sizeof *snv); char **sc = malloc((size_t)S_in * sizeof *sc);
    for (int s = 0; s < S_in; ++s) {
        sc[s] = strdup(f[9 + s]); sv[s] = split(sc[s], ':', &snv[s]);
        if (ipl < snv[s]) { int w = 1; for (const char *q = sv[s][ipl]; *q; ++q) w += *q == ','; if (w > width) width = w; }
    }
    const size_t cap = strlen(line) * 4 + 4096 + (size_t)S_in * (size_t)npl * 12;
    char *out = malloc(cap); size_t o = 0;
    #define OUT(...) do { o += (size_t)snprintf(out + o, cap - o, __VA_ARGS__); } while (0)
    OUT("%s\t%s\t%s\t%s\t", f[0], f[1], f[2], als[0]);
    if (nals == 1) OUT("."); else for (int i = 1; i < nals; ++i) OUT("%s%s", i > 1 ? "," : "", als[i]);
    OUT("\t%s\t%s\t", f[5], f[6]);
    {   /* INFO: QS follows the alleles (absent alleles: 0) */
        int ni; char *info = strdup(f[7]), **iv = split(info, ';', &ni);
        for (int i = 0; i < ni; ++i) {
            if (i) OUT(";");
            if (!strncmp(iv[i], "QS=", 3)) {
                int nq; char *qc = strdup(iv[i] + 3), **qv = split(qc, ',', &nq);
                OUT("QS=");
                for (int k = 0; k < nals; ++k) OUT("%s%.9g", k ? "," : "", amap[k] < nq ? (double)(float)atof(qv[amap[k]]) : 0.);
                free(qv); free(qc);
            } else OUT("%s", iv[i]);
        }
        free(iv); free(info);
    }
    OUT("\t%s", f[8]);
    int32_t *ori = malloc((size_t)width * 4);
    for (int s = 0; s < S_in; ++s) {
        OUT("\t");
        for (int w = 0; w < width; ++w) ori[w] = BCFGPU_INT32_VECTOR_END;
        if (ipl < snv[s]) {
            int np; char *pc = strdup(sv[s][ipl]), **pv = split(pc, ',', &np);
            for (int j = 0; j < np && j < width; ++j) ori[j] = !strcmp(pv[j], ".") ? BCFGPU_INT32_MISSING : atoi(pv[j]);
            free(pv); free(pc);
        } else ori[0] = BCFGPU_INT32_MISSING;
        for (int k = 0; k < nk; ++k) {
            if (k) OUT(":");
            if (k == ipl) {
                int printed = 0;
                for (int g = 0; g < npl; ++g) {
                    int32_t v = pl_map[g] < width ? ori[pl_map[g]] : BCFGPU_INT32_VECTOR_END;
                    if (v == BCFGPU_INT32_MISSING && unseen >= 0) {          /* an allele mpileup did not see: the unseen allele stands in */
                        int ia, ib; gt_alleles(pl_map[g], &ia, &ib);
                        int ko = gt_index(ia, unseen);
                        if ((ko < width ? ori[ko] : BCFGPU_INT32_VECTOR_END) == BCFGPU_INT32_MISSING) ko = gt_index(ib, unseen);
                        if ((ko < width ? ori[ko] : BCFGPU_INT32_VECTOR_END) == BCFGPU_INT32_MISSING) ko = gt_index(unseen, unseen);
                        v = ko < width ? ori[ko] : BCFGPU_INT32_VECTOR_END;
                    }
                    if (g == 0 && v == BCFGPU_INT32_VECTOR_END) v = BCFGPU_INT32_MISSING;
                    if (v == BCFGPU_INT32_VECTOR_END) break;
                    if (printed++) OUT(",");
                    if (v == BCFGPU_INT32_MISSING) OUT("."); else OUT("%d", v);
                }
            } else if (k < snv[s] && is_numberR(fmtR, n_fmtR, keys[k], strlen(keys[k])) && strcmp(sv[s][k], ".")) {
                int nv; char *vc2 = strdup(sv[s][k]), **vv = split(vc2, ',', &nv);      /* Number=R: new[k] = old[als_map[k]] */
                for (int a = 0; a < nals; ++a) OUT("%s%s", a ? "," : "", amap[a] < nv ? vv[amap[a]] : ".");
                free(vv); free(vc2);
            } else OUT("%s", k < snv[s] ? sv[s][k] : ".");
        }
    }
    #undef OUT
    free(ori);
    for (int s = 0; s < S_in; ++s) { free(sv[s]); free(sc[s]); }
    free(sv); free(snv); free(sc); free(keys); free(fmt); free(unseen_al); free(alts); free(altc); free(f); free(c);
    *unseen_io = unseen ? nals - 1 : unseen;
    return out;
}

/* next_line (vcfcall.c:501-605): the target of this record, or -1 when the record is not to be called */
static int pick_target(const char *chrom, int pos, const char *ref, char **alts, int nalt)
{
    int best = -1, bestn = 0, any = 0;
    char **als = malloc((size_t)(nalt + 1) * sizeof *als);
    als[0] = (char*)ref; for (int i = 0; i < nalt; ++i) als[1 + i] = alts[i];
    const int rec_indel = als_is_indel(als, nalt + 1) ? 1 : -1;
    for (int x = 0; x < n_tgt; ++x) {
        tgt_t *t = &tgt[tgt_sorted[x]];
        if (strcmp(t->chrom, chrom) || t->pos != pos) continue;
        any = 1;
        if (t->used) continue;
        vcmp_t vc; int n = 0;
        if (vcmp_set_ref(&vc, ref, t->als[0]) == 0) {
            n = 1;
            if (nalt > 0 && t->nals > 1) for (int i = 1; i < t->nals; ++i) n += vcmp_find_allele(&vc, alts, nalt, t->als[i]) >= 0;
        }
        n *= rec_indel * (als_is_indel(t->als, t->nals) ? 1 : -1);
        if (best < 0 || n > bestn) { best = tgt_sorted[x]; bestn = n; }
    }
    free(als);
    (void)any;
    return best;
}

/* ---- -t / -T / -r / -R: the sites to look at, as (sequence, first, last) with 1-based inclusive positions; last < 0: to the end ---- */
typedef struct { char *chrom; long beg, end; } sflt_t;
static sflt_t *sflt; static int n_sflt;
static void site_filter_add(const char *spec)
{
    sflt = realloc(sflt, (size_t)(n_sflt + 1) * sizeof *sflt);
    sflt_t *q = &sflt[n_sflt++];
    const char *c = strrchr(spec, ':');
    q->beg = 1; q->end = -1;
    if (!c) { q->chrom = strdup(spec); return; }
    q->chrom = strndup(spec, (size_t)(c - spec));
    char *e; q->beg = strtol(c + 1, &e, 10);
    if (e == c + 1) { free(q->chrom); q->chrom = strdup(spec); q->beg = 1; return; }      /* a ':' inside the sequence name */
    if (*e == '-') { q->end = e[1] ? strtol(e + 1, NULL, 10) : -1; } else q->end = q->beg;
}
static void site_filter_file(const char *path)
{
    FILE *f = fopen(path, "r");
    if (!f) DIE("cannot open %s\n", path);
    char ln[4096], c[1024]; long a, b;
    while (fgets(ln, sizeof ln, f)) {
        if (ln[0] == '#') continue;
        const int k = sscanf(ln, "%1023s %ld %ld", c, &a, &b);
        if (k < 1) continue;
        sflt = realloc(sflt, (size_t)(n_sflt + 1) * sizeof *sflt);
        sflt[n_sflt].chrom = strdup(c); sflt[n_sflt].beg = k >= 2 ? a : 1; sflt[n_sflt].end = k >= 3 ? b : k == 2 ? a : -1;
        ++n_sflt;
    }
    fclose(f);
}
static int site_filter_has(const char *line)
{
    const char *t = strchr(line, '\t');
    if (!t) return 0;
    const long pos = atol(t + 1);
    for (int i = 0; i < n_sflt; ++i)
        if (strlen(sflt[i].chrom) == (size_t)(t - line) && !strncmp(sflt[i].chrom, line, (size_t)(t - line)) && pos >= sflt[i].beg && (sflt[i].end < 0 || pos <= sflt[i].end)) return 1;
    return 0;
}

/* the sites `call` passes over before anything else (vcfcall.c:1095-1099): -V snps / indels by htslib's bcf_is_snp (every allele one
 * base that is not '*', or the symbolic <X> / <*>), and -- unless -M -- a reference allele that starts with N */
static int unwanted_site(const char *line, int acgt_only, int skip_kind)
{
    if (n_sflt && !site_filter_has(line)) return 1;
    const char *f = line; int tabs = 0;
    for (; *f && tabs < 3; ++f) if (*f == '\t') ++tabs;
    if (tabs < 3) return 0;
    const char *ref = f, *re = strchr(ref, '\t');
    if (!re) return 0;
    const char *alt = re + 1, *ae = strchr(alt, '\t');
    if (!ae) ae = alt + strlen(alt);
    if (skip_kind) {
        int is_snp = (re - ref == 1 && ref[0] != '*');
        if (!(ae - alt == 1 && alt[0] == '.'))
            for (const char *a = alt; is_snp && a < ae; ) {
                const char *e = memchr(a, ',', (size_t)(ae - a)); if (!e) e = ae;
                const size_t l = (size_t)(e - a);
                if (!((l == 1 && a[0] != '*') || (l == 3 && a[0] == '<' && (a[1] == 'X' || a[1] == '*') && a[2] == '>'))) is_snp = 0;
                a = e + 1;
            }
        if (skip_kind == 1 && is_snp) return 1;                  /* -V snps: CF_INDEL_ONLY */
        if (skip_kind == 2 && !is_snp) return 1;                 /* -V indels: CF_NO_INDEL */
    }
    return acgt_only && (ref[0] == 'N' || ref[0] == 'n');
}

int main(int argc, char **argv)
{
    {   /* call's long option names (vcfcall.c:946-981) are read as their short forms; -f is the old spelling of -a (vcfcall.c:995) */
        static const char *alias[][2] = {
            { "--variants-only", "-v" }, { "--multiallelic-caller", "-m" }, { "--keep-alts", "-A" }, { "--insert-missed", "-i" }, { "--constrain", "-C" },
            { "--targets-file", "-T" }, { "--prior", "-P" }, { "--output-type", "-O" }, { "--output", "-o" }, { "--group-samples", "-G" },
            { "--prior-freqs", "-F" }, { "--annotate", "-a" }, { "--format-fields", "-a" }, { "-f", "-a" }, { "--samples-file", "-S" }, { "--samples", "-s" },
            { "--pval-threshold", "-p" } };
        for (int i = 1; i < argc; ++i)
            for (size_t k = 0; k < sizeof alias / sizeof alias[0]; ++k) if (!strcmp(argv[i], alias[k][0])) argv[i] = (char *)alias[k][1];
    }
    int varonly = 0, out_tags = 0, keepalt = 0;
    int acgt_only = 1, skip_kind = 0;                           /* vcfcall.c:937 (CF_ACGT_ONLY is the default); -V: 1 = snps, 2 = indels */
    const char *tgt_file = NULL; double prior = 1.1e-3;
    const char *smpl_file = NULL, *smpl_list = NULL, *ploidy_file = NULL, *ploidy_alias = NULL, *grp_arg = NULL, *grp_tag = NULL;
    char prior_an_tag[64] = "", prior_ac_tag[64] = "";
    char out_mode = 'v'; const char *out_path = "-";
    int32_t gv_range[16]; int gv_n = 0;                         /* -g INT,...: gvcf_init (gvcf.c:47-73) */
    while (argc > 2 && argv[1][0] == '-') {
        if (!strcmp(argv[1], "-v")) { varonly = 1; ++argv; --argc; }
        else if (!strcmp(argv[1], "-m")) { ++argv; --argc; }                                     /* the multiallelic caller: the only one here */
        else if (!strcmp(argv[1], "-A")) { keepalt = 1; ++argv; --argc; }
        else if (!strcmp(argv[1], "-M") || !strcmp(argv[1], "--keep-masked-refs")) { acgt_only = 0; ++argv; --argc; }      /* vcfcall.c:1000 */
        else if (!strcmp(argv[1], "-N") || !strcmp(argv[1], "--skip-Ns")) { acgt_only = 1; ++argv; --argc; }               /* vcfcall.c:1001: the default */
        else if ((!strcmp(argv[1], "-V") || !strcmp(argv[1], "--skip-variants")) && argc > 3) {                              /* vcfcall.c:1032-1036 */
            if (!strcasecmp(argv[2], "snps")) skip_kind = 1; else if (!strcasecmp(argv[2], "indels")) skip_kind = 2;
            else DIE("Unknown skip category \"%s\" (-V argument must be \"snps\" or \"indels\")\n", argv[2]);
            argv += 2; argc -= 2;
        }
        else if (!strcmp(argv[1], "--threads") && argc > 3) { argv += 2; argc -= 2; }                                        /* (compression threads: nothing to do here) */
        else if (!strcmp(argv[1], "--no-version")) { ++argv; --argc; }                                                       /* (no ##bcftools_callVersion lines are written anyway) */
        else if (!strcmp(argv[1], "-i")) { insert_missed = 1; ++argv; --argc; }
        else if (!strcmp(argv[1], "-C") && argc > 3) { if (strcmp(argv[2], "alleles")) DIE("-C: only `alleles` is supported\n"); cals = 1; argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "-T") && argc > 3) { tgt_file = argv[2]; argv += 2; argc -= 2; }
        else if ((!strcmp(argv[1], "-t") || !strcmp(argv[1], "--targets") || !strcmp(argv[1], "-r") || !strcmp(argv[1], "--regions")) && argc > 3) {
            /* -t / -r CHR[:POS | :BEG-END],...: only the records whose POS lies there (bcf_sr_set_targets / _regions, vcfcall.c:612-626; -r without
             * an index is a filter over the stream here) */
            char *c = strdup(argv[2]); int nt; char **t = split(c, ',', &nt);
            for (int i = 0; i < nt; ++i) site_filter_add(t[i]);
            free(t); free(c); argv += 2; argc -= 2;
        }
        else if ((!strcmp(argv[1], "-R") || !strcmp(argv[1], "--regions-file")) && argc > 3) { site_filter_file(argv[2]); argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "-P") && argc > 3) { prior = atof(argv[2]); argv += 2; argc -= 2; }      /* vcfcall.c:931-943 */
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
        else if ((!strcmp(argv[1], "-g") || !strcmp(argv[1], "--gvcf")) && argc > 3) {
            char *c = strdup(argv[2]); int nt; char **t = split(c, ',', &nt);
            if (nt < 1 || nt > 16) DIE("Could not parse: --gvcf %s\n", argv[2]);
            for (int i = 0; i < nt; ++i) { char *e; gv_range[i] = (int32_t)strtol(t[i], &e, 10); if (e == t[i] || *e) DIE("Could not parse: --gvcf %s\n", argv[2]); }
            gv_n = nt; free(t); free(c); argv += 2; argc -= 2;
        }
        else if ((!strncmp(argv[1], "-g", 2) && argv[1][2]) || (!strncmp(argv[1], "-mg", 3) && argv[1][3])) {   /* -g0,2,5; the `-mg0` of test.pl:277 */
            char *c = strdup(argv[1] + (argv[1][1] == 'm' ? 3 : 2)); int nt; char **t = split(c, ',', &nt);
            if (nt < 1 || nt > 16) DIE("Could not parse: --gvcf %s\n", argv[1] + 2);
            for (int i = 0; i < nt; ++i) { char *e; gv_range[i] = (int32_t)strtol(t[i], &e, 10); if (e == t[i] || *e) DIE("Could not parse: --gvcf %s\n", argv[1] + 2); }
            gv_n = nt; free(t); free(c); ++argv; --argc;
        }
        else if (!strcmp(argv[1], "-S") && argc > 3) { smpl_file = argv[2]; argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "-s") && argc > 3) { smpl_list = argv[2]; smpl_file = argv[2]; argv += 2; argc -= 2; }          /* -s LIST: the names, comma-separated (vcfcall.c:1050) */
        else if (!strcmp(argv[1], "-p") && argc > 3) { argv += 2; argc -= 2; }                                                 /* --pval-threshold: read by the consensus caller only (vcfcall.c:1038) */
        else if (!strcmp(argv[1], "--ploidy-file") && argc > 3) { ploidy_file = argv[2]; argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "--ploidy") && argc > 3) { ploidy_alias = argv[2]; argv += 2; argc -= 2; }                       /* vcfcall.c:976, 827-855 */
        else if (!strcmp(argv[1], "-X")) { ploidy_alias = "X"; ++argv; --argc; }                                                  /* vcfcall.c:991 */
        else if (!strcmp(argv[1], "-Y")) { ploidy_alias = "Y"; ++argv; --argc; }                                                  /* vcfcall.c:992 */
        else break;
    }
    if (gv_n && varonly) DIE("The two options cannot be combined: --variants-only and --gvcf\n");       /* vcfcall.c:1085 */
    if (gv_n && cals) DIE("-g with -C alleles is not supported\n");
    if (argc != 2) { fprintf(stderr, "usage: bcfgpu_call [-v] [-M] [-V snps|indels] [-t|-r REGIONS] [-T|-R FILE] [-g INT,...] [-S samples.txt | -s NAME,...] [--ploidy-file file | --ploidy GRCh37|GRCh38|X|Y|1] [-G -|groups.txt [--group-samples-tag TAG]] [-F AN,AC] [-a GQ,GP] [-A] [-P theta] [-C alleles -T targets.tab [-i]] [-O v|z|u|b] [-o out] in.vcf|in.bcf\n"); return 2; }
    /* ploidy definition (ploidy.c): regions per sex, '*' lines = the sex's default; the last sex named is the default sex */
    preg_t *preg = NULL; int npreg = 0; char last_sex[64] = "";
    char *alias_text = NULL;
    if (ploidy_alias) {
        /* --ploidy ALIAS: the definitions `call` carries with it (vcfcall.c:138-199), in the format of a --ploidy-file: the
         * haploid stretches of the human sex chromosomes outside the pseudo-autosomal regions and the mitochondrion, by assembly;
         * "X" / "Y" / "1": males haploid / males haploid and females absent / everybody haploid, whatever the sequence */
        static const struct { const char *name; long x_par1_end, x_par2_beg, x_end, y_end; } asm_[2] = {
            { "GRCh37", 60000, 2699521, 154931043, 59373566 }, { "GRCh38", 9999, 2781480, 155701381, 57227415 } };
        size_t al = 0; FILE *m = open_memstream(&alias_text, &al);
        int known = 0;
        for (int k = 0; k < 2; ++k)
            if (!strcmp(ploidy_alias, asm_[k].name)) {
                for (int pre = 0; pre < 2; ++pre) {
                    const char *c = pre ? "chr" : "";
                    fprintf(m, "%sX 1 %ld M 1\n%sX %ld %ld M 1\n%sY 1 %ld M 1\n%sY 1 %ld F 0\n", c, asm_[k].x_par1_end, c, asm_[k].x_par2_beg, asm_[k].x_end, c, asm_[k].y_end, c, asm_[k].y_end);
                    fprintf(m, "%s 1 16569 M 1\n%s 1 16569 F 1\n", pre ? "chrM" : "MT", pre ? "chrM" : "MT");
                }
                fprintf(m, "* * * M 2\n* * * F 2\n");
                known = 1;
            }
        if (!strcmp(ploidy_alias, "X")) { fprintf(m, "* * * M 1\n* * * F 2\n"); known = 1; }
        if (!strcmp(ploidy_alias, "Y")) { fprintf(m, "* * * M 1\n* * * F 0\n"); known = 1; }
        if (!strcmp(ploidy_alias, "1")) { fprintf(m, "* * * * 1\n"); known = 1; }
        fclose(m);
        if (!known) DIE("--ploidy: GRCh37, GRCh38, X, Y or 1 (or a --ploidy-file)\n");
        if (ploidy_file) DIE("--ploidy and --ploidy-file exclude each other\n");
        ploidy_file = "--ploidy";
    }
    if (ploidy_file) {
        FILE *pf = alias_text ? fmemopen(alias_text, strlen(alias_text), "r") : fopen(ploidy_file, "r");
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
            char *lbuf = NULL;
            if (smpl_list) { lbuf = strdup(smpl_list); for (char *c = lbuf; *c; ++c) if (*c == ',') *c = '\n'; }
            FILE *sf = smpl_list ? fmemopen(lbuf, strlen(lbuf), "r") : fopen(smpl_file, "r");
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
            fclose(sf); free(lbuf);
            S = m;
        }
    }
    char *prev_chrom = NULL; long prev_pos0 = 0;
    if (cals) { if (!tgt_file) DIE("-C alleles needs -T targets\n"); tgt_parse(tgt_file); }
    else if (tgt_file) site_filter_file(tgt_file);           /* -T without -C alleles: the targets restrict the sites (vcfcall.c:612-617) */
    int rrc;
    while ((rrc = vio_read_line(fin, hdr, &buf, &bufcap)) > 0) {
        size_t l = strlen(buf);
        if (!l) continue;
        char *use = buf, *owned = NULL;
        if (cals) {                                              /* -C alleles: pair the record with a target, rewrite it */
            char *c2 = strdup(buf); int nf2; char **f2 = split(c2, '\t', &nf2);
            if (nf2 != 9 + S_in) DIE("malformed VCF\n");
            int nalt2 = 0; char *ac = strdup(f2[4]), **av = split(ac, ',', &nalt2);
            if (!strcmp(f2[4], ".")) nalt2 = 0;
            const int pos2 = atoi(f2[1]);
            const int ti = pick_target(f2[0], pos2, f2[3], av, nalt2);
            int skip = ti < 0;
            if (!skip) {
                tgt[ti].used = 1;
                int un = 0;
                for (int i = 0; i < nalt2; ++i) { const char *a = av[i]; if (!un && (a[0] == 'X' || (a[0] == '<' && (a[1] == 'X' || a[1] == '*') && a[2] == '>'))) un = 1 + i; }
                owned = constrain_line(buf, &tgt[ti], S_in, &un);
                if (!owned) skip = 1; else use = owned;
                if (!skip && unwanted_site(use, acgt_only, skip_kind)) { free(owned); owned = NULL; skip = 1; }   /* (the skipped record flushes no targets either: vcfcall.c:1095-1099 come before tgt_flush) */
                if (!skip && insert_missed) {                    /* tgt_flush (vcfcall.c:426-455) */
                    const long p0 = pos2 - 1;
                    if (!prev_chrom) flush_region(f2[0], 0, p0 - 1);
                    else if (strcmp(prev_chrom, f2[0])) { flush_region(prev_chrom, prev_pos0 + 1, 1L << 40); flush_region(f2[0], 0, p0 - 1); }
                    else flush_region(prev_chrom, prev_pos0, p0 - 1);
                    free(prev_chrom); prev_chrom = strdup(f2[0]); prev_pos0 = p0;
                }
            }
            free(av); free(ac); free(f2); free(c2);
            if (skip) continue;
        } else if (unwanted_site(use, acgt_only, skip_kind)) continue;
        if (n == cap) { cap = cap ? 2 * cap : 1024; recs = realloc(recs, (size_t)cap * sizeof *recs); }
        if (cals) push_event(0, n);
        rec_t *r = &recs[n++];
        r->line = strdup(use);
        free(owned);
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
                for (int i = 0; i < npreg && !found; ++i)                       /* a sex without a default of its own takes the "*" sex's (ploidy.c:122-127) */
                    if (preg[i].from < 0 && !strcmp(preg[i].sex, "*")) { pl = preg[i].ploidy; found = 1; }
            }
            r->ploidy[s] = (uint8_t)pl;
        }
    }
    if (rrc < 0) DIE("%s\n", vio_error());
    vio_close(fin);
    if (cals && insert_missed) {                                 /* the targets behind the last record, then the sequences without any */
        if (prev_chrom) flush_region(prev_chrom, prev_pos0, 1L << 40);
        for (int x = 0; x < n_tgt; ++x) if (!tgt[tgt_sorted[x]].used) flush_region(tgt[tgt_sorted[x]].chrom, 0, 1L << 40);
    }
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
    cfg.min_baseQ = 13; cfg.capQ = 60; cfg.call_theta = prior; cfg.call_flag = (varonly ? BCFGPU_CALL_VARONLY : 0) | (keepalt ? BCFGPU_CALL_KEEPALT : 0); cfg.n_grp = ngrp; cfg.ploidy_max = 2;
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
    if (gv_n) {                                                  /* gvcf_update_header, on the reader's header (vcfcall.c:661-666) */
        vio_hdr_append(hdr, "##INFO=<ID=END,Number=1,Type=Integer,Description=\"End position of the variant described in this record\">");
        vio_hdr_append(hdr, "##INFO=<ID=MinDP,Number=1,Type=Integer,Description=\"Minimum per-sample depth in this gVCF block\">");
    }
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
    /* ---- -g: gVCF blocks over the records that are written (vcfcall.c:1145-1149; gvcf_write, gvcf.c:88-226).  What
     * gvcf_write looks at goes to the device as arrays: may the record join (mcall() returned 1: the reference allele alone),
     * FORMAT/DP of every sample, position, sequence, INFO/END; the block table and the blocks' DP come back. ---- */
    int32_t *gv_w = NULL, *gv_blk = NULL, *gv_min = NULL, *gv_dp = NULL; bcfgpu_gvcf_block *gv_block = NULL; int gv_nw = 0;
    if (gv_n) {
        gv_w = malloc((size_t)(n + 1) * 4);                      /* record k is written record gv_w[k], or -1 */
        int32_t *pos = malloc((size_t)(n + 1) * 4), *rid = malloc((size_t)(n + 1) * 4), *endp = malloc((size_t)(n + 1) * 4);
        uint8_t *ro = malloc((size_t)n + 1);
        int32_t *dp = malloc(((size_t)n * S + 1) * 4);
        char **chroms = NULL; int nchrom = 0;
        for (int k = 0; k < n; ++k) {
            const rec_t *r = &recs[k];
            if (cs[k].ret < 0) { gv_w[k] = -1; continue; }
            const int w = gv_nw++;
            gv_w[k] = w;
            pos[w] = atoi(r->fld[1]) - 1; endp[w] = pos[w];
            int ci; for (ci = 0; ci < nchrom; ++ci) if (!strcmp(chroms[ci], r->fld[0])) break;
            if (ci == nchrom) { chroms = realloc(chroms, (size_t)(nchrom + 1) * sizeof *chroms); chroms[nchrom++] = r->fld[0]; }
            rid[w] = ci;
            ro[w] = cs[k].ret == 1;
            const char *e = strstr(r->fld[7], "END=");
            if (e && (e == r->fld[7] || e[-1] == ';')) endp[w] = atoi(e + 4) - 1;
            int nk, dpi = -1; char *fmt = strdup(r->fld[8]), **keys = split(fmt, ':', &nk);
            for (int i = 0; i < nk; ++i) if (!strcmp(keys[i], "DP")) dpi = i;
            free(keys); free(fmt);
            for (int s2 = 0; s2 < S; ++s2) {
                int32_t v = INT32_MIN;                           /* missing: the record stays as it is */
                if (dpi >= 0) {
                    char *smp = strdup(r->fld[9 + col[s2]]); int nv; char **vals = split(smp, ':', &nv);
                    if (dpi < nv && strcmp(vals[dpi], ".")) v = atoi(vals[dpi]);
                    free(vals); free(smp);
                }
                dp[(size_t)w * S + s2] = v;
            }
        }
        free(chroms);
        if (gv_nw) {
            void *d_pos = dev_upload(ctx, pos, (size_t)gv_nw * 4), *d_rid = dev_upload(ctx, rid, (size_t)gv_nw * 4), *d_end = dev_upload(ctx, endp, (size_t)gv_nw * 4);
            void *d_ro = dev_upload(ctx, ro, (size_t)gv_nw), *d_dp = dev_upload(ctx, dp, (size_t)gv_nw * S * 4);
            void *d_blk, *d_min, *d_block, *d_gdp;
            CHECK(bcfgpu_malloc(ctx, (size_t)gv_nw * 4, &d_blk)); CHECK(bcfgpu_malloc(ctx, (size_t)gv_nw * 4, &d_min));
            CHECK(bcfgpu_malloc(ctx, (size_t)gv_nw * sizeof(bcfgpu_gvcf_block), &d_block)); CHECK(bcfgpu_malloc(ctx, (size_t)gv_nw * S * 4, &d_gdp));
            bcfgpu_gvcf_in gi; memset(&gi, 0, sizeof gi);
            gi.n_sites = gv_nw; gi.n_range = gv_n; gi.dp_range = gv_range; gi.pos = d_pos; gi.rid = d_rid; gi.end = d_end; gi.ref_only = d_ro; gi.dp = d_dp;
            bcfgpu_gvcf_out go; memset(&go, 0, sizeof go);
            go.blk = d_blk; go.min_dp = d_min; go.block = d_block; go.dp = d_gdp;
            int32_t nb = 0;
            CHECK(bcfgpu_gvcf_blocks(ctx, &gi, &go, &nb));
            gv_blk = malloc((size_t)gv_nw * 4); gv_min = malloc((size_t)gv_nw * 4);
            gv_block = malloc((size_t)(nb + 1) * sizeof *gv_block); gv_dp = malloc(((size_t)nb * S + 1) * 4);
            CHECK(bcfgpu_memcpy_d2h(ctx, gv_blk, d_blk, (size_t)gv_nw * 4)); CHECK(bcfgpu_memcpy_d2h(ctx, gv_min, d_min, (size_t)gv_nw * 4));
            if (nb) { CHECK(bcfgpu_memcpy_d2h(ctx, gv_block, d_block, (size_t)nb * sizeof *gv_block)); CHECK(bcfgpu_memcpy_d2h(ctx, gv_dp, d_gdp, (size_t)nb * S * 4)); }
            CHECK(bcfgpu_sync(ctx));
        }
        free(pos); free(rid); free(endp); free(ro); free(dp);
    }
    /* ---- the record loop (vcfcall.c:1137-1147, mcall.c:1627-1681) ---- */
    const int n_out = cals ? n_events : n;
    for (int ev = 0; ev < n_out; ++ev) {
        if (cals && events[ev].is_missed) {                      /* -i: a target that met no record (tgt_flush_region, vcfcall.c:408-424) */
            const tgt_t *t = &tgt[events[ev].tgt];
            fprintf(LN, "%s\t%d\t.\t%s\t", t->chrom, t->pos, t->als[0]);
            if (t->nals < 2) fputc('.', LN);
            for (int i = 1; i < t->nals; ++i) fprintf(LN, "%s%s", i > 1 ? "," : "", t->als[i]);
            fputs("\t.\t.\t.\tGT", LN);
            for (int s2 = 0; s2 < S; ++s2) fputs("\t.", LN);
            fputc(0, LN); fflush(LN);
            if (vio_write_line(fout, hdr, ln_buf)) DIE("%s\n", vio_error());
            rewind(LN);
            continue;
        }
        const int k = cals ? events[ev].tgt : ev;
        const rec_t *r = &recs[k];
        const bcfgpu_call_site *c = &cs[k];
        if (c->ret == -2 || (varonly && c->ret == 0) || c->ret < 0) continue;
        const int nn = c->nals_new, ngn = nn * (nn + 1) / 2;
        int gv_min_dp = 0;
        if (gv_n) {
            const int w = gv_w[k], b = gv_blk[w];
            if (b >= 0) {                                        /* inside a block: one line when the block ends (gvcf.c:134-166) */
                const bcfgpu_gvcf_block *B = &gv_block[b];
                if (B->last_site != w) continue;
                int kf = k; while (gv_w[kf] != B->first_site) --kf;  /* the block's first record: alleles and genotypes are its */
                const rec_t *rf = &recs[kf]; const bcfgpu_call_site *cf = &cs[kf];
                fprintf(LN, "%s\t%d\t.\t%s\t", rf->fld[0], B->start_pos + 1, rf->fld[3]);
                {
                    const char *al[5] = { 0, 0, 0, 0, 0 };
                    for (int i = 0; i < rf->nals; ++i) if (cf->als_map[i] >= 0) al[cf->als_map[i]] = rf->als[i];
                    if (cf->nals_new < 2) fputc('.', LN);
                    for (int i = 1; i < cf->nals_new; ++i) fprintf(LN, "%s%s", i > 1 ? "," : "", al[i]);
                }
                fputs("\t.\t.\t", LN);
                if (B->start_pos + 1 < B->end1) fprintf(LN, "END=%d;", B->end1);
                fprintf(LN, "MinDP=%d\tGT:DP", B->min_dp);
                for (int s2 = 0; s2 < S; ++s2) {
                    const int g0 = gt[((size_t)kf * 2 + 0) * S + s2], g1 = gt[((size_t)kf * 2 + 1) * S + s2];
                    fputc('\t', LN);
                    if (g0 == BCFGPU_GT_MISSING) fputc('.', LN); else fprintf(LN, "%d", g0);
                    if (g1 != BCFGPU_GT_VECTOR_END) { fputc('/', LN); if (g1 == BCFGPU_GT_MISSING) fputc('.', LN); else fprintf(LN, "%d", g1); }
                    const int32_t v = gv_dp[(size_t)b * S + s2];
                    if (v == INT32_MIN) fputs(":.", LN); else fprintf(LN, ":%d", v);
                }
                fputc(0, LN); fflush(LN);
                if (vio_write_line(fout, hdr, ln_buf)) DIE("%s\n", vio_error());
                rewind(LN);
                continue;
            }
            if (c->ret == 1) gv_min_dp = gv_min[w];                /* a reference record outside the ranges keeps MinDP (gvcf.c:221-222) */
        }
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
            if (gv_min_dp == INT32_MIN) fputs(";MinDP=.", LN); else if (gv_min_dp) fprintf(LN, ";MinDP=%d", gv_min_dp);
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

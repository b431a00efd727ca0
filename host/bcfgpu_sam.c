/*  bcfgpu_sam.c -- `bcftools mpileup` over SAM / BAM files with every stage of the path on the device, in plain C over the
 *  C-ABI of include/bcfgpu.h (SNP and indel records).
 *
 *      bcfgpu_sam [options] -f ref.fa [-r CHR[:BEG[-END]],... | -R FILE] [-t [^]REG,... | -T [^]FILE] [-b FILE] file.sam|file.bam [...]         (mpileup's own spelling, mpileup.c:952-1003)
 *      bcfgpu_sam [options] ref.fa contig beg end file.sam|file.bam [...]                       (beg, end 1-based inclusive)
 *      options: -a TAG,..  --gvcf INT,..  -O v|z|u|b  -o FILE  -d INT  -s LIST  -S FILE  -G FILE  --ignore-RG
 *               -B  -E  -A  -q INT  -Q INT  -C INT  --ff INT  --rf INT  -I -o INT -e INT -h INT -m INT -F FLOAT -p -L INT   (as `bcftools mpileup`)
 *               --tile COLUMNS  columns per device tile (default 16384);  --gpus N  region shards, a process per shard
 *               --list-samples: print "sample <TAB> reads entering the pileup <TAB> files" and stop (no device needed)
 *
 *  A region is streamed through in TILES: the files are read in step with the tiles -- a position-sorted file no
 *  further than the tile's, and in the end the region's, last column -- a read stays in host memory while it can still cover a
 *  column, and what the device holds at any time is one tile's reads and columns.  Memory is bounded by the tile, not by the
 *  region (mpileup_reg() walks column by column with the iterator's buffer, mpileup.c:327-367); several regions run one after the
 *  other (mpileup.c:652-683).
 *
 *  What stays on the host is what mpileup.c and htslib's pileup do before any arithmetic: parsing (SAM text; BAM = BGZF +
 *  binary records, inflated block by block), the read -> sample map of bam_sample.c (@RG SM, RG:Z tags, -s/-S/-G), the read
 *  filters of mplp_func (mpileup.c:183-246: unmapped, --rf/--ff flags, reads of dropped read groups, -q, orphans), the
 *  iterator's per-file depth cap (bcfgpu_depth_cap_push, its buffer carried from tile to tile) and the pairing of overlapping
 *  mates (htslib overlap_push).  Then per tile, each a call on the flat read pool:
 *      bcfgpu_pool_upload          the tile's reads to HBM, once
 *      bcfgpu_pool_baq             BAQ (sam_prob_realn, mpileup.c:234)
 *      bcfgpu_pool_overlap_tweak   mate-overlap qualities (bam_mplp_init_overlaps, mpileup.c:640)
 *      bcfgpu_pool_pileup          the pileup columns of the tile, built in HBM
 *      bcfgpu_mpileup              bcf_call_glfgen x samples + bcf_call_combine per column (mpileup.c:343-347)
 *  and for the columns where some read is followed by an indel (mpileup.c:354-365):
 *      bcfgpu_gap_prep_tile (bcf_call_gap_prep on the candidate columns, in HBM) -> bcfgpu_mpileup on its indel tile
 *  with -C INT: bcfgpu_pool_baq + bcfgpu_pool_cap_mapq on every batch of reads as it comes off the files (mpileup.c:234-241),
 *  with --gvcf: bcfgpu_gvcf_blocks per tile, the block that reaches a tile's end joined with the next tile's first (gvcf.c:88-226);
 *  and the record loop writes what bcf_call2bcf (bam2bcf.c:756-906) puts in the record, in its order, under mpileup's header
 *  (mpileup.c:510-602), as VCF, bgzipped VCF or BCF (host/vcfio.c).  tests/test_c_host.py compares the whole output with the
 *  reference's goldens test/mpileup/mpileup.{1..11}.out, mpileup-SCR.out, indel-AD.1.out -- one tile and many.
 *  Not here: CRAM input, index files (a region far into a file is reached by reading up to it);
 *  a sample fed by several files has its reads merged by position (the reference appends file after file: same
 *  records unless a cell passes 255 usable reads, where errmod_cal's draw then meets the reads in another order).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <stdint.h>
#include <ctype.h>
#include <math.h>
#include <zlib.h>
#include <spawn.h>
#include <unistd.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <fcntl.h>
#include <signal.h>
#include <pthread.h>
#include <sys/syscall.h>
extern long syscall(long number, ...);       /* (unistd.h keeps it back under -std=c99 -D_POSIX_C_SOURCE) */
#include "bcfgpu.h"
#include "vcfio.h"

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s: %s (%d)\n", #call, bcfgpu_last_error(), rc_); exit(1); } } while (0)
#define DIE(...) do { fprintf(stderr, __VA_ARGS__); exit(1); } while (0)

static FILE *LN; static char *ln_buf; static size_t ln_len;      /* the record being written: a memory stream, framed by vcfio */
static vio_file *fout; static vio_hdr *hdr;
static void end_record(void)
{
    fputc(0, LN); fflush(LN);
    if (vio_write_line(fout, hdr, ln_buf)) { fprintf(stderr, "%s\n", vio_error()); exit(1); }
    rewind(LN);
}

typedef struct {
    int n, cap;                                   /* reads */
    int32_t *pos, *lq, *flag, *ncig, *cig_off, *seq_off, *smpl, *file, *end, *mpos, *isize, *rnext_same;
    uint8_t *mapq, *has_zq;
    char **qname;
    uint32_t *cig; size_t ncigs, cigcap;
    uint8_t *seq16, *qual, *zq; size_t nbase, basecap;
} pool_t;

static void *grow(void *p, size_t n) { p = realloc(p, n ? n : 1); if (!p) DIE("out of memory\n"); return p; }

static int nt16_of(char c)
{
    static const char *codes = "=ACMGRSVTWYHKDBN";
    const char *q = strchr(codes, toupper((unsigned char)c));
    return (q && *q) ? (int)(q - codes) : 15;
}

static char *read_contig(const char *path, const char *name, int *len)
{
    FILE *f = fopen(path, "r");
    if (!f) DIE("cannot open %s\n", path);
    char line[1 << 16], *seq = NULL;
    size_t n = 0, cap = 0;
    int in = 0;
    while (fgets(line, sizeof line, f)) {
        if (line[0] == '>') {
            char *e = line + 1;
            while (*e && !isspace((unsigned char)*e)) ++e;
            *e = 0;
            in = strcmp(line + 1, name) == 0;
            continue;
        }
        if (!in) continue;
        size_t l = strlen(line);
        while (l && isspace((unsigned char)line[l - 1])) --l;
        if (n + l + 1 > cap) { cap = (n + l + 1) * 2; seq = grow(seq, cap); }
        memcpy(seq + n, line, l); n += l;
    }
    fclose(f);
    if (!seq) DIE("contig %s not found in %s\n", name, path);
    seq[n] = 0; *len = (int)n;
    return seq;
}

/* ---- which reads belong to which output sample: bam_sample.c (bam_smpl_add_bam, bam_smpl_get_sample_id, -s/-S/-G) ---- */
typedef struct { char **key, **val; int n; } smap_t;                        /* a small string map; lookups are per @RG line */
static const char *smap_get(const smap_t *m, const char *k) { for (int i = 0; i < m->n; ++i) if (!strcmp(m->key[i], k)) return m->val[i]; return NULL; }
static void smap_set(smap_t *m, const char *k, const char *v)
{
    m->key = grow(m->key, (size_t)(m->n + 1) * sizeof *m->key); m->val = grow(m->val, (size_t)(m->n + 1) * sizeof *m->val);
    m->key[m->n] = strdup(k); m->val[m->n++] = strdup(v);
}
typedef struct { const char *fname; int default_idx, nrg; char **rg; int *rg_smpl; } sfile_t;
static struct {
    int ignore_rg, nsmpl; char **smpl;                                      /* output samples, in order of first appearance */
    int have_samples, sample_logic; smap_t samples;                         /* -s/-S: input sample -> output name; 1 include, 0 exclude */
    int have_rgs, rg_logic; smap_t rgs;                                     /* -G: "id" | "id\tfile" | "*\tfile" -> name or "\t" (keep) */
} SM;

static int rg_find(const sfile_t *f, const char *id) { for (int i = 0; i < f->nrg; ++i) if (!strcmp(f->rg[i], id)) return i; return -1; }
static int smpl_find(const char *name) { for (int i = 0; i < SM.nsmpl; ++i) if (!strcmp(SM.smpl[i], name)) return i; return -1; }

/* bsmpl_add_readgroup: name NULL = the read group is known but its reads are dropped; id "*" = the whole file */
static void rg_add(sfile_t *f, const char *id, const char *name)
{
    int is = -1;
    if (name && (is = smpl_find(name)) < 0) {
        SM.smpl = grow(SM.smpl, (size_t)(SM.nsmpl + 1) * sizeof *SM.smpl);
        SM.smpl[is = SM.nsmpl++] = strdup(name);
    }
    if (!strcmp(id, "*")) { f->default_idx = is; return; }
    if (rg_find(f, id) >= 0) return;                                        /* a repeated @RG ID: the first one counts */
    f->rg = grow(f->rg, (size_t)(f->nrg + 1) * sizeof *f->rg); f->rg_smpl = grow(f->rg_smpl, (size_t)(f->nrg + 1) * sizeof *f->rg_smpl);
    f->rg[f->nrg] = strdup(id); f->rg_smpl[f->nrg++] = is;
}
/* bsmpl_keep_readgroup: the -G list, most specific entry first; may rename the sample */
static int rg_keep(const sfile_t *f, const char *id, const char **name)
{
    char key[4096];
    const char *v = smap_get(&SM.rgs, id);
    if (!v) { snprintf(key, sizeof key, "%s\t%s", id, f->fname); v = smap_get(&SM.rgs, key); }
    if (!v) { snprintf(key, sizeof key, "*\t%s", f->fname); v = smap_get(&SM.rgs, key); }
    if ((!v && SM.rg_logic) || (v && !SM.rg_logic)) return 0;
    if (v && v[0] != '\t') *name = v;
    return 1;
}
/* one whitespace-delimited field with backslash escapes (bam_smpl_add_samples / bam_smpl_add_readgroups) */
static const char *next_field(const char *p, char *out, size_t cap)
{
    size_t n = 0; int esc = 0;
    while (*p && isspace((unsigned char)*p)) ++p;
    for (; *p; ++p) {
        if (*p == '\\' && !esc) { esc = 1; continue; }
        if (isspace((unsigned char)*p) && !esc) break;
        if (n + 1 < cap) out[n++] = *p;
        esc = 0;
    }
    out[n] = 0;
    return p;
}
/* hts_readlist: the rows of a file, or the comma-separated items of the argument */
static char **read_list(const char *arg, int is_file, int *n)
{
    char **rows = NULL; *n = 0;
    char *text = NULL;
    if (is_file) {
        FILE *f = fopen(arg, "r");
        if (!f) DIE("cannot open %s\n", arg);
        size_t len = 0, cap = 0; int c;
        while ((c = fgetc(f)) != EOF) { if (len + 2 > cap) { cap = cap ? 2 * cap : 4096; text = grow(text, cap); } text[len++] = (char)c; }
        fclose(f);
        if (!text) return NULL;
        text[len] = 0;
    } else text = strdup(arg);
    for (char *t = strtok(text, is_file ? "\r\n" : ","); t; t = strtok(NULL, is_file ? "\r\n" : ",")) {
        rows = grow(rows, (size_t)(*n + 1) * sizeof *rows); rows[(*n)++] = strdup(t);
    }
    free(text);
    return rows;
}
static void add_samples(const char *list, int is_file)                     /* -s / -S */
{
    if (list[0] != '^') SM.sample_logic = 1; else ++list;
    int n; char **rows = read_list(list, is_file, &n);
    if (!n) return;
    SM.have_samples = 1;
    for (int i = 0; i < n; ++i) {
        char a[1024], b[1024];
        const char *p = next_field(rows[i], a, sizeof a);
        next_field(p, b, sizeof b);
        if (!smap_get(&SM.samples, a)) smap_set(&SM.samples, a, b[0] ? b : a);
        free(rows[i]);
    }
    free(rows);
}
static void add_readgroups(const char *list)                               /* -G: rows "ID", "ID SAMPLE" or "ID FILE SAMPLE" */
{
    if (list[0] != '^') SM.rg_logic = 1; else ++list;
    int n; char **rows = read_list(list, 1, &n);
    if (!n) return;
    SM.have_rgs = 1;
    for (int i = 0; i < n; ++i) {
        char a[1024], b[1024], c[1024], key[2100];
        const char *p = next_field(rows[i], a, sizeof a);
        p = next_field(p, b, sizeof b);
        next_field(p, c, sizeof c);
        const char *val = c[0] ? c : b[0] ? b : "\t";
        if (c[0]) snprintf(key, sizeof key, "%s\t%s", a, b); else snprintf(key, sizeof key, "%s", a);
        const char *old = smap_get(&SM.rgs, key);
        if (!old) smap_set(&SM.rgs, key, val);
        else if (strcmp(old, val)) DIE("Error: The read group \"%s\" was assigned to two different samples: \"%s\" and \"%s\"\n", key, old, val);
        free(rows[i]);
    }
    free(rows);
}
/* bam_smpl_add_bam over the header text; 0: no read of the file can be used, the file is dropped */
static int add_file(sfile_t *f, const char *fname, const char *hdr_text)
{
    memset(f, 0, sizeof *f);
    f->fname = fname; f->default_idx = -1;
    if (SM.ignore_rg || !hdr_text || !hdr_text[0]) { rg_add(f, "*", fname); return 1; }
    int first_smpl = -1, nskipped = 0, n_file_smpl = 0;
    char **file_smpl = NULL;
    for (const char *line = hdr_text; line && *line; ) {
        const char *eol = strchr(line, '\n');
        const size_t len = eol ? (size_t)(eol - line) : strlen(line);
        if (len > 3 && !strncmp(line, "@RG", 3)) {
            char *l = malloc(len + 1); memcpy(l, line, len); l[len] = 0;
            if (len && l[len - 1] == '\r') l[len - 1] = 0;
            char *id = strstr(l, "\tID:"), *sm = strstr(l, "\tSM:");
            if (!id || !sm) { free(l); break; }                             /* the scan ends at an @RG without ID or SM */
            id += 4; sm += 4;
            id[strcspn(id, "\t")] = 0; sm[strcspn(sm, "\t")] = 0;
            if (!strcmp(id, "*") || !strcmp(id, "?")) DIE("Error: the read group IDs \"*\" and \"?\" have a special meaning: %s\n", fname);
            const char *name = sm;
            int accept = 1;
            if (SM.have_samples) {
                const char *ren = smap_get(&SM.samples, sm);
                if (!SM.sample_logic) accept = ren ? 0 : 1;
                else if (!ren) accept = 0;
                else name = ren;
            }
            if (accept && SM.have_rgs) accept = rg_keep(f, id, &name);
            if (accept) rg_add(f, id, name); else { rg_add(f, id, NULL); ++nskipped; }
            if (first_smpl < 0) first_smpl = smpl_find(name);
            int k; for (k = 0; k < n_file_smpl; ++k) if (!strcmp(file_smpl[k], name)) break;
            if (k == n_file_smpl) { file_smpl = grow(file_smpl, (size_t)(n_file_smpl + 1) * sizeof *file_smpl); file_smpl[n_file_smpl++] = strdup(name); }
            free(l);
        }
        line = eol ? eol + 1 : NULL;
    }
    for (int k = 0; k < n_file_smpl; ++k) free(file_smpl[k]);
    free(file_smpl);
    /* reads without a read group, or with one the header does not list */
    const char *null_name = NULL;
    int accept_null = 1;
    if (SM.have_rgs && !rg_keep(f, "?", &null_name)) accept_null = 0;
    if (SM.have_samples && first_smpl == -1) accept_null = 0;
    if (!accept_null && first_smpl == -1) return 0;
    if (!accept_null) return 1;
    if (n_file_smpl == 1 && !nskipped) { f->default_idx = first_smpl; return 1; }
    if (!null_name) null_name = first_smpl == -1 ? fname : SM.smpl[first_smpl];
    rg_add(f, "?", null_name);
    return 1;
}
static int sample_of(const sfile_t *f, const char *rg)                      /* bam_smpl_get_sample_id; rg NULL: no RG tag */
{
    if (f->default_idx >= 0) return f->default_idx;
    int i = rg_find(f, rg ? rg : "?");
    if (i < 0) i = rg_find(f, "?");
    return i < 0 ? -1 : f->rg_smpl[i];
}

/* ---- reading: SAM text or BAM; the read filters of mplp_func (mpileup.c:183-246) ---- */
static int rflag_require = 0, rflag_filter = 4 | 256 | 512 | 1024, min_mq = 0, keep_orphans = 0;
static int no_overlaps;                                                         /* mpileup -x: the mates' overlaps are left alone (mpileup.c:1005) */
/* -t / -T: the targets (mpileup.c:198-212, 330-335, 1033-1051): reads that overlap none of them never enter the pileup, columns outside
 * them give no record; a leading ^ turns the list into the columns to leave out (the reads are chosen as without it, as in the reference).
 * 0-based inclusive [beg, end]; end < 0: to the end of the sequence */
typedef struct { char *chrom; long beg, end; } target_t;
static target_t *target; static int n_target, target_incl = 1;
static void target_add(const char *chrom, size_t cl, long beg, long end)
{
    target = realloc(target, (size_t)(n_target + 1) * sizeof *target);
    target[n_target].chrom = strndup(chrom, cl); target[n_target].beg = beg; target[n_target].end = end; ++n_target;
}
static void target_spec(const char *spec)                                          /* CHR, CHR:POS, CHR:BEG-END (1-based) */
{
    const char *c = strrchr(spec, ':'); char *e = NULL;
    long a = c ? strtol(c + 1, &e, 10) : 0;
    if (!c || e == c + 1) { target_add(spec, strlen(spec), 0, -1); return; }
    const long b = *e == '-' ? (e[1] ? strtol(e + 1, NULL, 10) : 0) : a;
    target_add(spec, (size_t)(c - spec), a - 1, b ? b - 1 : -1);
}
static void target_file(const char *path)                                          /* CHROM <tab> POS [<tab> END], 1-based inclusive */
{
    FILE *f = fopen(path, "r");
    if (!f) { fprintf(stderr, "Could not read file \"%s\"\n", path); exit(1); }
    char ln[4096], c[1024]; long a, b;
    while (fgets(ln, sizeof ln, f)) {
        if (ln[0] == '#') continue;
        const int k = sscanf(ln, "%1023s %ld %ld", c, &a, &b);
        if (k >= 1) target_add(c, strlen(c), k >= 2 ? a - 1 : 0, k >= 3 ? b - 1 : k == 2 ? a - 1 : -1);
    }
    fclose(f);
}
/* does [beg, end] of `chrom` overlap a target?  *inside: it lies wholly inside one */
static int target_overlap(const char *chrom, long beg, long end, int *inside)
{
    int ov = 0;
    if (inside) *inside = 0;
    for (int i = 0; i < n_target; ++i) {
        if (strcmp(target[i].chrom, chrom)) continue;
        const long tb = target[i].beg, te = target[i].end < 0 ? (1L << 40) : target[i].end;
        if (beg <= te && end >= tb) { ov = 1; if (inside && beg >= tb && end <= te) *inside = 1; }
    }
    return ov;
}
static int target_keeps_read(const char *chrom, long beg, long end)                /* mpileup.c:198-212 */
{
    /* As the reference has it: a read goes on if it overlaps a listed stretch -- with the ^ form too (its "exclude only reads which are
     * fully contained" loop runs only when nothing overlaps, over no region, so a read clear of the list is dropped there as well). */
    return !n_target || target_overlap(chrom, beg, end, NULL);
}
static int target_keeps_column(const char *chrom, long pos)                        /* mpileup.c:330-335 */
{
    if (!n_target) return 1;
    const int ov = target_overlap(chrom, pos, pos, NULL);
    return target_incl ? ov : !ov;
}
static int illumina13;                                                          /* mpileup -6: qualities in the Illumina-1.3+ encoding (mpileup.c:216-221) */
static int defer_mq_filters = 0;
static int reg_beg = 0, reg_end = 0x7fffffff;   /* the region: only reads that overlap it enter the pool, as htslib's region iterator hands them out */      /* -C: sam_cap_mapq comes between the flag filters and the -q / orphan filters (mpileup.c:234-241) */

static void pool_add(pool_t *P, int file, int smpl, const char *qname, int flag, int pos, int mapq, int rnext_same, int mpos, int isize,
                     const uint32_t *cig, int ncig, int lq, const uint8_t *seq16, const uint8_t *qual)
{
    if (P->n == P->cap) {
        P->cap = P->cap ? 2 * P->cap : 1024;
        #define G(a) P->a = grow(P->a, (size_t)P->cap * sizeof *P->a)
        G(pos); G(lq); G(flag); G(ncig); G(cig_off); G(seq_off); G(smpl); G(file); G(end); G(mpos); G(isize); G(rnext_same); G(mapq); G(has_zq); G(qname);
        #undef G
    }
    const int r = P->n++;
    P->qname[r] = (char *)qname;                       /* borrowed: the live window owns the name */
    P->flag[r] = flag; P->pos[r] = pos; P->mapq[r] = (uint8_t)mapq; P->smpl[r] = smpl; P->file[r] = file; P->has_zq[r] = 0;
    P->rnext_same[r] = rnext_same; P->mpos[r] = mpos; P->isize[r] = isize;
    P->cig_off[r] = (int32_t)P->ncigs; P->ncig[r] = ncig;
    if (P->ncigs + ncig + 1 > P->cigcap) { P->cigcap = (P->ncigs + ncig + 1) * 2; P->cig = grow(P->cig, P->cigcap * 4); }
    int x = pos;
    for (int c = 0; c < ncig; ++c) {
        const int op = cig[c] & 15;
        P->cig[P->ncigs++] = cig[c];
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) x += (int)(cig[c] >> 4);
    }
    P->end[r] = x;
    P->lq[r] = lq; P->seq_off[r] = (int32_t)P->nbase;
    if (P->nbase + lq + 1 > P->basecap) {
        P->basecap = (P->nbase + lq + 1) * 2;
        P->seq16 = grow(P->seq16, P->basecap); P->qual = grow(P->qual, P->basecap); P->zq = grow(P->zq, P->basecap);
    }
    memcpy(P->seq16 + P->nbase, seq16, (size_t)lq); memcpy(P->qual + P->nbase, qual, (size_t)lq); memset(P->zq + P->nbase, 0, (size_t)lq);
    P->nbase += lq;
}
static int read_passes(int flag, int mapq)
{
    if (flag & 4) return 0;
    if (rflag_require && !(rflag_require & flag)) return 0;
    if (rflag_filter && (rflag_filter & flag)) return 0;
    if (defer_mq_filters) return 1;
    if (mapq < min_mq) return 0;
    if (!keep_orphans && (flag & 1) && !(flag & 2)) return 0;
    return 1;
}
static void contig_line(vio_hdr *h, const char *name, int nlen, long length)
{
    char buf[1200]; snprintf(buf, sizeof buf, "##contig=<ID=%.*s,length=%ld>", nlen, name, length);
    vio_hdr_append(h, buf);
}

/* ---- streaming input: one reader per file (SAM text, or BAM = BGZF + binary records, SAM spec 4.2), reads handed out in file
 * order.  A read leaves the reader only if it is on the region's contig, overlaps the region, passes the flag filters of
 * mplp_func (mpileup.c:183-246) and belongs to an output sample; a position-sorted file is read no further than the region's
 * end (htslib's region iterator stops there too). ---- */
typedef struct {
    char *qname; int flag, pos, mapq, rnext_same, mpos, isize, ncig, lq, end, smpl;
    uint32_t *cig; uint8_t *seq16, *qual;
} lrec_t;
static void lrec_free(lrec_t *r) { if (r) { free(r->qname); free(r->cig); free(r->seq16); free(r->qual); free(r); } }

typedef struct {
    const char *path; FILE *fp; int is_bam, eof, done, seen_contig;
    z_stream zs; int zs_on; uint8_t *zin;                   /* BAM: the BGZF stream, inflated member after member */
    uint8_t *buf; size_t buf_n, buf_rd, buf_cap;            /* BAM: inflated bytes not yet consumed */
    int tid, n_ref; char **ref_name; int32_t *ref_len;      /* BAM: the reference dictionary; tid = the region's contig */
    char *text;                                             /* the header text ('@' lines) */
    char *line; size_t line_cap; int have_line;             /* SAM: one text line of look-ahead */
    lrec_t *pend;                                           /* the next read parsed, not yet handed on (the parsing side's) */
    /* the consuming side: parsed reads wait in a ring (filled by a parsing thread, below); head = the next read, not yet taken */
    lrec_t *head; int drained;
    lrec_t **q; int q_rd, q_n, q_done;                      /* ring of RQ_CAP reads; q_done: the parsing side is through with the region */
    struct parser *owner;
} reader_t;

static int32_t le32(const uint8_t *p) { return (int32_t)((uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24); }

/* at least `need` inflated bytes at buf + buf_rd; 0 at the end of the file */
static int bz_fill(reader_t *r, size_t need)
{
    if (r->buf_n - r->buf_rd >= need) return 1;
    if (r->buf_rd) { memmove(r->buf, r->buf + r->buf_rd, r->buf_n - r->buf_rd); r->buf_n -= r->buf_rd; r->buf_rd = 0; }
    if (r->buf_cap < need + (1 << 16)) { r->buf_cap = need + (1 << 17); r->buf = grow(r->buf, r->buf_cap); }
    while (r->buf_n < need) {
        if (r->zs.avail_in == 0) {
            if (r->eof) break;
            const size_t got = fread(r->zin, 1, 1 << 16, r->fp);
            if (!got) { r->eof = 1; break; }
            r->zs.next_in = r->zin; r->zs.avail_in = (uInt)got;
        }
        r->zs.next_out = r->buf + r->buf_n; r->zs.avail_out = (uInt)(r->buf_cap - r->buf_n);
        const int rc = inflate(&r->zs, Z_NO_FLUSH);
        r->buf_n = r->buf_cap - r->zs.avail_out;
        if (rc == Z_STREAM_END) { if (inflateReset(&r->zs) != Z_OK) DIE("zlib: inflateReset failed\n"); }     /* the next BGZF block */
        else if (rc != Z_OK && rc != Z_BUF_ERROR) DIE("%s: corrupt BGZF block\n", r->path);
    }
    return r->buf_n - r->buf_rd >= need;
}

static void reader_close(reader_t *r)
{
    if (r->fp) fclose(r->fp);
    if (r->zs_on) inflateEnd(&r->zs);
    free(r->zin); free(r->buf); free(r->text); free(r->line);
    for (int i = 0; i < r->n_ref; ++i) free(r->ref_name[i]);
    free(r->ref_name); free(r->ref_len);
    lrec_free(r->pend);
    const char *p = r->path;
    memset(r, 0, sizeof *r); r->path = p;
}

/* opens the file and reads its header (r->text; BAM: the reference dictionary too) */
static void reader_open(reader_t *r, const char *path)
{
    memset(r, 0, sizeof *r);
    r->path = path; r->tid = -1;
    r->fp = fopen(path, "rb");
    if (!r->fp) DIE("cannot open %s\n", path);
    const int c0 = fgetc(r->fp), c1 = fgetc(r->fp);
    rewind(r->fp);
    if (c0 == 0x1f && c1 == 0x8b) {
        r->is_bam = 1; r->zin = grow(NULL, 1 << 16);
        if (inflateInit2(&r->zs, 15 + 16) != Z_OK) DIE("zlib: inflateInit2 failed\n");
        r->zs_on = 1;
        if (!bz_fill(r, 12) || memcmp(r->buf + r->buf_rd, "BAM\1", 4)) DIE("%s: not a BAM file\n", path);
        const size_t l_text = (uint32_t)le32(r->buf + r->buf_rd + 4);
        if (!bz_fill(r, 12 + l_text)) DIE("%s: truncated BAM header\n", path);
        r->text = grow(NULL, l_text + 1); memcpy(r->text, r->buf + r->buf_rd + 8, l_text); r->text[l_text] = 0;
        r->n_ref = le32(r->buf + r->buf_rd + 8 + l_text);
        r->buf_rd += 12 + l_text;
        r->ref_name = grow(NULL, (size_t)(r->n_ref + 1) * sizeof *r->ref_name); r->ref_len = grow(NULL, (size_t)(r->n_ref + 1) * sizeof *r->ref_len);
        for (int i = 0; i < r->n_ref; ++i) {
            if (!bz_fill(r, 4)) DIE("%s: truncated BAM header\n", path);
            const int l_name = le32(r->buf + r->buf_rd);
            if (l_name < 1 || !bz_fill(r, 8 + (size_t)l_name)) DIE("%s: truncated BAM header\n", path);
            r->ref_name[i] = grow(NULL, (size_t)l_name + 1); memcpy(r->ref_name[i], r->buf + r->buf_rd + 4, (size_t)l_name); r->ref_name[i][l_name] = 0;
            r->ref_len[i] = le32(r->buf + r->buf_rd + 4 + l_name);
            r->buf_rd += 8 + (size_t)l_name;
        }
        return;
    }
    /* SAM text: the '@' lines; the first other line stays as look-ahead */
    size_t tl = 0, tcap = 0;
    ssize_t n;
    while ((n = getline(&r->line, &r->line_cap, r->fp)) > 0) {
        if (r->line[0] != '@') { r->have_line = 1; break; }
        if (tl + (size_t)n + 1 > tcap) { tcap = (tl + (size_t)n + 1) * 2; r->text = grow(r->text, tcap); }
        memcpy(r->text + tl, r->line, (size_t)n); tl += (size_t)n;
    }
    if (!r->text) r->text = grow(NULL, 1);
    r->text[tl] = 0;
}

/* the value of the RG:Z tag in a BAM record's auxiliary data (SAM spec 4.2.4), or NULL */
static const char *bam_aux_rg(const uint8_t *a, const uint8_t *end)
{
    while (a + 3 <= end) {
        const int is_rg = a[0] == 'R' && a[1] == 'G';
        const char t = (char)a[2];
        a += 3;
        size_t l;
        switch (t) {
        case 'A': case 'c': case 'C': l = 1; break;
        case 's': case 'S': l = 2; break;
        case 'i': case 'I': case 'f': l = 4; break;
        case 'Z': case 'H': if (is_rg && t == 'Z') return (const char *)a; l = strnlen((const char *)a, (size_t)(end - a)) + 1; break;
        case 'B': {
            if (a + 5 > end) return NULL;
            const char st = (char)a[0]; const size_t cnt = (uint32_t)le32(a + 1);
            l = 5 + cnt * (st == 'c' || st == 'C' ? 1 : st == 's' || st == 'S' ? 2 : 4); break;
        }
        default: return NULL;
        }
        a += l;
    }
    return NULL;
}

static int ref_span_end(int pos, const uint32_t *cig, int ncig)
{
    int e = pos;
    for (int c = 0; c < ncig; ++c) { const int op = cig[c] & 15; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) e += (int)(cig[c] >> 4); }
    return e;
}
/* [pos, endpos) against [lo, hi) (a read without a reference span counts as one base: bam_endpos) */
static int overlaps(int pos, int end, int lo, int hi) { if (end == pos) end = pos + 1; return pos < hi && end > lo; }

static lrec_t *lrec_new(const char *qname, int flag, int pos, int mapq, int rnext_same, int mpos, int isize, const uint32_t *cig, int ncig, int lq, int smpl)
{
    lrec_t *x = grow(NULL, sizeof *x);
    x->qname = strdup(qname); x->flag = flag; x->pos = pos; x->mapq = mapq; x->rnext_same = rnext_same; x->mpos = mpos; x->isize = isize;
    x->ncig = ncig; x->lq = lq; x->smpl = smpl;
    x->cig = grow(NULL, (size_t)(ncig + 1) * 4); memcpy(x->cig, cig, (size_t)ncig * 4);
    x->end = ref_span_end(pos, cig, ncig);
    x->seq16 = grow(NULL, (size_t)lq + 1); x->qual = grow(NULL, (size_t)lq + 1);
    return x;
}

/* the next read of the file that can enter the pileup of region [reg_beg, reg_end) of `contig`, into r->pend; r->done at the end */
static void reader_parse(reader_t *r, const char *contig, const sfile_t *sf)
{
    if (r->pend || r->done) return;
    uint32_t cig[4096];
    if (r->is_bam) {
        if (r->tid < 0) {
            for (int i = 0; i < r->n_ref; ++i) if (!strcmp(r->ref_name[i], contig)) r->tid = i;
            if (r->tid < 0) { r->done = 1; return; }
        }
        for (;;) {
            if (!bz_fill(r, 4)) { r->done = 1; return; }
            const size_t bs = (uint32_t)le32(r->buf + r->buf_rd);
            if (bs < 32 || !bz_fill(r, 4 + bs)) DIE("%s: truncated BAM record\n", r->path);
            const uint8_t *b = r->buf + r->buf_rd;
            r->buf_rd += 4 + bs;
            const int refid = le32(b + 4), pos = le32(b + 8), l_name = b[12], mapq = b[13];
            const int n_cig = b[16] | b[17] << 8, flag = b[18] | b[19] << 8, l_seq = le32(b + 20);
            const int next_ref = le32(b + 24), next_pos = le32(b + 28), tlen = le32(b + 32);
            if (refid == r->tid) r->seen_contig = 1;
            /* a position-sorted file: nothing of the region follows a read past its end, or the reads of the next contig */
            if (r->seen_contig && (refid != r->tid || pos >= reg_end)) { r->done = 1; return; }
            if (refid != r->tid || !read_passes(flag, mapq)) continue;
            if (n_cig > 4096) DIE("%s: CIGAR with more than 4096 operations\n", r->path);
            const uint8_t *cg = b + 36 + l_name, *sq = cg + 4 * (size_t)n_cig, *ql = sq + (l_seq + 1) / 2, *aux = ql + l_seq;
            if (aux > b + 4 + bs) DIE("%s: malformed BAM record\n", r->path);
            for (int c = 0; c < n_cig; ++c) cig[c] = (uint32_t)le32(cg + 4 * c);
            if (!overlaps(pos, ref_span_end(pos, cig, n_cig), reg_beg, reg_end)) continue;
            if (!target_keeps_read(contig, pos, ref_span_end(pos, cig, n_cig) - 1)) continue;
            const int smpl = sample_of(sf, bam_aux_rg(aux, b + 4 + bs));
            if (smpl < 0) continue;
            lrec_t *x = lrec_new((const char *)b + 36, flag, pos, mapq, next_ref == refid, next_pos, tlen, cig, n_cig, l_seq, smpl);
            for (int i = 0; i < l_seq; ++i) { x->seq16[i] = (sq[i >> 1] >> ((~i & 1) << 2)) & 15; x->qual[i] = ql[i]; }
            if (illumina13) for (int i = 0; i < l_seq; ++i) x->qual[i] = x->qual[i] > 31 ? (uint8_t)(x->qual[i] - 31) : 0;
            r->pend = x;
            return;
        }
    }
    for (;;) {
        if (!r->have_line) {
            if (getline(&r->line, &r->line_cap, r->fp) <= 0) { r->done = 1; return; }
        }
        r->have_line = 0;
        char *line = r->line;
        { size_t l = strlen(line); while (l && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0; }
        char *fld[12]; int nf = 0; char *rest = NULL;
        for (char *s = line; nf < 11 && s; ) { fld[nf++] = s; s = strchr(s, '\t'); if (s) *s++ = 0; rest = s; }
        if (nf < 11) continue;
        const int flag = atoi(fld[1]), mapq = atoi(fld[4]), pos = atoi(fld[3]) - 1;
        const int on = !strcmp(fld[2], contig);
        if (on) r->seen_contig = 1;
        if (r->seen_contig && (!on || pos >= reg_end) && !(flag & 4)) { r->done = 1; return; }
        if (!on || !read_passes(flag, mapq)) continue;
        int ncig = 0;
        for (const char *c = fld[5]; *c && *c != '*'; ) {
            char *e; const long l = strtol(c, &e, 10);
            const char *ops = "MIDNSHP=X", *o = *e ? strchr(ops, *e) : NULL;
            if (!o || ncig == 4096) DIE("bad CIGAR in %s\n", r->path);
            cig[ncig++] = (uint32_t)l << 4 | (uint32_t)(o - ops);
            c = e + 1;
        }
        if (!overlaps(pos, ref_span_end(pos, cig, ncig), reg_beg, reg_end)) continue;
        if (!target_keeps_read(contig, pos, ref_span_end(pos, cig, ncig) - 1)) continue;
        const char *rg = NULL;
        for (char *t = rest; t && *t; ) {                                    /* the optional fields: RG:Z:<id> */
            char *e = strchr(t, '\t'); if (e) *e = 0;
            if (!strncmp(t, "RG:Z:", 5)) { rg = t + 5; break; }
            t = e ? e + 1 : NULL;
        }
        const int smpl = sample_of(sf, rg);
        if (smpl < 0) continue;
        const int lq = fld[9][0] == '*' ? 0 : (int)strlen(fld[9]);
        lrec_t *x = lrec_new(fld[0], flag, pos, mapq, !strcmp(fld[6], "=") || !strcmp(fld[6], fld[2]), atoi(fld[7]) - 1, atoi(fld[8]), cig, ncig, lq, smpl);
        const int noq = fld[10][0] == '*' && !fld[10][1];
        for (int i = 0; i < lq; ++i) { x->seq16[i] = (uint8_t)nt16_of(fld[9][i]); x->qual[i] = noq ? 0xff : (uint8_t)(fld[10][i] - 33); }
        if (illumina13) for (int i = 0; i < lq; ++i) x->qual[i] = x->qual[i] > 31 ? (uint8_t)(x->qual[i] - 31) : 0;
        r->pend = x;
        return;
    }
}

/* ---- parsing beside the device: up to N_PARSERS threads, thread t owning the readers f = t, t + N, ...  A thread parses the
 * next read of each of its files in turn into that file's ring and sleeps when all its rings are full; the main thread takes the
 * reads from the rings in file order, as it took them from the files.  (Parsing SAM text / inflating BAM was a quarter of a run's
 * wall time and waited for the device and the record writer: profiles/r4_sam_driver_probe.txt.)  One region at a time: the
 * threads are started when the region's files are open and joined before they are closed. ---- */
static int RQ_CAP = 8192;      /* reads a file's ring holds: 8192, fewer with many files (about a million reads waiting in all) */
#define N_PARSERS 8
struct parser {
    pthread_t th; pthread_mutex_t mu; pthread_cond_t data, space;
    reader_t *rdr; const sfile_t *sf; int first, step, n_files; const char *contig; int stop;
};
static void *parser_main(void *arg)
{
    struct parser *p = arg;
    for (;;) {
        int progress = 0, live = 0;
        for (int f = p->first; f < p->n_files; f += p->step) {
            reader_t *r = &p->rdr[f];
            if (r->q_done) continue;
            ++live;
            pthread_mutex_lock(&p->mu);
            const int room = RQ_CAP - r->q_n, stop = p->stop;
            pthread_mutex_unlock(&p->mu);
            if (stop) return NULL;
            if (!room) continue;
            int got = 0; lrec_t *batch[64];
            while (got < 64 && got < room) {                       /* parsed outside the lock, handed over in batches */
                reader_parse(r, p->contig, &p->sf[f]);
                if (!r->pend) break;
                batch[got++] = r->pend; r->pend = NULL;
            }
            pthread_mutex_lock(&p->mu);
            for (int i = 0; i < got; ++i) { r->q[(r->q_rd + r->q_n) % RQ_CAP] = batch[i]; ++r->q_n; }
            if (got < 64 && got < room) r->q_done = 1;              /* the file has nothing more for this region */
            pthread_cond_broadcast(&p->data);
            pthread_mutex_unlock(&p->mu);
            progress += got;
        }
        if (!live) return NULL;
        if (!progress) {                                            /* every ring of this thread is full: wait for the consumer */
            pthread_mutex_lock(&p->mu);
            int full = 1;
            for (int f = p->first; f < p->n_files && full; f += p->step) if (!p->rdr[f].q_done && p->rdr[f].q_n < RQ_CAP) full = 0;
            if (full && !p->stop) pthread_cond_wait(&p->space, &p->mu);
            const int stop = p->stop;
            pthread_mutex_unlock(&p->mu);
            if (stop) return NULL;
        }
    }
}
static struct parser parsers[N_PARSERS]; static int n_parsers;
static void parsers_start(reader_t *rdr, const sfile_t *sf, int F, const char *contig)
{
    n_parsers = F < N_PARSERS ? F : N_PARSERS;
    RQ_CAP = (1 << 20) / (F > 0 ? F : 1); if (RQ_CAP > 8192) RQ_CAP = 8192; if (RQ_CAP < 64) RQ_CAP = 64;
    for (int f = 0; f < F; ++f) { rdr[f].q = grow(NULL, (size_t)RQ_CAP * sizeof *rdr[f].q); rdr[f].q_rd = rdr[f].q_n = rdr[f].q_done = 0; rdr[f].head = NULL; rdr[f].drained = 0; rdr[f].owner = &parsers[f % n_parsers]; }
    for (int t = 0; t < n_parsers; ++t) {
        struct parser *p = &parsers[t];
        memset(p, 0, sizeof *p);
        pthread_mutex_init(&p->mu, NULL); pthread_cond_init(&p->data, NULL); pthread_cond_init(&p->space, NULL);
        p->rdr = rdr; p->sf = sf; p->first = t; p->step = n_parsers; p->n_files = F; p->contig = contig;
        if (pthread_create(&p->th, NULL, parser_main, p)) DIE("cannot start a parsing thread\n");
    }
}
static void parsers_stop(reader_t *rdr, int F)
{
    for (int t = 0; t < n_parsers; ++t) {
        struct parser *p = &parsers[t];
        pthread_mutex_lock(&p->mu); p->stop = 1; pthread_cond_broadcast(&p->space); pthread_mutex_unlock(&p->mu);
        pthread_join(p->th, NULL);
        pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->data); pthread_cond_destroy(&p->space);
    }
    for (int f = 0; f < F; ++f) {
        for (int i = 0; i < rdr[f].q_n; ++i) lrec_free(rdr[f].q[(rdr[f].q_rd + i) % RQ_CAP]);
        free(rdr[f].q); rdr[f].q = NULL; rdr[f].q_n = 0;
        lrec_free(rdr[f].head); rdr[f].head = NULL;
    }
    n_parsers = 0;
}
/* the next read of the file into r->head (NULL when the file has no more for the region) */
static void reader_fetch(reader_t *r)
{
    if (r->head || r->drained) return;
    struct parser *p = r->owner;
    pthread_mutex_lock(&p->mu);
    while (!r->q_n && !r->q_done) pthread_cond_wait(&p->data, &p->mu);
    if (r->q_n) { r->head = r->q[r->q_rd]; r->q_rd = (r->q_rd + 1) % RQ_CAP; if (r->q_n-- == RQ_CAP) pthread_cond_signal(&p->space); }
    else r->drained = 1;
    pthread_mutex_unlock(&p->mu);
}

/* the ##contig lines of the VCF header from the first file's dictionary (mpileup.c:533-540) */
static void header_contigs(const reader_t *r, vio_hdr *h)
{
    if (r->is_bam) { for (int i = 0; i < r->n_ref; ++i) contig_line(h, r->ref_name[i], (int)strlen(r->ref_name[i]), r->ref_len[i]); return; }
    for (const char *l = r->text; l && *l; ) {
        if (!strncmp(l, "@SQ\t", 4)) {
            const char *e = strchr(l, '\n'), *sn = strstr(l, "\tSN:"), *ln = strstr(l, "\tLN:");
            if (sn && ln && (!e || (sn < e && ln < e))) { sn += 4; contig_line(h, sn, (int)strcspn(sn, "\t\r\n"), atol(ln + 4)); }
        }
        l = strchr(l, '\n'); if (l) ++l;
    }
}

/* overlap_push (htslib sam.c) over the reads of one sample in file order: which pairs tweak_overlap_quality sees */
static int find_pairs(const pool_t *P, int r0, int r1, int32_t *pa, int32_t *pb)
{
    int np = 0, nb = 1;
    while (nb < 2 * (r1 - r0) + 1) nb <<= 1;
    int32_t *tab = malloc((size_t)nb * sizeof *tab);                         /* open addressing on the read name */
    uint8_t *paired = calloc((size_t)(r1 - r0) + 1, 1);
    for (int i = 0; i < nb; ++i) tab[i] = -1;
    for (int r = r0; r < r1; ++r) {
        const int f = P->flag[r];
        if ((f & 8) || !(f & 2)) continue;
        if (!P->rnext_same[r] || (abs(P->isize[r]) >= 2 * P->lq[r] && P->mpos[r] >= P->end[r])) continue;
        uint32_t h = 2166136261u;
        for (const char *c = P->qname[r]; *c; ++c) h = (h ^ (uint8_t)*c) * 16777619u;
        int slot = (int)(h & (uint32_t)(nb - 1));
        while (tab[slot] >= 0 && strcmp(P->qname[tab[slot]], P->qname[r])) slot = (slot + 1) & (nb - 1);
        if (tab[slot] < 0) {
            if (P->mpos[r] >= P->pos[r] || ((f & 1) && P->mpos[r] == -1)) tab[slot] = r;
        } else if (!paired[tab[slot] - r0]) {
            const int a = tab[slot];
            paired[a - r0] = 1;                    /* the slot stays occupied (probe chains); the name is done */
            if (P->end[a] > P->pos[r]) { pa[np] = a; pb[np] = r; ++np; }
        }
    }
    free(tab); free(paired);
    return np;
}

#define INSCNS_CAP 256

/* what bcf_call2bcf writes into a record (bam2bcf.c:756-906); alleles: the ready REF\tALT text */
static int fmt_flag = BCFGPU_INFO_VDB | BCFGPU_INFO_RPB;                     /* mpileup's default annotations + -a */


typedef struct { const uint8_t *pl, *sp; const uint16_t *dp4, *adf, *adr, *scr; const int32_t *qs; } planes_t;        /* host copies of bcfgpu_mplp_out's planes */

static void planes_free(planes_t *p) { free((void *)p->pl); free((void *)p->sp); free((void *)p->dp4); free((void *)p->adf); free((void *)p->adr); free((void *)p->scr); free((void *)p->qs); memset(p, 0, sizeof *p); }

static void put_counts(const char *lead, const int32_t *f, const int32_t *r, int n)
{
    fputs(lead, LN);
    for (int j = 0; j < n; ++j) fprintf(LN, "%s%d", j ? "," : "", (f ? f[j] : 0) + (r ? r[j] : 0));
}

/* what bcf_call2bcf writes into a record, in its order (bam2bcf.c:756-906); alleles: the ready REF\tALT text */
static void print_record(const char *contig, int pos1, const char *alleles, const char *prefix, const bcfgpu_site *c,
                         const planes_t *pp, size_t k, int S)
{
    const int na = c->n_alleles;
    fprintf(LN, "%s\t%d\t.\t%s\t0\t.\t%sDP=%u", contig, pos1, alleles, prefix, c->ori_depth);
    if (fmt_flag & BCFGPU_INFO_ADF) put_counts(";ADF=", c->adf_tot, NULL, na);
    if (fmt_flag & BCFGPU_INFO_ADR) put_counts(";ADR=", NULL, c->adr_tot, na);
    if (fmt_flag & BCFGPU_INFO_AD)  put_counts(";AD=", c->adf_tot, c->adr_tot, na);
    if (fmt_flag & BCFGPU_INFO_DPR) put_counts(";DPR=", c->adf_tot, c->adr_tot, na);
    if (fmt_flag & BCFGPU_INFO_SCR) fprintf(LN, ";SCR=%d", c->scr_tot);
    fputs(";I16=", LN);
    for (int j = 0; j < 16; ++j) {
        /* %g of a whole number below a million is its digits: most of the sixteen sums are (counts, sums of qualities) */
        const double v = (double)(float)c->anno[j];
        if (j) fputc(',', LN);
        if (v >= 0 && v < 1e6 && v == (double)(long)v && !(v == 0 && signbit(v))) {
            char t[8]; int n = 0; long u = (long)v;
            do { t[n++] = (char)('0' + u % 10); u /= 10; } while (u);
            while (n) fputc(t[--n], LN);
        } else fprintf(LN, "%g", v);
    }
    fputs(";QS=", LN);
    for (int j = 0; j < na; ++j) fprintf(LN, "%s%g", j ? "," : "", (double)c->qsum[j]);
    /* the bias statistics: HUGE_VAL = the tag is left out (bam2bcf.c:835-840) */
    {
        const char *tag[6] = { "VDB", "SGB", "RPB", "MQB", "MQSB", "BQB" };
        const float val[6] = { c->vdb, c->seg_bias, c->mwu_pos, c->mwu_mq, c->mwu_mqs, c->mwu_bq };
        for (int j = 0; j < 6; ++j) if (val[j] != HUGE_VALF) fprintf(LN, ";%s=%g", tag[j], (double)val[j]);
    }
    fprintf(LN, ";MQ0F=%g", c->ori_depth ? (double)((float)c->mq0 / (float)c->ori_depth) : 0.);
    fputs("\tPL", LN);
    if (fmt_flag & BCFGPU_FMT_DP) fputs(":DP", LN);
    if (fmt_flag & BCFGPU_FMT_DV) fputs(":DV", LN);
    if (fmt_flag & BCFGPU_FMT_SP) fputs(":SP", LN);
    if (fmt_flag & BCFGPU_FMT_DP4) fputs(":DP4", LN);
    if (fmt_flag & BCFGPU_FMT_ADF) fputs(":ADF", LN);
    if (fmt_flag & BCFGPU_FMT_ADR) fputs(":ADR", LN);
    if (fmt_flag & BCFGPU_FMT_AD) fputs(":AD", LN);
    if (fmt_flag & BCFGPU_FMT_DPR) fputs(":DPR", LN);
    if (fmt_flag & BCFGPU_FMT_SCR) fputs(":SCR", LN);
    if (fmt_flag & BCFGPU_FMT_QS) fputs(":QS", LN);
    const int x = na * (na + 1) / 2;
    const size_t Ss = (size_t)S;
    /* the samples' columns go to the writer as integer arrays, one per FORMAT key in the order of the keys above (a printf call
     * per number, and a parse of every number on the way into a BCF record, were three fifths of a run's wall time at 40 samples) */
    static int32_t *col[12]; static size_t colcap[12];
    int width[12], nk = 0;
    #define COL(w_) (width[nk] = (w_), (colcap[nk] < (size_t)S * (size_t)(w_) ? (colcap[nk] = (size_t)S * (size_t)(w_), col[nk] = grow(col[nk], colcap[nk] * 4)) : col[nk]), col[nk++])
    { int32_t *a = COL(x); for (int s = 0; s < S; ++s) for (int j = 0; j < x; ++j) a[(size_t)s * x + j] = pp->pl[(k * BCFGPU_MAX_PL + j) * Ss + s]; }
    const uint16_t *d = pp->dp4 + k * 4 * Ss;                                /* FORMAT/DP, DV, DP4 from DP4 (bam2bcf.c:851-886) */
    if (fmt_flag & BCFGPU_FMT_DP) { int32_t *a = COL(1); for (int s = 0; s < S; ++s) a[s] = d[s] + d[Ss + s] + d[2 * Ss + s] + d[3 * Ss + s]; }
    if (fmt_flag & BCFGPU_FMT_DV) { int32_t *a = COL(1); for (int s = 0; s < S; ++s) a[s] = d[2 * Ss + s] + d[3 * Ss + s]; }
    if (fmt_flag & BCFGPU_FMT_SP) { int32_t *a = COL(1); for (int s = 0; s < S; ++s) a[s] = pp->sp[k * Ss + s]; }
    if (fmt_flag & BCFGPU_FMT_DP4) { int32_t *a = COL(4); for (int s = 0; s < S; ++s) for (int j = 0; j < 4; ++j) a[(size_t)s * 4 + j] = d[(size_t)j * Ss + s]; }
    for (int which = 0; which < 4; ++which) {                                /* ADF, ADR, AD, DPR */
        static const int bit[4] = { BCFGPU_FMT_ADF, BCFGPU_FMT_ADR, BCFGPU_FMT_AD, BCFGPU_FMT_DPR };
        if (!(fmt_flag & bit[which])) continue;
        int32_t *a = COL(na);
        for (int s = 0; s < S; ++s)
            for (int j = 0; j < na; ++j) {
                const int f = pp->adf[(k * 5 + j) * Ss + s], r = pp->adr[(k * 5 + j) * Ss + s];
                a[(size_t)s * na + j] = which == 0 ? f : which == 1 ? r : f + r;
            }
    }
    if (fmt_flag & BCFGPU_FMT_SCR) { int32_t *a = COL(1); for (int s = 0; s < S; ++s) a[s] = pp->scr[k * Ss + s]; }
    if (fmt_flag & BCFGPU_FMT_QS) { int32_t *a = COL(na); for (int s = 0; s < S; ++s) for (int j = 0; j < na; ++j) a[(size_t)s * na + j] = pp->qs[(k * 5 + j) * Ss + s]; }
    #undef COL
    fputc(0, LN); fflush(LN);
    if (vio_write_record_int(fout, hdr, ln_buf, nk, width, (const int32_t *const *)col)) { fprintf(stderr, "%s\n", vio_error()); exit(1); }
    rewind(LN);
}

/* bcfgpu_mpileup over a tile; the site records and the planes come back to the host.  keep_*: the device copies of the
 * site records / PL / DP4 stay allocated for the caller (--gvcf works on them), else they are freed. */
static void run_mpileup(bcfgpu_ctx *ctx, const bcfgpu_tile *tile, int n, bcfgpu_site **site, planes_t *pp, void **keep_site, void **keep_pl, void **keep_dp4)
{
    void *d[8]; uint8_t *h[8];
    bcfgpu_mplp_out mo; memset(&mo, 0, sizeof mo);
    for (int w = 0; w < 8; ++w) {
        const size_t nb = bcfgpu_mplp_out_bytes(ctx, n, w);
        CHECK(bcfgpu_malloc(ctx, nb, &d[w]));
        if (w) CHECK(bcfgpu_memset(ctx, d[w], 0, nb));
    }
    mo.site = d[0]; mo.pl = d[1]; mo.dp4 = d[2]; mo.adf = d[3]; mo.adr = d[4]; mo.qs = d[5]; mo.scr = d[6]; mo.sp = d[7];
    CHECK(bcfgpu_mpileup(ctx, tile, &mo));
    CHECK(bcfgpu_sync(ctx));
    for (int w = 0; w < 8; ++w) {
        const size_t nb = bcfgpu_mplp_out_bytes(ctx, n, w);
        h[w] = malloc(nb ? nb : 1);
        CHECK(bcfgpu_memcpy_d2h(ctx, h[w], d[w], nb));
    }
    CHECK(bcfgpu_sync(ctx));
    *site = (bcfgpu_site *)h[0];
    pp->pl = h[1]; pp->dp4 = (const uint16_t *)h[2]; pp->adf = (const uint16_t *)h[3]; pp->adr = (const uint16_t *)h[4]; pp->qs = (const int32_t *)h[5];
    pp->scr = (const uint16_t *)h[6]; pp->sp = h[7];
    for (int w = 3; w < 8; ++w) bcfgpu_free(ctx, d[w]);
    if (keep_site) { *keep_site = d[0]; *keep_pl = d[1]; *keep_dp4 = d[2]; }
    else { bcfgpu_free(ctx, d[0]); bcfgpu_free(ctx, d[1]); bcfgpu_free(ctx, d[2]); }
}

/* ---- options (file scope: the tile loop and its helpers read them) ---- */
static int32_t gv_range[16]; static int gv_n = 0;                             /* mpileup --gvcf INT,.. (gvcf.c:44-67) */
static int max_depth = 250, baq_flag = 3, min_baseQ = 13, cap_thres = 0, no_indels = 0, max_indel_depth = 250;
static int openQ = 40, extQ = 20, tandemQ = 100, min_support = 1, per_sample_flt = 0; static double min_frac = 0.002;   /* mpileup.c:937-950 */
static int device = 0, tile_cols = 16384;

typedef struct { char *contig; int beg, end, open; } region_t;                /* 0-based [beg, end); open: to wherever the reads end (no END given) */
#define OPEN_END 0x3fffffff

/* ---- the device context, re-created when a tile needs more room than the last one had ---- */
static bcfgpu_ctx *ctx; static int ctx_S, ctx_sites; static uint64_t ctx_reads; static unsigned long long n_wide_cells;
static void ensure_ctx(int S, int n_sites, uint64_t n_reads)
{
    if (ctx && ctx_S == S && n_sites <= ctx_sites && n_reads <= ctx_reads) return;
    uint64_t rng = 0; int have_rng = 0;                                       /* errmod_cal's generator is the process's: it moves to the new context */
    if (ctx) { uint32_t nw = 0; CHECK(bcfgpu_truncated_cells(ctx, &nw)); n_wide_cells += nw; rng = bcfgpu_errmod_state(ctx); have_rng = 1; bcfgpu_destroy(ctx); ctx = NULL; }
    bcfgpu_cfg cfg; memset(&cfg, 0, sizeof cfg);
    ctx_S = S; ctx_sites = n_sites > ctx_sites ? n_sites : ctx_sites; ctx_reads = n_reads + n_reads / 2 + 4096;
    cfg.device = device; cfg.n_smpl = S; cfg.max_sites = ctx_sites; cfg.max_reads = ctx_reads;   /* every base is in <= 1 column */
    cfg.min_baseQ = min_baseQ; cfg.capQ = 60; cfg.errmod_theta = 0.; cfg.fmt_flag = fmt_flag;
    cfg.call_theta = 1.1e-3; cfg.n_grp = 1; cfg.ploidy_max = 2;
    CHECK(bcfgpu_create(&cfg, &ctx));
    if (have_rng) CHECK(bcfgpu_errmod_seed(ctx, rng));
}

/* ---- gVCF blocks across tiles (and adjacent regions): the reference's gvcf_write keeps a block open over any distance
 * (gvcf.c:88-226).  bcfgpu_gvcf_blocks ends every block with its call; the block that reaches a tile's last column is held
 * back here and joined with the first block of the next tile when gvcf_write would have gone on: the same sequence, the
 * next position, the same depth range (gvcf.c:130-131).  Per sample the smallest DP and the smallest (PL[1], PL[2]) pair in
 * that order; PL[0], REF and INFO/QS stay the first record's (gvcf.c:160-210). ---- */
static struct { int on, S; char *contig; int start_pos, end1, min_dp, range; char ref; float qs0, qs1; int32_t *dp; uint8_t *pl; } PB;
static void block_line(const char *contig, int start_pos, int end1, int min_dp, char refc, float qs0, float qs1, const uint8_t *pl, const int32_t *dp, int S)
{
    fprintf(LN, "%s\t%d\t.\t%c\t<*>\t.\t.\t", contig, start_pos + 1, refc);
    if (start_pos + 1 < end1) fprintf(LN, "END=%d;", end1);                   /* gvcf.c:150-151 */
    fprintf(LN, "MinDP=%d;QS=%g,%g\tPL:DP", min_dp, (double)qs0, (double)qs1);
    for (int s = 0; s < S; ++s) fprintf(LN, "\t%d,%d,%d:%d", pl[s], pl[(size_t)S + s], pl[2 * (size_t)S + s], dp[s]);
    end_record();
}
static void pending_flush(void)
{
    if (!PB.on) return;
    block_line(PB.contig, PB.start_pos, PB.end1, PB.min_dp, PB.ref, PB.qs0, PB.qs1, PB.pl, PB.dp, PB.S);
    PB.on = 0;
}
static int pending_joins(const char *contig, const bcfgpu_gvcf_block *B) { return PB.on && !strcmp(PB.contig, contig) && B->start_pos == PB.end1 && B->range == PB.range; }
static void pending_merge(const bcfgpu_gvcf_block *B, const uint8_t *pl, const int32_t *dp)
{
    const int S = PB.S;
    PB.end1 = B->end1; if (B->min_dp < PB.min_dp) PB.min_dp = B->min_dp;
    for (int s = 0; s < S; ++s) {
        if (dp[s] < PB.dp[s]) PB.dp[s] = dp[s];
        const uint8_t a = pl[(size_t)S + s], c = pl[2 * (size_t)S + s];
        if (a < PB.pl[(size_t)S + s] || (a == PB.pl[(size_t)S + s] && c < PB.pl[2 * (size_t)S + s])) { PB.pl[(size_t)S + s] = a; PB.pl[2 * (size_t)S + s] = c; }
    }
}
static void pending_set(const char *contig, const bcfgpu_gvcf_block *B, char refc, float qs0, float qs1, const uint8_t *pl, const int32_t *dp, int S)
{
    if (!PB.dp || PB.S != S) { PB.dp = grow(PB.dp, (size_t)S * 4); PB.pl = grow(PB.pl, (size_t)S * 3); PB.S = S; }
    free(PB.contig); PB.contig = strdup(contig);
    PB.on = 1; PB.start_pos = B->start_pos; PB.end1 = B->end1; PB.min_dp = B->min_dp; PB.range = B->range; PB.ref = refc; PB.qs0 = qs0; PB.qs1 = qs1;
    memcpy(PB.dp, dp, (size_t)S * 4); memcpy(PB.pl, pl, (size_t)S * 3);
}

static unsigned long long tot_entries, tot_pairs;
/* --timing: where the wall time of a run goes (seconds): reading and parsing the files, building a tile's pool, the device
 * stages of a tile (every call up to the records' planes on the host), writing the records */
#include <time.h>
static int want_timing; static double t_read, t_pool, t_dev, t_emit;
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
/* ---- a tile's records, written beside the device stages of the next tile: process_tile() hands everything the record loop reads over
 * as a job (host copies only: sites, planes, the indel columns' results, the gVCF blocks) and goes on; one worker thread takes the
 * jobs in order.  The pending gVCF block (PB) and the output are the worker's while it runs: whoever else needs them waits for it
 * (emit_wait). ---- */
typedef struct {
    int n_sites, t0, t1, S, nlive, open_block;
    const char *contig, *ref;
    int32_t *col_n; uint8_t *col_indel; int32_t *cand, *live;
    bcfgpu_site *site, *isite; planes_t snp_planes, ind_planes;
    int32_t *g_types, *g_maxins, *g_indelreg, *g_support; float *g_frac; int8_t *g_inscns;
    int32_t *gv_blk, *gv_dp; bcfgpu_gvcf_block *gv_block; uint8_t *gv_pl;
} emit_job_t;
static void emit_tile(emit_job_t *J)
{
    const int n_sites = J->n_sites, t0 = J->t0, t1 = J->t1, S = J->S, nlive = J->nlive, open_block = J->open_block;
    const char *contig = J->contig, *ref = J->ref;
    int32_t *col_n = J->col_n; uint8_t *col_indel = J->col_indel; int32_t *cand = J->cand, *live = J->live;
    bcfgpu_site *site = J->site, *isite = J->isite; planes_t snp_planes = J->snp_planes, ind_planes = J->ind_planes;
    int32_t *g_types = J->g_types, *g_maxins = J->g_maxins, *g_indelreg = J->g_indelreg, *g_support = J->g_support; float *g_frac = J->g_frac; int8_t *g_inscns = J->g_inscns;
    int32_t *gv_blk = J->gv_blk, *gv_dp = J->gv_dp; bcfgpu_gvcf_block *gv_block = J->gv_block; uint8_t *gv_pl = J->gv_pl;
    /* ---- the record loop: the SNP record of a column, then its indel record (mpileup.c:343-366) ---- */
    static const char *nt = "ACGTN";
    int jl = 0;
    const double tw1 = want_timing ? now_s() : 0.;
    for (int k = 0; k < n_sites; ++k) {
        if (col_n[k] == 0) continue;                                         /* no read: no record */
        if (!target_keeps_column(contig, t0 + k)) continue;                  /* outside the targets: no record (mpileup.c:330-335) */
        const bcfgpu_site *c = &site[k];
        if (gv_blk && gv_blk[k] >= 0) {                                      /* inside a block: one line when the block ends */
            const int b = gv_blk[k];
            const bcfgpu_gvcf_block *B = &gv_block[b];
            if (B->first_site == k) {                                        /* the block starts: does it continue the one held back? */
                if (PB.on && !pending_joins(contig, B)) pending_flush();
            }
            if (B->last_site == k) {
                const bcfgpu_site *f = &site[B->first_site];
                const char refc = nt[f->ori_ref < 0 || f->ori_ref > 4 ? 4 : f->ori_ref];
                const uint8_t *bpl = gv_pl + (size_t)b * 3 * S; const int32_t *bdp = gv_dp + (size_t)b * S;
                if (PB.on) pending_merge(B, bpl, bdp);                       /* (a block that did not join was flushed at its first site) */
                else if (b == open_block) pending_set(contig, B, refc, f->qsum[0], f->qsum[1], bpl, bdp, S);
                else block_line(contig, B->start_pos, B->end1, B->min_dp, refc, f->qsum[0], f->qsum[1], bpl, bdp, S);
                if (PB.on && b != open_block) pending_flush();               /* joined, and it ends inside this tile */
            }
        } else {
        pending_flush();                                                     /* a record that cannot join ends the block (gvcf.c:107) */
        char als[64]; int o = 0;
        als[o++] = nt[c->ori_ref < 0 || c->ori_ref > 4 ? 4 : c->ori_ref]; als[o++] = '\t';
        for (int j = 1; j < c->n_alleles; ++j) {
            if (j > 1) als[o++] = ',';
            if (j == c->unseen) { memcpy(als + o, "<*>", 3); o += 3; } else als[o++] = nt[c->a[j]];
        }
        if (c->n_alleles < 2) als[o++] = '.';
        als[o] = 0;
        print_record(contig, t0 + k + 1, als, "", c, &snp_planes, (size_t)k, S);
        }
        while (jl < nlive && cand[live[jl]] < k) ++jl;
        if (jl < nlive && cand[live[jl]] == k && isite[jl].ret == 0) {      /* (site jl of the indel tile = the jl-th candidate with ret == 0) */
            /* REF / ALT of an indel record (bam2bcf.c:767-790) */
            pending_flush();
            const int i = live[jl], p = t0 + k, ireg = g_indelreg[i], mi = g_maxins[i];
            char *txt = malloc((size_t)(5 * (ireg + mi + 8)) + 64), prefix[64];
            int t = 0;
            for (int j = 0; j <= ireg; ++j) txt[t++] = ref[p + j];
            txt[t++] = '\t';
            for (int a = 1; a < 4 && isite[jl].a[a] >= 0; ++a) {
                const int ai = isite[jl].a[a], ty = g_types[i * 4 + ai];
                if (a > 1) txt[t++] = ',';
                txt[t++] = ref[p];
                if (ty < 0) { for (int j = p + 1 - ty; j < p + 1 + ireg; ++j) txt[t++] = ref[j]; }
                else {
                    for (int j = 0; j < ty; ++j) txt[t++] = nt[g_inscns[(size_t)i * 4 * INSCNS_CAP + (size_t)ai * mi + j]];
                    for (int j = p + 1; j < p + 1 + ireg; ++j) txt[t++] = ref[j];
                }
            }
            txt[t] = 0;
            snprintf(prefix, sizeof prefix, "INDEL;IDV=%d;IMF=%g;", g_support[i], (double)g_frac[i]);
            print_record(contig, p + 1, txt, prefix, &isite[jl], &ind_planes, (size_t)jl, S);
            free(txt);
        }
    }
    /* a column without reads ends a block too (a gap in positions, gvcf.c:131): nothing stays open past it */
    if (gv_n && PB.on && (open_block < 0 || PB.end1 != t1)) pending_flush();
    free(site); planes_free(&snp_planes); free(isite); planes_free(&ind_planes);
    free(col_n); free(col_indel); free(cand); free(live);
    free(g_types); free(g_maxins); free(g_indelreg); free(g_support); free(g_frac); free(g_inscns);
    free(gv_blk); free(gv_block); free(gv_dp); free(gv_pl);
    if (want_timing) t_emit += now_s() - tw1;
}

static struct { pthread_t th; pthread_mutex_t mu; pthread_cond_t cv; emit_job_t job; int on, have, busy, stop; } EM;
static void *emit_main(void *arg)
{
    (void)arg;
    for (;;) {
        pthread_mutex_lock(&EM.mu);
        while (!EM.have && !EM.stop) pthread_cond_wait(&EM.cv, &EM.mu);
        if (!EM.have) { pthread_mutex_unlock(&EM.mu); return NULL; }
        emit_job_t j = EM.job; EM.have = 0; EM.busy = 1;
        pthread_cond_broadcast(&EM.cv);
        pthread_mutex_unlock(&EM.mu);
        emit_tile(&j);
        pthread_mutex_lock(&EM.mu); EM.busy = 0; pthread_cond_broadcast(&EM.cv); pthread_mutex_unlock(&EM.mu);
    }
}
/* the worker has written everything handed over so far (before anybody else touches the output, the pending block or the reference) */
static void emit_wait(void)
{
    if (!EM.on) return;
    pthread_mutex_lock(&EM.mu);
    while (EM.have || EM.busy) pthread_cond_wait(&EM.cv, &EM.mu);
    pthread_mutex_unlock(&EM.mu);
}
static void emit_submit(const emit_job_t *j)
{
    if (!EM.on) {
        pthread_mutex_init(&EM.mu, NULL); pthread_cond_init(&EM.cv, NULL);
        if (pthread_create(&EM.th, NULL, emit_main, NULL)) DIE("cannot start the record writer\n");
        EM.on = 1;
    }
    pthread_mutex_lock(&EM.mu);
    while (EM.have) pthread_cond_wait(&EM.cv, &EM.mu);          /* (one tile waiting while one is written) */
    EM.job = *j; EM.have = 1;
    pthread_cond_broadcast(&EM.cv);
    pthread_mutex_unlock(&EM.mu);
}
static void emit_finish(void)
{
    if (!EM.on) return;
    emit_wait();
    pthread_mutex_lock(&EM.mu); EM.stop = 1; pthread_cond_broadcast(&EM.cv); pthread_mutex_unlock(&EM.mu);
    pthread_join(EM.th, NULL);
    EM.on = 0; EM.stop = 0;
}

/* ---- one tile: columns [t0, t1) of `contig` from the reads in P (all the reads that overlap the tile, file-major; file f =
 * [first[f], first[f+1])).  Every stage on the device; the records of the tile are written in position order. ---- */
static void process_tile(pool_t *P, const int *first, int F, int S, const char *contig, const char *ref, int ref_len, int t0, int t1)
{
    const int n_sites = t1 - t0;
    const double tw0 = want_timing ? now_s() : 0.;
    ensure_ctx(S, n_sites, (uint64_t)P->nbase + 64);
    /* mate overlaps: htslib pairs the reads inside one file's iterator (bam_mplp_init_overlaps, mpileup.c:640) */
    int32_t *pa = malloc((size_t)(P->n + 1) * sizeof *pa), *pb = malloc((size_t)(P->n + 1) * sizeof *pb);
    int np = 0;
    if (!no_overlaps) for (int f = 0; f < F; ++f) np += find_pairs(P, first[f], first[f + 1], pa + np, pb + np);
    tot_pairs += (unsigned long long)np;
    /* a sample fed by several files (mpileup.c:275-293 appends file after file): its reads merged by position, files in
     * order at equal positions, as bcfgpu_pileup wants them; the pairs follow their reads */
    {
        int sorted = 1;
        int32_t *last = malloc((size_t)S * sizeof *last);
        for (int s = 0; s < S; ++s) last[s] = INT32_MIN;
        for (int r = 0; r < P->n && sorted; ++r) { if (P->pos[r] < last[P->smpl[r]]) sorted = 0; last[P->smpl[r]] = P->pos[r]; }
        free(last);
        if (!sorted) {
            int32_t *ord = malloc((size_t)P->n * sizeof *ord), *tmp = malloc((size_t)P->n * sizeof *tmp), *inv = malloc((size_t)P->n * sizeof *inv);
            for (int r = 0; r < P->n; ++r) ord[r] = r;
            for (int w = 1; w < P->n; w *= 2) {                              /* bottom-up merge sort: stable */
                for (int lo = 0; lo < P->n; lo += 2 * w) {
                    const int mid = lo + w < P->n ? lo + w : P->n, hi = lo + 2 * w < P->n ? lo + 2 * w : P->n;
                    int i = lo, j = mid, k = lo;
                    while (i < mid && j < hi) tmp[k++] = P->pos[ord[j]] < P->pos[ord[i]] ? ord[j++] : ord[i++];
                    while (i < mid) tmp[k++] = ord[i++];
                    while (j < hi) tmp[k++] = ord[j++];
                }
                int32_t *t = ord; ord = tmp; tmp = t;
            }
            for (int r = 0; r < P->n; ++r) inv[ord[r]] = r;
            #define PERM(a) do { void *n_ = malloc((size_t)P->n * sizeof *P->a); for (int r = 0; r < P->n; ++r) memcpy((char *)n_ + (size_t)r * sizeof *P->a, &P->a[ord[r]], sizeof *P->a); \
                                 memcpy(P->a, n_, (size_t)P->n * sizeof *P->a); free(n_); } while (0)
            PERM(pos); PERM(lq); PERM(flag); PERM(ncig); PERM(cig_off); PERM(seq_off); PERM(smpl); PERM(file); PERM(end); PERM(mpos); PERM(isize);
            PERM(rnext_same); PERM(mapq); PERM(has_zq); PERM(qname);
            #undef PERM
            for (int i = 0; i < np; ++i) { pa[i] = inv[pa[i]]; pb[i] = inv[pb[i]]; }
            free(ord); free(tmp); free(inv);
        }
    }

    bcfgpu_reads rd; memset(&rd, 0, sizeof rd);
    rd.n_reads = P->n; rd.r_pos = P->pos; rd.r_lq = P->lq; rd.r_flag = P->flag; rd.r_ncig = P->ncig; rd.r_cig_off = P->cig_off;
    rd.r_seq_off = P->seq_off; rd.cig = P->cig; rd.seq16 = P->seq16; rd.qual = P->qual; rd.zq = P->zq; rd.r_has_zq = P->has_zq;

    /* the pool goes up once and stays in HBM: BAQ (new qualities and ZQ bytes for the reads it applies to; not with -B), the
     * mate-overlap tweak, the pileup of the tile -- each on the copy the stage before left there */
    CHECK(bcfgpu_pool_upload(ctx, &rd, NULL, P->mapq));
    if (baq_flag) CHECK(bcfgpu_pool_baq(ctx, ref, ref_len, baq_flag, NULL));
    CHECK(bcfgpu_pool_overlap_tweak(ctx, np, pa, pb));
    free(pa); free(pb);
    bcfgpu_tile tile;
    int32_t *col_n = malloc((size_t)(n_sites + 1) * sizeof *col_n);
    uint8_t *col_indel = malloc((size_t)n_sites + 1);
    CHECK(bcfgpu_pool_pileup(ctx, P->smpl, NULL, t0, t1, ref, ref_len, &tile, col_n, col_indel));
    tot_entries += (unsigned long long)tile.n_reads;
    /* ---- indel candidates (mpileup.c:354-365): candidate columns -> bcf_call_gap_prep.  Before either pass of bcf_call_glfgen runs:
     * errmod_cal's draw for cells of more than 255 reads is planned over BOTH passes of the tile in the reference's visit order --
     * position by position the SNP pass, then the indel pass where gap_prep returned >= 0 (bcfgpu_errmod_plan) ---- */
    int nc = 0;
    int32_t *cand = malloc((size_t)(n_sites + 1) * sizeof *cand);
    for (int k = 0; k < n_sites; ++k)
        if (!no_indels && col_indel[k] && col_n[k] < max_indel_depth * S && target_keeps_column(contig, t0 + k)) cand[nc++] = k;     /* mpileup.c:330-335, :354 */
    bcfgpu_site *isite = NULL;
    planes_t ind_planes; memset(&ind_planes, 0, sizeof ind_planes); int32_t *live = NULL; int nlive = 0;
    int32_t *g_types = NULL, *g_maxins = NULL, *g_indelreg = NULL, *g_support = NULL; float *g_frac = NULL; int8_t *g_inscns = NULL;
    int32_t *gret = NULL;
    bcfgpu_tile ti; memset(&ti, 0, sizeof ti);
    if (nc) {
        /* everything stays in HBM: the entries of the candidates that pass the pooled support filter, the stage on the pool
         * bcfgpu_pileup left there, p->aux straight into the indel pass's tile -- the candidates with ret == 0, in order (the host
         * pool is passed for its ZQ bytes) */
        bcfgpu_indel_in in; memset(&in, 0, sizeof in);
        in.n_sites = nc; in.n_smpl = S; in.ref = ref;
        in.openQ = openQ; in.extQ = extQ; in.tandemQ = tandemQ; in.min_support = min_support; in.per_sample_flt = per_sample_flt; in.min_frac = min_frac;
        bcfgpu_indel_out out; memset(&out, 0, sizeof out);
        gret = malloc((size_t)nc * 4);
        g_types = malloc((size_t)nc * 16); g_inscns = malloc((size_t)nc * 4 * INSCNS_CAP); g_maxins = malloc((size_t)nc * 4);
        g_indelreg = malloc((size_t)nc * 4); g_support = malloc((size_t)nc * 4); g_frac = malloc((size_t)nc * 4);
        out.ret = gret; out.p_aux = NULL; out.indel_types = g_types; out.inscns = g_inscns; out.maxins = g_maxins;
        out.indelreg = g_indelreg; out.max_support = g_support; out.max_frac = g_frac;
        /* (with BAQ the ZQ bytes are the pool's, in HBM; with -B the reads' own tags, if any, go up from the host) */
        CHECK(bcfgpu_gap_prep_tile(ctx, nc, cand, baq_flag ? NULL : &rd, &in, &out, INSCNS_CAP, &ti));
        live = malloc((size_t)nc * 4);
        for (int i = 0; i < nc; ++i) if (gret[i] == 0) live[nlive++] = i;
    }
    {
        int32_t *live_col = malloc((size_t)(nlive + 1) * 4);                 /* the SNP-tile column of every site of the indel tile */
        for (int j = 0; j < nlive; ++j) live_col[j] = cand[live[j]];
        /* a column outside the targets is passed over before bcf_call_glfgen (mpileup.c:330-335): errmod_cal spends no draw there */
        uint8_t *visit = malloc((size_t)n_sites + 1);
        for (int k = 0; k < n_sites; ++k) visit[k] = (uint8_t)target_keeps_column(contig, t0 + k);
        CHECK(bcfgpu_errmod_plan_visit(ctx, &tile, visit, nlive ? &ti : NULL, live_col, NULL));
        free(live_col); free(visit);
    }
    /* ---- the SNP pass, then the indel pass on the tile gap_prep left (the columns where it returned 0, mpileup.c:354-360) ---- */
    void *d_site = NULL, *d_pl = NULL, *d_dp4 = NULL;
    bcfgpu_site *site = NULL;
    planes_t snp_planes;
    run_mpileup(ctx, &tile, n_sites, &site, &snp_planes, gv_n ? &d_site : NULL, &d_pl, &d_dp4);
    if (nlive) run_mpileup(ctx, &ti, nlive, &isite, &ind_planes, NULL, NULL, NULL);
    free(gret);

    /* ---- --gvcf: reference-only records collapse into blocks (gvcf_write, gvcf.c:88-226) on the planes still in HBM ---- */
    int32_t *gv_blk = NULL, *gv_dp = NULL; bcfgpu_gvcf_block *gv_block = NULL; uint8_t *gv_pl = NULL; int32_t nb = 0;
    int open_block = -1;                 /* the block that reaches the tile's last column and may go on in the next tile */
    if (gv_n) {
        int32_t *pos = malloc((size_t)n_sites * 4); uint8_t *brk = calloc((size_t)n_sites, 1);
        for (int k = 0; k < n_sites; ++k) { pos[k] = t0 + k; if (col_n[k] == 0 || !target_keeps_column(contig, t0 + k)) brk[k] |= 2; }
        for (int j = 0; j < nlive; ++j) if (isite[j].ret == 0) brk[cand[live[j]]] |= 1;      /* an indel record follows the SNP record */
        void *d_pos, *d_brk, *d_blk, *d_min, *d_block, *d_gdp, *d_gpl;
        CHECK(bcfgpu_malloc(ctx, (size_t)n_sites * 4, &d_pos)); CHECK(bcfgpu_malloc(ctx, (size_t)n_sites, &d_brk));
        CHECK(bcfgpu_malloc(ctx, (size_t)n_sites * 4, &d_blk)); CHECK(bcfgpu_malloc(ctx, (size_t)n_sites * 4, &d_min));
        CHECK(bcfgpu_malloc(ctx, (size_t)n_sites * sizeof(bcfgpu_gvcf_block), &d_block));
        CHECK(bcfgpu_malloc(ctx, (size_t)n_sites * S * 4, &d_gdp)); CHECK(bcfgpu_malloc(ctx, (size_t)n_sites * 3 * S, &d_gpl));
        CHECK(bcfgpu_memcpy_h2d(ctx, d_pos, pos, (size_t)n_sites * 4)); CHECK(bcfgpu_memcpy_h2d(ctx, d_brk, brk, (size_t)n_sites));
        bcfgpu_gvcf_in gi; memset(&gi, 0, sizeof gi);
        gi.n_sites = n_sites; gi.n_range = gv_n; gi.dp_range = gv_range; gi.pos = d_pos; gi.brk = d_brk;
        gi.site = d_site; gi.pl = d_pl; gi.dp4 = d_dp4;
        bcfgpu_gvcf_out go = { d_blk, d_min, d_block, d_gdp, d_gpl };
        CHECK(bcfgpu_gvcf_blocks(ctx, &gi, &go, &nb));
        gv_blk = malloc((size_t)n_sites * 4); gv_block = malloc((size_t)(nb + 1) * sizeof *gv_block);
        gv_dp = malloc((size_t)(nb + 1) * S * 4); gv_pl = malloc((size_t)(nb + 1) * 3 * S);
        CHECK(bcfgpu_memcpy_d2h(ctx, gv_blk, d_blk, (size_t)n_sites * 4)); CHECK(bcfgpu_memcpy_d2h(ctx, gv_block, d_block, (size_t)nb * sizeof *gv_block));
        CHECK(bcfgpu_memcpy_d2h(ctx, gv_dp, d_gdp, (size_t)nb * S * 4)); CHECK(bcfgpu_memcpy_d2h(ctx, gv_pl, d_gpl, (size_t)nb * 3 * S));
        CHECK(bcfgpu_sync(ctx));
        /* open at the tile's end: the last column is in a block and no indel record follows it */
        if (gv_blk[n_sites - 1] >= 0 && !(brk[n_sites - 1] & 1)) open_block = gv_blk[n_sites - 1];
        bcfgpu_free(ctx, d_pos); bcfgpu_free(ctx, d_brk); bcfgpu_free(ctx, d_blk); bcfgpu_free(ctx, d_min); bcfgpu_free(ctx, d_block);
        bcfgpu_free(ctx, d_gdp); bcfgpu_free(ctx, d_gpl); free(pos); free(brk);
        bcfgpu_free(ctx, d_site); bcfgpu_free(ctx, d_pl); bcfgpu_free(ctx, d_dp4);
    }

    /* ---- the records: handed to the writer (emit_tile), which runs beside the next tile's device stages ---- */
    if (want_timing) t_dev += now_s() - tw0;
    {
        emit_job_t J; memset(&J, 0, sizeof J);
        J.n_sites = n_sites; J.t0 = t0; J.t1 = t1; J.S = S; J.nlive = nlive; J.open_block = open_block; J.contig = contig; J.ref = ref;
        J.col_n = col_n; J.col_indel = col_indel; J.cand = cand; J.live = live; J.site = site; J.isite = isite; J.snp_planes = snp_planes; J.ind_planes = ind_planes;
        J.g_types = g_types; J.g_maxins = g_maxins; J.g_indelreg = g_indelreg; J.g_support = g_support; J.g_frac = g_frac; J.g_inscns = g_inscns;
        J.gv_blk = gv_blk; J.gv_dp = gv_dp; J.gv_block = gv_block; J.gv_pl = gv_pl;
        emit_submit(&J);
    }
}

/* ---- the live window: the reads of every file that passed the filters and the depth cap and may still cover a column ---- */
typedef struct { lrec_t **r; int n, cap; } lwin_t;
static void lwin_push(lwin_t *w, lrec_t *x) { if (w->n == w->cap) { w->cap = w->cap ? 2 * w->cap : 256; w->r = grow(w->r, (size_t)w->cap * sizeof *w->r); } w->r[w->n++] = x; }

static void pool_clear(pool_t *P) { P->n = 0; P->ncigs = 0; P->nbase = 0; }
static void pool_add_rec(pool_t *P, int file, const lrec_t *x)
{
    pool_add(P, file, x->smpl, x->qname, x->flag, x->pos, x->mapq, x->rnext_same, x->mpos, x->isize, x->cig, x->ncig, x->lq, x->seq16, x->qual);
}

/* mpileup -C INT (mpileup.c:234-241) for a batch of reads: after BAQ, sam_cap_mapq lowers the mapping quality of reads with
 * many mismatches or drops them; then the -q and orphan filters, which read_passes() left for here.  keep[i] = 0: dropped. */
static bcfgpu_ctx *cap_ctx;
static void cap_batch(lrec_t **b, int n, int S, const char *ref, int ref_len, uint8_t *keep)
{
    static pool_t Q;
    pool_clear(&Q);
    for (int i = 0; i < n; ++i) pool_add_rec(&Q, 0, b[i]);
    if (!cap_ctx) {
        bcfgpu_cfg c0; memset(&c0, 0, sizeof c0);
        c0.device = device; c0.n_smpl = S; c0.max_sites = 1; c0.max_reads = 64; c0.min_baseQ = min_baseQ; c0.capQ = 60; c0.n_grp = 1; c0.ploidy_max = 2;
        CHECK(bcfgpu_create(&c0, &cap_ctx));
    }
    bcfgpu_reads r0; memset(&r0, 0, sizeof r0);
    r0.n_reads = Q.n; r0.r_pos = Q.pos; r0.r_lq = Q.lq; r0.r_flag = Q.flag; r0.r_ncig = Q.ncig; r0.r_cig_off = Q.cig_off;
    r0.r_seq_off = Q.seq_off; r0.cig = Q.cig; r0.seq16 = Q.seq16; r0.qual = Q.qual; r0.zq = Q.zq; r0.r_has_zq = Q.has_zq;
    /* the batch goes up once; BAQ and the cap run on that copy (the qualities sam_cap_mapq sees are BAQ's) */
    CHECK(bcfgpu_pool_upload(cap_ctx, &r0, NULL, Q.mapq));
    if (baq_flag) CHECK(bcfgpu_pool_baq(cap_ctx, ref, ref_len, baq_flag, NULL));
    int32_t *capv = malloc((size_t)(n + 1) * sizeof *capv);
    CHECK(bcfgpu_pool_cap_mapq(cap_ctx, ref, ref_len, cap_thres, capv));
    for (int i = 0; i < n; ++i) {
        keep[i] = capv[i] >= 0;
        if (keep[i] && b[i]->mapq > capv[i]) b[i]->mapq = capv[i];
        if (b[i]->mapq < min_mq) keep[i] = 0;
        if (!keep_orphans && (b[i]->flag & 1) && !(b[i]->flag & 2)) keep[i] = 0;
    }
    free(capv);
}

static region_t parse_region(const char *s)
{
    region_t g; memset(&g, 0, sizeof g);
    const char *colon = strrchr(s, ':');
    g.beg = 0; g.end = -1;
    if (!colon) { g.contig = strdup(s); return g; }
    g.contig = strndup(s, (size_t)(colon - s));
    char *e; long b = strtol(colon + 1, &e, 10);
    if (e == colon + 1) { free(g.contig); g.contig = strdup(s); return g; }          /* a colon inside the contig's name */
    g.beg = (int)(b > 0 ? b - 1 : 0);
    if (*e == '-' && e[1]) g.end = (int)strtol(e + 1, NULL, 10);
    else if (*e != '-') g.end = (int)b;                                              /* "chr:pos": one position */
    return g;
}

/* ---- --gpus N: the parent of the shard processes.  It touches no device: it starts one process per shard, each writing its
 * record stream (uncompressed BCF) to an anonymous temporary file the parent holds open, and writes the streams out in shard
 * order -- which is genomic order -- once the shards are through.  With --gvcf the last block of a shard and the first of the
 * next are joined where gvcf_write would have gone on (the same rule as between two tiles).
 * (The records of `mpileup` are every column's PL / DP planes, encoded on the host of the shard that computed them: a device
 * gather -- bcfgpu_gather_bytes, what the call-side drivers use for their compacted call records -- would carry host-made
 * bytes through HBM and back.) ---- */
typedef struct { char *line; int is_block, pos1, end1, min_dp; } gline_t;
static int gline_parse(gline_t *g, char *line)
{
    g->line = line; g->is_block = 0;
    char *f[10]; int nf = 0;
    char *dup = strdup(line);
    for (char *s = dup; nf < 10 && s; ) { f[nf++] = s; s = strchr(s, '\t'); if (s) *s++ = 0; }
    if (nf >= 9 && !strcmp(f[4], "<*>") && strstr(f[7], "MinDP=") && !strcmp(f[8], "PL:DP")) {
        g->is_block = 1; g->pos1 = atoi(f[1]);
        const char *e = strstr(f[7], "END="); g->end1 = e == f[7] || (e && e[-1] == ';') ? atoi(e + 4) : g->pos1;
        g->min_dp = atoi(strstr(f[7], "MinDP=") + 6);
    }
    free(dup);
    return g->is_block;
}
static int dp_range_of(int min_dp) { int r = 0; while (r < gv_n && min_dp >= gv_range[r]) ++r; return r; }
/* a (held back) + b, both block lines of the same contig: the joined line, malloc'ed */
static char *gline_join(const gline_t *a, const gline_t *b)
{
    char *fa[10] = {0}, *fb[10] = {0}; int na = 0, nb = 0;
    char *da = strdup(a->line), *db = strdup(b->line);
    for (char *s = da; na < 9 && s; ) { fa[na++] = s; s = strchr(s, '\t'); if (s) *s++ = 0; if (na == 9) fa[9] = s; }
    for (char *s = db; nb < 9 && s; ) { fb[nb++] = s; s = strchr(s, '\t'); if (s) *s++ = 0; if (nb == 9) fb[9] = s; }
    const char *qs = strstr(fa[7], "QS=");
    char *out = NULL; size_t len = 0;
    FILE *m = open_memstream(&out, &len);
    fprintf(m, "%s\t%d\t.\t%s\t<*>\t.\t.\tEND=%d;MinDP=%d;%s\tPL:DP", fa[0], a->pos1, fa[3], b->end1, a->min_dp < b->min_dp ? a->min_dp : b->min_dp, qs ? qs : "");
    char *sa = fa[9], *sb = fb[9];
    while (sa && sb && *sa && *sb) {
        int pa[3], pb_[3], da_, db_;
        if (sscanf(sa, "%d,%d,%d:%d", &pa[0], &pa[1], &pa[2], &da_) != 4 || sscanf(sb, "%d,%d,%d:%d", &pb_[0], &pb_[1], &pb_[2], &db_) != 4) DIE("--gvcf: cannot join two blocks\n");
        if (pb_[1] < pa[1] || (pb_[1] == pa[1] && pb_[2] < pa[2])) { pa[1] = pb_[1]; pa[2] = pb_[2]; }
        fprintf(m, "\t%d,%d,%d:%d", pa[0], pa[1], pa[2], da_ < db_ ? da_ : db_);
        sa = strchr(sa, '\t'); if (sa) ++sa;
        sb = strchr(sb, '\t'); if (sb) ++sb;
    }
    fclose(m); free(da); free(db);
    return out;
}

static const char *shard_transport = "";
static int run_shards(int n_gpus, int argc0, char **argv0, int first_file, const char *ref_path, region_t *reg, int n_reg,
                      const char *out_path, char out_mode)
{
    /* the shards: the columns of all regions, in order, cut into n_gpus contiguous runs */
    long total = 0;
    for (int i = 0; i < n_reg; ++i) total += reg[i].end - reg[i].beg;
    const int n_sh = (long)n_gpus < total ? n_gpus : (total > 0 ? (int)total : 1);
    pid_t *pid = malloc((size_t)n_sh * sizeof *pid);
    int *rfd = malloc((size_t)n_sh * sizeof *rfd);
    extern char **environ;
    for (int k = 0; k < n_sh; ++k) {
        const long c0 = total * k / n_sh, c1 = total * (k + 1) / n_sh;
        char *rl = NULL; size_t rlen = 0; FILE *m = open_memstream(&rl, &rlen);
        long at = 0; int nr = 0;
        for (int i = 0; i < n_reg; ++i) {
            const long lo = at > c0 ? at : c0, hi = at + (reg[i].end - reg[i].beg) < c1 ? at + (reg[i].end - reg[i].beg) : c1;
            if (lo < hi) {
                if (reg[i].open && hi == at + (reg[i].end - reg[i].beg)) fprintf(m, "%s%s:%ld-", nr++ ? "," : "", reg[i].contig, reg[i].beg + (lo - at) + 1);
                else fprintf(m, "%s%s:%ld-%ld", nr++ ? "," : "", reg[i].contig, reg[i].beg + (lo - at) + 1, reg[i].beg + (hi - at));
            }
            at += reg[i].end - reg[i].beg;
        }
        fclose(m);
        /* the shard's records go to shared memory: an anonymous memory file (memfd_create: pages in RAM, no name in any file
         * system, nothing anyone could predict or replace) that this process holds open; the shard inherits the descriptor and
         * opens it by number.  (Where the kernel has no memfd_create: an unlinked temporary file, held the same way.) */
        int tfd = (int)syscall(SYS_memfd_create, "bcfgpu_shard", 0u);
        if (tfd >= 0) shard_transport = "memfd (shared memory)";
        else {
            FILE *tf = tmpfile();
            if (!tf) DIE("tmpfile failed\n");
            tfd = dup(fileno(tf));
            fclose(tf);
            shard_transport = "unlinked temporary file";
        }
        if (tfd < 0 || fcntl(tfd, F_SETFD, 0)) DIE("shard descriptor\n");
        char **av = malloc((size_t)(argc0 + 16) * sizeof *av);
        int n = 0;
        char sk[24], sfd[40]; snprintf(sk, sizeof sk, "%d", k); snprintf(sfd, sizeof sfd, "/dev/fd/%d", tfd);
        av[n++] = argv0[0]; av[n++] = "--shard"; av[n++] = strdup(sk);
        for (int i = 1; i < first_file; ++i) {                   /* the options, without -O / -o FILE / --output / --gpus / -r / -f and the old positional region */
            const char *o = argv0[i];
            if (!strcmp(o, "--gpus") || !strcmp(o, "--output") || !strcmp(o, "-O") || !strcmp(o, "-r") || !strcmp(o, "--regions") || !strcmp(o, "-R") || !strcmp(o, "--regions-file") || !strcmp(o, "-f") || !strcmp(o, "--fasta-ref")) { ++i; continue; }
            if (!strncmp(o, "-O", 2) && o[2]) continue;
            if (!strcmp(o, "-o")) { char *e; strtol(argv0[i + 1], &e, 10); if (*e) { ++i; continue; } }
            if (o[0] != '-') break;                              /* the positional form: ref.fa contig beg end come from -f / -r below */
            av[n++] = argv0[i];
            if (o[0] == '-' && i + 1 < first_file && argv0[i + 1][0] != '-' && strcmp(o, "-B") && strcmp(o, "-E") && strcmp(o, "-A") && strcmp(o, "-p") && strcmp(o, "-I") && strcmp(o, "-6") && strcmp(o, "--illumina1.3+") && strcmp(o, "--timing") && strcmp(o, "-x") && strcmp(o, "--ignore-overlaps") && strcmp(o, "--no-version")
                && strcmp(o, "--ignore-RG") && strcmp(o, "--list-samples")) av[n++] = argv0[++i];
        }
        av[n++] = "-f"; av[n++] = (char *)ref_path; av[n++] = "-r"; av[n++] = rl;
        av[n++] = "-O"; av[n++] = "u"; av[n++] = "--output"; av[n++] = strdup(sfd);
        for (int i = first_file; i < argc0; ++i) av[n++] = argv0[i];
        av[n] = NULL;
        if (posix_spawn(&pid[k], "/proc/self/exe", NULL, NULL, av, environ)) {
            for (int j = 0; j < k; ++j) { kill(pid[j], SIGTERM); waitpid(pid[j], NULL, 0); }          /* no orphans behind a failed start */
            DIE("cannot start shard %d\n", k);
        }
        rfd[k] = tfd;
        free(av);
    }
    int bad = 0;
    for (int k = 0; k < n_sh; ++k) { int st = 0; if (waitpid(pid[k], &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st)) bad = 1; }
    if (bad) DIE("a shard failed\n");
    fprintf(stderr, "[bcfgpu_sam] %d region shards, one process each; their record streams (uncompressed BCF through %s) are emitted in shard order on the host\n", n_sh, shard_transport);
    char *lb = NULL; size_t lcap = 0;
    gline_t held; char *held_line = NULL; memset(&held, 0, sizeof held);
    for (int k = 0; k < n_sh; ++k) {
        char path[40]; snprintf(path, sizeof path, "/dev/fd/%d", rfd[k]);
        vio_file *fk = vio_open_read(path);
        vio_hdr *hk = fk ? vio_read_hdr(fk) : NULL;
        if (!fk || !hk) DIE("shard %d: %s\n", k, vio_error());
        if (k == 0) {
            hdr = hk;
            fout = vio_open_write(out_path, out_mode);
            if (!fout || vio_write_hdr(fout, hdr)) DIE("%s\n", vio_error());
        }
        int rr, first = 1;
        while ((rr = vio_read_line(fk, hk, &lb, &lcap)) > 0) {
            if (!lb[0]) continue;
            gline_t g;
            if (gv_n && gline_parse(&g, lb)) {
                /* the shard's first line continues the block the shard before ended with? */
                if (held_line && first && !strncmp(held_line, lb, strcspn(lb, "\t") + 1) && g.pos1 == held.end1 + 1 && dp_range_of(g.min_dp) == dp_range_of(held.min_dp)) {
                    char *j = gline_join(&held, &g);
                    free(held_line); held_line = j; gline_parse(&held, held_line);
                } else {
                    if (held_line) { if (vio_write_line(fout, hdr, held_line)) DIE("%s\n", vio_error()); free(held_line); }
                    held_line = strdup(lb); gline_parse(&held, held_line);
                }
            } else {
                if (held_line) { if (vio_write_line(fout, hdr, held_line)) DIE("%s\n", vio_error()); free(held_line); held_line = NULL; }
                if (vio_write_line(fout, hdr, lb)) DIE("%s\n", vio_error());
            }
            first = 0;
        }
        if (rr < 0) DIE("shard %d: %s\n", k, vio_error());
        vio_close(fk);
        if (k) vio_hdr_free(hk);
    }
    if (held_line) { if (vio_write_line(fout, hdr, held_line)) DIE("%s\n", vio_error()); free(held_line); }
    free(lb);
    if (vio_close(fout)) DIE("%s\n", vio_error());
    return 0;
}

int main(int argc, char **argv)
{
    char out_mode = 'v'; const char *out_path = "-";                           /* mpileup -O, -o (mpileup.c:937-950) */
    int list_only = 0, n_gpus = 1, shard = -1;                                /* --gpus N: region shards, one process per shard; --shard K: this is shard K */
    const char *ref_path = NULL, *reg_arg = NULL, *reg_file = NULL, *file_list = NULL;
    {   /* mpileup's long option names (mpileup.c:952-1003) are read as their short forms */
        static const char *alias[][2] = {
            { "--count-orphans", "-A" }, { "--no-BAQ", "-B" }, { "--adjust-MQ", "-C" }, { "--max-depth", "-d" }, { "--redo-BAQ", "-E" },
            { "--read-groups", "-G" }, { "--min-MQ", "-q" }, { "--min-BQ", "-Q" }, { "--incl-flags", "--rf" }, { "--excl-flags", "--ff" },
            { "--annotate", "-a" }, { "--output-type", "-O" }, { "--samples", "-s" }, { "--samples-file", "-S" }, { "--ext-prob", "-e" },
            { "--gap-frac", "-F" }, { "--tandem-qual", "-h" }, { "--skip-indels", "-I" }, { "--max-idepth", "-L" }, { "--min-ireads", "-m" },
            { "--open-prob", "-o" }, { "--per-sample-mF", "-p" } };
        for (int i = 1; i < argc; ++i)
            for (size_t k = 0; k < sizeof alias / sizeof alias[0]; ++k) if (!strcmp(argv[i], alias[k][0])) argv[i] = (char *)alias[k][1];
    }
    char **argv0 = argv; const int argc0 = argc;
    while (argc > 2 && argv[1][0] == '-' && argv[1][1]) {
        if (!strcmp(argv[1], "-a")) {                                         /* mpileup -a, mpileup.c:parse_format_flag */
            /* parse_format_flag (mpileup.c:794-826): the names without regard to case, FORMAT tags with or without "FORMAT/" or "FMT/" */
            static const struct { const char *name; int bit; } tags[] = {
                { "DP", BCFGPU_FMT_DP }, { "DV", BCFGPU_FMT_DV }, { "SP", BCFGPU_FMT_SP }, { "DP4", BCFGPU_FMT_DP4 }, { "DPR", BCFGPU_FMT_DPR },
                { "AD", BCFGPU_FMT_AD }, { "ADF", BCFGPU_FMT_ADF }, { "ADR", BCFGPU_FMT_ADR }, { "SCR", BCFGPU_FMT_SCR }, { "QS", BCFGPU_FMT_QS },
                { "INFO/DPR", BCFGPU_INFO_DPR }, { "INFO/AD", BCFGPU_INFO_AD }, { "INFO/ADF", BCFGPU_INFO_ADF }, { "INFO/ADR", BCFGPU_INFO_ADR },
                { "INFO/SCR", BCFGPU_INFO_SCR } };
            char *list = strdup(argv[2]);
            for (char *t = strtok(list, ","); t; t = strtok(NULL, ",")) {
                const char *name = !strncasecmp(t, "FORMAT/", 7) ? t + 7 : !strncasecmp(t, "FMT/", 4) ? t + 4 : t;
                size_t i;
                for (i = 0; i < sizeof tags / sizeof tags[0]; ++i)
                    if (!strcasecmp(name, tags[i].name) && (name == t || strncasecmp(tags[i].name, "INFO/", 5))) { fmt_flag |= tags[i].bit; break; }
                if (i == sizeof tags / sizeof tags[0]) { fprintf(stderr, "Could not parse tag \"%s\" in \"%s\"\n", t, argv[2]); exit(1); }
            }
            free(list);
            argv += 2; argc -= 2;
        } else if (!strcmp(argv[1], "--gvcf") || !strcmp(argv[1], "-g")) {
            char *list = strdup(argv[2]);
            for (char *t = strtok(list, ","); t; t = strtok(NULL, ",")) { if (gv_n == 16) DIE("--gvcf: at most 16 limits\n"); gv_range[gv_n++] = atoi(t); }
            free(list);
            fmt_flag |= BCFGPU_FMT_DP;                                        /* mpileup.c:1101-1105 */
            argv += 2; argc -= 2;
        }
        else if (!strcmp(argv[1], "-O")) { out_mode = argv[2][0]; argv += 2; argc -= 2; }
        else if (!strncmp(argv[1], "-O", 2) && argv[1][2]) { out_mode = argv[1][2]; ++argv; --argc; }
        else if (!strcmp(argv[1], "-o")) {                                    /* -o INT: --open-prob, -o FILE: --output (the reference's own rule, mpileup.c:1073-1079) */
            char *e; const long v = strtol(argv[2], &e, 10);
            if (*e == 0) openQ = (int)v; else out_path = argv[2];
            argv += 2; argc -= 2;
        }
        else if (!strcmp(argv[1], "--output")) { out_path = argv[2]; argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "-f") || !strcmp(argv[1], "--fasta-ref")) { ref_path = argv[2]; argv += 2; argc -= 2; }      /* mpileup.c:1008,1056 */
        else if (!strcmp(argv[1], "-r") || !strcmp(argv[1], "--regions")) { reg_arg = argv[2]; argv += 2; argc -= 2; }          /* mpileup.c:1011,1057 */
        else if (!strcmp(argv[1], "-R") || !strcmp(argv[1], "--regions-file")) { reg_file = argv[2]; argv += 2; argc -= 2; }    /* mpileup.c:1031 */
        else if (!strcmp(argv[1], "-t") || !strcmp(argv[1], "--targets")) {                                                     /* mpileup.c:1033-1045 */
            const char *a = argv[2]; if (a[0] == '^') { ++a; target_incl = 0; }
            char *list = strdup(a);
            for (char *t = strtok(list, ","); t; t = strtok(NULL, ",")) target_spec(t);
            free(list); argv += 2; argc -= 2;
        }
        else if (!strcmp(argv[1], "-T") || !strcmp(argv[1], "--targets-file")) {                                                /* mpileup.c:1046-1051 */
            const char *a = argv[2]; if (a[0] == '^') { ++a; target_incl = 0; }
            target_file(a); argv += 2; argc -= 2;
        }
        else if (!strcmp(argv[1], "-b") || !strcmp(argv[1], "--bam-list")) { file_list = argv[2]; argv += 2; argc -= 2; }       /* mpileup.c:1072 */
        else if (!strcmp(argv[1], "-x") || !strcmp(argv[1], "--ignore-overlaps")) { no_overlaps = 1; ++argv; --argc; }          /* mpileup.c:1005 */
        else if (!strcmp(argv[1], "-P") || !strcmp(argv[1], "--platforms")) { argv += 2; argc -= 2; }                          /* read and never used by the reference either (mpileup.c:353, 1052) */
        else if (!strcmp(argv[1], "--no-version")) { ++argv; --argc; }                                                          /* (no ##bcftoolsVersion / ##bcftoolsCommand lines are written anyway) */
        else if (!strcmp(argv[1], "--threads")) { argv += 2; argc -= 2; }                                                      /* (the output's compression threads: nothing to do here) */
        else if (!strcmp(argv[1], "--timing")) { want_timing = 1; argv += 1; argc -= 1; }
        else if (!strcmp(argv[1], "--tile")) { tile_cols = atoi(argv[2]); if (tile_cols < 1) DIE("--tile: at least one column\n"); argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "-d")) { max_depth = atoi(argv[2]); argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "-s")) { add_samples(argv[2], 0); argv += 2; argc -= 2; }            /* mpileup.c:1058-1059,1087,1016 */
        else if (!strcmp(argv[1], "-S")) { add_samples(argv[2], 1); argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "-G")) { add_readgroups(argv[2]); argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "--ignore-RG")) { SM.ignore_rg = 1; ++argv; --argc; }
        else if (!strcmp(argv[1], "--list-samples")) { list_only = 1; ++argv; --argc; }
        else if (!strcmp(argv[1], "--gpus")) { n_gpus = atoi(argv[2]); argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "--shard")) { shard = atoi(argv[2]); argv += 2; argc -= 2; }                 /* host logic only: no device needed */
        else if (!strcmp(argv[1], "-6") || !strcmp(argv[1], "--illumina1.3+")) { illumina13 = 1; ++argv; --argc; }                  /* mpileup.c:1057 */
        else if (!strcmp(argv[1], "-B")) { baq_flag = 0; ++argv; --argc; }                             /* mpileup.c:1045,1062 */
        else if (!strcmp(argv[1], "-E")) { baq_flag = 7; ++argv; --argc; }
        else if (!strcmp(argv[1], "-A")) { keep_orphans = 1; ++argv; --argc; }
        else if (!strcmp(argv[1], "-q")) { min_mq = atoi(argv[2]); argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "-Q")) { min_baseQ = atoi(argv[2]); argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "-C")) { cap_thres = atoi(argv[2]); argv += 2; argc -= 2; }          /* mpileup.c:1069 */
        else if (!strcmp(argv[1], "-e")) { extQ = atoi(argv[2]); argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "-h")) { tandemQ = atoi(argv[2]); argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "-m")) { min_support = atoi(argv[2]); argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "-F")) { min_frac = atof(argv[2]); argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "-L")) { max_indel_depth = atoi(argv[2]); argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "-p")) { per_sample_flt = 1; ++argv; --argc; }
        else if (!strcmp(argv[1], "-I")) { no_indels = 1; ++argv; --argc; }
        else if (!strcmp(argv[1], "--ff")) { rflag_filter = (int)strtol(argv[2], NULL, 0); argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "--rf")) { rflag_require = (int)strtol(argv[2], NULL, 0); argv += 2; argc -= 2; }
        else break;
    }
    /* two spellings of what to run on: mpileup's own (-f REF [-r CHR[:BEG[-END]],...] files), or "ref.fa contig beg end files" */
    region_t *reg = NULL; int n_reg = 0;
    int n_in; char **in_path;
    int n_pos;                                                                /* input files named on the command line (the others: -b) */
    if (ref_path) {
        if (argc < 2 && !file_list) goto usage;
        n_in = n_pos = argc - 1; in_path = argv + 1;
        if (file_list) {
            /* -b FILE: a file of input paths, one per line, empty lines and trailing blanks allowed (read_file_list, mpileup.c:733-790) */
            FILE *fl = fopen(file_list, "r");
            if (!fl) DIE("cannot open %s\n", file_list);
            char **all = grow(NULL, (size_t)(n_in + 1) * sizeof *all);
            memcpy(all, in_path, (size_t)n_in * sizeof *all);
            char *ln = NULL; size_t lcap = 0;
            while (getline(&ln, &lcap, fl) > 0) {
                size_t l = strlen(ln);
                while (l && isspace((unsigned char)ln[l - 1])) ln[--l] = 0;
                if (!l) continue;
                all = grow(all, (size_t)(n_in + 2) * sizeof *all);
                all[n_in++] = strdup(ln);
            }
            free(ln); fclose(fl);
            if (n_in == n_pos) DIE("No files read from %s\n", file_list);
            in_path = all;
        }
        if (reg_arg) {
            char *list = strdup(reg_arg);
            for (char *t = strtok(list, ","); t; t = strtok(NULL, ",")) { reg = grow(reg, (size_t)(n_reg + 1) * sizeof *reg); reg[n_reg++] = parse_region(t); }
            free(list);
        }
        if (reg_file) {
            /* -R FILE: tab-delimited CHROM, POS and, optionally, END (1-based, inclusive), or CHROM alone; '#' starts a comment */
            FILE *fr = fopen(reg_file, "r");
            if (!fr) DIE("cannot open %s\n", reg_file);
            char *ln = NULL; size_t lcap = 0;
            while (getline(&ln, &lcap, fr) > 0) {
                size_t l = strlen(ln);
                while (l && isspace((unsigned char)ln[l - 1])) ln[--l] = 0;
                if (!l || ln[0] == '#') continue;
                char *c1 = strchr(ln, '\t'), *c2 = c1 ? strchr(c1 + 1, '\t') : NULL;
                char spec[1200];
                if (!c1) snprintf(spec, sizeof spec, "%s", ln);
                else { *c1 = 0; if (c2) *c2 = 0; snprintf(spec, sizeof spec, "%.1000s:%ld-%ld", ln, atol(c1 + 1), c2 ? atol(c2 + 1) : atol(c1 + 1)); }
                reg = grow(reg, (size_t)(n_reg + 1) * sizeof *reg); reg[n_reg++] = parse_region(spec);
            }
            free(ln); fclose(fr);
            if (!n_reg) DIE("no region in %s\n", reg_file);
        }
    } else {
        if (argc < 6) goto usage;
        ref_path = argv[1];
        reg = grow(NULL, sizeof *reg); n_reg = 1;
        reg[0].contig = strdup(argv[2]); reg[0].beg = atoi(argv[3]) - 1; reg[0].end = atoi(argv[4]);     /* 1-based inclusive -> 0-based [beg, end) */
        n_in = n_pos = argc - 5; in_path = argv + 5;
    }
    defer_mq_filters = cap_thres > 10;

    /* ---- the input files: their headers decide the samples (bam_smpl_add_bam) and give the ##contig lines ---- */
    reader_t *rdr = calloc((size_t)n_in, sizeof *rdr);
    sfile_t *sfile = calloc((size_t)n_in, sizeof *sfile);
    const char **kept_path = calloc((size_t)n_in, sizeof *kept_path);
    hdr = vio_hdr_new();
    { char b[4096]; snprintf(b, sizeof b, "##reference=file://%s", ref_path); vio_hdr_append(hdr, b); }
    int F = 0;                                                                /* files kept (mpileup.c:442-455 drops the others) */
    char **all_contig = NULL; int n_all = 0;                                   /* no -r: every sequence of the first file's dictionary */
    for (int i = 0; i < n_in; ++i) {
        reader_open(&rdr[F], in_path[i]);
        if (i == 0 && !reg) {
            if (rdr[F].is_bam) for (int k = 0; k < rdr[F].n_ref; ++k) { all_contig = grow(all_contig, (size_t)(n_all + 1) * sizeof *all_contig); all_contig[n_all++] = strdup(rdr[F].ref_name[k]); }
            else for (const char *l = rdr[F].text; l && *l; ) {
                if (!strncmp(l, "@SQ\t", 4)) { const char *sn = strstr(l, "\tSN:"), *e = strchr(l, '\n'); if (sn && (!e || sn < e)) { sn += 4; all_contig = grow(all_contig, (size_t)(n_all + 1) * sizeof *all_contig); all_contig[n_all++] = strndup(sn, strcspn(sn, "\t\r\n")); } }
                l = strchr(l, '\n'); if (l) ++l;
            }
        }
        if (!add_file(&sfile[F], in_path[i], rdr[F].text)) { reader_close(&rdr[F]); continue; }
        if (F == 0) header_contigs(&rdr[0], hdr);
        reader_close(&rdr[F]);
        kept_path[F++] = in_path[i];
    }
    if (!reg) { for (int k = 0; k < n_all; ++k) { reg = grow(reg, (size_t)(n_reg + 1) * sizeof *reg); reg[n_reg].contig = all_contig[k]; reg[n_reg].beg = 0; reg[n_reg++].end = -1; } }
    if (!n_reg) DIE("no region to run on: no -r and no sequence dictionary in the first file\n");
    const int S = SM.nsmpl;
    if (!F || !S) DIE("no sample left to call\n");
    char **sample = SM.smpl;
    /* the contigs' sequences (one at a time in memory).  A region without an END runs to wherever the reads end -- past the
     * contig's last base if they do, as the reference's iterator does (reference base N there); for the shards it is cut at
     * the contig's length and the last piece stays open */
    char *ref = NULL, *ref_name = NULL; int ref_len = 0;
    for (int i = 0; i < n_reg; ++i) if (reg[i].end < 0) { reg[i].open = 1; reg[i].end = OPEN_END; }
    if (n_gpus > 1 && shard < 0 && !list_only) {
        for (int i = 0; i < n_reg; ++i) if (reg[i].open) { int len = 0; char *sq = read_contig(ref_path, reg[i].contig, &len); reg[i].end = len > reg[i].beg ? len : reg[i].beg + 1; free(sq); }
        return run_shards(n_gpus, argc0, argv0, argc0 - n_pos, ref_path, reg, n_reg, out_path, out_mode);
    }
    if (shard > 0 && !list_only) { const int nd = bcfgpu_device_count(); device = nd > 0 ? shard % nd : 0; if (nd > 1) fprintf(stderr, "[bcfgpu_sam] shard %d on device %d of %d\n", shard, device, nd); }
    /* ---- the VCF header, in mpileup's order (mpileup.c:510-602) ---- */
    {
        #define HL(cond, text) do { if (cond) vio_hdr_append(hdr, text); } while (0)
        HL(1, "##ALT=<ID=*,Description=\"Represents allele(s) other than observed.\">");
        HL(1, "##INFO=<ID=INDEL,Number=0,Type=Flag,Description=\"Indicates that the variant is an INDEL.\">");
        HL(1, "##INFO=<ID=IDV,Number=1,Type=Integer,Description=\"Maximum number of raw reads supporting an indel\">");
        HL(1, "##INFO=<ID=IMF,Number=1,Type=Float,Description=\"Maximum fraction of raw reads supporting an indel\">");
        HL(1, "##INFO=<ID=DP,Number=1,Type=Integer,Description=\"Raw read depth\">");
        HL(fmt_flag & BCFGPU_INFO_VDB, "##INFO=<ID=VDB,Number=1,Type=Float,Description=\"Variant Distance Bias for filtering splice-site artefacts in RNA-seq data (bigger is better)\",Version=\"3\">");
        HL(fmt_flag & BCFGPU_INFO_RPB, "##INFO=<ID=RPB,Number=1,Type=Float,Description=\"Mann-Whitney U test of Read Position Bias (bigger is better)\">");
        HL(1, "##INFO=<ID=MQB,Number=1,Type=Float,Description=\"Mann-Whitney U test of Mapping Quality Bias (bigger is better)\">");
        HL(1, "##INFO=<ID=BQB,Number=1,Type=Float,Description=\"Mann-Whitney U test of Base Quality Bias (bigger is better)\">");
        HL(1, "##INFO=<ID=MQSB,Number=1,Type=Float,Description=\"Mann-Whitney U test of Mapping Quality vs Strand Bias (bigger is better)\">");
        HL(1, "##INFO=<ID=SGB,Number=1,Type=Float,Description=\"Segregation based metric.\">");
        HL(1, "##INFO=<ID=MQ0F,Number=1,Type=Float,Description=\"Fraction of MQ0 reads (smaller is better)\">");
        HL(1, "##INFO=<ID=I16,Number=16,Type=Float,Description=\"Auxiliary tag used for calling, see description of bcf_callret1_t in bam2bcf.h\">");
        HL(1, "##INFO=<ID=QS,Number=R,Type=Float,Description=\"Auxiliary tag used for calling\">");
        HL(1, "##FORMAT=<ID=PL,Number=G,Type=Integer,Description=\"List of Phred-scaled genotype likelihoods\">");
        HL(fmt_flag & BCFGPU_FMT_DP, "##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"Number of high-quality bases\">");
        HL(fmt_flag & BCFGPU_FMT_DV, "##FORMAT=<ID=DV,Number=1,Type=Integer,Description=\"Number of high-quality non-reference bases\">");
        HL(fmt_flag & BCFGPU_FMT_DPR, "##FORMAT=<ID=DPR,Number=R,Type=Integer,Description=\"Number of high-quality bases observed for each allele\">");
        HL(fmt_flag & BCFGPU_INFO_DPR, "##INFO=<ID=DPR,Number=R,Type=Integer,Description=\"Number of high-quality bases observed for each allele\">");
        HL(fmt_flag & BCFGPU_FMT_DP4, "##FORMAT=<ID=DP4,Number=4,Type=Integer,Description=\"Number of high-quality ref-fwd, ref-reverse, alt-fwd and alt-reverse bases\">");
        HL(fmt_flag & BCFGPU_FMT_SP, "##FORMAT=<ID=SP,Number=1,Type=Integer,Description=\"Phred-scaled strand bias P-value\">");
        HL(fmt_flag & BCFGPU_FMT_AD, "##FORMAT=<ID=AD,Number=R,Type=Integer,Description=\"Allelic depths (high-quality bases)\">");
        HL(fmt_flag & BCFGPU_FMT_ADF, "##FORMAT=<ID=ADF,Number=R,Type=Integer,Description=\"Allelic depths on the forward strand (high-quality bases)\">");
        HL(fmt_flag & BCFGPU_FMT_ADR, "##FORMAT=<ID=ADR,Number=R,Type=Integer,Description=\"Allelic depths on the reverse strand (high-quality bases)\">");
        HL(fmt_flag & BCFGPU_FMT_QS, "##FORMAT=<ID=QS,Number=R,Type=Integer,Description=\"Phred-score allele quality sum used by `call -mG` and `+trio-dnm`\">");
        HL(fmt_flag & BCFGPU_INFO_AD, "##INFO=<ID=AD,Number=R,Type=Integer,Description=\"Total allelic depths (high-quality bases)\">");
        HL(fmt_flag & BCFGPU_INFO_ADF, "##INFO=<ID=ADF,Number=R,Type=Integer,Description=\"Total allelic depths on the forward strand (high-quality bases)\">");
        HL(fmt_flag & BCFGPU_INFO_SCR, "##INFO=<ID=SCR,Number=1,Type=Integer,Description=\"Number of soft-clipped reads (at high-quality bases)\">");
        HL(fmt_flag & BCFGPU_FMT_SCR, "##FORMAT=<ID=SCR,Number=1,Type=Integer,Description=\"Per-sample number of soft-clipped reads (at high-quality bases)\">");
        HL(fmt_flag & BCFGPU_INFO_ADR, "##INFO=<ID=ADR,Number=R,Type=Integer,Description=\"Total allelic depths on the reverse strand (high-quality bases)\">");
        HL(gv_n, "##INFO=<ID=END,Number=1,Type=Integer,Description=\"End position of the variant described in this record\">");   /* gvcf.c:42-43 */
        HL(gv_n, "##INFO=<ID=MinDP,Number=1,Type=Integer,Description=\"Minimum per-sample depth in this gVCF block\">");
        #undef HL
        for (int s = 0; s < S; ++s) vio_hdr_add_sample(hdr, sample[s]);
    }
    if (!list_only) {
        fout = vio_open_write(out_path, out_mode);
        if (!fout || vio_write_hdr(fout, hdr)) DIE("%s\n", vio_error());
        LN = open_memstream(&ln_buf, &ln_len);
        if (!LN) DIE("open_memstream failed\n");
    }

    /* ---- the regions, one after the other (mpileup.c:652-683), each streamed through in tiles of tile_cols columns: the files
     * are read in step with the tiles (a sorted file no further than the tile's end), a read stays in memory while it can still
     * cover a column, and what is on the device at any time is one tile.  SURVEY 8e: "shards further cut into tiles". ---- */
    bcfgpu_depth_state *dcap = bcfgpu_depth_cap_new(F, max_depth);             /* the iterators' buffers (mpileup -d), one per file */
    if (!dcap) DIE("%s\n", bcfgpu_last_error());
    lwin_t *win = calloc((size_t)F, sizeof *win);
    pool_t P; memset(&P, 0, sizeof P);
    int *first = malloc((size_t)(F + 1) * sizeof *first);
    long long *n_in_smpl = calloc((size_t)S, sizeof *n_in_smpl); int *nf_smpl = calloc((size_t)S, sizeof *nf_smpl), *lastf_smpl = malloc((size_t)S * sizeof *lastf_smpl);
    for (int s = 0; s < S; ++s) lastf_smpl[s] = -1;
    unsigned long long n_reads_tot = 0, n_cols_tot = 0; int n_tiles = 0, max_span = 0;
    for (int g = 0; g < n_reg; ++g) {
        const char *contig = reg[g].contig;
        if (!ref_name || strcmp(ref_name, contig)) { emit_wait(); free(ref); free(ref_name); ref = read_contig(ref_path, contig, &ref_len); ref_name = strdup(contig); }    /* (the writer may still read the old sequence) */
        reg_beg = reg[g].beg; reg_end = reg[g].end;
        if (reg_end <= reg_beg) continue;
        for (int f = 0; f < F; ++f) { reader_open(&rdr[f], kept_path[f]); for (int i = 0; i < win[f].n; ++i) lrec_free(win[f].r[i]); win[f].n = 0; }
        bcfgpu_depth_cap_reset(dcap);
        parsers_start(rdr, sfile, F, contig);
        for (int t0 = reg_beg; t0 < reg_end; ) {
            const int t1 = t0 + tile_cols < reg_end ? t0 + tile_cols : reg_end;
            /* The tile's pool is the reads that overlap the tile widened by `margin` on both sides: the realignment of an indel
             * candidate reads a read's bases and qualities up to INDEL_WINDOW_SIZE columns beyond the tile (bam2bcf_indel.c:38,
             * 174-175), and those qualities carry the mate-overlap tweak of a mate that may lie wholly outside the tile -- so a
             * read's mate has to be in the pool with it.  margin = the longest reference span seen so far + the window. */
            int margin = max_span + 64, margin_read;
            const double tr0 = want_timing ? now_s() : 0.;
            /* stage 1: the reads that start before the tile's end come off the files, through -C (BAQ + sam_cap_mapq + the
             * deferred filters) and the depth cap, into the live window */
            do {                                                             /* (again when a longer read widened the margin) */
            margin_read = margin;
            for (int f = 0; f < F; ++f) {
                lrec_t **b = NULL; int nb = 0, bcap = 0;
                for (;;) {
                    reader_fetch(&rdr[f]);
                    lrec_t *x = rdr[f].head;
                    if (!x || x->pos >= t1 + margin) break;
                    rdr[f].head = NULL;
                    if (x->end - x->pos > max_span) max_span = x->end - x->pos;
                    if (nb == bcap) { bcap = bcap ? 2 * bcap : 256; b = grow(b, (size_t)bcap * sizeof *b); }
                    b[nb++] = x;
                }
                if (!nb) { free(b); continue; }
                uint8_t *keep = malloc((size_t)nb);
                memset(keep, 1, (size_t)nb);
                if (cap_thres > 10 && !list_only) cap_batch(b, nb, S, ref, ref_len, keep);
                int m = 0;
                for (int i = 0; i < nb; ++i) { if (keep[i]) b[m++] = b[i]; else lrec_free(b[i]); }
                if (max_depth > 0 && m) {
                    /* bcfgpu_depth_cap_push wants the batch as arrays */
                    int32_t *pos = malloc((size_t)m * 4), *ncg = malloc((size_t)m * 4), *coff = malloc((size_t)m * 4), *fl = malloc((size_t)m * 4);
                    size_t nc = 0; for (int i = 0; i < m; ++i) nc += (size_t)b[i]->ncig;
                    uint32_t *cg = malloc((nc + 1) * 4);
                    nc = 0;
                    for (int i = 0; i < m; ++i) { pos[i] = b[i]->pos; ncg[i] = b[i]->ncig; coff[i] = (int32_t)nc; fl[i] = f; memcpy(cg + nc, b[i]->cig, (size_t)b[i]->ncig * 4); nc += (size_t)b[i]->ncig; }
                    bcfgpu_reads r0; memset(&r0, 0, sizeof r0);
                    r0.n_reads = m; r0.r_pos = pos; r0.r_ncig = ncg; r0.r_cig_off = coff; r0.cig = cg;
                    CHECK(bcfgpu_depth_cap_push(dcap, &r0, fl, keep));
                    free(pos); free(ncg); free(coff); free(fl); free(cg);
                } else memset(keep, 1, (size_t)nb);
                for (int i = 0; i < m; ++i) {
                    if (!keep[i]) { lrec_free(b[i]); continue; }
                    lwin_push(&win[f], b[i]);
                    const int s = b[i]->smpl;
                    ++n_in_smpl[s]; ++n_reads_tot;
                    if (lastf_smpl[s] != f) { ++nf_smpl[s]; lastf_smpl[s] = f; }
                }
                free(keep); free(b);
            }
            margin = max_span + 64;
            } while (margin > margin_read);
            const double tr1 = want_timing ? now_s() : 0.;
            t_read += tr1 - tr0;
            if (!list_only) {
                /* stage 2: the tile's pool = the reads of the window that overlap the tile, file after file */
                pool_clear(&P);
                for (int f = 0; f < F; ++f) {
                    first[f] = P.n;
                    for (int i = 0; i < win[f].n; ++i) { const lrec_t *x = win[f].r[i]; if (overlaps(x->pos, x->end, t0 - margin, t1 + margin)) pool_add_rec(&P, f, x); }
                }
                first[F] = P.n;
                if (want_timing) t_pool += now_s() - tr1;
                if (P.n) { process_tile(&P, first, F, S, contig, ref, ref_len, t0, t1); ++n_tiles; }
                else { emit_wait(); pending_flush(); }                      /* columns without a read: a gap ends a gVCF block (gvcf.c:131) */
                n_cols_tot += (unsigned long long)(t1 - t0);
            }
            /* reads that end before the next tile's pool begins are through */
            int left = 0, next_pos = OPEN_END, more = 0;
            for (int f = 0; f < F; ++f) {
                int m = 0;
                for (int i = 0; i < win[f].n; ++i) { lrec_t *x = win[f].r[i]; if ((x->end == x->pos ? x->pos + 1 : x->end) > t1 - margin) win[f].r[m++] = x; else lrec_free(x); }
                win[f].n = m; left += m;
                reader_fetch(&rdr[f]);
                if (rdr[f].head) { more = 1; if (rdr[f].head->pos < next_pos) next_pos = rdr[f].head->pos; }
            }
            t0 = t1;
            if (!left) {
                if (!more) break;                                           /* nothing left on this region: every file is through */
                if (next_pos - margin > t0) t0 = next_pos - margin;         /* a stretch without reads: on to the next read */
            }
        }
        parsers_stop(rdr, F);
        for (int f = 0; f < F; ++f) reader_close(&rdr[f]);
    }
    if (list_only) {
        /* --list-samples: what the host side decided, one line per output sample -- name, reads that enter the pileup, files they
         * come from -- and nothing else (the read-group plumbing, the filters and the depth cap run without a device) */
        for (int s = 0; s < S; ++s) printf("%s\t%lld\t%d\n", sample[s], n_in_smpl[s], nf_smpl[s]);
        return 0;
    }
    emit_finish();
    pending_flush();
    if (ctx) { uint32_t nw = 0; CHECK(bcfgpu_truncated_cells(ctx, &nw)); n_wide_cells += nw; }
    if (n_wide_cells) fprintf(stderr, "[bcfgpu_sam] note: %llu (site, sample) cells of more than 255 usable reads were left to the first-255 rule "
                                      "instead of errmod_cal's draw\n", n_wide_cells);
    fprintf(stderr, "%llu reads of %d samples, %llu overlapping pairs, %llu pileup entries in %llu columns (%d tiles of <= %d)\n",
            n_reads_tot, S, tot_pairs, tot_entries, n_cols_tot, n_tiles, tile_cols);
    if (want_timing) fprintf(stderr, "[bcfgpu_sam] seconds: reading and parsing the files %.3f, tile pools %.3f, device stages %.3f, writing records %.3f\n", t_read, t_pool, t_dev, t_emit);
    if (vio_close(fout)) DIE("%s\n", vio_error());
    if (ctx) bcfgpu_destroy(ctx);
    if (cap_ctx) bcfgpu_destroy(cap_ctx);
    bcfgpu_depth_cap_free(dcap);
    return 0;
usage:
    fprintf(stderr, "usage: bcfgpu_sam [-a TAG,..] [--gvcf INT,..] [-O v|z|u|b] [-o out] [-d INT] [-s LIST | -S FILE] [-G FILE] [--ignore-RG]\n"
                    "                  [-B | -E] [-6] [-x] [-A] [-q INT] [-Q INT] [-C INT] [--ff INT] [--rf INT] [-I] [-o INT] [-e INT] [-h INT] [-m INT] [-F FLOAT] [-p] [-L INT]\n"
                    "                  [--tile COLUMNS] [--gpus N]\n"
                    "                  -f ref.fa [-r CHR[:BEG[-END]],... | -R FILE] [-b FILE] file.sam|file.bam [...]      (as `bcftools mpileup`)\n"
                    "              or  ref.fa contig beg end file.sam|file.bam [...]                    (beg, end 1-based inclusive)\n");
    return 2;
}

/*  bcfgpu_sam.c -- `bcftools mpileup` over SAM / BAM files with every stage of the path on the device, in plain C over the
 *  C-ABI of include/bcfgpu.h (SNP and indel records).
 *
 *      bcfgpu_sam [options] <ref.fa> <contig> <beg> <end> <file.sam|file.bam> [...]             (beg, end 1-based inclusive)
 *      options: -a TAG,..  --gvcf INT,..  -O v|z|u|b  -o FILE  -d INT  -s LIST  -S FILE  -G FILE  --ignore-RG
 *               -B  -E  -A  -q INT  -Q INT  --ff INT  --rf INT                                  (as `bcftools mpileup`)
 *               --list-samples: print "sample <TAB> reads entering the pileup <TAB> files" and stop (no device needed)
 *
 *  What stays on the host is what mpileup.c and htslib's pileup do before any arithmetic: parsing (SAM text; BAM = BGZF +
 *  binary records), the read -> sample map of bam_sample.c (@RG SM, RG:Z tags, -s/-S/-G), the read filters of mplp_func
 *  (mpileup.c:183-246: unmapped, --rf/--ff flags, reads of dropped read groups, -q, orphans), the iterator's per-file depth
 *  cap (bcfgpu_depth_cap) and the pairing of overlapping mates (htslib overlap_push).  Then, each a call on the flat read pool:
 *      bcfgpu_pool_upload          the pool to HBM, once
 *      bcfgpu_pool_baq             BAQ (sam_prob_realn, mpileup.c:234)
 *      bcfgpu_pool_overlap_tweak   mate-overlap qualities (bam_mplp_init_overlaps, mpileup.c:640)
 *      bcfgpu_pool_pileup          the pileup columns of the region, built in HBM
 *      bcfgpu_mpileup        bcf_call_glfgen x samples + bcf_call_combine per column (mpileup.c:343-347)
 *  and for the columns where some read is followed by an indel (mpileup.c:354-365):
 *      bcfgpu_gap_prep_tile (bcf_call_gap_prep on the candidate columns, in HBM) -> bcfgpu_mpileup on its indel tile
 *  and the record loop writes what bcf_call2bcf (bam2bcf.c:756-906) puts in the record, in its order, under mpileup's header
 *  (mpileup.c:510-602), as VCF, bgzipped VCF or BCF (host/vcfio.c).  tests/test_c_host.py compares the whole output with the
 *  reference's goldens test/mpileup/mpileup.{1..11}.out and mpileup-SCR.out.
 *  Not here: CRAM input, sam_cap_mapq (-C; htslib's source is not in the reference tree and no golden exercises it), BED
 *  files (-l/-T), --illumina1.3+; a sample fed by several files has its reads merged by position (the reference appends file
 *  after file: same records unless a cell passes 255 usable reads, where errmod's subsampling depends on the order anyway).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <ctype.h>
#include <math.h>
#include <zlib.h>
#include <spawn.h>
#include <unistd.h>
#include <sys/types.h>
#include <sys/wait.h>
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
static int defer_mq_filters = 0;
static int reg_beg = 0, reg_end = 0x7fffffff;   /* the region: only reads that overlap it enter the pool, as htslib's region iterator hands them out */      /* -C: sam_cap_mapq comes between the flag filters and the -q / orphan filters (mpileup.c:234-241) */

static void pool_add(pool_t *P, int file, int smpl, const char *qname, int flag, int pos, int mapq, int rnext_same, int mpos, int isize,
                     const uint32_t *cig, int ncig, int lq, const uint8_t *seq16, const uint8_t *qual)
{
    {   /* [pos, endpos) against the region (a read without a reference span counts as one base: bam_endpos) */
        int e = pos;
        for (int c = 0; c < ncig; ++c) { const int op = cig[c] & 15; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) e += (int)(cig[c] >> 4); }
        if (e == pos) e = pos + 1;
        if (pos >= reg_end || e <= reg_beg) return;
    }
    if (P->n == P->cap) {
        P->cap = P->cap ? 2 * P->cap : 1024;
        #define G(a) P->a = grow(P->a, (size_t)P->cap * sizeof *P->a)
        G(pos); G(lq); G(flag); G(ncig); G(cig_off); G(seq_off); G(smpl); G(file); G(end); G(mpos); G(isize); G(rnext_same); G(mapq); G(has_zq); G(qname);
        #undef G
    }
    const int r = P->n++;
    P->qname[r] = strdup(qname);
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

typedef struct { uint8_t *d; size_t n; } blob_t;
static blob_t slurp(const char *path)
{
    blob_t b = { NULL, 0 };
    FILE *f = fopen(path, "rb");
    if (!f) DIE("cannot open %s\n", path);
    size_t cap = 0, got;
    do { if (b.n + (1 << 16) > cap) { cap = cap ? 2 * cap : 1 << 20; b.d = grow(b.d, cap + 1); } got = fread(b.d + b.n, 1, 1 << 16, f); b.n += got; } while (got);
    fclose(f);
    b.d[b.n] = 0;
    return b;
}
/* a BGZF file (a series of gzip members, SAM spec 4.1) inflated whole */
static blob_t bgzf_inflate(const blob_t in)
{
    blob_t out = { NULL, 0 };
    size_t cap = 0, at = 0;
    while (at < in.n) {
        z_stream z; memset(&z, 0, sizeof z);
        if (inflateInit2(&z, 15 + 16) != Z_OK) DIE("zlib: inflateInit2 failed\n");
        z.next_in = in.d + at; z.avail_in = (uInt)(in.n - at > (1u << 30) ? (1u << 30) : in.n - at);
        int rc;
        do {
            if (out.n + (1 << 16) > cap) { cap = cap ? 2 * cap : 1 << 20; out.d = grow(out.d, cap); }
            z.next_out = out.d + out.n; z.avail_out = (uInt)(cap - out.n);
            rc = inflate(&z, Z_NO_FLUSH);
            out.n = cap - z.avail_out;
            if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) DIE("zlib: corrupt BGZF block\n");
        } while (rc != Z_STREAM_END);
        at = (size_t)(z.next_in - in.d);
        inflateEnd(&z);
    }
    return out;
}
static int32_t le32(const uint8_t *p) { return (int32_t)((uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24); }

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

/* one input file: its header decides the samples (add_file), its reads on `contig` that pass the filters join the pool.
 * Returns 0 when the file is dropped (no usable read group).  h: the VCF header to receive ##contig lines, or NULL. */
static int read_file(const char *path, const char *contig, int file, sfile_t *sf, pool_t *P, vio_hdr *h)
{
    blob_t raw = slurp(path);
    uint8_t seqbuf[1 << 16], qualbuf[1 << 16];
    if (raw.n >= 4 && raw.d[0] == 0x1f && raw.d[1] == 0x8b) {
        /* ---- BAM (SAM spec 4.2) ---- */
        blob_t b = bgzf_inflate(raw);
        free(raw.d);
        if (b.n < 12 || memcmp(b.d, "BAM\1", 4)) DIE("%s: not a BAM file\n", path);
        const size_t l_text = (uint32_t)le32(b.d + 4);
        char *text = malloc(l_text + 1); memcpy(text, b.d + 8, l_text); text[l_text] = 0;
        size_t at = 8 + l_text;
        const int n_ref = le32(b.d + at); at += 4;
        int tid = -1;
        for (int i = 0; i < n_ref; ++i) {
            const int l_name = le32(b.d + at); const char *name = (const char *)b.d + at + 4;
            const int l_ref = le32(b.d + at + 4 + l_name);
            if (!strcmp(name, contig)) tid = i;
            at += 8 + (size_t)l_name;
            (void)l_ref;
        }
        if (!add_file(sf, path, text)) { free(text); free(b.d); return 0; }
        if (h) {
            size_t a2 = 12 + l_text;
            for (int i = 0; i < n_ref; ++i) {
                const int l_name = le32(b.d + a2);
                contig_line(h, (const char *)b.d + a2 + 4, l_name - 1, le32(b.d + a2 + 4 + l_name));
                a2 += 8 + (size_t)l_name;
            }
        }
        free(text);
        while (at + 36 <= b.n) {
            const uint8_t *r = b.d + at;
            const size_t bs = (uint32_t)le32(r);
            if (at + 4 + bs > b.n) DIE("%s: truncated BAM record\n", path);
            at += 4 + bs;
            const int refid = le32(r + 4), pos = le32(r + 8), l_name = r[12], mapq = r[13];
            const int n_cig = r[16] | r[17] << 8, flag = r[18] | r[19] << 8, l_seq = le32(r + 20);
            const int next_ref = le32(r + 24), next_pos = le32(r + 28), tlen = le32(r + 32);
            if (refid != tid || refid < 0 || !read_passes(flag, mapq)) continue;
            const char *qname = (const char *)r + 36;
            const uint8_t *cg = r + 36 + l_name, *sq = cg + 4 * (size_t)n_cig, *ql = sq + (l_seq + 1) / 2, *aux = ql + l_seq;
            const int smpl = sample_of(sf, bam_aux_rg(aux, r + 4 + bs));
            if (smpl < 0) continue;
            if (l_seq > (int)sizeof seqbuf) DIE("%s: read longer than %d\n", path, (int)sizeof seqbuf);
            uint32_t cig[4096];
            if (n_cig > 4096) DIE("%s: CIGAR with more than 4096 operations\n", path);
            for (int c = 0; c < n_cig; ++c) cig[c] = (uint32_t)le32(cg + 4 * c);
            for (int i = 0; i < l_seq; ++i) { seqbuf[i] = (sq[i >> 1] >> ((~i & 1) << 2)) & 15; qualbuf[i] = ql[i]; }
            pool_add(P, file, smpl, qname, flag, pos, mapq, next_ref == refid, next_pos, tlen, cig, n_cig, l_seq, seqbuf, qualbuf);
        }
        free(b.d);
        return 1;
    }
    /* ---- SAM text ---- */
    char *txt = (char *)raw.d, *body = txt;
    while (*body == '@') { char *e = strchr(body, '\n'); if (!e) { body += strlen(body); break; } body = e + 1; }
    {
        const char save = *body; *body = 0;
        const int ok = add_file(sf, path, txt);
        if (ok && h)
            for (const char *l = txt; l && *l; ) {                          /* @SQ -> ##contig (mpileup.c:533-540) */
                if (!strncmp(l, "@SQ\t", 4)) {
                    const char *e = strchr(l, '\n'), *sn = strstr(l, "\tSN:"), *ln = strstr(l, "\tLN:");
                    if (sn && ln && (!e || (sn < e && ln < e))) { sn += 4; contig_line(h, sn, (int)strcspn(sn, "\t\r\n"), atol(ln + 4)); }
                }
                l = strchr(l, '\n'); if (l) ++l;
            }
        *body = save;
        if (!ok) { free(raw.d); return 0; }
    }
    for (char *line = body; line && *line; ) {
        char *eol = strchr(line, '\n');
        if (eol) *eol = 0;
        char *next = eol ? eol + 1 : NULL;
        { size_t l = strlen(line); if (l && line[l - 1] == '\r') line[l - 1] = 0; }
        char *fld[12]; int nf = 0; char *rest = NULL;
        for (char *s = line; nf < 11 && s; ) { fld[nf++] = s; s = strchr(s, '\t'); if (s) *s++ = 0; rest = s; }
        line = next;
        if (nf < 11) continue;
        const int flag = atoi(fld[1]), mapq = atoi(fld[4]);
        if (strcmp(fld[2], contig) || !read_passes(flag, mapq)) continue;
        const char *rg = NULL;
        for (char *t = rest; t && *t; ) {                                    /* the optional fields: RG:Z:<id> */
            char *e = strchr(t, '\t'); if (e) *e = 0;
            if (!strncmp(t, "RG:Z:", 5)) { rg = t + 5; break; }
            t = e ? e + 1 : NULL;
        }
        const int smpl = sample_of(sf, rg);
        if (smpl < 0) continue;
        uint32_t cig[4096]; int ncig = 0;
        for (const char *c = fld[5]; *c && *c != '*'; ) {
            char *e; const long l = strtol(c, &e, 10);
            const char *ops = "MIDNSHP=X", *o = *e ? strchr(ops, *e) : NULL;
            if (!o || ncig == 4096) DIE("bad CIGAR in %s\n", path);
            cig[ncig++] = (uint32_t)l << 4 | (uint32_t)(o - ops);
            c = e + 1;
        }
        const int lq = fld[9][0] == '*' ? 0 : (int)strlen(fld[9]);
        if (lq > (int)sizeof seqbuf) DIE("%s: read longer than %d\n", path, (int)sizeof seqbuf);
        for (int i = 0; i < lq; ++i) { seqbuf[i] = (uint8_t)nt16_of(fld[9][i]); qualbuf[i] = fld[10][0] == '*' && !fld[10][1] ? 0xff : (uint8_t)(fld[10][i] - 33); }
        pool_add(P, file, smpl, fld[0], flag, atoi(fld[3]) - 1, mapq, !strcmp(fld[6], "=") || !strcmp(fld[6], fld[2]), atoi(fld[7]) - 1, atoi(fld[8]),
                 cig, ncig, lq, seqbuf, qualbuf);
    }
    free(raw.d);
    return 1;
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

/* the reads with keep[r] != 0 stay, in order; first[] (the pool is file-major) follows */
static void pool_keep(pool_t *P, int *first, int F, const uint8_t *keep)
{
    int m = 0;
    for (int s = 0, r = 0; s < F; ++s) {
        const int e = first[s + 1];
        first[s] = m;
        for (; r < e; ++r) {
            if (!keep[r]) { free(P->qname[r]); continue; }
            if (m != r) {
                #define MV(a) P->a[m] = P->a[r]
                MV(pos); MV(lq); MV(flag); MV(ncig); MV(cig_off); MV(seq_off); MV(smpl); MV(file); MV(end); MV(mpos); MV(isize); MV(rnext_same); MV(mapq); MV(has_zq); MV(qname);
                #undef MV
            }
            ++m;
        }
    }
    first[F] = m; P->n = m;
}

typedef struct { const uint8_t *pl, *sp; const uint16_t *dp4, *adf, *adr, *scr; const int32_t *qs; } planes_t;        /* host copies of bcfgpu_mplp_out's planes */

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
    for (int j = 0; j < 16; ++j) fprintf(LN, "%s%g", j ? "," : "", (double)(float)c->anno[j]);
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
    for (int s = 0; s < S; ++s) {
        fputc('\t', LN);
        for (int j = 0; j < x; ++j) fprintf(LN, "%s%d", j ? "," : "", pp->pl[(k * BCFGPU_MAX_PL + j) * Ss + s]);
        const uint16_t *d = pp->dp4 + k * 4 * Ss + s;                        /* FORMAT/DP, DV, DP4 from DP4 (bam2bcf.c:851-886) */
        if (fmt_flag & BCFGPU_FMT_DP) fprintf(LN, ":%d", d[0] + d[Ss] + d[2 * Ss] + d[3 * Ss]);
        if (fmt_flag & BCFGPU_FMT_DV) fprintf(LN, ":%d", d[2 * Ss] + d[3 * Ss]);
        if (fmt_flag & BCFGPU_FMT_SP) fprintf(LN, ":%d", pp->sp[k * Ss + s]);
        if (fmt_flag & BCFGPU_FMT_DP4) fprintf(LN, ":%d,%d,%d,%d", d[0], d[Ss], d[2 * Ss], d[3 * Ss]);
        for (int which = 0; which < 4; ++which) {                            /* ADF, ADR, AD, DPR */
            static const int bit[4] = { BCFGPU_FMT_ADF, BCFGPU_FMT_ADR, BCFGPU_FMT_AD, BCFGPU_FMT_DPR };
            if (!(fmt_flag & bit[which])) continue;
            fputc(':', LN);
            for (int j = 0; j < na; ++j) {
                const int f = pp->adf[(k * 5 + j) * Ss + s], r = pp->adr[(k * 5 + j) * Ss + s];
                fprintf(LN, "%s%d", j ? "," : "", which == 0 ? f : which == 1 ? r : f + r);
            }
        }
        if (fmt_flag & BCFGPU_FMT_SCR) fprintf(LN, ":%d", pp->scr[k * Ss + s]);
        if (fmt_flag & BCFGPU_FMT_QS) { fputc(':', LN); for (int j = 0; j < na; ++j) fprintf(LN, "%s%d", j ? "," : "", pp->qs[(k * 5 + j) * Ss + s]); }
    }
    end_record();
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

int main(int argc, char **argv)
{
    int32_t gv_range[16]; int gv_n = 0;                                       /* mpileup --gvcf INT,.. (gvcf.c:44-67) */
    char out_mode = 'v'; const char *out_path = "-"; int max_depth = 250;      /* mpileup -O, -o, -d (mpileup.c:937-950) */
    int baq_flag = 3, min_baseQ = 13, list_only = 0;
    int n_gpus = 1, shard = -1;                                               /* --gpus N: region shards, one process per shard; --shard K: this is shard K */
    char **argv0 = argv; const int argc0 = argc;
    int cap_thres = 0;                                                        /* mpileup -C (adjust-MQ), mpileup.c:938 */
    int openQ = 40, extQ = 20, tandemQ = 100, min_support = 1, per_sample_flt = 0, no_indels = 0, max_indel_depth = 250; double min_frac = 0.002;   /* mpileup.c:937-950 */
    while (argc > 2 && argv[1][0] == '-') {
        if (!strcmp(argv[1], "-a")) {                                         /* mpileup -a, mpileup.c:parse_format_flag */
            static const struct { const char *name; int bit; } tags[] = {
                { "DP", BCFGPU_FMT_DP }, { "DV", BCFGPU_FMT_DV }, { "SP", BCFGPU_FMT_SP }, { "DP4", BCFGPU_FMT_DP4 }, { "DPR", BCFGPU_FMT_DPR },
                { "AD", BCFGPU_FMT_AD }, { "ADF", BCFGPU_FMT_ADF }, { "ADR", BCFGPU_FMT_ADR }, { "INFO/DPR", BCFGPU_INFO_DPR },
                { "INFO/AD", BCFGPU_INFO_AD }, { "INFO/ADF", BCFGPU_INFO_ADF }, { "INFO/ADR", BCFGPU_INFO_ADR },
                { "SCR", BCFGPU_FMT_SCR }, { "FMT/SCR", BCFGPU_FMT_SCR }, { "FORMAT/SCR", BCFGPU_FMT_SCR }, { "INFO/SCR", BCFGPU_INFO_SCR },
                { "QS", BCFGPU_FMT_QS }, { "FMT/QS", BCFGPU_FMT_QS }, { "FORMAT/QS", BCFGPU_FMT_QS } };
            char *list = strdup(argv[2]);
            for (char *t = strtok(list, ","); t; t = strtok(NULL, ",")) {
                size_t i;
                for (i = 0; i < sizeof tags / sizeof tags[0]; ++i) if (!strcmp(t, tags[i].name)) { fmt_flag |= tags[i].bit; break; }
                if (i == sizeof tags / sizeof tags[0]) DIE("unknown tag %s\n", t);
            }
            free(list);
            argv += 2; argc -= 2;
        } else if (!strcmp(argv[1], "--gvcf")) {
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
        else if (!strcmp(argv[1], "-d")) { max_depth = atoi(argv[2]); argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "-s")) { add_samples(argv[2], 0); argv += 2; argc -= 2; }            /* mpileup.c:1058-1059,1087,1016 */
        else if (!strcmp(argv[1], "-S")) { add_samples(argv[2], 1); argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "-G")) { add_readgroups(argv[2]); argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "--ignore-RG")) { SM.ignore_rg = 1; ++argv; --argc; }
        else if (!strcmp(argv[1], "--list-samples")) { list_only = 1; ++argv; --argc; }
        else if (!strcmp(argv[1], "--gpus")) { n_gpus = atoi(argv[2]); argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "--shard")) { shard = atoi(argv[2]); argv += 2; argc -= 2; }                 /* host logic only: no device needed */
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
    if (argc < 6) {
        fprintf(stderr, "usage: bcfgpu_sam [-a TAG,..] [--gvcf INT,..] [-O v|z|u|b] [-o out] [-d INT] [-s LIST | -S FILE] [-G FILE] [--ignore-RG]\n"
                        "                  [-B | -E] [-A] [-q INT] [-Q INT] [-C INT] [--ff INT] [--rf INT] [-I] [-o INT] [-e INT] [-h INT] [-m INT] [-F FLOAT] [-p] [-L INT]\n"
                        "                  ref.fa contig beg end file.sam|file.bam [...]\n");
        return 2;
    }
    defer_mq_filters = cap_thres > 10;
    const char *contig = argv[2];
    const int beg = atoi(argv[3]) - 1, end = atoi(argv[4]);                 /* 0-based [beg, end) */
    if (n_gpus > 1 && shard < 0 && !list_only) {
        /* ---- several GPUs (the reference's -r regions + `bcftools concat`, mpileup.c:652-683, vcfconcat.c:420):
         * the region is cut into contiguous shards, a process per shard (shard k on device k mod the devices present), each
         * writing its records as VCF text; rank order is genomic order, so the files are written out one after the other, the
         * header from the first.  Sites are independent, and a shard reads the reads that overlap it: the records equal the
         * single-process run's.  (This process makes no device call before it starts the others.) ---- */
        if (gv_n) DIE("--gpus with --gvcf: a block would end at every shard boundary; not supported\n");
        const int n_sh = n_gpus < end - beg ? n_gpus : (end - beg > 0 ? end - beg : 1);
        pid_t *pid = malloc((size_t)n_sh * sizeof *pid);
        char (*tmp)[256] = malloc((size_t)n_sh * sizeof *tmp);
        for (int k = 0; k < n_sh; ++k) {
            const long b0 = beg + (long)(end - beg) * k / n_sh, e0 = beg + (long)(end - beg) * (k + 1) / n_sh;
            snprintf(tmp[k], sizeof tmp[k], "/tmp/bcfgpu_sam.%d.%d.vcf", (int)getpid(), k);
            char **av = malloc((size_t)(argc0 + 12) * sizeof *av);
            int n = 0;
            av[n++] = argv0[0];
            static char sb[16][24];
            snprintf(sb[0], 24, "%d", k); snprintf(sb[1], 24, "%ld", b0 + 1); snprintf(sb[2], 24, "%ld", e0);
            av[n++] = "--shard"; av[n++] = sb[0];
            const int first_pos = argc0 - argc + 1;              /* index of ref.fa in argv0 */
            for (int i = 1; i < first_pos; ++i) {                /* the options, without -O / -o FILE / --output / --gpus */
                const char *o = argv0[i];
                if (!strcmp(o, "--gpus") || !strcmp(o, "--output") || !strcmp(o, "-O")) { ++i; continue; }
                if (!strncmp(o, "-O", 2) && o[2]) continue;
                if (!strcmp(o, "-o")) { char *e; strtol(argv0[i + 1], &e, 10); if (*e) { ++i; continue; } }
                av[n++] = argv0[i];
            }
            av[n++] = "-O"; av[n++] = "v"; av[n++] = "--output"; av[n++] = tmp[k];
            av[n++] = argv0[first_pos]; av[n++] = argv0[first_pos + 1];
            char *rb = strdup(sb[1]), *re = strdup(sb[2]);
            av[n++] = rb; av[n++] = re;
            for (int i = first_pos + 4; i < argc0; ++i) av[n++] = argv0[i];
            av[n] = NULL;
            extern char **environ;
            if (posix_spawn(&pid[k], "/proc/self/exe", NULL, NULL, av, environ)) DIE("cannot start shard %d\n", k);
            free(av);
        }
        int bad = 0;
        for (int k = 0; k < n_sh; ++k) { int st = 0; if (waitpid(pid[k], &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st)) bad = 1; }
        if (bad) { for (int k = 0; k < n_sh; ++k) unlink(tmp[k]); DIE("a shard failed\n"); }
        fprintf(stderr, "[bcfgpu_sam] %d region shards, one process each; ordered emit on the host (records are text: no device gather)\n", n_sh);
        vio_file *f0 = vio_open_read(tmp[0]);
        if (!f0) DIE("%s\n", vio_error());
        hdr = vio_read_hdr(f0);
        if (!hdr) DIE("%s\n", vio_error());
        fout = vio_open_write(out_path, out_mode);
        if (!fout || vio_write_hdr(fout, hdr)) DIE("%s\n", vio_error());
        char *lb = NULL; size_t lcap = 0; int rr;
        while ((rr = vio_read_line(f0, hdr, &lb, &lcap)) > 0) if (lb[0] && vio_write_line(fout, hdr, lb)) DIE("%s\n", vio_error());
        if (rr < 0) DIE("%s\n", vio_error());
        vio_close(f0); unlink(tmp[0]);
        for (int k = 1; k < n_sh; ++k) {                         /* the other shards wrote records only */
            FILE *fk = fopen(tmp[k], "r");
            if (!fk) DIE("cannot read %s\n", tmp[k]);
            ssize_t nl;
            while ((nl = getline(&lb, &lcap, fk)) > 0) {
                if (lb[nl - 1] == '\n') lb[nl - 1] = 0;
                if (lb[0] && lb[0] != '#' && vio_write_line(fout, hdr, lb)) DIE("%s\n", vio_error());
            }
            fclose(fk); unlink(tmp[k]);
        }
        free(lb);
        if (vio_close(fout)) DIE("%s\n", vio_error());
        return 0;
    }
    reg_beg = beg; reg_end = end;
    int device = 0;                                                           /* shard k of --gpus runs on device k mod the devices present */
    if (shard > 0 && !list_only) { const int nd = bcfgpu_device_count(); device = nd > 0 ? shard % nd : 0; if (nd > 1) fprintf(stderr, "[bcfgpu_sam] shard %d on device %d of %d\n", shard, device, nd); }
    const int n_in = argc - 5, n_sites = end - beg;
    int ref_len = 0;
    char *ref = read_contig(argv[1], contig, &ref_len);
    pool_t P; memset(&P, 0, sizeof P);
    int *first = malloc((size_t)(n_in + 1) * sizeof *first);                  /* the pool is file-major: file f = [first[f], first[f+1]) */
    sfile_t *sfile = calloc((size_t)n_in, sizeof *sfile);
    /* ---- the VCF header, in mpileup's order (mpileup.c:510-602) ---- */
    hdr = vio_hdr_new();
    { char b[4096]; snprintf(b, sizeof b, "##reference=file://%s", argv[1]); vio_hdr_append(hdr, b); }
    int F = 0;                                                                /* files kept (mpileup.c:442-455 drops the others) */
    for (int i = 0; i < n_in; ++i) { first[F] = P.n; if (read_file(argv[5 + i], contig, F, &sfile[F], &P, F == 0 ? hdr : NULL)) ++F; }
    first[F] = P.n;
    const int S = SM.nsmpl;
    if (!F || !S) DIE("no sample left to call\n");
    char **sample = SM.smpl;
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
    /* ---- mpileup -C INT (mpileup.c:234-241): after BAQ, sam_cap_mapq lowers the mapping quality of reads with many
     * mismatches or drops them; then the -q and orphan filters, which read_passes() left for here ---- */
    if (cap_thres > 10 && P.n && !list_only) {
        bcfgpu_cfg c0; memset(&c0, 0, sizeof c0);
        c0.device = device; c0.n_smpl = S; c0.max_sites = 1; c0.max_reads = 64; c0.min_baseQ = min_baseQ; c0.capQ = 60; c0.n_grp = 1; c0.ploidy_max = 2;
        bcfgpu_ctx *cx = NULL;
        CHECK(bcfgpu_create(&c0, &cx));
        bcfgpu_reads r0; memset(&r0, 0, sizeof r0);
        r0.n_reads = P.n; r0.r_pos = P.pos; r0.r_lq = P.lq; r0.r_flag = P.flag; r0.r_ncig = P.ncig; r0.r_cig_off = P.cig_off;
        r0.r_seq_off = P.seq_off; r0.cig = P.cig; r0.seq16 = P.seq16; r0.qual = P.qual; r0.zq = P.zq; r0.r_has_zq = P.has_zq;
        /* the pool goes up once; BAQ and the cap run on that copy (the qualities sam_cap_mapq sees are BAQ's) */
        CHECK(bcfgpu_pool_upload(cx, &r0, NULL, P.mapq));
        if (baq_flag) CHECK(bcfgpu_pool_baq(cx, ref, ref_len, baq_flag, NULL));
        int32_t *capv = malloc((size_t)P.n * sizeof *capv);
        CHECK(bcfgpu_pool_cap_mapq(cx, ref, ref_len, cap_thres, capv));
        uint8_t *keep = malloc((size_t)P.n);
        for (int r = 0; r < P.n; ++r) {
            keep[r] = capv[r] >= 0;
            if (keep[r] && P.mapq[r] > capv[r]) P.mapq[r] = (uint8_t)capv[r];
            if (P.mapq[r] < min_mq) keep[r] = 0;
            if (!keep_orphans && (P.flag[r] & 1) && !(P.flag[r] & 2)) keep[r] = 0;
        }
        pool_keep(&P, first, F, keep);
        free(keep); free(capv);
        bcfgpu_destroy(cx);
    }
    /* ---- the per-file depth cap of the pileup iterator (mpileup -d, mpileup.c:646): reads it drops leave the pool ---- */
    if (max_depth > 0 && P.n) {
        bcfgpu_reads r0; memset(&r0, 0, sizeof r0);
        r0.n_reads = P.n; r0.r_pos = P.pos; r0.r_ncig = P.ncig; r0.r_cig_off = P.cig_off; r0.cig = P.cig;
        uint8_t *keep = malloc((size_t)P.n);
        CHECK(bcfgpu_depth_cap(&r0, P.file, F, max_depth, keep));
        pool_keep(&P, first, F, keep);
        free(keep);
    }

    if (list_only) {
        /* --list-samples: what the host side decided, one line per output sample -- name, reads that enter the pileup, files they
         * come from -- and nothing else (the read-group plumbing, the filters and the depth cap run without a device) */
        for (int s = 0; s < S; ++s) {
            int nr = 0, nf = 0, lastf = -1;
            for (int r = 0; r < P.n; ++r) if (P.smpl[r] == s) { ++nr; if (P.file[r] != lastf) { ++nf; lastf = P.file[r]; } }
            printf("%s\t%d\t%d\n", sample[s], nr, nf);
        }
        return 0;
    }
    bcfgpu_cfg cfg; memset(&cfg, 0, sizeof cfg);
    cfg.device = device; cfg.n_smpl = S; cfg.max_sites = n_sites; cfg.max_reads = (uint64_t)P.nbase + 64;   /* every base is in <= 1 column */
    cfg.min_baseQ = min_baseQ; cfg.capQ = 60; cfg.errmod_theta = 0.; cfg.fmt_flag = fmt_flag;
    cfg.call_theta = 1.1e-3; cfg.n_grp = 1; cfg.ploidy_max = 2;
    bcfgpu_ctx *ctx = NULL;
    CHECK(bcfgpu_create(&cfg, &ctx));

    /* mate overlaps: htslib pairs the reads inside one file's iterator (bam_mplp_init_overlaps, mpileup.c:640) */
    int32_t *pa = malloc((size_t)(P.n + 1) * sizeof *pa), *pb = malloc((size_t)(P.n + 1) * sizeof *pb);
    int np = 0;
    for (int f = 0; f < F; ++f) np += find_pairs(&P, first[f], first[f + 1], pa + np, pb + np);
    /* a sample fed by several files (mpileup.c:275-293 appends file after file): its reads merged by position, files in
     * order at equal positions, as bcfgpu_pileup wants them; the pairs follow their reads */
    {
        int sorted = 1;
        int32_t *last = malloc((size_t)S * sizeof *last);
        for (int s = 0; s < S; ++s) last[s] = INT32_MIN;
        for (int r = 0; r < P.n && sorted; ++r) { if (P.pos[r] < last[P.smpl[r]]) sorted = 0; last[P.smpl[r]] = P.pos[r]; }
        free(last);
        if (!sorted) {
            int32_t *ord = malloc((size_t)P.n * sizeof *ord), *tmp = malloc((size_t)P.n * sizeof *tmp), *inv = malloc((size_t)P.n * sizeof *inv);
            for (int r = 0; r < P.n; ++r) ord[r] = r;
            for (int w = 1; w < P.n; w *= 2) {                               /* bottom-up merge sort: stable */
                for (int lo = 0; lo < P.n; lo += 2 * w) {
                    const int mid = lo + w < P.n ? lo + w : P.n, hi = lo + 2 * w < P.n ? lo + 2 * w : P.n;
                    int i = lo, j = mid, k = lo;
                    while (i < mid && j < hi) tmp[k++] = P.pos[ord[j]] < P.pos[ord[i]] ? ord[j++] : ord[i++];
                    while (i < mid) tmp[k++] = ord[i++];
                    while (j < hi) tmp[k++] = ord[j++];
                }
                int32_t *t = ord; ord = tmp; tmp = t;
            }
            for (int r = 0; r < P.n; ++r) inv[ord[r]] = r;
            #define PERM(a) do { void *n_ = malloc((size_t)P.n * sizeof *P.a); for (int r = 0; r < P.n; ++r) memcpy((char *)n_ + (size_t)r * sizeof *P.a, &P.a[ord[r]], sizeof *P.a); \
                                 memcpy(P.a, n_, (size_t)P.n * sizeof *P.a); free(n_); } while (0)
            PERM(pos); PERM(lq); PERM(flag); PERM(ncig); PERM(cig_off); PERM(seq_off); PERM(smpl); PERM(file); PERM(end); PERM(mpos); PERM(isize);
            PERM(rnext_same); PERM(mapq); PERM(has_zq); PERM(qname);
            #undef PERM
            for (int i = 0; i < np; ++i) { pa[i] = inv[pa[i]]; pb[i] = inv[pb[i]]; }
            free(ord); free(tmp); free(inv);
        }
    }

    bcfgpu_reads rd; memset(&rd, 0, sizeof rd);
    rd.n_reads = P.n; rd.r_pos = P.pos; rd.r_lq = P.lq; rd.r_flag = P.flag; rd.r_ncig = P.ncig; rd.r_cig_off = P.cig_off;
    rd.r_seq_off = P.seq_off; rd.cig = P.cig; rd.seq16 = P.seq16; rd.qual = P.qual; rd.zq = P.zq; rd.r_has_zq = P.has_zq;

    /* the pool goes up once and stays in HBM: BAQ (new qualities and ZQ bytes for the reads it applies to; not with -B), the
     * mate-overlap tweak, the pileup of the region -- each on the copy the stage before left there */
    CHECK(bcfgpu_pool_upload(ctx, &rd, NULL, P.mapq));
    if (baq_flag) CHECK(bcfgpu_pool_baq(ctx, ref, ref_len, baq_flag, NULL));
    CHECK(bcfgpu_pool_overlap_tweak(ctx, np, pa, pb));
    /* the pileup of the region and the SNP pass */
    bcfgpu_tile tile;
    int32_t *col_n = malloc((size_t)(n_sites + 1) * sizeof *col_n);
    uint8_t *col_indel = malloc((size_t)n_sites + 1);
    CHECK(bcfgpu_pool_pileup(ctx, P.smpl, NULL, beg, end, ref, ref_len, &tile, col_n, col_indel));
    void *d_site, *d_pl, *d_dp4;
    bcfgpu_site *site = NULL;
    planes_t snp_planes;
    run_mpileup(ctx, &tile, n_sites, &site, &snp_planes, gv_n ? &d_site : NULL, &d_pl, &d_dp4);

    /* ---- indel records (mpileup.c:354-365): candidate columns -> bcf_call_gap_prep -> second pass with p->aux ---- */
    int nc = 0;
    int32_t *cand = malloc((size_t)(n_sites + 1) * sizeof *cand);
    for (int k = 0; k < n_sites; ++k)
        if (!no_indels && col_indel[k] && col_n[k] < max_indel_depth * S) cand[nc++] = k;     /* mpileup.c:354 */      /* max_indel_depth */
    bcfgpu_site *isite = NULL;
    planes_t ind_planes; memset(&ind_planes, 0, sizeof ind_planes); int32_t *live = NULL; int nlive = 0;
    int32_t *g_types = NULL, *g_maxins = NULL, *g_indelreg = NULL, *g_support = NULL; float *g_frac = NULL; int8_t *g_inscns = NULL;
    if (nc) {
        /* everything stays in HBM: the candidates' entries, the stage on the pool bcfgpu_pileup left there, p->aux straight
         * into the indel pass's tile over all candidate columns (the host pool is passed for its ZQ bytes) */
        bcfgpu_indel_in in; memset(&in, 0, sizeof in);
        in.n_sites = nc; in.n_smpl = S; in.ref = ref;
        in.openQ = openQ; in.extQ = extQ; in.tandemQ = tandemQ; in.min_support = min_support; in.per_sample_flt = per_sample_flt; in.min_frac = min_frac;
        bcfgpu_indel_out out; memset(&out, 0, sizeof out);
        int32_t *gret = malloc((size_t)nc * 4);
        g_types = malloc((size_t)nc * 16); g_inscns = malloc((size_t)nc * 4 * INSCNS_CAP); g_maxins = malloc((size_t)nc * 4);
        g_indelreg = malloc((size_t)nc * 4); g_support = malloc((size_t)nc * 4); g_frac = malloc((size_t)nc * 4);
        out.ret = gret; out.p_aux = NULL; out.indel_types = g_types; out.inscns = g_inscns; out.maxins = g_maxins;
        out.indelreg = g_indelreg; out.max_support = g_support; out.max_frac = g_frac;
        bcfgpu_tile ti;
        /* (with BAQ the ZQ bytes are the pool's, in HBM; with -B the reads' own tags, if any, go up from the host) */
        CHECK(bcfgpu_gap_prep_tile(ctx, nc, cand, baq_flag ? NULL : &rd, &in, &out, INSCNS_CAP, &ti));
        live = malloc((size_t)nc * 4);
        for (int i = 0; i < nc; ++i) if (gret[i] == 0) live[nlive++] = i;
        if (nlive) run_mpileup(ctx, &ti, nc, &isite, &ind_planes, NULL, NULL, NULL);      /* records of columns with ret < 0 are not used */
        free(gret);
    }

    {   /* cells cut to their first 255 usable reads (errmod_cal would draw a random 255: bcfgpu.h, bcfgpu_truncated_cells) */
        uint32_t ncut = 0;
        CHECK(bcfgpu_truncated_cells(ctx, &ncut));
        if (ncut) fprintf(stderr, "[bcfgpu_sam] warning: %u (site, sample) cells held more than 255 usable reads and were cut to their first 255 "
                                  "(DP, AD, QS count the kept reads only; bcftools subsamples at random inside errmod_cal): lower -d or split the sample's files\n", ncut);
    }
    /* ---- --gvcf: reference-only records collapse into blocks (gvcf_write, gvcf.c:88-226) on the planes still in HBM ---- */
    int32_t *gv_blk = NULL, *gv_dp = NULL; bcfgpu_gvcf_block *gv_block = NULL; uint8_t *gv_pl = NULL;
    if (gv_n) {
        int32_t *pos = malloc((size_t)n_sites * 4); uint8_t *brk = calloc((size_t)n_sites, 1);
        for (int k = 0; k < n_sites; ++k) { pos[k] = beg + k; if (col_n[k] == 0) brk[k] |= 2; }
        for (int j = 0; j < nlive; ++j) if (isite[live[j]].ret == 0) brk[cand[live[j]]] |= 1;      /* an indel record follows the SNP record */
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
        int32_t nb = 0;
        CHECK(bcfgpu_gvcf_blocks(ctx, &gi, &go, &nb));
        gv_blk = malloc((size_t)n_sites * 4); gv_block = malloc((size_t)(nb + 1) * sizeof *gv_block);
        gv_dp = malloc((size_t)(nb + 1) * S * 4); gv_pl = malloc((size_t)(nb + 1) * 3 * S);
        CHECK(bcfgpu_memcpy_d2h(ctx, gv_blk, d_blk, (size_t)n_sites * 4)); CHECK(bcfgpu_memcpy_d2h(ctx, gv_block, d_block, (size_t)nb * sizeof *gv_block));
        CHECK(bcfgpu_memcpy_d2h(ctx, gv_dp, d_gdp, (size_t)nb * S * 4)); CHECK(bcfgpu_memcpy_d2h(ctx, gv_pl, d_gpl, (size_t)nb * 3 * S));
        CHECK(bcfgpu_sync(ctx));
        bcfgpu_free(ctx, d_pos); bcfgpu_free(ctx, d_brk); bcfgpu_free(ctx, d_blk); bcfgpu_free(ctx, d_min); bcfgpu_free(ctx, d_block);
        bcfgpu_free(ctx, d_gdp); bcfgpu_free(ctx, d_gpl); free(pos); free(brk);
    }

    /* ---- the record loop: the SNP record of a column, then its indel record (mpileup.c:343-366) ---- */
    static const char *nt = "ACGTN";
    int jl = 0;
    for (int k = 0; k < n_sites; ++k) {
        if (col_n[k] == 0) continue;                                         /* no read: no record */
        const bcfgpu_site *c = &site[k];
        if (gv_blk && gv_blk[k] >= 0) {                                      /* inside a block: one line when the block ends */
            const int b = gv_blk[k];
            const bcfgpu_gvcf_block *B = &gv_block[b];
            if (B->last_site == k) {
                const bcfgpu_site *f = &site[B->first_site];
                fprintf(LN, "%s\t%d\t.\t%c\t<*>\t.\t.\t", contig, B->start_pos + 1, nt[f->ori_ref < 0 || f->ori_ref > 4 ? 4 : f->ori_ref]);
                if (B->start_pos + 1 < B->end1) fprintf(LN, "END=%d;", B->end1);                  /* gvcf.c:150-151 */
                fprintf(LN, "MinDP=%d;QS=%g,%g\tPL:DP", B->min_dp, (double)f->qsum[0], (double)f->qsum[1]);
                for (int s = 0; s < S; ++s)
                    fprintf(LN, "\t%d,%d,%d:%d", gv_pl[((size_t)b * 3) * S + s], gv_pl[((size_t)b * 3 + 1) * S + s], gv_pl[((size_t)b * 3 + 2) * S + s],
                           gv_dp[(size_t)b * S + s]);
                end_record();
            }
        } else {
        char als[64]; int o = 0;
        als[o++] = nt[c->ori_ref < 0 || c->ori_ref > 4 ? 4 : c->ori_ref]; als[o++] = '\t';
        for (int j = 1; j < c->n_alleles; ++j) {
            if (j > 1) als[o++] = ',';
            if (j == c->unseen) { memcpy(als + o, "<*>", 3); o += 3; } else als[o++] = nt[c->a[j]];
        }
        if (c->n_alleles < 2) als[o++] = '.';
        als[o] = 0;
        print_record(contig, beg + k + 1, als, "", c, &snp_planes, (size_t)k, S);
        }
        while (jl < nlive && cand[live[jl]] < k) ++jl;
        if (jl < nlive && cand[live[jl]] == k && isite[live[jl]].ret == 0) {
            /* REF / ALT of an indel record (bam2bcf.c:767-790) */
            const int i = live[jl], p = beg + k, ireg = g_indelreg[i], mi = g_maxins[i];
            char *txt = malloc((size_t)(5 * (ireg + mi + 8)) + 64), prefix[64];
            int t = 0;
            for (int j = 0; j <= ireg; ++j) txt[t++] = ref[p + j];
            txt[t++] = '\t';
            for (int a = 1; a < 4 && isite[i].a[a] >= 0; ++a) {
                const int ai = isite[i].a[a], ty = g_types[i * 4 + ai];
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
            print_record(contig, p + 1, txt, prefix, &isite[i], &ind_planes, (size_t)i, S);
            free(txt);
        }
    }
    fprintf(stderr, "%d reads of %d samples, %d overlapping pairs, %llu pileup entries in %d columns\n",
            P.n, S, np, (unsigned long long)tile.n_reads, n_sites);
    if (gv_n) { bcfgpu_free(ctx, d_site); bcfgpu_free(ctx, d_pl); bcfgpu_free(ctx, d_dp4); }
    if (vio_close(fout)) DIE("%s\n", vio_error());
    bcfgpu_destroy(ctx);
    return 0;
}

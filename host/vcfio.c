/*  vcfio.c -- see vcfio.h.  VCF header bookkeeping, BGZF framing (zlib) and the VCF text <-> BCF2 record codec. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <stdarg.h>
#include <math.h>
#include <zlib.h>
#include "vcfio.h"

static char g_err[512];
const char *vio_error(void) { return g_err; }
static int fail(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); return -1; }

/* ---- growing byte buffer ---- */
typedef struct { char *s; size_t l, m; } sbuf;
static void sb_need(sbuf *b, size_t n) { if (b->l + n + 1 > b->m) { b->m = (b->l + n + 1) * 2; b->s = realloc(b->s, b->m); if (!b->s) { fprintf(stderr, "out of memory\n"); exit(1); } } }
static void sb_put(sbuf *b, const void *p, size_t n) { sb_need(b, n); memcpy(b->s + b->l, p, n); b->l += n; b->s[b->l] = 0; }
static void sb_puts(sbuf *b, const char *s) { sb_put(b, s, strlen(s)); }
static void sb_putc(sbuf *b, char c) { sb_put(b, &c, 1); }
static void sb_printf(sbuf *b, const char *fmt, ...)
{
    char tmp[64]; va_list ap; va_start(ap, fmt); const int n = vsnprintf(tmp, sizeof tmp, fmt, ap); va_end(ap);
    sb_put(b, tmp, (size_t)(n < (int)sizeof tmp ? n : (int)sizeof tmp - 1));
}
static void sb_u32(sbuf *b, uint32_t v) { sb_put(b, &v, 4); }       /* little-endian hosts only (as the rest of the repo) */

/* ---- header ---- */
enum { T_FLAG = 0, T_INT = 1, T_FLOAT = 2, T_STR = 3, T_NONE = -1 };
typedef struct { char *id; size_t len; int info_type, fmt_type; } dict_ent;     /* len = strlen(id): a record looks a dozen keys up */
struct vio_hdr { char **line; int n_line, m_line; dict_ent *dict; int n_dict; char **ctg; int n_ctg; char **smpl; int n_smpl; };

static int dict_find(const vio_hdr *h, const char *id, size_t len)
{
    for (int i = 0; i < h->n_dict; ++i) if (h->dict[i].id && h->dict[i].len == len && !memcmp(h->dict[i].id, id, len)) return i;
    return -1;
}
static int dict_at(vio_hdr *h, const char *id, size_t len, int idx)
{
    int i = dict_find(h, id, len);
    if (i >= 0) return i;
    if (idx < 0) idx = h->n_dict;
    if (idx >= h->n_dict) {
        h->dict = realloc(h->dict, (size_t)(idx + 1) * sizeof *h->dict);
        for (int k = h->n_dict; k <= idx; ++k) { h->dict[k].id = NULL; h->dict[k].len = 0; h->dict[k].info_type = h->dict[k].fmt_type = T_NONE; }
        h->n_dict = idx + 1;
    }
    h->dict[idx].id = malloc(len + 1); memcpy(h->dict[idx].id, id, len); h->dict[idx].id[len] = 0; h->dict[idx].len = len;
    return idx;
}
/* value of KEY= inside the <...> of a meta line, or NULL */
static const char *attr(const char *ln, const char *key, size_t *len)
{
    const size_t kl = strlen(key);
    const char *p = strchr(ln, '<');
    int inq = 0;
    for (p = p ? p + 1 : NULL; p && *p; ++p) {
        if (*p == '"') inq = !inq;
        if (inq) continue;
        if ((p[-1] == '<' || p[-1] == ',') && !strncmp(p, key, kl) && p[kl] == '=') {
            const char *v = p + kl + 1, *e = v;
            if (*e == '"') { for (++e; *e && *e != '"'; ++e) {} if (*e) ++e; }
            else while (*e && *e != ',' && *e != '>') ++e;
            *len = (size_t)(e - v);
            return v;
        }
    }
    return NULL;
}
static int type_of(const char *t, size_t n)
{
    if (n == 4 && !strncmp(t, "Flag", 4)) return T_FLAG;
    if (n == 7 && !strncmp(t, "Integer", 7)) return T_INT;
    if (n == 5 && !strncmp(t, "Float", 5)) return T_FLOAT;
    return T_STR;
}
vio_hdr *vio_hdr_new(void)
{
    vio_hdr *h = calloc(1, sizeof *h);
    vio_hdr_append(h, "##fileformat=VCFv4.2");
    vio_hdr_append(h, "##FILTER=<ID=PASS,Description=\"All filters passed\">");
    return h;
}
void vio_hdr_free(vio_hdr *h)
{
    if (!h) return;
    for (int i = 0; i < h->n_line; ++i) free(h->line[i]);
    for (int i = 0; i < h->n_dict; ++i) free(h->dict[i].id);
    for (int i = 0; i < h->n_ctg; ++i) free(h->ctg[i]);
    for (int i = 0; i < h->n_smpl; ++i) free(h->smpl[i]);
    free(h->line); free(h->dict); free(h->ctg); free(h->smpl); free(h);
}
int vio_hdr_append(vio_hdr *h, const char *line)
{
    size_t n = strlen(line);
    while (n && (line[n - 1] == '\n' || line[n - 1] == '\r')) --n;
    if (n < 2 || line[0] != '#' || line[1] != '#') return fail("not a meta line: %.40s", line);
    char *ln = malloc(n + 1); memcpy(ln, line, n); ln[n] = 0;
    int idx_attr = -1;                                           /* ",IDX=n": the dictionary index a BCF header spells out; not kept in the text */
    {
        size_t xl; const char *ix = attr(ln, "IDX", &xl);
        if (ix && ix > ln + 5 && ix[-5] == ',') {
            idx_attr = atoi(ix);
            char *from = (char*)ix - 5;
            memmove(from, ix + xl, strlen(ix + xl) + 1);
            n = strlen(ln);
        }
    }
    const int is_info = !strncmp(ln, "##INFO=<", 8), is_fmt = !strncmp(ln, "##FORMAT=<", 10), is_flt = !strncmp(ln, "##FILTER=<", 10);
    if (is_info || is_fmt || is_flt) {
        size_t il, tl;
        const char *id = attr(ln, "ID", &il), *ty = attr(ln, "Type", &tl);
        if (!id) { free(ln); return fail("meta line without ID: %.60s", line); }
        /* the same tag declared twice: the first declaration stands (bcf_hdr_append skips duplicates) */
        const int known = dict_find(h, id, il);
        if (known >= 0 && ((is_info && h->dict[known].info_type != T_NONE) || (is_fmt && h->dict[known].fmt_type != T_NONE))) { free(ln); return 0; }
        if (!h->n_dict && !(il == 4 && !strncmp(id, "PASS", 4))) dict_at(h, "PASS", 4, 0);
        const int d = dict_at(h, id, il, idx_attr);
        if (is_info) h->dict[d].info_type = ty ? type_of(ty, tl) : T_STR;
        if (is_fmt) h->dict[d].fmt_type = ty ? type_of(ty, tl) : T_STR;
    } else if (!strncmp(ln, "##contig=<", 10)) {
        size_t il; const char *id = attr(ln, "ID", &il);
        if (id) {
            h->ctg = realloc(h->ctg, (size_t)(h->n_ctg + 1) * sizeof *h->ctg);
            h->ctg[h->n_ctg] = malloc(il + 1); memcpy(h->ctg[h->n_ctg], id, il); h->ctg[h->n_ctg][il] = 0; ++h->n_ctg;
        }
    }
    if (h->n_line == h->m_line) { h->m_line = h->m_line ? 2 * h->m_line : 32; h->line = realloc(h->line, (size_t)h->m_line * sizeof *h->line); }
    h->line[h->n_line++] = ln;
    return 0;
}
int vio_hdr_remove(vio_hdr *h, const char *kind, const char *id)
{
    char pre[64]; snprintf(pre, sizeof pre, "##%s=<", kind);
    for (int i = 0; i < h->n_line; ++i) {
        if (strncmp(h->line[i], pre, strlen(pre))) continue;
        size_t il; const char *v = attr(h->line[i], "ID", &il);
        if (!v || il != strlen(id) || strncmp(v, id, il)) continue;
        free(h->line[i]);
        memmove(h->line + i, h->line + i + 1, (size_t)(h->n_line - i - 1) * sizeof *h->line);
        --h->n_line;
        return 1;
    }
    return 0;
}
int vio_hdr_add_sample(vio_hdr *h, const char *name)
{
    h->smpl = realloc(h->smpl, (size_t)(h->n_smpl + 1) * sizeof *h->smpl);
    h->smpl[h->n_smpl++] = strdup(name);
    return 0;
}
int vio_hdr_subset(vio_hdr *h, int n, const int *keep)
{
    char **s = malloc((size_t)(n ? n : 1) * sizeof *s);
    for (int i = 0; i < n; ++i) { if (keep[i] < 0 || keep[i] >= h->n_smpl) { free(s); return fail("sample index out of range"); } s[i] = strdup(h->smpl[keep[i]]); }
    for (int i = 0; i < h->n_smpl; ++i) free(h->smpl[i]);
    free(h->smpl); h->smpl = s; h->n_smpl = n;
    return 0;
}
int vio_hdr_nsamples(const vio_hdr *h) { return h->n_smpl; }
const char *vio_hdr_sample(const vio_hdr *h, int i) { return h->smpl[i]; }
int vio_hdr_nlines(const vio_hdr *h) { return h->n_line; }
const char *vio_hdr_line(const vio_hdr *h, int i) { return h->line[i]; }
static char *hdr_text(const vio_hdr *h, size_t *len, int with_idx)
{
    sbuf b = {0, 0, 0};
    int ci = 0;
    for (int i = 0; i < h->n_line; ++i) {
        const char *ln = h->line[i];
        int idx = -1;
        if (with_idx) {                                          /* BCF: every dictionary line carries its index (hrec IDX of htslib) */
            size_t il; const char *id;
            if ((!strncmp(ln, "##INFO=<", 8) || !strncmp(ln, "##FORMAT=<", 10) || !strncmp(ln, "##FILTER=<", 10)) && (id = attr(ln, "ID", &il))) idx = dict_find(h, id, il);
            else if (!strncmp(ln, "##contig=<", 10)) idx = ci++;
        }
        const size_t l = strlen(ln);
        if (idx >= 0 && l && ln[l - 1] == '>') { sb_put(&b, ln, l - 1); sb_printf(&b, ",IDX=%d>", idx); }
        else sb_puts(&b, ln);
        sb_putc(&b, '\n');
    }
    sb_puts(&b, "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO");
    if (h->n_smpl) sb_puts(&b, "\tFORMAT");
    for (int i = 0; i < h->n_smpl; ++i) { sb_putc(&b, '\t'); sb_puts(&b, h->smpl[i]); }
    sb_putc(&b, '\n');
    if (len) *len = b.l;
    return b.s;
}
char *vio_hdr_text(const vio_hdr *h, size_t *len) { return hdr_text(h, len, 0); }
vio_hdr *vio_hdr_parse(const char *text, size_t len)
{
    vio_hdr *h = calloc(1, sizeof *h);
    const char *p = text, *end = text + len;
    while (p < end && *p) {
        const char *e = memchr(p, '\n', (size_t)(end - p));
        if (!e) e = end;
        size_t n = (size_t)(e - p);
        while (n && p[n - 1] == '\r') --n;
        char *ln = malloc(n + 1); memcpy(ln, p, n); ln[n] = 0;
        if (n > 1 && ln[0] == '#' && ln[1] == '#') { if (vio_hdr_append(h, ln)) { free(ln); vio_hdr_free(h); return NULL; } }
        else if (n > 1 && ln[0] == '#') {
            int col = 0;
            for (char *t = strtok(ln, "\t"); t; t = strtok(NULL, "\t"), ++col) if (col >= 9) vio_hdr_add_sample(h, t);
        }
        free(ln);
        p = e < end ? e + 1 : end;
    }
    if (!h->n_dict) dict_at(h, "PASS", 4, 0);
    /* PASS is always declared: a header without the line gets it right after ##fileformat (bcf_hdr_parse does the same) */
    int has_pass = 0;
    for (int i = 0; i < h->n_line; ++i) if (!strncmp(h->line[i], "##FILTER=<ID=PASS,", 18) || !strncmp(h->line[i], "##FILTER=<ID=PASS>", 18)) has_pass = 1;
    if (!has_pass) {
        if (h->n_line == h->m_line) { h->m_line = h->m_line ? 2 * h->m_line : 32; h->line = realloc(h->line, (size_t)h->m_line * sizeof *h->line); }
        const int at = h->n_line && !strncmp(h->line[0], "##fileformat", 12) ? 1 : 0;
        memmove(h->line + at + 1, h->line + at, (size_t)(h->n_line - at) * sizeof *h->line);
        h->line[at] = strdup("##FILTER=<ID=PASS,Description=\"All filters passed\">");
        ++h->n_line;
    }
    return h;
}

/* ---- streams: plain file, or gzip members (BGZF) ---- */
struct vio_file {
    FILE *fp; int is_write; char mode;
    sbuf blk;                       /* write: the pending uncompressed block; read: decompressed bytes not yet consumed */
    size_t rd;                      /* read: position in blk */
    int gz, bcf, eof;
    z_stream zs; unsigned char *zin;
    sbuf rec;
};
#define BGZF_BLOCK 0xff00

static int bgzf_flush_block(vio_file *f, const char *data, size_t n)
{
    unsigned char out[BGZF_BLOCK + 1024];
    z_stream z; memset(&z, 0, sizeof z);
    if (deflateInit2(&z, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return fail("deflateInit2");
    z.next_in = (unsigned char*)data; z.avail_in = (uInt)n; z.next_out = out + 18; z.avail_out = sizeof out - 26;
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) { deflateEnd(&z); return fail("deflate"); }
    const size_t clen = z.total_out;
    deflateEnd(&z);
    static const unsigned char hd[16] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0 };
    memcpy(out, hd, 16);
    const unsigned bsize = (unsigned)(clen + 25);
    out[16] = (unsigned char)(bsize & 0xff); out[17] = (unsigned char)(bsize >> 8);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, NULL, 0), (const unsigned char*)data, (uInt)n), isz = (uint32_t)n;
    memcpy(out + 18 + clen, &crc, 4); memcpy(out + 22 + clen, &isz, 4);
    return fwrite(out, 1, clen + 26, f->fp) == clen + 26 ? 0 : fail("write error");
}
static int out_bytes(vio_file *f, const void *p, size_t n)
{
    if (f->mode == 'v' || f->mode == 'u') return fwrite(p, 1, n, f->fp) == n ? 0 : fail("write error");   /* -Ou: the BCF stream as it is, no BGZF framing (htslib mode "wbu") */
    const char *c = p;
    while (n) {
        size_t k = BGZF_BLOCK - f->blk.l; if (k > n) k = n;
        sb_put(&f->blk, c, k); c += k; n -= k;
        if (f->blk.l == BGZF_BLOCK) { if (bgzf_flush_block(f, f->blk.s, f->blk.l)) return -1; f->blk.l = 0; }
    }
    return 0;
}
vio_file *vio_open_write(const char *path, char mode)
{
    if (mode != 'v' && mode != 'z' && mode != 'u' && mode != 'b') { fail("unknown output mode %c", mode); return NULL; }
    vio_file *f = calloc(1, sizeof *f);
    f->fp = strcmp(path, "-") ? fopen(path, "wb") : stdout;
    if (!f->fp) { fail("cannot open %s", path); free(f); return NULL; }
    f->is_write = 1; f->mode = mode; f->bcf = mode == 'u' || mode == 'b';
    return f;
}
int vio_close(vio_file *f)
{
    int rc = 0;
    if (!f) return 0;
    if (f->is_write) {
        if (f->mode != 'v' && f->mode != 'u') {
            if (f->blk.l && bgzf_flush_block(f, f->blk.s, f->blk.l)) rc = -1;
            /* the 28-byte empty block that marks the end of a BGZF file (SAM specification 4.1.2) */
            static const unsigned char eof[28] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
            if (fwrite(eof, 1, 28, f->fp) != 28) rc = -1;
        }
        if (fflush(f->fp)) rc = -1;
    } else if (f->gz) inflateEnd(&f->zs);
    if (f->fp && f->fp != stdout && f->fp != stdin) fclose(f->fp);
    free(f->blk.s); free(f->rec.s); free(f->zin); free(f);
    return rc;
}
/* read side: make at least `want` bytes available at blk.s + rd (fewer at the end of the input) */
static int fill(vio_file *f, size_t want)
{
    if (f->blk.s && (f->rd > (1u << 20) || f->rd == f->blk.l)) { memmove(f->blk.s, f->blk.s + f->rd, f->blk.l - f->rd); f->blk.l -= f->rd; f->rd = 0; }
    while (f->blk.l - f->rd < want && !f->eof) {
        if (!f->gz) {
            sb_need(&f->blk, 1 << 16);
            const size_t k = fread(f->blk.s + f->blk.l, 1, 1 << 16, f->fp);
            if (!k) f->eof = 1;
            f->blk.l += k;
        } else {
            if (!f->zs.avail_in) {
                f->zs.next_in = f->zin; f->zs.avail_in = (uInt)fread(f->zin, 1, 1 << 16, f->fp);
                if (!f->zs.avail_in) { f->eof = 1; break; }
            }
            sb_need(&f->blk, 1 << 17);
            f->zs.next_out = (unsigned char*)f->blk.s + f->blk.l; f->zs.avail_out = 1 << 17;
            const int r = inflate(&f->zs, Z_NO_FLUSH);
            f->blk.l += (1 << 17) - f->zs.avail_out;
            if (r == Z_STREAM_END) inflateReset(&f->zs);         /* the next gzip member (BGZF block) */
            else if (r != Z_OK && r != Z_BUF_ERROR) return fail("inflate: corrupt input");
        }
    }
    if (f->blk.s) f->blk.s[f->blk.l] = 0;
    return 0;
}
vio_file *vio_open_read(const char *path)
{
    vio_file *f = calloc(1, sizeof *f);
    f->fp = strcmp(path, "-") ? fopen(path, "rb") : stdin;
    if (!f->fp) { fail("cannot open %s", path); free(f); return NULL; }
    if (fill(f, 2)) { vio_close(f); return NULL; }
    if (f->blk.l >= 2 && (unsigned char)f->blk.s[0] == 0x1f && (unsigned char)f->blk.s[1] == 0x8b) {
        f->gz = 1; f->zin = malloc(1 << 16);
        memset(&f->zs, 0, sizeof f->zs);
        if (inflateInit2(&f->zs, 15 + 16) != Z_OK) { fail("inflateInit2"); vio_close(f); return NULL; }
        size_t n = f->blk.l; memcpy(f->zin, f->blk.s, n); f->blk.l = 0; f->rd = 0;
        n += fread(f->zin + n, 1, (1 << 16) - n, f->fp);
        f->zs.next_in = f->zin; f->zs.avail_in = (uInt)n; f->eof = 0;
    }
    if (fill(f, 5)) { vio_close(f); return NULL; }
    f->bcf = f->blk.l - f->rd >= 5 && !memcmp(f->blk.s + f->rd, "BCF\2\2", 5);
    return f;
}
vio_hdr *vio_read_hdr(vio_file *f)
{
    if (f->bcf) {
        if (fill(f, 9) || f->blk.l - f->rd < 9) { fail("truncated BCF header"); return NULL; }
        uint32_t l; memcpy(&l, f->blk.s + f->rd + 5, 4);
        if (fill(f, 9 + (size_t)l) || f->blk.l - f->rd < 9 + (size_t)l) { fail("truncated BCF header"); return NULL; }
        vio_hdr *h = vio_hdr_parse(f->blk.s + f->rd + 9, strnlen(f->blk.s + f->rd + 9, l));
        f->rd += 9 + l;
        return h;
    }
    /* VCF: every line that starts with '#' */
    sbuf t = {0, 0, 0};
    for (;;) {
        if (fill(f, 1)) { free(t.s); return NULL; }
        if (f->rd == f->blk.l || f->blk.s[f->rd] != '#') break;
        for (;;) {
            char *nl = memchr(f->blk.s + f->rd, '\n', f->blk.l - f->rd);
            if (nl) { sb_put(&t, f->blk.s + f->rd, (size_t)(nl + 1 - (f->blk.s + f->rd))); f->rd = (size_t)(nl + 1 - f->blk.s); break; }
            sb_put(&t, f->blk.s + f->rd, f->blk.l - f->rd); f->rd = f->blk.l;
            if (fill(f, 1) || f->rd == f->blk.l) break;
        }
    }
    vio_hdr *h = vio_hdr_parse(t.s ? t.s : "", t.l);
    free(t.s);
    return h;
}
int vio_write_hdr(vio_file *f, const vio_hdr *h)
{
    size_t n; char *t = hdr_text(h, &n, f->bcf);
    int rc;
    if (f->bcf) {
        const uint32_t l = (uint32_t)n + 1;
        rc = out_bytes(f, "BCF\2\2", 5) || out_bytes(f, &l, 4) || out_bytes(f, t, n + 1);
    } else rc = out_bytes(f, t, n);
    free(t);
    return rc;
}

/* ---- BCF2 typed values ---- */
enum { BT_NULL = 0, BT_INT8 = 1, BT_INT16 = 2, BT_INT32 = 3, BT_FLOAT = 5, BT_CHAR = 7 };
#define I_MISSING  INT32_MIN
#define I_VEND     (INT32_MIN + 1)
static const uint32_t F_MISSING = 0x7F800001u, F_VEND = 0x7F800002u;

static void enc_int1(sbuf *b, int32_t x);
static void enc_size(sbuf *b, int n, int type)
{
    if (n >= 15) { sb_putc(b, (char)(15 << 4 | type)); enc_int1(b, n); }
    else sb_putc(b, (char)(n << 4 | type));
}
static void enc_int1(sbuf *b, int32_t x)
{
    if (x == I_VEND) { enc_size(b, 1, BT_INT8); sb_putc(b, (char)0x81); }
    else if (x == I_MISSING) { enc_size(b, 1, BT_INT8); sb_putc(b, (char)0x80); }
    else if (x <= 127 && x >= -120) { enc_size(b, 1, BT_INT8); sb_putc(b, (char)x); }
    else if (x <= 32767 && x >= -32760) { int16_t z = (int16_t)x; enc_size(b, 1, BT_INT16); sb_put(b, &z, 2); }
    else { enc_size(b, 1, BT_INT32); sb_put(b, &x, 4); }
}
/* n values, wsize per vector (the size in the descriptor); smallest type that holds every value (sentinels aside) */
static void enc_vint(sbuf *b, int n, const int32_t *a, int wsize)
{
    if (n <= 0) { enc_size(b, 0, BT_NULL); return; }
    if (n == 1) { enc_int1(b, a[0]); return; }
    if (wsize <= 0) wsize = n;
    int32_t mx = INT32_MIN + 1, mn = INT32_MAX;
    for (int i = 0; i < n; ++i) { if (a[i] == I_MISSING || a[i] == I_VEND) continue; if (mx < a[i]) mx = a[i]; if (mn > a[i]) mn = a[i]; }
    if (mx <= 127 && mn >= -120) {
        enc_size(b, wsize, BT_INT8);
        for (int i = 0; i < n; ++i) sb_putc(b, a[i] == I_VEND ? (char)0x81 : a[i] == I_MISSING ? (char)0x80 : (char)a[i]);
    } else if (mx <= 32767 && mn >= -32760) {
        enc_size(b, wsize, BT_INT16);
        for (int i = 0; i < n; ++i) { int16_t z = a[i] == I_VEND ? (int16_t)0x8001 : a[i] == I_MISSING ? (int16_t)0x8000 : (int16_t)a[i]; sb_put(b, &z, 2); }
    } else { enc_size(b, wsize, BT_INT32); sb_put(b, a, (size_t)n * 4); }
}
static void enc_vchar(sbuf *b, const char *s, size_t n) { enc_size(b, (int)n, BT_CHAR); sb_put(b, s, n); }
static uint32_t float_bits(const char *s, size_t n)
{
    if (n == 1 && s[0] == '.') return F_MISSING;
    char tmp[64]; if (n > 63) n = 63; memcpy(tmp, s, n); tmp[n] = 0;
    const float f = strtof(tmp, NULL); uint32_t u; memcpy(&u, &f, 4); return u;
}
static int32_t int_val(const char *s, size_t n)
{
    if (n == 1 && s[0] == '.') return I_MISSING;
    char tmp[32]; if (n > 31) n = 31; memcpy(tmp, s, n); tmp[n] = 0;
    return (int32_t)strtol(tmp, NULL, 10);
}
static int count_char(const char *s, size_t n, char c) { int k = 0; for (size_t i = 0; i < n; ++i) k += s[i] == c; return k; }

/* one VCF text record -> BCF2 (l_shared, l_indiv, shared, indiv) appended to `out` */
/* vals != NULL: the per-sample columns are not in `line` (which ends with the FORMAT keys) but in integer arrays, n_keys of them,
 * vals[k][s * width[k] + j] (vio_write_record_int) */
static int encode_record(const vio_hdr *h, const char *line, sbuf *out, int n_keys, const int *width, const int32_t *const *vals)
{
    const char *fld[10]; size_t fl[10]; int nf = 0;
    const char *p = line;
    while (nf < 9) {
        const char *e = strchr(p, '\t');
        fld[nf] = p; fl[nf] = e ? (size_t)(e - p) : strlen(p); ++nf;
        if (!e) { p = NULL; break; }
        p = e + 1;
    }
    if (nf < 8) return fail("VCF record with fewer than 8 columns");
    const char *samples = (nf == 9 && p) ? p : NULL;
    sbuf sh = {0, 0, 0}, in = {0, 0, 0};
    int ci = -1;
    for (int i = 0; i < h->n_ctg; ++i) if (strlen(h->ctg[i]) == fl[0] && !memcmp(h->ctg[i], fld[0], fl[0])) { ci = i; break; }
    if (ci < 0) return fail("contig %.*s is not in the header", (int)fl[0], fld[0]);
    const int32_t pos = (int32_t)int_val(fld[1], fl[1]) - 1;
    int n_allele = 1;
    if (!(fl[4] == 1 && fld[4][0] == '.')) n_allele += 1 + count_char(fld[4], fl[4], ',');
    /* INFO first: rlen may come from END */
    int n_info = 0; int32_t rlen = (int32_t)fl[3];
    sbuf inf = {0, 0, 0};
    if (!(fl[7] == 1 && fld[7][0] == '.')) {
        const char *q = fld[7], *qe = fld[7] + fl[7];
        while (q < qe) {
            const char *e = memchr(q, ';', (size_t)(qe - q)); if (!e) e = qe;
            const char *eq = memchr(q, '=', (size_t)(e - q));
            const size_t kl = eq ? (size_t)(eq - q) : (size_t)(e - q);
            if (kl) {
                const int d = dict_find(h, q, kl);
                if (d < 0 || h->dict[d].info_type == T_NONE) { free(sh.s); free(inf.s); return fail("INFO tag %.*s is not defined in the header", (int)kl, q); }
                enc_int1(&inf, d);
                const char *v = eq ? eq + 1 : e; const size_t vl = (size_t)(e - v);
                const int ty = h->dict[d].info_type;
                if (ty == T_FLAG || !eq) enc_size(&inf, 0, BT_NULL);
                else if (ty == T_STR) enc_vchar(&inf, v, vl);
                else {
                    const int nv = 1 + count_char(v, vl, ',');
                    int32_t *a = malloc((size_t)nv * 4);
                    const char *t = v;
                    for (int i = 0; i < nv; ++i) {
                        const char *te = memchr(t, ',', (size_t)(e - t)); if (!te) te = e;
                        if (ty == T_INT) a[i] = int_val(t, (size_t)(te - t)); else { const uint32_t u = float_bits(t, (size_t)(te - t)); memcpy(&a[i], &u, 4); }
                        t = te + 1;
                    }
                    if (ty == T_INT) { enc_vint(&inf, nv, a, -1); if (kl == 3 && !strncmp(q, "END", 3) && nv == 1 && a[0] != I_MISSING && a[0] > pos) rlen = a[0] - pos; }
                    else { enc_size(&inf, nv, BT_FLOAT); sb_put(&inf, a, (size_t)nv * 4); }
                    free(a);
                }
                ++n_info;
            }
            q = e + 1;
        }
    }
    int32_t i32 = ci; sb_put(&sh, &i32, 4); sb_put(&sh, &pos, 4); sb_put(&sh, &rlen, 4);
    { const uint32_t qb = float_bits(fld[5], fl[5]); sb_u32(&sh, qb); }
    int n_fmt = 0, n_sample = 0;
    /* FORMAT keys */
    int fkey[64];
    if (nf == 9 && (samples || vals) && h->n_smpl) {
        n_sample = h->n_smpl;
        const char *q = fld[8], *qe = fld[8] + fl[8];
        while (q < qe && n_fmt < 64) {
            const char *e = memchr(q, ':', (size_t)(qe - q)); if (!e) e = qe;
            const int d = dict_find(h, q, (size_t)(e - q));
            if (d < 0 || h->dict[d].fmt_type == T_NONE) { free(sh.s); free(inf.s); return fail("FORMAT tag %.*s is not defined in the header", (int)(e - q), q); }
            fkey[n_fmt++] = d;
            q = e + 1;
        }
    }
    sb_u32(&sh, (uint32_t)n_allele << 16 | (uint32_t)n_info);
    sb_u32(&sh, (uint32_t)n_fmt << 24 | (uint32_t)n_sample);
    if (fl[2] == 1 && fld[2][0] == '.') enc_size(&sh, 0, BT_CHAR); else enc_vchar(&sh, fld[2], fl[2]);
    enc_vchar(&sh, fld[3], fl[3]);
    if (n_allele > 1) {
        const char *q = fld[4], *qe = fld[4] + fl[4];
        while (q <= qe) { const char *e = memchr(q, ',', (size_t)(qe - q)); if (!e) e = qe; enc_vchar(&sh, q, (size_t)(e - q)); q = e + 1; }
    }
    if (fl[6] == 1 && fld[6][0] == '.') enc_size(&sh, 0, BT_NULL);
    else {
        const int nv = 1 + count_char(fld[6], fl[6], ';');
        int32_t a[64]; const char *q = fld[6], *qe = fld[6] + fl[6];
        for (int i = 0; i < nv && i < 64; ++i) {
            const char *e = memchr(q, ';', (size_t)(qe - q)); if (!e) e = qe;
            const int d = dict_find(h, q, (size_t)(e - q));
            if (d < 0) { free(sh.s); free(inf.s); return fail("FILTER %.*s is not defined in the header", (int)(e - q), q); }
            a[i] = d; q = e + 1;
        }
        enc_vint(&sh, nv < 64 ? nv : 64, a, -1);
    }
    sb_put(&sh, inf.s ? inf.s : "", inf.l);
    free(inf.s);
    if (n_fmt && vals) {
        /* per-sample fields from the caller's arrays: what the text path below would have parsed out of the line */
        const int S = n_sample;
        if (n_fmt != n_keys) { free(sh.s); return fail("%d FORMAT keys, %d columns of values", n_fmt, n_keys); }
        for (int k = 0; k < n_fmt; ++k) {
            const int d = fkey[k];
            if (h->dict[d].fmt_type != T_INT || !strcmp(h->dict[d].id, "GT") || width[k] < 1) { free(sh.s); free(in.s); return fail("FORMAT/%s cannot be written from integer columns", h->dict[d].id); }
            enc_int1(&in, d);
            /* the vector's length is that of the sample with most values, as the text path sizes it: columns where every sample
             * ends early (VIO_INT_VEND) are not written */
            int w = 1;
            for (int s2 = 0; s2 < S && w < width[k]; ++s2) {
                const int32_t *v = vals[k] + (size_t)s2 * width[k];
                int c = width[k];
                while (c > 1 && v[c - 1] == VIO_INT_VEND) --c;
                if (c > w) w = c;
            }
            if (w == width[k]) { if (S * w == 1) enc_int1(&in, vals[k][0]); else enc_vint(&in, S * w, vals[k], w); }
            else {
                int32_t *t = malloc((size_t)S * (size_t)w * sizeof *t);
                for (int s2 = 0; s2 < S; ++s2) memcpy(t + (size_t)s2 * w, vals[k] + (size_t)s2 * width[k], (size_t)w * sizeof *t);
                if (S * w == 1) enc_int1(&in, t[0]); else enc_vint(&in, S * w, t, w);
                free(t);
            }
        }
    } else
    /* per-sample fields: the columns of every sample split once */
    if (n_fmt) {
        const int S = n_sample;
        const char **sv = malloc((size_t)S * (size_t)n_fmt * sizeof *sv); size_t *sl = calloc((size_t)S * (size_t)n_fmt, sizeof *sl);
        const char *q = samples;
        for (int s = 0; s < S; ++s) {
            const char *e = q ? strchr(q, '\t') : NULL, *qe = q ? (e ? e : q + strlen(q)) : NULL;
            const char *t = q;
            for (int k = 0; k < n_fmt; ++k) {
                if (!t || t > qe) { sv[(size_t)s * n_fmt + k] = "."; sl[(size_t)s * n_fmt + k] = 1; continue; }
                const char *te = memchr(t, ':', (size_t)(qe - t)); if (!te) te = qe;
                sv[(size_t)s * n_fmt + k] = t; sl[(size_t)s * n_fmt + k] = (size_t)(te - t);
                t = te < qe ? te + 1 : NULL;
            }
            q = e ? e + 1 : NULL;
        }
        for (int k = 0; k < n_fmt; ++k) {
            const int d = fkey[k], ty = h->dict[d].fmt_type;
            const int is_gt = !strcmp(h->dict[d].id, "GT");
            enc_int1(&in, d);
            if (is_gt) {
                int w = 1;
                for (int s = 0; s < S; ++s) { const char *v = sv[(size_t)s * n_fmt + k]; const size_t vl = sl[(size_t)s * n_fmt + k]; const int c = 1 + count_char(v, vl, '/') + count_char(v, vl, '|'); if (c > w) w = c; }
                int32_t *a = malloc((size_t)S * (size_t)w * 4);
                for (int s = 0; s < S; ++s) {
                    const char *v = sv[(size_t)s * n_fmt + k], *ve = v + sl[(size_t)s * n_fmt + k];
                    int j = 0, phased = 0;
                    while (v <= ve && j < w) {
                        const char *e = v; while (e < ve && *e != '/' && *e != '|') ++e;
                        const int32_t al = (e - v == 1 && v[0] == '.') || e == v ? -1 : int_val(v, (size_t)(e - v));
                        a[(size_t)s * w + j++] = (al + 1) << 1 | phased;
                        if (e >= ve) break;
                        phased = *e == '|'; v = e + 1;
                    }
                    for (; j < w; ++j) a[(size_t)s * w + j] = I_VEND;
                }
                enc_vint(&in, S * w, a, w);
                free(a);
            } else if (ty == T_STR) {
                size_t w = 0;
                for (int s = 0; s < S; ++s) if (sl[(size_t)s * n_fmt + k] > w) w = sl[(size_t)s * n_fmt + k];
                enc_size(&in, (int)w, BT_CHAR);
                for (int s = 0; s < S; ++s) { sb_put(&in, sv[(size_t)s * n_fmt + k], sl[(size_t)s * n_fmt + k]); for (size_t z = sl[(size_t)s * n_fmt + k]; z < w; ++z) sb_putc(&in, 0); }
            } else {
                int w = 1;
                for (int s = 0; s < S; ++s) { const int c = 1 + count_char(sv[(size_t)s * n_fmt + k], sl[(size_t)s * n_fmt + k], ','); if (c > w) w = c; }
                int32_t *a = malloc((size_t)S * (size_t)w * 4);
                for (int s = 0; s < S; ++s) {
                    const char *v = sv[(size_t)s * n_fmt + k], *ve = v + sl[(size_t)s * n_fmt + k];
                    int j = 0;
                    while (j < w) {
                        const char *e = memchr(v, ',', (size_t)(ve - v)); if (!e) e = ve;
                        if (ty == T_INT) a[(size_t)s * w + j] = int_val(v, (size_t)(e - v)); else { const uint32_t u = float_bits(v, (size_t)(e - v)); memcpy(&a[(size_t)s * w + j], &u, 4); }
                        ++j;
                        if (e >= ve) break;
                        v = e + 1;
                    }
                    for (; j < w; ++j) { if (ty == T_INT) a[(size_t)s * w + j] = I_VEND; else memcpy(&a[(size_t)s * w + j], &F_VEND, 4); }
                }
                if (ty == T_INT) { if (S * w == 1) { enc_int1(&in, a[0]); } else enc_vint(&in, S * w, a, w); }
                else { enc_size(&in, w, BT_FLOAT); sb_put(&in, a, (size_t)S * (size_t)w * 4); }
                free(a);
            }
        }
        free(sv); free(sl);
    }
    sb_u32(out, (uint32_t)sh.l); sb_u32(out, (uint32_t)in.l);
    sb_put(out, sh.s, sh.l); sb_put(out, in.s ? in.s : "", in.l);
    free(sh.s); free(in.s);
    return 0;
}

/* ---- BCF2 -> text ---- */
typedef struct { const unsigned char *p, *e; int bad; } rd_t;      /* bad: a read ran past e (a truncated or malformed record) */
#define RD_NEED(r, k) ((r)->e && (size_t)((r)->e - (r)->p) < (size_t)(k) ? ((r)->bad = 1, (r)->p = (r)->e, 0) : 1)
static int dec_size(rd_t *r, int *type);
static int32_t dec_int(rd_t *r, int type)
{
    int32_t v = 0;
    if (!RD_NEED(r, type == BT_INT8 ? 1 : type == BT_INT16 ? 2 : type == BT_INT32 ? 4 : 0)) return 0;
    if (type == BT_INT8) { v = (int8_t)*r->p; r->p += 1; if (v == -128) v = I_MISSING; else if (v == -127) v = I_VEND; }
    else if (type == BT_INT16) { int16_t z; memcpy(&z, r->p, 2); r->p += 2; v = z; if (z == INT16_MIN) v = I_MISSING; else if (z == INT16_MIN + 1) v = I_VEND; }
    else if (type == BT_INT32) { memcpy(&v, r->p, 4); r->p += 4; }
    return v;
}
static int dec_size(rd_t *r, int *type)
{
    *type = 0;
    if (!RD_NEED(r, 1)) return 0;
    const unsigned b = *r->p++;
    *type = b & 0xf;
    int n = b >> 4;
    if (n == 15) { int t2; const int n1 = dec_size(r, &t2); (void)n1; n = dec_int(r, t2); }
    return n;
}
static int dec_typed_int(rd_t *r) { int t; const int n = dec_size(r, &t); int32_t v = 0; for (int i = 0; i < n; ++i) v = dec_int(r, t); return v; }
static void put_float(sbuf *b, uint32_t u)
{
    if (u == F_MISSING) { sb_putc(b, '.'); return; }
    float f; memcpy(&f, &u, 4);
    sb_printf(b, "%g", (double)f);
}
static int tsz(int t) { return t == BT_INT8 || t == BT_CHAR ? 1 : t == BT_INT16 ? 2 : 4; }
static int decode_record(const vio_hdr *h, const unsigned char *sh, uint32_t lsh, const unsigned char *in, uint32_t lin, sbuf *b)
{
    rd_t r = { sh, sh + lsh, 0 };
    if (lsh < 24) return fail("truncated BCF record");
    #define SKIP(rr, k) do { const long long k_ = (long long)(k); if (k_ < 0 || !RD_NEED(rr, k_)) return fail("truncated BCF record"); (rr)->p += k_; } while (0)
    int32_t chrom, pos, rlen; uint32_t qb, nai, nfs;
    memcpy(&chrom, r.p, 4); memcpy(&pos, r.p + 4, 4); memcpy(&rlen, r.p + 8, 4); memcpy(&qb, r.p + 12, 4); memcpy(&nai, r.p + 16, 4); memcpy(&nfs, r.p + 20, 4);
    r.p += 24; (void)rlen;
    const int n_allele = (int)(nai >> 16), n_info = (int)(nai & 0xffff), n_fmt = (int)(nfs >> 24), n_sample = (int)(nfs & 0xffffff);
    if (chrom < 0 || chrom >= h->n_ctg) return fail("BCF record with an unknown contig index");
    sb_puts(b, h->ctg[chrom]); sb_printf(b, "\t%d\t", pos + 1);
    int t, n = dec_size(&r, &t);
    if (n < 0 || !RD_NEED(&r, n)) return fail("truncated BCF record");
    if (n) sb_put(b, r.p, (size_t)n); else sb_putc(b, '.');
    r.p += n; sb_putc(b, '\t');
    for (int a = 0; a < n_allele; ++a) {
        n = dec_size(&r, &t);
        if (a == 1) sb_putc(b, '\t'); else if (a > 1) sb_putc(b, ',');
        if (n < 0 || !RD_NEED(&r, n)) return fail("truncated BCF record");
        sb_put(b, r.p, (size_t)n); r.p += n;
    }
    if (n_allele == 1) sb_puts(b, "\t.");
    sb_putc(b, '\t');
    put_float(b, qb); sb_putc(b, '\t');
    n = dec_size(&r, &t);
    if (!n) sb_putc(b, '.');
    for (int i = 0; i < n; ++i) { const int32_t d = dec_int(&r, t); if (i) sb_putc(b, ';'); if (d < 0 || d >= h->n_dict || !h->dict[d].id) return fail("BCF record with an unknown FILTER index"); sb_puts(b, h->dict[d].id); }
    sb_putc(b, '\t');
    if (!n_info) sb_putc(b, '.');
    for (int i = 0; i < n_info; ++i) {
        const int d = dec_typed_int(&r);
        if (d < 0 || d >= h->n_dict || !h->dict[d].id) return fail("BCF record with an unknown INFO index");
        if (i) sb_putc(b, ';');
        sb_puts(b, h->dict[d].id);
        n = dec_size(&r, &t);
        if (!n) continue;                                        /* a flag */
        sb_putc(b, '=');
        if (n < 0 || !RD_NEED(&r, (long long)n * tsz(t))) return fail("truncated BCF record");
        if (t == BT_CHAR) { sb_put(b, r.p, strnlen((const char*)r.p, (size_t)n)); r.p += n; }
        else for (int k = 0; k < n; ++k) {
            if (t == BT_FLOAT) { uint32_t u; memcpy(&u, r.p, 4); r.p += 4; if (u == F_VEND) { r.p += 4 * (n - k - 1); break; } if (k) sb_putc(b, ','); put_float(b, u); }
            else { const int32_t v = dec_int(&r, t); if (v == I_VEND) { r.p += tsz(t) * (n - k - 1); break; } if (k) sb_putc(b, ','); if (v == I_MISSING) sb_putc(b, '.'); else sb_printf(b, "%d", v); }
        }
    }
    if (r.bad) return fail("truncated BCF record");
    if (!n_sample) return 0;
    /* FORMAT: the keys, then every sample's values from the per-key blocks */
    rd_t q = { in, in + lin, 0 };
    struct { int d, t, n; const unsigned char *p; } fm[64];
    sb_putc(b, '\t');
    for (int k = 0; k < n_fmt && k < 64; ++k) {
        fm[k].d = dec_typed_int(&q);
        if (fm[k].d < 0 || fm[k].d >= h->n_dict || !h->dict[fm[k].d].id) return fail("BCF record with an unknown FORMAT index");
        fm[k].n = dec_size(&q, &fm[k].t); fm[k].p = q.p;
        if (fm[k].n < 0) return fail("truncated BCF record");
        SKIP(&q, (long long)fm[k].n * tsz(fm[k].t) * n_sample);          /* the block of this key lies inside the record */
        if (k) sb_putc(b, ':');
        sb_puts(b, h->dict[fm[k].d].id);
    }
    if (q.bad) return fail("truncated BCF record");
    if (!n_fmt) sb_putc(b, '.');
    for (int s = 0; s < n_sample; ++s) {
        sb_putc(b, '\t');
        if (!n_fmt) sb_putc(b, '.');
        for (int k = 0; k < n_fmt && k < 64; ++k) {
            if (k) sb_putc(b, ':');
            rd_t v = { fm[k].p + (size_t)s * (size_t)fm[k].n * (size_t)tsz(fm[k].t), NULL, 0 };   /* (checked above: inside the record) */
            const int is_gt = !strcmp(h->dict[fm[k].d].id, "GT");
            if (fm[k].t == BT_CHAR) { const size_t l = strnlen((const char*)v.p, (size_t)fm[k].n); if (l) sb_put(b, v.p, l); else sb_putc(b, '.'); continue; }
            int printed = 0;
            for (int j = 0; j < fm[k].n; ++j) {
                if (fm[k].t == BT_FLOAT) { uint32_t u; memcpy(&u, v.p, 4); v.p += 4; if (u == F_VEND) break; if (j) sb_putc(b, ','); put_float(b, u); }
                else {
                    const int32_t x = dec_int(&v, fm[k].t);
                    if (x == I_VEND) break;
                    if (is_gt) { if (j) sb_putc(b, (x & 1) ? '|' : '/'); if (x >> 1) sb_printf(b, "%d", (x >> 1) - 1); else sb_putc(b, '.'); }
                    else { if (j) sb_putc(b, ','); if (x == I_MISSING) sb_putc(b, '.'); else sb_printf(b, "%d", x); }
                }
                ++printed;
            }
            if (!printed) sb_putc(b, '.');
        }
    }
    return 0;
}

int vio_write_line(vio_file *f, const vio_hdr *h, const char *line)
{
    if (!f->bcf) return out_bytes(f, line, strlen(line)) || out_bytes(f, "\n", 1);
    f->rec.l = 0;
    if (encode_record(h, line, &f->rec, 0, NULL, NULL)) return -1;
    return out_bytes(f, f->rec.s, f->rec.l);
}
int vio_write_record_int(vio_file *f, const vio_hdr *h, const char *head, int n_keys, const int *width, const int32_t *const *vals)
{
    f->rec.l = 0;
    if (f->bcf) {
        if (encode_record(h, head, &f->rec, n_keys, width, vals)) return -1;
        return out_bytes(f, f->rec.s, f->rec.l);
    }
    /* VCF text: the head, then every sample's columns (a vector ends at VIO_INT_VEND; an empty one and a missing value are '.');
     * written straight into the buffer, room for a sample's columns reserved at once */
    sb_put(&f->rec, head, strlen(head));
    size_t per = 2;
    for (int k = 0; k < n_keys; ++k) per += (size_t)width[k] * 12 + 1;
    for (int s = 0; s < h->n_smpl; ++s) {
        sb_need(&f->rec, per);
        char *o = f->rec.s + f->rec.l;
        *o++ = '\t';
        for (int k = 0; k < n_keys; ++k) {
            if (k) *o++ = ':';
            const int32_t *a = vals[k] + (size_t)s * (size_t)width[k];
            for (int j = 0; j < width[k]; ++j) {
                if (a[j] == I_VEND) { if (!j) *o++ = '.'; break; }
                if (j) *o++ = ',';
                if (a[j] == I_MISSING) { *o++ = '.'; continue; }
                char t[16]; int n = 0;
                uint32_t u = a[j] < 0 ? (uint32_t)(-(int64_t)a[j]) : (uint32_t)a[j];
                if (a[j] < 0) *o++ = '-';
                do { t[n++] = (char)('0' + u % 10); u /= 10; } while (u);
                while (n) *o++ = t[--n];
            }
        }
        f->rec.l = (size_t)(o - f->rec.s);
    }
    sb_putc(&f->rec, '\n');
    return out_bytes(f, f->rec.s, f->rec.l);
}
int vio_read_line(vio_file *f, const vio_hdr *h, char **line, size_t *cap)
{
    sbuf b = { *line, 0, *cap };
    int got = 0;
    if (f->bcf) {
        if (fill(f, 8)) return -1;
        if (f->blk.l - f->rd >= 8) {
            uint32_t ls, li; memcpy(&ls, f->blk.s + f->rd, 4); memcpy(&li, f->blk.s + f->rd + 4, 4);
            if (fill(f, 8 + (size_t)ls + li) || f->blk.l - f->rd < 8 + (size_t)ls + li) return fail("truncated BCF record");
            const unsigned char *p = (const unsigned char*)f->blk.s + f->rd + 8;
            if (decode_record(h, p, ls, p + ls, li, &b)) { *line = b.s; *cap = b.m; return -1; }
            f->rd += 8 + (size_t)ls + li;
            got = 1;
        }
    } else {
        for (;;) {
            if (fill(f, 1)) return -1;
            if (f->rd == f->blk.l) break;
            char *nl = memchr(f->blk.s + f->rd, '\n', f->blk.l - f->rd);
            const size_t k = nl ? (size_t)(nl - (f->blk.s + f->rd)) : f->blk.l - f->rd;
            sb_put(&b, f->blk.s + f->rd, k);
            got = 1;
            f->rd += k + (nl ? 1 : 0);
            if (nl) break;
        }
        while (b.l && b.s[b.l - 1] == '\r') b.s[--b.l] = 0;
        if (got && !b.l) { *line = b.s; *cap = b.m; return vio_read_line(f, h, line, cap); }   /* blank line */
    }
    if (!b.s) sb_need(&b, 1);
    b.s[b.l] = 0;
    *line = b.s; *cap = b.m;
    return got;
}

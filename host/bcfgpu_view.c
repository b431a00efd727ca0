/*  bcfgpu_view.c -- format conversion on the pipe boundary: VCF / bgzipped VCF / BCF2 in, any of them out (what
 *  `bcftools view [-O v|z|u|b]` does for a file that needs no filtering; used the way test.pl:1194-1195 uses it, to turn
 *  the BCF output of the drivers back into text).
 *
 *      bcfgpu_view [-O v|z|u|b] [-o out] [-H] [--int-columns] <in|->            -H: records only, no header
 *  --int-columns: a record whose per-sample columns are all integers (no GT, no '.' inside a vector but as a whole value) is written
 *  through vio_write_record_int -- the columns as integer arrays, the way host/bcfgpu_sam hands them over -- instead of as its text
 *  line; the output must not differ (tests/test_vcfio.py).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <ctype.h>
#include "vcfio.h"


/* 0: the record went out through vio_write_record_int; 1: it does not qualify (the caller writes the text line) */
static int int_pad = 0;         /* --int-columns-pad N: the columns handed over N values wider than any sample needs (a caller with fixed-width planes) */
static int write_int_columns(vio_file *fo, const vio_hdr *h, char *line, long *n_done)
{
    const int S = vio_hdr_nsamples(h);
    char *f8 = line; int tabs = 0;
    for (char *p = line; *p; ++p) if (*p == '\t' && ++tabs == 8) { f8 = p + 1; break; }
    if (tabs < 8 || !S) return 1;
    char *smp = strchr(f8, '\t');
    if (!smp) return 1;
    int nk = 1; for (char *p = f8; p < smp; ++p) if (*p == ':') ++nk;
    if (nk > 32 || (smp - f8 >= 2 && !strncmp(f8, "GT", 2) && (f8[2] == ':' || f8 + 2 == smp))) return 1;
    /* widths: the largest number of values any sample has for the key */
    int width[32]; for (int k = 0; k < nk; ++k) width[k] = 1;
    const char *q = smp + 1;
    for (int s = 0; s < S; ++s) {
        int k = 0, w = 1;
        for (;; ++q) {
            if (*q == ',') ++w;
            else if (*q == ':' || *q == '\t' || !*q) { if (k < nk && w > width[k]) width[k] = w; ++k; w = 1; if (*q != ':') break; }
            else if (!isdigit((unsigned char)*q) && *q != '-' && *q != '.') return 1;
        }
        if (k != nk) return 1;                                 /* trailing fields dropped: leave it to the text path */
        if (*q) ++q;
    }
    int32_t *col[32];
    for (int k = 0; k < nk; ++k) { width[k] += int_pad; col[k] = malloc((size_t)S * (size_t)width[k] * 4); }
    q = smp + 1;
    int ok = 1;
    for (int s = 0; s < S && ok; ++s)
        for (int k = 0; k < nk; ++k) {
            int j = 0;
            for (;;) {
                char *e; long v;
                if (*q == '.') { v = VIO_INT_MISSING; e = (char *)q + 1; } else v = strtol(q, &e, 10);
                if (e == q) { ok = 0; break; }
                col[k][(size_t)s * width[k] + j++] = (int32_t)v;
                q = e;
                if (*q == ',') { ++q; continue; }
                break;
            }
            for (; j < width[k]; ++j) col[k][(size_t)s * width[k] + j] = VIO_INT_VEND;
            if (*q == ':' || *q == '\t') ++q;
        }
    int rc = 1;
    if (ok) {
        *smp = 0;                                                /* the head: CHROM .. FORMAT */
        rc = vio_write_record_int(fo, h, line, nk, width, (const int32_t *const *)col) ? 1 : 0;
        if (rc) { *smp = '\t'; }
        else ++*n_done;
    }
    for (int k = 0; k < nk; ++k) free(col[k]);
    return rc;
}

int main(int argc, char **argv)
{
    char mode = 'v'; const char *out = "-", *in = NULL; int no_hdr = 0, int_cols = 0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-O") && i + 1 < argc) mode = argv[++i][0];
        else if (!strncmp(argv[i], "-O", 2) && argv[i][2]) mode = argv[i][2];
        else if (!strcmp(argv[i], "-o") && i + 1 < argc) out = argv[++i];
        else if (!strcmp(argv[i], "-H")) no_hdr = 1;
        else if (!strcmp(argv[i], "--int-columns")) int_cols = 1;
        else if (!strcmp(argv[i], "--int-columns-pad") && i + 1 < argc) { int_cols = 1; int_pad = atoi(argv[++i]); }
        else in = argv[i];
    }
    if (!in) { fprintf(stderr, "usage: bcfgpu_view [-O v|z|u|b] [-o out] [-H] <in|->\n"); return 2; }
    vio_file *fi = vio_open_read(in);
    if (!fi) { fprintf(stderr, "%s\n", vio_error()); return 1; }
    vio_hdr *h = vio_read_hdr(fi);
    if (!h) { fprintf(stderr, "%s\n", vio_error()); return 1; }
    vio_file *fo = vio_open_write(out, mode);
    if (!fo) { fprintf(stderr, "%s\n", vio_error()); return 1; }
    if (!no_hdr && vio_write_hdr(fo, h)) { fprintf(stderr, "%s\n", vio_error()); return 1; }
    char *line = NULL; size_t cap = 0; int rc;
    long n_int = 0;
    while ((rc = vio_read_line(fi, h, &line, &cap)) > 0) {
        if (int_cols && write_int_columns(fo, h, line, &n_int) == 0) continue;
        if (vio_write_line(fo, h, line)) { fprintf(stderr, "%s\n", vio_error()); return 1; }
    }
    if (int_cols) fprintf(stderr, "%ld records written from integer columns\n", n_int);
    if (rc < 0) { fprintf(stderr, "%s\n", vio_error()); return 1; }
    free(line);
    if (vio_close(fo)) { fprintf(stderr, "%s\n", vio_error()); return 1; }
    vio_close(fi); vio_hdr_free(h);
    return 0;
}

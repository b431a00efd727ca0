/*  vcfio.h -- VCF headers and VCF / BCF2 record streams for the host drivers: what htslib's vcf.c does on the pipe
 *  boundary of `bcftools mpileup | bcftools call` (hts_open / bcf_hdr_write / bcf_write of mpileup.c:502-602,288-316 and
 *  vcfcall.c:703-710; output modes of version.c:67-82).  Host-side record I/O only: nothing here computes.
 *
 *  The drivers build and consume records as VCF text lines; this module frames them:
 *      'v' plain VCF, 'z' bgzip-compressed VCF, 'u' uncompressed BCF2 (BGZF blocks of stored data), 'b' compressed BCF2.
 *  BCF2 records are encoded from / decoded to the text line with the header's dictionaries, following the BCF2.2
 *  specification (typed values, smallest integer type that holds a vector, string dictionary in header order with
 *  PASS = 0, contig dictionary in header order).
 */
#ifndef VCFIO_H
#define VCFIO_H
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

typedef struct vio_hdr vio_hdr;
typedef struct vio_file vio_file;

/* ---- header ---- */
vio_hdr *vio_hdr_new(void);                                     /* ##fileformat=VCFv4.2 + FILTER PASS, as bcf_hdr_init("w") */
vio_hdr *vio_hdr_parse(const char *text, size_t len);           /* meta lines + #CHROM line */
void vio_hdr_free(vio_hdr *h);
int  vio_hdr_append(vio_hdr *h, const char *line);              /* one "##..." line (with or without the newline) */
int  vio_hdr_remove(vio_hdr *h, const char *kind, const char *id);   /* kind: "INFO", "FORMAT", "FILTER"; the dictionary keeps the id */
int  vio_hdr_add_sample(vio_hdr *h, const char *name);
int  vio_hdr_subset(vio_hdr *h, int n, const int *keep);        /* the samples keep[0..n) of the current list, in that order */
int  vio_hdr_nsamples(const vio_hdr *h);
const char *vio_hdr_sample(const vio_hdr *h, int i);
char *vio_hdr_text(const vio_hdr *h, size_t *len);              /* malloc'ed: every meta line and the #CHROM line */
int  vio_hdr_nlines(const vio_hdr *h);
const char *vio_hdr_line(const vio_hdr *h, int i);              /* meta line i (no newline) */

/* ---- streams ---- */
vio_file *vio_open_write(const char *path, char mode);          /* path "-" = stdout; mode 'v', 'z', 'u' or 'b' */
int  vio_write_hdr(vio_file *f, const vio_hdr *h);
int  vio_write_line(vio_file *f, const vio_hdr *h, const char *line);   /* one VCF record as text, without the newline */
/* A record whose per-sample columns are integers, handed over as arrays instead of text: `head` = the first nine columns (CHROM .. FORMAT,
 * tab separated, no newline); for the k-th FORMAT key the n_samples x width[k] values vals[k][s * width[k] + j] (VIO_INT_VEND after a sample's
 * last value, VIO_INT_MISSING for '.').  The same bytes as vio_write_line of the full text line, as VCF and as BCF -- without a number being
 * printed and parsed back on the way into a BCF record. */
#define VIO_INT_MISSING INT32_MIN
#define VIO_INT_VEND    (INT32_MIN + 1)
int  vio_write_record_int(vio_file *f, const vio_hdr *h, const char *head, int n_keys, const int *width, const int32_t *const *vals);
vio_file *vio_open_read(const char *path);                      /* path "-" = stdin; VCF, bgzipped VCF or BCF2, detected */
vio_hdr *vio_read_hdr(vio_file *f);
int  vio_read_line(vio_file *f, const vio_hdr *h, char **line, size_t *cap);   /* 1: a record (as VCF text) in *line, 0: end, <0: error */
int  vio_close(vio_file *f);
const char *vio_error(void);

#endif

/*  bcfgpu_view.c -- format conversion on the pipe boundary: VCF / bgzipped VCF / BCF2 in, any of them out (what
 *  `bcftools view [-O v|z|u|b]` does for a file that needs no filtering; used the way test.pl:1194-1195 uses it, to turn
 *  the BCF output of the drivers back into text).
 *
 *      bcfgpu_view [-O v|z|u|b] [-o out] [-H] <in|->            -H: records only, no header
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "vcfio.h"

int main(int argc, char **argv)
{
    char mode = 'v'; const char *out = "-", *in = NULL; int no_hdr = 0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-O") && i + 1 < argc) mode = argv[++i][0];
        else if (!strncmp(argv[i], "-O", 2) && argv[i][2]) mode = argv[i][2];
        else if (!strcmp(argv[i], "-o") && i + 1 < argc) out = argv[++i];
        else if (!strcmp(argv[i], "-H")) no_hdr = 1;
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
    while ((rc = vio_read_line(fi, h, &line, &cap)) > 0)
        if (vio_write_line(fo, h, line)) { fprintf(stderr, "%s\n", vio_error()); return 1; }
    if (rc < 0) { fprintf(stderr, "%s\n", vio_error()); return 1; }
    free(line);
    if (vio_close(fo)) { fprintf(stderr, "%s\n", vio_error()); return 1; }
    vio_close(fi); vio_hdr_free(h);
    return 0;
}

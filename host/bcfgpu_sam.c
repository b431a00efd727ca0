/*  bcfgpu_sam.c -- `bcftools mpileup` over SAM files with every stage of the path on the device, in plain C over the
 *  C-ABI of include/bcfgpu.h (SNP and indel records; one sample per file, in file order).
 *
 *      bcfgpu_sam [-a TAG,TAG,..] <ref.fa> <contig> <beg> <end> <file.sam> [<file.sam> ...]     (beg, end 1-based inclusive)
 *
 *  What stays on the host is what mpileup.c and htslib's pileup do before any arithmetic: parsing, the read filters of
 *  mplp_func (mpileup.c:183-246: unmapped, secondary / QC-fail / duplicate, orphans) and the pairing of overlapping mates
 *  (htslib overlap_push).  Then, each a call on the flat read pool:
 *      bcfgpu_baq            BAQ (sam_prob_realn, mpileup.c:234)
 *      bcfgpu_overlap_tweak  mate-overlap qualities (bam_mplp_init_overlaps, mpileup.c:640)
 *      bcfgpu_pileup         the pileup columns of the region, built in HBM
 *      bcfgpu_mpileup        bcf_call_glfgen x samples + bcf_call_combine per column (mpileup.c:343-347)
 *  and for the columns where some read is followed by an indel (mpileup.c:354-365):
 *      bcfgpu_pileup_entries -> bcfgpu_gap_prep (bcf_call_gap_prep) -> bcfgpu_pileup_indel_tile -> bcfgpu_mpileup
 *  and the record loop prints, VCF-like, what bcf_call2bcf (bam2bcf.c:756-906) puts in the record:
 *      CHROM POS . REF ALT 0 . DP=..;I16=..;QS=..;VDB=..;SGB=..;RPB=..;MQB=..;MQSB=..;BQB=..;MQ0F=..   PL   <PL of every sample>
 *  (indel records: INDEL;IDV=..;IMF=.. in front; -a adds DP, DV, SP, DP4, AD, ADF, ADR, DPR, INFO/AD, INFO/ADF, INFO/ADR,
 *  INFO/DPR in bcf_call2bcf's order).  The lines are the data lines
 *  of `bcftools mpileup`'s VCF: tests/test_c_host.py compares them, byte for byte, with the reference's goldens
 *  test/mpileup/mpileup.{1,2,4,5}.out.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <ctype.h>
#include <math.h>
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
    int32_t *pos, *lq, *flag, *ncig, *cig_off, *seq_off, *smpl, *end, *mpos, *isize, *rnext_same;
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

/* one SAM file = one sample: reads on `contig` that pass mplp_func's filters, appended to the pool */
static void read_sam(const char *path, const char *contig, int smpl, pool_t *P, char **sample, vio_hdr *h)
{
    FILE *f = fopen(path, "r");
    if (!f) DIE("cannot open %s\n", path);
    static char line[1 << 20];
    *sample = NULL;
    while (fgets(line, sizeof line, f)) {
        if (line[0] == '@') {
            /* @SQ -> ##contig (mpileup.c:533-540, the first file's header); the first @RG's SM names the file's sample (bam_sample.c) */
            if (h && !strncmp(line, "@SQ\t", 4)) {
                const char *sn = strstr(line, "\tSN:"), *ln = strstr(line, "\tLN:");
                if (sn && ln) {
                    char buf[1024]; int l = 0; sn += 4;
                    while (sn[l] && sn[l] != '\t' && sn[l] != '\n') ++l;
                    snprintf(buf, sizeof buf, "##contig=<ID=%.*s,length=%d>", l, sn, atoi(ln + 4));
                    vio_hdr_append(h, buf);
                }
            } else if (!*sample && !strncmp(line, "@RG\t", 4)) {
                const char *sm = strstr(line, "\tSM:");
                if (sm) { int l = 0; sm += 4; while (sm[l] && sm[l] != '\t' && sm[l] != '\n' && sm[l] != '\r') ++l; *sample = malloc((size_t)l + 1); memcpy(*sample, sm, (size_t)l); (*sample)[l] = 0; }
            }
            continue;
        }
        char *fld[12]; int nf = 0;
        for (char *s = line; nf < 11 && s; ) { fld[nf++] = s; s = strchr(s, '\t'); if (s) *s++ = 0; }
        if (nf < 11) continue;
        { char *e = fld[10]; while (*e && *e != '\t' && *e != '\n' && *e != '\r') ++e; *e = 0; }
        const int flag = atoi(fld[1]);
        if (strcmp(fld[2], contig) || (flag & 4)) continue;
        if (flag & (256 | 512 | 1024)) continue;                             /* --ff UNMAP,SECONDARY,QCFAIL,DUP */
        if ((flag & 1) && !(flag & 2)) continue;                             /* orphans (no -A) */
        if (P->n == P->cap) {
            P->cap = P->cap ? 2 * P->cap : 1024;
            #define G(a) P->a = grow(P->a, (size_t)P->cap * sizeof *P->a)
            G(pos); G(lq); G(flag); G(ncig); G(cig_off); G(seq_off); G(smpl); G(end); G(mpos); G(isize); G(rnext_same); G(mapq); G(has_zq); G(qname);
            #undef G
        }
        const int r = P->n++;
        P->qname[r] = strdup(fld[0]);
        P->flag[r] = flag; P->pos[r] = atoi(fld[3]) - 1; P->mapq[r] = (uint8_t)atoi(fld[4]); P->smpl[r] = smpl; P->has_zq[r] = 0;
        P->rnext_same[r] = !strcmp(fld[6], "=") || !strcmp(fld[6], fld[2]);
        P->mpos[r] = atoi(fld[7]) - 1; P->isize[r] = atoi(fld[8]);
        /* CIGAR */
        P->cig_off[r] = (int32_t)P->ncigs; P->ncig[r] = 0;
        int x = P->pos[r];
        for (const char *c = fld[5]; *c && *c != '*'; ) {
            char *e; const long l = strtol(c, &e, 10);
            const char *ops = "MIDNSHP=X", *o = strchr(ops, *e);
            if (!o) DIE("bad CIGAR in %s\n", path);
            if (P->ncigs == P->cigcap) { P->cigcap = P->cigcap ? 2 * P->cigcap : 4096; P->cig = grow(P->cig, P->cigcap * 4); }
            P->cig[P->ncigs++] = (uint32_t)l << 4 | (uint32_t)(o - ops);
            ++P->ncig[r];
            if (*e == 'M' || *e == 'D' || *e == 'N' || *e == '=' || *e == 'X') x += (int)l;
            c = e + 1;
        }
        P->end[r] = x;
        /* SEQ / QUAL */
        const int lq = fld[9][0] == '*' ? 0 : (int)strlen(fld[9]);
        P->lq[r] = lq; P->seq_off[r] = (int32_t)P->nbase;
        if (P->nbase + lq + 1 > P->basecap) {
            P->basecap = (P->nbase + lq + 1) * 2;
            P->seq16 = grow(P->seq16, P->basecap); P->qual = grow(P->qual, P->basecap); P->zq = grow(P->zq, P->basecap);
        }
        for (int i = 0; i < lq; ++i) {
            P->seq16[P->nbase + i] = (uint8_t)nt16_of(fld[9][i]);
            P->qual[P->nbase + i] = fld[10][0] == '*' ? 0xff : (uint8_t)(fld[10][i] - 33);
            P->zq[P->nbase + i] = 0;
        }
        P->nbase += lq;
    }
    fclose(f);
    if (!*sample) *sample = strdup(path);                   /* no read group: the file name (bam_sample.c) */
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

typedef struct { const uint8_t *pl, *dp4, *adf, *adr, *sp; } planes_t;        /* host copies of bcfgpu_mplp_out's planes */

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
    const int x = na * (na + 1) / 2;
    const size_t Ss = (size_t)S;
    for (int s = 0; s < S; ++s) {
        fputc('\t', LN);
        for (int j = 0; j < x; ++j) fprintf(LN, "%s%d", j ? "," : "", pp->pl[(k * BCFGPU_MAX_PL + j) * Ss + s]);
        const uint8_t *d = pp->dp4 + k * 4 * Ss + s;                         /* FORMAT/DP, DV, DP4 from DP4 (bam2bcf.c:851-886) */
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
    }
    end_record();
}

int main(int argc, char **argv)
{
    int32_t gv_range[16]; int gv_n = 0;                                       /* mpileup --gvcf INT,.. (gvcf.c:44-67) */
    char out_mode = 'v'; const char *out_path = "-"; int max_depth = 250;      /* mpileup -O, -o, -d (mpileup.c:937-950) */
    while (argc > 2 && argv[1][0] == '-') {
        if (!strcmp(argv[1], "-a")) {                                         /* mpileup -a, mpileup.c:parse_format_flag */
            static const struct { const char *name; int bit; } tags[] = {
                { "DP", BCFGPU_FMT_DP }, { "DV", BCFGPU_FMT_DV }, { "SP", BCFGPU_FMT_SP }, { "DP4", BCFGPU_FMT_DP4 }, { "DPR", BCFGPU_FMT_DPR },
                { "AD", BCFGPU_FMT_AD }, { "ADF", BCFGPU_FMT_ADF }, { "ADR", BCFGPU_FMT_ADR }, { "INFO/DPR", BCFGPU_INFO_DPR },
                { "INFO/AD", BCFGPU_INFO_AD }, { "INFO/ADF", BCFGPU_INFO_ADF }, { "INFO/ADR", BCFGPU_INFO_ADR } };
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
        else if (!strcmp(argv[1], "-o")) { out_path = argv[2]; argv += 2; argc -= 2; }
        else if (!strcmp(argv[1], "-d")) { max_depth = atoi(argv[2]); argv += 2; argc -= 2; }
        else break;
    }
    if (argc < 6) { fprintf(stderr, "usage: bcfgpu_sam [-a TAG,..] [--gvcf INT,..] [-O v|z|u|b] [-o out] [-d INT] ref.fa contig beg end file.sam [file.sam ...]\n"); return 2; }
    const char *contig = argv[2];
    const int beg = atoi(argv[3]) - 1, end = atoi(argv[4]);                 /* 0-based [beg, end) */
    const int S = argc - 5, n_sites = end - beg;
    int ref_len = 0;
    char *ref = read_contig(argv[1], contig, &ref_len);
    pool_t P; memset(&P, 0, sizeof P);
    int *first = malloc((size_t)(S + 1) * sizeof *first);
    /* ---- the VCF header, in mpileup's order (mpileup.c:510-602) ---- */
    hdr = vio_hdr_new();
    { char b[4096]; snprintf(b, sizeof b, "##reference=file://%s", argv[1]); vio_hdr_append(hdr, b); }
    char **sample = malloc((size_t)S * sizeof *sample);
    for (int s = 0; s < S; ++s) { first[s] = P.n; read_sam(argv[5 + s], contig, s, &P, &sample[s], s == 0 ? hdr : NULL); }
    first[S] = P.n;
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
        HL(fmt_flag & BCFGPU_INFO_AD, "##INFO=<ID=AD,Number=R,Type=Integer,Description=\"Total allelic depths (high-quality bases)\">");
        HL(fmt_flag & BCFGPU_INFO_ADF, "##INFO=<ID=ADF,Number=R,Type=Integer,Description=\"Total allelic depths on the forward strand (high-quality bases)\">");
        HL(fmt_flag & BCFGPU_INFO_ADR, "##INFO=<ID=ADR,Number=R,Type=Integer,Description=\"Total allelic depths on the reverse strand (high-quality bases)\">");
        HL(gv_n, "##INFO=<ID=END,Number=1,Type=Integer,Description=\"End position of the variant described in this record\">");   /* gvcf.c:42-43 */
        HL(gv_n, "##INFO=<ID=MinDP,Number=1,Type=Integer,Description=\"Minimum per-sample depth in this gVCF block\">");
        #undef HL
        for (int s = 0; s < S; ++s) vio_hdr_add_sample(hdr, sample[s]);
    }
    fout = vio_open_write(out_path, out_mode);
    if (!fout || vio_write_hdr(fout, hdr)) DIE("%s\n", vio_error());
    LN = open_memstream(&ln_buf, &ln_len);
    if (!LN) DIE("open_memstream failed\n");
    /* ---- the per-file depth cap of the pileup iterator (mpileup -d, mpileup.c:646): reads it drops leave the pool ---- */
    if (max_depth > 0 && P.n) {
        bcfgpu_reads r0; memset(&r0, 0, sizeof r0);
        r0.n_reads = P.n; r0.r_pos = P.pos; r0.r_ncig = P.ncig; r0.r_cig_off = P.cig_off; r0.cig = P.cig;
        uint8_t *keep = malloc((size_t)P.n);
        CHECK(bcfgpu_depth_cap(&r0, P.smpl, S, max_depth, keep));
        int m = 0;
        for (int s = 0, r = 0; s < S; ++s) {
            const int e = first[s + 1];
            first[s] = m;
            for (; r < e; ++r) {
                if (!keep[r]) { free(P.qname[r]); continue; }
                if (m != r) {
                    #define MV(a) P.a[m] = P.a[r]
                    MV(pos); MV(lq); MV(flag); MV(ncig); MV(cig_off); MV(seq_off); MV(smpl); MV(end); MV(mpos); MV(isize); MV(rnext_same); MV(mapq); MV(has_zq); MV(qname);
                    #undef MV
                }
                ++m;
            }
        }
        first[S] = m; P.n = m;
        free(keep);
    }

    bcfgpu_cfg cfg; memset(&cfg, 0, sizeof cfg);
    cfg.device = 0; cfg.n_smpl = S; cfg.max_sites = n_sites; cfg.max_reads = (uint64_t)P.nbase + 64;   /* every base is in <= 1 column */
    cfg.min_baseQ = 13; cfg.capQ = 60; cfg.errmod_theta = 0.; cfg.fmt_flag = fmt_flag;
    cfg.call_theta = 1.1e-3; cfg.n_grp = 1; cfg.ploidy_max = 2;
    bcfgpu_ctx *ctx = NULL;
    CHECK(bcfgpu_create(&cfg, &ctx));

    bcfgpu_reads rd; memset(&rd, 0, sizeof rd);
    rd.n_reads = P.n; rd.r_pos = P.pos; rd.r_lq = P.lq; rd.r_flag = P.flag; rd.r_ncig = P.ncig; rd.r_cig_off = P.cig_off;
    rd.r_seq_off = P.seq_off; rd.cig = P.cig; rd.seq16 = P.seq16; rd.qual = P.qual; rd.zq = P.zq; rd.r_has_zq = P.has_zq;

    /* BAQ: new qualities for the reads it applies to */
    uint8_t *q1 = malloc(P.nbase + 1), *zq = malloc(P.nbase + 1), *q2 = malloc(P.nbase + 1);
    int32_t *ret = malloc((size_t)(P.n + 1) * sizeof *ret);
    CHECK(bcfgpu_baq(ctx, &rd, ref, ref_len, 3, q1, zq, ret));
    rd.qual = q1;
    for (int r = 0; r < P.n; ++r) P.has_zq[r] = ret[r] == 0;                 /* the "ZQ" tag sam_prob_realn leaves on the read */
    rd.zq = zq;
    /* mate overlaps, sample by sample */
    int32_t *pa = malloc((size_t)(P.n + 1) * sizeof *pa), *pb = malloc((size_t)(P.n + 1) * sizeof *pb);
    int np = 0;
    for (int s = 0; s < S; ++s) np += find_pairs(&P, first[s], first[s + 1], pa + np, pb + np);
    CHECK(bcfgpu_overlap_tweak(ctx, &rd, np, pa, pb, q2));
    rd.qual = q2;
    /* the pileup of the region and the SNP pass */
    bcfgpu_tile tile;
    int32_t *col_n = malloc((size_t)(n_sites + 1) * sizeof *col_n);
    uint8_t *col_indel = malloc((size_t)n_sites + 1);
    CHECK(bcfgpu_pileup(ctx, &rd, P.mapq, P.smpl, beg, end, ref, ref_len, &tile, col_n, col_indel));
    bcfgpu_mplp_out mo; memset(&mo, 0, sizeof mo);
    void *d_site, *d_pl, *d_dp4, *d_adf, *d_adr, *d_sp;
    const size_t nb_site = (size_t)n_sites * sizeof(bcfgpu_site), nb_pl = (size_t)n_sites * BCFGPU_MAX_PL * S, nb_dp4 = (size_t)n_sites * 4 * S,
                 nb_ad = (size_t)n_sites * 5 * S, nb_sp = (size_t)n_sites * S;
    CHECK(bcfgpu_malloc(ctx, nb_site, &d_site)); CHECK(bcfgpu_malloc(ctx, nb_pl, &d_pl)); CHECK(bcfgpu_malloc(ctx, nb_dp4, &d_dp4));
    CHECK(bcfgpu_malloc(ctx, nb_ad, &d_adf)); CHECK(bcfgpu_malloc(ctx, nb_ad, &d_adr)); CHECK(bcfgpu_malloc(ctx, nb_sp, &d_sp));
    CHECK(bcfgpu_memset(ctx, d_pl, 0, nb_pl)); CHECK(bcfgpu_memset(ctx, d_adf, 0, nb_ad)); CHECK(bcfgpu_memset(ctx, d_adr, 0, nb_ad));
    CHECK(bcfgpu_memset(ctx, d_sp, 0, nb_sp));
    mo.site = d_site; mo.pl = d_pl; mo.dp4 = d_dp4; mo.adf = d_adf; mo.adr = d_adr; mo.sp = d_sp;
    CHECK(bcfgpu_mpileup(ctx, &tile, &mo));
    CHECK(bcfgpu_sync(ctx));
    bcfgpu_site *site = malloc(nb_site);
    uint8_t *pl = malloc(nb_pl), *dp4 = malloc(nb_dp4), *adf = malloc(nb_ad), *adr = malloc(nb_ad), *sp = malloc(nb_sp);
    CHECK(bcfgpu_memcpy_d2h(ctx, site, d_site, nb_site)); CHECK(bcfgpu_memcpy_d2h(ctx, pl, d_pl, nb_pl));
    CHECK(bcfgpu_memcpy_d2h(ctx, dp4, d_dp4, nb_dp4)); CHECK(bcfgpu_memcpy_d2h(ctx, adf, d_adf, nb_ad));
    CHECK(bcfgpu_memcpy_d2h(ctx, adr, d_adr, nb_ad)); CHECK(bcfgpu_memcpy_d2h(ctx, sp, d_sp, nb_sp));
    const planes_t snp_planes = { pl, dp4, adf, adr, sp };
    CHECK(bcfgpu_sync(ctx));

    /* ---- indel records (mpileup.c:354-365): candidate columns -> bcf_call_gap_prep -> second pass with p->aux ---- */
    int nc = 0;
    int32_t *cand = malloc((size_t)(n_sites + 1) * sizeof *cand);
    int64_t cap = 0;
    for (int k = 0; k < n_sites; ++k)
        if (col_indel[k] && col_n[k] < 250 * S) { cand[nc++] = k; cap += col_n[k]; }      /* max_indel_depth */
    bcfgpu_site *isite = NULL;
    planes_t ind_planes = { NULL, NULL, NULL, NULL, NULL }; int32_t *live = NULL; int nlive = 0;
    int32_t *g_types = NULL, *g_maxins = NULL, *g_indelreg = NULL, *g_support = NULL; float *g_frac = NULL; int8_t *g_inscns = NULL;
    if (nc) {
        int32_t *so = malloc(((size_t)nc * S + 1) * sizeof *so), *pr = malloc((size_t)(cap + 1) * 4), *pq = malloc((size_t)(cap + 1) * 4),
                *pi = malloc((size_t)(cap + 1) * 4), *cpos = malloc((size_t)nc * 4);
        CHECK(bcfgpu_pileup_entries(ctx, nc, cand, so, pr, pq, pi, cap));
        for (int i = 0; i < nc; ++i) cpos[i] = beg + cand[i];
        bcfgpu_indel_in in; memset(&in, 0, sizeof in);
        in.n_sites = nc; in.n_smpl = S; in.pos = cpos; in.smpl_off = so; in.p_read = pr; in.p_qpos = pq; in.p_indel = pi; in.ref = ref;
        in.openQ = 40; in.extQ = 20; in.tandemQ = 100; in.min_support = 1; in.per_sample_flt = 0; in.min_frac = 0.002;   /* mpileup.c:937-950 */
        bcfgpu_indel_out out; memset(&out, 0, sizeof out);
        int32_t *gret = malloc((size_t)nc * 4);
        uint32_t *aux = malloc((size_t)(cap + 1) * 4);
        g_types = malloc((size_t)nc * 16); g_inscns = malloc((size_t)nc * 4 * INSCNS_CAP); g_maxins = malloc((size_t)nc * 4);
        g_indelreg = malloc((size_t)nc * 4); g_support = malloc((size_t)nc * 4); g_frac = malloc((size_t)nc * 4);
        out.ret = gret; out.p_aux = aux; out.indel_types = g_types; out.inscns = g_inscns; out.maxins = g_maxins;
        out.indelreg = g_indelreg; out.max_support = g_support; out.max_frac = g_frac;
        CHECK(bcfgpu_gap_prep(ctx, &rd, &in, &out, INSCNS_CAP));
        live = malloc((size_t)nc * 4);
        int32_t *lcols = malloc((size_t)nc * 4);
        uint32_t *laux = malloc((size_t)(cap + 1) * 4);
        int64_t nl = 0;
        for (int i = 0; i < nc; ++i)
            if (gret[i] == 0) {
                live[nlive] = i; lcols[nlive++] = cand[i];
                for (int e = so[(size_t)i * S]; e < so[(size_t)(i + 1) * S]; ++e) laux[nl++] = aux[e];
            }
        if (nlive) {
            bcfgpu_tile ti;
            CHECK(bcfgpu_pileup_indel_tile(ctx, nlive, lcols, laux, nl, &ti));
            const size_t b_pl = (size_t)nlive * BCFGPU_MAX_PL * S, b_dp4 = (size_t)nlive * 4 * S, b_ad = (size_t)nlive * 5 * S, b_sp = (size_t)nlive * S;
            void *d_is, *d_p[5];
            const size_t b_p[5] = { b_pl, b_dp4, b_ad, b_ad, b_sp };
            CHECK(bcfgpu_malloc(ctx, (size_t)nlive * sizeof(bcfgpu_site), &d_is));
            for (int j = 0; j < 5; ++j) { CHECK(bcfgpu_malloc(ctx, b_p[j], &d_p[j])); CHECK(bcfgpu_memset(ctx, d_p[j], 0, b_p[j])); }
            bcfgpu_mplp_out io; memset(&io, 0, sizeof io);
            io.site = d_is; io.pl = d_p[0]; io.dp4 = d_p[1]; io.adf = d_p[2]; io.adr = d_p[3]; io.sp = d_p[4];
            CHECK(bcfgpu_mpileup(ctx, &ti, &io));
            CHECK(bcfgpu_sync(ctx));
            isite = malloc((size_t)nlive * sizeof *isite);
            uint8_t *h_p[5];
            CHECK(bcfgpu_memcpy_d2h(ctx, isite, d_is, (size_t)nlive * sizeof *isite));
            for (int j = 0; j < 5; ++j) { h_p[j] = malloc(b_p[j]); CHECK(bcfgpu_memcpy_d2h(ctx, h_p[j], d_p[j], b_p[j])); }
            CHECK(bcfgpu_sync(ctx));
            ind_planes.pl = h_p[0]; ind_planes.dp4 = h_p[1]; ind_planes.adf = h_p[2]; ind_planes.adr = h_p[3]; ind_planes.sp = h_p[4];
            bcfgpu_free(ctx, d_is);
            for (int j = 0; j < 5; ++j) bcfgpu_free(ctx, d_p[j]);
        }
    }

    /* ---- --gvcf: reference-only records collapse into blocks (gvcf_write, gvcf.c:88-226) on the planes still in HBM ---- */
    int32_t *gv_blk = NULL, *gv_dp = NULL; bcfgpu_gvcf_block *gv_block = NULL; uint8_t *gv_pl = NULL;
    if (gv_n) {
        int32_t *pos = malloc((size_t)n_sites * 4); uint8_t *brk = calloc((size_t)n_sites, 1);
        for (int k = 0; k < n_sites; ++k) { pos[k] = beg + k; if (col_n[k] == 0) brk[k] |= 2; }
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
        if (jl < nlive && cand[live[jl]] == k && isite[jl].ret == 0) {
            /* REF / ALT of an indel record (bam2bcf.c:767-790) */
            const int i = live[jl], p = beg + k, ireg = g_indelreg[i], mi = g_maxins[i];
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
    fprintf(stderr, "%d reads of %d samples, %d overlapping pairs, %llu pileup entries in %d columns\n",
            P.n, S, np, (unsigned long long)tile.n_reads, n_sites);
    bcfgpu_free(ctx, d_site); bcfgpu_free(ctx, d_pl); bcfgpu_free(ctx, d_dp4); bcfgpu_free(ctx, d_adf); bcfgpu_free(ctx, d_adr); bcfgpu_free(ctx, d_sp);
    if (vio_close(fout)) DIE("%s\n", vio_error());
    bcfgpu_destroy(ctx);
    return 0;
}

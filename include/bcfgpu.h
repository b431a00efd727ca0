/*  bcfgpu.h -- C-ABI of the MI355X-native `bcftools mpileup | bcftools call -m` hot path.
 *
 *  This is the drop-in boundary.  Every entry point is `extern "C"`, takes
 *  plain pointers and sizes, returns 0 or a negative BCFGPU_E_* code and never
 *  exits the process (the reference's error()/exit(-1), version.c:43-50, is
 *  left to the CLI layer).  Each entry cites the reference interface it
 *  replaces (paths relative to the bcftools source tree):
 *
 *    bcfgpu_create / bcfgpu_destroy      <- bcf_call_init / bcf_call_destroy   bam2bcf.h:135-136 (bam2bcf.c:43-77)
 *                                           + mcall_init / mcall_destroy       call.h:135,139   (mcall.c:361-438)
 *    bcfgpu_pack_read                    <- the per-read accessors used by bcf_call_glfgen
 *                                           (bam2bcf.c:80-114 get_position, :189-220, mpileup.c:253-273)
 *    bcfgpu_mpileup                      <- bcf_callaux_clean + bcf_call_glfgen x n_smpl + bcf_call_combine
 *                                           for every site of a tile          bam2bcf.h:137-138,142
 *                                           (call sites mpileup.c:343-347 and :357-360)
 *    bcfgpu_gap_prep                     <- bcf_call_gap_prep                  bam2bcf.h:141 (bam2bcf_indel.c:99-470)
 *    bcfgpu_gap_prep_stats               <- (measurement only)
 *    bcfgpu_baq                          <- sam_prob_realn (htslib realn.c), call site mpileup.c:234
 *    bcfgpu_overlap_tweak                <- tweak_overlap_quality (htslib sam.c) of the pileup engine, switched on at mpileup.c:640
 *    bcfgpu_pileup                       <- the columns of bam_mplp_auto (htslib) as mpileup_reg() walks them, mpileup.c:320-347,
 *                                           with the per-read accessors of bcfgpu_pack_read
 *    bcfgpu_pileup_indel_tile            <- the second pileup pass of mpileup_reg() with p->aux set, mpileup.c:354-360
 *    bcfgpu_pileup_entries               <- the bam_pileup1_t fields bcf_call_gap_prep reads (b, qpos, indel), bam2bcf_indel.c:106-128
 *    bcfgpu_gvcf_blocks                  <- gvcf_write (gvcf.c:88-226) over the records of a tile, call site mpileup.c:297-316
 *    bcfgpu_mcall                        <- mcall()                            call.h:131 (mcall.c:1430-1684)
 *                                           incl. the per-record prologue of vcfcall.c:1096-1115
 *    bcfgpu_pipeline                     <- the `mpileup -Ou | call -m` pipe with PL/QS/I16 kept in HBM
 *
 *  Memory model: all bulk arrays are *device* pointers (HBM).  Hosts that do
 *  not link HIP use bcfgpu_malloc/free/memcpy_*.  Kernels are enqueued on the
 *  context's HIP stream; bcfgpu_sync() waits for them.
 *
 *  Data layout (one "tile" = a batch of pileup columns):
 *     site x sample x read  CSR:  plp_off[site*n_smpl + smpl] .. plp_off[..+1]
 *     index the per-read arrays `rd` (u32) and `epos` (u8) [and `aux` (u32) at
 *     indel sites].  Reads keep the order in which bcf_call_glfgen would see
 *     them (plp[] order, mpileup.c:275-293).
 */
#ifndef BCFGPU_H
#define BCFGPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BCFGPU_VERSION 2          /* 2: DP4/ADF/ADR/SCR planes are u16, the QS planes i32 (cells of any depth keep their counts) */

/* error codes */
#define BCFGPU_OK          0
#define BCFGPU_E_ARG      -1   /* bad argument / NULL pointer                       */
#define BCFGPU_E_NOMEM    -2   /* host or device allocation failed                  */
#define BCFGPU_E_HIP      -3   /* a HIP runtime call failed (see bcfgpu_last_error) */
#define BCFGPU_E_DEPTH    -4   /* a (site,sample) cell holds more pileup entries than the context's scratch for over-deep cells
                                  (several thousand stay in LDS, more go through a global scratch sized from max_reads), or more than
                                  65535 reads of one base and strand (the u16 count planes).  Cells with more than 255 *usable* reads are
                                  not an error: see bcfgpu_truncated_cells */
#define BCFGPU_E_NODEV    -5   /* no HIP device: the product path has no CPU fallback  */
#define BCFGPU_E_RANGE    -6   /* tile larger than the context's capacity           */

/* annotation flags: values identical to B2B_* (bam2bcf.h:46-62) */
#define BCFGPU_FMT_DP      (1<<0)
#define BCFGPU_FMT_SP      (1<<1)
#define BCFGPU_FMT_DV      (1<<2)
#define BCFGPU_FMT_DP4     (1<<3)
#define BCFGPU_FMT_DPR     (1<<4)
#define BCFGPU_INFO_DPR    (1<<5)
#define BCFGPU_FMT_AD      (1<<6)
#define BCFGPU_FMT_ADF     (1<<7)
#define BCFGPU_FMT_ADR     (1<<8)
#define BCFGPU_INFO_AD     (1<<9)
#define BCFGPU_INFO_ADF    (1<<10)
#define BCFGPU_INFO_ADR    (1<<11)
#define BCFGPU_INFO_SCR    (1<<12)
#define BCFGPU_FMT_SCR     (1<<13)
#define BCFGPU_INFO_VDB    (1<<14)
#define BCFGPU_INFO_RPB    (1<<15)
#define BCFGPU_FMT_QS      (1<<16)

/* call flags: values identical to CALL_* (call.h:32-39) */
#define BCFGPU_CALL_KEEPALT   1
#define BCFGPU_CALL_VARONLY   (1<<1)
#define BCFGPU_CALL_FMT_PV4   (1<<5)
#define BCFGPU_CALL_FMT_GQ    (1<<6)
#define BCFGPU_CALL_FMT_GP    (1<<7)

#define BCFGPU_MAX_ALLELES 5          /* B2B_MAX_ALLELES, bam2bcf.h:64 */
#define BCFGPU_MAX_PL      15         /* 5*(5+1)/2 */
#define BCFGPU_NPOS        100        /* bca->npos,  bam2bcf.c:55 */
#define BCFGPU_NQUAL       60         /* bca->nqual, bam2bcf.c:58 */
#define BCFGPU_MAX_DEPTH   255        /* reads of a (site,sample) cell errmod_cal takes without subsampling (htslib errmod.c); every count,
                                         QS sum, I16 sum and histogram is over ALL usable reads whatever their number (bam2bcf.c:203-252) */

/* sentinels, identical to htslib's bcf_int32_missing / bcf_int32_vector_end */
#define BCFGPU_INT32_MISSING     (INT32_MIN)
#define BCFGPU_INT32_VECTOR_END  (INT32_MIN+1)
/* genotype encodings in the int8 GT planes */
#define BCFGPU_GT_MISSING     (-1)
#define BCFGPU_GT_VECTOR_END  (-2)

/* ---- packed read record `rd` (u32), one per pileup entry ------------------
 *   bits  0..7   baseQ    bam_get_qual(b)[qpos]                       bam2bcf.c:191 (:183 at indels)
 *   bits  8..15  mapQ     b->core.qual (255 kept; DEF_MAPQ applied on device, bam2bcf.c:196)
 *   bits 16..19  nt16     bam_seqi(bam_get_seq(b), qpos)              bam2bcf.c:189,241
 *   bit  20      is_rev   bam_is_rev(b)
 *   bit  21      softclip PLP_HAS_SOFT_CLIP(p->cd.i)                  bam2bcf.h:66
 *   bit  22      is_del   p->is_del
 *   bit  23      skip     p->is_refskip || (b->core.flag&BAM_FUNMAP)  bam2bcf.c:173
 *   bits 24..31  tail     min(qpos, l_qseq-1-qpos, 255)  (CAP_DIST applied on device, bam2bcf.c:218-220)
 * `epos` (u8): (int)((double)get_position()/(len+1)*100), bam2bcf.c:80-114,234-235
 * `aux` (u32): p->aux as left by bcf_call_gap_prep, type<<16|seqQ<<8|indelQ (bam2bcf_indel.c:423)
 */
#define BCFGPU_RD_REV    (1u<<20)
#define BCFGPU_RD_SCLIP  (1u<<21)
#define BCFGPU_RD_DEL    (1u<<22)
#define BCFGPU_RD_SKIP   (1u<<23)

typedef struct bcfgpu_ctx bcfgpu_ctx;

/* configuration = the fields of bcf_callaux_t (bam2bcf.h:69-87) and call_t
 * (call.h:72-123) that the hot path reads */
typedef struct {
    int32_t device;        /* HIP device ordinal */
    int32_t n_smpl;        /* samples per site (bcf_call_t.n) */
    int32_t max_sites;     /* tile capacity */
    uint64_t max_reads;    /* tile capacity */
    /* mpileup side */
    int32_t min_baseQ;     /* mpileup -Q, default 13 (mpileup.c:937-950) */
    int32_t capQ;          /* 60 (bam2bcf.c:48) */
    double  errmod_theta;  /* <=0 -> CALL_DEFTHETA 0.83 (bam2bcf.c:38,46) */
    int32_t fmt_flag;      /* BCFGPU_FMT_* | BCFGPU_INFO_*; mpileup default VDB|RPB (mpileup.c:950) */
    /* call side */
    double  call_theta;    /* call -P, default 1.1e-3 (vcfcall.c:931-943); <=0: no prior */
    int32_t call_flag;     /* BCFGPU_CALL_KEEPALT | BCFGPU_CALL_VARONLY */
    int32_t output_tags;   /* BCFGPU_CALL_FMT_GQ | BCFGPU_CALL_FMT_GP | BCFGPU_CALL_FMT_PV4 */
    int32_t n_grp;         /* number of -G sample groups; <=1: one pooled group */
    int32_t grp_tag_is_qs; /* -G with FORMAT/QS (1) or FORMAT/AD (0) as frequency source */
    int32_t ploidy_max;    /* ploidy_max(args->ploidy) used for the prior's allele count (vcfcall.c:654-655, mcall.c:397-405); 0 -> 2 */
} bcfgpu_cfg;

/* one tile of pileup columns (device pointers) */
typedef struct {
    int32_t  n_sites;
    int32_t  is_indel;        /* 0: SNP pass (ref_base>=0); 1: indel pass (ref_base=-1, aux used) */
    uint64_t n_reads;
    const int8_t   *ref16;    /* [n_sites] 4-bit reference code handed to bcf_call_glfgen (mpileup.c:341-342) */
    const uint32_t *plp_off;  /* [n_sites*n_smpl+1] */
    const uint32_t *rd;       /* [n_reads] */
    const uint8_t  *epos;     /* [n_reads] */
    const uint32_t *aux;      /* [n_reads] or NULL */
} bcfgpu_tile;

/* per-site result of the mpileup stage = the scalar part of bcf_call_t
 * (bam2bcf.h:111-129) after bcf_call_combine */
typedef struct {
    int32_t a[5];             /* allele order, -1 = unused */
    int32_t n_alleles, unseen, ori_ref, shift;
    int32_t ret;              /* bcf_call_combine's return: 0, or -1 (indel site without alt) */
    uint32_t depth, ori_depth, mq0;
    float   qsum[5];          /* INFO/QS */
    float   vdb, mwu_pos, mwu_mq, mwu_bq, mwu_mqs, seg_bias;   /* +inf = HUGE_VAL = tag omitted */
    int32_t adf_tot[5], adr_tot[5];   /* ADF[0..4], ADR[0..4]: site totals, bam2bcf.c:676-690 */
    int32_t scr_tot;          /* SCR[0] */
    int32_t pad;
    double  anno[16];         /* INFO/I16 before the float cast (bam2bcf.c:831) */
} bcfgpu_site;

/* per-sample results of the mpileup stage, SoA planes in HBM.
 * plane strides are n_smpl; the per-site block of each array holds `planes`
 * planes whatever n_alleles is, so addressing does not depend on the data:
 *      pl [site][BCFGPU_MAX_PL][n_smpl]      u8   PL (<=255, bam2bcf.c:645-647); first n_alleles*(n_alleles+1)/2 planes valid
 *      dp4[site][4][n_smpl]                  u16  DP4 (anno[0..3], bam2bcf.c:650-659)
 *      adf/adr[site][5][n_smpl]              u16  ADF/ADR in *allele order* (bam2bcf.c:668-697); first n_alleles planes valid
 *      qs [site][5][n_smpl]                  i32  FMT/QS in allele order (bam2bcf.c:698-712)
 *      scr[site][n_smpl]                     u16  SCR[1+i]
 *      sp [site][n_smpl]                     u8   FMT/SP: Phred-scaled two-sided Fisher exact test of DP4 (bam2bcf.c:867-885)
 * The count planes are 16 bits wide so that a cell deeper than 255 reads keeps its counts over all of its reads, as
 * bcf_call_glfgen does (only errmod_cal's input is cut to 255, bam2bcf.c:256); a count past 65535 is BCFGPU_E_DEPTH.
 */
typedef struct {
    bcfgpu_site *site;        /* [n_sites] */
    uint8_t  *pl;
    uint16_t *dp4;
    uint16_t *adf, *adr;      /* may be NULL when no AD-type flag is set */
    int32_t  *qs;             /* may be NULL unless BCFGPU_FMT_QS or grouped calling on QS */
    uint16_t *scr;            /* may be NULL unless an SCR flag is set */
    uint8_t  *sp;             /* may be NULL unless BCFGPU_FMT_SP is set */
} bcfgpu_mplp_out;

/* input of the call stage when it is used on its own (e.g. on records parsed
 * from a VCF/BCF): everything mcall() reads from the record */
typedef struct {
    int32_t n_sites;
    int32_t n_gt_max;         /* plane count of `pl` per site (>= max over sites of nals*(nals+1)/2) */
    int32_t n_al_max;         /* plane count of `ad` per site */
    int32_t reserved;
    const int32_t *nals;      /* [n_sites] rec->n_allele */
    const int32_t *unseen;    /* [n_sites] index of <*>/X or 0 if none (vcfcall.c:1102-1111) */
    const int32_t *pl;        /* [site][n_gt_max][n_smpl]  FORMAT/PL incl. missing / vector_end sentinels */
    const float   *qs;        /* [site][5] INFO/QS, entries beyond those present = 0 (mcall.c:1456-1464) */
    const int32_t *ad;        /* [site][n_al_max][n_smpl] FORMAT/AD|QS for -G, or NULL */
    const uint8_t *ploidy;    /* [n_smpl] (call->ploidy) or NULL = all diploid; constant over the tile */
    const int32_t *grp;       /* [n_smpl] group id of each sample, or NULL */
    const int32_t *prior_an;  /* [n_sites] -F AN or NULL */
    const int32_t *prior_ac;  /* [site][4] -F AC, missing/vector_end sentinels allowed, or NULL */
    const float   *i16;       /* [site][16] INFO/I16, or NULL: then DP4/MQ/PV4 are not produced (has_i16 = 0) */
} bcfgpu_call_in;

/* per-site result of the call stage */
typedef struct {
    int32_t ret;              /* mcall()'s return value: nals_new, 0, or -2 */
    int32_t nals_new;         /* alleles kept */
    int32_t als_new;          /* bit mask of kept alleles (call->als_new) */
    int32_t als_map[5];       /* old -> new allele index or -1 (mcall.c:547-557) */
    int32_t ac[5];            /* call->ac: per kept allele */
    int32_t an;               /* INFO/AN */
    int32_t qual_missing;     /* 1 when QUAL is '.' (mcall.c:1644) */
    float   qual;             /* rec->qual */
    int32_t pl_dropped;       /* 1 when FORMAT/PL is removed (mcall.c:1583) */
    /* from INFO/I16 (mcall.c:1659-1679), when I16 is available (fused pipeline, or bcfgpu_call_in.i16): */
    int32_t has_i16;          /* 0: the four fields below are not set */
    int32_t dp4[4];           /* INFO/DP4 */
    int32_t mq;               /* INFO/MQ */
    int32_t pv4_tested;       /* with BCFGPU_CALL_FMT_PV4: 1 when INFO/PV4 is written (test16: both ref and alt reads present) */
    float   pv4[4];           /* INFO/PV4: strand (Fisher exact), baseQ, mapQ, tail-distance bias (t-tests), ccall.c:89-138 */
} bcfgpu_call_site;

/* per-sample results of the call stage:
 *      gt [site][2][n_smpl]               i8   allele index, BCFGPU_GT_MISSING or BCFGPU_GT_VECTOR_END
 *      pl [site][n_gt_max][n_smpl]        i32  trimmed PL (mcall.c:1158-1194) incl. sentinels; first ngts_new planes valid
 *      gq [site][n_smpl]                  i32  FORMAT/GQ   (NULL unless requested)
 *      gp [site][n_gt_max][n_smpl]        f32  FORMAT/GP   (NULL unless requested); missing = NaN 0x7F800001, vector_end = 0x7F800002
 */
typedef struct {
    bcfgpu_call_site *site;
    int8_t  *gt;
    int32_t *pl;
    int32_t *gq;
    float   *gp;
} bcfgpu_call_out;

/* ---- life cycle ------------------------------------------------------------ */
int  bcfgpu_create(const bcfgpu_cfg *cfg, bcfgpu_ctx **out);
void bcfgpu_destroy(bcfgpu_ctx *ctx);
const char *bcfgpu_last_error(void);
int  bcfgpu_device_count(void);
/* sizeof() of {cfg, tile, site, mplp_out, call_in, call_site, call_out, timing}: lets a binding check its struct mirrors */
void bcfgpu_abi_sizes(int32_t out[8]);

/* ---- device memory helpers (thin wrappers of hipMalloc/hipMemcpy) ---------- */
int  bcfgpu_malloc(bcfgpu_ctx *ctx, size_t bytes, void **dptr);
int  bcfgpu_free(bcfgpu_ctx *ctx, void *dptr);
int  bcfgpu_memcpy_h2d(bcfgpu_ctx *ctx, void *dst, const void *src, size_t bytes);
int  bcfgpu_memcpy_d2h(bcfgpu_ctx *ctx, void *dst, const void *src, size_t bytes);
int  bcfgpu_memset(bcfgpu_ctx *ctx, void *dst, int value, size_t bytes);
int  bcfgpu_sync(bcfgpu_ctx *ctx);
/* Page-locked host memory for the buffers a caller hands to the entries that take HOST pointers (bcfgpu_pileup's read pool,
 * bcfgpu_baq, ...): from such memory an upload is a DMA transfer that runs beside the kernels of other contexts, from
 * ordinary memory the runtime stages it through the CPU.  Optional: every entry accepts ordinary memory. */
int  bcfgpu_host_alloc(size_t bytes, void **ptr);
int  bcfgpu_host_free(void *ptr);
/* Cells of the launches since the last call that held more than BCFGPU_MAX_DEPTH usable reads.  errmod_cal (htslib errmod.c)
 * shuffles such a cell's reads with hts_drand48 and keeps 255 -- a draw from a process-wide generator that depends on the
 * order in which the whole run visits its cells.  Here the LIKELIHOODS of such a cell (p[25], hence its PLs) come from its
 * first 255 usable reads; everything else the reference computes over all reads -- DP4, AD/ADF/ADR, QS, SCR, I16, the bias-test
 * histograms, depth -- is over all reads here too.  The counter tells how many cells' PLs may deviate from a reference run
 * (cells that bcfgpu_errmod_plan drew for are not counted: see there).
 * With mpileup's default -d 250 per file (bcfgpu_depth_cap) such cells arise only where the cap lets reads through. */
int  bcfgpu_truncated_cells(bcfgpu_ctx *ctx, uint32_t *n_cells);
/* errmod_cal's own rule for such cells: the random draw, replayed.  hts_drand48 is a 48-bit linear congruential generator,
 * one per process, started from htslib's default seed; a cell of n > 255 reads takes n - 1 numbers from it (ks_shuffle) and
 * keeps the first 255 of the shuffled order; the cells draw in the order mpileup_reg() visits them -- position by position,
 * the samples of the SNP pass, then the samples of the indel pass where bcf_call_gap_prep returned >= 0 (mpileup.c:343-360).
 * bcfgpu_errmod_plan ranks the over-deep cells of a tile's two passes in that order, jumps the generator to each cell's place
 * and marks the 255 reads it keeps; the NEXT bcfgpu_mpileup / bcfgpu_pipeline of each of the two tiles uses the marks (its
 * likelihoods are then errmod_cal's, and bcfgpu_truncated_cells does not count those cells), and the context's generator
 * moves on by the numbers drawn, as the process-wide one does.
 *   snp     the SNP pass's tile (or NULL);  indel  the indel pass's tile as bcfgpu_gap_prep_tile left it (or NULL)
 *   indel_cols  HOST [indel->n_sites]: the SNP-tile column of every site of the indel tile (bcfgpu_gap_prep_tile: the entries of
 *               its `cols` with ret == 0, in order)
 *   indel_ret   HOST [indel->n_sites] or NULL: bcf_call_gap_prep's return values where the indel tile holds columns it turned
 *               away as well (bcfgpu_pileup_indel_tile over all candidates) -- only sites with 0 are visited
 * Synchronises the stream.  Without a plan the first 255 usable reads of such a cell are taken (see above).
 * bcfgpu_errmod_seed / _state: the generator's 48-bit state (a new context starts at 0x1234ABCD330E, a fresh process). */
int  bcfgpu_errmod_plan(bcfgpu_ctx *ctx, const bcfgpu_tile *snp, const bcfgpu_tile *indel, const int32_t *indel_cols, const int32_t *indel_ret);
/* The same with the columns mpileup_reg() passes over before either pass (positions outside the -t / -T targets, mpileup.c:330-335:
 * no bcf_call_glfgen there, hence no draw): snp_visit HOST [snp->n_sites] or NULL, 0 = the column is not visited.  (Indel sites on
 * such columns: give them a non-zero indel_ret, or leave them out of the indel tile.) */
int  bcfgpu_errmod_plan_visit(bcfgpu_ctx *ctx, const bcfgpu_tile *snp, const uint8_t *snp_visit, const bcfgpu_tile *indel,
                              const int32_t *indel_cols, const int32_t *indel_ret);
int  bcfgpu_errmod_seed(bcfgpu_ctx *ctx, uint64_t state);
uint64_t bcfgpu_errmod_state(bcfgpu_ctx *ctx);
/* enqueue on an externally owned hipStream_t (e.g. torch's current stream); NULL = the context's own */
int  bcfgpu_set_stream(bcfgpu_ctx *ctx, void *hip_stream);

/* ---- host-side read packer -------------------------------------------------
 * Fills one `rd`/`epos` pair from the fields of a bam_pileup1_t / bam1_t.
 * cigar = bam_get_cigar(b) (BAM encoding, len<<4|op), want_epos = fmt_flag has
 * RPB or VDB (bam2bcf.c:232). */
void bcfgpu_pack_read(int nt16, int baseQ, int mapQ, int is_rev, int has_softclip,
                      int is_del, int is_refskip_or_unmapped, int qpos, int l_qseq,
                      const uint32_t *cigar, int n_cigar, int want_epos,
                      uint32_t *rd, uint8_t *epos);

/* ---- the hot path ------------------------------------------------------------ */
/* glfgen (+errmod_cal) for every (site,sample) and combine (+SGB/MWU/VDB) for every site of the tile */
int  bcfgpu_mpileup(bcfgpu_ctx *ctx, const bcfgpu_tile *tile, const bcfgpu_mplp_out *out);

/* mcall for every site of `in`; ploidy/groups/prior as described above */
int  bcfgpu_mcall(bcfgpu_ctx *ctx, const bcfgpu_call_in *in, const bcfgpu_call_out *out);

/* mpileup stage followed by the call stage with PL/QS kept in HBM: `mpileup | call -m` for the tile's records, SNP tiles and
 * indel tiles alike (the reference pipes both kinds of record through mcall(), mpileup.c:357-364 -> vcfcall.c:1137; an indel
 * record has no <*> allele, vcfcall.c:1102-1111).  A site where the mpileup stage wrote no record (site.ret < 0: an indel
 * column without an ALT allele, bam2bcf.c:611) gets a call record with ret = 0 and is not an error.
 * `ploidy` and `grp` as in bcfgpu_call_in (device pointers or NULL).  `mout` receives the
 * mpileup-stage results (input of the call stage), `cout` the calls. */
int  bcfgpu_pipeline(bcfgpu_ctx *ctx, const bcfgpu_tile *tile, const uint8_t *ploidy, const int32_t *grp,
                     const bcfgpu_mplp_out *mout, const bcfgpu_call_out *cout);

/* ---- indel candidates: bcf_call_gap_prep (bam2bcf.h:141, bam2bcf_indel.c:99-470) for a batch of positions --------
 * Every stage runs on the device (csrc/gap_prep.hip, csrc/indel.hip): candidate typing, the insertion and per-sample
 * consensus, the realignment of every read against every candidate type (probaln_glocal, "the bottleneck",
 * bam2bcf_indel.c:335) and indelQ / seqQ with the choice of the <= 4 output types.  The host side of this call uploads
 * the arrays, launches, and brings the results back.  All pointers here are HOST pointers; bcfgpu_gap_prep_tile is the
 * form for callers whose reads are already in HBM (after bcfgpu_pileup).
 * Reads are a flat pool: r_* arrays indexed by read, cig/seq16/qual/zq pools indexed through r_cig_off / r_seq_off
 * (seq16: one 4-bit nt16 code per byte; qual: the qualities the pileup sees; zq: "ZQ" tag bytes, r_has_zq flags).
 * Pileup entries of (site k, sample s): smpl_off[k*n_smpl+s] .. smpl_off[k*n_smpl+s+1]-1 into p_read/p_qpos/p_indel. */
typedef struct {
    int32_t n_reads;
    const int32_t *r_pos, *r_lq, *r_flag, *r_ncig, *r_cig_off, *r_seq_off;
    const uint32_t *cig;
    const uint8_t *seq16, *qual, *zq, *r_has_zq;
} bcfgpu_reads;

typedef struct {
    int32_t n_sites, n_smpl;
    const int32_t *pos;          /* [n_sites] 0-based position of the base before the indel */
    const int32_t *smpl_off;     /* [n_sites*n_smpl+1] */
    const int32_t *p_read, *p_qpos, *p_indel;
    const char *ref;             /* NUL-terminated reference of the contig */
    int32_t openQ, extQ, tandemQ, min_support, per_sample_flt;   /* bcf_callaux_t, mpileup -o -e -h -m -p */
    double min_frac;             /* mpileup -F */
} bcfgpu_indel_in;

typedef struct {
    int32_t *ret;                /* [n_sites] return value of bcf_call_gap_prep: 0 or -1 */
    uint32_t *p_aux;             /* [entries] p->aux = type<<16 | seqQ<<8 | indelQ (valid where ret==0) */
    int32_t *indel_types;        /* [n_sites][4] bca->indel_types */
    int8_t  *inscns;             /* [n_sites][4*inscns_cap] bca->inscns (stride maxins[k] inside a site) */
    int32_t *maxins, *indelreg, *max_support;
    float   *max_frac;
} bcfgpu_indel_out;

int  bcfgpu_gap_prep(bcfgpu_ctx *ctx, const bcfgpu_reads *reads, const bcfgpu_indel_in *in, const bcfgpu_indel_out *out,
                     int inscns_cap);

/* ---- BAQ: sam_prob_realn(b, ref, ref_len, flag) of htslib realn.c as mpileup applies it to every read before the
 * pileup (mpileup.c:234, flag = 1 apply | 2 extended; `mpileup -E` redoes it with 7 = recompute).  For every read of
 * the pool: the banded glocal pair-HMM forward-backward against its reference window, the per-base posterior of the
 * aligned column, and the base-quality cap that follows.  HOST pointers.  `reads->qual` is not modified:
 *   qual_out[r_seq_off[r] + i]  new quality of base i (= qual when the read is left alone, ret[r] < 0)
 *   zq_out  [r_seq_off[r] + i]  the "ZQ" tag byte (64 + cap offset), 0 when the read is left alone
 *   ret[r]                      0 applied, -1 left alone (no sequence, qual 0xff, N in CIGAR, no aligned base)
 * Reads that already carry BQ/ZQ tags are the caller's business (realn.c:60-90); this entry is the computation. */
int  bcfgpu_baq(bcfgpu_ctx *ctx, const bcfgpu_reads *reads, const char *ref, int32_t ref_len, int flag,
                uint8_t *qual_out, uint8_t *zq_out, int32_t *ret);

/* ---- mpileup -C INT: sam_cap_mapq(b, ref, ref_len, thres) of htslib sam.c as mplp_func applies it to every read after
 * BAQ (mpileup.c:235-239, only when thres > 10): a cap on the read's mapping quality from its mismatches against the
 * reference (their count and capped base qualities, the aligned length, the clipped bases).  HOST pointers; reads->qual =
 * the qualities after bcfgpu_baq.  cap[r]: -1 = the read is dropped (mpileup.c:237), else the caller lowers the read's
 * mapping quality to cap[r] when it is above it (:238).  The filters that follow in mplp_func (-q, orphans) are the caller's. */
int  bcfgpu_cap_mapq(bcfgpu_ctx *ctx, const bcfgpu_reads *reads, const char *ref, int32_t ref_len, int32_t thres, int32_t *cap);

/* ---- mate overlaps: what bam_mplp_init_overlaps() (mpileup.c:640) makes htslib's pileup do to read pairs whose mates
 * overlap (sam.c tweak_overlap_quality): at every reference position both reads cover with an aligned base, equal
 * bases pool their qualities in the first read (at most 200) and different bases keep 0.8 of the better quality (the
 * first read's on ties); the other read's base gets quality 0.  Pair p is (pair_a[p] = the mate that entered the
 * pileup first, pair_b[p]); which reads pair up (proper pair, same contig, first mate still buffered) is the pileup
 * engine's bookkeeping and stays with the caller.  A read may be in one pair only.  HOST pointers; `reads->qual` (the
 * qualities after BAQ) is not modified: qual_out is the whole quality pool with the pairs' bases rewritten. */
int  bcfgpu_overlap_tweak(bcfgpu_ctx *ctx, const bcfgpu_reads *reads, int32_t n_pairs, const int32_t *pair_a,
                          const int32_t *pair_b, uint8_t *qual_out);

/* ---- the pileup itself: the columns bam_mplp_auto() hands to mpileup_reg() (mpileup.c:320-347), built on the device
 * from the reads of a region instead of being packed read by read on the host (bcfgpu_pack_read) and sent over PCIe
 * at 5 bytes per (read, position): the read pool is ~30x smaller than the tile it expands to.
 *   reads    the flat pool (HOST pointers), already filtered as mplp_func does (mpileup.c:183-246), with the qualities
 *            the pileup should see (after bcfgpu_baq / bcfgpu_overlap_tweak); r_pos on the region's contig
 *   r_mapq   [n_reads] mapping qualities;  r_smpl [n_reads] sample index of each read.  The reads of one sample must
 *            come in ascending r_pos order (a position-sorted BAM; a sample spread over several files: merged).
 *   [beg,end) the region, one column per position (columns without reads are there too: every cell empty);
 *   ref/ref_len  the contig's sequence, for the column's reference base (N past its end)
 *   tile     out: DEVICE pointers into the context's workspace, valid until the next bcfgpu_pileup on this context;
 *            ready for bcfgpu_mpileup / bcfgpu_pipeline (a context created with max_sites >= end-beg and max_reads >=
 *            tile->n_reads).  Entries of a cell are in the order of the reads.
 *   col_n    out, HOST [end-beg] or NULL: pileup entries per column (0: the reference emits no record there)
 *   col_indel out, HOST [end-beg] or NULL: 1 when some read of the column is followed by an indel (the columns
 *            bcf_call_gap_prep looks at, bam2bcf_indel.c:106-113)
 * The per-file depth cap of the iterator (mpileup -d) is the caller's to apply to the pool first: bcfgpu_depth_cap. */
int  bcfgpu_pileup(bcfgpu_ctx *ctx, const bcfgpu_reads *reads, const uint8_t *r_mapq, const int32_t *r_smpl,
                   int32_t beg, int32_t end, const char *ref, int32_t ref_len,
                   bcfgpu_tile *tile, int32_t *col_n, uint8_t *col_indel);

/* The same call for a pool in the form BAM records hold it, for hosts that feed the device over PCIe (the pool's bytes are what
 * bounds that path): bases as two 4-bit codes per byte, qualities optionally as 4-bit indices into a palette of <= 16 values
 * (binned qualities; after BAQ the values are too many and qual4 is left NULL).  `reads` as for bcfgpu_pileup, except that
 * reads->seq16 is not read, nor reads->qual when qual4 is given; the pool is expanded on the device.
 *   seq4     base i of read r is nibble r_seq_off[r] + i: byte (r_seq_off[r]+i)/2, the HIGH nibble when that index is even
 *            (bam_get_seq's order; a host that copies the records' sequence bytes keeps every r_seq_off even)
 *   qual4    NULL, or palette indices addressed the same way;  palette[j] = the quality index j stands for
 *   n_bases  bases the pools hold (the largest r_seq_off[r] + r_lq[r]);  n_cig  operations in reads->cig
 *   smpl_off NULL, or [n_smpl+1]: the reads of sample s are smpl_off[s] .. smpl_off[s+1]-1 of the pool (then r_smpl is not
 *            read and the call makes no pass over the reads on the host at all)
 *   recs     NULL, or the per-read fields as 12-byte records (bcfgpu_read12), see there
 * bcfgpu_gap_prep_tile, bcfgpu_pileup_entries and bcfgpu_pileup_indel_tile follow it as they follow bcfgpu_pileup. */
typedef struct {
    int32_t  pos;                /* r_pos */
    uint16_t lq;                 /* r_lq */
    uint8_t  ncig;               /* r_ncig */
    uint8_t  flag8;              /* bit 0: reverse strand (BAM flag 16), bit 1: unmapped (4) -- what the stages read of r_flag */
    uint8_t  mapq;               /* r_mapq */
    uint8_t  pad[3];
} bcfgpu_read12;

typedef struct {
    const uint8_t *seq4, *qual4;
    uint8_t palette[16];
    int64_t n_bases, n_cig;
    const int32_t *smpl_off;
    int32_t qual_bits;           /* bits per palette index in qual4: 0 or 4 = two per byte (above); 2 = four per byte, index i of a
                                    byte in bits 7-2i..6-2i (the first base highest; a palette of <= 4 values: the four bins of
                                    current sequencers); reads then start at multiples of four bases when copied bytewise */
    const bcfgpu_read12 *recs;   /* NULL, or [n_reads]: the per-read arrays as one 12-byte record a read (half the bytes of the six
                                    arrays and r_mapq, which are then not read).  The pools are then dense: read r's bases start at
                                    the sum of the earlier reads' lengths, each rounded up to a multiple of four, its CIGAR at the
                                    sum of the earlier reads' operation counts (r_seq_off / r_cig_off are formed on the device) */
} bcfgpu_packed;

int  bcfgpu_pileup_packed(bcfgpu_ctx *ctx, const bcfgpu_reads *reads, const bcfgpu_packed *pk, const uint8_t *r_mapq,
                          const int32_t *r_smpl, int32_t beg, int32_t end, const char *ref, int32_t ref_len,
                          bcfgpu_tile *tile, int32_t *col_n, uint8_t *col_indel);

/* ---- the read pool of a region kept in HBM: what mplp_func does to a read (mpileup.c:183-246: BAQ :234, the -C cap :235-239)
 * and what the iterator does to pairs (:640) without the pool crossing PCIe between the stages.  bcfgpu_baq, bcfgpu_cap_mapq,
 * bcfgpu_overlap_tweak and bcfgpu_pileup each take the pool from the host and (but for the last) hand their result back; here the
 * pool is uploaded once -- packed (bcfgpu_packed) or a byte per base -- and the stages work on that copy, in mplp_func's order:
 *   bcfgpu_pool_upload        reads / pk / r_mapq as for bcfgpu_pileup[_packed]; replaces the context's pool; the caller's arrays
 *                             are its own again when the call returns (so is `ref` of the calls below)
 *   bcfgpu_pool_baq           sam_prob_realn on every read: the pool's qualities become the new ones, the ZQ bytes stay in HBM
 *                             (bcfgpu_gap_prep_tile finds them there when its `reads` is NULL); ret: HOST [n_reads] as for
 *                             bcfgpu_baq, or NULL
 *   bcfgpu_pool_cap_mapq      sam_cap_mapq: cap: HOST [n_reads] out as for bcfgpu_cap_mapq; the pool's mapping qualities above
 *                             their cap are lowered to it (mpileup.c:238); dropping the reads with cap < 0 (:237) and the -q /
 *                             orphan filters that follow are the caller's: bcfgpu_pool_keep
 *   bcfgpu_pool_keep          keep: HOST [n_reads], 0 = the read does not enter the pileup (filters, bcfgpu_depth_cap); NULL = all
 *   bcfgpu_pool_overlap_tweak the pairs as for bcfgpu_overlap_tweak; the pool's qualities are rewritten in place
 *   bcfgpu_pool_pileup        the tile, as bcfgpu_pileup builds it (r_smpl, or smpl_off as in bcfgpu_packed)
 *   bcfgpu_pool_download      the pool's current qualities / ZQ bytes / mapping qualities back to the host (any may be NULL): tests,
 *                             and callers that want BQ/ZQ tags written
 * The pool lives in the context's workspace until the next bcfgpu_pool_upload / bcfgpu_pileup[_packed] on that context. */
int  bcfgpu_pool_upload(bcfgpu_ctx *ctx, const bcfgpu_reads *reads, const bcfgpu_packed *pk, const uint8_t *r_mapq);
int  bcfgpu_pool_baq(bcfgpu_ctx *ctx, const char *ref, int32_t ref_len, int flag, int32_t *ret);
int  bcfgpu_pool_cap_mapq(bcfgpu_ctx *ctx, const char *ref, int32_t ref_len, int32_t thres, int32_t *cap);
int  bcfgpu_pool_keep(bcfgpu_ctx *ctx, const uint8_t *keep);
int  bcfgpu_pool_overlap_tweak(bcfgpu_ctx *ctx, int32_t n_pairs, const int32_t *pair_a, const int32_t *pair_b);
int  bcfgpu_pool_pileup(bcfgpu_ctx *ctx, const int32_t *r_smpl, const int32_t *smpl_off, int32_t beg, int32_t end,
                        const char *ref, int32_t ref_len, bcfgpu_tile *tile, int32_t *col_n, uint8_t *col_indel);
int  bcfgpu_pool_download(bcfgpu_ctx *ctx, uint8_t *qual, uint8_t *zq, uint8_t *r_mapq);

/* ---- the pileup iterator's per-file depth cap: mpileup -d (mpileup.c:646 bam_mplp_set_maxcnt; htslib sam.c bam_plp_push) -------
 * Host helper (integer bookkeeping of the read buffer, no device work).  A read is dropped when it starts at the position of
 * the read kept last while max_depth or more reads are still buffered: reads kept earlier whose end (bam_endpos) is not before
 * that position.  The first read of every start position is always kept, so a column can exceed max_depth by a little.
 * One sample = one input file (mpileup counts per file).  reads: the pool after the read filters of mplp_func, every sample's
 * reads in position order; keep: out [n_reads], 1 = the read enters the pileup.  max_depth <= 0: keep everything. */
int  bcfgpu_depth_cap(const bcfgpu_reads *reads, const int32_t *r_smpl, int32_t n_smpl, int32_t max_depth, uint8_t *keep);
/* The same for a caller that streams a sorted file through the region in batches (host/bcfgpu_sam cuts a region into tiles): the
 * iterator's buffer (the ends of the reads kept so far, the position of the read kept last, per file) lives in the state between
 * the calls, so the batches' keep[] together are what one call over all the reads would give.  _reset: a new region (the
 * reference restarts its iterators per region, mpileup.c:652-683). */
typedef struct bcfgpu_depth_state bcfgpu_depth_state;
bcfgpu_depth_state *bcfgpu_depth_cap_new(int32_t n_smpl, int32_t max_depth);
int  bcfgpu_depth_cap_push(bcfgpu_depth_state *st, const bcfgpu_reads *reads, const int32_t *r_smpl, uint8_t *keep);
void bcfgpu_depth_cap_reset(bcfgpu_depth_state *st);
void bcfgpu_depth_cap_free(bcfgpu_depth_state *st);

/* The pileup entries of selected columns of the last bcfgpu_pileup on this context, in the form bcfgpu_gap_prep takes them
 * (bcfgpu_indel_in: what bcf_call_gap_prep reads of bam_pileup1_t): for column cols[i] and sample s the entries
 * smpl_off[i*n_smpl+s] .. smpl_off[i*n_smpl+s+1]-1 of p_read (index into the read pool), p_qpos, p_indel.
 * HOST pointers; smpl_off has n_cols*n_smpl+1 elements, the p_* arrays `cap` elements (>= the sum of col_n over cols). */
int  bcfgpu_pileup_entries(bcfgpu_ctx *ctx, int32_t n_cols, const int32_t *cols, int32_t *smpl_off,
                           int32_t *p_read, int32_t *p_qpos, int32_t *p_indel, int64_t cap);

/* The indel pass's tile (mpileup.c:357-360: the same pileup entries with p->aux set by bcf_call_gap_prep, ref_base = -1) for
 * selected columns of the last bcfgpu_pileup on this context -- the columns where bcfgpu_gap_prep returned 0.
 * aux: HOST, one word per entry of those columns in bcfgpu_pileup_entries' order (bcfgpu_indel_out.p_aux of the same
 * columns); n_aux must equal their entry count.  tile: out, DEVICE pointers into the context's workspace (is_indel = 1),
 * valid until the next call of this function or of bcfgpu_pileup on this context. */
int  bcfgpu_pileup_indel_tile(bcfgpu_ctx *ctx, int32_t n_cols, const int32_t *cols, const uint32_t *aux, int64_t n_aux,
                              bcfgpu_tile *tile);

/* bcf_call_gap_prep for candidate columns of the last bcfgpu_pileup on this context with everything staying in HBM: the
 * entries of the columns are listed on the device (what bcfgpu_pileup_entries would download), the stage runs on the read
 * pool bcfgpu_pileup left there, and p->aux goes straight into the indel pass's tile (what bcfgpu_pileup_indel_tile would
 * build from an uploaded array).  The host-pointer chain bcfgpu_pileup_entries -> bcfgpu_gap_prep -> bcfgpu_pileup_indel_tile
 * gives the same results.
 *   cols   HOST [n_cols], ascending column indices of the last bcfgpu_pileup (the candidates: col_indel != 0)
 *   reads  HOST, the pool handed to bcfgpu_pileup, or NULL: only zq / r_has_zq are read (the "ZQ" bytes bcfgpu_baq left;
 *          uploaded here because the pileup itself does not need them)
 *   par    the options of bcfgpu_indel_in (openQ ... min_frac) and `ref`; its array pointers are ignored
 *   out    HOST arrays per column of `cols` as for bcfgpu_gap_prep (a column with ret[i] = -1 has INDEL_NULL types and zeros
 *          elsewhere: what bca holds after such a call is read by no one, mpileup.c:354); p_aux may be NULL (the words stay
 *          on the device), else it receives the words of the TILE's entries (below), tile->n_reads of them
 *   tile   out: DEVICE pointers, the indel pass's tile (is_indel = 1, aux set), ready for bcfgpu_mpileup: the columns with
 *          ret[i] == 0 and only those, in the order of `cols` -- site j of the tile is the j-th such column (mpileup.c:354-360
 *          runs the indel pass where bcf_call_gap_prep returned >= 0).  n_sites = 0 when there is none.
 *          Valid until the next bcfgpu_gap_prep_tile / bcfgpu_gap_prep / bcfgpu_pileup* call on this context.
 * Without per_sample_flt the pooled support filter of bam2bcf_indel.c:150-154 is decided on the device from two counts the
 * pileup left per column, and nothing else -- no entry, no workgroup of the stage -- touches a column that fails it. */
int  bcfgpu_gap_prep_tile(bcfgpu_ctx *ctx, int32_t n_cols, const int32_t *cols, const bcfgpu_reads *reads,
                          const bcfgpu_indel_in *par, const bcfgpu_indel_out *out, int inscns_cap, bcfgpu_tile *tile);

/* ---- gVCF blocks (mpileup --gvcf, gvcf.c:88-226) ---------------------------------
 * gvcf_write() collapses runs of reference-only records into blocks: a record can join when it has only REF and <*>
 * (mpileup.c:309-315) and the smallest per-sample FORMAT/DP falls into a range > 0 of `dp_range` (gvcf.c:112-128); a block
 * ends where that range changes, at a record that cannot join, at a gap in positions or a new sequence (gvcf.c:130-131).
 * Per block and sample: the smallest DP and the smallest (PL[1], PL[2]) pair in that order (gvcf.c:192-210), PL[0], alleles
 * and INFO/QS are the first record's.  On the device this is a site-local rule for block starts, a prefix sum for block
 * numbers and a per-sample reduction over the block's sites, on the planes the mpileup stage left in HBM.
 * A block never spans two calls: a caller that cuts a region into tiles joins the last block of one call with the first
 * of the next where the rule above allows it (the per-sample rule is associative).
 */
typedef struct {
    int32_t first_site, last_site;    /* sites of the block (indices into the tile) */
    int32_t start_pos;                /* rec->pos of the first record, 0-based (gvcf.c:183) */
    int32_t end1;                     /* INFO/END, 1-based (gvcf.c:139-143); written only when start_pos+1 < end1 (gvcf.c:150) */
    int32_t min_dp;                   /* INFO/MinDP (gvcf.c:188,192) */
    int32_t range;                    /* 1-based index into dp_range the block belongs to */
} bcfgpu_gvcf_block;

typedef struct {
    int32_t n_sites, n_range;
    const int32_t *dp_range;          /* HOST [n_range], the --gvcf list as given (gvcf.c:44-67) */
    const int32_t *pos;               /* [n_sites] rec->pos (0-based), ascending within a sequence */
    const int32_t *rid;               /* [n_sites] sequence of the record, or NULL: all the same */
    const uint8_t *brk;               /* [n_sites] bit 0: a record that cannot join (the indel record of mpileup.c:354-365)
                                         follows this site at the same position -- the block ends here and its END stops
                                         one short (gvcf.c:139); bit 1: the site has no record at all (a column without
                                         reads): it joins nothing and blk is -1; or NULL */
    const bcfgpu_site *site;          /* the mpileup stage's output for the tile (n_alleles, unseen are read) */
    const uint8_t *pl;                /* bcfgpu_mplp_out.pl  */
    const uint16_t *dp4;              /* bcfgpu_mplp_out.dp4: FORMAT/DP is the sum of the four (bam2bcf.c:853-858) */
    /* The form `bcftools call -g` needs (vcfcall.c:1145-1149: gvcf_write(.., ret==1)): the records are whatever the caller
     * read, so what gvcf_write looks at comes as arrays of its own, and site / pl / dp4 are not read (may be NULL):
     *   ref_only [n_sites] u8: 1 = the record may join a block (mcall() kept the reference allele only)
     *   dp       [n_sites][n_smpl] i32: FORMAT/DP as it stands in the record (missing = INT32_MIN: the record stays alone)
     *   end      [n_sites] i32 or NULL: 0-based last position the record covers (INFO/END - 1 of a record that is a block
     *            already, gvcf.c:215-218); NULL: pos
     * A record that follows a block at the block's last position cuts it one short (gvcf.c:139) whichever form is used.
     * out->pl may be NULL in this form (call drops FORMAT/PL of reference-only records: a block carries GT and DP). */
    const uint8_t *ref_only;
    const int32_t *dp;
    const int32_t *end;
} bcfgpu_gvcf_in;

/*      blk    [n_sites]            i32  block of the site, or -1: its record is written as it is
 *      min_dp [n_sites]            i32  smallest per-sample DP of the site (INFO/MinDP of an uncollapsed reference record, gvcf.c:221-222)
 *      block  [n_sites]                 the blocks in order (capacity n_sites)
 *      dp     [block][n_smpl]      i32  FORMAT/DP of the block
 *      pl     [block][3][n_smpl]   u8   FORMAT/PL of the block */
typedef struct {
    int32_t *blk, *min_dp;
    bcfgpu_gvcf_block *block;
    int32_t *dp;
    uint8_t *pl;
} bcfgpu_gvcf_out;
/* n_blocks: HOST, out.  Synchronises the context's stream (the block count decides the launch of the reduction). */
int  bcfgpu_gvcf_blocks(bcfgpu_ctx *ctx, const bcfgpu_gvcf_in *in, const bcfgpu_gvcf_out *out, int32_t *n_blocks);

/* statistics of the last bcfgpu_gap_prep call on this context (SURVEY 8d "indel stage unit": DP cells per second):
 * jobs = (site, candidate type, read) realignments, passes = forward passes run (a second parameter set is tried when
 * the first score exceeds 5, bam2bcf_indel.c:351), dp_cells = sum over passes of l_query * (2*bw+1) * 3 */
typedef struct {
    uint64_t n_jobs, n_passes, dp_cells;
    float kernel_ms;             /* probaln_kernel launches, HIP events on the context's stream */
    float prepare_ms, finalize_ms, total_ms;   /* host typing/consensus, host scoring, whole call (wall clock) */
    uint64_t n_wide;             /* jobs whose band (|type| + 3 clipped as probaln_glocal clips it) is wider than the widest
                                  * register-resident class (types of about 8 bp and more): the row of the pair-HMM lives in LDS */
    uint64_t n_scratch;          /* ... of these, the jobs too wide (or too long) for LDS as well: rolling rows in a global scratch
                                  * buffer (bands past 300, reads past 65 535 bases) */
    uint32_t band_jobs[8];       /* jobs by band: <= 10 (a register class per width), 11-15, 16-31, 32-43 (the row in registers, D re-run),
                                  * 44-58, 59-73 (the row in LDS), 74-300 (in LDS, sixteen jobs a wavefront), beyond (scratch) */
} bcfgpu_gap_stats;
int  bcfgpu_gap_prep_stats(const bcfgpu_ctx *ctx, bcfgpu_gap_stats *out);

/* ---- several GPUs: region shards and the ordered gather (the reference's -r regions + `bcftools concat`,
 * mpileup.c:652-683, vcfconcat.c:420) --------------------------------------------------------------------------------
 * Sites are independent, so a region is cut into contiguous shards, one context (one GPU, one host thread or process)
 * each, with no exchange on the data path.  What goes to the writer on rank 0 is the records that will be written: they are
 * compacted on the device -- per record a fixed header with the call record and the mpileup-stage site record, then the
 * site's GT planes [2][n_smpl] i8 (padded to 4 bytes) and its trimmed PL planes [n_gt][n_smpl] i32, the whole padded to
 * 16 bytes, records back to back in site order -- and gathered in rank order, which is genomic order. */
typedef struct {
    int32_t site;                 /* site0 + index of the site in the shard's tile */
    int32_t n_gt;                 /* PL planes that follow: nals_new*(nals_new+1)/2, or 0 when FORMAT/PL is dropped */
    uint32_t bytes;               /* size of the whole record: the next one starts `bytes` further */
    int32_t pad;
    bcfgpu_call_site call;
    bcfgpu_site mplp;             /* zero when no mpileup-stage record was given */
} bcfgpu_call_rec;

/* The records of a tile's call stage that `bcftools call` would write (ret >= 0; with variants_only = 1 also ret != 0,
 * vcfcall.c:1140-1144; variants_only = 2: only records with an ALT allele called, i.e. what -v keeps, whatever the context's
 * call_flag), packed into d_buf (device, cap_bytes).  msite: the mpileup stage's site records (device) or NULL;
 * n_gt_planes: plane count of cout->pl per site.  n_bytes / n_rec (host, out): what was written.  Synchronises the stream. */
int  bcfgpu_compact_calls(bcfgpu_ctx *ctx, int32_t n_sites, int32_t site0, const bcfgpu_site *msite, const bcfgpu_call_out *cout,
                          int32_t n_gt_planes, int32_t variants_only, void *d_buf, uint64_t cap_bytes, uint64_t *n_bytes, uint32_t *n_rec);

/* The same without a word coming back to the host: everything is queued on the context's stream.  d_counts: DEVICE, four
 * 64-bit words the caller owns (so that several compactions can be in flight): [0] bytes of all records, [1] records,
 * [2] != 0 when they did not fit cap_bytes (nothing was written then).  bcfgpu_compact_counts waits for the stream and
 * reads them (BCFGPU_E_RANGE when [2] is set); a caller with streams of its own may read them any other way. */
int  bcfgpu_compact_calls_async(bcfgpu_ctx *ctx, int32_t n_sites, int32_t site0, const bcfgpu_site *msite, const bcfgpu_call_out *cout,
                                int32_t n_gt_planes, int32_t variants_only, void *d_buf, uint64_t cap_bytes, uint64_t *d_counts);
int  bcfgpu_compact_counts(bcfgpu_ctx *ctx, const uint64_t *d_counts, uint64_t *n_bytes, uint32_t *n_rec);

/* One communicator over the contexts of a node, rank i = ctxs[i] (RCCL ncclCommInitAll; every context on its own device;
 * librccl is loaded at this call, a single context needs none). */
typedef struct bcfgpu_comm bcfgpu_comm;
int  bcfgpu_comm_init_all(bcfgpu_ctx *const *ctxs, int32_t n, bcfgpu_comm **out);
void bcfgpu_comm_destroy(bcfgpu_comm *comm);
/* The gather: every rank calls it (from its own thread) with the same counts[n] -- the byte counts of all ranks, which
 * the caller shares through host memory.  Rank 0 ends up with rank 0's, rank 1's, ... bytes back to back in d_recv (device,
 * sum of counts); the others send d_send.  ncclGroupStart / ncclRecv x (n-1) on rank 0, one ncclSend per peer / ncclGroupEnd,
 * enqueued on each context's stream: bcfgpu_sync(ctx) waits for it. */
int  bcfgpu_gather_bytes(bcfgpu_comm *comm, int32_t rank, const void *d_send, const uint64_t *counts, void *d_recv);

/* byte sizes of the output planes for a tile of n_sites (n_smpl from the context) */
size_t bcfgpu_mplp_out_bytes(const bcfgpu_ctx *ctx, int n_sites, int which /*0 site,1 pl,2 dp4,3 adf,4 adr,5 qs,6 scr,7 sp*/);

/* timing of the last launches (ms, measured with HIP events on the stream the kernels ran on) */
typedef struct { float glfgen_ms, combine_ms, mcall_ms, total_ms; } bcfgpu_timing;
/* on=1: time each call and wait for it; on=2: only record events (no host wait), bcfgpu_timing_get() then returns
 * the per-launch averages since the previous get; on=0: off */
int  bcfgpu_timing_enable(bcfgpu_ctx *ctx, int on);
int  bcfgpu_timing_get(bcfgpu_ctx *ctx, bcfgpu_timing *t);

#ifdef __cplusplus
}
#endif
#endif

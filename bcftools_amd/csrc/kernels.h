// kernels.h -- parameter blocks and launchers of the HIP kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include <stdint.h>
#include "../../include/bcfgpu.h"

// Diagnostics (tools/ablate_kernel.sh builds with -DBCFGPU_DIAG): parts of a kernel can be switched off to time the rest.
// The product build has no such switch: BCFGPU_ABL() is a compile-time 0 and the parameter blocks carry no mask.
#ifdef BCFGPU_DIAG
#define BCFGPU_ABL(P, bit) (((P).ablate & (bit)) != 0)
#define BCFGPU_ABL_MASK(P) ((P).ablate)
#define BCFGPU_ABL_FIELD int ablate;
#else
#define BCFGPU_ABL(P, bit) false
#define BCFGPU_ABL_MASK(P) 0
#define BCFGPU_ABL_FIELD
#endif

namespace bcfgpu {

// histogram layout of one site (bcf_callaux_t's bias-test arrays, bam2bcf.h:77)
enum : int {
    H_REF_POS = 0, H_REF_MQ = 100, H_REF_BQ = 160, H_ALT_OFF = 220,      // every ALT array sits H_ALT_OFF after its REF array
    H_ALT_POS = H_REF_POS + H_ALT_OFF, H_ALT_MQ = H_REF_MQ + H_ALT_OFF, H_ALT_BQ = H_REF_BQ + H_ALT_OFF,
    H_FWD_MQS = 440, H_REV_MQS = 500, H_SIZE = 560
};
// site totals glfgen_kernel accumulates read-parallel: anno[4..15] (bam2bcf.c:221-226), then ori_depth and mq0
enum : int { SITE_NSUM = 14 };

// per (site,sample) result of the glfgen kernel = bcf_callret1_t (bam2bcf.h:90-108), SoA planes over
// ncells = n_sites*n_smpl.  Everything but p is an exact integer.
// Most cells show a single base (the reference's): their 15 genotype likelihoods take three values -- 0 for the
// homozygote of that base b, A = (float)bsum[b] for every genotype without b, and H = (float)(4.343 * ln2 * n) for the
// heterozygotes with b (errmod_cal's lhet term alone, a function of the read count n) -- so such a cell stores A and b and
// combine_kernel rebuilds the planes it needs; cells with two or more bases store all 15 values.
struct CallretPlanes {
    float    *p15;    // [ncells][16]  upper triangle of p[5][5]: index k*(k+1)/2+j for j<=k (a 64-byte record per cell); written for CR_FULL cells only
    float    *pa;     // [ncells]      A of a single-base cell
    uint64_t *qs64;   // [ncells]      QS[0..3] packed 4 x u16
    uint32_t *adf;    // [ncells]      ADF[0..3] packed 4 x u8
    uint32_t *adr;    // [ncells]      ADR[0..3] packed 4 x u8
    uint32_t *cnt4;   // [ncells]      anno[0..3] packed 4 x u8
    uint32_t *misc;   // [ncells]      code | SCR<<8; code = the cell's base 0..4, or CR_FULL: all 15 likelihoods are in p15;
                      //               | CR_WIDE: the cell has more than 255 usable reads -- its packed counts above are those of the 255
                      //               reads errmod_cal took, qs64 is WIDE_QS_MARK | the index of its WideRec, which has the counts over all reads
    struct WideRec *wide;   // [wide_cap]
};

enum : uint32_t { CR_FULL = 0x80, CR_WIDE = 0x40 };
// qs64 of a CR_WIDE cell: this mark (QS[3] = 0xffff cannot come from 255 reads of quality <= 63) | the index of its WideRec
#define WIDE_QS_MARK (0xffffull << 48)

// What bcf_call_glfgen leaves for a cell of more than 255 usable reads, over ALL of them (bam2bcf.c:203-226): such cells are
// rare (never under mpileup's default -d 250 with one file per sample), so the planes above keep their packed form and these
// records are handed out by an atomic counter; a tile of n reads has at most n/256 of them.
struct WideRec {
    uint32_t qs[4];        // QS[0..3]
    uint32_t ad[4];        // ADF[b] | ADR[b] << 16
    uint32_t cnt[2];       // anno[0] | anno[1] << 16, anno[2] | anno[3] << 16
    uint32_t scr, n;       // SCR; usable reads (bcf_call_glfgen's return value)
    uint32_t cell, pad[3];
};
static_assert(sizeof(WideRec) == 64, "one cache-line-aligned record per over-deep cell");

struct GlfgenParams {
    int n_sites, n_smpl, is_indel;
    int min_baseQ, capQ, fmt_flag;
    int hist_slots;                 // >0: per-workgroup LDS histograms with that many site slots; 0: global atomics
    int lds_cap;                    // read keys held in LDS per workgroup round (multiple of 16, <= 16384)
    int part_cols;                  // LDS columns per partial sum of phase A (power of two >= 4; 12 * hist_slots * part_cols <= 2048: the slot region)
    uint32_t n_reads;               // length of rd/epos (bounds of the vector loads)
    BCFGPU_ABL_FIELD
    const int8_t   *ref16;
    const uint32_t *off;
    const uint32_t *rd;
    const uint8_t  *epos;
    const uint32_t *aux;
    const double *fk, *beta, *lhet;
    const CallretPlanes *crp;       // the output planes' addresses, in device memory: read where the stores are (glfgen.hip)
    int *hist;                      // [n_sites][H_SIZE], zeroed before launch
    unsigned long long *site_sums;  // [n_sites][SITE_NSUM] site totals of anno[4..15], ori_depth, mq0 (exact integers), zeroed before launch
    int *err;                       // device error word
    unsigned int *trunc;            // cells whose likelihoods come from their first 255 usable reads (counter)
    uint32_t *wide_ctr;             // [0] WideRecs handed out by this launch (zeroed before launch)
    const uint32_t *draw_bits;      // NULL, or a bit per read of the tile: the 255 reads bcfgpu_errmod_plan drew for every over-deep cell
    uint32_t wide_cap;              // records in crp->wide
    // cells whose pileup does not fit the LDS key window: listed by the tile launch, worked on by the launch that follows
    uint32_t *deep_list;            // [deep_cap][2]: cell, offset of its keys in deep_keys
    uint32_t *deep_ctr;             // [0] cells listed, [1] keys handed out, [2] set when the list or the scratch ran out (zeroed before launch)
    uint16_t *deep_keys;            // [deep_key_cap] key scratch
    uint32_t deep_cap, deep_key_cap;
    uint16_t *keys;                 // NULL: one fused kernel; else [n_reads + 64] the keys between the phase-A and the phase-B launch
#ifdef BCFGPU_DIAG
    unsigned long long *stamps;     // [16] cycle totals per kernel phase
#endif
};

// errmod_cal's draw for over-deep cells (draw.hip): the generator's position, and the plan of the passes about to run
struct DrawState {
    uint64_t x = 0x1234ABCD330EULL;      // hts_drand48's state: htslib's default seed in a fresh process
    const uint32_t *rd[2] = {nullptr, nullptr};   // the read records of the planned SNP / indel tile (a plan serves one launch per pass)
    uint32_t *bits[2] = {nullptr, nullptr};       // their bitmaps
    uint32_t n_planned = 0;
};

struct CombineParams {
    int n_sites, n_smpl, is_indel, fmt_flag;
    const int8_t *ref16;
    CallretPlanes cr;
    const int *hist;
    const unsigned long long *site_sums;   // [n_sites][SITE_NSUM] from glfgen_kernel
    const double *mw;               // [6][6][50]
    BCFGPU_ABL_FIELD
    int vec4;                       // set by launch_combine: n_smpl % 4 == 0 and all planes 16-byte aligned
    bcfgpu_mplp_out out;
};

struct McallParams {
    int n_sites, n_smpl;
    int n_gt_max, n_al_max;         // plane counts of pl / ad
    int pl_is_u8;                   // 1: pl planes are the u8 planes of the mpileup stage (stride BCFGPU_MAX_PL)
    int call_flag, output_tags, n_grp;
    double theta;                   // log-scaled prior or 0
    const double *pl2p;
    const int32_t *nals, *unseen;   // per site (NULL with pl_is_u8: taken from msite)
    const bcfgpu_site *msite;       // mpileup-stage site structs (fused path) or NULL
    const void *pl;
    const float *qs;                // [site][5] or NULL (fused: msite->qsum)
    const int32_t *ad;              // i32 planes or NULL
    const uint16_t *ad_u16, *ad_u16b; const int32_t *qs_i32;   // fused -G sources (mpileup-stage ADF/ADR or QS planes, stride 5)
    const uint8_t *ploidy;
    const int32_t *grp;
    int32_t *grp_rng;               // [n_grp][3] workspace: first sample, last sample + 1 and number of samples of every group (launch_mcall fills it)
    float *grp_q;                   // [n_sites][n_grp][5] workspace: the groups' allele-frequency sums (grp_qsum_kernel -> mcall_kernel)
    const int32_t *prior_an, *prior_ac;
    const float *i16;               // [site][16] INFO/I16 or NULL (fused: msite->anno)
    bcfgpu_call_out out;
    int out_n_gt_max;               // plane count of out.pl / out.gp
    int *err;                       // device error word
    int small_too;                  // mcall_kernel<5, 15, ...> also takes the sites of at most three alleles (launch_mcall)
    BCFGPU_ABL_FIELD
};

// ---- the read pool of a region kept in HBM (bcfgpu_pool_upload; pileup.hip, baq.hip, overlap.hip, capmapq.hip) ----
// The caller's per-read arrays as they were handed over, the pools one byte per base; the stages replace `qual` / `r_mapq`
// in place or swing the pointer to a buffer of their own.
struct DevPool {
    int valid, n_reads;
    uint32_t n_bases, n_cig;
    const int32_t *r_pos, *r_lq, *r_flag, *r_ncig, *r_cig_off, *r_seq_off;
    uint8_t *r_mapq;
    const uint32_t *cig;
    const uint8_t *seq16;
    uint8_t *qual;
    uint8_t *zq, *r_has_zq;          // the "ZQ" bytes bcfgpu_pool_baq left and which reads have them; NULL before
    uint8_t *keep;                   // [n_reads] 0 = the read does not enter the pileup (bcfgpu_pool_keep); NULL = all do
    int ext_valid, ext_lo, ext_hi;   // [lowest start, highest end) of the reads on the reference, once something asked for it
    int qual_slot;                   // the workspace slot `qual` lives in (bcfgpu_pool_baq writes the new qualities to another one)
};

// ---- bcf_call_gap_prep on the device (gap_prep.hip, indel.hip) ----
// the caller's arrays in HBM (bcfgpu_reads, bcfgpu_indel_in) and the slice [ref_lo, ref_hi) of the contig the batch touches
struct GapIn {
    int n_sites, n_smpl, n_reads;
    const int32_t *pos, *smpl_off, *p_read, *p_qpos, *p_indel;
    const int32_t *r_pos, *r_lq, *r_flag, *r_ncig, *r_cig_off, *r_seq_off;
    const uint32_t *cig;
    const uint8_t *seq16, *qual, *zq, *r_has_zq;
    const char *ref; long ref_lo, ref_hi;        // positions outside the slice read as NUL
    int openQ, extQ, tandemQ, min_support, per_sample_flt;
    double min_frac;
};
// what bcf_call_gap_prep keeps in local variables for one position
struct GapSite {
    int32_t live;                                // 0: bcf_call_gap_prep returns -1 before the realignment
    int32_t n_types, ref_type, l_run, max_ins, N, indelreg, left, right, pos, max_rd_len, max_ref2;
    int32_t e0;                                  // first pileup entry of the site
    uint32_t job0, job_end;                      // jobs of the site: job0 + t*N + K (type-major); job_end: running total, all sites
    uint32_t ref2_0;                             // ref2 rows of the site: ref2_0 + (t*n_smpl + s)*max_ref2
    uint32_t ins0;                               // insertion consensus of the site: ins0 + t*max_ins
    uint32_t q8_0; int32_t qstride;              // packed queries of the site's entries: 8-byte unit q8_0 + K*(qstride/8); qstride = max_rd_len rounded up to 8
    int32_t types[64];                           // ascending (bam2bcf_indel.c:145-171); a dropped insertion becomes 0 (:279)
};
// per pileup entry: the read's sample (-1: the read is not realigned) and its window coordinates
struct GapEntry { int32_t smpl, qbeg, qend, tbeg, tend; uint32_t q8; };
struct GapTotals {
    unsigned long long n_jobs, ref2_bytes, ins_bytes;
    unsigned long long n_passes, dp_cells;       // statistics of the realignment (bcfgpu_gap_stats)
    unsigned long long qpack8;                   // 8-byte units of the packed-query pool
    int32_t max_L, max_bw, n_live, max_ref2, max_qstride, max_N;   // max_N: most pileup entries of a live site
    uint32_t n_wide;                             // jobs whose band fits neither the registers nor LDS (or whose lengths exceed 16 bits)
    int32_t max_eff;                             // their widest band
    uint32_t n_lds;                              // jobs of the LDS class (bands PROBALN_BW_MAX + 1 .. PROBALN_LDS_MAX)
};
// realignment jobs (indel.hip): bands of half-width PROBALN_BW_MIN..PROBALN_BW_MAX run with the row in registers, one kernel
// instantiation per width (narrower bands in the smallest); wider ones, up to PROBALN_LDS_MAX, with the row in LDS (class
// PROBALN_CLS_LDS, sorted by band width and cut into PROBALN_LDS_GROUPS launches by the LDS a wavefront needs; PROBALN_CLS_LDS16:
// sixteen jobs a wavefront for the bands no 64 jobs fit LDS with); what is wider still, or longer than the 16-bit fields of a
// PJob, from rolling rows in a global scratch buffer (PROBALN_CLS_WIDE)
#define PROBALN_BW_MIN 3
#define PROBALN_BW_MAX 10
#define PROBALN_LDS_MAX 73                       // 64 jobs a wavefront: (2 * 73 + 1 -> 152 cells + 2) x 64 lanes x 16 bytes + the emission table <= 160 KiB
#define PROBALN_LDS16_MAX 300                    // 16 jobs a wavefront: 610 cells x 16 lanes x 16 bytes
#define PROBALN_LDS16_SPLIT 103                  // ... in two launches: bands to 103 (210 cells: 58 KB, two workgroups a CU) and the rest
#define PROBALN_LDS_GROUPS 5
#define PROBALN_CLS_LDS 11u
#define PROBALN_CLS_LDS16 12u
#define PROBALN_CLS_WIDE 14u
#define PROBALN_CLS_NONE 15u
struct PJob { uint32_t ref_off, q8; uint16_t l_ref, l_query, eff, pad; };      // one decoded job: offsets into ref2 / the packed queries (8-byte units)
struct ProbalnQueue {
    uint32_t cls_begin[17];                      // first sorted slot of every class (class = key >> 13)
    uint32_t next1[16], next2[16];               // work counters of the two passes
    uint32_t n2[16];                             // jobs listed for the second parameter set, per class
    uint32_t lds_begin[PROBALN_LDS_GROUPS + 3];  // class PROBALN_CLS_LDS: first sorted slot of every band-width group; then class PROBALN_CLS_LDS16: bands to PROBALN_LDS16_SPLIT, wider
    uint32_t lds_next[PROBALN_LDS_GROUPS + 2];   // work counters of the groups
};
// the widest band of every group of the LDS class (host and device), and the cells a lane's column needs for a band: whole
// groups of eight cells and a guard cell at either end
#define PROBALN_LDS_CAPS {15, 31, 43, 58, PROBALN_LDS_MAX}     /* the first three groups run with the row in registers (4, 8, 11 groups of eight cells) */
#define PROBALN_LDS_CELLS(cap) ((((2 * (cap) + 1) + 7) & ~7) + 2)
struct ProbalnParams {
    GapIn gin;
    const GapSite *sites;
    const GapEntry *ent;
    int n_sites, n_jobs;
    const uint8_t *ref2;                         // realignment targets, base codes 0..4
    const uint8_t *qpack;                        // packed queries (gap_qpack_kernel)
    const float *q2p;                            // 10^(-q/10) as float, q = 0..255 (htslib g_qual2prob)
    double2 *emt;                                // [256] emission values by packed query byte (probaln_emt_kernel)
    int32_t *score1, *score2;                    // sc<<8 | norm, bam2bcf_indel.c:348-356
    PJob *pjob;                                  // [n_jobs]
    uint32_t *key_in, *val_in;                   // [n_jobs] sort keys and job numbers as probaln_jobs_kernel writes them
    const uint32_t *key_sorted, *val_sorted;     // after the radix sort
    uint32_t *list2;                             // [n_jobs] jobs of the second pass, class c from cls_begin[c] on
    ProbalnQueue *queue;
    uint32_t *wide; GapTotals *tot;              // the jobs left for the wide-band kernel
    uint32_t wide_first; int wide_count;         // wide-band launch: jobs wide[wide_first .. +wide_count)
    int ncell;                                   // scratch cells per row (>= 3*(2*bw+1)+6 for the widest band)
    size_t scratch_stride;                       // jobs per chunk; scratch is [2][ncell][stride] doubles
    double *scratch;
    int force_wide;                              // tests: every job through the rolling-row version
};
void launch_probaln_jobs(const ProbalnParams &p, hipStream_t s);
void launch_probaln_bounds(const ProbalnParams &p, hipStream_t s);
int launch_probaln_exact(const ProbalnParams &p, hipStream_t s, int n_cu, hipStream_t *side, hipEvent_t *ev);      // (the LDS class too)
void launch_probaln_wide(const ProbalnParams &p, hipStream_t s);
void launch_gap_entries(const GapIn &in, const GapSite *sites, int n_ent, GapEntry *ent, int max_qstride, uint8_t *qpack, hipStream_t s);

size_t glfgen_lds_bytes(int cap, int hist_slots);
void launch_glfgen(const GlfgenParams &p, hipStream_t s);
void launch_combine(const CombineParams &p, hipStream_t s);
void launch_mcall(const McallParams &p, hipStream_t s);

}  // namespace bcfgpu

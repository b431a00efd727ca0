/*  bcforacle.h -- CPU oracle for the mpileup -> call -m hot path.
 *
 *  TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 *  algorithm (bam2bcf.c, mcall.c and the htslib routines they call).  Only
 *  tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 *  it; the product library (bcftools_amd/csrc) never links or calls it.
 *
 *  Parity pins (see tests/test_oracle_golden_*.py):
 *    - glfgen + errmod_cal + combine: test/mpileup/mpileup.3.out from
 *      mpileup.1.sam (BAQ-free golden of the reference's own test-suite)
 *    - mcall: test/mpileup*.vcf, call-G*.vcf, call.af-fixation.vcf, mpileup.hwe.vcf,
 *      mpileup.X.vcf against their .out goldens
 *    - probaln_glocal / sam_prob_realn (BAQ) / bcf_call_gap_prep / mate-overlap tweak: test/mpileup/mpileup.{1,2,4,5}.out
 *      and indel-AD.1.out (goldens made with BAQ on), every record and field
 *
 *  It takes and fills the same SoA structures as the device library
 *  (include/bcfgpu.h), with HOST pointers.
 */
#ifndef BCFORACLE_H
#define BCFORACLE_H

#include "../include/bcfgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_errmod orc_errmod;

/* htslib errmod.c: errmod_init(depcorr) / errmod_cal() -- call sites bam2bcf.c:51,256 */
orc_errmod *orc_errmod_init(double depcorr);
void orc_errmod_destroy(orc_errmod *em);
int  orc_errmod_cal(const orc_errmod *em, int n, int m, uint16_t *bases, float *q);
/* cells deeper than 255 reads (see errmod.c): the reference's random draw (rule 0, default; orc_srand48_reset() puts the
 * generator back to the state a fresh process has) or the first 255 reads in pileup order (rule 1) */
void orc_errmod_deep_rule(int rule);
void orc_srand48_reset(void);
double orc_drand48(void);
uint64_t orc_rand48_state(void);
const double *orc_errmod_fk(const orc_errmod *em);
const double *orc_errmod_beta(const orc_errmod *em);
const double *orc_errmod_lhet(const orc_errmod *em);

/* htslib kfunc.c */
double orc_kf_erfc(double x);
double orc_kf_betai(double a, double b, double x);
/* test16 (ccall.c:103-138) on INFO/I16: the four PV4 p-values; is_tested as in anno16_t; returns -1 at depth 0 */
int orc_test16(const float anno[16], double p[4], int *is_tested);
double orc_kt_fisher_exact(int n11, int n12, int n21, int n22, double *left, double *right, double *two);

/* bam2bcf.c:281-530 */
double orc_calc_vdb(const int *pos, int npos);
double orc_calc_mwu_bias(const int *a, const int *b, int n);

/* Per (site,sample) result of bcf_call_glfgen, for kernel-level checks */
typedef struct {
    float    p[25];
    double   anno[16];
    int32_t  QS[4], ADF[4], ADR[4];
    int32_t  SCR, n;          /* n = return value of bcf_call_glfgen (-1 when _n==0) */
    uint32_t ori_depth, mq0;
} orc_callret;

/* mpileup stage over a tile: bcf_callaux_clean + glfgen x n_smpl + combine, per site
 * (mpileup.c:343-347).  `ret_dbg` (may be NULL) receives [n_sites*n_smpl] callrets.
 * Returns 0 or BCFGPU_E_*. */
int orc_mpileup(const bcfgpu_cfg *cfg, const bcfgpu_tile *tile, const bcfgpu_mplp_out *out, orc_callret *ret_dbg);

/* call stage over a tile: mcall() per site (mcall.c:1430-1684) */
int orc_mcall(const bcfgpu_cfg *cfg, const bcfgpu_call_in *in, const bcfgpu_call_out *out);

/* htslib probaln.c / realn.c (see probaln.c) */
int orc_probaln_glocal(const uint8_t *ref, int l_ref, const uint8_t *query, int l_query, const uint8_t *iqual,
                       double d, double e, int bw, int *state, uint8_t *q);
int orc_sam_prob_realn(int pos, int l_qseq, const uint8_t *seq4, uint8_t *qual, const uint32_t *cigar, int n_cigar,
                       const char *ref, int ref_len, int flag, uint8_t *zq);

/* bcf_call_gap_prep (bam2bcf_indel.c:99-470) over a flat read model, see indel_oracle.c */
int orc_gap_prep(int n, const int *smpl_off, const int *p_read, const int *p_qpos, const int *p_indel,
                 const int *r_pos, const int *r_lq, const int *r_flag, const int *r_ncig, const int *r_cig_off,
                 const uint32_t *cig, const int *r_seq_off, const uint8_t *seq16, const uint8_t *qualp,
                 const uint8_t *zqp, const uint8_t *r_has_zq,
                 int pos, const char *ref, int openQ, int extQ, int tandemQ, int min_support, double min_frac,
                 int per_sample_flt,
                 uint32_t *p_aux, int *indel_types, char *inscns_out, int inscns_cap, int *maxins_out, int *indelreg_out,
                 int *max_support_out, float *max_frac_out);

/* FORMAT/SP from DP4 (bam2bcf.c:867-885) */
int orc_format_sp(int fwd_ref, int rev_ref, int fwd_alt, int rev_alt);
/* mate-overlap quality tweak (htslib sam.c tweak_overlap_quality; mpileup.c:640) on the read pool: pair p is
 * (pair_a[p] arrived first, pair_b[p]); `qual` is the pool of qualities, modified in place */
int orc_overlap_tweak(const bcfgpu_reads *rd, int32_t n_pairs, const int32_t *pair_a, const int32_t *pair_b, uint8_t *qual);

/* gvcf_write (gvcf.c:88-226) over n records: dp [n][S], pl [n][3][S] (FORMAT/DP and FORMAT/PL of the records), is_ref as
 * mpileup.c:309-315 decides it; outputs as bcfgpu_gvcf_out but with int32 PL.  Returns the number of blocks. */
int orc_gvcf_blocks(int n, int S, const int32_t *pos, const int32_t *rid, const uint8_t *brk, const uint8_t *is_ref,
                    const int32_t *dp, const int32_t *pl, const int32_t *dp_range, int n_range,
                    int32_t *blk, int32_t *min_dp_out, bcfgpu_gvcf_block *block, int32_t *dp_out, int32_t *pl_out);

#ifdef __cplusplus
}
#endif
#endif

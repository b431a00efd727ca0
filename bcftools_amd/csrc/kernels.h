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
#define BCFGPU_ABL_FIELD int ablate;
#else
#define BCFGPU_ABL(P, bit) false
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
struct CallretPlanes {
    float    *p15;    // [15][ncells]  upper triangle of p[5][5]: index k*(k+1)/2+j for j<=k
    uint64_t *qs64;   // [ncells]      QS[0..3] packed 4 x u16
    uint32_t *adf;    // [ncells]      ADF[0..3] packed 4 x u8
    uint32_t *adr;    // [ncells]      ADR[0..3] packed 4 x u8
    uint32_t *cnt4;   // [ncells]      anno[0..3] packed 4 x u8
    uint32_t *misc;   // [ncells]      mq0 | SCR<<8 | ori_depth<<16
};

struct GlfgenParams {
    int n_sites, n_smpl, is_indel;
    int min_baseQ, capQ, fmt_flag;
    int hist_slots;                 // >0: per-workgroup LDS histograms with that many site slots; 0: global atomics
    int lds_cap;                    // read keys held in LDS per workgroup round (multiple of 16, <= 16384)
    uint32_t n_reads;               // length of rd/epos (bounds of the vector loads)
    BCFGPU_ABL_FIELD
    const int8_t   *ref16;
    const uint32_t *off;
    const uint32_t *rd;
    const uint8_t  *epos;
    const uint32_t *aux;
    const double *fk, *beta, *lhet;
    CallretPlanes cr;
    int *hist;                      // [n_sites][H_SIZE], zeroed before launch
    unsigned long long *site_sums;  // [n_sites][SITE_NSUM] site totals of anno[4..15], ori_depth, mq0 (exact integers), zeroed before launch
    int *err;                       // device error word
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
    const uint8_t *ad_u8, *ad_u8b; const uint16_t *qs_u16;   // fused -G sources (mpileup-stage ADF/ADR or QS planes, stride 5)
    const uint8_t *ploidy;
    const int32_t *grp;
    const int32_t *prior_an, *prior_ac;
    const float *i16;               // [site][16] INFO/I16 or NULL (fused: msite->anno)
    bcfgpu_call_out out;
    int out_n_gt_max;               // plane count of out.pl / out.gp
    BCFGPU_ABL_FIELD
};

// one realignment job of bcf_call_gap_prep: probaln_glocal(ref2+ref_off, l_ref, query+query_off, l_query, qq+query_off, {.., bw})
struct ProbalnJob { uint32_t ref_off, query_off; int32_t l_ref, l_query, bw, flags; };   // query_off: into the reads' seq16/qual pools; flags&1: ZQ present
struct ProbalnParams {
    int n_jobs, ncell;              // ncell: scratch cells per row (>= 3*(2*bw+1)+6 for the widest band)
    int force_scratch;              // diagnostics build only (-DBCFGPU_DIAG): every job through the rolling-row version
    size_t scratch_stride;          // jobs rounded up; scratch is [2][ncell][stride] doubles
    const ProbalnJob *jobs;
    const uint8_t *ref2, *query, *qq, *zq;   // consensus windows (0..4 codes); the reads' seq16 / qual / ZQ pools as the caller holds them
    const float *q2p;               // 10^(-q/10) as float, q = 0..255 (htslib g_qual2prob)
    double *scratch;
    int32_t *score1, *score2;       // sc<<8 | norm, bam2bcf_indel.c:348-356
};
void launch_probaln(const ProbalnParams &p, hipStream_t s);
// host-side job pools of bcfgpu_gap_prep, one per preparing thread; offsets inside a pool are pool-relative until rebased
struct ProbalnPools { std::vector<ProbalnJob> jobs; std::vector<uint8_t> ref2pool; int max_bw = 0; };

size_t glfgen_lds_bytes(int cap, int hist_slots);
void launch_glfgen(const GlfgenParams &p, hipStream_t s);
void launch_combine(const CombineParams &p, hipStream_t s);
void launch_mcall(const McallParams &p, hipStream_t s);

}  // namespace bcfgpu

// api.hip -- the C-ABI of include/bcfgpu.h: context, device-memory helpers, launch sequencing.
//
// There is deliberately no CPU fallback here: without a HIP device bcfgpu_create() fails with
// BCFGPU_E_NODEV (the only host-side compute entry, bcfgpu_pack_read, is the data packer).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <algorithm>
#include "kernels.h"
#include "tables.h"

using namespace bcfgpu;

static thread_local std::string g_err;
static int set_err(int code, const char *what, hipError_t e = hipSuccess)
{
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof buf, "%s", what);
    g_err = buf;
    return code;
}
int bcfgpu_set_error(int code, const char *what) { return set_err(code, what); }
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return set_err(BCFGPU_E_HIP, #call, e_); } while (0)

struct bcfgpu_ctx {
    bcfgpu_cfg cfg;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // constant tables in HBM
    double *d_fk = nullptr, *d_beta = nullptr, *d_lhet = nullptr, *d_pl2p = nullptr, *d_mw = nullptr;
    float *d_q2p = nullptr;         // 10^(-q/10) as float (htslib g_qual2prob), for the realignment kernel
    double call_theta_log = 0;
    // workspaces sized by cfg.max_sites / cfg.max_reads
    int *d_hist = nullptr, *d_err = nullptr;       // d_err: [0] error word, [1] cells past 255 usable reads, [2..4] counters of glfgen's deep-cell list, [5] WideRecs of the launch
    uint16_t *d_keys = nullptr;                    // glfgen in two launches (BCFGPU_GLFGEN_SPLIT=1): 2 bytes per read between them
    int32_t *d_grp_rng = nullptr;                  // mcall: sample range of every -G group
    float *d_grp_q = nullptr; size_t grp_q_bytes = 0;   // mcall -G: the groups' frequency sums of a launch (grow-only)
    uint32_t *d_deep_list = nullptr; uint16_t *d_deep_keys = nullptr; uint32_t deep_cap = 0, deep_key_cap = 0, wide_cap = 0;
    CallretPlanes *d_crp = nullptr;
    unsigned long long *d_site_sums = nullptr;
    CallretPlanes cr{};
    size_t ncells_cap = 0;
    // timing
    int timing = 0;                 // 1: time every launch sequence and wait for it; 2: record only, resolve in timing_get
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bcfgpu_timing last{};
    std::vector<hipEvent_t> pool;   // mode 2: 4 events per launch sequence; created once and reused after every timing_get
    size_t pool_used = 0;           // events handed out since the last timing_get
    std::vector<int> pool_call;     // mode 2: 1 if the sequence included the call kernel
    std::vector<void*> owned;
    bcfgpu_gap_stats gap{};         // statistics of the last bcfgpu_gap_prep
    // grow-only device workspaces of the indel / BAQ stages (GiB-sized scratch: not reallocated per call)
    struct Ws { void *p = nullptr; size_t bytes = 0; };
    alignas(16) unsigned char pileup_state[256] = {0};   // csrc/pileup.hip: the parameters of the last bcfgpu_pileup
    alignas(16) unsigned char pool_state[256] = {0};     // kernels.h DevPool: the read pool bcfgpu_pool_upload left in HBM
    Ws ws[152];                    // grow-only device workspaces of the host-fed stages (0-15: BAQ / overlaps, 16-35: pileup, 36-39: gVCF / indel tile, 40-103: gap_prep, 104-135: the resident read pool and its stages, 136-143: errmod_cal's draw, 144-151: BAQ's smaller band classes, which run beside the main one)
    int n_cu = 256;                // compute units of the device (grid size of the work-queue kernels)
    hipStream_t side[8] = {};      // created on first use: the realignment kernels of different band widths run side by side
    hipEvent_t side_ev[9] = {};    // [0..7] a side stream's work is done, [8] the fork point on the main stream
    Ws pinned[8];                  // grow-only pinned host staging buffers
    DrawState draw;                // errmod_cal's generator and the plan of the next launches (draw.hip)
};

extern "C" {

void *bcfgpu_internal_ws(bcfgpu_ctx *c, int slot, size_t bytes);

const char *bcfgpu_last_error(void) { return g_err.c_str(); }

int bcfgpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void bcfgpu_abi_sizes(int32_t *out)
{
    out[0] = (int32_t)sizeof(bcfgpu_cfg); out[1] = (int32_t)sizeof(bcfgpu_tile); out[2] = (int32_t)sizeof(bcfgpu_site);
    out[3] = (int32_t)sizeof(bcfgpu_mplp_out); out[4] = (int32_t)sizeof(bcfgpu_call_in); out[5] = (int32_t)sizeof(bcfgpu_call_site);
    out[6] = (int32_t)sizeof(bcfgpu_call_out); out[7] = (int32_t)sizeof(bcfgpu_timing);
}

static int dev_alloc(bcfgpu_ctx *c, void **p, size_t bytes)
{
    *p = nullptr;
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return set_err(BCFGPU_E_NOMEM, "hipMalloc", e);
    c->owned.push_back(*p);
    return 0;
}

int bcfgpu_create(const bcfgpu_cfg *cfg, bcfgpu_ctx **out)
{
    if (!cfg || !out || cfg->n_smpl <= 0) return set_err(BCFGPU_E_ARG, "bcfgpu_create: bad arguments");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_err(BCFGPU_E_NODEV, "bcfgpu_create: no HIP device (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return set_err(BCFGPU_E_ARG, "bcfgpu_create: device ordinal out of range");
    HIPCHK(hipSetDevice(cfg->device));
    bcfgpu_ctx *c = new bcfgpu_ctx();
    c->cfg = *cfg;
    if (c->cfg.capQ <= 0) c->cfg.capQ = 60;
    if (c->cfg.capQ > 63) { delete c; return set_err(BCFGPU_E_ARG, "bcfgpu_create: capQ must be <= 63 (the reference fixes it at 60, bam2bcf.c:48)"); }
    if (c->cfg.min_baseQ < 0) c->cfg.min_baseQ = 0;
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return set_err(BCFGPU_E_HIP, "hipStreamCreate", e); }
    c->stream = c->own_stream;
    for (int i = 0; i < 4; ++i) hipEventCreate(&c->ev[i]);
    { int ncu = 0; if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && ncu > 0) c->n_cu = ncu; }

    // tables
    const double theta = cfg->errmod_theta <= 0. ? 0.83 : cfg->errmod_theta;     // CALL_DEFTHETA, bam2bcf.c:38,46
    std::vector<double> fk, beta, lhet;
    build_errmod_tables(1. - theta, fk, beta, lhet);
    double pl2p[256], mw[6 * 6 * 50];
    build_pl2p(pl2p);
    build_mw_table(mw);
    int rc = 0;
    if ((rc = dev_alloc(c, (void**)&c->d_fk, fk.size() * 8)) || (rc = dev_alloc(c, (void**)&c->d_beta, beta.size() * 8 + 64)) ||
        (rc = dev_alloc(c, (void**)&c->d_lhet, lhet.size() * 8)) || (rc = dev_alloc(c, (void**)&c->d_pl2p, sizeof pl2p)) ||
        (rc = dev_alloc(c, (void**)&c->d_mw, sizeof mw)) || (rc = dev_alloc(c, (void**)&c->d_q2p, 256 * sizeof(float))) || (rc = dev_alloc(c, (void**)&c->d_err, 8 * sizeof(int)))) {
        bcfgpu_destroy(c); return rc;
    }
    if (cfg->n_grp > 1 && (rc = dev_alloc(c, (void**)&c->d_grp_rng, (size_t)cfg->n_grp * 3 * sizeof(int32_t)))) { bcfgpu_destroy(c); return rc; }
    hipMemcpy(c->d_fk, fk.data(), fk.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(c->d_beta, beta.data(), beta.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(c->d_lhet, lhet.data(), lhet.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(c->d_pl2p, pl2p, sizeof pl2p, hipMemcpyHostToDevice);
    hipMemcpy(c->d_mw, mw, sizeof mw, hipMemcpyHostToDevice);
    { float q2p[256]; for (int i = 0; i < 256; ++i) q2p[i] = (float)std::pow(10., -i / 10.); hipMemcpy(c->d_q2p, q2p, sizeof q2p, hipMemcpyHostToDevice); }
    hipMemset(c->d_err, 0, 8 * sizeof(int));

    // the prior: theta <- log(theta * sum_{i<n} 1/i), n = ploidy_max * nsamples (mcall.c:396-416, vcfcall.c:654-655)
    c->call_theta_log = 0;
    if (cfg->call_theta > 0) {
        const int pm = cfg->ploidy_max > 0 ? cfg->ploidy_max : 2;
        const int n = pm * cfg->n_smpl;
        double aM = 1;
        for (int i = 2; i < n; i++) aM += 1. / i;
        double t = cfg->call_theta * aM;
        if (t >= 1) t = 0.99;
        c->call_theta_log = std::log(t);
    }

    // workspaces
    const size_t ncells = (size_t)(cfg->max_sites > 0 ? cfg->max_sites : 0) * cfg->n_smpl;
    c->ncells_cap = ncells;
    if (ncells) {
        if ((rc = dev_alloc(c, (void**)&c->d_hist, (size_t)cfg->max_sites * H_SIZE * sizeof(int))) ||
            (rc = dev_alloc(c, (void**)&c->cr.p15, ncells * 16 * sizeof(float))) ||
            (rc = dev_alloc(c, (void**)&c->cr.qs64, ncells * 8)) ||
            (rc = dev_alloc(c, (void**)&c->cr.adf, ncells * 4)) || (rc = dev_alloc(c, (void**)&c->cr.adr, ncells * 4)) ||
            (rc = dev_alloc(c, (void**)&c->cr.cnt4, ncells * 4)) || (rc = dev_alloc(c, (void**)&c->d_site_sums, (size_t)cfg->max_sites * SITE_NSUM * 8)) ||
            (rc = dev_alloc(c, (void**)&c->cr.misc, ncells * 4)) || (rc = dev_alloc(c, (void**)&c->cr.pa, ncells * 4))) {
            bcfgpu_destroy(c); return rc;
        }
        // glfgen's list of cells deeper than its LDS key window, and the scratch their keys go to (2 bytes per pileup entry):
        // up to 1024 such cells per tile, with at most 32 Mi entries between them (or the whole tile's, when it is smaller)
#ifdef BCFGPU_DIAG      // the two-launch form of glfgen is a measurement variant (make DIAG=1), never the product path
        if (const char *sp = getenv("BCFGPU_GLFGEN_SPLIT")) if (atoi(sp) && (rc = dev_alloc(c, (void**)&c->d_keys, ((size_t)cfg->max_reads + 64) * 2))) { bcfgpu_destroy(c); return rc; }
#endif
        c->deep_cap = 1024;
        c->deep_key_cap = (uint32_t)std::min<uint64_t>((uint64_t)cfg->max_reads + 16 * 1024, 32u << 20);
        if ((rc = dev_alloc(c, (void**)&c->d_deep_list, (size_t)c->deep_cap * 8)) || (rc = dev_alloc(c, (void**)&c->d_deep_keys, (size_t)c->deep_key_cap * 2 + 64))) {
            bcfgpu_destroy(c); return rc;
        }
        // the records of cells past 255 usable reads (kernels.h WideRec): a tile of n reads has at most n / 256 such cells
        c->wide_cap = (uint32_t)std::min<uint64_t>((uint64_t)cfg->max_reads / 256 + 1, 1u << 26);
        if ((rc = dev_alloc(c, (void**)&c->cr.wide, (size_t)c->wide_cap * sizeof(WideRec)))) { bcfgpu_destroy(c); return rc; }
        // the table of plane addresses as glfgen_kernel reads it at the point of its stores (kernels.h: GlfgenParams::crp)
        if ((rc = dev_alloc(c, (void**)&c->d_crp, sizeof(CallretPlanes)))) { bcfgpu_destroy(c); return rc; }
        e = hipMemcpy(c->d_crp, &c->cr, sizeof(CallretPlanes), hipMemcpyHostToDevice);
        if (e != hipSuccess) { bcfgpu_destroy(c); return set_err(BCFGPU_E_HIP, "plane table upload", e); }
    }
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { bcfgpu_destroy(c); return set_err(BCFGPU_E_HIP, "table upload", e); }
    *out = c;
    return BCFGPU_OK;
}

void bcfgpu_destroy(bcfgpu_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->cfg.device);
    if (c->own_stream) hipStreamSynchronize(c->own_stream);
    for (void *p : c->owned) hipFree(p);
    if (c->d_grp_q) hipFree(c->d_grp_q);
    for (auto &w : c->ws) if (w.p) hipFree(w.p);
    for (auto &w : c->pinned) if (w.p) hipHostFree(w.p);
    for (hipEvent_t e : c->pool) hipEventDestroy(e);
    for (int i = 0; i < 4; ++i) if (c->ev[i]) hipEventDestroy(c->ev[i]);
    for (int i = 0; i < 8; ++i) if (c->side[i]) { hipStreamSynchronize(c->side[i]); hipStreamDestroy(c->side[i]); }
    for (int i = 0; i < 9; ++i) if (c->side_ev[i]) hipEventDestroy(c->side_ev[i]);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
}

int bcfgpu_malloc(bcfgpu_ctx *c, size_t bytes, void **dptr)
{
    if (!c || !dptr) return set_err(BCFGPU_E_ARG, "bcfgpu_malloc: bad arguments");
    hipSetDevice(c->cfg.device);
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 16);
    if (e != hipSuccess) return set_err(BCFGPU_E_NOMEM, "hipMalloc", e);
    return 0;
}
int bcfgpu_free(bcfgpu_ctx *c, void *dptr)
{
    if (!c) return set_err(BCFGPU_E_ARG, "bcfgpu_free: bad arguments");
    HIPCHK(hipFree(dptr));
    return 0;
}
int bcfgpu_memcpy_h2d(bcfgpu_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c) return set_err(BCFGPU_E_ARG, "bcfgpu_memcpy_h2d: bad arguments");
    if (!bytes) return 0;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
int bcfgpu_memcpy_d2h(bcfgpu_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c) return set_err(BCFGPU_E_ARG, "bcfgpu_memcpy_d2h: bad arguments");
    if (!bytes) return 0;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
int bcfgpu_memset(bcfgpu_ctx *c, void *dst, int value, size_t bytes)
{
    if (!c) return set_err(BCFGPU_E_ARG, "bcfgpu_memset: bad arguments");
    if (!bytes) return 0;
    HIPCHK(hipMemsetAsync(dst, value, bytes, c->stream));
    return 0;
}
int bcfgpu_sync(bcfgpu_ctx *c)
{
    if (!c) return set_err(BCFGPU_E_ARG, "bcfgpu_sync: bad arguments");
    HIPCHK(hipStreamSynchronize(c->stream));
    int err = 0;
    HIPCHK(hipMemcpy(&err, c->d_err, sizeof(int), hipMemcpyDeviceToHost));
    if (err) {
        hipMemset(c->d_err, 0, sizeof(int));
        return set_err(err, err == BCFGPU_E_DEPTH ? "a (site,sample) cell holds more pileup entries than the scratch for over-deep cells takes, or more than 65535 reads of one base and strand" :
                            err == BCFGPU_E_RANGE ? "a record is outside the supported range (more than 5 alleles, or more genotypes / alleles than the planes hold): its ret is -2" : "device-side error");
    }
    return 0;
}
int bcfgpu_truncated_cells(bcfgpu_ctx *c, uint32_t *n_cells)
{
    if (!c || !n_cells) return set_err(BCFGPU_E_ARG, "bcfgpu_truncated_cells: bad arguments");
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(n_cells, c->d_err + 1, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (*n_cells) hipMemset(c->d_err + 1, 0, sizeof(int));
    return 0;
}
int bcfgpu_set_stream(bcfgpu_ctx *c, void *hip_stream)
{
    if (!c) return set_err(BCFGPU_E_ARG, "bcfgpu_set_stream: bad arguments");
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return 0;
}

// get_position (bam2bcf.c:80-114) + the per-read fields of bcf_call_glfgen
void bcfgpu_pack_read(int nt16, int baseQ, int mapQ, int is_rev, int has_softclip, int is_del,
                      int is_refskip_or_unmapped, int qpos, int l_qseq, const uint32_t *cigar, int n_cigar,
                      int want_epos, uint32_t *rd, uint8_t *epos)
{
    int tail = l_qseq - 1 - qpos;
    if (tail > qpos) tail = qpos;
    if (tail < 0) tail = 0;
    if (tail > 255) tail = 255;
    *rd = (uint32_t)(baseQ & 0xff) | (uint32_t)(mapQ & 0xff) << 8 | (uint32_t)(nt16 & 0xf) << 16
        | (is_rev ? BCFGPU_RD_REV : 0) | (has_softclip ? BCFGPU_RD_SCLIP : 0) | (is_del ? BCFGPU_RD_DEL : 0)
        | (is_refskip_or_unmapped ? BCFGPU_RD_SKIP : 0) | (uint32_t)tail << 24;
    int e = 0;
    if (want_epos && cigar) {
        int n_tot_bases = 0, iread = 0, edist = qpos + 1;
        for (int ic = 0; ic < n_cigar; ic++) {
            const int op = cigar[ic] & 0xf, len = (int)(cigar[ic] >> 4);
            if (op == 0 /*M*/ || op == 7 /*=*/ || op == 8 /*X*/ || op == 1 /*I*/) { n_tot_bases += len; iread += len; }
            else if (op == 4 /*S*/) { iread += len; if (iread <= qpos) edist -= len; }
        }
        e = (int)((double)edist / (n_tot_bases + 1) * BCFGPU_NPOS);
        if (e < 0) e = 0;
        if (e > BCFGPU_NPOS - 1) e = BCFGPU_NPOS - 1;
    }
    *epos = (uint8_t)e;
}

// htslib sam.c bam_plp_push: "if (iter->tid == b->core.tid && iter->pos == b->core.pos && iter->mp->cnt > iter->maxcnt)" the
// read is dropped.  bam_plp_auto pushes a read only when the buffer cannot yield the next column, i.e. iter->pos is the start of
// the read kept last and every earlier column has been handed out: the buffer then holds the kept reads that end at or after
// that position (reads are released at the first column they no longer cover), and mp->cnt counts them plus the list's tail node.
struct bcfgpu_depth_state {
    struct St { std::vector<int32_t> ends; int32_t last_pos = INT32_MIN; };      // ends: a min-heap of the buffered reads' ends
    std::vector<St> st;
    int32_t max_depth;
};
bcfgpu_depth_state *bcfgpu_depth_cap_new(int32_t n_smpl, int32_t max_depth)
{
    if (n_smpl <= 0) { set_err(BCFGPU_E_ARG, "bcfgpu_depth_cap_new: bad arguments"); return nullptr; }
    bcfgpu_depth_state *d = new bcfgpu_depth_state();
    d->st.resize(n_smpl); d->max_depth = max_depth;
    return d;
}
void bcfgpu_depth_cap_free(bcfgpu_depth_state *d) { delete d; }
void bcfgpu_depth_cap_reset(bcfgpu_depth_state *d) { if (d) for (auto &S : d->st) { S.ends.clear(); S.last_pos = INT32_MIN; } }
int bcfgpu_depth_cap_push(bcfgpu_depth_state *d, const bcfgpu_reads *rd, const int32_t *r_smpl, uint8_t *keep)
{
    if (!d || !rd || !keep || rd->n_reads < 0 || (rd->n_reads && !r_smpl)) return set_err(BCFGPU_E_ARG, "bcfgpu_depth_cap_push: bad arguments");
    const int n = rd->n_reads, n_smpl = (int)d->st.size(), max_depth = d->max_depth;
    if (max_depth <= 0) { for (int r = 0; r < n; ++r) keep[r] = 1; return 0; }
    auto cmp = [](int32_t a, int32_t b) { return a > b; };
    for (int r = 0; r < n; ++r) {
        const int s = r_smpl[r];
        if (s < 0 || s >= n_smpl) return set_err(BCFGPU_E_ARG, "bcfgpu_depth_cap: sample index out of range");
        bcfgpu_depth_state::St &S = d->st[s];
        const int32_t p = rd->r_pos[r];
        if (p < S.last_pos) return set_err(BCFGPU_E_ARG, "bcfgpu_depth_cap: the reads of a sample are not in position order");
        int32_t e = p;                                          // pos + bam_cigar2rlen: the raw end, not bam_endpos (which makes a read
        const uint32_t *cg = rd->cig + rd->r_cig_off[r];        // without reference bases one long; bam_plp_push says so too)
        for (int k = 0; k < rd->r_ncig[r]; ++k) { const int op = cg[k] & 0xf; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) e += (int32_t)(cg[k] >> 4); }
        while (!S.ends.empty() && S.ends.front() < p) { std::pop_heap(S.ends.begin(), S.ends.end(), cmp); S.ends.pop_back(); }
        if (p == S.last_pos && (int64_t)S.ends.size() + 1 > max_depth) { keep[r] = 0; continue; }
        keep[r] = 1;
        // linked into the buffer only when it ends past the iterator's position (= the start of the read kept last): a read
        // without reference bases that is not the first of its start position is not (it never reaches a column either way)
        if (e > S.last_pos) { S.ends.push_back(e); std::push_heap(S.ends.begin(), S.ends.end(), cmp); }
        S.last_pos = p;
    }
    return 0;
}
int bcfgpu_depth_cap(const bcfgpu_reads *rd, const int32_t *r_smpl, int32_t n_smpl, int32_t max_depth, uint8_t *keep)
{
    if (!rd || !keep || rd->n_reads < 0 || (rd->n_reads && !r_smpl) || n_smpl <= 0) return set_err(BCFGPU_E_ARG, "bcfgpu_depth_cap: bad arguments");
    bcfgpu_depth_state d;
    d.st.resize(n_smpl); d.max_depth = max_depth;
    return bcfgpu_depth_cap_push(&d, rd, r_smpl, keep);
}

size_t bcfgpu_mplp_out_bytes(const bcfgpu_ctx *c, int n_sites, int which)
{
    if (!c || n_sites < 0) return 0;
    const size_t S = c->cfg.n_smpl, n = n_sites;
    switch (which) {
        case 0: return n * sizeof(bcfgpu_site);
        case 1: return n * BCFGPU_MAX_PL * S;
        case 2: return n * 4 * S * 2;
        case 3: case 4: return n * 5 * S * 2;
        case 5: return n * 5 * S * 4;
        case 6: return n * S * 2;
        case 7: return n * S;
    }
    return 0;
}

int bcfgpu_host_alloc(size_t bytes, void **ptr)
{
    if (!ptr) return set_err(BCFGPU_E_ARG, "bcfgpu_host_alloc: NULL");
    *ptr = nullptr;
    hipError_t e = hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return set_err(BCFGPU_E_NOMEM, "hipHostMalloc", e);
    return 0;
}
int bcfgpu_host_free(void *ptr)
{
    if (!ptr) return 0;
    hipError_t e = hipHostFree(ptr);
    return e == hipSuccess ? 0 : set_err(BCFGPU_E_HIP, "hipHostFree", e);
}

int bcfgpu_timing_enable(bcfgpu_ctx *c, int on)
{
    if (!c) return BCFGPU_E_ARG;
    c->timing = on;
    if (on == 2) while (c->pool.size() < 4 * 64) { hipEvent_t e; hipEventCreate(&e); c->pool.push_back(e); }   // 64 sequences' worth up front
    return 0;
}
int bcfgpu_timing_get(bcfgpu_ctx *c, bcfgpu_timing *t)
{
    if (!c || !t) return set_err(BCFGPU_E_ARG, "bcfgpu_timing_get: bad arguments");
    if (c->timing == 2) {
        // averages over every launch sequence recorded since the last call
        bcfgpu_timing a{};
        const size_t n = c->pool_call.size();
        if (n) {
            HIPCHK(hipStreamSynchronize(c->stream));
            int ncall = 0;
            for (size_t k = 0; k < n; ++k) {
                hipEvent_t *e = &c->pool[4 * k];
                float g = 0, m = 0, q = 0, tot = 0;
                hipEventElapsedTime(&g, e[0], e[1]);
                hipEventElapsedTime(&m, e[1], e[2]);
                if (c->pool_call[k]) { hipEventElapsedTime(&q, e[2], e[3]); hipEventElapsedTime(&tot, e[0], e[3]); ncall++; }
                else hipEventElapsedTime(&tot, e[0], e[2]);
                a.glfgen_ms += g; a.combine_ms += m; a.mcall_ms += q; a.total_ms += tot;
            }
            a.glfgen_ms /= n; a.combine_ms /= n; a.total_ms /= n;
            if (ncall) a.mcall_ms /= ncall;
            c->pool_used = 0; c->pool_call.clear();              // the events stay: the next sequences record into them again
        }
        c->last = a;
    }
    *t = c->last;
    return 0;
}

// the four events of the current launch sequence
static hipEvent_t *seq_events(bcfgpu_ctx *c)
{
    if (c->timing == 2) {
        // no event is created inside a timed loop once the pool has grown to the loop's length (timing_enable(2) pre-grows it)
        while (c->pool.size() < c->pool_used + 4) { hipEvent_t e; hipEventCreate(&e); c->pool.push_back(e); }
        c->pool_call.push_back(0);
        c->pool_used += 4;
        return &c->pool[c->pool_used - 4];
    }
    return c->ev;
}

static int check_tile(bcfgpu_ctx *c, const bcfgpu_tile *t)
{
    if (!t || t->n_sites < 0) return set_err(BCFGPU_E_ARG, "bad tile");
    if (t->n_sites > c->cfg.max_sites || t->n_reads > c->cfg.max_reads)
        return set_err(BCFGPU_E_RANGE, "tile exceeds the capacity the context was created with");
    if (t->n_sites && (!t->plp_off || (!t->is_indel && !t->ref16) || (t->n_reads && (!t->rd || !t->epos))))
        return set_err(BCFGPU_E_ARG, "tile: NULL array");
    if (t->is_indel && t->n_reads && !t->aux) return set_err(BCFGPU_E_ARG, "indel tile without aux");
    if (((uintptr_t)t->rd & 15) || ((uintptr_t)t->epos & 15) || (t->is_indel && ((uintptr_t)t->aux & 15)))
        return set_err(BCFGPU_E_ARG, "tile: rd/epos/aux must be 16-byte aligned");
    if (t->n_reads >> 32) return set_err(BCFGPU_E_RANGE, "tile: more than 2^32-1 reads");
    return 0;
}

static int enqueue_mpileup(bcfgpu_ctx *c, const bcfgpu_tile *tile, const bcfgpu_mplp_out *out)
{
    const int S = c->cfg.n_smpl;
    GlfgenParams g{};
    g.n_sites = tile->n_sites; g.n_smpl = S; g.is_indel = tile->is_indel;
    g.min_baseQ = c->cfg.min_baseQ; g.capQ = c->cfg.capQ; g.fmt_flag = c->cfg.fmt_flag;
    // one workgroup = 256 consecutive cells = at most (255/S)+2 sites
    const int slots = 255 / S + 2;
    g.hist_slots = slots <= 8 ? slots : 0;
    // LDS: two bytes per read of the workgroup's span, sized from the tile's mean depth (a 256-cell sum: +4 % and a
    // constant cover its spread) so that shallow tiles leave room for more workgroups per CU.  160 KiB are handed out in
    // 128 blocks of 1280 bytes: the span gets what lets the most workgroups (at most six: the kernel runs five to six
    // wavefronts per SIMD) share a CU while still holding `want` keys; a span that does not fit is worked off in rounds.
    {
        const double mean = (double)tile->n_reads / ((double)tile->n_sites * S);
        long want = (long)(mean * 256 * 1.03) + 64;
        want = std::max(2048L, (want + 15) & ~15L);
        int cap = 0;
        for (int wgs = 6; wgs >= 3 && !cap; --wgs) {
            const long budget = (long)(128 / wgs) * 1280 - 32;
            long c = (budget - (long)glfgen_lds_bytes(0, g.hist_slots)) / 2;
            c = std::min(c & ~15L, 16384L);
            if (c >= want || wgs == 3) cap = (int)std::max(c, 2048L);     // the whole window of the tier: `want` only chooses the tier
        }
        g.lds_cap = cap;
        int pc = 64;
        while (pc > 4 && 12 * std::max(g.hist_slots, 1) * pc > 2048) pc >>= 1;
        g.part_cols = pc;
    }
    g.n_reads = (uint32_t)tile->n_reads;
#ifdef BCFGPU_DIAG
    { const char *ab = getenv("BCFGPU_ABLATE"); g.ablate = ab ? atoi(ab) : 0; }
#endif
    g.ref16 = tile->ref16; g.off = tile->plp_off; g.rd = tile->rd; g.epos = tile->epos; g.aux = tile->aux;
    g.fk = c->d_fk; g.beta = c->d_beta; g.lhet = c->d_lhet;
    g.crp = c->d_crp;
    // the callret planes are addressed with ncells of *this* tile
    g.hist = c->d_hist; g.err = c->d_err; g.site_sums = c->d_site_sums;
    g.trunc = reinterpret_cast<unsigned int*>(c->d_err + 1);
    g.deep_list = c->d_deep_list; g.deep_ctr = reinterpret_cast<uint32_t*>(c->d_err + 2); g.deep_keys = c->d_deep_keys;
    g.deep_cap = c->deep_cap; g.deep_key_cap = c->deep_key_cap;
    g.keys = c->d_keys;
    {   // the plan bcfgpu_errmod_plan made for this pass of this tile, if any: it serves this one launch
        const int kind = tile->is_indel ? 1 : 0;
        g.draw_bits = (c->draw.rd[kind] && c->draw.rd[kind] == tile->rd) ? c->draw.bits[kind] : nullptr;
        c->draw.rd[kind] = nullptr;
    }
    g.wide_ctr = reinterpret_cast<uint32_t*>(c->d_err + 5); g.wide_cap = c->wide_cap;
    HIPCHK(hipMemsetAsync(c->d_err + 2, 0, 4 * sizeof(int), c->stream));
#ifdef BCFGPU_DIAG
    {   // phase stamps of glfgen_kernel: totals of the previous launch are printed, then cleared
        static unsigned long long *d_st = nullptr;
        if (!d_st) { hipMalloc(&d_st, 16 * 8); hipMemset(d_st, 0, 16 * 8); }
        if (getenv("BCFGPU_STAMPS")) {
            unsigned long long h[16]; hipStreamSynchronize(c->stream); hipMemcpy(h, d_st, sizeof h, hipMemcpyDeviceToHost);
            unsigned long long tot = 0; for (int i = 0; i < 10; ++i) tot += h[i];
            if (tot) { fprintf(stderr, "[glfgen stamps %%]"); for (int i = 0; i < 10; ++i) fprintf(stderr, " %.1f", 100.0 * h[i] / tot); fprintf(stderr, "  (total %.3g wave-cycles)\n", (double)tot); }
            hipMemset(d_st, 0, 16 * 8);
            g.stamps = d_st;                          // the stamps cost atomics: only when asked for
        }
    }
#endif
    HIPCHK(hipMemsetAsync(c->d_hist, 0, (size_t)tile->n_sites * H_SIZE * sizeof(int), c->stream));
    HIPCHK(hipMemsetAsync(c->d_site_sums, 0, (size_t)tile->n_sites * SITE_NSUM * 8, c->stream));
    hipEvent_t *ev = c->timing ? seq_events(c) : nullptr;
    if (ev) hipEventRecord(ev[0], c->stream);
    launch_glfgen(g, c->stream);
    if (ev) hipEventRecord(ev[1], c->stream);
    CombineParams k{};
    k.n_sites = tile->n_sites; k.n_smpl = S; k.is_indel = tile->is_indel; k.fmt_flag = c->cfg.fmt_flag;
    k.ref16 = tile->ref16; k.cr = c->cr; k.hist = c->d_hist; k.site_sums = c->d_site_sums; k.mw = c->d_mw; k.out = *out;
#ifdef BCFGPU_DIAG
    { const char *ab = getenv("BCFGPU_ABLATE"); k.ablate = ab ? atoi(ab) : 0; }
#endif
    launch_combine(k, c->stream);
    if (ev) hipEventRecord(ev[2], c->stream);
    HIPCHK(hipGetLastError());
    return 0;
}

static void finish_timing(bcfgpu_ctx *c, bool with_call)
{
    if (c->timing != 1) return;
    hipEventSynchronize(with_call ? c->ev[3] : c->ev[2]);
    bcfgpu_timing t{};
    hipEventElapsedTime(&t.glfgen_ms, c->ev[0], c->ev[1]);
    hipEventElapsedTime(&t.combine_ms, c->ev[1], c->ev[2]);
    if (with_call) hipEventElapsedTime(&t.mcall_ms, c->ev[2], c->ev[3]);
    hipEventElapsedTime(&t.total_ms, c->ev[0], with_call ? c->ev[3] : c->ev[2]);
    c->last = t;
}

int bcfgpu_mpileup(bcfgpu_ctx *c, const bcfgpu_tile *tile, const bcfgpu_mplp_out *out)
{
    if (!c || !out || !out->site || !out->pl || !out->dp4) return set_err(BCFGPU_E_ARG, "bcfgpu_mpileup: bad arguments");
    int rc = check_tile(c, tile);
    if (rc) return rc;
    if (tile->n_sites == 0) return 0;
    hipSetDevice(c->cfg.device);
    rc = enqueue_mpileup(c, tile, out);
    if (rc) return rc;
    finish_timing(c, false);
    return 0;
}

// call -G: the groups' frequency sums of a launch, [n_sites][n_grp][5] floats (grow-only; sized from the launch, not from
// cfg.max_sites: a call-only context has max_sites = 0)
static float *grp_q_for(bcfgpu_ctx *c, int n_sites)
{
    if (c->cfg.n_grp <= 1 || n_sites <= 0) return nullptr;
    const size_t bytes = (size_t)n_sites * c->cfg.n_grp * 5 * sizeof(float);
    if (bytes <= c->grp_q_bytes) return c->d_grp_q;
    if (c->d_grp_q) { hipStreamSynchronize(c->stream); hipFree(c->d_grp_q); c->d_grp_q = nullptr; c->grp_q_bytes = 0; }
    void *p = nullptr;
    if (hipMalloc(&p, bytes + bytes / 4) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    c->d_grp_q = (float*)p; c->grp_q_bytes = bytes + bytes / 4;
    return c->d_grp_q;
}

int bcfgpu_mcall(bcfgpu_ctx *c, const bcfgpu_call_in *in, const bcfgpu_call_out *out)
{
    if (!c || !in || !out || !out->site || !out->gt) return set_err(BCFGPU_E_ARG, "bcfgpu_mcall: bad arguments");
    if (in->n_sites == 0) return 0;
    if (!in->nals || !in->unseen || !in->pl || (!in->qs && !(c->cfg.n_grp > 1)))
        return set_err(BCFGPU_E_ARG, "bcfgpu_mcall: NULL input array");
    if (c->cfg.n_grp > 1 && (!in->grp || !in->ad)) return set_err(BCFGPU_E_ARG, "bcfgpu_mcall: -G needs grp and ad");
    if (in->n_gt_max < 1 || in->n_gt_max > BCFGPU_MAX_PL) return set_err(BCFGPU_E_ARG, "bcfgpu_mcall: n_gt_max out of range");
    if (in->ad && (in->n_al_max < 1 || in->n_al_max > BCFGPU_MAX_ALLELES)) return set_err(BCFGPU_E_ARG, "bcfgpu_mcall: n_al_max out of range");
    hipSetDevice(c->cfg.device);
    McallParams m{};
    m.n_sites = in->n_sites; m.n_smpl = c->cfg.n_smpl; m.n_gt_max = in->n_gt_max; m.n_al_max = in->n_al_max;
    m.pl_is_u8 = 0; m.call_flag = c->cfg.call_flag; m.output_tags = c->cfg.output_tags; m.n_grp = c->cfg.n_grp;
    m.theta = c->call_theta_log; m.pl2p = c->d_pl2p;
    m.nals = in->nals; m.unseen = in->unseen; m.msite = nullptr; m.pl = in->pl; m.qs = in->qs; m.ad = in->ad;
    m.grp_rng = c->d_grp_rng; m.grp_q = grp_q_for(c, m.n_sites);
    if (c->cfg.n_grp > 1 && !m.grp_q) return set_err(BCFGPU_E_NOMEM, "call -G: workspace of the groups' frequency sums");
    m.ploidy = in->ploidy; m.grp = c->cfg.n_grp > 1 ? in->grp : nullptr; m.prior_an = in->prior_an; m.prior_ac = in->prior_ac; m.i16 = in->i16;
    m.out = *out; m.out_n_gt_max = in->n_gt_max; m.err = c->d_err;
    if (c->timing == 1) hipEventRecord(c->ev[2], c->stream);
#ifdef BCFGPU_DIAG
    { const char *ab = getenv("BCFGPU_ABLATE"); m.ablate = ab ? atoi(ab) : 0; }
#endif
    launch_mcall(m, c->stream);
    if (c->timing == 1) { hipEventRecord(c->ev[3], c->stream); hipEventSynchronize(c->ev[3]);
        bcfgpu_timing t{}; hipEventElapsedTime(&t.mcall_ms, c->ev[2], c->ev[3]); t.total_ms = t.mcall_ms; c->last = t; }
    HIPCHK(hipGetLastError());
    return 0;
}

int bcfgpu_pipeline(bcfgpu_ctx *c, const bcfgpu_tile *tile, const uint8_t *ploidy, const int32_t *grp,
                    const bcfgpu_mplp_out *mout, const bcfgpu_call_out *cout)
{
    if (!c || !mout || !cout || !mout->site || !mout->pl || !mout->dp4 || !cout->site || !cout->gt)
        return set_err(BCFGPU_E_ARG, "bcfgpu_pipeline: bad arguments");
    int rc = check_tile(c, tile);
    if (rc) return rc;
    if (tile->n_sites == 0) return 0;
    if (c->cfg.n_grp > 1 && (!grp || (c->cfg.grp_tag_is_qs ? !mout->qs : (!mout->adf || !mout->adr))))
        return set_err(BCFGPU_E_ARG, "bcfgpu_pipeline: -G needs grp and the AD (or QS) planes");
    hipSetDevice(c->cfg.device);
    rc = enqueue_mpileup(c, tile, mout);
    if (rc) return rc;
    McallParams m{};
    m.n_sites = tile->n_sites; m.n_smpl = c->cfg.n_smpl; m.n_gt_max = BCFGPU_MAX_PL; m.n_al_max = 5;
    m.pl_is_u8 = 1; m.call_flag = c->cfg.call_flag; m.output_tags = c->cfg.output_tags; m.n_grp = c->cfg.n_grp;
    m.theta = c->call_theta_log; m.pl2p = c->d_pl2p;
    m.msite = mout->site; m.pl = mout->pl; m.qs = nullptr; m.ad = nullptr;
    m.qs_i32 = (c->cfg.n_grp > 1 && c->cfg.grp_tag_is_qs) ? mout->qs : nullptr;
    if (c->cfg.n_grp > 1 && !c->cfg.grp_tag_is_qs) { m.ad_u16 = mout->adf; m.ad_u16b = mout->adr; }   // FORMAT/AD = ADF+ADR (bam2bcf.c:892-896)
    m.grp_rng = c->d_grp_rng; m.grp_q = grp_q_for(c, m.n_sites);
    if (c->cfg.n_grp > 1 && !m.grp_q) return set_err(BCFGPU_E_NOMEM, "call -G: workspace of the groups' frequency sums");
    m.ploidy = ploidy; m.grp = c->cfg.n_grp > 1 ? grp : nullptr;
    m.out = *cout; m.out_n_gt_max = BCFGPU_MAX_PL; m.err = c->d_err;
#ifdef BCFGPU_DIAG
    { const char *ab = getenv("BCFGPU_ABLATE"); m.ablate = ab ? atoi(ab) : 0; }
#endif
    launch_mcall(m, c->stream);
    if (c->timing == 1) hipEventRecord(c->ev[3], c->stream);
    else if (c->timing == 2) { hipEventRecord(c->pool[c->pool_used - 1], c->stream); c->pool_call.back() = 1; }
    HIPCHK(hipGetLastError());
    finish_timing(c, true);
    return 0;
}

bcfgpu_gap_stats *bcfgpu_internal_gap_stats(bcfgpu_ctx *c) { return &c->gap; }
DrawState *bcfgpu_internal_draw_state(bcfgpu_ctx *c) { return &c->draw; }
const bcfgpu_cfg *bcfgpu_internal_cfg(const bcfgpu_ctx *c) { return c ? &c->cfg : nullptr; }
void *bcfgpu_internal_pileup_state(bcfgpu_ctx *c) { return c ? c->pileup_state : nullptr; }
void *bcfgpu_internal_pool_state(bcfgpu_ctx *c) { return c ? c->pool_state : nullptr; }

// workspace `slot` of at least `bytes` (contents undefined); nullptr when the allocation fails
void *bcfgpu_internal_ws(bcfgpu_ctx *c, int slot, size_t bytes)
{
    if (!c || slot < 0 || slot >= 152) return nullptr;
    auto &w = c->ws[slot];
    if (w.bytes >= bytes && w.p) return w.p;
    hipSetDevice(c->cfg.device);
    if (w.p) { hipStreamSynchronize(c->stream); hipFree(w.p); w.p = nullptr; w.bytes = 0; }
    const size_t want = bytes + bytes / 8 + 256;          // a little slack so that slowly growing batches settle
    if (hipMalloc(&w.p, want) != hipSuccess) { w.p = nullptr; return nullptr; }
    w.bytes = want;
    return w.p;
}

// pinned host staging buffer `slot` of at least `bytes` (contents undefined); nullptr when the allocation fails
void *bcfgpu_internal_pinned(bcfgpu_ctx *c, int slot, size_t bytes)
{
    if (!c || slot < 0 || slot >= 8) return nullptr;
    auto &w = c->pinned[slot];
    if (w.bytes >= bytes && w.p) return w.p;
    hipSetDevice(c->cfg.device);
    if (w.p) { hipStreamSynchronize(c->stream); hipHostFree(w.p); w.p = nullptr; w.bytes = 0; }
    const size_t want = bytes + bytes / 8 + 256;
    if (hipHostMalloc(&w.p, want, hipHostMallocDefault) != hipSuccess) { w.p = nullptr; return nullptr; }
    w.bytes = want;
    return w.p;
}

// The realignment and BAQ stages fork their band classes onto side streams meant to run beside one another.  The HIP runtime
// multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default); two streams of one queue run their
// kernels one after the other, and with this context's streams beside the host program's own that happens: the realignment's
// second chain of classes started only when the wide-band stream it shared a queue with had drained (29.0 ms a 16 384-column
// tile; 25.6 ms with eight queues, profiles/r5_hw_queues.txt).  The variable is read when the runtime starts, i.e. at the first
// HIP call of the process: set here, when the library is loaded, unless the user has set it.
__attribute__((constructor)) static void bcfgpu_runtime_knobs() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

// for the stages implemented in their own translation units: bind the device, hand out the stream and shared tables
int bcfgpu_internal_n_cu(const bcfgpu_ctx *c) { return c ? c->n_cu : 256; }

// side streams (fork / join around independent launches); 0 on success
int bcfgpu_internal_side(bcfgpu_ctx *c, hipStream_t **streams, hipEvent_t **events)
{
    if (!c) return -1;
    hipSetDevice(c->cfg.device);
    if (!c->side[0]) {
        for (int i = 0; i < 8; ++i) if (hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking) != hipSuccess) return -1;
        for (int i = 0; i < 9; ++i) if (hipEventCreateWithFlags(&c->side_ev[i], hipEventDisableTiming) != hipSuccess) return -1;
    }
    *streams = c->side; *events = c->side_ev;
    return 0;
}

int bcfgpu_internal_device(bcfgpu_ctx *c, hipStream_t *stream, const float **q2p)
{
    if (!c) return -1;
    hipSetDevice(c->cfg.device);
    if (stream) *stream = c->stream;
    if (q2p) *q2p = c->d_q2p;
    return 0;
}

int bcfgpu_gap_prep_stats(const bcfgpu_ctx *c, bcfgpu_gap_stats *out)
{
    if (!c || !out) return set_err(BCFGPU_E_ARG, "bcfgpu_gap_prep_stats: bad arguments");
    *out = c->gap;
    return 0;
}

}  // extern "C"

// draw.hip -- errmod_cal's random draw for cells of more than 255 usable reads (htslib errmod.c: "then sample 255 bases":
// ks_shuffle(uint16_t, n, bases); n = 255, call site bam2bcf.c:256).
//
// ks_shuffle (htslib ksort.h) walks i = n .. 2, j = (int)(hts_drand48() * i), swap(a[j], a[i-1]); hts_drand48 (hts_os.c /
// os/rand.c) is the 48-bit linear congruential generator of rand48 -- X <- (0x5DEECE66D X + 0xB) mod 2^48, returning X / 2^48
// after the step -- started from 0x1234ABCD330E in every process: ONE generator for the whole run, so the 255 reads a cell
// keeps depend on every deeper-than-255 cell the run visited before it.  mpileup_reg() visits, position by position, the
// samples of the SNP pass and then -- where bcf_call_gap_prep returned >= 0 -- the samples of the indel pass (mpileup.c:343-360).
//
// An LCG can be jumped ahead, so the draw is replayed in parallel: bcfgpu_errmod_plan() ranks the deep cells of a tile's two
// passes in that visit order, gives each the number of draws before it (a deep cell of n reads draws n - 1 numbers), and one
// lane per deep cell jumps the generator to its place, shuffles the cell's usable reads and marks the 255 that stay in a
// bitmap over the tile's reads.  glfgen_kernel, meeting a cell of more than 255 usable reads, keeps the marked ones (or,
// without a plan, the first 255: bcfgpu_truncated_cells counts those).  The generator's position is carried in the context
// from call to call, as the process-wide one is from position to position.
//
// PARITY UNPINNED (like the oracle's restatement, oracle/errmod.c): htslib's source is not in the reference tree and no
// golden of the reference's tests reaches 256 reads in a cell; tests/test_gpu_draw.py compares with the oracle's restatement,
// which tests/test_oracle_errmod_deep.py checks against libc's rand48.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstring>
#include <vector>
#include <algorithm>
#include "kernels.h"

struct bcfgpu_ctx;
int bcfgpu_set_error(int code, const char *what);
extern "C" int bcfgpu_internal_device(bcfgpu_ctx *ctx, hipStream_t *stream, const float **q2p);
extern "C" void *bcfgpu_internal_ws(bcfgpu_ctx *ctx, int slot, size_t bytes);
extern "C" const bcfgpu_cfg *bcfgpu_internal_cfg(const bcfgpu_ctx *ctx);
extern "C" bcfgpu::DrawState *bcfgpu_internal_draw_state(bcfgpu_ctx *ctx);

namespace bcfgpu {

#define RAND48_A 0x5DEECE66DULL
#define RAND48_C 0xBULL
#define RAND48_M 0xFFFFFFFFFFFFULL

// the state after k steps from x: x -> A_k x + C_k (mod 2^48), (A_k, C_k) by doubling
__host__ __device__ inline uint64_t rand48_jump(uint64_t x, uint64_t k)
{
    uint64_t a_acc = 1, c_acc = 0, a = RAND48_A, c = RAND48_C;
    while (k) {
        if (k & 1) { a_acc = (a * a_acc) & RAND48_M; c_acc = (a * c_acc + c) & RAND48_M; }
        c = ((a + 1) * c) & RAND48_M;
        a = (a * a) & RAND48_M;
        k >>= 1;
    }
    return (a_acc * x + c_acc) & RAND48_M;
}

struct DrawEnt { uint64_t key; uint32_t n, beg, end, kind; uint64_t off; };    // one deep cell: visit key, usable reads, its entries [beg, end) of the tile, pass, draws before it

struct DrawScanParams {
    int n_sites, n_smpl, is_indel, min_baseQ;
    const uint32_t *off, *rd;
    const int32_t *cols;            // indel pass: SNP-tile column of every site; NULL for the SNP pass
    const int32_t *ret;             // indel pass: bcf_call_gap_prep's return per site (only 0 is visited); NULL = all
    const uint8_t *visit;           // SNP pass: 0 = mpileup_reg() passes the column over before bcf_call_glfgen (a position outside -t / -T targets, mpileup.c:330-335); NULL = all
    DrawEnt *ent; uint32_t *n_ent; uint32_t cap;
    uint32_t *bits;                 // the pass's bitmap (for the cells of columns without an indel pass)
    unsigned long long *tot_usable;
};

// which entries of a cell reach bases[] (bam2bcf.c:173-203): the same tests as glfgen_kernel's phase A
__device__ __forceinline__ bool draw_usable(uint32_t w, bool indel, uint32_t min_baseQ)
{
    if (w & BCFGPU_RD_SKIP) return false;
    if (indel) return true;
    return !(w & BCFGPU_RD_DEL) && (w & 0xff) >= min_baseQ;
}

// one lane per cell: the cells of more than 255 usable reads are listed with their visit key
__global__ __launch_bounds__(256) void draw_scan_kernel(const DrawScanParams P)
{
    const long cell = (long)blockIdx.x * 256 + threadIdx.x;
    if (cell >= (long)P.n_sites * P.n_smpl) return;
    const uint32_t b = P.off[cell], e = P.off[cell + 1];
    if (e - b <= BCFGPU_MAX_DEPTH) return;
    const int site = (int)(cell / P.n_smpl), s = (int)(cell - (long)site * P.n_smpl);
    uint32_t n = 0;
    for (uint32_t i = b; i < e; ++i) n += draw_usable(P.rd[i], P.is_indel != 0, (uint32_t)P.min_baseQ) ? 1u : 0u;
    if (n <= BCFGPU_MAX_DEPTH) return;
    if ((P.is_indel && P.ret && P.ret[site] != 0) || (!P.is_indel && P.visit && !P.visit[site])) {
        // the pass does not run there (mpileup.c:354; :330-335 for a column outside the targets): no draw is spent on the cell and
        // nothing of it is written out; its first 255 usable reads are marked so that the likelihood kernel has a complete plan and
        // counts nothing as cut
        uint32_t m = 0;
        for (uint32_t i = b; i < e && m < BCFGPU_MAX_DEPTH; ++i)
            if (draw_usable(P.rd[i], P.is_indel != 0, (uint32_t)P.min_baseQ)) { atomicOr(&P.bits[i >> 5], 1u << (i & 31)); ++m; }
        return;
    }
    const uint32_t slot = atomicAdd(P.n_ent, 1u);
    if (slot >= P.cap) return;
    const long col = P.cols ? P.cols[site] : site;
    DrawEnt &d = P.ent[slot];
    d.key = ((unsigned long long)col * 2 + (P.is_indel ? 1 : 0)) * (unsigned long long)P.n_smpl + (unsigned long long)s;
    d.n = n; d.beg = b; d.end = e; d.kind = P.is_indel ? 1u : 0u; d.off = 0;
    atomicAdd(P.tot_usable, (unsigned long long)n);
}

struct DrawShuffleParams {
    int n_ent, min_baseQ;
    const DrawEnt *ent;             // in visit order, `off` = draws before the cell
    const uint32_t *rd[2];          // the two passes' read records
    uint32_t *bits[2];              // their bitmaps (a bit per read of the tile), zeroed
    const unsigned long long *idx_off;   // [n_ent] start of the cell's slice of `idx`
    uint32_t *idx;                  // scratch: positions of the usable reads
    unsigned long long x0;          // the generator before the first draw of this plan
};

// one lane per deep cell: ks_shuffle over the cell's usable reads, the first 255 of the shuffled order marked
__global__ __launch_bounds__(64) void draw_shuffle_kernel(const DrawShuffleParams P)
{
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= P.n_ent) return;
    const DrawEnt d = P.ent[k];
    const uint32_t *rd = P.rd[d.kind];
    uint32_t *a = P.idx + P.idx_off[k];
    uint32_t n = 0;
    for (uint32_t i = d.beg; i < d.end; ++i) if (draw_usable(rd[i], d.kind != 0, (uint32_t)P.min_baseQ)) a[n++] = i;
    unsigned long long x = rand48_jump(P.x0, d.off);
    for (uint32_t i = n; i > 1; --i) {
        x = (RAND48_A * x + RAND48_C) & RAND48_M;
        // hts_drand48: ldexp(x0, -48) + ldexp(x1, -32) + ldexp(x2, -16) of the state's three 16-bit words = x / 2^48, exactly
        const double r = (double)x * (1.0 / 281474976710656.0);
        const uint32_t j = (uint32_t)(int)(r * (double)i);
        const uint32_t t = a[j]; a[j] = a[i - 1]; a[i - 1] = t;
    }
    uint32_t *bits = P.bits[d.kind];
    for (uint32_t i = 0; i < BCFGPU_MAX_DEPTH; ++i) atomicOr(&bits[a[i] >> 5], 1u << (a[i] & 31));
}

}  // namespace bcfgpu

using namespace bcfgpu;

#define DR_CHK(call) do { if ((call) != hipSuccess) return bcfgpu_set_error(BCFGPU_E_HIP, #call); } while (0)

extern "C" int bcfgpu_errmod_plan_visit(bcfgpu_ctx *ctx, const bcfgpu_tile *snp, const uint8_t *snp_visit, const bcfgpu_tile *indel, const int32_t *indel_cols,
                                        const int32_t *indel_ret);
extern "C" int bcfgpu_errmod_plan(bcfgpu_ctx *ctx, const bcfgpu_tile *snp, const bcfgpu_tile *indel, const int32_t *indel_cols, const int32_t *indel_ret)
{
    return bcfgpu_errmod_plan_visit(ctx, snp, nullptr, indel, indel_cols, indel_ret);
}
extern "C" int bcfgpu_errmod_plan_visit(bcfgpu_ctx *ctx, const bcfgpu_tile *snp, const uint8_t *snp_visit, const bcfgpu_tile *indel, const int32_t *indel_cols,
                                        const int32_t *indel_ret)
{
    if (!ctx || (!snp && !indel)) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_errmod_plan: bad arguments");
    if (indel && indel->n_sites && !indel_cols) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_errmod_plan: the indel tile's columns are needed");
    hipStream_t stream = nullptr;
    if (bcfgpu_internal_device(ctx, &stream, nullptr)) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_errmod_plan: bad context");
    const bcfgpu_cfg *cfg = bcfgpu_internal_cfg(ctx);
    DrawState &D = *bcfgpu_internal_draw_state(ctx);
    const int S = cfg->n_smpl;
    D.rd[0] = D.rd[1] = nullptr; D.bits[0] = D.bits[1] = nullptr; D.n_planned = 0;
    const bcfgpu_tile *tiles[2] = { snp, indel };
    uint64_t n_reads[2] = { snp ? snp->n_reads : 0, indel ? indel->n_reads : 0 };
    // the bitmaps (a bit per read of each tile), the list, its counters
    const uint32_t cap = (uint32_t)std::min<uint64_t>((n_reads[0] + n_reads[1]) / (BCFGPU_MAX_DEPTH + 1) + 1, 1u << 24);
    uint32_t *bits[2] = { nullptr, nullptr };
    for (int t = 0; t < 2; ++t) {
        if (!tiles[t] || !n_reads[t]) continue;
        const size_t words = (size_t)((n_reads[t] + 31) / 32);
        bits[t] = (uint32_t*)bcfgpu_internal_ws(ctx, 136 + t, words * 4 + 64);
        if (!bits[t]) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_errmod_plan: device workspace");
        DR_CHK(hipMemsetAsync(bits[t], 0, words * 4, stream));
    }
    DrawEnt *d_ent = (DrawEnt*)bcfgpu_internal_ws(ctx, 138, (size_t)cap * sizeof(DrawEnt) + 64);
    unsigned long long *d_ctr = (unsigned long long*)bcfgpu_internal_ws(ctx, 139, 64);
    if (!d_ent || !d_ctr) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_errmod_plan: device workspace");
    DR_CHK(hipMemsetAsync(d_ctr, 0, 64, stream));
    int32_t *d_cols = nullptr, *d_ret = nullptr;
    if (indel && indel->n_sites) {
        d_cols = (int32_t*)bcfgpu_internal_ws(ctx, 140, (size_t)indel->n_sites * 8 + 64);
        if (!d_cols) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_errmod_plan: device workspace");
        DR_CHK(hipMemcpyAsync(d_cols, indel_cols, (size_t)indel->n_sites * 4, hipMemcpyHostToDevice, stream));
        if (indel_ret) { d_ret = d_cols + indel->n_sites; DR_CHK(hipMemcpyAsync(d_ret, indel_ret, (size_t)indel->n_sites * 4, hipMemcpyHostToDevice, stream)); }
    }
    uint8_t *d_visit = nullptr;
    if (snp && snp_visit && snp->n_sites) {
        d_visit = (uint8_t*)bcfgpu_internal_ws(ctx, 132, (size_t)snp->n_sites + 64);
        if (!d_visit) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_errmod_plan: device workspace");
        DR_CHK(hipMemcpyAsync(d_visit, snp_visit, (size_t)snp->n_sites, hipMemcpyHostToDevice, stream));
    }
    for (int t = 0; t < 2; ++t) {
        if (!tiles[t] || !tiles[t]->n_sites || !n_reads[t]) continue;
        DrawScanParams Q{};
        Q.n_sites = tiles[t]->n_sites; Q.n_smpl = S; Q.is_indel = t; Q.min_baseQ = cfg->min_baseQ < 0 ? 0 : cfg->min_baseQ;
        Q.off = tiles[t]->plp_off; Q.rd = tiles[t]->rd; Q.cols = t ? d_cols : nullptr; Q.ret = t ? d_ret : nullptr; Q.visit = t ? nullptr : d_visit;
        Q.bits = bits[t]; Q.ent = d_ent; Q.n_ent = reinterpret_cast<uint32_t*>(d_ctr); Q.cap = cap; Q.tot_usable = d_ctr + 1;
        const long ncells = (long)Q.n_sites * S;
        hipLaunchKernelGGL(draw_scan_kernel, dim3((unsigned)((ncells + 255) / 256)), dim3(256), 0, stream, Q);
    }
    unsigned long long ctr[2] = {0, 0};
    DR_CHK(hipMemcpyAsync(ctr, d_ctr, 16, hipMemcpyDeviceToHost, stream));
    DR_CHK(hipStreamSynchronize(stream));
    const uint32_t n_ent = (uint32_t)std::min<unsigned long long>(ctr[0] & 0xffffffffull, cap);
    for (int t = 0; t < 2; ++t) if (tiles[t]) { D.rd[t] = tiles[t]->rd; D.bits[t] = bits[t]; }
    if (n_ent == 0) return BCFGPU_OK;
    // visit order and the draws before every cell: the list is short (a cell in it holds more than 255 reads), ranked on the host
    std::vector<DrawEnt> ent(n_ent);
    DR_CHK(hipMemcpy(ent.data(), d_ent, (size_t)n_ent * sizeof(DrawEnt), hipMemcpyDeviceToHost));
    std::sort(ent.begin(), ent.end(), [](const DrawEnt &a, const DrawEnt &b) { return a.key < b.key; });
    std::vector<unsigned long long> idx_off(n_ent);
    unsigned long long draws = 0, usable = 0;
    for (uint32_t k = 0; k < n_ent; ++k) { ent[k].off = draws; draws += ent[k].n - 1; idx_off[k] = usable; usable += ent[k].n; }
    unsigned long long *d_ioff = (unsigned long long*)bcfgpu_internal_ws(ctx, 141, (size_t)n_ent * 8 + 64);
    uint32_t *d_idx = (uint32_t*)bcfgpu_internal_ws(ctx, 142, (size_t)usable * 4 + 64);
    if (!d_ioff || !d_idx) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_errmod_plan: device workspace");
    DR_CHK(hipMemcpyAsync(d_ent, ent.data(), (size_t)n_ent * sizeof(DrawEnt), hipMemcpyHostToDevice, stream));
    DR_CHK(hipMemcpyAsync(d_ioff, idx_off.data(), (size_t)n_ent * 8, hipMemcpyHostToDevice, stream));
    DrawShuffleParams H{};
    H.n_ent = (int)n_ent; H.min_baseQ = cfg->min_baseQ < 0 ? 0 : cfg->min_baseQ; H.ent = d_ent;
    H.rd[0] = snp ? snp->rd : nullptr; H.rd[1] = indel ? indel->rd : nullptr; H.bits[0] = bits[0]; H.bits[1] = bits[1];
    H.idx_off = d_ioff; H.idx = d_idx; H.x0 = D.x;
    hipLaunchKernelGGL(draw_shuffle_kernel, dim3((n_ent + 63) / 64), dim3(64), 0, stream, H);
    DR_CHK(hipGetLastError());
    DR_CHK(hipStreamSynchronize(stream));                     // (ent / idx_off are this call's host vectors)
    D.x = rand48_jump(D.x, draws);
    D.n_planned = n_ent;
    return BCFGPU_OK;
}

extern "C" int bcfgpu_errmod_seed(bcfgpu_ctx *ctx, uint64_t state)
{
    if (!ctx) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_errmod_seed: bad arguments");
    DrawState &D = *bcfgpu_internal_draw_state(ctx);
    D.x = state & RAND48_M;
    return BCFGPU_OK;
}
extern "C" uint64_t bcfgpu_errmod_state(bcfgpu_ctx *ctx) { return ctx ? bcfgpu_internal_draw_state(ctx)->x : 0; }

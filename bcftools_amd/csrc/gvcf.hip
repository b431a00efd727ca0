// gvcf.hip -- the gVCF blocks of `bcftools mpileup --gvcf` (gvcf_write, gvcf.c:88-226) over the records of a tile.
//
// The reference merges reference-only records one by one into an open block.  Whether a record opens a new block depends
// only on itself and the record before it (its depth range, sequence, position), so block starts are a site-local rule,
// block numbers a prefix sum over the starts, and the block's per-sample values a reduction over its sites:
//   gvcf_site_kernel   one wavefront per site: smallest FORMAT/DP over the samples -> range (gvcf.c:106-128)
//   gvcf_head_kernel   one lane per site: does the site open a block (gvcf.c:130-131)
//   (hipcub inclusive sum of the heads = block number + 1)
//   gvcf_block_kernel  one lane per site: block table (first/last site, start, END)
//   gvcf_reduce_kernel one lane per (block, sample): min DP, smallest (PL[1], PL[2]) pair, PL[0] of the first record
//                      (gvcf.c:171-213); coalesced over samples.
// Everything is integer: bit-exact against the oracle's sequential restatement (oracle/gvcf.c).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include "kernels.h"

extern "C" int bcfgpu_internal_device(bcfgpu_ctx *ctx, hipStream_t *stream, const float **q2p);
extern "C" void *bcfgpu_internal_ws(bcfgpu_ctx *ctx, int slot, size_t bytes);
extern "C" const bcfgpu_cfg *bcfgpu_internal_cfg(const bcfgpu_ctx *c);
int bcfgpu_set_error(int code, const char *what);

namespace bcfgpu {

#define GVCF_MAX_RANGE 16

struct GvcfParams {
    int n_sites, n_smpl, n_range;
    int dp_range[GVCF_MAX_RANGE];
    const int32_t *pos, *rid;
    const uint8_t *brk;
    const bcfgpu_site *site;
    const uint8_t *pl; const uint16_t *dp4;
    const uint8_t *ref_only; const int32_t *dp32, *end;      // the `call -g` form (bcfgpu.h)
    int32_t *range, *head, *scan;       // workspace
    int32_t *blk, *min_dp;
    bcfgpu_gvcf_block *block;
    int32_t *dp_out;
    uint8_t *pl_out;
};

__global__ __launch_bounds__(256) void gvcf_site_kernel(const GvcfParams P)
{
    const int site = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (site >= P.n_sites) return;
    const int S = P.n_smpl;
    int m = INT32_MAX;
    if (P.dp32) {
        const int32_t *d = P.dp32 + (size_t)site * S;
        for (int s = lane; s < S; s += 64) m = min(m, d[s]);
    } else {
        const uint16_t *d = P.dp4 + (size_t)site * 4 * S;
        for (int s = lane; s < S; s += 64) m = min(m, (int)d[s] + d[S + s] + d[2 * S + s] + d[3 * S + s]);
    }
    for (int o = 32; o; o >>= 1) m = min(m, __shfl_xor(m, o, 64));
    if (lane) return;
    // REF and <*> only (mpileup.c:309-315); `call -g`: the records mcall() left with the reference allele alone
    const bool is_ref = (P.ref_only ? P.ref_only[site] != 0 : P.site[site].n_alleles == 2 && P.site[site].unseen == 1) && !(P.brk && (P.brk[site] & 2));
    int r = 0;
    if (is_ref) { while (r < P.n_range && m >= P.dp_range[r]) ++r; }
    P.min_dp[site] = is_ref ? m : 0;
    P.range[site] = r;
}

// may site i continue the block site i-1 is in?
__device__ __forceinline__ bool gvcf_joins(const GvcfParams &P, int i)
{
    if (i <= 0) return false;
    const int r = P.range[i];
    if (!r || P.range[i - 1] != r) return false;
    if (P.rid && P.rid[i] != P.rid[i - 1]) return false;
    if (P.pos[i] > (P.end ? P.end[i - 1] : P.pos[i - 1]) + 1) return false;
    return !(P.brk && (P.brk[i - 1] & 1));
}

__global__ __launch_bounds__(256) void gvcf_head_kernel(const GvcfParams P)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.n_sites) return;
    P.head[i] = P.range[i] && !gvcf_joins(P, i);
}

__global__ __launch_bounds__(256) void gvcf_block_kernel(const GvcfParams P)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.n_sites) return;
    if (!P.range[i]) { P.blk[i] = -1; return; }
    const int b = P.scan[i] - 1;
    P.blk[i] = b;
    bcfgpu_gvcf_block *B = &P.block[b];
    if (P.head[i]) { B->first_site = i; B->start_pos = P.pos[i]; B->range = P.range[i]; }
    if (i + 1 == P.n_sites || !gvcf_joins(P, i + 1)) {
        B->last_site = i;
        // gvcf.c:139-141: a record at the block's last position (the indel record after the SNP record: flagged in brk when it
        // is not in the list, the next record when it is) cuts the block one short
        const int e0 = P.end ? P.end[i] : P.pos[i];
        const bool cut = (P.brk && (P.brk[i] & 1)) ||
                         (i + 1 < P.n_sites && !(P.brk && (P.brk[i + 1] & 2)) && (!P.rid || P.rid[i + 1] == P.rid[i]) && P.pos[i + 1] == e0);
        B->end1 = e0 + 1 - (cut ? 1 : 0);
    }
}

__global__ __launch_bounds__(64) void gvcf_reduce_kernel(const GvcfParams P, int chunks)
{
    const int b = blockIdx.x / chunks, s = (blockIdx.x % chunks) * 64 + threadIdx.x;
    const int S = P.n_smpl;
    const int first = P.block[b].first_site, last = P.block[b].last_site;
    if (blockIdx.x % chunks == 0 && threadIdx.x == 0) {
        int m = P.min_dp[first];
        for (int i = first + 1; i <= last; ++i) m = min(m, P.min_dp[i]);
        P.block[b].min_dp = m;
    }
    if (s >= S) return;
    if (P.dp32) {                                              // `call -g`: FORMAT/DP only
        int dp = P.dp32[(size_t)first * S + s];
        for (int i = first + 1; i <= last; ++i) dp = min(dp, P.dp32[(size_t)i * S + s]);
        P.dp_out[(size_t)b * S + s] = dp;
        return;
    }
    const uint16_t *d = P.dp4 + (size_t)first * 4 * S + s; const uint8_t *p = P.pl + (size_t)first * BCFGPU_MAX_PL * S + s;
    int dp = (int)d[0] + d[S] + d[2 * S] + d[3 * S];
    const int pl0 = p[0];
    int pl1 = p[S], pl2 = p[2 * S];
    for (int i = first + 1; i <= last; ++i) {
        d += (size_t)4 * S; p += (size_t)BCFGPU_MAX_PL * S;
        dp = min(dp, (int)d[0] + d[S] + d[2 * S] + d[3 * S]);
        const int a = p[S], c = p[2 * S];
        if (a < pl1 || (a == pl1 && c < pl2)) { pl1 = a; pl2 = c; }
    }
    P.dp_out[(size_t)b * S + s] = dp;
    uint8_t *o = P.pl_out + (size_t)b * 3 * S + s;
    o[0] = (uint8_t)pl0; o[S] = (uint8_t)pl1; o[2 * S] = (uint8_t)pl2;
}

}  // namespace bcfgpu

using namespace bcfgpu;

#define GV_CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return bcfgpu_set_error(BCFGPU_E_HIP, hipGetErrorString(e_)); } while (0)

extern "C" int bcfgpu_gvcf_blocks(bcfgpu_ctx *ctx, const bcfgpu_gvcf_in *in, const bcfgpu_gvcf_out *out, int32_t *n_blocks)
{
    if (!ctx || !in || !out || !n_blocks) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_gvcf_blocks: null argument");
    if (in->n_sites < 0 || in->n_range < 1 || in->n_range > GVCF_MAX_RANGE || !in->dp_range)
        return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_gvcf_blocks: n_sites < 0 or 1..16 depth ranges expected");
    *n_blocks = 0;
    if (!in->n_sites) return BCFGPU_OK;
    const bool call_form = in->dp != nullptr || in->ref_only != nullptr;
    if (!in->pos || !out->blk || !out->min_dp || !out->block || !out->dp ||
        (call_form ? (!in->dp || !in->ref_only) : (!in->site || !in->pl || !in->dp4 || !out->pl)))
        return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_gvcf_blocks: null plane");
    hipStream_t stream;
    bcfgpu_internal_device(ctx, &stream, nullptr);
    const bcfgpu_cfg *cfg = bcfgpu_internal_cfg(ctx);
    const int n = in->n_sites;
    GvcfParams P;
    P.n_sites = n; P.n_smpl = cfg->n_smpl; P.n_range = in->n_range;
    for (int i = 0; i < GVCF_MAX_RANGE; ++i) P.dp_range[i] = i < in->n_range ? in->dp_range[i] : INT32_MAX;
    P.pos = in->pos; P.rid = in->rid; P.brk = in->brk; P.site = in->site; P.pl = in->pl; P.dp4 = in->dp4;
    P.ref_only = in->ref_only; P.dp32 = in->dp; P.end = in->end;
    P.blk = out->blk; P.min_dp = out->min_dp; P.block = out->block; P.dp_out = out->dp; P.pl_out = out->pl;
    size_t tmp_bytes = 0;
    GV_CHK(hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, (int32_t*)nullptr, (int32_t*)nullptr, n, stream));
    int32_t *w = (int32_t*)bcfgpu_internal_ws(ctx, 32, (size_t)n * 3 * sizeof(int32_t));
    void *d_tmp = bcfgpu_internal_ws(ctx, 33, tmp_bytes + 16);
    if (!w || !d_tmp) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_gvcf_blocks: workspace");
    P.range = w; P.head = w + n; P.scan = w + 2 * (size_t)n;
    hipLaunchKernelGGL(gvcf_site_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, P);
    hipLaunchKernelGGL(gvcf_head_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, P);
    GV_CHK(hipcub::DeviceScan::InclusiveSum(d_tmp, tmp_bytes, P.head, P.scan, n, stream));
    hipLaunchKernelGGL(gvcf_block_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, P);
    int32_t nb = 0;
    GV_CHK(hipMemcpyAsync(&nb, P.scan + (n - 1), sizeof nb, hipMemcpyDeviceToHost, stream));
    GV_CHK(hipStreamSynchronize(stream));
    if (nb < 0 || nb > n) return bcfgpu_set_error(BCFGPU_E_HIP, "bcfgpu_gvcf_blocks: block count out of range");
    if (nb) {
        const int chunks = (P.n_smpl + 63) / 64;
        hipLaunchKernelGGL(gvcf_reduce_kernel, dim3((unsigned)nb * chunks), dim3(64), 0, stream, P, chunks);
    }
    GV_CHK(hipGetLastError());
    *n_blocks = nb;
    return BCFGPU_OK;
}

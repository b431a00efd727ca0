// overlap.hip -- the mate-overlap quality tweak of the pileup engine, for all read pairs of a pool at once.
//
// `bcftools mpileup` switches it on with bam_mplp_init_overlaps() (mpileup.c:640); htslib's pileup then passes every
// second mate that overlaps its first mate through tweak_overlap_quality() (sam.c): at each reference position both
// reads cover with an aligned base, equal bases pool their qualities in the first read (capped at 200) and different
// bases keep 0.8 of the better quality; the other read's base drops to quality 0.  Which reads form a pair (proper pair,
// same contig, the first mate still buffered) is the pileup engine's bookkeeping and stays with the caller; this is the
// per-base arithmetic, one lane per pair: two cursors walk the CIGARs to their common aligned columns.
#include <hip/hip_runtime.h>
#include <cstring>
#include <vector>
#include "kernels.h"

extern "C" int bcfgpu_internal_device(bcfgpu_ctx *ctx, hipStream_t *stream, const float **q2p);
extern "C" void *bcfgpu_internal_ws(bcfgpu_ctx *ctx, int slot, size_t bytes);
int bcfgpu_set_error(int code, const char *what);

namespace bcfgpu {

struct OverlapParams {
    int n_pairs;
    const int32_t *pair_a, *pair_b;
    const int32_t *r_pos, *r_ncig, *r_cig_off, *r_seq_off;
    const uint32_t *cig;
    const uint8_t *seq16;
    uint8_t *qual;                  // in/out
};

__global__ __launch_bounds__(64) void overlap_kernel(const OverlapParams P)
{
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= P.n_pairs) return;
    const int ra = P.pair_a[p], rb = P.pair_b[p];
    const uint32_t *ca = P.cig + P.r_cig_off[ra], *cb = P.cig + P.r_cig_off[rb];
    const int na = P.r_ncig[ra], nb = P.r_ncig[rb];
    const uint8_t *sa = P.seq16 + P.r_seq_off[ra], *sb = P.seq16 + P.r_seq_off[rb];
    uint8_t *qa = P.qual + P.r_seq_off[ra], *qb = P.qual + P.r_seq_off[rb];
    // the current aligned (M/=/X) block of each read: reference start x, query start y, columns left l
    int ka = 0, kb = 0, xa = P.r_pos[ra], ya = 0, la = 0, xb = P.r_pos[rb], yb = 0, lb = 0;
    for (;;) {
        while (la == 0 && ka < na) {
            const uint32_t c = ca[ka++];
            const int op = c & 0xf, l = (int)(c >> 4);
            if (op == 0 || op == 7 || op == 8) la = l;
            else if (op == 2 || op == 3) xa += l;
            else if (op == 1 || op == 4) ya += l;
        }
        while (lb == 0 && kb < nb) {
            const uint32_t c = cb[kb++];
            const int op = c & 0xf, l = (int)(c >> 4);
            if (op == 0 || op == 7 || op == 8) lb = l;
            else if (op == 2 || op == 3) xb += l;
            else if (op == 1 || op == 4) yb += l;
        }
        if (la == 0 || lb == 0) break;
        if (xa < xb) { const int d = min(xb - xa, la); xa += d; ya += d; la -= d; continue; }
        if (xb < xa) { const int d = min(xa - xb, lb); xb += d; yb += d; lb -= d; continue; }
        const int m = min(la, lb);
        for (int i = 0; i < m; ++i) {
            const int va = qa[ya + i], vb = qb[yb + i];
            int oa, ob;
            if (sa[ya + i] == sb[yb + i]) { oa = min(200, va + vb); ob = 0; }
            else if (va >= vb) { oa = (int)(0.8 * va); ob = 0; }
            else { ob = (int)(0.8 * vb); oa = 0; }
            qa[ya + i] = (uint8_t)oa; qb[yb + i] = (uint8_t)ob;
        }
        xa += m; ya += m; la -= m; xb += m; yb += m; lb -= m;
    }
}

}  // namespace bcfgpu

using namespace bcfgpu;

extern "C" int bcfgpu_overlap_tweak(bcfgpu_ctx *ctx, const bcfgpu_reads *rd, int32_t n_pairs, const int32_t *pair_a,
                                    const int32_t *pair_b, uint8_t *qual_out)
{
    if (!ctx || !rd || n_pairs < 0 || (n_pairs && (!pair_a || !pair_b)) || !qual_out || rd->n_reads < 0)
        return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_overlap_tweak: bad arguments");
    hipStream_t stream = nullptr;
    if (bcfgpu_internal_device(ctx, &stream, nullptr)) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_overlap_tweak: bad context");
    const int n = rd->n_reads;
    size_t nbase = 0, ncig = 0;
    for (int r = 0; r < n; ++r) {
        const size_t e = (size_t)rd->r_seq_off[r] + rd->r_lq[r], c = (size_t)rd->r_cig_off[r] + rd->r_ncig[r];
        if (e > nbase) nbase = e;
        if (c > ncig) ncig = c;
    }
    // every read may be in one pair only (a second pair would race with the first on the read's qualities)
    {
        std::vector<uint8_t> seen((size_t)n, 0);
        for (int p = 0; p < n_pairs; ++p) {
            const int a = pair_a[p], b = pair_b[p];
            if (a < 0 || a >= n || b < 0 || b >= n || a == b || seen[a] || seen[b])
                return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_overlap_tweak: a read index is out of range or used twice");
            seen[a] = seen[b] = 1;
        }
    }
    if (n_pairs == 0 || nbase == 0) { if (nbase) std::memcpy(qual_out, rd->qual, nbase); return BCFGPU_OK; }
    // context workspaces (shared with the BAQ stage's slots; both finish their stream work before returning)
    void *d_pa = bcfgpu_internal_ws(ctx, 7, (size_t)n_pairs * 4), *d_pb = bcfgpu_internal_ws(ctx, 8, (size_t)n_pairs * 4),
         *d_pos = bcfgpu_internal_ws(ctx, 9, (size_t)n * 4), *d_ncig = bcfgpu_internal_ws(ctx, 10, (size_t)n * 4),
         *d_coff = bcfgpu_internal_ws(ctx, 11, (size_t)n * 4), *d_soff = bcfgpu_internal_ws(ctx, 12, (size_t)n * 4),
         *d_cig = bcfgpu_internal_ws(ctx, 13, (ncig + 4) * 4), *d_seq = bcfgpu_internal_ws(ctx, 14, nbase + 16),
         *d_qual = bcfgpu_internal_ws(ctx, 15, nbase + 16);
    if (!d_pa || !d_pb || !d_pos || !d_ncig || !d_coff || !d_soff || !d_cig || !d_seq || !d_qual)
        return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_overlap_tweak: device workspace");
    #define OV_CHK(call) do { if ((call) != hipSuccess) return bcfgpu_set_error(BCFGPU_E_HIP, #call); } while (0)
    OV_CHK(hipMemcpyAsync(d_pa, pair_a, (size_t)n_pairs * 4, hipMemcpyHostToDevice, stream));
    OV_CHK(hipMemcpyAsync(d_pb, pair_b, (size_t)n_pairs * 4, hipMemcpyHostToDevice, stream));
    OV_CHK(hipMemcpyAsync(d_pos, rd->r_pos, (size_t)n * 4, hipMemcpyHostToDevice, stream));
    OV_CHK(hipMemcpyAsync(d_ncig, rd->r_ncig, (size_t)n * 4, hipMemcpyHostToDevice, stream));
    OV_CHK(hipMemcpyAsync(d_coff, rd->r_cig_off, (size_t)n * 4, hipMemcpyHostToDevice, stream));
    OV_CHK(hipMemcpyAsync(d_soff, rd->r_seq_off, (size_t)n * 4, hipMemcpyHostToDevice, stream));
    OV_CHK(hipMemcpyAsync(d_cig, rd->cig, ncig * 4, hipMemcpyHostToDevice, stream));
    OV_CHK(hipMemcpyAsync(d_seq, rd->seq16, nbase, hipMemcpyHostToDevice, stream));
    OV_CHK(hipMemcpyAsync(d_qual, rd->qual, nbase, hipMemcpyHostToDevice, stream));
    OverlapParams P{};
    P.n_pairs = n_pairs; P.pair_a = (const int32_t*)d_pa; P.pair_b = (const int32_t*)d_pb;
    P.r_pos = (const int32_t*)d_pos; P.r_ncig = (const int32_t*)d_ncig; P.r_cig_off = (const int32_t*)d_coff; P.r_seq_off = (const int32_t*)d_soff;
    P.cig = (const uint32_t*)d_cig; P.seq16 = (const uint8_t*)d_seq; P.qual = (uint8_t*)d_qual;
    hipLaunchKernelGGL(overlap_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, stream, P);
    OV_CHK(hipGetLastError());
    OV_CHK(hipMemcpyAsync(qual_out, d_qual, nbase, hipMemcpyDeviceToHost, stream));
    OV_CHK(hipStreamSynchronize(stream));
    #undef OV_CHK
    return BCFGPU_OK;
}

extern "C" void *bcfgpu_internal_pool_state(bcfgpu_ctx *ctx);

// The same tweak on the pool bcfgpu_pool_upload left in HBM (after bcfgpu_pool_baq): the pairs go up, nothing comes back.
extern "C" int bcfgpu_pool_overlap_tweak(bcfgpu_ctx *ctx, int32_t n_pairs, const int32_t *pair_a, const int32_t *pair_b)
{
    if (!ctx || n_pairs < 0 || (n_pairs && (!pair_a || !pair_b))) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pool_overlap_tweak: bad arguments");
    hipStream_t stream = nullptr;
    if (bcfgpu_internal_device(ctx, &stream, nullptr)) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pool_overlap_tweak: bad context");
    const DevPool &D = *static_cast<const DevPool*>(bcfgpu_internal_pool_state(ctx));
    if (!D.valid) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pool_overlap_tweak: no read pool on this context (bcfgpu_pool_upload)");
    const int n = D.n_reads;
    {   // every read may be in one pair only (a second pair would race with the first on the read's qualities)
        std::vector<uint8_t> seen((size_t)n, 0);
        for (int p = 0; p < n_pairs; ++p) {
            const int a = pair_a[p], b = pair_b[p];
            if (a < 0 || a >= n || b < 0 || b >= n || a == b || seen[a] || seen[b])
                return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pool_overlap_tweak: a read index is out of range or used twice");
            seen[a] = seen[b] = 1;
        }
    }
    if (n_pairs == 0 || D.n_bases == 0) return BCFGPU_OK;
    void *d_pa = bcfgpu_internal_ws(ctx, 123, (size_t)n_pairs * 4 + 64), *d_pb = bcfgpu_internal_ws(ctx, 124, (size_t)n_pairs * 4 + 64);
    if (!d_pa || !d_pb) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_pool_overlap_tweak: device workspace");
    if (hipMemcpyAsync(d_pa, pair_a, (size_t)n_pairs * 4, hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipMemcpyAsync(d_pb, pair_b, (size_t)n_pairs * 4, hipMemcpyHostToDevice, stream) != hipSuccess)
        return bcfgpu_set_error(BCFGPU_E_HIP, "bcfgpu_pool_overlap_tweak: upload");
    OverlapParams P{};
    P.n_pairs = n_pairs; P.pair_a = (const int32_t*)d_pa; P.pair_b = (const int32_t*)d_pb;
    P.r_pos = D.r_pos; P.r_ncig = D.r_ncig; P.r_cig_off = D.r_cig_off; P.r_seq_off = D.r_seq_off;
    P.cig = D.cig; P.seq16 = D.seq16; P.qual = D.qual;
    hipLaunchKernelGGL(overlap_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, stream, P);
    if (hipGetLastError() != hipSuccess) return bcfgpu_set_error(BCFGPU_E_HIP, "bcfgpu_pool_overlap_tweak: launch");
    if (hipStreamSynchronize(stream) != hipSuccess) return bcfgpu_set_error(BCFGPU_E_HIP, "bcfgpu_pool_overlap_tweak");      // (the pair arrays are the caller's)
    return BCFGPU_OK;
}


// capmapq.hip -- sam_cap_mapq(b, ref, ref_len, thres) of htslib sam.c as `mpileup -C INT` applies it to every read after
// BAQ (mpileup.c:235-239): from the mismatches of the read against the reference -- their number, the sum of their base
// qualities (each capped at 33), the aligned length and the clipped bases -- a cap on the read's mapping quality, or -1:
// the read is dropped.  One lane per read: a walk over the CIGAR and the aligned bases (integer counts), then
//     t = q - 4.343 ln( prod_{i<mm} len/(i+1) ) + clip_q/5;   t > thres: -1;   cap = (int)( sqrt((thres - max(t,0))/thres) thres + .499 )
// in double, in the reference's operation order.
//
// htslib is not part of the reference tree and no golden of the reference's tests runs `mpileup -C`: the function is
// restated from htslib's published source and checked against the oracle's restatement only (parity unpinned).
#include <hip/hip_runtime.h>
#include <cstring>
#include <climits>
#include "kernels.h"

extern "C" int bcfgpu_internal_device(bcfgpu_ctx *ctx, hipStream_t *stream, const float **q2p);
extern "C" void *bcfgpu_internal_ws(bcfgpu_ctx *ctx, int slot, size_t bytes);
int bcfgpu_set_error(int code, const char *what);

namespace bcfgpu {

struct CapParams {
    int n_reads, thres;
    const int32_t *r_pos, *r_lq, *r_ncig, *r_cig_off, *r_seq_off;
    const uint32_t *cig;
    const uint8_t *seq16, *qual;
    const char *ref; long ref_lo, ref_hi;       // the slice of the contig the reads can touch; outside it: past the end
    int32_t *out;
};

// seq_nt16_table of htslib for the letters a reference holds (IUPAC, case-insensitive; anything else is N = 15)
__device__ __forceinline__ int cap_nt16_of(char c)
{
    switch (c) {
        case 'A': case 'a': return 1;  case 'C': case 'c': return 2;  case 'G': case 'g': return 4;  case 'T': case 't': return 8;
        case '=': return 0;
        case 'M': case 'm': return 3;  case 'R': case 'r': return 5;  case 'S': case 's': return 6;  case 'V': case 'v': return 7;
        case 'W': case 'w': return 9;  case 'Y': case 'y': return 10; case 'H': case 'h': return 11; case 'K': case 'k': return 12;
        case 'D': case 'd': return 13; case 'B': case 'b': return 14;
        case 'U': case 'u': return 8;  case '0': return 1; case '1': return 2; case '2': return 4; case '3': return 8;   /* the rest of htslib's seq_nt16_table */
        default: return 15;
    }
}

__global__ __launch_bounds__(256) void cap_mapq_kernel(const CapParams P)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= P.n_reads) return;
    const uint32_t *cigar = P.cig + P.r_cig_off[r];
    const uint8_t *seq = P.seq16 + P.r_seq_off[r], *qual = P.qual + P.r_seq_off[r];
    const int ncig = P.r_ncig[r];
    const int thres = P.thres < 0 ? 40 : P.thres;
    int mm = 0, q = 0, len = 0, clip_q = 0, y = 0;
    long x = P.r_pos[r];
    auto past_end = [&](long p) { return p >= P.ref_hi || p < P.ref_lo || P.ref[p - P.ref_lo] == '\0'; };
    for (int i = 0; i < ncig; ++i) {
        const int l = (int)(cigar[i] >> 4), op = (int)(cigar[i] & 0xf);
        if (op == 0 || op == 7 || op == 8) {
            int j;
            for (j = 0; j < l; ++j) {
                const int z = y + j;
                if (past_end(x + j)) break;                                // out of bounds
                const int c1 = seq[z] & 15, c2 = cap_nt16_of(P.ref[x + j - P.ref_lo]);
                if (c2 != 15 && c1 != 15 && qual[z] >= 13) {               // not ambiguous
                    ++len;
                    if (c1 && c1 != c2 && qual[z] >= 13) { ++mm; q += qual[z] > 33 ? 33 : qual[z]; }   // mismatch
                }
            }
            if (j < l) break;
            x += l; y += l; len += l;
        } else if (op == 2) {
            int j;
            for (j = 0; j < l; ++j) if (past_end(x + j)) break;
            if (j < l) break;
            x += l;
        } else if (op == 4) {
            for (int j = 0; j < l; ++j) clip_q += qual[y + j];
            y += l;
        } else if (op == 5) clip_q += 13 * l;
        else if (op == 1) y += l;
        else if (op == 3) x += l;
    }
    double t = 1;
    for (int i = 0; i < mm; ++i) t *= (double)len / (i + 1);
    t = q - 4.343 * log(t) + clip_q / 5.;
    int res;
    if (t > thres) res = -1;
    else {
        if (t < 0) t = 0;
        t = sqrt((thres - t) / thres) * thres;
        res = (int)(t + .499);
    }
    P.out[r] = res;
}

}  // namespace bcfgpu

using namespace bcfgpu;

namespace bcfgpu {
// mpileup.c:238: a read whose mapping quality is above its cap is lowered to it
__global__ __launch_bounds__(256) void cap_apply_kernel(int n, const int32_t *cap, uint8_t *mapq)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < n && cap[r] >= 0 && (int)mapq[r] > cap[r]) mapq[r] = (uint8_t)cap[r];
}
}

extern "C" void *bcfgpu_internal_pool_state(bcfgpu_ctx *ctx);
int bcfgpu_internal_pool_extent(bcfgpu_ctx *ctx, int *lo, int *hi);

// The same on the pool in HBM (after bcfgpu_pool_baq): the caps come back for the caller's filters (cap < 0: the read is
// dropped, mpileup.c:237), the pool's mapping qualities are lowered in place.
extern "C" int bcfgpu_pool_cap_mapq(bcfgpu_ctx *ctx, const char *ref, int32_t ref_len, int32_t thres, int32_t *cap)
{
    if (!ctx || !cap || (ref_len > 0 && !ref)) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pool_cap_mapq: bad arguments");
    hipStream_t st = nullptr;
    if (bcfgpu_internal_device(ctx, &st, nullptr)) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pool_cap_mapq: bad context");
    const DevPool &D = *static_cast<const DevPool*>(bcfgpu_internal_pool_state(ctx));
    if (!D.valid) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pool_cap_mapq: no read pool on this context (bcfgpu_pool_upload)");
    const int n = D.n_reads;
    if (!n) return BCFGPU_OK;
    int lo = 0, hi = 0;
    if (bcfgpu_internal_pool_extent(ctx, &lo, &hi)) return bcfgpu_set_error(BCFGPU_E_HIP, "bcfgpu_pool_cap_mapq: pool extent");
    if (lo < 0) lo = 0;
    if (hi > ref_len) hi = ref_len;
    if (hi < lo) hi = lo;
    CapParams P{};
    P.n_reads = n; P.thres = thres;
    P.r_pos = D.r_pos; P.r_lq = D.r_lq; P.r_ncig = D.r_ncig; P.r_cig_off = D.r_cig_off; P.r_seq_off = D.r_seq_off;
    P.cig = D.cig; P.seq16 = D.seq16; P.qual = D.qual;
    char *d_ref = (char*)bcfgpu_internal_ws(ctx, 125, (size_t)(hi - lo) + 64);
    P.out = (int32_t*)bcfgpu_internal_ws(ctx, 126, (size_t)n * 4 + 64);
    if (!d_ref || !P.out) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_pool_cap_mapq: device workspace");
    if (hi > lo && hipMemcpyAsync(d_ref, ref + lo, (size_t)(hi - lo), hipMemcpyHostToDevice, st) != hipSuccess)
        return bcfgpu_set_error(BCFGPU_E_HIP, "bcfgpu_pool_cap_mapq: upload");
    P.ref = d_ref; P.ref_lo = lo; P.ref_hi = hi;
    hipLaunchKernelGGL(cap_mapq_kernel, dim3((n + 255) / 256), dim3(256), 0, st, P);
    hipLaunchKernelGGL(cap_apply_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, (const int32_t*)P.out, D.r_mapq);
    if (hipGetLastError() != hipSuccess) return bcfgpu_set_error(BCFGPU_E_HIP, "bcfgpu_pool_cap_mapq: launch");
    if (hipMemcpyAsync(cap, P.out, (size_t)n * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return bcfgpu_set_error(BCFGPU_E_HIP, "bcfgpu_pool_cap_mapq: download");
    return BCFGPU_OK;
}

extern "C" int bcfgpu_cap_mapq(bcfgpu_ctx *ctx, const bcfgpu_reads *rd, const char *ref, int32_t ref_len, int32_t thres, int32_t *cap)
{
    if (!ctx || !rd || !cap || rd->n_reads < 0 || (ref_len > 0 && !ref)) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_cap_mapq: bad arguments");
    hipStream_t st = nullptr;
    if (bcfgpu_internal_device(ctx, &st, nullptr)) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_cap_mapq: bad context");
    const int n = rd->n_reads;
    if (!n) return BCFGPU_OK;
    size_t nbase = 0, ncig = 0;
    long lo = LONG_MAX, hi = 0;
    for (int r = 0; r < n; ++r) {
        const size_t e = (size_t)rd->r_seq_off[r] + rd->r_lq[r], c = (size_t)rd->r_cig_off[r] + rd->r_ncig[r];
        if (e > nbase) nbase = e;
        if (c > ncig) ncig = c;
        long x = rd->r_pos[r];
        if (x < lo) lo = x;
        for (int k = 0; k < rd->r_ncig[r]; ++k) { const uint32_t cg = rd->cig[rd->r_cig_off[r] + k]; const int op = cg & 15; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) x += cg >> 4; }
        if (x > hi) hi = x;
    }
    if (lo < 0) lo = 0;
    if (hi > ref_len) hi = ref_len;
    if (hi < lo) hi = lo;
    #define CQ_CHK(call) do { if ((call) != hipSuccess) return bcfgpu_set_error(BCFGPU_E_HIP, #call); } while (0)
    auto up = [&](int slot, const void *src, size_t bytes) -> void* {
        void *d = bcfgpu_internal_ws(ctx, slot, bytes + 16);
        if (d && bytes && hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return nullptr;
        return d;
    };
    CapParams P{};
    P.n_reads = n; P.thres = thres;
    P.r_pos = (const int32_t*)up(0, rd->r_pos, (size_t)n * 4); P.r_lq = (const int32_t*)up(1, rd->r_lq, (size_t)n * 4);
    P.r_ncig = (const int32_t*)up(2, rd->r_ncig, (size_t)n * 4); P.r_cig_off = (const int32_t*)up(3, rd->r_cig_off, (size_t)n * 4);
    P.r_seq_off = (const int32_t*)up(4, rd->r_seq_off, (size_t)n * 4); P.cig = (const uint32_t*)up(5, rd->cig, ncig * 4);
    P.seq16 = (const uint8_t*)up(6, rd->seq16, nbase); P.qual = (const uint8_t*)up(7, rd->qual, nbase);
    P.ref = (const char*)up(8, ref + lo, (size_t)(hi - lo)); P.ref_lo = lo; P.ref_hi = hi;
    P.out = (int32_t*)bcfgpu_internal_ws(ctx, 9, (size_t)n * 4 + 16);
    if (!P.r_pos || !P.r_lq || !P.r_ncig || !P.r_cig_off || !P.r_seq_off || !P.cig || !P.seq16 || !P.qual || !P.ref || !P.out)
        return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_cap_mapq: device workspace");
    hipLaunchKernelGGL(cap_mapq_kernel, dim3((n + 255) / 256), dim3(256), 0, st, P);
    CQ_CHK(hipGetLastError());
    CQ_CHK(hipMemcpyAsync(cap, P.out, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    CQ_CHK(hipStreamSynchronize(st));
    #undef CQ_CHK
    return BCFGPU_OK;
}

// baq.hip -- BAQ (base alignment quality): htslib's sam_prob_realn() for a pool of reads, as `bcftools mpileup`
// applies it to every read before the pileup (mpileup.c:234).
//
// Per read: reference window and band (realn.c), the banded glocal pair-HMM of probaln_glocal() with the full
// forward and backward passes and the posterior maximum per query base (probaln.c), then the quality cap
// (plain or extended).  One lane per read.  The usual bands (half-width 7 or 8) keep the current row of both passes in
// registers: of the forward matrix only the odd rows go to memory (unscaled, [row][cell][lane]: a wavefront's reads side by
// side), and the backward pass re-forms the even ones from the row below while it forms their posterior (baq_fb_reg).  Wider
// bands keep both matrices in a scratch buffer [row][cell][read] (baq_fb_scratch).  All arithmetic is fp64 in the
// reference's operation order, so state / posterior quality, and therefore the new base qualities, are identical to
// the CPU's.  The host half (window, band, 2-bit reference) is a few integer operations per read.
#include <hip/hip_runtime.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include <type_traits>
#include <cstring>
#include <cstdlib>
#include "kernels.h"

struct bcfgpu_ctx;
int bcfgpu_set_error(int code, const char *what);
extern "C" int bcfgpu_internal_device(bcfgpu_ctx *ctx, hipStream_t *stream, const float **q2p);
extern "C" void *bcfgpu_internal_ws(bcfgpu_ctx *ctx, int slot, size_t bytes);
extern "C" void *bcfgpu_internal_pinned(bcfgpu_ctx *ctx, int slot, size_t bytes);
extern "C" int bcfgpu_internal_side(bcfgpu_ctx *ctx, hipStream_t **streams, hipEvent_t **events);

namespace bcfgpu {

#define EI .25
#define EM .33333333333

struct BaqJob {
    uint32_t ref_off, seq_off, cig_off;       // into the window pool / the reads' seq16+qual pools / the cigar pool
    int32_t l_ref, l_query, bw, xb, pos, n_cigar, ret;
};

#define BAQ_ROWS_KEPT(max_lq) (((max_lq) + 1) / 2 + 1)      // forward rows the register-row classes keep per read: the odd ones
struct BaqParams {
    int n_jobs, ncell, max_lq, flag;
    size_t stride;                            // jobs per chunk (scratch row stride)
    const BaqJob *jobs;
    const uint8_t *tref, *seq16, *qual;       // 0..4 codes of the windows; the reads' 4-bit bases and qualities
    const float *q2p;                         // 10^(-Q/10) as float, Q = 0..255
    const uint32_t *cig;
    double *F, *B, *S;                        // [max_lq+1][ncell][stride] forward / backward; [max_lq+2][stride] scales
    int32_t *state;                           // [pool offset] posterior state per base (k-1)<<2 | {0 M, 1 I}
    uint8_t *q, *tmp;                         // [pool offset] posterior quality; scratch for the extended cap
    int32_t *wstate; uint8_t *wq, *wleft;     // the same three for the register-row classes, a wavefront's reads side by side: [job >> 6][base][lane]
                                              // (a lane's own 100-byte stretch made every access of the cap 64 cache lines, and the
                                              // two stores per row of the backward pass 64 partial sectors each)
    uint8_t *qual_out, *zq_out;
    int lds_rows;                             // scratch class: 1 = the working rows in LDS (baq_fb_lds; the launch carries 2 * ncell * BAQ_WIDE_LANES doubles)
};

__device__ __forceinline__ int set_u(int b, int i, int k) { int x = i - b; x = x > 0 ? x : 0; return (k - x + 1) * 3; }
__device__ __forceinline__ int nt16_to_4(int c) { return (int)((0x4444444344424104ull >> (4 * (c & 15))) & 7); }

// forward, backward and posterior maximum with both matrices in the scratch buffer: any band width
__device__ void baq_fb_scratch(const BaqParams &P, int job, const BaqJob &j, const uint8_t *ref, const uint8_t *seq,
                               const uint8_t *iqual, int32_t *state, uint8_t *q)
{
    const int l_query = j.l_query, l_ref = j.l_ref;
    const size_t st = P.stride;
    #define FM(M_, i, c) (M_)[((size_t)(i) * P.ncell + (c)) * st + job]
    #define SC(i) P.S[(size_t)(i) * st + job]
    int bw = l_ref > l_query ? l_ref : l_query;
    if (bw > j.bw) bw = j.bw;
    if (bw < abs(l_ref - l_query)) bw = abs(l_ref - l_query);
    const int bw2 = bw * 2 + 1, nc = bw2 * 3 + 6;
    const double d = 0.001, e_ = 0.1;                    // realn.c: conf = { 0.001, 0.1, 10 }, bw overridden
    double m[9];
    const double sM = 1. / (2 * l_query + 2), sI = sM;
    m[0] = (1 - d - d) * (1 - sM); m[1] = m[2] = d * (1 - sM);
    m[3] = (1 - e_) * (1 - sI); m[4] = e_ * (1 - sI); m[5] = 0.;
    m[6] = 1 - e_; m[7] = 0.; m[8] = e_;
    const double bM = (1 - d) / l_ref, bI = d / l_ref;
    // rows 0..l_query of both matrices start as zeros
    for (int i = 0; i <= l_query; ++i)
        for (int c = 0; c < nc; ++c) { FM(P.F, i, c) = 0.; FM(P.B, i, c) = 0.; }
    auto qy = [&](int i) { return nt16_to_4(seq[i]); };
    auto qp = [&](int i) { return (double)P.q2p[iqual[i]]; };   // (float)pow(10, -Q/10), probaln.c
    // ---- forward ----
    FM(P.F, 0, set_u(bw, 0, 0)) = 1.; SC(0) = 1.;
    {
        double sum = 0.;
        const int beg = 1, end = l_ref < bw + 1 ? l_ref : bw + 1;
        const double q0 = qp(0);
        const int y0 = qy(0);
        for (int k = beg; k <= end; ++k) {
            const double e = (ref[k - 1] > 3 || y0 > 3) ? 1. : ref[k - 1] == y0 ? 1. - q0 : q0 * EM;
            const int u = set_u(bw, 1, k);
            const double a = e * bM, b = EI * bI;
            FM(P.F, 1, u) = a; FM(P.F, 1, u + 1) = b;
            sum += a + b;
        }
        SC(1) = sum;
        const int _beg = set_u(bw, 1, beg), _end = set_u(bw, 1, end) + 2;
        for (int k = _beg; k <= _end; ++k) FM(P.F, 1, k) /= sum;
    }
    for (int i = 2; i <= l_query; ++i) {
        const double qli = qp(i - 1);
        const int qyi = qy(i - 1);
        int beg = 1, end = l_ref, x;
        x = i - bw; beg = beg > x ? beg : x;
        x = i + bw; end = end < x ? end : x;
        double sum = 0.;
        for (int k = beg; k <= end; ++k) {
            const double e = (ref[k - 1] > 3 || qyi > 3) ? 1. : ref[k - 1] == qyi ? 1. - qli : qli * EM;
            const int u = set_u(bw, i, k), v11 = set_u(bw, i - 1, k - 1), v10 = set_u(bw, i - 1, k), v01 = set_u(bw, i, k - 1);
            const double f0 = e * (m[0] * FM(P.F, i - 1, v11) + m[3] * FM(P.F, i - 1, v11 + 1) + m[6] * FM(P.F, i - 1, v11 + 2));
            const double f1 = EI * (m[1] * FM(P.F, i - 1, v10) + m[4] * FM(P.F, i - 1, v10 + 1));
            const double f2 = m[2] * FM(P.F, i, v01) + m[8] * FM(P.F, i, v01 + 2);
            FM(P.F, i, u) = f0; FM(P.F, i, u + 1) = f1; FM(P.F, i, u + 2) = f2;
            sum += f0 + f1 + f2;
        }
        SC(i) = sum;
        const int _beg = set_u(bw, i, beg), _end = set_u(bw, i, end) + 2;
        const double r = 1. / sum;
        for (int k = _beg; k <= _end; ++k) FM(P.F, i, k) *= r;
    }
    {
        double sum = 0.;
        for (int k = 1; k <= l_ref; ++k) {
            const int u = set_u(bw, l_query, k);
            if (u < 3 || u >= bw2 * 3 + 3) continue;
            sum += FM(P.F, l_query, u) * sM + FM(P.F, l_query, u + 1) * sI;
        }
        SC(l_query + 1) = sum;
    }
    // ---- backward ----
    {
        const double sl = SC(l_query), sl1 = SC(l_query + 1);
        for (int k = 1; k <= l_ref; ++k) {
            const int u = set_u(bw, l_query, k);
            if (u < 3 || u >= bw2 * 3 + 3) continue;
            FM(P.B, l_query, u) = sM / sl / sl1; FM(P.B, l_query, u + 1) = sI / sl / sl1;
        }
    }
    for (int i = l_query - 1; i >= 1; --i) {
        int beg = 1, end = l_ref, x;
        double y = (i > 1);
        const double qli1 = qp(i);
        const int qyi1 = qy(i);
        x = i - bw; beg = beg > x ? beg : x;
        x = i + bw; end = end < x ? end : x;
        for (int k = end; k >= beg; --k) {
            const int u = set_u(bw, i, k), v11 = set_u(bw, i + 1, k + 1), v10 = set_u(bw, i + 1, k), v01 = set_u(bw, i, k + 1);
            const double e = (k >= l_ref ? 0 : (ref[k] > 3 || qyi1 > 3) ? 1. : ref[k] == qyi1 ? 1. - qli1 : qli1 * EM) * FM(P.B, i + 1, v11);
            const double b10 = FM(P.B, i + 1, v10 + 1), b01 = FM(P.B, i, v01 + 2);
            FM(P.B, i, u) = e * m[0] + EI * m[1] * b10 + m[2] * b01;
            FM(P.B, i, u + 1) = e * m[3] + EI * m[4] * b10;
            FM(P.B, i, u + 2) = (e * m[6] + m[8] * b01) * y;
        }
        const int _beg = set_u(bw, i, beg), _end = set_u(bw, i, end) + 2;
        y = 1. / SC(i);
        for (int k = _beg; k <= _end; ++k) FM(P.B, i, k) *= y;
    }
    // (the backward termination b[0] of probaln.c only feeds a debugging value)
    // ---- MAP ----
    for (int i = 1; i <= l_query; ++i) {
        double sum = 0., max = 0.;
        int beg = 1, end = l_ref, x, max_k = -1;
        x = i - bw; beg = beg > x ? beg : x;
        x = i + bw; end = end < x ? end : x;
        for (int k = beg; k <= end; ++k) {
            const int u = set_u(bw, i, k);
            double z;
            z = FM(P.F, i, u) * FM(P.B, i, u);         if (z > max) { max = z; max_k = (k - 1) << 2 | 0; } sum += z;
            z = FM(P.F, i, u + 1) * FM(P.B, i, u + 1); if (z > max) { max = z; max_k = (k - 1) << 2 | 1; } sum += z;
        }
        max /= sum;
        state[i - 1] = max_k;
        const int kq = (int)(-4.343 * log(1. - max) + .499);
        q[i - 1] = (uint8_t)(kq > 100 ? 99 : kq);
    }
    #undef FM
    #undef SC
}


// The same arithmetic, cell by cell, with the two rows a pass works on in LDS (rowA / rowB: this lane's columns, cell c at
// [c * LANES]) instead of in the scratch matrices.  In baq_fb_scratch every cell waits for global-memory round trips -- the
// row above, and the cell it has just stored -- and a read with a 40-base indel (a band of 43: 87 cells x 100 rows x three
// passes) takes 65 ms whatever else the chip is doing.  Here the forward pass reads the row above from LDS, carries the cell
// to the left in registers and only STORES its scaled rows to the scratch matrix F (the posterior needs f x b per cell); the
// backward pass keeps its rows in LDS alone and forms a row's posterior maximum as soon as the row is scaled, from F's row
// (loads nothing waits for but that row's own sums).  Every expression is baq_fb_scratch's, in its order.
template <int LANES>
__device__ void baq_fb_lds(const BaqParams &P, int job, const BaqJob &j, const uint8_t *ref, const uint8_t *seq,
                           const uint8_t *iqual, int32_t *state, uint8_t *q, double *rowA, double *rowB, double *rowF, uint8_t *lref)
{
    const int l_query = j.l_query, l_ref = j.l_ref;
    const size_t st = P.stride;
    #define FG(i, c) P.F[((size_t)(i) * P.ncell + (c)) * st + job]
    #define SC(i) P.S[(size_t)(i) * st + job]
    #define LR(buf, c) (buf)[(size_t)(c) * LANES]
    int bw = l_ref > l_query ? l_ref : l_query;
    if (bw > j.bw) bw = j.bw;
    if (bw < abs(l_ref - l_query)) bw = abs(l_ref - l_query);
    const int bw2 = bw * 2 + 1, nc = bw2 * 3 + 6;
    const double d = 0.001, e_ = 0.1;
    double m[9];
    const double sM = 1. / (2 * l_query + 2), sI = sM;
    m[0] = (1 - d - d) * (1 - sM); m[1] = m[2] = d * (1 - sM);
    m[3] = (1 - e_) * (1 - sI); m[4] = e_ * (1 - sI); m[5] = 0.;
    m[6] = 1 - e_; m[7] = 0.; m[8] = e_;
    const double bM = (1 - d) / l_ref, bI = d / l_ref;
    auto qy = [&](int i) { return nt16_to_4(seq[i]); };
    auto qp = [&](int i) { return (double)P.q2p[iqual[i]]; };
    double *prev = rowA, *cur = rowB;
    // the window's bases into LDS, eight loads in flight at a time (a load per cell from device memory is a wait per cell)
    #define RF(k) lref[(size_t)(k) * LANES]
    for (int k0 = 0; k0 < l_ref; k0 += 8) {
        uint8_t t[8];
        #pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = ref[k0 + u < l_ref ? k0 + u : l_ref - 1];
        #pragma unroll
        for (int u = 0; u < 8; ++u) if (k0 + u < l_ref) RF(k0 + u) = t[u];
    }
    // ---- forward ----
    SC(0) = 1.;
    for (int c = 0; c < nc; ++c) LR(cur, c) = 0.;
    {
        double sum = 0.;
        const int beg = 1, end = l_ref < bw + 1 ? l_ref : bw + 1;
        const double q0 = qp(0);
        const int y0 = qy(0);
        for (int k = beg; k <= end; ++k) {
            const int rk = RF(k - 1);
            const double e = (rk > 3 || y0 > 3) ? 1. : rk == y0 ? 1. - q0 : q0 * EM;
            const int u = set_u(bw, 1, k);
            const double a = e * bM, b = EI * bI;
            LR(cur, u) = a; LR(cur, u + 1) = b;
            sum += a + b;
        }
        SC(1) = sum;
        const int _beg = set_u(bw, 1, beg), _end = set_u(bw, 1, end) + 2;
        for (int k = _beg; k <= _end; ++k) { const double v = LR(cur, k) / sum; LR(cur, k) = v; FG(1, k) = v; }
    }
    for (int i = 2; i <= l_query; ++i) {
        { double *t = prev; prev = cur; cur = t; }
        for (int c = 0; c < nc; ++c) LR(cur, c) = 0.;
        const double qli = qp(i - 1);
        const int qyi = qy(i - 1);
        int beg = 1, end = l_ref, x;
        x = i - bw; beg = beg > x ? beg : x;
        x = i + bw; end = end < x ? end : x;
        double sum = 0.;
        double lM = 0., lD = 0.;                        // f[i][k-1]: M and D of the cell to the left (zeros in front of the band)
        for (int k = beg; k <= end; ++k) {
            const int rk = RF(k - 1);
            const double e = (rk > 3 || qyi > 3) ? 1. : rk == qyi ? 1. - qli : qli * EM;
            const int u = set_u(bw, i, k), v11 = set_u(bw, i - 1, k - 1), v10 = set_u(bw, i - 1, k);
            const double f0 = e * (m[0] * LR(prev, v11) + m[3] * LR(prev, v11 + 1) + m[6] * LR(prev, v11 + 2));
            const double f1 = EI * (m[1] * LR(prev, v10) + m[4] * LR(prev, v10 + 1));
            const double f2 = m[2] * lM + m[8] * lD;
            LR(cur, u) = f0; LR(cur, u + 1) = f1; LR(cur, u + 2) = f2;
            sum += f0 + f1 + f2;
            lM = f0; lD = f2;
        }
        SC(i) = sum;
        const int _beg = set_u(bw, i, beg), _end = set_u(bw, i, end) + 2;
        const double r = 1. / sum;
        for (int k = _beg; k <= _end; ++k) { const double v = LR(cur, k) * r; LR(cur, k) = v; FG(i, k) = v; }
    }
    double sl1;
    {
        double sum = 0.;
        for (int k = 1; k <= l_ref; ++k) {
            const int u = set_u(bw, l_query, k);
            if (u < 3 || u >= bw2 * 3 + 3) continue;
            sum += LR(cur, u) * sM + LR(cur, u + 1) * sI;
        }
        sl1 = sum;
    }
    // the posterior maximum of row i from the scaled backward row `b` (LDS) and the scaled forward row (scratch)
    auto map_row = [&](int i, const double *b) {
        double sum = 0., max = 0.;
        int beg = 1, end = l_ref, x, max_k = -1;
        x = i - bw; beg = beg > x ? beg : x;
        x = i + bw; end = end < x ? end : x;
        // the scaled forward row from the scratch matrix into LDS, eight loads in flight at a time
        const int c0 = set_u(bw, i, beg), c1 = set_u(bw, i, end) + 1;
        for (int c = c0; c <= c1; c += 8) {
            double t[8];
            #pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = FG(i, c + u <= c1 ? c + u : c1);
            #pragma unroll
            for (int u = 0; u < 8; ++u) if (c + u <= c1) LR(rowF, c + u) = t[u];
        }
        for (int k = beg; k <= end; ++k) {
            const int u = set_u(bw, i, k);
            double z;
            z = LR(rowF, u) * LR(b, u);         if (z > max) { max = z; max_k = (k - 1) << 2 | 0; } sum += z;
            z = LR(rowF, u + 1) * LR(b, u + 1); if (z > max) { max = z; max_k = (k - 1) << 2 | 1; } sum += z;
        }
        max /= sum;
        state[i - 1] = max_k;
        const int kq = (int)(-4.343 * log(1. - max) + .499);
        q[i - 1] = (uint8_t)(kq > 100 ? 99 : kq);
    };
    // ---- backward, each row's posterior maximum right behind it ----
    double *nxt = prev;                                 // (the forward rows are done with: the two buffers serve the backward pass)
    for (int c = 0; c < nc; ++c) LR(cur, c) = 0.;
    {
        const double sl = SC(l_query);
        for (int k = 1; k <= l_ref; ++k) {
            const int u = set_u(bw, l_query, k);
            if (u < 3 || u >= bw2 * 3 + 3) continue;
            LR(cur, u) = sM / sl / sl1; LR(cur, u + 1) = sI / sl / sl1;
        }
    }
    map_row(l_query, cur);
    for (int i = l_query - 1; i >= 1; --i) {
        { double *t = nxt; nxt = cur; cur = t; }
        for (int c = 0; c < nc; ++c) LR(cur, c) = 0.;
        int beg = 1, end = l_ref, x;
        double y = (i > 1);
        const double qli1 = qp(i);
        const int qyi1 = qy(i);
        x = i - bw; beg = beg > x ? beg : x;
        x = i + bw; end = end < x ? end : x;
        double rD = 0.;                                 // b[i][k+1]'s D, unscaled (zero past the band)
        for (int k = end; k >= beg; --k) {
            const int u = set_u(bw, i, k), v11 = set_u(bw, i + 1, k + 1), v10 = set_u(bw, i + 1, k);
            const int rk = k >= l_ref ? 4 : RF(k);
            const double e = (k >= l_ref ? 0 : (rk > 3 || qyi1 > 3) ? 1. : rk == qyi1 ? 1. - qli1 : qli1 * EM) * LR(nxt, v11);
            const double b10 = LR(nxt, v10 + 1), b01 = rD;
            const double vM = e * m[0] + EI * m[1] * b10 + m[2] * b01;
            const double vI = e * m[3] + EI * m[4] * b10;
            const double vD = (e * m[6] + m[8] * b01) * y;
            LR(cur, u) = vM; LR(cur, u + 1) = vI; LR(cur, u + 2) = vD;
            rD = vD;
        }
        const int _beg = set_u(bw, i, beg), _end = set_u(bw, i, end) + 2;
        y = 1. / SC(i);
        for (int k = _beg; k <= _end; ++k) LR(cur, k) *= y;
        map_row(i, cur);
    }
    #undef FG
    #undef SC
    #undef LR
    #undef RF
}


__device__ __forceinline__ double uniform_f64(double x)      // the first running lane's value, in scalar registers
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// The same with the current row in registers, for bands of half-width <= BW (the default band of 7 fits): the forward pass
// keeps its scaled row in registers and stores every other one (the posterior needs f*b per cell), the backward pass keeps
// its row in registers, reads the stored forward rows back and re-forms the others.
// Layout: ALWAYS SLIDING, as in probaln_fwd_exact (indel.hip) -- position p (1..W, W = 2 BW + 1) of row i is reference column
// k = p + i - bw - 1, for every row.  (htslib's set_u() pins the band to column 0 for the first bw rows; the layout is storage
// only: the cells and the order of every sum are the reference's.)  Then
//   forward   f[i][k] reads f[i-1][k-1] = old p,  f[i-1][k] = old p + 1,  f[i][k-1] = new p - 1        (ascending p, in place)
//   backward  b[i][k] reads b[i+1][k+1] = old p,  b[i+1][k] = old p - 1,  b[i][k+1] = new p + 1        (descending p, in place)
// with no choice between a sliding and a resting band in any cell.  Cells under the band's lower edge (k < 1) are zero in the
// forward rows by construction (their three neighbours are); in the backward rows they hold values no live cell reads (a live
// cell's neighbours have k >= 1) and their posterior term is 0 * b = 0.  So only the upper edge -- k > l_ref, or a band narrower
// than BW -- needs masks, and a read of 100 bases in a window of 107 has no such row at all.  Every row body exists twice, each
// straight-line code: the fast one (every cell of the row alive for EVERY read of the wavefront, no N under the band: no range
// test, no mask, an emission = one comparison of byte codes and a select of the row's two values) and the edge one (masks, N,
// reads of other lengths).  Each pass is a run of edge rows, a run of fast rows while every read of the wavefront qualifies, and
// edge rows to the end -- separate loops, not a choice per row: with both bodies behind one branch the compiler forms the
// products they share (m0 * M[p], ... -- 75 of them) in front of the branch and keeps them alive across it, 150 registers the
// kernel does not have.  (Round 4's single body tested plo <= p <= phi and chose between the resting and the sliding neighbour
// in every cell: 47 instructions a cell for the 19 fp64 operations of the forward recurrence.)
// The reference window travels in W bytes of registers, a byte code per position (forward: one down per row, backward: one up).
// The backward pass counts its rows down from the longest read of the wavefront, a shorter read joining at its own last row, so
// that the row number -- odd or even: which posterior body runs -- is the same for all lanes.
template <int BW>
__device__ __forceinline__ void baq_fb_reg(const BaqParams &P, int job, const BaqJob &j, const uint8_t *ref, const uint8_t *seq,
                                           const uint8_t *iqual, int bw, const float *lq2p)
{
    constexpr int W = 2 * BW + 1, NB = (W + 3) / 4, NP = 2 * BW + 3;
    using Fast = std::true_type; using Edge = std::false_type;
    const int l_query = j.l_query, l_ref = j.l_ref;
    // A wavefront's rows are one contiguous block, [row][cell][lane]: consecutive stores are 512 bytes apart.  (With the
    // rows of all reads of a launch interleaved -- [row][cell][read] -- consecutive stores of a wavefront were megabytes
    // apart, every one on a DRAM page and a TLB entry of its own: 1.8 TB/s.)  M and I only: the posterior never reads D;
    // M and I of a cell side by side in a lane's 16 bytes: one dwordx4 store / load per cell.
    // (the wavefront's base and the row's offset are the same for all lanes -- the row number is: scalar address arithmetic, the
    // lane's 16 or 8 bytes the only vector part)
    const unsigned wave = blockIdx.x, lane = threadIdx.x;         // = job >> 6, job & 63
    double2 *const Fw = reinterpret_cast<double2*>(P.F) + (size_t)wave * (size_t)BAQ_ROWS_KEPT(P.max_lq) * NP * 64;
    double *const Sw = P.S + (size_t)wave * (size_t)(P.max_lq + 2) * 64;
    int32_t *const stw = P.wstate + (size_t)wave * (size_t)P.max_lq * 64;
    uint8_t *const qw = P.wq + (size_t)wave * (size_t)P.max_lq * 64;
    #define FR2(i, p) (Fw + ((size_t)((i) >> 1) * NP + (size_t)(p)) * 64)[lane]       // the odd rows 1, 3, 5, ... in slots 0, 1, 2, ...
    #define SC(i) (Sw + (size_t)(i) * 64)[lane]
    const double d = 0.001, e_ = 0.1;
    const double sM = 1. / (2 * l_query + 2), sI = sM;
    const double m0 = (1 - d - d) * (1 - sM), m1 = d * (1 - sM), m2 = m1;
    const double m3 = (1 - e_) * (1 - sI), m4 = e_ * (1 - sI);
    const double m6 = 1 - e_, m8 = e_;
    const double m1q = EI * m1, m4q = EI * m4;            // (EI = 1/4: scaling by a power of two commutes with rounding)
    // The fast rows run only while all reads of the wavefront have one length: then the coefficients are the same in every lane and
    // sit in scalar registers there (14 vector registers of the 256 the two passes need; the edge rows use the lanes' own).
    const int lq_u = __builtin_amdgcn_readfirstlane(l_query);
    const double m0u = uniform_f64(m0), m1qu = uniform_f64(m1q), m2u = uniform_f64(m2), m3u = uniform_f64(m3), m4qu = uniform_f64(m4q);
    const double bM = (1 - d) / l_ref, bI = d / l_ref;
    const int top = 2 * bw + 1, lq1 = l_query - 1, lr1 = l_ref - 1;
    double M[W + 2], I[W + 2], D[W + 2];
    #pragma unroll
    for (int p = 0; p < W + 2; ++p) M[p] = I[p] = D[p] = 0.;
    // window: the code of the reference base under position p in byte p - 1 (forward: ref[k - 1], k the column of p in this row);
    // 0 outside the reference -- a dead cell's emission is never used
    uint32_t wv[NB];
    #pragma unroll
    for (int b = 0; b < NB; ++b) wv[b] = 0;
    #pragma unroll
    for (int p = 1; p <= W; ++p) { const int k = p - bw; wv[(p - 1) >> 2] |= (uint32_t)((k >= 1 && k <= l_ref) ? ref[k - 1] : 0) << (8 * ((p - 1) & 3)); }
    auto wbyte = [&](int p) { return (wv[(p - 1) >> 2] >> (8 * ((p - 1) & 3))) & 0xffu; };
    auto wdown = [&](uint32_t top_code) {                  // every position one down, top_code into position W
        #pragma unroll
        for (int b = 0; b + 1 < NB; ++b) wv[b] = __builtin_amdgcn_alignbyte(wv[b + 1], wv[b], 1);
        wv[NB - 1] >>= 8;
        wv[(W - 1) >> 2] |= top_code << (8 * ((W - 1) & 3));
    };
    auto wup = [&](uint32_t bottom_code) {                 // every position one up, bottom_code into position 1
        #pragma unroll
        for (int b = NB - 1; b >= 1; --b) wv[b] = __builtin_amdgcn_alignbyte(wv[b], wv[b - 1], 3);
        wv[0] = (wv[0] << 8) | bottom_code;
        if (W & 3) wv[NB - 1] &= (1u << (8 * (W & 3))) - 1u;
    };
    auto any_n = [&]() { uint32_t a = 0;
        #pragma unroll
        for (int b = 0; b < NB; ++b) a |= wv[b];
        return (a & 0x04040404u) != 0; };
    auto ref_at = [&](int k) { return (uint32_t)ref[k < 0 ? 0 : k > lr1 ? lr1 : k]; };   // (clamped: requested whether or not it is used)
    // ---- forward ----
    // The bytes a row needs (the query base, its quality, the reference base that slides into the band) are requested one row
    // ahead, BEFORE the stores of the row in between: vmcnt counts loads and stores in issue order, so a load requested
    // after a row's stores is only known to be there when those stores are.
    double s_last;                                         // the scale of the last row
    {   // row 1: k = p - bw
        double sum = 0.;
        const int end = l_ref < bw + 1 ? l_ref : bw + 1;
        const double q0 = (double)lq2p[iqual[0]];
        const uint32_t y0 = (uint32_t)nt16_to_4(seq[0]);
        #pragma unroll
        for (int p = 1; p <= W; ++p) {
            const bool live = p > bw && p <= bw + end;
            const uint32_t rb = wbyte(p);
            const double e = (rb > 3 || y0 > 3) ? 1. : rb == y0 ? 1. - q0 : q0 * EM;
            const double a = live ? e * bM : 0., b = live ? EI * bI : 0.;
            M[p] = a; I[p] = b;
            sum += a + b;
        }
        SC(1) = sum;
        s_last = sum;
        #pragma unroll
        for (int p = 1; p <= W; ++p) {
            FR2(1, p) = make_double2(M[p], I[p]);          // (unscaled, like every stored row: see below)
            M[p] /= sum; I[p] /= sum;
        }
    }
    uint32_t nq = iqual[1 < lq1 ? 1 : lq1], ns = seq[1 < lq1 ? 1 : lq1], nr = ref_at(W + 2 - bw - 2);
    auto fwd_row = [&](const int i, auto kind) {
        constexpr bool FAST = decltype(kind)::value;
        const uint32_t cq = nq, cs = ns, cr = nr;
        {   // row i + 1's bytes
            const int in = i < lq1 ? i : lq1;
            nq = iqual[in]; ns = seq[in]; nr = ref_at(W + (i + 1) - bw - 2);
        }
        const double qli = (double)lq2p[cq];
        const uint32_t qyi = (uint32_t)nt16_to_4((int)cs);
        const double ex = qyi > 3 ? 1. : 1. - qli, ey = qyi > 3 ? 1. : qli * EM;      // the row's emission over a matching / another base
        wdown(W + i - bw - 1 <= l_ref ? cr : 0u);
        double sum = 0.;
        if (FAST) {
            #pragma unroll
            for (int p = 1; p <= W; ++p) {
                const double e = wbyte(p) == qyi ? ex : ey;
                const double f0 = e * (m0u * M[p] + m3u * I[p] + m6 * D[p]);
                const double f1 = m1qu * M[p + 1] + m4qu * I[p + 1];
                const double f2 = m2u * M[p - 1] + m8 * D[p - 1];
                sum += f0 + f1 + f2;
                M[p] = f0; I[p] = f1; D[p] = f2;
            }
        } else {
            const int hi = l_ref - (i - bw) + 1 < top ? l_ref - (i - bw) + 1 : top;
            #pragma unroll
            for (int p = 1; p <= W; ++p) {
                const bool live = p <= hi;
                const uint32_t rb = wbyte(p);
                double e = rb > 3 ? 1. : rb == qyi ? ex : ey;
                e = live ? e : 0.;
                const double tv = live ? 1. : 0.;
                const double f0 = e * (m0 * M[p] + m3 * I[p] + m6 * D[p]);
                const double f1 = m1q * M[p + 1] + m4q * I[p + 1];
                const double f2 = tv * (m2 * M[p - 1] + m8 * D[p - 1]);
                sum += f0 + f1 + f2;
                M[p] = f0; I[p] = f1; D[p] = f2;
            }
        }
        SC(i) = sum;
        s_last = sum;
        const double r = 1. / sum;
        // Only the ODD rows go to memory, and UNSCALED (M', I'; the scale is stored anyway): the backward pass re-forms an
        // even row from the odd row below it while it forms that row's posterior.  Unscaled, because the row above needs D of
        // this one, which is not stored: D' is the recurrence D'[p] = m2 M'[p-1] + m8 D'[p-1] over the UNSCALED M' -- re-run
        // in the same order it gives the same bits, and the scaled row is re-formed by the same multiplication as here.
#ifdef BAQ_EXP_NOSTORE       // experiment: the forward rows are not written (never true at run time)
        if (P.n_jobs < 0)
#endif
        if (i & 1) {
            #pragma unroll
            for (int p = 1; p <= W; ++p) {
                // (non-temporal, like the loads that bring the row back: 6 GB a launch pass through once each way, a hundred rows
                // apart -- the forward pass alone, bound by these stores, 1.67 -> 1.18 ms for 482 k reads)
                typedef double v2d __attribute__((ext_vector_type(2)));
                v2d t; t.x = M[p]; t.y = I[p];
                __builtin_nontemporal_store(t, reinterpret_cast<v2d*>(&FR2(i, p)));
            }
        }
        #pragma unroll
        for (int p = 1; p <= W; ++p) { M[p] *= r; I[p] *= r; D[p] *= r; }
    };
    {
        int i = 2;
        // fast rows: every cell of the row inside the reference, for every read of the wavefront, no N about to be under the band
        for (;; ++i) {
            const bool fast = i <= l_query && l_query == lq_u && bw == BW && i + bw <= l_ref && !any_n() && nr < 4;
            if (__builtin_amdgcn_ballot_w64(!fast) != 0) break;
            fwd_row(i, Fast());
        }
        #pragma unroll 1
        for (; i <= l_query; ++i) fwd_row(i, Edge());
    }
    double s_end = 0.;                                     // f[l_query + 1]: over the last row (dead cells are zero)
    #pragma unroll
    for (int p = 1; p <= W; ++p) s_end += M[p] * sM + I[p] * sI;
    SC(l_query + 1) = s_end;
#if defined(BAQ_EXP_PHASE) && BAQ_EXP_PHASE == 1   // experiment: the forward pass only
    if (l_query > 0) { qw[lane] = (uint8_t)M[1]; return; }
#endif
    // ---- backward with the posterior maximum of every row ----
    // Iteration i: b[i] from b[i+1] (not for the last row), then the posterior of row i.  The window is the forward pass's: the
    // step's emission is that of (row i+1, column k+1) = the forward window of row i+1 at the same position; then it moves one
    // up and is row i's own, which the re-forming of an even row reads.
    // fr0 / fr1: the stored (odd, unscaled) row the next posterior works from -- its own for an odd row, the row below for an even
    // one; requested while the rows in between are worked on
    // (60 registers beside the 60 of the backward row and what a cell needs: more than the 256 of two wavefronts a SIMD -- the kernel
    // runs one wavefront a SIMD, with up to 512, see BAQ_WAVES.  Staging the row in LDS instead (copied there by global_load_lds, no
    // destination register) fits two wavefronts and was 7 % faster on its own, but its 16 KB a wavefront keep the wide-band class's
    // workgroups, which need 50 KB of LDS each, off the compute units: the handful of reads of that class then ran after the others
    // instead of beside them, 55 ms for a pool of 4.9 M reads that is through the register classes in 45.)
    double fr0[W + 2], fr1[W + 2];
    fr0[0] = fr1[0] = fr0[W + 1] = fr1[W + 1] = 0.;
    auto fr_request = [&](int row) {
        #pragma unroll
        for (int p = 1; p <= W; ++p) {
            typedef double v2d __attribute__((ext_vector_type(2)));
            const v2d t = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(&FR2(row, p)));
            fr0[p] = t.x; fr1[p] = t.y;
        }
    };
    auto fr = [&](int p) { return make_double2(fr0[p], fr1[p]); };
    bool fr_scaled = false;                                // row 1 in fr has been divided by its sum
    // bytes of the query: hi = base i (row i+1's, the step's), lo = base i-1 (row i's own); the scale of rows i and i-1; the
    // reference base that enters the window from below -- all requested an iteration ahead
    uint32_t q_hi = 0, s_hi = 0, q_lo = 0, s_lo = 0, r_in = 0;
    double sc_i = 1., sc_n = 1.;
    auto bwd_row = [&](const int i, auto kind) {
        constexpr bool FAST = decltype(kind)::value;
        if (!FAST && i == l_query) {                       // this read's last row: where its backward pass starts
            const int lo = bw + 2 - l_query > 1 ? bw + 2 - l_query : 1;
            const int hi = l_ref - l_query + bw + 1 < top ? l_ref - l_query + bw + 1 : top;
            const double vM = sM / s_last / s_end, vI = sI / s_last / s_end;
            #pragma unroll
            for (int p = 1; p <= W; ++p) {
                const bool in = p >= lo && p <= hi;
                M[p] = in ? vM : 0.; I[p] = in ? vI : 0.;
            }
            q_lo = iqual[lq1]; s_lo = seq[lq1];
            sc_i = s_last; sc_n = SC(l_query - 1 > 1 ? l_query - 1 : 1);
            fr_request((l_query & 1) ? l_query : l_query - 1);
        }
        const int i2 = i - 2 > 0 ? i - 2 : 0;
        const uint32_t rq_q = iqual[i2], rq_s = seq[i2], rq_r = ref_at(i - bw - 2);
        const double rq_sc = SC(i - 2 > 1 ? i - 2 : 1);
        const double r_i = 1. / sc_i;                     // (row 1 was scaled by a division, every other row by this reciprocal)
        if (FAST || i < l_query) {
            const double qli = (double)lq2p[q_hi];
            const uint32_t qyi = (uint32_t)nt16_to_4((int)s_hi);
            const double ex = qyi > 3 ? 1. : 1. - qli, ey = qyi > 3 ? 1. : qli * EM;
            if (FAST) {
                double b01 = 0.;                           // b[i][k+1].D, unscaled: the row's D state lives in this chain only
                #pragma unroll
                for (int p = W; p >= 1; --p) {
                    const double e = (wbyte(p) == qyi ? ex : ey) * M[p];
                    const double b10 = I[p - 1];
                    M[p] = e * m0u + m1qu * b10 + m2u * b01;
                    I[p] = e * m3u + m4qu * b10;
                    b01 = e * m6 + m8 * b01;              // (times y = 1; row 1's D, y = 0, is read by nothing)
                }
            } else {
                const int lo = bw + 2 - i > 1 ? bw + 2 - i : 1;
                const int hi = l_ref - i + bw + 1 < top ? l_ref - i + bw + 1 : top;
                const double y = (i > 1);
                double b01 = 0.;
                #pragma unroll
                for (int p = W; p >= 1; --p) {
                    const bool live = p >= lo && p <= hi;
                    const uint32_t rb = wbyte(p);
                    const double em = p + i - bw - 1 >= l_ref ? 0. : rb > 3 ? 1. : rb == qyi ? ex : ey;    // column k+1 beyond the reference: no emission
                    const double e = em * M[p];
                    const double b10 = I[p - 1];
                    const double nM = e * m0 + m1q * b10 + m2 * b01;
                    const double nI = e * m3 + m4q * b10;
                    const double nD = (e * m6 + m8 * b01) * y;
                    M[p] = live ? nM : 0.; I[p] = live ? nI : 0.; b01 = live ? nD : 0.;
                }
            }
            #pragma unroll
            for (int p = 1; p <= W; ++p) { M[p] *= r_i; I[p] *= r_i; }       // (the D state of a backward row is read within the row only)
            wup(i - bw - 1 >= 0 && i - bw - 1 < l_ref ? r_in : 0u);
        }
        // posterior of row i (probaln.c MAP): first maximum over k ascending, M before I.  A dead cell's term is 0 (its forward
        // value is), and a zero never beats the running maximum: no range test.
        double sum = 0., max = 0.;
        int best = -1;                                    // p << 2 | state of the maximum
        // Row 1 in fr (at i = 2, or at i = 1 for a read of one base) is divided by its sum once, in place -- as the forward pass
        // did -- and from then on counts as scaled.
        if (!FAST && i <= 2 && !fr_scaled) {
            const double s1 = i == 1 ? sc_i : sc_n;
            #pragma unroll
            for (int p = 1; p <= W; ++p) { fr0[p] /= s1; fr1[p] /= s1; }
            fr_scaled = true;
        }
        if (i & 1) {
            // an odd row: its own stored values, scaled as the forward pass scaled them
            const double r_o = i == 1 ? 1. : r_i;
            #pragma unroll
            for (int p = 1; p <= W; ++p) {
                const double2 t = fr(p);
                const double fM = t.x * r_o, fI = t.y * r_o;
                double z;
                z = fM * M[p]; if (z > max) { max = z; best = p << 2 | 0; } sum += z;
                z = fI * I[p]; if (z > max) { max = z; best = p << 2 | 1; } sum += z;
            }
            // the stored row the next two posteriors work from: row i - 2 (row i - 1 is re-formed from it, then it is its own)
#ifdef BAQ_EXP_NOLOAD       // experiment: the forward rows are not read back (never true at run time): the backward pass on stale values
            if (P.n_jobs < 0)
#endif
            if (i >= 3) fr_request(i - 2);
        } else {
            // an even row: re-formed cell by cell from the stored row below (fr = row i - 1, unscaled) exactly as the forward pass
            // formed it -- the row below scaled by its own factor (row 1: divided, above), its D by the recurrence over the unscaled
            // M', the three-term sums in the forward pass's order, then this row's scale
            const double qli = (double)lq2p[q_lo];        // row i's own base and quality
            const uint32_t qyi = (uint32_t)nt16_to_4((int)s_lo);
            const double ex = qyi > 3 ? 1. : 1. - qli, ey = qyi > 3 ? 1. : qli * EM;
            if (FAST) {
                const double r_b = 1. / sc_n;             // the row below's scale: 1 / SC(i - 1)
                const double2 t1 = fr(1);
                double gM = t1.x * r_b, gI = t1.y * r_b, dp = 0., mp = 0., mc = t1.x;     // the row below at p, scaled; D'[p - 1], M'[p - 1], M'[p] unscaled
                #pragma unroll
                for (int p = 1; p <= W; ++p) {
                    const double dc = m2u * mp + m8 * dp;                                // D'[p]
                    const double gD = dc * r_b;
                    const double2 tn = fr(p + 1);
                    const double uM = tn.x * r_b, uI = tn.y * r_b;
                    const double e = wbyte(p) == qyi ? ex : ey;
                    const double f0 = e * (m0u * gM + m3u * gI + m6 * gD);
                    const double f1 = m1qu * uM + m4qu * uI;
                    const double fM = f0 * r_i, fI = f1 * r_i;
                    double z;
                    z = fM * M[p]; if (z > max) { max = z; best = p << 2 | 0; } sum += z;
                    z = fI * I[p]; if (z > max) { max = z; best = p << 2 | 1; } sum += z;
                    dp = dc; gM = uM; gI = uI; mp = mc; mc = tn.x;
                }
            } else {
                const bool b1 = i == 2;                                       // the row below is row 1: already divided by its sum, D = 0
                const double r_b = b1 ? 1. : 1. / sc_n;
                const int hib = l_ref - (i - 1) + bw + 1 < top ? l_ref - (i - 1) + bw + 1 : top;     // the row below's live cells end here
                const int hi = l_ref - i + bw + 1 < top ? l_ref - i + bw + 1 : top;
                double dp = 0.;
                #pragma unroll
                for (int p = 1; p <= W; ++p) {
                    const double2 tp = fr(p - 1), tc = fr(p), tn = fr(p + 1);
                    const double dc = (!b1 && p <= hib) ? m2 * tp.x + m8 * dp : 0.;
                    const uint32_t rb = wbyte(p);
                    const double e = rb > 3 ? 1. : rb == qyi ? ex : ey;
                    const double gM = tc.x * r_b, gI = tc.y * r_b, gD = dc * r_b;
                    const double uM = tn.x * r_b, uI = tn.y * r_b;
                    const double f0 = e * (m0 * gM + m3 * gI + m6 * gD);
                    const double f1 = m1q * uM + m4q * uI;
                    const double fM = p <= hi ? f0 * r_i : 0., fI = p <= hi ? f1 * r_i : 0.;
                    double z;
                    z = fM * M[p]; if (z > max) { max = z; best = p << 2 | 0; } sum += z;
                    z = fI * I[p]; if (z > max) { max = z; best = p << 2 | 1; } sum += z;
                    dp = dc;
                }
            }
        }
        max /= sum;
        (stw + (size_t)(i - 1) * 64)[lane] = best < 0 ? -1 : ((i - bw - 2) << 2) + best;      // (k - 1) << 2 | state, k = p + i - bw - 1
        const int kq = (int)(-4.343 * log(1. - max) + .499);
        (qw + (size_t)(i - 1) * 64)[lane] = (uint8_t)(kq > 100 ? 99 : kq);
        q_hi = q_lo; s_hi = s_lo; q_lo = rq_q; s_lo = rq_s; r_in = rq_r;
        sc_i = sc_n; sc_n = rq_sc;
    };
    {
        // the longest read of the wavefront (the lanes that run: the others have returned)
        int i = 0;
        {
            bool todo = true;
            for (;;) {
                const uint64_t m = __builtin_amdgcn_ballot_w64(todo);
                if (m == 0) break;
                const int v = __builtin_amdgcn_readlane(l_query, (int)__builtin_ctzll(m));
                i = v > i ? v : i;
                if (l_query <= v) todo = false;
            }
        }
        // A fast iteration: every lane past its own last row, the step's and the row's cells all inside the reference, not the rows 2
        // and 1 (row 1 was scaled by a division and has no D), no N in the window before or after it moves.
        auto fast_at = [&](int r) { return r < l_query && l_query == lq_u && r > 2 && bw == BW && r + bw + 1 <= l_ref && !any_n() && r_in < 4; };
        #pragma unroll 1
        for (; i >= 1; --i) {
            if (__builtin_amdgcn_ballot_w64(!fast_at(i)) == 0) break;
            if (i <= l_query) bwd_row(i, Edge());
        }
        for (; i >= 1; --i) {
            if (__builtin_amdgcn_ballot_w64(!fast_at(i)) != 0) break;
            bwd_row(i, Fast());
        }
        #pragma unroll 1
        for (; i >= 1; --i)
            if (i <= l_query) bwd_row(i, Edge());
    }
    #undef FR2
    #undef SC
}

// the quality cap of sam_prob_realn (realn.c) from the posterior states and qualities.
// Per aligned block of the CIGAR the reference computes bq[i] = (state agrees with the alignment ? q[i] : 0) and, extended,
// replaces it by min(running maximum from the left, running maximum from the right) in five passes over three byte arrays;
// here: one ascending pass that leaves the left maxima in the scratch array and one descending pass that forms the right
// maximum, the minimum, the ZQ byte and the new quality, four bases per trip with all loads of a trip issued before the
// first is used (a lane's bytes are its own 100-byte stretch: every access is a round trip to L2, and one wait per byte
// made this phase a third of the kernel).
__device__ void baq_cap(const BaqParams &P, const BaqJob &j, const uint8_t *iqual, const int32_t *state_, const uint8_t *q_, uint8_t *left_, const int pst,
                        uint8_t *qout, uint8_t *zout)
{
    // state / q / left of base i at [i * pst]: pst = 64 for a wavefront's reads side by side (the lanes' loads of one base are one
    // stretch of memory), 1 for a read's own arrays (the scratch class)
    struct { const int32_t *p; int st; __device__ int operator[](int i) const { return p[(size_t)i * st]; } } state{state_, pst};
    struct { const uint8_t *p; int st; __device__ uint32_t operator[](int i) const { return p[(size_t)i * st]; } } q{q_, pst};
    struct { uint8_t *p; int st; __device__ uint8_t &operator[](int i) const { return p[(size_t)i * st]; } } left{left_, pst};
    const uint32_t *cigar = P.cig + j.cig_off;
    const bool apply = (P.flag & 1) != 0, extend = (P.flag & 2) != 0;
    int x = j.pos, y = 0;
    for (int k = 0; k < j.n_cigar; ++k) {
        const int op = cigar[k] & 0xf, l = (int)(cigar[k] >> 4);
        if (op == 0 || op == 7 || op == 8) {
            const int exp0 = x - j.xb - y;                      // the state an aligned base i of this block must have: exp0 + i
            auto bqv = [&](int st, uint32_t qq, int i) -> uint32_t { return ((st & 3) != 0 || (st >> 2) != exp0 + i) ? 0u : qq; };
            if (extend) {
                uint32_t L = 0;
                int i = y;
                for (; i + 4 <= y + l; i += 4) {
                    const int s0 = state[i], s1 = state[i + 1], s2 = state[i + 2], s3 = state[i + 3];
                    const uint32_t q0 = q[i], q1 = q[i + 1], q2 = q[i + 2], q3 = q[i + 3];
                    L = max(L, bqv(s0, q0, i)); const uint32_t l0 = L;
                    L = max(L, bqv(s1, q1, i + 1)); const uint32_t l1 = L;
                    L = max(L, bqv(s2, q2, i + 2)); const uint32_t l2 = L;
                    L = max(L, bqv(s3, q3, i + 3));
                    left[i] = (uint8_t)l0; left[i + 1] = (uint8_t)l1; left[i + 2] = (uint8_t)l2; left[i + 3] = (uint8_t)L;
                }
                for (; i < y + l; ++i) { L = max(L, bqv(state[i], q[i], i)); left[i] = (uint8_t)L; }
            }
            uint32_t R = 0;
            auto fin = [&](int i, int st, uint32_t qq, uint32_t lf, uint32_t iq) {
                uint32_t v = bqv(st, qq, i), z;
                if (extend) { R = max(R, v); v = lf < R ? lf : R; z = iq <= v ? 0u : iq - v; }
                else { v = iq < v ? iq : v; if (((st & 3) != 0 || (st >> 2) != exp0 + i)) v = 0; z = iq - v; }
                zout[i] = (uint8_t)(64 + z);
                qout[i] = apply ? (uint8_t)(iq - z) : (uint8_t)iq;
            };
            int i = y + l - 1;
            for (; i - 3 >= y; i -= 4) {
                const int s0 = state[i], s1 = state[i - 1], s2 = state[i - 2], s3 = state[i - 3];
                const uint32_t q0 = q[i], q1 = q[i - 1], q2 = q[i - 2], q3 = q[i - 3];
                const uint32_t i0 = iqual[i], i1 = iqual[i - 1], i2 = iqual[i - 2], i3 = iqual[i - 3];
                uint32_t l0 = 0, l1 = 0, l2 = 0, l3 = 0;
                if (extend) { l0 = left[i]; l1 = left[i - 1]; l2 = left[i - 2]; l3 = left[i - 3]; }
                fin(i, s0, q0, l0, i0); fin(i - 1, s1, q1, l1, i1); fin(i - 2, s2, q2, l2, i2); fin(i - 3, s3, q3, l3, i3);
            }
            for (; i >= y; --i) fin(i, state[i], q[i], extend ? left[i] : 0u, iqual[i]);
            x += l; y += l;
        } else if (op == 4 || op == 1) {
            for (int i = y; i < y + l && i < j.l_query; ++i) { zout[i] = 64; qout[i] = iqual[i]; }     // bases outside the aligned blocks keep their quality
            y += l;
        } else if (op == 2) x += l;
    }
    for (int i = y; i < j.l_query; ++i) { zout[i] = 64; qout[i] = iqual[i]; }      // (a CIGAR shorter than the query)
}

// band classes: bands of up to 7 (a read without indels: 7) and of 8 (realn.c trims the window to the read's length + 7 or
// + 8: a deletion of even length gives 8) keep their rows in registers; wider ones (indels of 8 and more) go through scratch
#define BAQ_BWM 7
#define BAQ_BWM2 8
#define BAQ_BWM3 16         // bands of 9 .. 16 (reads with an indel of 6 .. 13 bases: most of the reads past band 8) are a class of their own:
#define BAQ_MID_LANES 32    // the LDS rows of such a band are short enough for 32 reads a wavefront (the wider ones: BAQ_WIDE_LANES).  (A
                            // register row of 33 cells does not fit: 198 registers for the forward row alone, 1 600 spilled.)
#define BAQ_WIDE_LANES 8

// Wavefronts per SIMD the compiler is to leave room for: one for the register-row classes -- the two rows of the backward pass, the
// stored forward row and a cell's operands are 330 - 390 registers (with two wavefronts and 256 the row loops spill: 1.2x - 2x slower).
#ifndef BAQ_WAVES
#define BAQ_WAVES(bwm) ((bwm) > 0 ? 1 : 2)     // (bwm <= 0: the LDS-row classes)
#endif
template <int BWM>          // 0: any band, the rows in LDS (or both matrices in scratch), BAQ_WIDE_LANES reads a wavefront; < 0: the same with -BWM reads
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BAQ_WAVES(BWM), 8))) void baq_kernel(const BaqParams P)
{
    __shared__ float s_q2p[256];
    for (int t = threadIdx.x; t < 256; t += 64) s_q2p[t] = P.q2p[t];
    __syncthreads();
    // The scratch class (BWM = 0: reads with an indel of eight bases or more, a few in a thousand) is a handful of wavefronts with a
    // long way to go each -- every cell a round trip to the scratch rows -- so a wavefront takes BAQ_WIDE_LANES reads, not 64:
    // eight times the wavefronts in flight for the same reads (64 reads a wavefront: 65 ms for 17 000 reads beside 46 ms for the
    // other 4.9 million).
    constexpr int LANES = BWM > 0 ? 64 : BWM == 0 ? BAQ_WIDE_LANES : -BWM;
    if (threadIdx.x >= LANES) return;
    const int job = blockIdx.x * LANES + threadIdx.x;
    if (job >= P.n_jobs) return;
    const BaqJob j = P.jobs[job];
    const uint8_t *seq = P.seq16 + j.seq_off, *iqual = P.qual + j.seq_off;
    uint8_t *qout = P.qual_out + j.seq_off, *zout = P.zq_out + j.seq_off;
    if (j.ret < 0 || j.l_ref <= 0 || j.l_query <= 0) {
        for (int i = 0; i < j.l_query; ++i) { qout[i] = iqual[i]; zout[i] = 0; }
        return;
    }
    const uint8_t *ref = P.tref + j.ref_off;
    // the register-row classes keep state / posterior quality / left maxima of a wavefront's reads side by side
    const size_t wv = ((size_t)(job >> 6) * (size_t)P.max_lq) * 64 + (size_t)(job & 63);
    int32_t *state = BWM > 0 ? P.wstate + wv : P.state + j.seq_off;
    uint8_t *q = BWM > 0 ? P.wq + wv : P.q + j.seq_off;
    uint8_t *left = BWM > 0 ? P.wleft + wv : P.tmp + 2 * (size_t)j.seq_off;
    const int pst = BWM > 0 ? 64 : 1;
    if (BWM > 0) {
        int bw = j.l_ref > j.l_query ? j.l_ref : j.l_query;
        if (bw > j.bw) bw = j.bw;
        if (bw < abs(j.l_ref - j.l_query)) bw = abs(j.l_ref - j.l_query);
        baq_fb_reg<(BWM > 0 ? BWM : 1)>(P, job, j, ref, seq, iqual, bw, s_q2p);
    } else if (P.lds_rows && j.l_ref <= P.max_lq + P.ncell / 6 + 16) {        // (the window fits the LDS copy: always, realn.c trims it to the read and its band)
        extern __shared__ double s_rows[];                   // [3][ncell][LANES] doubles, then the window's bases [max l_ref][LANES]
        const size_t rw = (size_t)P.ncell * LANES;
        baq_fb_lds<LANES>(P, job, j, ref, seq, iqual, state, q, s_rows + threadIdx.x, s_rows + rw + threadIdx.x, s_rows + 2 * rw + threadIdx.x,
                          reinterpret_cast<uint8_t*>(s_rows + 3 * rw) + threadIdx.x);
    } else baq_fb_scratch(P, job, j, ref, seq, iqual, state, q);
#if !defined(BAQ_EXP_PHASE) || BAQ_EXP_PHASE == 0
    baq_cap(P, j, iqual, state, q, left, pst, qout, zout);
#endif
}

// ---- the same stage on the pool bcfgpu_pool_upload left in HBM: the host half above as a kernel ----
// One lane per read: the window and band of realn.c; the read's job goes to the list of its band class (class 0 = the
// register-row kernel's band, class 1 = wider), reads that BAQ leaves alone get ret = -1 and keep their qualities.
struct BaqPrepParams {
    DevPool D;
    int ref_len;                     // bases of the contig (up to its terminating NUL)
    BaqJob *jobs0, *jobs1, *jobs2, *jobs3;
    int *counts;                     // [0],[1],[7],[6] jobs per class (band <= 7, 8, <= 16, wider)  [2] widest band  [3] longest query  [4] lowest xb  [5] highest xe
    int32_t *ret; uint8_t *has_zq;
};
__global__ __launch_bounds__(256) void baq_prep_kernel(const BaqPrepParams P)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    BaqJob j{};
    int cls = -1, b = 0, lq = 0, wlo = INT32_MAX, whi = 0;      // the read's class (-1: no job), band, query length, window
    if (r < P.D.n_reads) {
        const int l_qseq = P.D.r_lq[r], pos = P.D.r_pos[r], n_cigar = P.D.r_ncig[r];
        const uint32_t soff = (uint32_t)P.D.r_seq_off[r], coff = (uint32_t)P.D.r_cig_off[r];
        const uint32_t *cigar = P.D.cig + coff;
        bool ok = !(l_qseq <= 0 || P.D.qual[soff] == 0xff || (P.D.r_flag[r] & 4));
        int x = pos, y = 0, yb = -1, ye = -1, xb = -1, xe = -1;
        for (int k = 0; ok && k < n_cigar; ++k) {
            const int op = cigar[k] & 0xf, l = (int)(cigar[k] >> 4);
            if (op == 0 || op == 7 || op == 8) {
                if (yb < 0) yb = y;
                if (xb < 0) xb = x;
                ye = y + l; xe = x + l;
                x += l; y += l;
            } else if (op == 4 || op == 1) y += l;
            else if (op == 2) x += l;
            else if (op == 3) ok = false;
        }
        if (xb == -1) ok = false;
        P.ret[r] = ok ? 0 : -1; P.has_zq[r] = ok ? 1 : 0;
        if (ok) {
            int bw = 7;
            if (abs((xe - xb) - (ye - yb)) > bw) bw = abs((xe - xb) - (ye - yb)) + 3;
            xb -= yb + bw / 2; if (xb < 0) xb = 0;
            xe += l_qseq - ye + bw / 2;
            if (xe - xb - l_qseq > bw) { xb += (xe - xb - l_qseq - bw) / 2; xe -= (xe - xb - l_qseq - bw) / 2; }
            if (xe > P.ref_len) xe = P.ref_len;
            if (xe < xb) xe = xb;
            j.seq_off = soff; j.cig_off = coff; j.l_query = l_qseq; j.n_cigar = n_cigar; j.pos = pos;
            j.l_ref = xe - xb; j.bw = bw; j.xb = xb; j.ret = 0; j.ref_off = (uint32_t)xb;      // (relative to the slice through P.tref - lowest xb)
            b = j.l_ref > j.l_query ? j.l_ref : j.l_query;
            if (b > j.bw) b = j.bw;
            if (b < abs(j.l_ref - j.l_query)) b = abs(j.l_ref - j.l_query);
            cls = b <= BAQ_BWM ? 0 : b <= BAQ_BWM2 ? 1 : b <= BAQ_BWM3 ? 3 : 2;
            lq = l_qseq;
            if (j.l_ref > 0) { wlo = xb; whi = xe; }
        }
    }
    // One atomic per WORKGROUP and class, and the maxima only when they would change a counter: every lane, or every wavefront, on
    // the same few counters queue up behind one another in L2 (3.7 ms for 4.9 M reads with one set of atomics a wavefront).
    __shared__ int s_cnt[4][4], s_base[4], s_red[4][4];
    const int wave = threadIdx.x >> 6;
    unsigned long long m[4];
    #pragma unroll
    for (int c = 0; c < 4; ++c) { m[c] = __builtin_amdgcn_ballot_w64(cls == c); if (lane == 0) s_cnt[wave][c] = (int)__popcll(m[c]); }
    int wb = cls == 2 ? b : 1;
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) { wb = max(wb, __shfl_xor(wb, o)); lq = max(lq, __shfl_xor(lq, o)); wlo = min(wlo, __shfl_xor(wlo, o)); whi = max(whi, __shfl_xor(whi, o)); }
    if (lane == 0) { s_red[wave][0] = wb; s_red[wave][1] = lq; s_red[wave][2] = wlo; s_red[wave][3] = whi; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int c = threadIdx.x, tot = s_cnt[0][c] + s_cnt[1][c] + s_cnt[2][c] + s_cnt[3][c];
        s_base[c] = tot ? atomicAdd(&P.counts[c == 2 ? 6 : c == 3 ? 7 : c], tot) : 0;
    } else if (threadIdx.x == 64) {
        int xb_ = 1, xl = 1, xlo = INT32_MAX, xhi = 0;
        for (int w = 0; w < 4; ++w) { xb_ = max(xb_, s_red[w][0]); xl = max(xl, s_red[w][1]); xlo = min(xlo, s_red[w][2]); xhi = max(xhi, s_red[w][3]); }
        volatile int *cnt = P.counts;                      // (a stale value only costs an atomic that changes nothing)
        if (xb_ > 1 && xb_ > cnt[2]) atomicMax(&P.counts[2], xb_);
        if (xl > 1 && xl > cnt[3]) atomicMax(&P.counts[3], xl);
        if (xlo != INT32_MAX) { if (xlo < cnt[4]) atomicMin(&P.counts[4], xlo); if (xhi > cnt[5]) atomicMax(&P.counts[5], xhi); }
    }
    __syncthreads();
    #pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (cls != c) continue;
        int base = s_base[c];
        for (int w = 0; w < wave; ++w) base += s_cnt[w][c];
        (c == 0 ? P.jobs0 : c == 1 ? P.jobs1 : c == 2 ? P.jobs2 : P.jobs3)[base + (int)__popcll(m[c] & ((1ull << lane) - 1))] = j;
    }
}

// The wide-band class's jobs in descending order of their band (a counting sort, one workgroup: the class is a few reads in a
// thousand).  A wavefront of the class takes BAQ_WIDE_LANES reads and sweeps to the widest band among them: in arrival order
// nearly every wavefront had one of the widest (a pool with indels of 8 - 40 bases: mean band 34, mean of a wavefront's maximum
// 70), and the widest, which run longest, start first.
__global__ __launch_bounds__(1024) void baq_sort_wide_kernel(const BaqJob *in, int n, BaqJob *out)
{
    __shared__ int s_hist[1024], s_part[1024];
    auto key = [](const BaqJob &j) {
        int b = j.l_ref > j.l_query ? j.l_ref : j.l_query;
        if (b > j.bw) b = j.bw;
        if (b < abs(j.l_ref - j.l_query)) b = abs(j.l_ref - j.l_query);
        return 1023 - (b > 1023 ? 1023 : b);               // (ascending key = descending band)
    };
    const int tid = threadIdx.x;
    s_hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) atomicAdd(&s_hist[key(in[i])], 1);
    __syncthreads();
    // exclusive prefix of the 1024 counts
    const int v = s_hist[tid];
    s_part[tid] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int t = tid >= o ? s_part[tid - o] : 0;
        __syncthreads();
        s_part[tid] += t;
        __syncthreads();
    }
    s_hist[tid] = s_part[tid] - v;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) { const BaqJob j = in[i]; out[atomicAdd(&s_hist[key(j)], 1)] = j; }
}

__global__ __launch_bounds__(256) void baq_ref4_kernel(const char *ref, size_t n, uint8_t *out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const char c = ref[i];
    out[i] = (c == 'A' || c == 'a') ? 0 : (c == 'C' || c == 'c') ? 1 : (c == 'G' || c == 'g') ? 2 : (c == 'T' || c == 't') ? 3 : 4;
}

}  // namespace bcfgpu

using namespace bcfgpu;

#define BQ_CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { cleanup(); return bcfgpu_set_error(BCFGPU_E_HIP, #call); } } while (0)

// the scratch class: BAQ_WIDE_LANES reads a wavefront; the working rows in LDS when two rows of the class's widest band fit (bands
// to about 200), else every cell through the scratch matrices
template <int BWM>          // 0: the wide class, -BAQ_MID_LANES: bands up to BAQ_BWM3
static void launch_baq_wide(BaqParams P, hipStream_t st)
{
    constexpr int LANES = BWM == 0 ? BAQ_WIDE_LANES : -BWM;
    // three rows of the class's widest band (two working rows, the forward row of the posterior) and the window's bases: a window is
    // at most the read and its band long (realn.c trims it to that)
    const size_t lds = 3 * (size_t)P.ncell * LANES * sizeof(double) + ((size_t)P.max_lq + (size_t)P.ncell / 6 + 16) * LANES;
    P.lds_rows = lds <= 150 * 1024 ? 1 : 0;
    if (P.lds_rows && lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(baq_kernel<BWM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(baq_kernel<BWM>, dim3((P.n_jobs + LANES - 1) / LANES), dim3(64), P.lds_rows ? lds : 0, st, P);
}

extern "C" int bcfgpu_baq(bcfgpu_ctx *ctx, const bcfgpu_reads *rd, const char *ref, int32_t ref_len, int flag,
                          uint8_t *qual_out, uint8_t *zq_out, int32_t *ret)
{
    if (!ctx || !rd || !ref || !qual_out || !zq_out || !ret || rd->n_reads < 0)
        return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_baq: bad arguments");
    hipStream_t stream = nullptr;
    const float *d_q2p = nullptr;
    if (bcfgpu_internal_device(ctx, &stream, &d_q2p)) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_baq: bad context");
    const int n = rd->n_reads;
    if (n == 0) return BCFGPU_OK;
    static const uint8_t nt4[256] = {
#define N4 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4
        N4,N4,N4,N4,
        4,0,4,1,4,4,4,2,4,4,4,4,4,4,4,4, 4,4,4,4,3,4,4,4,4,4,4,4,4,4,4,4,
        4,0,4,1,4,4,4,2,4,4,4,4,4,4,4,4, 4,4,4,4,3,4,4,4,4,4,4,4,4,4,4,4,
        N4,N4,N4,N4,N4,N4,N4,N4
#undef N4
    };
    // ---- host half: window and band of every read (realn.c), 0..4 codes of the window ----
    std::vector<BaqJob> jobs(n);
    std::vector<uint8_t> tref;
    size_t nbase = 0, ncig = 0;
    int max_lq = 1;
    for (int r = 0; r < n; ++r) {
        BaqJob &j = jobs[r];
        const int l_qseq = rd->r_lq[r], pos = rd->r_pos[r];
        const uint32_t *cigar = rd->cig + rd->r_cig_off[r];
        j.seq_off = (uint32_t)rd->r_seq_off[r]; j.cig_off = (uint32_t)rd->r_cig_off[r];
        j.l_query = l_qseq; j.n_cigar = rd->r_ncig[r]; j.pos = pos; j.ret = -1; j.l_ref = 0; j.bw = 0; j.xb = 0; j.ref_off = 0;
        if ((size_t)j.seq_off + l_qseq > nbase) nbase = (size_t)j.seq_off + l_qseq;
        if ((size_t)j.cig_off + j.n_cigar > ncig) ncig = (size_t)j.cig_off + j.n_cigar;
        if (l_qseq > max_lq) max_lq = l_qseq;
        ret[r] = -1;
        if (l_qseq == 0 || rd->qual[j.seq_off] == 0xff || (rd->r_flag[r] & 4)) continue;
        int x = pos, y = 0, yb = -1, ye = -1, xb = -1, xe = -1;
        bool has_n = false;
        for (int k = 0; k < j.n_cigar; ++k) {
            const int op = cigar[k] & 0xf, l = (int)(cigar[k] >> 4);
            if (op == 0 || op == 7 || op == 8) {
                if (yb < 0) yb = y;
                if (xb < 0) xb = x;
                ye = y + l; xe = x + l;
                x += l; y += l;
            } else if (op == 4 || op == 1) y += l;
            else if (op == 2) x += l;
            else if (op == 3) { has_n = true; break; }
        }
        if (has_n || xb == -1) continue;
        int bw = 7;
        if (std::abs((xe - xb) - (ye - yb)) > bw) bw = std::abs((xe - xb) - (ye - yb)) + 3;
        xb -= yb + bw / 2; if (xb < 0) xb = 0;
        xe += l_qseq - ye + bw / 2;
        if (xe - xb - l_qseq > bw) { xb += (xe - xb - l_qseq - bw) / 2; xe -= (xe - xb - l_qseq - bw) / 2; }
        j.ref_off = (uint32_t)tref.size();
        int k;
        for (k = xb; k < xe && k < ref_len && ref[k]; ++k) tref.push_back(nt4[(uint8_t)ref[k]]);
        xe = k;
        j.l_ref = xe - xb; j.bw = bw; j.xb = xb; j.ret = 0;
        ret[r] = 0;
    }
    if (tref.size() >> 32) return bcfgpu_set_error(BCFGPU_E_RANGE, "bcfgpu_baq: pool too large, use fewer reads per call");
    // band classes: class 0 = bands that fit the register-row kernel (the default band of 7), class 1 = wide bands
    // (long indels), run separately with both matrices in scratch and their own row width
    auto eff_bw = [](const BaqJob &j) { int b = j.l_ref > j.l_query ? j.l_ref : j.l_query; if (b > j.bw) b = j.bw;
                                        if (b < std::abs(j.l_ref - j.l_query)) b = std::abs(j.l_ref - j.l_query); return b; };
#ifdef BCFGPU_DIAG
    const bool force_scratch = [] { const char *ab = getenv("BCFGPU_ABLATE"); return ab && (atoi(ab) & 512); }();
#else
    const bool force_scratch = false;
#endif
    std::vector<BaqJob> cls[4];
    int cls_bw[4] = {1, 1, 1, 1};
    for (const BaqJob &j : jobs) {
        const int b = j.ret < 0 ? 1 : eff_bw(j), c = force_scratch ? 2 : b <= BAQ_BWM ? 0 : b <= BAQ_BWM2 ? 1 : b <= BAQ_BWM3 ? 3 : 2;
        cls[c].push_back(j);
        if (b > cls_bw[c]) cls_bw[c] = b;
    }

    // ---- device half ----
    // grow-only workspaces of the context (slots 7..): GiB-sized scratch is not reallocated per call
    void *d_jobs = nullptr, *d_F = nullptr, *d_B = nullptr, *d_S = nullptr;
    auto cleanup = [&]() {};
    void *d_tref = bcfgpu_internal_ws(ctx, 7, tref.size() + 16), *d_seq = bcfgpu_internal_ws(ctx, 8, nbase + 16),
         *d_qual = bcfgpu_internal_ws(ctx, 9, nbase + 16), *d_cig = bcfgpu_internal_ws(ctx, 10, (ncig + 4) * 4),
         *d_state = bcfgpu_internal_ws(ctx, 11, (nbase + 4) * 4), *d_q = bcfgpu_internal_ws(ctx, 12, nbase + 16),
         *d_tmp = bcfgpu_internal_ws(ctx, 13, 2 * nbase + 16), *d_qo = bcfgpu_internal_ws(ctx, 14, nbase + 16),
         *d_zo = bcfgpu_internal_ws(ctx, 15, nbase + 16);
    if (!d_tref || !d_seq || !d_qual || !d_cig || !d_state || !d_q || !d_tmp || !d_qo || !d_zo)
        return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_baq: device workspace");
    BQ_CHK(hipMemcpyAsync(d_tref, tref.data(), tref.size(), hipMemcpyHostToDevice, stream));
    BQ_CHK(hipMemcpyAsync(d_seq, rd->seq16, nbase, hipMemcpyHostToDevice, stream));
    BQ_CHK(hipMemcpyAsync(d_qual, rd->qual, nbase, hipMemcpyHostToDevice, stream));
    BQ_CHK(hipMemcpyAsync(d_cig, rd->cig, ncig * 4, hipMemcpyHostToDevice, stream));
    BQ_CHK(hipMemsetAsync(d_zo, 0, nbase, stream));
    BQ_CHK(hipMemcpyAsync(d_qo, rd->qual, nbase, hipMemcpyHostToDevice, stream));      // bases no job covers keep their quality
    BaqParams P{};
    P.flag = flag; P.max_lq = max_lq;
    P.tref = (const uint8_t*)d_tref; P.seq16 = (const uint8_t*)d_seq; P.qual = (const uint8_t*)d_qual; P.cig = (const uint32_t*)d_cig;
    P.q2p = d_q2p; P.state = (int32_t*)d_state; P.q = (uint8_t*)d_q; P.tmp = (uint8_t*)d_tmp;
    P.qual_out = (uint8_t*)d_qo; P.zq_out = (uint8_t*)d_zo;
    // (the wide class in descending order of the band, as baq_sort_wide_kernel leaves it in the pool form)
    std::stable_sort(cls[2].begin(), cls[2].end(), [&](const BaqJob &x, const BaqJob &y) { return eff_bw(x) > eff_bw(y); });
    for (int c = 0; c < 4; ++c) {
        const size_t nj = cls[c].size();
        if (!nj) continue;
        const bool reg = c < 2;
        P.ncell = reg ? 2 * (2 * (c ? BAQ_BWM2 : BAQ_BWM) + 3) : 3 * (2 * cls_bw[c] + 1) + 6;       // doubles per matrix row
        const size_t per_mat = (size_t)(reg ? BAQ_ROWS_KEPT(max_lq) : max_lq + 2) * P.ncell * sizeof(double);    // one matrix of one read (register-row classes: the odd rows)
        const size_t per_job = reg ? per_mat : 2 * per_mat;
        size_t chunk = ((size_t)2 << 30) / per_job;
        chunk = chunk < 64 ? 64 : (chunk & ~(size_t)63);
        if (chunk > nj) chunk = (nj + 63) & ~(size_t)63;
        P.stride = chunk;
        // (slots 0..6 are shared with the indel stage; both stages finish their stream work before returning)
        d_jobs = bcfgpu_internal_ws(ctx, 0, nj * sizeof(BaqJob));
        d_F = bcfgpu_internal_ws(ctx, 4, per_mat * chunk);
        d_B = reg ? nullptr : bcfgpu_internal_ws(ctx, 1, per_mat * chunk);
        d_S = bcfgpu_internal_ws(ctx, 2, (size_t)(max_lq + 2) * chunk * sizeof(double));
        if (!d_jobs || !d_F || (!reg && !d_B) || !d_S) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_baq: device workspace");
        if (reg) {                                                   // state / posterior quality / left maxima, a wavefront's reads side by side
            const size_t wn = chunk * (size_t)max_lq;
            uint8_t *d_w = (uint8_t*)bcfgpu_internal_ws(ctx, 5, wn * 6 + 64);
            if (!d_w) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_baq: device workspace");
            P.wstate = (int32_t*)d_w; P.wq = d_w + wn * 4; P.wleft = d_w + wn * 5;
        }
        BQ_CHK(hipMemcpyAsync(d_jobs, cls[c].data(), nj * sizeof(BaqJob), hipMemcpyHostToDevice, stream));
        P.F = (double*)d_F; P.B = (double*)d_B; P.S = (double*)d_S;
        for (size_t j0 = 0; j0 < nj; j0 += chunk) {
            P.n_jobs = (int)(nj - j0 < chunk ? nj - j0 : chunk);
            P.jobs = (const BaqJob*)d_jobs + j0;
            if (c == 0)      hipLaunchKernelGGL(baq_kernel<BAQ_BWM>, dim3((P.n_jobs + 63) / 64), dim3(64), 0, stream, P);
            else if (c == 1) hipLaunchKernelGGL(baq_kernel<BAQ_BWM2>, dim3((P.n_jobs + 63) / 64), dim3(64), 0, stream, P);
            else if (c == 3) launch_baq_wide<-BAQ_MID_LANES>(P, stream);
            else             launch_baq_wide<0>(P, stream);
        }
        BQ_CHK(hipGetLastError());
    }
    BQ_CHK(hipMemcpyAsync(qual_out, d_qo, nbase, hipMemcpyDeviceToHost, stream));
    BQ_CHK(hipMemcpyAsync(zq_out, d_zo, nbase, hipMemcpyDeviceToHost, stream));
    BQ_CHK(hipStreamSynchronize(stream));
    return BCFGPU_OK;
}

extern "C" void *bcfgpu_internal_pool_state(bcfgpu_ctx *ctx);

// sam_prob_realn on every read of the pool in HBM: the pool's qualities become the new ones, the ZQ bytes stay there for
// bcfgpu_gap_prep_tile.  The host part: the contig's length, one wait for the job counts (they size the scratch and the
// reference slice to upload), launches.
extern "C" int bcfgpu_pool_baq(bcfgpu_ctx *ctx, const char *ref, int32_t ref_len, int flag, int32_t *ret)
{
    if (!ctx || !ref || ref_len < 0) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pool_baq: bad arguments");
    hipStream_t stream = nullptr;
    const float *d_q2p = nullptr;
    if (bcfgpu_internal_device(ctx, &stream, &d_q2p)) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pool_baq: bad context");
    DevPool &D = *static_cast<DevPool*>(bcfgpu_internal_pool_state(ctx));
    if (!D.valid) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pool_baq: no read pool on this context (bcfgpu_pool_upload)");
    const int n = D.n_reads;
    if (n == 0) return BCFGPU_OK;
    auto cleanup = [&]() {};
    const size_t nbase = D.n_bases;
    BaqPrepParams Q{};
    Q.D = D; Q.ref_len = (int)strnlen(ref, (size_t)ref_len);
    Q.jobs0 = (BaqJob*)bcfgpu_internal_ws(ctx, 0, (size_t)n * sizeof(BaqJob) + 64);
    Q.jobs1 = (BaqJob*)bcfgpu_internal_ws(ctx, 3, (size_t)n * sizeof(BaqJob) + 64);
    Q.jobs2 = (BaqJob*)bcfgpu_internal_ws(ctx, 127, (size_t)n * sizeof(BaqJob) + 64);
    Q.jobs3 = (BaqJob*)bcfgpu_internal_ws(ctx, 135, (size_t)n * sizeof(BaqJob) + 64);
    Q.counts = (int*)bcfgpu_internal_ws(ctx, 115, 64);
    Q.ret = (int32_t*)bcfgpu_internal_ws(ctx, 116, (size_t)n * 4 + 64);
    Q.has_zq = (uint8_t*)bcfgpu_internal_ws(ctx, 117, (size_t)n + 64);
    const int qo_slot = D.qual_slot == 118 ? 119 : 118;                 // not the buffer the pool's qualities are in now
    uint8_t *d_qo = (uint8_t*)bcfgpu_internal_ws(ctx, qo_slot, nbase + 64);
    uint8_t *d_zo = (uint8_t*)bcfgpu_internal_ws(ctx, 120, nbase + 64);
    void *d_state = bcfgpu_internal_ws(ctx, 11, (nbase + 4) * 4), *d_q = bcfgpu_internal_ws(ctx, 12, nbase + 16), *d_tmp = bcfgpu_internal_ws(ctx, 13, 2 * nbase + 16);
    if (!Q.jobs0 || !Q.jobs1 || !Q.jobs2 || !Q.jobs3 || !Q.counts || !Q.ret || !Q.has_zq || !d_qo || !d_zo || !d_state || !d_q || !d_tmp)
        return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_pool_baq: device workspace");
    const int init[8] = {0, 0, 1, 1, INT32_MAX, 0, 0, 0};
    int counts[8];
    BQ_CHK(hipMemcpyAsync(Q.counts, init, sizeof init, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(baq_prep_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, Q);
    BQ_CHK(hipMemcpyAsync(counts, Q.counts, sizeof counts, hipMemcpyDeviceToHost, stream));
    BQ_CHK(hipMemcpyAsync(d_qo, D.qual, nbase, hipMemcpyDeviceToDevice, stream));     // reads without a job keep their qualities
    BQ_CHK(hipMemsetAsync(d_zo, 0, nbase, stream));
    BQ_CHK(hipStreamSynchronize(stream));
    // the reference slice the windows touch, as 0..4 codes
    const int lo = counts[4] == INT32_MAX ? 0 : counts[4], hi = std::max(counts[5], lo);
    char *d_refc = (char*)bcfgpu_internal_ws(ctx, 121, (size_t)(hi - lo) + 64);
    uint8_t *d_ref4 = (uint8_t*)bcfgpu_internal_ws(ctx, 7, (size_t)(hi - lo) + 64);
    if (!d_refc || !d_ref4) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_pool_baq: device workspace");
    if (hi > lo) {
        // (through the context's page-locked staging buffer: the call returns with the copy in flight, `ref` is the caller's)
        char *h_ref = (char*)bcfgpu_internal_pinned(ctx, 6, (size_t)(hi - lo));
        if (!h_ref) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_pool_baq: host staging");
        std::memcpy(h_ref, ref + lo, (size_t)(hi - lo));
        BQ_CHK(hipMemcpyAsync(d_refc, h_ref, (size_t)(hi - lo), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(baq_ref4_kernel, dim3((unsigned)(((size_t)(hi - lo) + 255) / 256)), dim3(256), 0, stream, d_refc, (size_t)(hi - lo), d_ref4);
    }
    BaqParams P{};
    P.flag = flag; P.max_lq = counts[3];
    P.tref = d_ref4 - lo;                      // (jobs carry absolute window starts)
    P.seq16 = D.seq16; P.qual = D.qual; P.cig = D.cig;
    P.q2p = d_q2p; P.state = (int32_t*)d_state; P.q = (uint8_t*)d_q; P.tmp = (uint8_t*)d_tmp;
    P.qual_out = d_qo; P.zq_out = d_zo;
    // The four band classes are independent (their reads are disjoint): the three small ones (band 8: a few per cent of the reads;
    // bands to 16 and wider ones: a handful, whose workgroups share the compute units' LDS, so the second of them starts when the
    // first has drained -- 23 ms together beside the main class's 39; a raised wave priority for the main class changed nothing)
    // go to side streams with row buffers of their own and are launched FIRST, so that their few hundred
    // wavefronts run beside the main class instead of after it -- a launch of 178 wavefronts takes as long as its slowest
    // wavefront, half a millisecond with the chip otherwise idle.
    hipStream_t *side = nullptr; hipEvent_t *sev = nullptr;
    if (bcfgpu_internal_side(ctx, &side, &sev)) return bcfgpu_set_error(BCFGPU_E_HIP, "bcfgpu_pool_baq: side streams");
    BaqJob *jobs2 = Q.jobs2;                               // the wide class's jobs, sorted by band (before anything else is on the chip:
    if (counts[6] > 0) {                                   // one workgroup, 20 us alone, 15 ms behind a launch that fills it)
        jobs2 = (BaqJob*)bcfgpu_internal_ws(ctx, 128, (size_t)counts[6] * sizeof(BaqJob) + 64);
        if (!jobs2) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_pool_baq: device workspace");
        hipLaunchKernelGGL(baq_sort_wide_kernel, dim3(1), dim3(1024), 0, stream, Q.jobs2, counts[6], jobs2);
    }
    BQ_CHK(hipEventRecord(sev[0], stream));
    static const int slotF[4] = { 4, 144, 147, 149 }, slotS[4] = { 2, 145, 148, 150 }, slotW[4] = { 5, 146, 1, 151 };    // (W of the LDS-row classes: their second matrix)
    bool used_side[4] = { false, false, false, false };
    for (int ci = 0; ci < 4; ++ci) {
        const int c = ci == 0 ? 2 : ci == 1 ? 3 : ci == 2 ? 1 : 0;                // the small classes first, the slowest of them in front
        const size_t nj = (size_t)counts[c == 2 ? 6 : c == 3 ? 7 : c];
        if (!nj) continue;
        const bool reg = c < 2;
        hipStream_t st = c == 0 ? stream : side[c - 1];
        if (c) { BQ_CHK(hipStreamWaitEvent(st, sev[0], 0)); used_side[c] = true; }
        P.ncell = reg ? 2 * (2 * (c ? BAQ_BWM2 : BAQ_BWM) + 3) : 3 * (2 * (c == 3 ? BAQ_BWM3 : counts[2]) + 1) + 6;       // doubles per matrix row
        const size_t per_mat = (size_t)(reg ? BAQ_ROWS_KEPT(P.max_lq) : P.max_lq + 2) * P.ncell * sizeof(double);  // one matrix of one read (register-row classes: the odd rows)
        const size_t per_job = reg ? per_mat : 2 * per_mat;
        size_t chunk = ((size_t)24 << 30) / per_job;                               // (up to 24 GiB of scratch of the 288 GB: one launch for ~9e5 reads of 100 bases -- every launch ends with a round of wavefronts that does not fill the chip)
        chunk = chunk < 64 ? 64 : (chunk & ~(size_t)63);
        if (chunk > nj) chunk = (nj + 63) & ~(size_t)63;
        P.stride = chunk;
        void *d_F = bcfgpu_internal_ws(ctx, slotF[c], per_mat * chunk), *d_B = reg ? nullptr : bcfgpu_internal_ws(ctx, slotW[c], per_mat * chunk);
        void *d_S = bcfgpu_internal_ws(ctx, slotS[c], (size_t)(P.max_lq + 2) * chunk * sizeof(double));
        if (!d_F || (!reg && !d_B) || !d_S) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_pool_baq: device workspace");
        if (reg) {                                                   // state / posterior quality / left maxima, a wavefront's reads side by side
            const size_t wn = chunk * (size_t)P.max_lq;
            uint8_t *d_w = (uint8_t*)bcfgpu_internal_ws(ctx, slotW[c], wn * 6 + 64);
            if (!d_w) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_pool_baq: device workspace");
            P.wstate = (int32_t*)d_w; P.wq = d_w + wn * 4; P.wleft = d_w + wn * 5;
        }
        P.F = (double*)d_F; P.B = (double*)d_B; P.S = (double*)d_S;
        for (size_t j0 = 0; j0 < nj; j0 += chunk) {
            P.n_jobs = (int)(nj - j0 < chunk ? nj - j0 : chunk);
            P.jobs = (c == 0 ? Q.jobs0 : c == 1 ? Q.jobs1 : c == 2 ? jobs2 : Q.jobs3) + j0;
            if (c == 0)      hipLaunchKernelGGL(baq_kernel<BAQ_BWM>, dim3((P.n_jobs + 63) / 64), dim3(64), 0, st, P);
            else if (c == 1) hipLaunchKernelGGL(baq_kernel<BAQ_BWM2>, dim3((P.n_jobs + 63) / 64), dim3(64), 0, st, P);
            else if (c == 3) launch_baq_wide<-BAQ_MID_LANES>(P, st);
            else             launch_baq_wide<0>(P, st);
        }
        BQ_CHK(hipGetLastError());
    }
    for (int c = 1; c < 4; ++c)
        if (used_side[c]) { BQ_CHK(hipEventRecord(sev[c], side[c - 1])); BQ_CHK(hipStreamWaitEvent(stream, sev[c], 0)); }
    if (ret) {
        BQ_CHK(hipMemcpyAsync(ret, Q.ret, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
        BQ_CHK(hipStreamSynchronize(stream));
    }
    D.qual = d_qo; D.qual_slot = qo_slot; D.zq = d_zo; D.r_has_zq = Q.has_zq;
    return BCFGPU_OK;
}


// indel.hip -- the realignment scores of bcf_call_gap_prep: probaln_glocal() forward pass for every
// (site, indel type, read) job of a batch (bam2bcf_indel.c:291-370; "this is the bottleneck", :335).
//
// One lane per job: the banded 3-state pair-HMM is a chain of dependent rows, and the jobs of a site share window
// and band, so 64 neighbouring jobs run in lockstep.  Each lane keeps two rolling rows of the scaled forward matrix
// in a scratch buffer laid out [cell][job] (coalesced across lanes).  All arithmetic is fp64 in the reference's
// order (htslib probaln.c), so the integer scores match the CPU path.
#include <hip/hip_runtime.h>
#include <math.h>
#include <type_traits>
#include <algorithm>
#include "kernels.h"

namespace bcfgpu {

#define EI .25
#define EM .33333333333

// a read's bases and qualities as the caller holds them: 4-bit codes, raw qualities, optional ZQ bytes (BAQ offsets).
// bam2bcf_indel.c:339-345: query = seq_nt16_int of the base; quality = qual (+ ZQ - 64), capped to [7, 30]
struct QSrc {
    const uint8_t *seq, *qual, *zq;
    __device__ __forceinline__ int base(int i) const { return (int)((0x4444444344424104ull >> (4 * (seq[i] & 15))) & 7); }
    __device__ __forceinline__ int q(int i) const
    {
        int v = zq ? (int)(uint8_t)(qual[i] + (zq[i] - 64)) : (int)qual[i];
        v = v > 30 ? 30 : v;
        return v < 7 ? 7 : v;
    }
};

__device__ __forceinline__ int set_u(int b, int i, int k) { int x = i - b; x = x > 0 ? x : 0; return (k - x + 1) * 3; }

// forward score of one job with gap-open d / gap-ext e (probaln_par_t {d, e, bw})
__device__ int probaln_fwd(const uint8_t *ref, int l_ref, const QSrc qs, int l_query,
                           const float *q2p, double d, double e_, int cbw, double *row0, double *row1, size_t stride, int ncell)
{
    if (l_ref <= 0 || l_query <= 0) return 0;
    int bw = l_ref > l_query ? l_ref : l_query;
    if (bw > cbw) bw = cbw;
    if (bw < abs(l_ref - l_query)) bw = abs(l_ref - l_query);
    const int bw2 = bw * 2 + 1;
    if (bw2 * 3 + 6 > ncell) return INT_MIN;              // scratch too small for this band (host sizes it)
    double m[9];
    const double sM = 1. / (2 * l_query + 2), sI = sM;
    m[0] = (1 - d - d) * (1 - sM); m[1] = m[2] = d * (1 - sM);
    m[3] = (1 - e_) * (1 - sI); m[4] = e_ * (1 - sI); m[5] = 0.;
    m[6] = 1 - e_; m[7] = 0.; m[8] = e_;
    const double bM = (1 - d) / l_ref, bI = d / l_ref;
    #define F(r, c) (r)[(size_t)(c) * stride]
    double *fi = row1, *fi1 = row0;
    const int nc = bw2 * 3 + 6;
    double p = 1., Pr1 = 0.;            // running product of the scaling factors s[i] (s[0] = 1)
    // f[1]
    for (int c = 0; c < nc; ++c) F(fi, c) = 0.;
    {
        double sum = 0.;
        const int beg = 1, end = l_ref < bw + 1 ? l_ref : bw + 1;
        const double q0 = (double)q2p[qs.q(0)];
        const int qy0 = qs.base(0);
        for (int k = beg; k <= end; ++k) {
            const double e = (ref[k - 1] > 3 || qy0 > 3) ? 1. : ref[k - 1] == qy0 ? 1. - q0 : q0 * EM;
            const int u = set_u(bw, 1, k);
            const double a = e * bM, b = EI * bI;
            F(fi, u) = a; F(fi, u + 1) = b;
            sum += a + b;
        }
        const int _beg = set_u(bw, 1, beg), _end = set_u(bw, 1, end) + 2;
        for (int k = _beg; k <= _end; ++k) F(fi, k) /= sum;
        p *= sum;
        if (p < 1e-100) { Pr1 += -4.343 * log(p); p = 1.; }
    }
    for (int i = 2; i <= l_query; ++i) {
        double *t = fi; fi = fi1; fi1 = t;
        for (int c = 0; c < nc; ++c) F(fi, c) = 0.;
        const double qli = (double)q2p[qs.q(i - 1)];
        const int qyi = qs.base(i - 1);
        int beg = 1, end = l_ref, x;
        x = i - bw; beg = beg > x ? beg : x;
        x = i + bw; end = end < x ? end : x;
        double sum = 0.;
        for (int k = beg; k <= end; ++k) {
            const double e = (ref[k - 1] > 3 || qyi > 3) ? 1. : ref[k - 1] == qyi ? 1. - qli : qli * EM;
            const int u = set_u(bw, i, k), v11 = set_u(bw, i - 1, k - 1), v10 = set_u(bw, i - 1, k), v01 = set_u(bw, i, k - 1);
            const double f0 = e * (m[0] * F(fi1, v11) + m[3] * F(fi1, v11 + 1) + m[6] * F(fi1, v11 + 2));
            const double f1 = EI * (m[1] * F(fi1, v10) + m[4] * F(fi1, v10 + 1));
            const double f2 = m[2] * F(fi, v01) + m[8] * F(fi, v01 + 2);
            F(fi, u) = f0; F(fi, u + 1) = f1; F(fi, u + 2) = f2;
            sum += f0 + f1 + f2;
        }
        const int _beg = set_u(bw, i, beg), _end = set_u(bw, i, end) + 2;
        const double r = 1. / sum;
        for (int k = _beg; k <= _end; ++k) F(fi, k) *= r;
        p *= sum;
        if (p < 1e-100) { Pr1 += -4.343 * log(p); p = 1.; }
    }
    {   // f[l_query+1]
        double sum = 0.;
        for (int k = 1; k <= l_ref; ++k) {
            const int u = set_u(bw, l_query, k);
            if (u < 3 || u >= bw2 * 3 + 3) continue;
            sum += F(fi, u) * sM + F(fi, u + 1) * sI;
        }
        p *= sum;
        if (p < 1e-100) { Pr1 += -4.343 * log(p); p = 1.; }
    }
    Pr1 += -4.343 * log(p * l_ref * l_query);
    #undef F
    return (int)(Pr1 + .499);
}

// ---- the register-resident forward pass: one lane per job, the band exactly as wide as the template ----------------
// A row of the band has W = 2*BW+1 cells x 3 states, kept in registers in an ALWAYS-SLIDING layout: position p (1..W) of
// row i is reference column k = p + i - bw - 1, so the diagonal neighbour of p is the old p and the upper one the old
// p+1 from the first row on (htslib's set_u() layout pins the band to column 0 for the first bw rows; the layout is
// storage only, the cells and the order of every sum are the reference's).  Cells under the band's lower edge (k < 1)
// are zero by construction -- their three neighbours are -- so only the upper edge (k > l_ref, or p > 2*bw+1 for a job
// whose band is narrower than the template's) needs masks.  Two row bodies, each straight-line code:
//   * fast rows: the band inside the reference for EVERY job of the wavefront, no N under it: 18 fp64 operations per
//     cell (EI = 1/4 is folded into the two I-state coefficients: scaling by a power of two commutes with rounding) and
//     4 integer ones for the emission (match / mismatch value of this query base, chosen by comparing 3-bit codes);
//   * edge rows: the same with the emission zeroed and the D state masked past the upper edge, N handled.
// The wavefront runs fast rows while all its jobs qualify (a wave-uniform branch: with a divergent one the register
// allocator keeps both variants' operands apart and the kernel needs twice the registers), then edge rows to the end.
// The jobs of a launch are sorted by (band, query length, l_ref - l_query), so the jobs of a wavefront leave the fast
// rows within a row or two of one another and finish together.  The query travels as one byte per base
// (code | quality << 3, written once per pileup entry by gap_qpack_kernel, eight rows per 8-byte load); the emission
// values 1 - 10^(-q/10) and 10^(-q/10) / 3 of all 256 byte values sit in a 4 KB LDS table (N in the read: both 1).
template <int BW>
__device__ __forceinline__ int probaln_fwd_exact(const uint8_t *ref, int l_ref, const uint8_t *qp, int l_query,
                                                 const double2 *emt, double d, double e_, int bw)
{
    constexpr int W = 2 * BW + 1, NB = (W + 3) / 4;
    double M[W + 2], I[W + 2], D[W + 2];
    #pragma unroll
    for (int p = 0; p < W + 2; ++p) M[p] = I[p] = D[p] = 0.;
    const double sM = 1. / (2 * l_query + 2), sI = sM;
    const double m0 = (1 - d - d) * (1 - sM), m1 = d * (1 - sM), m2 = m1;
    const double m3 = (1 - e_) * (1 - sI), m4 = e_ * (1 - sI);
    const double m6 = 1 - e_, m8 = e_;
    const double m1q = EI * m1, m4q = EI * m4;
    const double bM = (1 - d) / l_ref, bI = d / l_ref;
    const int top = 2 * bw + 1;
    // reference window: the base code of position p in byte p-1 of wv[] (a byte per position: the comparison with the query
    // base reads it through an operand byte-select, no shift or mask); row 1: k = p - bw
    uint32_t wv[NB];
    #pragma unroll
    for (int j = 0; j < NB; ++j) wv[j] = 0;
    #pragma unroll
    for (int p = 1; p <= W; ++p) { const int k = p - bw; wv[(p - 1) >> 2] |= (uint32_t)((k >= 1 && k <= l_ref) ? ref[k - 1] : 0) << (8 * ((p - 1) & 3)); }
    auto wbyte = [&](int p) { return (wv[(p - 1) >> 2] >> (8 * ((p - 1) & 3))) & 0xffu; };
    auto wshift = [&](uint32_t top_code) {                    // every position one down, top_code into position W
        #pragma unroll
        for (int j = 0; j + 1 < NB; ++j) wv[j] = __builtin_amdgcn_alignbyte(wv[j + 1], wv[j], 1);
        wv[NB - 1] >>= 8;
        wv[(W - 1) >> 2] |= top_code << (8 * ((W - 1) & 3));
    };
    double prod = 1., Pr1 = 0.;
    int result = 0;
    uint64_t qw = *(const uint64_t*)qp;
    auto finish = [&]() {             // f[l_query+1] over the cells of the last row (dead ones are zero), then the score
        double fsum = 0.;
        #pragma unroll
        for (int p = 1; p <= W; ++p) fsum += M[p] * sM + I[p] * sI;
        prod *= fsum;
        if (prod < 1e-100) { Pr1 += -4.343 * log(prod); prod = 1.; }
        Pr1 += -4.343 * log(prod * l_ref * l_query);
        result = (int)(Pr1 + .499);
    };
    {   // f[1]
        double sum = 0.;
        const int end = l_ref < bw + 1 ? l_ref : bw + 1;
        const int qb = (int)(qw & 0xff);
        const double2 em = emt[qb];
        const uint32_t qy = qb & 7;
        #pragma unroll
        for (int p = 1; p <= W; ++p) {
            const double lv = (p > bw && p <= bw + end) ? 1. : 0.;
            const uint32_t rb = wbyte(p);
            const double e = rb > 3 ? 1. : rb == qy ? em.x : em.y;
            const double a = lv * (e * bM), b = lv * (EI * bI);
            M[p] = a; I[p] = b;
            sum += a + b;
        }
        #pragma unroll
        for (int p = 1; p <= W; ++p) { M[p] /= sum; I[p] /= sum; }
        prod *= sum;
        if (prod < 1e-100) { Pr1 += -4.343 * log(prod); prod = 1.; }
        if (l_query == 1) finish();
    }
    int i = 2;
    if (__builtin_amdgcn_ballot_w64(bw != BW || l_query < 3 || l_ref < bw + 2) == 0) {
        // Fast rows (the last row of a job is always an edge row: it ends with finish()).  Row i shifts reference base
        // ref[i + bw - 1] into the window and reads query byte i - 1: both streams come eight rows at a time as one 8-byte
        // load each, issued a group ahead of its first use.
        const uint8_t *rp = ref + bw + 1;                     // ref[i + bw - 1] for i = 2
        uint64_t rq, rq_next, qw_next;
        __builtin_memcpy(&rq, rp, 8);
        __builtin_memcpy(&rq_next, rp + 8, 8);
        qw_next = *(const uint64_t*)(qp + 8);
        qw >>= 8;                                             // byte of row 2
        for (;; ++i) {
            const bool fast = i < l_query && i + bw <= l_ref;
            if (__builtin_amdgcn_ballot_w64(!fast) != 0) break;
            const uint32_t nb = (uint32_t)(rq & 0xff);
            // an N about to enter the window (or already in it from row 1): the masked rows handle it
            uint32_t anyn = nb;
            #pragma unroll
            for (int j = 0; j < NB; ++j) anyn |= wv[j];
            if (__builtin_amdgcn_ballot_w64((anyn & 0x04040404u) != 0) != 0) break;
            wshift(nb);
            const int qb = (int)(qw & 0xff);
            const double2 em = emt[qb];
            const uint32_t qyi = qb & 7;
            // the streams move on: row i+1 reads the next byte; every eighth row the group loaded eight rows ago takes over
            if (((i - 1) & 7) == 0) { rq = rq_next; __builtin_memcpy(&rq_next, rp + (i - 1) + 8, 8); } else rq >>= 8;
            if ((i & 7) == 0) { qw = qw_next; qw_next = *(const uint64_t*)(qp + i + 8); } else qw >>= 8;
            double sum = 0.;
            #pragma unroll
            for (int p = 1; p <= W; ++p) {
                const double e = wbyte(p) == qyi ? em.x : em.y;
                const double f0 = e * (m0 * M[p] + m3 * I[p] + m6 * D[p]);
                if (p == 1) { const double f1 = m1q * M[p + 1] + m4q * I[p + 1]; sum += f0 + f1; M[p] = f0; I[p] = f1; D[p] = 0.; }
                else if (p == W) { const double f2 = m2 * M[p - 1] + m8 * D[p - 1]; sum += f0 + f2; M[p] = f0; I[p] = 0.; D[p] = f2; }
                else {
                    const double f1 = m1q * M[p + 1] + m4q * I[p + 1];
                    const double f2 = m2 * M[p - 1] + m8 * D[p - 1];
                    sum += f0 + f1 + f2;
                    M[p] = f0; I[p] = f1; D[p] = f2;
                }
            }
            const double r = 1. / sum;
            #pragma unroll
            for (int p = 1; p <= W; ++p) { M[p] *= r; I[p] *= r; D[p] *= r; }
            prod *= sum;
            if (prod < 1e-100) { Pr1 += -4.343 * log(prod); prod = 1.; }
        }
    }
    for (; i <= l_query; ++i) {       // edge rows
        qw = *(const uint64_t*)(qp + ((i - 1) & ~7)) >> (8 * ((i - 1) & 7));
        const int qb = (int)(qw & 0xff);
        const double2 em = emt[qb];
        const uint32_t qyi = qb & 7;
        const int kt = i - bw - 1 + W;
        wshift(kt <= l_ref ? (uint32_t)ref[kt - 1] : 0u);
        const int hi = l_ref - (i - bw) + 1 < top ? l_ref - (i - bw) + 1 : top;
        double sum = 0.;
        #pragma unroll
        for (int p = 1; p <= W; ++p) {
            const bool live = p <= hi;
            const uint32_t rb = wbyte(p);
            double e = rb > 3 ? 1. : rb == qyi ? em.x : em.y;
            e = live ? e : 0.;
            const double tv = live ? 1. : 0.;
            const double f0 = e * (m0 * M[p] + m3 * I[p] + m6 * D[p]);
            const double f1 = m1q * M[p + 1] + m4q * I[p + 1];
            const double f2 = tv * (m2 * M[p - 1] + m8 * D[p - 1]);
            sum += f0 + f1 + f2;
            M[p] = f0; I[p] = f1; D[p] = f2;
        }
        const double r = 1. / sum;
        #pragma unroll
        for (int p = 1; p <= W; ++p) { M[p] *= r; I[p] *= r; D[p] *= r; }
        prod *= sum;
        if (prod < 1e-100) { Pr1 += -4.343 * log(prod); prod = 1.; }
        if (i == l_query) finish();
    }
    return result;
}

// tpos2qpos, bam2bcf_indel.c:40-66
__device__ int gap_tpos2qpos(int cpos, int n_cigar, const uint32_t *cigar, int tpos, int is_left, int *_tpos)
{
    int x = cpos, y = 0, last_y = 0;
    *_tpos = cpos;
    for (int k = 0; k < n_cigar; ++k) {
        const int op = cigar[k] & 0xf, l = (int)(cigar[k] >> 4);
        if (op == 0 || op == 7 || op == 8) {
            if (cpos > tpos) return y;
            if (x + l > tpos) { *_tpos = tpos; return y + (tpos - x); }
            x += l; y += l; last_y = y;
        } else if (op == 1 || op == 4) y += l;
        else if (op == 2 || op == 3) {
            if (x + l > tpos) { *_tpos = is_left ? x : x + l; return y; }
            x += l;
        }
    }
    *_tpos = x;
    return last_y;
}

// What a read contributes to every realignment of its site (it does not depend on the candidate type): the part of the
// read inside the window and where it starts and ends on the reference -- the two tpos2qpos() calls of
// bam2bcf_indel.c:326-327 -- the read's sample, and where its packed query goes.  One lane per pileup entry.
__global__ __launch_bounds__(256) void gap_entry_kernel(const GapIn in, const GapSite *sites, int n_ent, GapEntry *ent)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_ent) return;
    int lo = 0, hi = in.n_sites - 1;                        // the site of entry e: the last one with smpl_off[site*n] <= e
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (in.smpl_off[(size_t)mid * in.n_smpl] <= e) lo = mid; else hi = mid - 1; }
    const GapSite &S = sites[lo];
    if (!S.live) { GapEntry z{}; z.smpl = -1; ent[e] = z; return; }     // (gap_qpack_kernel looks at every entry)
    const int32_t *soff = in.smpl_off + (size_t)lo * in.n_smpl;
    int a = 0, b = in.n_smpl - 1;                           // the sample of entry e: the last s with soff[s] <= e
    while (a < b) { const int mid = (a + b + 1) >> 1; if (soff[mid] <= e) a = mid; else b = mid - 1; }
    const int r = in.p_read[e];
    const uint32_t *cigar = in.cig + in.r_cig_off[r];
    const int ncig = in.r_ncig[r];
    bool skip = (in.r_flag[r] & 4) != 0;                    // unmapped reads (:319)
    for (int k = 0; k < ncig; ++k) if ((cigar[k] & 0xf) == 3) skip = true;        // reads with a reference skip (:321-323)
    GapEntry g{};
    g.smpl = skip ? -1 : a;
    g.q8 = S.q8_0 + (uint32_t)(e - S.e0) * (uint32_t)(S.qstride >> 3);
    if (!skip) {
        g.qbeg = gap_tpos2qpos(in.r_pos[r], ncig, cigar, S.left, 0, &g.tbeg);
        g.qend = gap_tpos2qpos(in.r_pos[r], ncig, cigar, S.right, 1, &g.tend);
    }
    ent[e] = g;
}

// The query of every entry as the realignment reads it (bam2bcf_indel.c:339-345), one byte per base: code 0..4 |
// capped quality << 3, the entry's qend - qbeg bytes from ent[e].q8 * 8 on.  A lane per (entry, group of eight bases).
__global__ __launch_bounds__(256) void gap_qpack_kernel(const GapIn in, const GapEntry *ent, int n_ent, int chunks, uint8_t *qpack)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int e = (int)(idx / chunks), c = (int)(idx - (long)e * chunks);
    if (e >= n_ent) return;
    const GapEntry g = ent[e];
    const int lq = g.qend - g.qbeg;
    if (g.smpl < 0 || c * 8 >= lq) return;
    const int r = in.p_read[e];
    const size_t qo = (size_t)in.r_seq_off[r] + g.qbeg + (size_t)c * 8;
    const bool has_zq = in.zq && in.r_has_zq && in.r_has_zq[r];
    // eight bases, qualities (and ZQ bytes) as one unaligned 8-byte load each (the pools end in >= 8 bytes of slack)
    uint64_t sv, qv, zv = 0;
    __builtin_memcpy(&sv, in.seq16 + qo, 8);
    __builtin_memcpy(&qv, in.qual + qo, 8);
    if (has_zq) __builtin_memcpy(&zv, in.zq + qo, 8);
    const int nb = lq - c * 8 < 8 ? lq - c * 8 : 8;
    uint64_t v = 0;
    #pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int base = (int)((0x4444444344424104ull >> (4 * ((sv >> (8 * k)) & 15))) & 7);
        int q = (int)((qv >> (8 * k)) & 0xff);
        if (has_zq) q = (int)(uint8_t)(q + ((int)((zv >> (8 * k)) & 0xff) - 64));
        q = q > 30 ? 30 : q < 7 ? 7 : q;
        v |= (uint64_t)(base | q << 3) << (8 * k);
    }
    if (nb < 8) v &= (1ull << (8 * nb)) - 1;
    *(uint64_t*)(qpack + ((size_t)g.q8 + c) * 8) = v;
}

void launch_gap_entries(const GapIn &in, const GapSite *sites, int n_ent, GapEntry *ent, int max_qstride, uint8_t *qpack, hipStream_t s)
{
    if (n_ent <= 0) return;
    hipLaunchKernelGGL(gap_entry_kernel, dim3((n_ent + 255) / 256), dim3(256), 0, s, in, sites, n_ent, ent);
    const int chunks = max_qstride >> 3;
    if (chunks > 0) {
        const long lanes = (long)n_ent * chunks;
        hipLaunchKernelGGL(gap_qpack_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, in, ent, n_ent, chunks, qpack);
    }
}

// One realignment of bam2bcf_indel.c:313-357: read K of the site against candidate type t.  Jobs are numbered
// job0 + t*N + K inside a site.
struct JobDesc { const uint8_t *ref; int l_ref, l_query, bw, eff; uint32_t ref_off, q8; QSrc qs; bool skip; };
__device__ __forceinline__ JobDesc decode_job(const ProbalnParams &P, uint32_t job)
{
    JobDesc d{};
    const GapIn &in = P.gin;
    int lo = 0, hi = P.n_sites - 1;                         // the first site whose running job total exceeds `job`
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (P.sites[mid].job_end > job) hi = mid; else lo = mid + 1; }
    const GapSite &S = P.sites[lo];
    const uint32_t rel = job - S.job0;
    const int t = (int)(rel / (uint32_t)S.N), K = (int)(rel % (uint32_t)S.N), e = S.e0 + K;
    const GapEntry g = P.ent[e];
    d.skip = g.smpl < 0;
    if (d.skip) return d;
    const int r = in.p_read[e];
    const int ty = S.types[t], aty = abs(ty);
    int tbeg = g.tbeg;
    if (ty < 0) tbeg = tbeg - aty > S.left ? tbeg - aty : S.left;
    d.ref_off = S.ref2_0 + (uint32_t)(((size_t)t * in.n_smpl + g.smpl) * S.max_ref2 + (tbeg - S.left));
    d.ref = P.ref2 + d.ref_off;
    d.l_ref = g.tend - tbeg + aty;
    d.l_query = g.qend - g.qbeg;
    d.bw = aty + 3;
    // the band probaln_glocal really uses (probaln.c): min(bw, max(l_ref, l_query)), at least |l_ref - l_query|
    int eff = d.l_ref > d.l_query ? d.l_ref : d.l_query;
    if (eff > d.bw) eff = d.bw;
    if (eff < abs(d.l_ref - d.l_query)) eff = abs(d.l_ref - d.l_query);
    d.eff = eff;
    d.q8 = g.q8;
    const size_t qo = (size_t)in.r_seq_off[r] + g.qbeg;
    d.qs = QSrc{in.seq16 + qo, in.qual + qo, (in.zq && in.r_has_zq && in.r_has_zq[r]) ? in.zq + qo : nullptr};
    return d;
}

// ---- the jobs, decoded once: a 16-byte record and a sort key per job ----
// key = band class << 13 | min(l_query, 255) << 5 | clamp(l_ref - l_query + 16, 0, 31); classes PROBALN_BW_MIN..PROBALN_BW_MAX
// run in probaln_exact_kernel<class> (bands narrower than PROBALN_BW_MIN in its kernel), wider bands are listed for
// probaln_wide_kernel, jobs without a realignment (skipped reads, an empty side: score 0, the arrays are zeroed) sort last.
__global__ __launch_bounds__(256) void probaln_jobs_kernel(const ProbalnParams P)
{
    const uint32_t job = blockIdx.x * 256u + threadIdx.x;
    if (job >= (uint32_t)P.n_jobs) return;
    const JobDesc j = decode_job(P, job);
    uint32_t cls = PROBALN_CLS_NONE;
    PJob pj{};
    if (!j.skip && j.l_ref > 0 && j.l_query > 0) {
        if (j.eff > PROBALN_LDS16_MAX || j.l_ref > 65535 || j.l_query > 65535 || P.force_wide) {
            P.wide[atomicAdd(&P.tot->n_wide, 1u)] = job;
            atomicMax(&P.tot->max_eff, j.eff);
            cls = PROBALN_CLS_WIDE;
        } else {
            cls = j.eff > PROBALN_LDS_MAX ? PROBALN_CLS_LDS16 : j.eff > PROBALN_BW_MAX ? PROBALN_CLS_LDS : j.eff < PROBALN_BW_MIN ? PROBALN_BW_MIN : j.eff;
            if (cls >= PROBALN_CLS_LDS) atomicAdd(&P.tot->n_lds, 1u);
            pj.ref_off = j.ref_off; pj.q8 = j.q8; pj.l_ref = (uint16_t)j.l_ref; pj.l_query = (uint16_t)j.l_query; pj.eff = (uint16_t)j.eff;
        }
    }
    int dl = j.l_ref - j.l_query + 16;
    dl = dl < 0 ? 0 : dl > 31 ? 31 : dl;
    const uint32_t lq = (uint32_t)(j.l_query > 255 ? 255 : j.l_query < 0 ? 0 : j.l_query);
    P.pjob[job] = pj;
    // the LDS class is sorted by band width first: a wavefront's jobs share the sweep over the widest band among them
    // (... then by which of the two sequences is the longer one -- the band's live cells sit at opposite ends of it -- then by length)
    const uint32_t dir = j.l_ref > j.l_query ? 1u : 0u;
    P.key_in[job] = cls == PROBALN_CLS_LDS ? cls << 13 | (uint32_t)j.eff << 6 | dir << 5 | lq >> 3
                  : cls == PROBALN_CLS_LDS16 ? cls << 13 | (uint32_t)(j.eff >> 2 > 127 ? 127 : j.eff >> 2) << 6 | dir << 5 | lq >> 3
                  : cls << 13 | lq << 5 | (uint32_t)dl;
    P.val_in[job] = job;
}

// first sorted slot of every class: cls_begin[c] = number of keys with class < c, c = 0..16
__global__ void probaln_bounds_kernel(const uint32_t *key_sorted, int n, ProbalnQueue *q)
{
    const int c = threadIdx.x;
    if (c > 16) return;
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((key_sorted[mid] >> 13) >= (uint32_t)c) hi = mid; else lo = mid + 1; }
    q->cls_begin[c] = (uint32_t)lo;
    if (c < 16) { q->next1[c] = 0; q->next2[c] = 0; q->n2[c] = 0; }
    if (c <= PROBALN_LDS_GROUPS + 2) {
        // group g of the LDS class: the jobs with band widths in (cap[g-1], cap[g]]; lds_begin[g] = keys below class | (cap[g-1] + 1) << 6;
        // the last two ranges are the sixteen-jobs-a-wavefront class (its key holds the band width >> 2)
        const int caps[PROBALN_LDS_GROUPS] = PROBALN_LDS_CAPS;
        const uint32_t thr = c == 0 ? PROBALN_CLS_LDS << 13 : c == PROBALN_LDS_GROUPS ? PROBALN_CLS_LDS16 << 13
                           : c == PROBALN_LDS_GROUPS + 1 ? PROBALN_CLS_LDS16 << 13 | (uint32_t)((PROBALN_LDS16_SPLIT >> 2) + 1) << 6
                           : c == PROBALN_LDS_GROUPS + 2 ? (PROBALN_CLS_LDS16 + 1) << 13 : PROBALN_CLS_LDS << 13 | (uint32_t)(caps[c - 1] + 1) << 6;
        int a = 0, b = n;
        while (a < b) { const int mid = (a + b) >> 1; if (key_sorted[mid] >= thr) b = mid; else a = mid + 1; }
        q->lds_begin[c] = (uint32_t)a;
        if (c <= PROBALN_LDS_GROUPS + 1) q->lds_next[c] = 0;
    }
}

// PASS 1: the jobs of class BW in sorted order, parameter set {1e-4, 1e-2}; jobs scoring above 5 are listed for
// PASS 2 (the second parameter set {1e-6, 1e-3}, bam2bcf_indel.c:293-294, 346-356).  One wavefront per workgroup; the
// grid is sized for the machine, not for the class: a wavefront takes the next 64 jobs of its class from a counter
// until none are left (every wavefront reaches that test, also when the class is empty).
template <int BW, int PASS>
__global__ __launch_bounds__(64) void probaln_exact_kernel(const ProbalnParams P)
{
    // (the emission table is read from device memory -- 4 KB, resident in every CU's cache -- not from LDS: a workgroup of this
    // kernel then needs no LDS at all and fits beside the wide-band classes' workgroups, which take a CU's whole LDS)
    const double2 *s_emt = P.emt;
    ProbalnQueue *Q = P.queue;
    const uint32_t c0 = Q->cls_begin[BW];
    const uint32_t n = PASS == 1 ? Q->cls_begin[BW + 1] - c0 : Q->n2[BW];
    if (n == 0) return;                                   // (the same for every wavefront of the launch)
    unsigned long long cells = 0, passes = 0;
    const double gd = PASS == 1 ? 1e-4 : 1e-6, ge = PASS == 1 ? 1e-2 : 1e-3;
    for (;;) {
        uint32_t base = 0;
        if (threadIdx.x == 0) base = atomicAdd(PASS == 1 ? &Q->next1[BW] : &Q->next2[BW], 64u);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= n) break;
        const uint32_t i = base + threadIdx.x;
        const bool have = i < n;
        uint32_t job = 0;
        int sc = 0;
        if (have) {
            job = PASS == 1 ? P.val_sorted[c0 + i] : P.list2[c0 + i];
            const PJob j = P.pjob[job];
            sc = probaln_fwd_exact<BW>(P.ref2 + j.ref_off, j.l_ref, P.qpack + (size_t)j.q8 * 8, j.l_query, s_emt, gd, ge, j.eff);
            int l = (int)(100. * sc / j.l_query + .499);
            if (l > 255) l = 255;
            const int v = sc << 8 | l;
            if (PASS == 1) P.score1[job] = v;
            P.score2[job] = v;
            cells += (unsigned long long)j.l_query * (2 * j.eff + 1) * 3;
            ++passes;
        }
        if (PASS == 1) {
            const bool again = have && sc > 5;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(again);
            if (m) {
                uint32_t at = 0;
                if (threadIdx.x == 0) at = atomicAdd(&Q->n2[BW], (uint32_t)__popcll(m));
                at = __builtin_amdgcn_readfirstlane(at);
                if (again) P.list2[c0 + at + __popcll(m & ((1ull << threadIdx.x) - 1))] = job;
            }
        }
    }
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) { passes += __shfl_xor(passes, o); cells += __shfl_xor(cells, o); }
    if (threadIdx.x == 0 && passes) { atomicAdd(&P.tot->n_passes, passes); atomicAdd(&P.tot->dp_cells, cells); }
}

// ---- bands wider than the register classes (indels of 8 bases and more: bw = |type| + 3, bam2bcf_indel.c:293-294): the row in LDS ----
// One lane per job as before, one wavefront per workgroup, the band in the same always-sliding layout as above -- position p
// of row i is reference column k = p + i - bw - 1 -- but the row lives in LDS, [cell][lane] doubles (conflict-free: a wavefront's
// lanes read consecutive 8-byte words), and is updated IN PLACE: the new cell p needs the old cells p (diagonal) and p + 1
// (above), so a sweep over ascending p that carries the old cell p in registers can overwrite slot p as it goes.  Two things
// keep the LDS bytes at 16 per cell instead of 48 (two rows of three states):
//   * the row is stored UNSCALED; the factor 1/s_i of the row (a division for row 1, as probaln.c has it) is applied where the
//     next row reads a value -- the same single rounding as scaling the stored row;
//   * D is not stored: D'[p] = m2 M'[p-1] + m8 D'[p-1] is a recurrence over the row's own unscaled M', so the next row's sweep
//     re-runs it one cell behind its own -- the same operations in the same order, hence the same bits (what baq.hip does for
//     its even rows).
// Every sum keeps the reference's order; cells outside the band or the reference are exact zeros, as in the edge rows above.
// A wavefront sweeps to the widest band and the longest query among its 64 jobs (sorted by band, then length).
__device__ __forceinline__ int wave_max_i(int v)
{
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(v, o); v = w > v ? w : v; }
    return v;
}
// LS: the lanes that hold a job (64, or 16 for the bands no 64 jobs fit LDS with); cell p of this lane's column at [p * LS], the
// cell's M' and I' side by side (one 16-byte LDS access).  The sweep runs in groups of eight cells: a group's reads of the row
// above go out a whole group ahead of their use (they touch cells this group's stores do not), and so does the 8-byte load of
// the bases under the next group.  A group all of whose cells are inside every job's band and reference, with no N under
// them, runs without masks (a wave-uniform choice, as in the register classes).
__device__ __forceinline__ int wave_min_i(int v)
{
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(v, o); v = w < v ? w : v; }
    return v;
}
template <int LS>
__device__ __forceinline__ int probaln_fwd_lds(const uint8_t *ref, int l_ref, const uint8_t *qp, int l_query, int bw, const double2 *emt,
                               double d, double e_, double2 *sR, bool active, bool owns)
{
    if (!active) { l_ref = 0; l_query = 0; bw = 0; }
    const int W = active ? 2 * bw + 1 : 0;
    const int Wmax = wave_max_i(W), Lmax = wave_max_i(l_query);
    if (Wmax == 0) return 0;
    const int Wc = (Wmax + 7) & ~7;                                          // whole groups: cells 0 .. Wc + 1 are this lane's
    const double sMc = 1. / (2 * l_query + 2), sIc = sMc;
    const double m0 = (1 - d - d) * (1 - sMc), m1 = d * (1 - sMc), m2 = m1;
    const double m3 = (1 - e_) * (1 - sIc), m4 = e_ * (1 - sIc);
    const double m6 = 1 - e_, m8 = e_;
    const double m1q = EI * m1, m4q = EI * m4;
    const double bM = (1 - d) / l_ref, bI = d / l_ref;
    double prod = 1., Pr1 = 0., scale = 1.;
    uint64_t qw = *(const uint64_t*)qp;                      // (a lane without a job reads its pool's first bytes: any value will do)
    {   // f[1]: k = p - bw; stored scaled (a division, probaln.c): the factor the next row applies to it is 1
        double sum = 0.;
        const int end = l_ref < bw + 1 ? l_ref : bw + 1;
        const int qb = (int)(qw & 0xff);
        const double2 em = emt[qb];
        const uint32_t qy = qb & 7;
        if (owns) { sR[0] = make_double2(0., 0.); sR[(size_t)(Wc + 1) * LS] = make_double2(0., 0.); }
        for (int p = 1; p <= Wc; ++p) {
            const int k = p - bw;
            const bool live = p <= W && k >= 1 && k <= end;
            const uint32_t rb = live ? ref[k - 1] : 0u;
            const double e = rb > 3 ? 1. : rb == qy ? em.x : em.y;
            const double a = live ? e * bM : 0., b = live ? EI * bI : 0.;
            if (owns) sR[(size_t)p * LS] = make_double2(a, b);
            sum += a + b;
        }
        for (int p = 1; p <= Wc; ++p) {
            const double2 c = sR[(size_t)p * LS];
            if (owns) sR[(size_t)p * LS] = make_double2(c.x / sum, c.y / sum);
        }
        prod *= sum;
        if (prod < 1e-100) { Pr1 += -4.343 * log(prod); prod = 1.; }
    }
    // Rows 2 .. the longest query of the wavefront, run by every lane (see probaln_fwd_regs): a lane takes its score when its own
    // last row is done and sweeps on over its own column with values nobody reads.
    auto row = [&](int i, bool mine, int hi, int hio, int clear, int pa, int pz) {           // pa, pz: first cell of the first / last group to sweep
        const bool first = i == 2;                                             // the row above is row 1: it has no D
        qw = ((i - 1) & 7) == 0 ? *(const uint64_t*)(qp + (i - 1)) : qw >> 8;
        const int qb = (int)(qw & 0xff);
        const double2 em = emt[qb];
        const uint32_t qyi = qb & 7;
        const uint8_t *rrow = ref + (i - bw - 2);                              // the base under cell p: rrow[p] (the pool is padded on both sides)
        double Mo_un = 0., Do_un = 0.;                                         // the row above, unscaled: M'[p-1], D'[p-1] (zeros in front of the first group)
        const double2 c1 = sR[(size_t)pa * LS];
        double Mc_un = c1.x, Mc = c1.x * scale, Ic = c1.y * scale;             // ... its cell p: M' unscaled, M and I scaled
        double Mn = 0., Dn = 0.;                                               // this row: M[p-1], D[p-1]
        double sum = 0.;
        uint64_t rw_next;
        __builtin_memcpy(&rw_next, rrow + pa, 8);
        double2 nx[8], nn[8];
        #pragma unroll
        for (int u = 0; u < 8; ++u) nx[u] = sR[(size_t)(pa + 1 + u) * LS];
        for (int p0 = pa; p0 <= pz; p0 += 8) {
            const uint64_t rw = rw_next;
            __builtin_memcpy(&rw_next, rrow + p0 + 8, 8);
            double2 *cR = sR + (size_t)p0 * LS;
            if (p0 + 8 <= pz) {
                #pragma unroll
                for (int u = 0; u < 8; ++u) nn[u] = cR[(size_t)(9 + u) * LS];
            }
            const bool masked = first || p0 + 7 > clear || __builtin_amdgcn_ballot_w64(mine && (rw & 0x0404040404040404ull) != 0) != 0;
            if (!masked) {
                #pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double Mx_un = nx[u].x, Mx = Mx_un * scale, Ix = nx[u].y * scale;
                    Do_un = m2 * Mo_un + m8 * Do_un;
                    const double Ds = Do_un * scale;
                    const uint32_t rb = (uint32_t)(rw >> (8 * u)) & 0xffu;
                    const double e = rb == qyi ? em.x : em.y;
                    const double f0 = e * (m0 * Mc + m3 * Ic + m6 * Ds);
                    const double f1 = m1q * Mx + m4q * Ix;
                    const double f2 = m2 * Mn + m8 * Dn;
                    sum += f0 + f1 + f2;
                    if (LS == 64 || owns) cR[(size_t)u * LS] = make_double2(f0, f1);
                    Mo_un = Mc_un; Mc_un = Mx_un; Mc = Mx; Ic = Ix;
                    Mn = f0; Dn = f2;
                }
            } else {
                #pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int p = p0 + u;
                    const double Mx_un = nx[u].x, Mx = Mx_un * scale, Ix = nx[u].y * scale;
                    const double tvo = (!first && p <= hio) ? 1. : 0.;
                    Do_un = tvo * (m2 * Mo_un + m8 * Do_un);                   // D' of the row above at p, as that row formed it
                    const double Ds = Do_un * scale;
                    const bool live = p <= hi;
                    const uint32_t rb = (uint32_t)(rw >> (8 * u)) & 0xffu;
                    double e = rb > 3 ? 1. : rb == qyi ? em.x : em.y;
                    e = live ? e : 0.;
                    const double tv = live ? 1. : 0.;
                    const double f0 = e * (m0 * Mc + m3 * Ic + m6 * Ds);
                    const double f1 = tv * (m1q * Mx + m4q * Ix);              // (past `hi` the cell above may be an earlier row's: zero, as the full sweep has it)
                    const double f2 = tv * (m2 * Mn + m8 * Dn);
                    sum += f0 + f1 + f2;
                    if (LS == 64 || owns) cR[(size_t)u * LS] = make_double2(f0, f1);
                    Mo_un = Mc_un; Mc_un = Mx_un; Mc = Mx; Ic = Ix;
                    Mn = f0; Dn = f2;
                }
            }
            #pragma unroll
            for (int u = 0; u < 8; ++u) nx[u] = nn[u];
        }
        scale = 1. / sum;
        prod *= sum;
        if (prod < 1e-100) { Pr1 += -4.343 * log(prod); prod = 1.; }
    };
    int result = 0;
    for (int i = 2; ; ++i) {
        if (i - 1 == l_query) {       // this lane's last row is done: f[l_query + 1] over its cells inside the reference, then the score
            int hl = l_ref - (l_query - bw) + 1; hl = hl < W ? hl : W;
            double fsum = 0.;
            for (int p = 1; p <= hl; ++p) {
                const double2 c = sR[(size_t)p * LS];
                fsum += (c.x * scale) * sMc + (c.y * scale) * sIc;
            }
            double pr = prod * fsum, P1 = Pr1;
            if (pr < 1e-100) { P1 += -4.343 * log(pr); pr = 1.; }
            P1 += -4.343 * log(pr * l_ref * l_query);
            result = (int)(P1 + .499);
        }
        if (i > Lmax) break;
        const bool mine = i <= l_query;
        int hi = l_ref - (i - bw) + 1; hi = hi < W ? hi : W;                   // cells past it lie beyond the reference (or this job's band)
        int hio = l_ref - (i - 1 - bw) + 1; hio = hio < W ? hio : W;           // the same for the row above (its D is re-run)
        const int clear = wave_min_i(mine ? (hi < hio ? hi : hio) : 0x7fffffff);
        // Cells in front of the reference's first column (p < bw + 2 - i) are zeros since row 1 and stay zeros; cells past `hi` are
        // zeros in this row and are not read by any later row (hi only falls): the sweep covers the groups of eight in between,
        // over all lanes.  (What a skipped group holds past `hi` is an earlier row's: nothing reads it, see f[l_query + 1] above.)
        int lo = bw + 2 - i; lo = lo < 1 ? 1 : lo;
        const int lo_min = wave_min_i(mine ? lo : 0x7fffffff), hi_max = wave_max_i(mine ? hi : 0);
        if (hi_max < 1) continue;
        row(i, mine, hi, hio, clear, ((lo_min - 1) & ~7) + 1, ((hi_max - 1) & ~7) + 1);
    }
    return result;
}

// The same sweep with the whole row in registers: NC groups of eight cells, everything unrolled, so that every cell is a named
// register pair (M', I') -- four registers a cell instead of the six of probaln_fwd_exact (D re-run, see above), which is what
// lets a band of 43 (88 cells) fit the 512 registers a lane has at one wavefront per SIMD.  The register file of a CU is three
// times its LDS: bands up to 43 run here with every SIMD busy, where their rows in LDS leave room for one or two wavefronts a CU.
template <int NC>
__device__ __forceinline__ int probaln_fwd_regs(const uint8_t *ref, int l_ref, const uint8_t *qp, int l_query, int bw, const double2 *emt,
                                                double d, double e_, bool active)
{
    constexpr int WC = NC * 8;
    if (!active) { l_ref = 0; l_query = 0; bw = 0; }
    const int W = active ? 2 * bw + 1 : 0;
    const int Wmax = wave_max_i(W), Lmax = wave_max_i(l_query);
    if (Wmax == 0) return 0;
    double2 R[WC + 2];                                       // cells 0 .. WC + 1 (the two ends stay zero)
    const double sMc = 1. / (2 * l_query + 2), sIc = sMc;
    const double m0 = (1 - d - d) * (1 - sMc), m1 = d * (1 - sMc), m2 = m1;
    const double m3 = (1 - e_) * (1 - sIc), m4 = e_ * (1 - sIc);
    const double m6 = 1 - e_, m8 = e_;
    const double m1q = EI * m1, m4q = EI * m4;
    const double bM = (1 - d) / l_ref, bI = d / l_ref;
    double prod = 1., Pr1 = 0., scale = 1.;
    uint64_t qw = *(const uint64_t*)qp;                      // (a lane without a job reads its pool's first bytes: any value will do)
    {   // f[1]: k = p - bw
        double sum = 0.;
        const int end = l_ref < bw + 1 ? l_ref : bw + 1;
        const int qb = (int)(qw & 0xff);
        const double2 em = emt[qb];
        const uint32_t qy = qb & 7;
        R[0] = make_double2(0., 0.); R[WC + 1] = make_double2(0., 0.);
        #pragma unroll
        for (int c = 0; c < NC; ++c) {
            uint64_t rw;
            __builtin_memcpy(&rw, ref + (8 * c - bw), 8);     // the bases under cells 8c + 1 .. 8c + 8: ref[k - 1], k = p - bw (the pool is padded)
            #pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int p = 8 * c + u + 1, k = p - bw;
                const bool live = p <= W && k >= 1 && k <= end;
                const uint32_t rb = live ? (uint32_t)(rw >> (8 * u)) & 0xffu : 0u;
                const double e = rb > 3 ? 1. : rb == qy ? em.x : em.y;
                const double a = live ? e * bM : 0., b = live ? EI * bI : 0.;
                R[p] = make_double2(a, b);
                sum += a + b;
            }
        }
        #pragma unroll
        for (int p = 1; p <= WC; ++p) { R[p].x /= sum; R[p].y /= sum; }        // (row 1 is scaled by a division, probaln.c; later rows keep their factor for the next row's reads: x * 1 = x here)
        prod *= sum;
        if (prod < 1e-100) { Pr1 += -4.343 * log(prod); prod = 1.; }
    }
    // Rows 2 .. the longest query of the wavefront, run by EVERY lane: a lane takes its score when its own last row is done
    // (f[l_query + 1] below reads the row, it does not change it) and computes on with values nobody reads -- a lane-divergent
    // branch around the sweep would keep two copies of the row's registers apart.
    auto row = [&](int i, bool mine, int hi, int hio, int clear, int ca, int cz) {           // ca, cz: first and last group of eight to sweep
        const bool first = i == 2;                            // the row above is row 1: it has no D
        qw = ((i - 1) & 7) == 0 ? *(const uint64_t*)(qp + (i - 1)) : qw >> 8;
        const int qb = (int)(qw & 0xff);
        const double2 em = emt[qb];
        const uint32_t qyi = qb & 7;
        const uint8_t *rrow = ref + (i - bw - 2);
        auto sc = [&](double x) { return x * scale; };
        double Mo_un = 0., Do_un = 0.;
        double Mc_un = R[1].x, Mc = sc(R[1].x), Ic = sc(R[1].y);
        double Mn = 0., Dn = 0.;
        double sum = 0.;
        uint64_t rw_next;
        __builtin_memcpy(&rw_next, rrow + 1, 8);
        #pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int p0 = 1 + 8 * c;
            if (c < ca) {                                     // (wave-uniform) a group of zeros in front of the reference: step over it
                if (c + 1 < NC) { __builtin_memcpy(&rw_next, rrow + p0 + 8, 8); Mc_un = R[p0 + 8].x; Mc = sc(Mc_un); Ic = sc(R[p0 + 8].y); }
            } else if (c <= cz) {
                const uint64_t rw = rw_next;
                if (c + 1 < NC) __builtin_memcpy(&rw_next, rrow + p0 + 8, 8);
                const bool masked = first || p0 + 7 > clear || __builtin_amdgcn_ballot_w64(mine && (rw & 0x0404040404040404ull) != 0) != 0;
                if (!masked) {
                    #pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const double2 nx = R[p0 + u + 1];
                        const double Mx_un = nx.x, Mx = Mx_un * scale, Ix = nx.y * scale;
                        Do_un = m2 * Mo_un + m8 * Do_un;
                        const double Ds = Do_un * scale;
                        const uint32_t rb = (uint32_t)(rw >> (8 * u)) & 0xffu;
                        const double e = rb == qyi ? em.x : em.y;
                        const double f0 = e * (m0 * Mc + m3 * Ic + m6 * Ds);
                        const double f1 = m1q * Mx + m4q * Ix;
                        const double f2 = m2 * Mn + m8 * Dn;
                        sum += f0 + f1 + f2;
                        R[p0 + u] = make_double2(f0, f1);
                        Mo_un = Mc_un; Mc_un = Mx_un; Mc = Mx; Ic = Ix;
                        Mn = f0; Dn = f2;
                        if (NC > 4) __builtin_amdgcn_sched_barrier(0);         // (a cell's instructions stay together)
                    }
                } else {
                    #pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int p = p0 + u;
                        const double2 nx = R[p0 + u + 1];
                        const double Mx_un = nx.x, Mx = sc(Mx_un), Ix = sc(nx.y);
                        const double tvo = (!first && p <= hio) ? 1. : 0.;
                        Do_un = tvo * (m2 * Mo_un + m8 * Do_un);
                        const double Ds = Do_un * scale;
                        const bool live = p <= hi;
                        const uint32_t rb = (uint32_t)(rw >> (8 * u)) & 0xffu;
                        double e = rb > 3 ? 1. : rb == qyi ? em.x : em.y;
                        e = live ? e : 0.;
                        const double tv = live ? 1. : 0.;
                        const double f0 = e * (m0 * Mc + m3 * Ic + m6 * Ds);
                        const double f1 = tv * (m1q * Mx + m4q * Ix);          // (past `hi` the cell above may be an earlier row's: zero, as the full sweep has it)
                        const double f2 = tv * (m2 * Mn + m8 * Dn);
                        sum += f0 + f1 + f2;
                        R[p0 + u] = make_double2(f0, f1);
                        Mo_un = Mc_un; Mc_un = Mx_un; Mc = Mx; Ic = Ix;
                        Mn = f0; Dn = f2;
                        if (NC > 4) __builtin_amdgcn_sched_barrier(0);         // (a cell's instructions stay together)
                    }
                }
            }
        }
        scale = 1. / sum;
        prod *= sum;
        if (prod < 1e-100) { Pr1 += -4.343 * log(prod); prod = 1.; }
    };
    int result = 0;
    for (int i = 2; ; ++i) {
        if (i - 1 == l_query) {       // this lane's last row is done: f[l_query + 1] over its cells inside the reference, then the score
            int hl = l_ref - (l_query - bw) + 1; hl = hl < W ? hl : W;
            double fsum = 0.;
            #pragma unroll
            for (int p = 1; p <= WC; ++p) {
                const double keep = p <= hl ? 1. : 0.;        // (a cell past it holds an earlier row's value, or a zero: it adds +0 either way)
                fsum += keep * ((R[p].x * scale) * sMc + (R[p].y * scale) * sIc);
            }
            double pr = prod * fsum, P1 = Pr1;
            if (pr < 1e-100) { P1 += -4.343 * log(pr); pr = 1.; }
            P1 += -4.343 * log(pr * l_ref * l_query);
            result = (int)(P1 + .499);
        }
        if (i > Lmax) break;
        const bool mine = i <= l_query;
        int hi = l_ref - (i - bw) + 1; hi = hi < W ? hi : W;
        int hio = l_ref - (i - 1 - bw) + 1; hio = hio < W ? hio : W;
        const int clear = wave_min_i(mine ? (hi < hio ? hi : hio) : 0x7fffffff);
        // (the groups of eight in front of the reference's first column hold zeros and stay zeros, those past `hi` of every lane are
        // zeros in this row and read by no later row: see probaln_fwd_lds)
        int lo = bw + 2 - i; lo = lo < 1 ? 1 : lo;
        const int lo_min = wave_min_i(mine ? lo : 0x7fffffff), hi_max = wave_max_i(mine ? hi : 0);
        if (hi_max < 1) continue;
        row(i, mine, hi, hio, clear, (lo_min - 1) >> 3, (hi_max - 1) >> 3);
    }
    return result;
}

// Group `grp` of the LDS class through the register-resident sweep (the groups whose bands fit it): one lane per job, both
// parameter sets one after the other.
template <int NC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void probaln_regs_kernel(const ProbalnParams P, int grp)
{
    const double2 *s_emt = P.emt;                             // (device memory, not LDS: see probaln_exact_kernel)
    __builtin_amdgcn_s_setprio(2);                            // (one wavefront a SIMD with a long chunk each: ahead of the register classes' three or four)
    ProbalnQueue *Q = P.queue;
    const uint32_t c0 = Q->lds_begin[grp], n = Q->lds_begin[grp + 1] - c0;
    if (n == 0) return;
    unsigned long long cells = 0, passes = 0;
    for (;;) {
        uint32_t base = 0;
        if (threadIdx.x == 0) base = atomicAdd(&Q->lds_next[grp], 64u);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= n) break;
        const uint32_t i = base + threadIdx.x;
        const bool have = i < n;
        uint32_t job = 0;
        PJob j{};
        if (have) { job = P.val_sorted[c0 + i]; j = P.pjob[job]; }
        const bool fits = have && 2 * (int)j.eff + 1 <= NC * 8;                // (always: the groups are cut by band width)
        const uint8_t *ref = P.ref2 + j.ref_off, *qp = P.qpack + (size_t)j.q8 * 8;
        int s1 = 0, s2 = 0;
        bool go = fits;
        #pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            const int sc = probaln_fwd_regs<NC>(ref, j.l_ref, qp, j.l_query, j.eff, s_emt, pass ? 1e-6 : 1e-4, pass ? 1e-3 : 1e-2, go);
            if (pass == 0) { s1 = s2 = sc; go = fits && sc > 5; if (__builtin_amdgcn_ballot_w64(go) == 0) break; }
            else if (go) s2 = sc;
        }
        if (fits) {
            auto pack = [&](int sc) { int l = (int)(100. * sc / j.l_query + .499); if (l > 255) l = 255; return sc << 8 | l; };
            P.score1[job] = pack(s1);
            P.score2[job] = pack(s2);
            const unsigned long long c1 = (unsigned long long)j.l_query * (2 * j.eff + 1) * 3;
            cells += go ? 2 * c1 : c1;
            passes += go ? 2 : 1;
        }
    }
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) { passes += __shfl_xor(passes, o); cells += __shfl_xor(cells, o); }
    if (threadIdx.x == 0 && passes) { atomicAdd(&P.tot->n_passes, passes); atomicAdd(&P.tot->dp_cells, cells); }
}

// Group `grp` of the LDS class (LS = 64), or the sixteen-jobs class (LS = 16, grp = PROBALN_LDS_GROUPS): both parameter sets of a
// job in the lane that holds it (nearly every job of a realigned column scores above 5 with the first).  wcells: the cells
// the launch's LDS holds per lane.
template <int LS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void probaln_lds_kernel(const ProbalnParams P, int grp, int wcells)
{
    extern __shared__ double2 s_lds[];                     // (M', I')[wcells][LS], then the emission table
    // These wavefronts are few and each has a long way to go alone (one or two a CU), beside launches that fill every SIMD with
    // three or four wavefronts of their own: at equal priority a wavefront here would get a quarter of its SIMD's issue slots and
    // the whole stage would wait for it.
    __builtin_amdgcn_s_setprio(3);
    double2 *s_emt = s_lds + (size_t)wcells * LS;
    ProbalnQueue *Q = P.queue;
    const uint32_t c0 = Q->lds_begin[grp], n = Q->lds_begin[grp + 1] - c0;
    if (n == 0) return;
    for (int b = threadIdx.x; b < 256; b += 64) {
        const double ql = (double)P.q2p[b >> 3];
        s_emt[b] = (b & 7) > 3 ? make_double2(1., 1.) : make_double2(1. - ql, ql * EM);
    }
    __syncthreads();
    const bool owns = threadIdx.x < LS;                    // (LS = 16: the other lanes hold no job and own no column)
    double2 *sR = s_lds + (owns ? threadIdx.x : 0);
    unsigned long long cells = 0, passes = 0;
    for (;;) {
        uint32_t base = 0;
        if (threadIdx.x == 0) base = atomicAdd(&Q->lds_next[grp], (uint32_t)LS);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= n) break;
        const uint32_t i = base + threadIdx.x;
        const bool have = threadIdx.x < LS && i < n;
        uint32_t job = 0;
        PJob j{};
        if (have) { job = P.val_sorted[c0 + i]; j = P.pjob[job]; }
        const bool fits = have && PROBALN_LDS_CELLS((int)j.eff) <= wcells;     // (always: the groups are cut by band width)
        const uint8_t *ref = P.ref2 + j.ref_off, *qp = P.qpack + (size_t)j.q8 * 8;
        int s1 = 0, s2 = 0;
        bool again = fits;
        #pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            const int sc = probaln_fwd_lds<LS>(ref, j.l_ref, qp, j.l_query, j.eff, s_emt, pass ? 1e-6 : 1e-4, pass ? 1e-3 : 1e-2, sR, again, owns);
            if (pass == 0) { s1 = s2 = sc; again = fits && sc > 5; if (__builtin_amdgcn_ballot_w64(again) == 0) break; }
            else if (again) s2 = sc;
        }
        if (fits) {
            auto pack = [&](int sc) { int l = (int)(100. * sc / j.l_query + .499); if (l > 255) l = 255; return sc << 8 | l; };
            P.score1[job] = pack(s1);
            P.score2[job] = pack(s2);
            const unsigned long long c1 = (unsigned long long)j.l_query * (2 * j.eff + 1) * 3;
            cells += again ? 2 * c1 : c1;
            passes += again ? 2 : 1;
        }
    }
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) { passes += __shfl_xor(passes, o); cells += __shfl_xor(cells, o); }
    if (threadIdx.x == 0 && passes) { atomicAdd(&P.tot->n_passes, passes); atomicAdd(&P.tot->dp_cells, cells); }
}

// The listed jobs with bands wider than PROBALN_LDS16_MAX (or lengths past 16 bits): two rolling rows per job in the scratch buffer, both parameter sets.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void probaln_wide_kernel(const ProbalnParams P)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    unsigned long long passes = 0, cells = 0;
    if (i < P.wide_count) {
        const uint32_t job = P.wide[P.wide_first + i];
        const JobDesc j = decode_job(P, job);
        const size_t stride = P.scratch_stride;
        double *row0 = P.scratch + i, *row1 = P.scratch + (size_t)P.ncell * stride + i;
        int s1 = 0, s2 = 0;
        double gd = 1e-4, ge = 1e-2;
        #pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            const int sc = probaln_fwd(j.ref, j.l_ref, j.qs, j.l_query, P.q2p, gd, ge, j.bw, row0, row1, stride, P.ncell);
            int l = (int)(100. * sc / j.l_query + .499);
            if (l > 255) l = 255;
            const int v = sc << 8 | l;
            ++passes;
            if (pass == 0) { s1 = s2 = v; if (sc <= 5) break; gd = 1e-6; ge = 1e-3; }
            else s2 = v;
        }
        P.score1[job] = s1;
        P.score2[job] = s2;
        cells = (unsigned long long)j.l_query * (2 * j.eff + 1) * 3 * passes;
    }
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) { passes += __shfl_xor(passes, o); cells += __shfl_xor(cells, o); }
    if (threadIdx.x == 0 && passes) { atomicAdd(&P.tot->n_passes, passes); atomicAdd(&P.tot->dp_cells, cells); }
}

// the emission values of a (query base, capped quality) byte: (match, mismatch) = (1 - 10^(-q/10), 10^(-q/10) / 3); an N in the read: (1, 1)
__global__ void probaln_emt_kernel(const float *q2p, double2 *emt)
{
    const int b = threadIdx.x;
    const double ql = (double)q2p[b >> 3];
    emt[b] = (b & 7) > 3 ? make_double2(1., 1.) : make_double2(1. - ql, ql * EM);
}
void launch_probaln_jobs(const ProbalnParams &p, hipStream_t s)
{
    hipLaunchKernelGGL(probaln_emt_kernel, dim3(1), dim3(256), 0, s, p.q2p, p.emt);
    if (p.n_jobs > 0) hipLaunchKernelGGL(probaln_jobs_kernel, dim3((p.n_jobs + 255) / 256), dim3(256), 0, s, p);
}
void launch_probaln_bounds(const ProbalnParams &p, hipStream_t s)
{
    hipLaunchKernelGGL(probaln_bounds_kernel, dim3(1), dim3(64), 0, s, p.key_sorted, p.n_jobs, p.queue);
}
template <int BW> static void launch_exact_class(const ProbalnParams &p, hipStream_t s, unsigned grid)
{
    hipLaunchKernelGGL((probaln_exact_kernel<BW, 1>), dim3(grid), dim3(64), 0, s, p);
    hipLaunchKernelGGL((probaln_exact_kernel<BW, 2>), dim3(grid), dim3(64), 0, s, p);
}
// Both passes of every band width, on four streams between a fork and a join on the caller's (side[0..2], ev[0..2] and ev[8]
// from the context; the runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues -- api.hip raises it to
// eight when the library is loaded: on the default four, two of these streams shared a queue and ran one after the other;
// the three register-row classes on a fifth stream of their own, beside both chains, was slower: profiles/r5_hw_queues.txt).  Every launch is a grid of persistent wavefronts pulling 64 jobs at a time from its class's counter, so a class with
// few jobs costs one such chunk, not a launch's worth of idle machine; the two chains of machine-filling launches share the
// CUs wavefront by wavefront.  The classes whose wavefronts are few and slow -- the bands past 43, one or two wavefronts a CU
// in LDS -- go first on a stream of their own and run beside everything else.
int launch_probaln_exact(const ProbalnParams &p, hipStream_t s, int n_cu, hipStream_t *side, hipEvent_t *ev)
{
    if (p.n_jobs <= 0) return 0;
    // wavefronts the machine holds at the kernels' occupancy (2-3 per SIMD), but no more than the jobs can fill
    unsigned grid = (unsigned)n_cu * 4u * 3u;
    const unsigned need = (unsigned)((p.n_jobs + 63) / 64);
    if (grid > need) grid = need;
    hipStream_t tail = side[0], second = side[1], tail2 = side[2];
    if (hipEventRecord(ev[8], s) != hipSuccess) return -1;
    if (hipStreamWaitEvent(tail, ev[8], 0) != hipSuccess || hipStreamWaitEvent(second, ev[8], 0) != hipSuccess ||
        hipStreamWaitEvent(tail2, ev[8], 0) != hipSuccess) return -1;
    {
        // (always launched: a job's band is |type| + 3 OR the difference of its two lengths, so a column of 1-3 base types has
        // wide-band jobs as soon as one of its reads carries a long indel elsewhere in the window; an empty class costs a launch
        // of workgroups that leave at once)
        static const int caps[PROBALN_LDS_GROUPS] = PROBALN_LDS_CAPS;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(probaln_lds_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(probaln_lds_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
        // the sixteen-jobs class: bands to PROBALN_LDS16_SPLIT (the bulk: 58 KB, two workgroups a CU) on one stream, the rest behind it
        for (int h = 0; h < 2; ++h) {
            const int wcells = PROBALN_LDS_CELLS(h == 0 ? PROBALN_LDS16_SPLIT : PROBALN_LDS16_MAX);
            const size_t lds = (size_t)wcells * 16 * 16 + 256 * sizeof(double2);
            const unsigned per_cu = (unsigned)((160 * 1024) / lds);
            hipLaunchKernelGGL(probaln_lds_kernel<16>, dim3(std::min((unsigned)n_cu * per_cu, need)), dim3(64), lds, tail, p, PROBALN_LDS_GROUPS + h, wcells);
        }
        // bands 44 .. 73, 64 jobs a wavefront, on a stream of their own beside the sixteen-jobs class
        for (int g = PROBALN_LDS_GROUPS - 1; g >= 3; --g) {
            const int wcells = PROBALN_LDS_CELLS(caps[g]);
            const size_t lds = (size_t)wcells * 64 * 16 + 256 * sizeof(double2);
            const unsigned per_cu = (unsigned)((160 * 1024) / lds);
            hipLaunchKernelGGL(probaln_lds_kernel<64>, dim3(std::min((unsigned)n_cu * per_cu, need)), dim3(64), lds, tail2, p, g, wcells);
        }
        // bands 11 .. 43: the row in registers, one wavefront a SIMD
        hipLaunchKernelGGL(probaln_regs_kernel<11>, dim3(std::min((unsigned)n_cu * 4u, need)), dim3(64), 0, second, p, 2);
        hipLaunchKernelGGL(probaln_regs_kernel<8>, dim3(std::min((unsigned)n_cu * 4u, need)), dim3(64), 0, s, p, 1);
        hipLaunchKernelGGL(probaln_regs_kernel<4>, dim3(std::min((unsigned)n_cu * 4u, need)), dim3(64), 0, second, p, 0);
    }
    // (other layouts measured with eight hardware queues, none better: every narrow band in one chain, in either order; the register-row
    // classes on a stream of their own beside one or two chains; two chains of two wavefronts a SIMD: profiles/r5_hw_queues.txt)
    launch_exact_class<10>(p, second, grid); launch_exact_class<9>(p, s, grid);
    launch_exact_class<8>(p, second, grid);  launch_exact_class<7>(p, s, grid);
    launch_exact_class<6>(p, second, grid);  launch_exact_class<5>(p, s, grid);
    launch_exact_class<4>(p, second, grid);  launch_exact_class<3>(p, s, grid);
    if (hipEventRecord(ev[0], tail) != hipSuccess || hipStreamWaitEvent(s, ev[0], 0) != hipSuccess) return -1;
    if (hipEventRecord(ev[1], second) != hipSuccess || hipStreamWaitEvent(s, ev[1], 0) != hipSuccess) return -1;
    if (hipEventRecord(ev[2], tail2) != hipSuccess || hipStreamWaitEvent(s, ev[2], 0) != hipSuccess) return -1;
    return 0;
}
void launch_probaln_wide(const ProbalnParams &p, hipStream_t s)
{
    if (p.wide_count > 0) hipLaunchKernelGGL(probaln_wide_kernel, dim3((p.wide_count + 63) / 64), dim3(64), 0, s, p);
}

}  // namespace bcfgpu

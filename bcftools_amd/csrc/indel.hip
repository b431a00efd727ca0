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
        if (j.eff > PROBALN_BW_MAX || j.l_ref > 65535 || j.l_query > 65535 || P.force_wide) {
            P.wide[atomicAdd(&P.tot->n_wide, 1u)] = job;
            atomicMax(&P.tot->max_eff, j.eff);
            cls = PROBALN_CLS_WIDE;
        } else {
            cls = j.eff < PROBALN_BW_MIN ? PROBALN_BW_MIN : j.eff;
            pj.ref_off = j.ref_off; pj.q8 = j.q8; pj.l_ref = (uint16_t)j.l_ref; pj.l_query = (uint16_t)j.l_query; pj.eff = (uint16_t)j.eff;
        }
    }
    int dl = j.l_ref - j.l_query + 16;
    dl = dl < 0 ? 0 : dl > 31 ? 31 : dl;
    P.pjob[job] = pj;
    P.key_in[job] = cls << 13 | (uint32_t)(j.l_query > 255 ? 255 : j.l_query < 0 ? 0 : j.l_query) << 5 | (uint32_t)dl;
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
}

// PASS 1: the jobs of class BW in sorted order, parameter set {1e-4, 1e-2}; jobs scoring above 5 are listed for
// PASS 2 (the second parameter set {1e-6, 1e-3}, bam2bcf_indel.c:293-294, 346-356).  One wavefront per workgroup; the
// grid is sized for the machine, not for the class: a wavefront takes the next 64 jobs of its class from a counter
// until none are left (every wavefront reaches that test, also when the class is empty).
template <int BW, int PASS>
__global__ __launch_bounds__(64) void probaln_exact_kernel(const ProbalnParams P)
{
    __shared__ double2 s_emt[256];
    ProbalnQueue *Q = P.queue;
    const uint32_t c0 = Q->cls_begin[BW];
    const uint32_t n = PASS == 1 ? Q->cls_begin[BW + 1] - c0 : Q->n2[BW];
    if (n == 0) return;                                   // (the same for every wavefront of the launch)
    for (int b = threadIdx.x; b < 256; b += 64) {
        const double ql = (double)P.q2p[b >> 3];
        s_emt[b] = (b & 7) > 3 ? make_double2(1., 1.) : make_double2(1. - ql, ql * EM);
    }
    __syncthreads();
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

// The listed jobs with bands wider than PROBALN_BW_MAX: two rolling rows per job in the scratch buffer, both parameter sets.
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

void launch_probaln_jobs(const ProbalnParams &p, hipStream_t s)
{
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
// Both passes of every band width.  The widths are independent of one another, and the narrow classes hold most jobs while
// a wide one may hold a few dozen whose single wavefront takes as long as a large class does: each width runs on a stream
// of its own between a fork and a join on the caller's stream (side[8], ev[9] from the context), widest first.
int launch_probaln_exact(const ProbalnParams &p, hipStream_t s, int n_cu, hipStream_t *side, hipEvent_t *ev)
{
    if (p.n_jobs <= 0) return 0;
    // wavefronts the machine holds at the kernels' occupancy (2-3 per SIMD), but no more than the jobs can fill
    unsigned grid = (unsigned)n_cu * 4u * 3u;
    const unsigned need = (unsigned)((p.n_jobs + 63) / 64);
    if (grid > need) grid = need;
    if (hipEventRecord(ev[8], s) != hipSuccess) return -1;
    for (int i = 0; i < 8; ++i) {
        if (hipStreamWaitEvent(side[i], ev[8], 0) != hipSuccess) return -1;
        switch (i) {
            case 0: launch_exact_class<10>(p, side[i], grid); break;
            case 1: launch_exact_class<9>(p, side[i], grid); break;
            case 2: launch_exact_class<8>(p, side[i], grid); break;
            case 3: launch_exact_class<7>(p, side[i], grid); break;
            case 4: launch_exact_class<6>(p, side[i], grid); break;
            case 5: launch_exact_class<5>(p, side[i], grid); break;
            case 6: launch_exact_class<4>(p, side[i], grid); break;
            default: launch_exact_class<3>(p, side[i], grid); break;
        }
        if (hipEventRecord(ev[i], side[i]) != hipSuccess || hipStreamWaitEvent(s, ev[i], 0) != hipSuccess) return -1;
    }
    return 0;
}
void launch_probaln_wide(const ProbalnParams &p, hipStream_t s)
{
    if (p.wide_count > 0) hipLaunchKernelGGL(probaln_wide_kernel, dim3((p.wide_count + 63) / 64), dim3(64), 0, s, p);
}

}  // namespace bcfgpu

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

// The same forward pass with the scaled row held in registers, for bands of half-width <= BWM (|indel| <= BWM-3: the
// common case).  A row has 2*bw+1 live cells x 3 states; position p of the register row is reference column
// k = p + x - 1 with x = max(0, i - bw), exactly the slot set_u() assigns in the rolling-row version, so row i is
// computed in place from row i-1: while the band hugs the left edge (i <= bw) the diagonal neighbour of p is p-1,
// afterwards the band slides by one column per row and it is p itself.  The reference window travels in one 64-bit
// register (3 bits per base, shifted as the band slides).  No scratch memory: the rolling-row version moves
// 48 bytes per cell through HBM and is bandwidth-bound; this one is bound by fp64 issue.
template <int BWM>
__device__ int probaln_fwd_reg(const uint8_t *ref, int l_ref, const QSrc qs, int l_query,
                               const float *q2p, double d, double e_, int bw)
{
    constexpr int NP = 2 * BWM + 3;                       // positions 0 .. 2*BWM+2
    double M[NP], I[NP], D[NP];
    #pragma unroll
    for (int p = 0; p < NP; ++p) M[p] = I[p] = D[p] = 0.;
    const int bw2 = bw * 2 + 1;
    double m[9];
    const double sM = 1. / (2 * l_query + 2), sI = sM;
    m[0] = (1 - d - d) * (1 - sM); m[1] = m[2] = d * (1 - sM);
    m[3] = (1 - e_) * (1 - sI); m[4] = e_ * (1 - sI); m[5] = 0.;
    m[6] = 1 - e_; m[7] = 0.; m[8] = e_;
    const double bM = (1 - d) / l_ref, bI = d / l_ref;
    // reference window: base of position p (column k = p + x - 1) in bits [3p, 3p+3); x = 0 to begin with
    uint64_t rw = 0;
    #pragma unroll
    for (int p = 2; p < NP; ++p) rw |= (uint64_t)(p - 2 < l_ref ? ref[p - 2] : 4) << (3 * p);
    double prod = 1., Pr1 = 0.;
    // f[1]
    {
        double sum = 0.;
        const int end = l_ref < bw + 1 ? l_ref : bw + 1;
        const double q0 = (double)q2p[qs.q(0)];
        const int qy = qs.base(0);
        #pragma unroll
        for (int p = 2; p < NP; ++p) {
            if (p - 1 <= end) {
                const int rb = (int)((rw >> (3 * p)) & 7);
                const double e = (rb > 3 || qy > 3) ? 1. : rb == qy ? 1. - q0 : q0 * EM;
                const double a = e * bM, b = EI * bI;
                M[p] = a; I[p] = b;
                sum += a + b;
            }
        }
        #pragma unroll
        for (int p = 2; p < NP; ++p)
            if (p - 1 <= end) { M[p] /= sum; I[p] /= sum; D[p] /= sum; }
        prod *= sum;
        if (prod < 1e-100) { Pr1 += -4.343 * log(prod); prod = 1.; }
    }
    int x = 0;                                            // first column of the band minus one: max(0, i - bw)
    // one row; `slide`: x grows by one on this row (i > bw).  Two instantiations: the rows before and after the band leaves
    // the left edge differ in which old cells are the diagonal and the upper neighbour, and a run-time choice costs ten
    // selects per position
    auto row = [&](auto slide_c, int i) {
        constexpr bool slide = decltype(slide_c)::value;
        const double qli = (double)q2p[qs.q(i - 1)];
        const int qyi = qs.base(i - 1);
        if (slide) {
            ++x;
            rw >>= 3;
            const int nk = (NP - 1) + x - 2;              // reference index of the new top position
            rw |= (uint64_t)(nk < l_ref ? ref[nk] : 4) << (3 * (NP - 1));
        }
        const int end = l_ref < i + bw ? l_ref : i + bw;
        const int plo = x == 0 ? 2 : 1, phi = end - x + 1;
        double sum = 0.;
        // in place, ascending p: the left neighbour is the new [p-1]; the diagonal one is the old [p] once the band
        // slides, the old [p-1] (carried) before; the upper one is the old [p+1] resp. the old [p]
        double cM = M[0], cI = I[0], cD = D[0];           // old [p-1]
        M[0] = I[0] = D[0] = 0.;
        #pragma unroll
        for (int p = 1; p < NP; ++p) {
            const double oM = M[p], oI = I[p], oD = D[p];
            const double nM = p + 1 < NP ? M[p + 1 < NP ? p + 1 : p] : 0., nI = p + 1 < NP ? I[p + 1 < NP ? p + 1 : p] : 0.;
            double f0 = 0., f1 = 0., f2 = 0.;
            if (p >= plo && p <= phi) {
                const int rb = (int)((rw >> (3 * p)) & 7);
                const double e = (rb > 3 || qyi > 3) ? 1. : rb == qyi ? 1. - qli : qli * EM;
                const double gM = slide ? oM : cM, gI = slide ? oI : cI, gD = slide ? oD : cD;
                const double uM = slide ? nM : oM, uI = slide ? nI : oI;
                f0 = e * (m[0] * gM + m[3] * gI + m[6] * gD);
                f1 = EI * (m[1] * uM + m[4] * uI);
                f2 = m[2] * M[p - 1] + m[8] * D[p - 1];
                sum += f0 + f1 + f2;
            }
            M[p] = f0; I[p] = f1; D[p] = f2;
            cM = oM; cI = oI; cD = oD;
        }
        const double r = 1. / sum;
        #pragma unroll
        for (int p = 1; p < NP; ++p) { M[p] *= r; I[p] *= r; D[p] *= r; }
        prod *= sum;
        if (prod < 1e-100) { Pr1 += -4.343 * log(prod); prod = 1.; }
    };
    {
        int i = 2;
        for (; i <= l_query && i <= bw; ++i) row(std::false_type{}, i);
        for (; i <= l_query; ++i) row(std::true_type{}, i);
    }
    {   // f[l_query+1]: columns k = 1..l_ref whose slot lies inside the band of the last row
        double sum = 0.;
        const int phi = l_ref - x + 1 < bw2 ? l_ref - x + 1 : bw2;
        const int plo = x == 0 ? 2 : 1;
        #pragma unroll
        for (int p = 1; p < NP; ++p)
            if (p >= plo && p <= phi) sum += M[p] * sM + I[p] * sI;
        prod *= sum;
        if (prod < 1e-100) { Pr1 += -4.343 * log(prod); prod = 1.; }
    }
    Pr1 += -4.343 * log(prod * l_ref * l_query);
    return (int)(Pr1 + .499);
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

// One realignment of bam2bcf_indel.c:313-357: read K of the site against candidate type t.  Jobs are numbered
// job0 + t*N + K inside a site (neighbouring lanes: neighbouring reads of one type, i.e. the same band and window).
// What a read contributes to every realignment of its site (it does not depend on the candidate type): the part of the
// read inside the window and where it starts and ends on the reference -- the two tpos2qpos() calls of
// bam2bcf_indel.c:326-327 -- and the read's sample.  One lane per pileup entry.
__global__ __launch_bounds__(256) void gap_entry_kernel(const GapIn in, const GapSite *sites, int n_ent, GapEntry *ent)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_ent) return;
    int lo = 0, hi = in.n_sites - 1;                        // the site of entry e: the last one with smpl_off[site*n] <= e
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (in.smpl_off[(size_t)mid * in.n_smpl] <= e) lo = mid; else hi = mid - 1; }
    const GapSite &S = sites[lo];
    if (!S.live) return;
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
    if (!skip) {
        g.qbeg = gap_tpos2qpos(in.r_pos[r], ncig, cigar, S.left, 0, &g.tbeg);
        g.qend = gap_tpos2qpos(in.r_pos[r], ncig, cigar, S.right, 1, &g.tend);
    }
    ent[e] = g;
}
void launch_gap_entries(const GapIn &in, const GapSite *sites, int n_ent, GapEntry *ent, hipStream_t s)
{
    if (n_ent > 0) hipLaunchKernelGGL(gap_entry_kernel, dim3((n_ent + 255) / 256), dim3(256), 0, s, in, sites, n_ent, ent);
}

// One realignment of bam2bcf_indel.c:313-357: read K of the site against candidate type t.  Jobs are numbered
// job0 + t*N + K inside a site (neighbouring lanes: neighbouring reads of one type, i.e. the same band and window).
struct JobDesc { const uint8_t *ref; int l_ref, l_query, bw, eff; QSrc qs; bool skip; };
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
    d.ref = P.ref2 + (size_t)S.ref2_0 + ((size_t)t * in.n_smpl + g.smpl) * S.max_ref2 + (tbeg - S.left);
    d.l_ref = g.tend - tbeg + aty;
    d.l_query = g.qend - g.qbeg;
    d.bw = aty + 3;
    // the band probaln_glocal really uses (probaln.c): min(bw, max(l_ref, l_query)), at least |l_ref - l_query|
    int eff = d.l_ref > d.l_query ? d.l_ref : d.l_query;
    if (eff > d.bw) eff = d.bw;
    if (eff < abs(d.l_ref - d.l_query)) eff = abs(d.l_ref - d.l_query);
    d.eff = eff;
    const size_t qo = (size_t)in.r_seq_off[r] + g.qbeg;
    d.qs = QSrc{in.seq16 + qo, in.qual + qo, (in.zq && in.r_has_zq && in.r_has_zq[r]) ? in.zq + qo : nullptr};
    return d;
}

#define PROBALN_BWM 6
// WIDE = false: every job; bands up to PROBALN_BWM run here with the row in registers, the others are listed.
// WIDE = true: the listed jobs, two rolling rows per job in the scratch buffer.
template <bool WIDE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void probaln_kernel(const ProbalnParams P)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    uint32_t job = 0;
    bool have;
    if (WIDE) { have = i < P.wide_count; if (have) job = P.wide[P.wide_first + i]; }
    else { have = i < P.n_jobs; job = (uint32_t)i; }
    unsigned long long passes = 0, cells = 0;
    if (have) {
        const JobDesc j = decode_job(P, job);
        bool run = !j.skip;
        const bool degenerate = j.l_ref <= 0 || j.l_query <= 0;           // probaln_glocal has nothing to align: score 0
        if (!WIDE && run && !degenerate && (j.eff > PROBALN_BWM || P.force_scratch)) {
            P.wide[atomicAdd(&P.tot->n_wide, 1u)] = job;
            atomicMax(&P.tot->max_eff, j.eff);
            run = false;
        }
        if (run) {
            const size_t stride = P.scratch_stride;
            double *row0 = WIDE ? P.scratch + i : nullptr, *row1 = WIDE ? P.scratch + (size_t)P.ncell * stride + i : nullptr;
            // apf1 = {1e-4, 1e-2, bw}; a second parameter set apf2 = {1e-6, 1e-3, bw} is tried when the first score exceeds 5
            // (bam2bcf_indel.c:293-294, 346-356)
            int s1 = 0, s2 = 0;
            double gd = 1e-4, ge = 1e-2;
            #pragma unroll 1
            for (int pass = 0; pass < 2; ++pass) {
                int sc;
                if (degenerate) sc = 0;
                else if (WIDE) sc = probaln_fwd(j.ref, j.l_ref, j.qs, j.l_query, P.q2p, gd, ge, j.bw, row0, row1, stride, P.ncell);
                else sc = probaln_fwd_reg<PROBALN_BWM>(j.ref, j.l_ref, j.qs, j.l_query, P.q2p, gd, ge, j.eff);
                int l = (int)(100. * sc / j.l_query + .499);
                if (l > 255) l = 255;
                const int v = sc << 8 | l;
                ++passes;
                if (pass == 0) { s1 = s2 = v; if (sc <= 5) break; gd = 1e-6; ge = 1e-3; }
                else s2 = v;
            }
            P.score1[job] = s1;
            P.score2[job] = s2;
            if (!degenerate) cells = (unsigned long long)j.l_query * (2 * j.eff + 1) * 3 * passes;
        }
    }
    // statistics: one atomic per wavefront
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) { passes += __shfl_xor(passes, o); cells += __shfl_xor(cells, o); }
    if (threadIdx.x == 0 && passes) { atomicAdd(&P.tot->n_passes, passes); atomicAdd(&P.tot->dp_cells, cells); }
}

void launch_probaln(const ProbalnParams &p, hipStream_t s, bool wide_pass)
{
    if (wide_pass) { if (p.wide_count > 0) hipLaunchKernelGGL(probaln_kernel<true>, dim3((p.wide_count + 63) / 64), dim3(64), 0, s, p); }
    else if (p.n_jobs > 0) hipLaunchKernelGGL(probaln_kernel<false>, dim3((p.n_jobs + 63) / 64), dim3(64), 0, s, p);
}

}  // namespace bcfgpu

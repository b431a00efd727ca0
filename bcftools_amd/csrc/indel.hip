// indel.hip -- the realignment scores of bcf_call_gap_prep: probaln_glocal() forward pass for every
// (site, indel type, read) job of a batch (bam2bcf_indel.c:291-370; "this is the bottleneck", :335).
//
// One lane per job: the banded 3-state pair-HMM is a chain of dependent rows, and the jobs of a site share window
// and band, so 64 neighbouring jobs run in lockstep.  Each lane keeps two rolling rows of the scaled forward matrix
// in a scratch buffer laid out [cell][job] (coalesced across lanes).  All arithmetic is fp64 in the reference's
// order (htslib probaln.c), so the integer scores match the CPU path.
#include <hip/hip_runtime.h>
#include <math.h>
#include "kernels.h"

namespace bcfgpu {

#define EI .25
#define EM .33333333333

__device__ __forceinline__ int set_u(int b, int i, int k) { int x = i - b; x = x > 0 ? x : 0; return (k - x + 1) * 3; }

// forward score of one job with gap-open d / gap-ext e (probaln_par_t {d, e, bw})
__device__ int probaln_fwd(const uint8_t *ref, int l_ref, const uint8_t *query, int l_query, const uint8_t *iqual,
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
        const double q0 = (double)q2p[iqual[0]];
        for (int k = beg; k <= end; ++k) {
            const double e = (ref[k - 1] > 3 || query[0] > 3) ? 1. : ref[k - 1] == query[0] ? 1. - q0 : q0 * EM;
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
        const double qli = (double)q2p[iqual[i - 1]];
        const uint8_t qyi = query[i - 1];
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

__global__ __launch_bounds__(64) void probaln_kernel(const ProbalnParams P)
{
    const int job = blockIdx.x * 64 + threadIdx.x;
    if (job >= P.n_jobs) return;
    const ProbalnJob j = P.jobs[job];
    const size_t stride = P.scratch_stride;
    double *row0 = P.scratch + job, *row1 = P.scratch + (size_t)P.ncell * stride + job;
    const uint8_t *ref = P.ref2 + j.ref_off, *query = P.query + j.query_off, *qq = P.qq + j.query_off;
    // apf1 = {1e-4, 1e-2, bw}, apf2 = {1e-6, 1e-3, bw}  (bam2bcf_indel.c:293-294)
    int sc = probaln_fwd(ref, j.l_ref, query, j.l_query, qq, P.q2p, 1e-4, 1e-2, j.bw, row0, row1, stride, P.ncell);
    int l = (int)(100. * sc / j.l_query + .499);
    if (l > 255) l = 255;
    int s1 = sc << 8 | l, s2 = s1;
    if (sc > 5) {
        sc = probaln_fwd(ref, j.l_ref, query, j.l_query, qq, P.q2p, 1e-6, 1e-3, j.bw, row0, row1, stride, P.ncell);
        l = (int)(100. * sc / j.l_query + .499);
        if (l > 255) l = 255;
        s2 = sc << 8 | l;
    }
    P.score1[job] = s1;
    P.score2[job] = s2;
}

void launch_probaln(const ProbalnParams &p, hipStream_t s)
{
    if (p.n_jobs == 0) return;
    hipLaunchKernelGGL(probaln_kernel, dim3((p.n_jobs + 63) / 64), dim3(64), 0, s, p);
}

}  // namespace bcfgpu

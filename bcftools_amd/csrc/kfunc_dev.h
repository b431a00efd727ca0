// kfunc_dev.h -- device restatements of the htslib kfunc.c routines the path needs outside of one kernel:
// kf_lgamma, kt_fisher_exact (FMT/SP in combine_kernel, PV4 in mcall_kernel), kf_betai and bcftools' own test16 (PV4).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace bcfgpu {

// kf_lgamma and kt_fisher_exact (htslib kfunc.c), for FMT/SP (bam2bcf.c:867-885): the two-sided Fisher exact test of a
// sample's DP4 table, with kfunc's incremental walk over the hypergeometric terms (re-anchored through lgamma at every
// n11 divisible by 11, as there) so that the rounded Phred value matches.
__device__ inline double dev_kf_lgamma(double z)
{
    double x = 0;
    x += 0.1659470187408462e-06 / (z + 7);
    x += 0.9934937113930748e-05 / (z + 6);
    x -= 0.1385710331296526     / (z + 5);
    x += 12.50734324009056      / (z + 4);
    x -= 176.6150291498386      / (z + 3);
    x += 771.3234287757674      / (z + 2);
    x -= 1259.139216722289      / (z + 1);
    x += 676.5203681218835      / z;
    x += 0.9999999999995183;
    return log(x) - 5.58106146679532777 - z + (z - 0.5) * log(z + 6.5);
}
__device__ __forceinline__ double dev_lbinom(int n, int k)
{
    if (k == 0 || n == k) return 0;
    return dev_kf_lgamma(n + 1) - dev_kf_lgamma(k + 1) - dev_kf_lgamma(n - k + 1);
}
struct HgAcc { int n11, n1_, n_1, n; double p; };
__device__ __forceinline__ double dev_hypergeo(const HgAcc &a)
{
    return exp(dev_lbinom(a.n1_, a.n11) + dev_lbinom(a.n - a.n1_, a.n_1 - a.n11) - dev_lbinom(a.n, a.n_1));
}
// the term for a new n11 (only n11 changes): one multiplication from the neighbouring term where kfunc does that
__device__ inline double dev_hypergeo_step(int n11, HgAcc &a)
{
    if (n11 % 11 && n11 + a.n - a.n1_ - a.n_1) {
        if (n11 == a.n11 + 1) {
            a.p *= (double)(a.n1_ - a.n11) / n11 * (a.n_1 - a.n11) / (n11 + a.n - a.n1_ - a.n_1);
            a.n11 = n11;
            return a.p;
        }
        if (n11 == a.n11 - 1) {
            a.p *= (double)a.n11 / (a.n1_ - n11) * (a.n11 + a.n - a.n1_ - a.n_1) / (a.n_1 - n11);
            a.n11 = n11;
            return a.p;
        }
    }
    a.n11 = n11;
    a.p = dev_hypergeo(a);
    return a.p;
}
__device__ inline double dev_fisher_two_sided(int n11, int n12, int n21, int n22)
{
    const int n1_ = n11 + n12, n_1 = n11 + n21, n = n11 + n12 + n21 + n22;
    const int mx = n_1 < n1_ ? n_1 : n1_;
    int mn = n1_ + n_1 - n;
    if (mn < 0) mn = 0;
    if (mn == mx) return 1.;
    HgAcc a; a.n11 = n11; a.n1_ = n1_; a.n_1 = n_1; a.n = n;
    const double q = a.p = dev_hypergeo(a);
    int i, j;
    double left, right;
    double p = dev_hypergeo_step(mn, a);
    for (left = 0., i = mn + 1; p < 0.99999999 * q && i <= mx; ++i) { left += p; p = dev_hypergeo_step(i, a); }
    --i;
    if (p < 1.00000001 * q) left += p;
    p = dev_hypergeo_step(mx, a);
    for (right = 0., j = mx - 1; p < 0.99999999 * q && j >= 0; --j) { right += p; p = dev_hypergeo_step(j, a); }
    if (p < 1.00000001 * q) right += p;
    const double two = left + right;
    return two > 1. ? 1. : two;
}

// kf_betai (htslib kfunc.c): regularised incomplete beta function, continued fraction by the modified Lentz algorithm
__device__ inline double dev_kf_betai_aux(double a, double b, double x)
{
    if (x == 0.) return 0.;
    if (x == 1.) return 1.;
    double f = 1., C = f, D = 0.;
    for (int j = 1; j < 200; ++j) {
        const int m = j >> 1;
        const double aa = (j & 1) ? -(a + m) * (a + b + m) * x / ((a + 2 * m) * (a + 2 * m + 1))
                                  : m * (b - m) * x / ((a + 2 * m - 1) * (a + 2 * m));
        D = 1. + aa * D;
        if (D < 1e-290) D = 1e-290;
        C = 1. + aa / C;
        if (C < 1e-290) C = 1e-290;
        D = 1. / D;
        const double d = C * D;
        f *= d;
        if (fabs(d - 1.) < 1e-14) break;
    }
    return exp(dev_kf_lgamma(a + b) - dev_kf_lgamma(a) - dev_kf_lgamma(b) + a * log(x) + b * log(1. - x)) / a / f;
}
__device__ inline double dev_kf_betai(double a, double b, double x)
{
    return x < (a + 1.) / (a + b + 2.) ? dev_kf_betai_aux(a, b, x) : 1. - dev_kf_betai_aux(b, a, 1. - x);
}
// ttest, ccall.c:89-101
__device__ inline double dev_ttest(int n1, int n2, const float *a)
{
    if (n1 == 0 || n2 == 0 || n1 + n2 < 3) return 1.0;
    const double u1 = (double)a[0] / n1, u2 = (double)a[2] / n2;
    if (u1 <= u2) return 1.;
    const double t = (u1 - u2) / sqrt(((a[1] - n1 * u1 * u1) + (a[3] - n2 * u2 * u2)) / (n1 + n2 - 2) * (1. / n1 + 1. / n2));
    const double v = n1 + n2 - 2;
    return t < 0. ? 1. : .5 * dev_kf_betai(.5 * v, .5, v / (v + t * t));
}
// test16, ccall.c:103-138, test k of the four (0: strand bias by Fisher's exact test, 1..3: baseQ, mapQ, tail distance)
__device__ inline double dev_test16_one(const float *anno, int k)
{
    if (k == 0) return dev_fisher_two_sided((int)anno[0], (int)anno[1], (int)anno[2], (int)anno[3]);
    return dev_ttest((int)(anno[0] + anno[1]), (int)(anno[2] + anno[3]), anno + 4 * k);
}

}  // namespace bcfgpu

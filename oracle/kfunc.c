/*  kfunc.c -- ORACLE (test infrastructure only): the htslib kfunc routines the
 *  hot path calls, restated from their published algorithms.
 *
 *    kf_erfc          bam2bcf.c:341 (VDB)        Hart's rational approximation as
 *                                                given by West (2009), double precision
 *    kt_fisher_exact  bam2bcf.c:878 (FMT/SP)     two-sided Fisher exact test with the
 *                                                incremental hypergeometric walk
 *    kf_lgamma        (used by the above)        Lanczos, g=5.5? 8-term form
 *    kf_betai         ccall.c:91,100 (PV4)       regularised incomplete beta function, continued fraction evaluated with
 *                                                the modified Lentz algorithm
 *    test16           ccall.c:89-138 (PV4)       (bcftools' own: restated from the reference) strand bias by Fisher's
 *                                                exact test, baseQ / mapQ / tail-distance bias by one-sided t-tests
 *
 *  htslib is not part of /root/reference; parity of these is
 *  pinned through the reference's VDB / SP / PV4 golden values only (PV4: the INFO/PV4 values of
 *  test/mpileup.c.1.out against the INFO/I16 of test/mpileup.c.vcf, tests/test_oracle_pv4.py).
 */
#include <math.h>
#include <stdlib.h>
#include "bcforacle.h"

static double kf_lgamma(double z)
{
    double x = 0;
    x += 0.1659470187408462e-06 / (z+7);
    x += 0.9934937113930748e-05 / (z+6);
    x -= 0.1385710331296526     / (z+5);
    x += 12.50734324009056      / (z+4);
    x -= 176.6150291498386      / (z+3);
    x += 771.3234287757674      / (z+2);
    x -= 1259.139216722289      / (z+1);
    x += 676.5203681218835      / z;
    x += 0.9999999999995183;
    return log(x) - 5.58106146679532777 - z + (z-0.5) * log(z+6.5);
}

double orc_kf_erfc(double x)
{
    const double p0 = 220.2068679123761;
    const double p1 = 221.2135961699311;
    const double p2 = 112.0792914978709;
    const double p3 = 33.912866078383;
    const double p4 = 6.37396220353165;
    const double p5 = .7003830644436881;
    const double p6 = .03526249659989109;
    const double q0 = 440.4137358247522;
    const double q1 = 793.8265125199484;
    const double q2 = 637.3336333788311;
    const double q3 = 296.5642487796737;
    const double q4 = 86.78073220294608;
    const double q5 = 16.06417757920695;
    const double q6 = 1.755667163182642;
    const double q7 = .08838834764831844;
    double expntl, z, p;
    z = fabs(x) * M_SQRT2;
    if (z > 37.) return x > 0.? 0. : 2.;
    expntl = exp(z * z * - .5);
    if (z < 10. / M_SQRT2)
        p = expntl * ((((((p6 * z + p5) * z + p4) * z + p3) * z + p2) * z + p1) * z + p0)
            / (((((((q7 * z + q6) * z + q5) * z + q4) * z + q3) * z + q2) * z + q1) * z + q0);
    else p = expntl / 2.506628274631001 / (z + 1. / (z + 2. / (z + 3. / (z + 4. / (z + .65)))));
    return x > 0.? 2. * p : 2. * (1. - p);
}

static double lbinom(int n, int k)
{
    if (k == 0 || n == k) return 0;
    return kf_lgamma(n+1) - kf_lgamma(k+1) - kf_lgamma(n-k+1);
}

static double hypergeo(int n11, int n1_, int n_1, int n)
{
    return exp(lbinom(n1_, n11) + lbinom(n-n1_, n_1-n11) - lbinom(n, n_1));
}

typedef struct { int n11, n1_, n_1, n; double p; } hgacc_t;

static double hypergeo_acc(int n11, int n1_, int n_1, int n, hgacc_t *aux)
{
    if (n1_ || n_1 || n) {
        aux->n11 = n11; aux->n1_ = n1_; aux->n_1 = n_1; aux->n = n;
    } else {    /* only n11 changed */
        if (n11%11 && n11 + aux->n - aux->n1_ - aux->n_1) {
            if (n11 == aux->n11 + 1) {
                aux->p *= (double)(aux->n1_ - aux->n11) / n11
                    * (aux->n_1 - aux->n11) / (n11 + aux->n - aux->n1_ - aux->n_1);
                aux->n11 = n11;
                return aux->p;
            }
            if (n11 == aux->n11 - 1) {
                aux->p *= (double)aux->n11 / (aux->n1_ - n11)
                    * (aux->n11 + aux->n - aux->n1_ - aux->n_1) / (aux->n_1 - n11);
                aux->n11 = n11;
                return aux->p;
            }
        }
        aux->n11 = n11;
    }
    aux->p = hypergeo(aux->n11, aux->n1_, aux->n_1, aux->n);
    return aux->p;
}

double orc_kt_fisher_exact(int n11, int n12, int n21, int n22, double *_left, double *_right, double *two)
{
    int i, j, max, min;
    double p, q, left, right;
    hgacc_t aux;
    int n1_, n_1, n;

    n1_ = n11 + n12; n_1 = n11 + n21; n = n11 + n12 + n21 + n22;
    max = (n_1 < n1_) ? n_1 : n1_;
    min = n1_ + n_1 - n;
    if (min < 0) min = 0;
    *two = *_left = *_right = 1.;
    if (min == max) return 1.;
    q = hypergeo_acc(n11, n1_, n_1, n, &aux);
    /* left tail */
    p = hypergeo_acc(min, 0, 0, 0, &aux);
    for (left = 0., i = min + 1; p < 0.99999999 * q && i<=max; ++i)
        left += p, p = hypergeo_acc(i, 0, 0, 0, &aux);
    --i;
    if (p < 1.00000001 * q) left += p;
    else --i;
    /* right tail */
    p = hypergeo_acc(max, 0, 0, 0, &aux);
    for (right = 0., j = max - 1; p < 0.99999999 * q && j>=0; --j)
        right += p, p = hypergeo_acc(j, 0, 0, 0, &aux);
    ++j;
    if (p < 1.00000001 * q) right += p;
    else ++j;
    *two = left + right;
    if (*two > 1.) *two = 1.;
    if (abs(i - n11) < abs(j - n11)) right = 1. - left + q;
    else left = 1.0 - right + q;
    *_left = left; *_right = right;
    return q;
}

/* FORMAT/SP, bam2bcf.c:867-885 */
int orc_format_sp(int fwd_ref, int rev_ref, int fwd_alt, int rev_alt)
{
    if ( fwd_ref+rev_ref<2 || fwd_alt+rev_alt<2 || fwd_ref+fwd_alt<2 || rev_ref+rev_alt<2 ) return 0;
    double left, right, two;
    orc_kt_fisher_exact(fwd_ref, rev_ref, fwd_alt, rev_alt, &left, &right, &two);
    int x = (int)(-4.343 * log(two) + .499);
    if (x > 255) x = 255;
    return x;
}

/* kf_betai (htslib kfunc.c): I_x(a,b) by its continued fraction, modified Lentz algorithm */
#define KF_GAMMA_EPS 1e-14
#define KF_TINY 1e-290
static double kf_betai_aux(double a, double b, double x)
{
    double C, D, f;
    int j;
    if (x == 0.) return 0.;
    if (x == 1.) return 1.;
    f = 1.; C = f; D = 0.;
    for (j = 1; j < 200; ++j) {
        double aa, d;
        int m = j >> 1;
        aa = (j & 1) ? -(a + m) * (a + b + m) * x / ((a + 2*m) * (a + 2*m + 1))
                     : m * (b - m) * x / ((a + 2*m - 1) * (a + 2*m));
        D = 1. + aa * D;
        if (D < KF_TINY) D = KF_TINY;
        C = 1. + aa / C;
        if (C < KF_TINY) C = KF_TINY;
        D = 1. / D;
        d = C * D;
        f *= d;
        if (fabs(d - 1.) < KF_GAMMA_EPS) break;
    }
    return exp(kf_lgamma(a+b) - kf_lgamma(a) - kf_lgamma(b) + a * log(x) + b * log(1.-x)) / a / f;
}
double orc_kf_betai(double a, double b, double x)
{
    return x < (a + 1.) / (a + b + 2.) ? kf_betai_aux(a, b, x) : 1. - kf_betai_aux(b, a, 1. - x);
}

/* ttest, ccall.c:89-101 */
static double ttest(int n1, int n2, const float a[4])
{
    double t, v, u1, u2;
    if (n1 == 0 || n2 == 0 || n1 + n2 < 3) return 1.0;
    u1 = (double)a[0] / n1; u2 = (double)a[2] / n2;
    if (u1 <= u2) return 1.;
    t = (u1 - u2) / sqrt(((a[1] - n1 * u1 * u1) + (a[3] - n2 * u2 * u2)) / (n1 + n2 - 2) * (1./n1 + 1./n2));
    v = n1 + n2 - 2;
    return t < 0. ? 1. : .5 * orc_kf_betai(.5*v, .5, v/(v+t*t));
}

/* test16, ccall.c:103-138: p[4], is_tested; returns -1 when the depth is 0 */
int orc_test16(const float anno[16], double p[4], int *is_tested)
{
    double left, right;
    int i;
    p[0] = p[1] = p[2] = p[3] = 1.;
    *is_tested = 0;
    const float depth = anno[0] + anno[1] + anno[2] + anno[3];
    *is_tested = (anno[0] + anno[1] > 0 && anno[2] + anno[3] > 0);
    if (depth == 0) { *is_tested = 0; return -1; }
    orc_kt_fisher_exact((int)anno[0], (int)anno[1], (int)anno[2], (int)anno[3], &left, &right, &p[0]);
    for (i = 1; i < 4; ++i)
        p[i] = ttest((int)(anno[0] + anno[1]), (int)(anno[2] + anno[3]), anno + 4*i);
    return 0;
}

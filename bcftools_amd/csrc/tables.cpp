// tables.cpp -- host-side construction of the constant tables the kernels gather from.
//
//   errmod tables (fk, beta, lhet): the coefficients of htslib's errmod_init(), which
//   bcf_call_init() builds with depcorr = 1 - theta (bam2bcf.c:51).  errmod_cal() indexes beta as
//   q<<16 | n<<8 | k; the device copy is stored q<<16 | k<<8 | n (depth fastest): the lanes of a wavefront walk
//   their reads in lock step, so at one step they share k (and mostly q) and differ in the depth n of their
//   cells -- with n fastest a gather touches a couple of cache lines instead of one per lane.
//   pl2p: call_init_pl2p (mcall.c:56-61).
//   mw:   the Mann-Whitney table of mw.h, rebuilt by its own generating recursion
//         (mw.h:32-37), used by calc_mwu_bias (bam2bcf.c:483).
//
// These run once per context on the host CPU (about 0.3 s for beta) and are uploaded to HBM.
#include <cmath>
#include <cstdlib>
#include <vector>
#include "tables.h"

namespace bcfgpu {

void build_errmod_tables(double depcorr, std::vector<double> &fk, std::vector<double> &beta, std::vector<double> &lhet)
{
    const double eta = 0.03;
    fk.assign(256, 0.0);
    beta.assign((size_t)256 * 256 * 64, 0.0);
    lhet.assign((size_t)256 * 256, 0.0);
    fk[0] = 1.0;
    for (int n = 1; n < 256; ++n) fk[n] = std::pow(1. - depcorr, n) * (1.0 - eta) + eta;

    std::vector<double> lC((size_t)256 * 256, 0.0);
    for (int n = 1; n < 256; ++n)
        for (int k = 1; k <= n; ++k)
            lC[n << 8 | k] = lgamma(n + 1) - lgamma(k + 1) - lgamma(n - k + 1);

    for (int q = 1; q < 64; ++q) {
        const double e = std::pow(10.0, -q / 10.0);
        const double le = std::log(e), le1 = std::log(1.0 - e);
        for (int n = 1; n <= 255; ++n) {
            double *b = &beta[(size_t)q << 16 | n];
            long double sum = 0.0L, sum1 = 0.0L;
            for (int k = n; k >= 0; --k, sum1 = sum) {
                sum = sum1 + expl(lC[n << 8 | k] + k * le + (n - k) * le1);
                b[(size_t)k << 8] = -10. / M_LN10 * logl(sum1 / sum);
            }
        }
    }
    for (int n = 0; n < 256; ++n)
        for (int k = 0; k < 256; ++k)
            lhet[n << 8 | k] = lC[n << 8 | k] - M_LN2 * n;
}

void build_pl2p(double *pl2p)
{
    for (int i = 0; i < 256; i++) pl2p[i] = std::pow(10., -i / 10.);
}

static double mw_rec(int n, int m, int U)
{
    if (U < 0) return 0;
    if (n == 0 || m == 0) return U == 0 ? 1 : 0;
    return (double)n / (n + m) * mw_rec(n - 1, m, U - m) + (double)m / (n + m) * mw_rec(n, m - 1, U);
}

void build_mw_table(double *mw /* [6][6][50] */)
{
    for (int i = 2; i < 8; i++)
        for (int j = 2; j < 8; j++)
            for (int k = 0; k < 50; k++)
                mw[((i - 2) * 6 + (j - 2)) * 50 + k] = mw_rec(i, j, k);
}

}  // namespace bcfgpu

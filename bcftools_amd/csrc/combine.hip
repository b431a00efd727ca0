// combine.hip -- bcf_call_combine (+ calc_SegBias, calc_mwu_bias x4, calc_vdb) for every site of a tile.
//
// Replaces bam2bcf.c:558-754, :281-342, :440-530.  One 64-lane workgroup (a single wavefront) per site.
//
// Order-sensitive pieces are replayed in the reference's order:
//   * qsum[j] += (float)QS[j]/sum over samples (bam2bcf.c:569-575) decides the ALT allele
//     order, so it is a sequential float32 sum, one lane per allele, over LDS-staged QS;
//   * sum_min (bam2bcf.c:642) is a sequential double sum (lane 0).
// Integer totals (DP4, AD, I16, depth, ...) are exact in any order and use wave reductions.
// SGB is a sum of log/exp terms whose last bits already differ from glibc's, so it uses a
// tree reduction (compared with a relative tolerance).
#include <hip/hip_runtime.h>
#include <math.h>
#include <float.h>
#include "kernels.h"
#include "kfunc_dev.h"

namespace bcfgpu {

#define WG 64             // one wavefront per site: thousands of sites in flight hide the sequential chains
#define CHUNK 1024        // samples staged in LDS per round

__device__ __forceinline__ int nt16_int_c(int c) { return (int)((0x4444444344424104ull >> (4 * (c & 15))) & 7); }
__device__ __forceinline__ int tri_c(int j, int k) { return j <= k ? k * (k + 1) / 2 + j : j * (j + 1) / 2 + k; }

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// the same for per-lane 32-bit partial sums whose sum over 16 lanes stays below 2^32: four adds inside each row of 16 lanes by
// data-parallel-primitive moves (no trip through the LDS crossbar), the four row sums added as scalars
__device__ __forceinline__ unsigned long long wave_sum_u32(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);     // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);     // quad_perm [2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false);    // row_half_mirror
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false);    // row_mirror
    return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16)
         + (uint32_t)__builtin_amdgcn_readlane((int)v, 32) + (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}
__device__ __forceinline__ double wave_sum_f64(double v)
{
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// kf_erfc (htslib kfunc.c; Hart/West rational approximation), used by calc_vdb (bam2bcf.c:341)
__device__ double dev_kf_erfc(double x)
{
    const double p0 = 220.2068679123761, p1 = 221.2135961699311, p2 = 112.0792914978709, p3 = 33.912866078383,
                 p4 = 6.37396220353165, p5 = .7003830644436881, p6 = .03526249659989109;
    const double q0 = 440.4137358247522, q1 = 793.8265125199484, q2 = 637.3336333788311, q3 = 296.5642487796737,
                 q4 = 86.78073220294608, q5 = 16.06417757920695, q6 = 1.755667163182642, q7 = .08838834764831844;
    double expntl, z, p;
    z = fabs(x) * M_SQRT2;
    if (z > 37.) return x > 0. ? 0. : 2.;
    expntl = exp(z * z * -.5);
    if (z < 10. / M_SQRT2)
        p = expntl * ((((((p6 * z + p5) * z + p4) * z + p3) * z + p2) * z + p1) * z + p0)
            / (((((((q7 * z + q6) * z + q5) * z + q4) * z + q3) * z + q2) * z + q1) * z + q0);
    else p = expntl / 2.506628274631001 / (z + 1. / (z + 2. / (z + 3. / (z + 4. / (z + .65)))));
    return x > 0. ? 2. * p : 2. * (1. - p);
}

__device__ __forceinline__ uint32_t dev_format_sp(int fr, int rr, int fa, int ra)
{
    if (fr + rr < 2 || fa + ra < 2 || fr + fa < 2 || rr + ra < 2) return 0;
    int x = (int)(-4.343 * log(dev_fisher_two_sided(fr, rr, fa, ra)) + .499);
    return (uint32_t)(x > 255 ? 255 : x);
}

// calc_vdb, bam2bcf.c:281-342, called by the whole wavefront (result valid in every lane).  The first loop's float sum of
// pos[i]*i is a sum of integers: while the total stays below 2^24 every partial sum is exact, so it is the integer total
// (wave reduction); otherwise lane 0 replays the loop.  The second loop rounds to float after every addition, so it stays
// sequential on lane 0, but its terms pos[i]*|i - mean| are formed by all lanes first (adding the +0 of an empty bin
// changes nothing).  s_term: 128 doubles of LDS.
__device__ float wave_calc_vdb(const int *pos, int lane, double *s_term)
{
    const int readlen = 100, nparam = 15;
    const float param[15][3] = { {3,0.079f,18}, {4,0.09f,19.8f}, {5,0.1f,20.5f}, {6,0.11f,21.5f},
        {7,0.125f,21.6f}, {8,0.135f,22}, {9,0.14f,22.2f}, {10,0.153f,22.3f}, {15,0.19f,22.8f},
        {20,0.22f,23.2f}, {30,0.26f,23.4f}, {40,0.29f,23.5f}, {50,0.35f,23.65f}, {100,0.5f,23.7f},
        {200,0.7f,23.7f} };
    int i;
    const int i0 = 2 * lane, i1 = i0 + 1;
    const int p0 = i0 < readlen ? pos[i0] : 0, p1 = i1 < readlen ? pos[i1] : 0;
    int dp = p0 + p1;
    unsigned long long mp = (unsigned long long)p0 * i0 + (unsigned long long)p1 * i1;
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) { dp += __shfl_xor(dp, o); mp += __shfl_xor(mp, o); }
    if (dp < 2) return HUGE_VALF;
    float mean_pos = (float)mp;
    if (mp >= (1ull << 24)) {
        mean_pos = 0;
        if (lane == 0)
            for (i = 0; i < readlen; i++) { if (!pos[i]) continue; mean_pos += pos[i] * i; }
        mean_pos = __shfl(mean_pos, 0);
    }
    mean_pos /= dp;
    if (i0 < readlen) s_term[i0] = p0 * fabs((double)((float)i0 - mean_pos));
    if (i1 < readlen) s_term[i1] = p1 * fabs((double)((float)i1 - mean_pos));
    __syncthreads();
    float mean_diff = 0;
    if (lane == 0) {
        const double2 *t2 = reinterpret_cast<const double2*>(s_term);
        for (i = 0; i < readlen / 2; i++) {
            const double2 t = t2[i];
            mean_diff = (float)((double)mean_diff + t.x);
            mean_diff = (float)((double)mean_diff + t.y);
        }
    }
    mean_diff = __shfl(mean_diff, 0);
    mean_diff /= dp;
    int ipos = (int)mean_diff;
    if (dp == 2)
        return (float)((2 * readlen - 2 * (ipos + 1) - 1) * (ipos + 1) / (readlen - 1) / (readlen * 0.5));
    if (dp >= 200) i = nparam;
    else {
        for (i = 0; i < nparam; i++)
            if (param[i][0] >= dp) break;
    }
    float pshift, pscale;
    if (i == nparam) { pscale = param[nparam - 1][1]; pshift = param[nparam - 1][2]; }
    else if (i > 0 && param[i][0] != dp) {
        pscale = (float)((param[i - 1][1] + param[i][1]) * 0.5);
        pshift = (float)((param[i - 1][2] + param[i][2]) * 0.5);
    } else { pscale = param[i][1]; pshift = param[i][2]; }
    return (float)(0.5 * dev_kf_erfc(-(double)((mean_diff - pshift) * pscale)));
}

// calc_mwu_bias, bam2bcf.c:440-484.  The loop over the bins (bam2bcf.c:445-465) is summed by the wavefront, two bins
// per lane: U = sum_i a_i * (nb_before_i + b_i / 2) has only integer and half-integer terms far below 2^53, so the
// double sum is exact in any order; the int product of the b_i == 0 branch is kept as the reference writes it.
__device__ __forceinline__ void dev_mwu_sums(const int *a, const int *b, int n, int lane, int &na, int &nb, double &U)
{
    const int i0 = 2 * lane, i1 = i0 + 1;
    const int a0 = i0 < n ? a[i0] : 0, b0 = i0 < n ? b[i0] : 0, a1 = i1 < n ? a[i1] : 0, b1 = i1 < n ? b[i1] : 0;
    int incl = b0 + b1;
    #pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    const int nb0 = incl - (b0 + b1), nb1 = nb0 + b0;       // reads of b before bin i0 / i1
    double u = 0;
    if (a0) u += b0 ? a0 * (nb0 + b0 * 0.5) : (double)(a0 * nb0);
    if (a1) u += b1 ? a1 * (nb1 + b1 * 0.5) : (double)(a1 * nb1);
    int as = a0 + a1;
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) { as += __shfl_xor(as, o); u += __shfl_xor(u, o); }
    na = as; nb = __shfl(incl, 63); U = u;
}
__device__ float dev_mwu_tail(int na, int nb, double U, const double *mw)
{
    if (!na || !nb) return HUGE_VALF;
    if (na == 1 || nb == 1) return 1.0f;
    double mean = ((double)na * nb) * 0.5;
    if (na == 2 || nb == 2) return (float)(U > mean ? (2.0 * mean - U) / mean : U / mean);
    double var2 = ((double)na * nb) * (na + nb + 1) / 12.0;
    if (na >= 8 || nb >= 8) return (float)exp(-0.5 * (U - mean) * (U - mean) / var2);
    // exact: na,nb in 3..7, U < na*nb <= 49 -> always inside the mw.h table (bam2bcf.c:382)
    int iu = (int)U;
    double m = (iu >= 0 && iu < 50) ? mw[((na - 2) * 6 + (nb - 2)) * 50 + iu] : 0.0;
    return (float)(m * sqrt(2 * M_PI * var2));
}

__device__ __forceinline__ double dev_logsumexp2(double a, double b)
{
    if (a > b) return log(1 + exp(b - a)) + a;
    else       return log(1 + exp(a - b)) + b;
}

struct SiteShared {
    int a[5]; int n_alleles, unseen, ori_ref, x;
    float qsum_out[5];
    int g[15], g1[15], g2[15];
    unsigned long long tot[32];   // integer totals
    double segb_sum; double sum_min;
    float bias[6];
};

// Sequential sums in the reference's order (they decide allele order and the PL shift, so they are not tree-reduced).
// One lane walks an LDS array; the next 16 values are requested while the current 16 are added, so the chain runs at
// the latency of the dependent adds rather than that of the LDS.
__device__ __forceinline__ float seq_sum_f32(float acc, const float *v, int n)
{
    const float4 *q = reinterpret_cast<const float4*>(v);
    const int nb = n >> 4;
    float4 c0, c1, c2, c3;
    if (nb > 0) { c0 = q[0]; c1 = q[1]; c2 = q[2]; c3 = q[3]; }
    for (int b = 0; b < nb; ++b) {
        const float4 a0 = c0, a1 = c1, a2 = c2, a3 = c3;
        if (b + 1 < nb) { c0 = q[4 * b + 4]; c1 = q[4 * b + 5]; c2 = q[4 * b + 6]; c3 = q[4 * b + 7]; }
        acc += a0.x; acc += a0.y; acc += a0.z; acc += a0.w; acc += a1.x; acc += a1.y; acc += a1.z; acc += a1.w;
        acc += a2.x; acc += a2.y; acc += a2.z; acc += a2.w; acc += a3.x; acc += a3.y; acc += a3.z; acc += a3.w;
    }
    for (int i = nb << 4; i < n; ++i) acc += v[i];
    return acc;
}
__device__ __forceinline__ double seq_sum_f64(double acc, const double *v, int n)
{
    const double2 *q = reinterpret_cast<const double2*>(v);
    const int nb = n >> 3;
    double2 c0, c1, c2, c3;
    if (nb > 0) { c0 = q[0]; c1 = q[1]; c2 = q[2]; c3 = q[3]; }
    for (int b = 0; b < nb; ++b) {
        const double2 a0 = c0, a1 = c1, a2 = c2, a3 = c3;
        if (b + 1 < nb) { c0 = q[4 * b + 4]; c1 = q[4 * b + 5]; c2 = q[4 * b + 6]; c3 = q[4 * b + 7]; }
        acc += a0.x; acc += a0.y; acc += a1.x; acc += a1.y; acc += a2.x; acc += a2.y; acc += a3.x; acc += a3.y;
    }
    for (int i = nb << 3; i < n; ++i) acc += v[i];
    return acc;
}

struct SampleTotals { uint32_t adf[5], adr[5], scr, ori, mq0, cnt[4]; };

// The per-sample outputs of bcf_call_combine for one chunk of samples of a site with NAL alleles (NAL = 0: a dead indel
// site, totals only): PL from the NAL(NAL+1)/2 genotype likelihoods in allele order (bam2bcf.c:634-651), DP4, AD/ADF/ADR,
// QS, SCR planes, per-lane integer totals, and each sample's minimum for the sequential sum_min.
// V = 4: a lane takes four consecutive samples, so that every plane is read with 16-byte loads and the u8 planes are
// written four bytes at a time (the caller checks that n_smpl and the plane addresses allow it); V = 1 otherwise.
template <int V, class T4> __device__ __forceinline__ void load_v(const T4 *p, T4 (&d)[V])
{
    if constexpr (V == 4) { const uint4 u = *reinterpret_cast<const uint4*>(p); __builtin_memcpy(d, &u, 16); }
    else d[0] = p[0];
}
template <int V> __device__ __forceinline__ void store_bytes(uint8_t *p, const uint32_t (&b)[V])
{
    if constexpr (V == 4) *reinterpret_cast<uint32_t*>(p) = b[0] | b[1] << 8 | b[2] << 16 | b[3] << 24;
    else p[0] = (uint8_t)b[0];
}
template <int V> __device__ __forceinline__ void store_u16(uint16_t *p, const uint32_t (&b)[V])
{
    if constexpr (V == 4) *reinterpret_cast<uint2*>(p) = make_uint2(b[0] | b[1] << 16, b[2] | b[3] << 16);
    else p[0] = (uint16_t)b[0];
}
template <int V> __device__ __forceinline__ void store_i32(int32_t *p, const uint32_t (&b)[V])
{
    if constexpr (V == 4) *reinterpret_cast<uint4*>(p) = make_uint4(b[0], b[1], b[2], b[3]);
    else p[0] = (int32_t)b[0];
}
template <int NAL, int V>
__device__ __forceinline__ void sample_planes(const CombineParams &P, SampleTotals &T, const int (&gs)[15], const int (&g1s)[15], const int (&g2s)[15], const int (&as)[5],
                                              int is, long c0, int base, int cn, int tid, long ncells, double *s_min, uint32_t &wide_seen)
{
    constexpr int X = NAL * (NAL + 1) / 2;
    const size_t Ss = (size_t)P.n_smpl;
    for (int i = tid * V; i < cn; i += WG * V) {
        const int s = base + i;
        const long cell = c0 + s;
        float mn[V];
        #pragma unroll
        for (int v = 0; v < V; ++v) mn[v] = 0.f;
        uint32_t cnt4[V], adf[V], adr[V], misc[V];
        load_v<V>(P.cr.cnt4 + cell, cnt4); load_v<V>(P.cr.adf + cell, adf); load_v<V>(P.cr.adr + cell, adr); load_v<V>(P.cr.misc + cell, misc);
        if (NAL > 0) {
            float pv[X > 0 ? X : 1][V];
            // single-base cells: the planes from (A, b, n), see CallretPlanes; the rest from the stored likelihoods
            float pa[V], ph[V];
            load_v<V>(P.cr.pa + cell, pa);
            bool any_full = false;
            #pragma unroll
            for (int v = 0; v < V; ++v) {
                const int nrd = (int)((cnt4[v] & 0xff) + ((cnt4[v] >> 8) & 0xff) + ((cnt4[v] >> 16) & 0xff) + (cnt4[v] >> 24));
                float h = 0.0f;
                if (nrd > 0) { h = (float)(-4.343 * (0.0 - M_LN2 * nrd)); if (h < 0.0f) h = 0.0f; }    // lhet[n<<8|n] = lhet[n<<8|0] = -ln2*n (tables.cpp)
                ph[v] = h;
                any_full |= (misc[v] & CR_FULL) != 0;
            }
            #pragma unroll
            for (int v = 0; v < V; ++v) mn[v] = FLT_MAX;
            #pragma unroll
            for (int j = 0; j < X; ++j) {
                const int b1 = g1s[j], b2 = g2s[j];                // the two bases of genotype j (scalar)
                #pragma unroll
                for (int v = 0; v < V; ++v) {
                    const int b = (int)(misc[v] & 7);
                    pv[j][v] = b1 == b2 ? (b1 == b ? 0.0f : pa[v]) : ((b1 == b || b2 == b) ? ph[v] : pa[v]);
                }
            }
            if (__any(any_full)) {
                #pragma unroll
                for (int j = 0; j < X; ++j) {
                    #pragma unroll
                    for (int v = 0; v < V; ++v)
                        if (misc[v] & CR_FULL) pv[j][v] = P.cr.p15[(size_t)(c0 + s + v) * 16 + gs[j]];
                }
            }
            #pragma unroll
            for (int j = 0; j < X; ++j) {
                #pragma unroll
                for (int v = 0; v < V; ++v) if (mn[v] > pv[j][v]) mn[v] = pv[j][v];
            }
            uint8_t *PL = P.out.pl + (size_t)is * BCFGPU_MAX_PL * Ss + s;
            #pragma unroll
            for (int j = 0; j < X; ++j) {
                uint32_t yb[V];
                #pragma unroll
                for (int v = 0; v < V; ++v) {
                    int y = (int)((double)(pv[j][v] - mn[v]) + .499);
                    if (y > 255) y = 255;
                    yb[v] = (uint32_t)y;
                }
                store_bytes<V>(PL + (size_t)j * Ss, yb);
            }
        }
        #pragma unroll
        for (int v = 0; v < V; ++v) s_min[i + v] = (double)mn[v];        // widened here, by all lanes, for the sequential sum
        uint32_t any_wide = 0;
        #pragma unroll
        for (int v = 0; v < V; ++v) any_wide |= misc[v];
        any_wide &= CR_WIDE;
        if (NAL > 0 && !BCFGPU_ABL(P, 512)) {
            uint16_t *DP4 = P.out.dp4 + (size_t)is * 4 * Ss + s;
            uint32_t b[V];
            #pragma unroll
            for (int j = 0; j < 4; ++j) {
                #pragma unroll
                for (int v = 0; v < V; ++v) b[v] = (cnt4[v] >> (8 * j)) & 0xff;
                store_u16<V>(DP4 + (size_t)j * Ss, b);
            }
            #pragma unroll
            for (int v = 0; v < V; ++v) { b[v] = (misc[v] >> 8) & 0xff; T.scr += b[v]; }
            if (P.out.scr) store_u16<V>(P.out.scr + (size_t)is * Ss + s, b);
            unsigned long long qs64[V];
            #pragma unroll
            for (int v = 0; v < V; ++v) qs64[v] = P.out.qs ? P.cr.qs64[cell + v] : 0ull;
            #pragma unroll
            for (int j = 0; j < NAL; ++j) {
                const int aj = as[j];                         // scalar
                uint32_t vf[V], vr[V];
                #pragma unroll
                for (int v = 0; v < V; ++v) {
                    vf[v] = aj < 4 ? (adf[v] >> (8 * aj)) & 0xff : 0;
                    vr[v] = aj < 4 ? (adr[v] >> (8 * aj)) & 0xff : 0;
                    T.adf[j] += vf[v]; T.adr[j] += vr[v];
                }
                if (P.out.adf) store_u16<V>(P.out.adf + ((size_t)is * 5 + j) * Ss + s, vf);
                if (P.out.adr) store_u16<V>(P.out.adr + ((size_t)is * 5 + j) * Ss + s, vr);
                if (P.out.qs) {
                    uint32_t q[V];
                    #pragma unroll
                    for (int v = 0; v < V; ++v) q[v] = (uint32_t)(aj < 4 ? (qs64[v] >> (16 * aj)) & 0xffff : 0);
                    store_i32<V>(P.out.qs + ((size_t)is * 5 + j) * Ss + s, q);
                }
            }
        }
        wide_seen |= any_wide;
        #pragma unroll
        for (int v = 0; v < V; ++v) {
            #pragma unroll
            for (int j = 0; j < 4; ++j) T.cnt[j] += (cnt4[v] >> (8 * j)) & 0xff;
        }
    }
}

// Cells of more than 255 usable reads (rare: none under mpileup's default -d 250 with a file per sample).  What sample_planes
// took from the packed planes for such a cell are the counts of the 255 reads its likelihoods were made of; its WideRec
// (kernels.h) has the counts over all reads, as bcf_call_glfgen leaves them (bam2bcf.c:203-226).  A pass of its own over the
// site's samples, entered only when sample_planes met such a cell: the cell's plane entries are written again, and what it
// added to the lane's totals is replaced (the same expressions subtracted: exact in modular arithmetic).
__device__ __forceinline__ void fix_wide_cells(const CombineParams &P, SampleTotals &T, const int (&as)[5], int nal, bool live, int is, long c0, int tid)
{
    const size_t Ss = (size_t)P.n_smpl;
    for (int s = tid; s < P.n_smpl; s += WG) {
        const uint32_t misc = P.cr.misc[c0 + s];
        if (!(misc & CR_WIDE)) continue;
        const uint32_t cnt4 = P.cr.cnt4[c0 + s], adf = P.cr.adf[c0 + s], adr = P.cr.adr[c0 + s];
        const WideRec *w = P.cr.wide + (uint32_t)P.cr.qs64[c0 + s];
        const uint32_t wc[4] = { w->cnt[0] & 0xffffu, w->cnt[0] >> 16, w->cnt[1] & 0xffffu, w->cnt[1] >> 16 };
        for (int j = 0; j < 4; ++j) T.cnt[j] += wc[j] - ((cnt4 >> (8 * j)) & 0xff);
        if (!live) continue;
        for (int j = 0; j < 4; ++j) P.out.dp4[((size_t)is * 4 + j) * Ss + s] = (uint16_t)wc[j];
        T.scr += w->scr - ((misc >> 8) & 0xff);
        if (P.out.scr) P.out.scr[(size_t)is * Ss + s] = (uint16_t)w->scr;
        for (int j = 0; j < nal; ++j) {
            const int aj = as[j];
            const uint32_t ad = aj < 4 ? w->ad[aj] : 0u, nf = ad & 0xffffu, nr = ad >> 16;
            T.adf[j] += nf - (aj < 4 ? (adf >> (8 * aj)) & 0xff : 0);
            T.adr[j] += nr - (aj < 4 ? (adr >> (8 * aj)) & 0xff : 0);
            if (P.out.adf) P.out.adf[((size_t)is * 5 + j) * Ss + s] = (uint16_t)nf;
            if (P.out.adr) P.out.adr[((size_t)is * 5 + j) * Ss + s] = (uint16_t)nr;
            if (P.out.qs) P.out.qs[((size_t)is * 5 + j) * Ss + s] = (int32_t)(aj < 4 ? w->qs[aj] : 0u);
        }
    }
}

template <int V>
#ifndef COMB_WAVES
#define COMB_WAVES 4         // wavefronts per SIMD the register budget is held to
#endif
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(COMB_WAVES, COMB_WAVES))) void combine_kernel(const CombineParams P)
{
    __shared__ SiteShared sh;
    __shared__ unsigned long long s_stage[CHUNK];   // QS (4 x u16) per sample, later float mins
    __shared__ float s_q[4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int is = blockIdx.x;
    const int S = P.n_smpl;
    const long ncells = (long)P.n_sites * S;
    const long c0 = (long)is * S;
    bcfgpu_site *site = &P.out.site[is];

    const int ref_base = P.is_indel ? -1 : P.ref16[is];
    int ref4;
    if (ref_base >= 0) { ref4 = nt16_int_c(ref_base); if (ref4 > 4) ref4 = 4; } else ref4 = 0;

    if (tid < 32) sh.tot[tid] = 0;
    if (tid == 0) { sh.segb_sum = 0; sh.sum_min = 0; }

    // ---- qsum (bam2bcf.c:569-575): every thread normalises its samples' QS in parallel, then lane j of
    // wave 0 adds allele j's fractions in sample order (a sequential float32 sum, as in the reference) ----
    float myq = 0.f;
    float *s_frt = reinterpret_cast<float*>(s_stage);          // [4 alleles][CHUNK/2] normalised QS of the staged samples
    constexpr int HC = CHUNK / 2;
    // (a chunk's packed QS words are all requested before the first is used, and the next chunk's before this chunk's sequential
    // sum is run: the loop was a chain of memory round trips, one per 64 samples)
    constexpr int QU = HC / WG;
    unsigned long long qv[QU];
    auto fetch_qs = [&](int base) {
        #pragma unroll
        for (int u = 0; u < QU; ++u) { const int i = base + tid + u * WG; qv[u] = i < S ? P.cr.qs64[c0 + i] : 0ull; }
    };
    fetch_qs(0);
    for (int base = 0; base < S; base += HC) {
        const int cn = min(HC, S - base);
        __syncthreads();
        #pragma unroll
        for (int u = 0; u < QU; ++u) {
            const int i = tid + u * WG;
            if (i >= cn) break;
            const unsigned long long v = qv[u];
            float q0 = (float)(int)(v & 0xffff), q1 = (float)(int)((v >> 16) & 0xffff);
            float q2 = (float)(int)((v >> 32) & 0xffff), q3 = (float)(int)((v >> 48) & 0xffff);
            if ((uint32_t)(v >> 48) == 0xffffu) {               // WIDE_QS_MARK: a cell of more than 255 usable reads, QS over all of them
                const WideRec *w = P.cr.wide + (uint32_t)v;
                q0 = (float)(int)w->qs[0]; q1 = (float)(int)w->qs[1]; q2 = (float)(int)w->qs[2]; q3 = (float)(int)w->qs[3];
            }
            float sum = 0;
            sum += q0; sum += q1; sum += q2; sum += q3;
            // q / sum is +0 for q = 0 and 1 for q = sum, exactly; most cells show one base, the same one for all the samples of a site:
            // an allele none of the wavefront's 64 samples needs a real division for (a wave-uniform test) takes the select
            auto frac = [&](float q) -> float {
                if (__any(q != 0.f && q != sum)) return sum != 0.f ? q / sum : 0.f;
                return q != 0.f ? 1.0f : 0.0f;
            };
            const float4 f = make_float4(frac(q0), frac(q1), frac(q2), frac(q3));
            // adding +0 for empty samples leaves the running sum unchanged
            s_frt[i] = f.x; s_frt[HC + i] = f.y; s_frt[2 * HC + i] = f.z; s_frt[3 * HC + i] = f.w;
        }
        __syncthreads();
        if (base + HC < S) fetch_qs(base + HC);
        if (tid < 4 && !BCFGPU_ABL(P, 8192)) myq = seq_sum_f32(myq, s_frt + tid * HC, cn);
    }
    if (tid < 4) s_q[tid] = myq;
    __syncthreads();

    // ---- allele ordering (bam2bcf.c:577-632), lane 0 ----
    if (tid == 0) {
        float qsum[5] = { s_q[0], s_q[1], s_q[2], s_q[3], 0.f };
        int idx[5] = {0, 1, 2, 3, 4};
        for (int i = 1; i < 4; i++)
            for (int j = i; j > 0 && qsum[idx[j]] < qsum[idx[j - 1]]; j--) { int t = idx[j]; idx[j] = idx[j - 1]; idx[j - 1] = t; }
        for (int i = 0; i < 5; i++) { sh.a[i] = -1; sh.qsum_out[i] = 0; }
        sh.unseen = -1;
        sh.a[0] = ref4;
        int i, j;
        for (i = 3, j = 1; i >= 0; i--) {
            const int ipos = idx[i];
            if (ipos == ref4) sh.qsum_out[0] = qsum[ipos];
            else {
                if (!qsum[ipos]) break;
                sh.qsum_out[j] = qsum[ipos];
                sh.a[j++] = ipos;
            }
        }
        int ret = 0;
        if (ref_base >= 0) {
            if (((ref4 < 4 && j < 4) || (ref4 == 4 && j < 5)) && i >= 0) { sh.unseen = j; sh.a[j++] = idx[i]; }
            sh.n_alleles = j;
        } else {
            sh.n_alleles = j;
            if (j == 1) ret = -1;
        }
        sh.ori_ref = ref_base >= 0 ? nt16_int_c(ref_base) : -1;
        sh.x = sh.n_alleles * (sh.n_alleles + 1) / 2;
        int z = 0;
        for (i = 0; i < sh.n_alleles; ++i)
            for (j = 0; j <= i; ++j) { sh.g1[z] = sh.a[j]; sh.g2[z] = sh.a[i]; sh.g[z++] = tri_c(sh.a[j], sh.a[i]); }
        site->ret = ret;
    }
    __syncthreads();
    const int nal = sh.n_alleles, x = sh.x;
    const bool dead = (P.is_indel && nal == 1);     // bcf_call_combine returned -1 (bam2bcf.c:611)

    // ---- per-sample planes + integer totals ----
    // per-lane partial totals (u32 is ample: a lane sees S/64 samples of <= 4 x 65535 reads); anno[4..15] arrive as site totals
    SampleTotals T;
    #pragma unroll
    for (int j = 0; j < 5; ++j) T.adf[j] = T.adr[j] = 0;
    T.scr = T.ori = T.mq0 = 0; T.cnt[0] = T.cnt[1] = T.cnt[2] = T.cnt[3] = 0;
    // the allele order is uniform over the wavefront: keep it in scalar registers so that plane addresses are scalar
    int gs[15], g1s[15], g2s[15], as[5];
    #pragma unroll
    for (int j = 0; j < 15; ++j) {
        gs[j] = __builtin_amdgcn_readfirstlane(j < x ? sh.g[j] : 0);
        g1s[j] = __builtin_amdgcn_readfirstlane(j < x ? sh.g1[j] : 0); g2s[j] = __builtin_amdgcn_readfirstlane(j < x ? sh.g2[j] : 0);
    }
    #pragma unroll
    for (int j = 0; j < 5; ++j) as[j] = __builtin_amdgcn_readfirstlane(j < nal ? sh.a[j] : 4);
    double *s_min = reinterpret_cast<double*>(s_stage);
    uint32_t wide_seen = 0;
    for (int base = 0; base < S; base += CHUNK) {
        const int cn = min(CHUNK, S - base);
        __syncthreads();
        const bool live = !dead && !BCFGPU_ABL(P, 256);
        #define PLANES(V_) switch (live ? nal : 0) { \
            case 1: sample_planes<1, V_>(P, T, gs, g1s, g2s, as, is, c0, base, cn, tid, ncells, s_min, wide_seen); break; \
            case 2: sample_planes<2, V_>(P, T, gs, g1s, g2s, as, is, c0, base, cn, tid, ncells, s_min, wide_seen); break; \
            case 3: sample_planes<3, V_>(P, T, gs, g1s, g2s, as, is, c0, base, cn, tid, ncells, s_min, wide_seen); break; \
            case 4: sample_planes<4, V_>(P, T, gs, g1s, g2s, as, is, c0, base, cn, tid, ncells, s_min, wide_seen); break; \
            case 5: sample_planes<5, V_>(P, T, gs, g1s, g2s, as, is, c0, base, cn, tid, ncells, s_min, wide_seen); break; \
            default: sample_planes<0, V_>(P, T, gs, g1s, g2s, as, is, c0, base, cn, tid, ncells, s_min, wide_seen); break; }
        PLANES(V)
        #undef PLANES
        __syncthreads();
        // bam2bcf.c:642, sum_min += the sample's smallest likelihood, in sample order.  A cell whose reads all show one base has
        // a zero there (nine in ten do): x + 0.0 is x, so only the other terms are added, in their order -- found by the whole
        // wavefront 64 samples at a time, added by the scalar walk over the ballot (every lane computes the same sum).
        if (!dead && !BCFGPU_ABL(P, 8192)) {
            double acc = sh.sum_min;
            for (int b0 = 0; b0 < cn; b0 += WG) {
                const double v = b0 + tid < cn ? s_min[b0 + tid] : 0.0;
                unsigned long long m = __ballot(v != 0.0);
                while (m) {
                    const int l = __builtin_ctzll(m);
                    m &= m - 1;
                    const unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)bits, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(bits >> 32), l);
                    acc += __builtin_bit_cast(double, (unsigned long long)hi << 32 | lo);
                }
            }
            __syncthreads();                                          // (sh.sum_min was read by every lane above)
            if (tid == 0) sh.sum_min = acc;
        }
    }
    if (__any(wide_seen != 0)) {
        int as5[5];
        #pragma unroll
        for (int j = 0; j < 5; ++j) as5[j] = as[j];
        fix_wide_cells(P, T, as5, nal, !dead && !BCFGPU_ABL(P, 256), is, c0, tid);
        __syncthreads();                                          // the SP / SGB passes below read the DP4 planes
    }
    // FMT/SP (bam2bcf.c:867-885): a Fisher exact test per sample, in its own pass -- its loops over the table's margins
    // diverge between lanes, and only samples with at least two reads in every margin enter them
    if (P.out.sp && (P.fmt_flag & BCFGPU_FMT_SP) && !dead) {
        // (from the DP4 planes this wavefront has just written: they hold the counts over all reads, whatever the cell's depth)
        const uint16_t *d4 = P.out.dp4 + (size_t)is * 4 * S;
        for (int s = tid; s < S; s += WG)
            P.out.sp[(size_t)is * S + s] = (uint8_t)dev_format_sp(d4[s], d4[(size_t)S + s], d4[2 * (size_t)S + s], d4[3 * (size_t)S + s]);
    }
    uint32_t (&t_adf)[5] = T.adf; uint32_t (&t_adr)[5] = T.adr; uint32_t (&t_cnt)[4] = T.cnt;
    const uint32_t t_scr = T.scr, t_ori = T.ori, t_mq0 = T.mq0;
    // wave reductions of the integer totals, then one LDS atomic per wave
    {
        unsigned long long v;
        // (a lane's partial sum is at most S / 64 samples of 65 535 reads: sixteen of them fit 32 bits up to 262 144 samples)
        const bool narrow = S <= 262144;
        auto wsum = [&](uint32_t x) -> unsigned long long { return narrow ? wave_sum_u32(x) : wave_sum_u64((unsigned long long)x); };
        #pragma unroll
        for (int j = 0; j < 5; ++j) { v = wsum(t_adf[j]); if (lane == 0 && v) atomicAdd(&sh.tot[j], v); }
        #pragma unroll
        for (int j = 0; j < 5; ++j) { v = wsum(t_adr[j]); if (lane == 0 && v) atomicAdd(&sh.tot[5 + j], v); }
        v = wsum(t_scr); if (lane == 0 && v) atomicAdd(&sh.tot[10], v);
        v = wsum(t_ori); if (lane == 0 && v) atomicAdd(&sh.tot[11], v);
        v = wsum(t_mq0); if (lane == 0 && v) atomicAdd(&sh.tot[12], v);
        #pragma unroll
        for (int j = 0; j < 4; ++j) { v = wsum(t_cnt[j]); if (lane == 0 && v) atomicAdd(&sh.tot[13 + j], v); }
        if (tid < 12) sh.tot[17 + tid] = P.site_sums[(size_t)is * SITE_NSUM + tid];
        if (tid == 12) sh.tot[29] = P.site_sums[(size_t)is * SITE_NSUM + 12];     // ori_depth and mq0 come as site totals
        if (tid == 13) sh.tot[30] = P.site_sums[(size_t)is * SITE_NSUM + 13];
    }
    __syncthreads();

    // ---- calc_SegBias (bam2bcf.c:494-530): tree-reduced sum of per-sample terms ----
    const double an0 = (double)sh.tot[13], an1 = (double)sh.tot[14], an2 = (double)sh.tot[15], an3 = (double)sh.tot[16];
    const int nr = (int)(an2 + an3);
    if (nr && !dead && !BCFGPU_ABL(P, 2048)) {
        const int avg_dp = (int)((an0 + an1 + nr) / S);
        double M = floor((double)nr / avg_dp + 0.5);
        if (M > S) M = S;
        else if (M == 0) M = 1;
        const double f = M / 2. / S;
        const double p = (double)nr / S;
        const double q = (double)nr / M;
        const double log2 = log(2.0);
        // The per-sample term depends only on the sample's alt-read count oi (0..510), and most samples of a site share a
        // handful of values: count the samples per oi (zeros by ballot, the rest by LDS atomics), then evaluate the
        // log/exp expression once per occurring count.  (The sum is tree-reduced and compared with a tolerance anyway.)
        int *s_oc = reinterpret_cast<int*>(s_stage);           // [512]
        for (int i = tid; i < 512; i += WG) s_oc[i] = 0;
        __syncthreads();
        int nzero = 0;
        double part_deep = 0;
        const uint16_t *d4s = P.out.dp4 + (size_t)is * 4 * S;
        for (int sb = 0; sb < S; sb += 4 * WG) {
          // (four rounds of 64 samples requested together: the loop is a chain of round trips to the planes otherwise)
          int oi4[4];
          #pragma unroll
          for (int u = 0; u < 4; ++u) {
              const int s = sb + u * WG + tid;
              oi4[u] = s < S ? (int)d4s[2 * (size_t)S + s] + (int)d4s[3 * (size_t)S + s] : -1;     // the DP4 planes written above (counts over all reads)
          }
          #pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (sb + u * WG >= S) break;
            const int oi = oi4[u];
            nzero += __popcll(__ballot(oi == 0));
            if (oi > 0 && oi < 512) atomicAdd(&s_oc[oi], 1);
            else if (oi >= 512) {                                 // a cell far past 255 reads: its term on its own
                double tmp = dev_logsumexp2(log(2 * (1 - f)), log(f) + oi * log2 - q);
                tmp += log(f) + oi * log(q / p) - q + p;
                part_deep += tmp;
            }
          }
        }
        __syncthreads();
        double part = part_deep;
        for (int k0 = 0; k0 < 512; k0 += WG) {
            const int oi = k0 + tid;
            const int cnt = oi == 0 ? nzero : s_oc[oi];
            if (!__any(cnt > 0)) continue;
            double tmp;
            if (oi) {
                tmp = dev_logsumexp2(log(2 * (1 - f)), log(f) + oi * log2 - q);
                tmp += log(f) + oi * log(q / p) - q + p;
            } else
                tmp = log(2 * f * (1 - f) * exp(-q) + f * f * exp(-2 * q) + (1 - f) * (1 - f)) + p;
            if (cnt > 0) part += cnt * tmp;
        }
        part = wave_sum_f64(part);
        if (lane == 0) atomicAdd(&sh.segb_sum, part);
    }
    // ---- MWU x4 and VDB from the site histograms (bam2bcf.c:735-751): staged in LDS, one lane per test ----
    if (tid < 6) sh.bias[tid] = 0.f;
    int *s_h = reinterpret_cast<int*>(s_stage);
    __syncthreads();
    if (!dead) {
        const int *h = P.hist + (long)is * H_SIZE;
        for (int i = tid; i < H_SIZE; i += WG) s_h[i] = h[i];
    }
    __syncthreads();
    if (!dead && !BCFGPU_ABL(P, 4096)) {
        const int *h = s_h;
        // the four Mann-Whitney tests: bin sums by the whole wavefront, the closed forms in lanes 1..4; lane 0 the VDB
        int na_t = 0, nb_t = 0; double U_t = 0;
        #pragma unroll
        for (int t = 1; t < 5; ++t) {
            const int *ha = h + (t == 1 ? H_REF_POS : t == 2 ? H_REF_MQ : t == 3 ? H_REF_BQ : H_FWD_MQS);
            const int *hb = h + (t == 1 ? H_ALT_POS : t == 2 ? H_ALT_MQ : t == 3 ? H_ALT_BQ : H_REV_MQS);
            int na, nb; double U;
            dev_mwu_sums(ha, hb, t == 1 ? BCFGPU_NPOS : BCFGPU_NQUAL, lane, na, nb, U);
            if (tid == t) { na_t = na; nb_t = nb; U_t = U; }
        }
        if (tid >= 1 && tid < 5) sh.bias[tid] = (tid == 1 && !(P.fmt_flag & BCFGPU_INFO_RPB)) ? 0.f : dev_mwu_tail(na_t, nb_t, U_t, P.mw);
        const float vdb = (P.fmt_flag & BCFGPU_INFO_VDB) ? wave_calc_vdb(h + H_ALT_POS, lane, reinterpret_cast<double*>(s_stage) + 512) : 0.f;
        if (tid == 0) sh.bias[0] = vdb;
    }
    __syncthreads();

    // ---- the site struct ----
    if (tid == 0) {
        for (int i = 0; i < 5; ++i) { site->a[i] = sh.a[i]; site->qsum[i] = sh.qsum_out[i]; }
        site->n_alleles = nal; site->unseen = sh.unseen; site->ori_ref = sh.ori_ref;
        site->pad = 0;
        if (dead) {
            site->shift = 0; site->depth = site->ori_depth = site->mq0 = 0; site->scr_tot = 0;
            site->vdb = site->mwu_pos = site->mwu_mq = site->mwu_bq = site->mwu_mqs = site->seg_bias = 0.f;
            for (int i = 0; i < 5; ++i) site->adf_tot[i] = site->adr_tot[i] = 0;
            for (int i = 0; i < 16; ++i) site->anno[i] = 0;
        } else {
            site->shift = (int)(sh.sum_min + .499);
            site->depth = (uint32_t)(sh.tot[13] + sh.tot[14] + sh.tot[15] + sh.tot[16]);
            site->ori_depth = (uint32_t)(sh.tot[11] + sh.tot[29]); site->mq0 = (uint32_t)(sh.tot[12] + sh.tot[30]);
            site->scr_tot = (int)sh.tot[10];
            for (int i = 0; i < 5; ++i) { site->adf_tot[i] = (int)sh.tot[i]; site->adr_tot[i] = (int)sh.tot[5 + i]; }
            for (int i = 0; i < 4; ++i) site->anno[i] = (double)sh.tot[13 + i];
            for (int i = 0; i < 12; ++i) site->anno[4 + i] = (double)sh.tot[17 + i];
            site->seg_bias = nr ? (float)sh.segb_sum : HUGE_VALF;
            site->vdb = sh.bias[0]; site->mwu_pos = sh.bias[1]; site->mwu_mq = sh.bias[2];
            site->mwu_bq = sh.bias[3]; site->mwu_mqs = sh.bias[4];
        }
    }
}

void launch_combine(const CombineParams &p, hipStream_t s)
{
    if (p.n_sites == 0) return;
    CombineParams q = p;
    // four samples per lane need every site's row of every plane to start on a 16-byte (u8 planes: 4-byte) boundary
    auto al = [](const void *ptr, uintptr_t a) { return ptr == nullptr || reinterpret_cast<uintptr_t>(ptr) % a == 0; };
    q.vec4 = (p.n_smpl % 4 == 0) && !BCFGPU_ABL(p, 1024) &&
             al(p.cr.p15, 16) && al(p.cr.pa, 16) && al(p.cr.cnt4, 16) && al(p.cr.adf, 16) && al(p.cr.adr, 16) && al(p.cr.misc, 16) && al(p.cr.qs64, 8) &&
             al(p.out.pl, 4) && al(p.out.dp4, 8) && al(p.out.scr, 8) && al(p.out.adf, 8) && al(p.out.adr, 8) && al(p.out.qs, 16);
    if (q.vec4) hipLaunchKernelGGL(combine_kernel<4>, dim3(q.n_sites), dim3(WG), 0, s, q);
    else hipLaunchKernelGGL(combine_kernel<1>, dim3(q.n_sites), dim3(WG), 0, s, q);
}

}  // namespace bcfgpu

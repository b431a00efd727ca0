// glfgen.hip -- bcf_call_glfgen + errmod_cal for every (site,sample) cell of a tile.
//
// Replaces bam2bcf.c:147-258 and the errmod_cal() it calls (htslib errmod.c).
//
// Mapping: one lane per (site,sample) cell, 256 consecutive cells per workgroup.  A lane
// replays the reference's per-read loop over its own reads, so every order-sensitive
// accumulation (the double sums bsum[] of errmod_cal, the float sums of its epilogue)
// happens in the reference's order and the results are bit-identical to the CPU path.
//
// Data movement: the reads of the workgroup's 256 cells are one contiguous span of the
// `rd`/`epos` arrays (CSR order).  The span is staged into LDS with coalesced 16-byte loads and
// every lane then walks its own slice in LDS; HBM sees each input byte exactly once.
//
// errmod_cal() sorts the n 16-bit codes and walks them from the largest down.  Only the
// relative order of codes with the same base matters (per-base accumulators), and within a
// base the code order is the order of key7 = q<<1|strand.  The sort is therefore replaced by
// a per-base counting pass over the 128 possible key7 values (u8 counters in LDS, one column
// per lane, conflict-free) followed by a walk over the set bits of a 128-bit presence mask.
// Reads carrying the site's reference base -- almost all of them -- are counted directly in
// the per-read loop; only the others are kept as codes (in place, in the lane's LDS slice).
//
// Site-wide bias histograms (bam2bcf.c:228-252) are integer counts: hot bins (mapQ>=59) are
// counted in registers, the rest with LDS atomics; one flush of global atomics per workgroup.
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace bcfgpu {

#define WG 256
#define DEF_MAPQ 20
#define CAP_DIST 25

// seq_nt16_int packed in nibbles: {4,0,1,4,2,4,4,4,3,4,4,4,4,4,4,4}
__device__ __forceinline__ int nt16_int(int c) { return (int)((0x4444444344424104ull >> (4 * (c & 15))) & 7); }

__device__ __forceinline__ int tri(int j, int k) { return k * (k + 1) / 2 + j; }   // j<=k

struct WalkState {
    uint64_t mlo, mhi;
};

// errmod_cal's descending walk for one base: the lane's counts are in s_cnt (column `tid`), the
// presence mask in (mlo,mhi); `left` reads to process.  Returns bsum[base].
__device__ __forceinline__ double walk_base(const uint32_t *s_cnt, const double *s_fk, const double *beta,
                                            int tid, int n, int left, uint64_t mlo, uint64_t mhi)
{
    int rem = 0, rev = 0;
    uint32_t cc = 0, w0 = 0, w1 = 0;
    const double *brow = beta;
    double bs = 0;
    while (__any(left > 0)) {
        if (left > 0) {
            if (rem == 0) {
                uint32_t key;
                if (mhi) { const int k = 63 - __clzll((long long)mhi); mhi &= ~(1ull << k); key = k + 64; }
                else     { const int k = 63 - __clzll((long long)mlo); mlo &= ~(1ull << k); key = k; }
                rem = (s_cnt[(key >> 2) * WG + tid] >> (8 * (key & 3))) & 0xff;
                rev = key & 1;
                brow = beta + ((size_t)(key >> 1) << 16 | (size_t)n << 8);
            }
            const double f = s_fk[rev ? w1 : w0];
            bs += f * brow[cc];
            ++cc; w1 += rev; w0 += 1 - rev;
            --rem; --left;
        }
    }
    return bs;
}

__global__ __launch_bounds__(WG) void glfgen_kernel(const GlfgenParams P)
{
    extern __shared__ __align__(16) unsigned char smem[];
    // LDS carve-up: fk[256] f64 | cnt[32][WG] u32 | rd[cap] u32 | epos[cap] u8 | hist[slots][H_SIZE] i32
    const int cap = P.lds_cap;
    double   *s_fk  = reinterpret_cast<double*>(smem);
    uint32_t *s_cnt = reinterpret_cast<uint32_t*>(smem + 2048);
    uint32_t *s_rd  = reinterpret_cast<uint32_t*>(smem + 2048 + 32 * WG * 4);
    uint8_t  *s_ep  = smem + 2048 + 32 * WG * 4 + ((size_t)cap + 4) * 4;
    int      *s_hist = reinterpret_cast<int*>(smem + 2048 + 32 * WG * 4 + ((size_t)cap + 4) * 4 + (size_t)cap + 32);
    __shared__ unsigned int s_next;

    const int tid = threadIdx.x;
    const long ncells = (long)P.n_sites * P.n_smpl;
    const long cell0 = (long)blockIdx.x * WG;
    const long cell = cell0 + tid;
    const bool active = cell < ncells;
    const int site0 = (int)(cell0 / P.n_smpl);

    s_fk[tid] = P.fk[tid];
    for (int i = tid; i < P.hist_slots * H_SIZE; i += WG) s_hist[i] = 0;

    const int is_indel = P.is_indel;
    int site = 0, ref_base = -1, ref4 = 4;
    uint32_t beg = 0, end = 0;
    if (active) {
        site = (int)(cell / P.n_smpl);
        beg = P.off[cell]; end = P.off[cell + 1];
        if (!is_indel) { ref_base = P.ref16[site]; ref4 = nt16_int(ref_base); }
    }
    const int primary = is_indel ? 0 : ref4;          // the base whose reads are counted on the fly
    int *hist = P.hist_slots ? s_hist + (site - site0) * H_SIZE : P.hist + (long)site * H_SIZE;
    const bool want_epos = (P.fmt_flag & (BCFGPU_INFO_RPB | BCFGPU_INFO_VDB)) != 0;
    const bool want_scr = (P.fmt_flag & (BCFGPU_INFO_SCR | BCFGPU_FMT_SCR)) != 0;
    const uint32_t span_end = P.off[min(cell0 + WG, ncells)];
    const uint32_t n_reads_tot = P.n_reads;

    bool done = !active;
    if (active && end - beg > (uint32_t)cap) { atomicExch(P.err, BCFGPU_E_DEPTH); done = true; }   // cannot be staged
    uint32_t base = P.off[cell0];

    for (;;) {
        // ---- stage [base, lim) of rd/epos into LDS, 16 bytes per lane per load ----
        const uint32_t abase = base & ~3u;                       // 16-byte aligned start of the u32 stream
        const uint32_t lim = min(abase + (uint32_t)cap, span_end);
        {
            const uint32_t nvec = (lim - abase + 3) >> 2;
            const uint4 *src = reinterpret_cast<const uint4*>(P.rd + abase);
            uint4 *dst = reinterpret_cast<uint4*>(s_rd);
            for (uint32_t v = tid; v < nvec; v += WG) {
                if (abase + 4 * v + 3 < n_reads_tot) dst[v] = src[v];
                else {
                    uint32_t t4[4] = {0, 0, 0, 0};
                    for (int k = 0; k < 4; ++k) if (abase + 4 * v + k < n_reads_tot) t4[k] = P.rd[abase + 4 * v + k];
                    dst[v] = make_uint4(t4[0], t4[1], t4[2], t4[3]);
                }
            }
            if (want_epos) {
                const uint32_t ebase = base & ~15u;              // 16-byte aligned start of the u8 stream
                const uint32_t nv16 = (lim - ebase + 15) >> 4;
                const uint4 *es = reinterpret_cast<const uint4*>(P.epos + ebase);
                uint4 *ed = reinterpret_cast<uint4*>(s_ep);
                for (uint32_t v = tid; v < nv16; v += WG) {
                    if (ebase + 16 * v + 15 < n_reads_tot) ed[v] = es[v];
                    else for (int k = 0; k < 16; ++k) s_ep[16 * v + k] = (ebase + 16 * v + k < n_reads_tot) ? P.epos[ebase + 16 * v + k] : 0;
                }
            }
        }
        if (tid == 0) s_next = 0xffffffffu;
        __syncthreads();

        const bool part = !done && beg >= base && end <= lim;    // this lane's slice is resident
        const uint32_t lbeg = part ? beg - abase : 0;            // slice start in s_rd
        const uint32_t ebeg = part ? beg - (base & ~15u) : 0;    // slice start in s_ep
        const uint32_t cnt_raw = part ? end - beg : 0;

        // ---- pass 0: the per-read loop of bcf_call_glfgen (bam2bcf.c:170-253) ----
        #pragma unroll
        for (int k = 0; k < 32; ++k) s_cnt[k * WG + tid] = 0;
        uint64_t qs64 = 0, c64 = 0, mlo = 0, mhi = 0;
        uint32_t adf = 0, adr = 0, cnt4 = 0, mq0 = 0, scr = 0, ori_depth = 0;
        uint32_t t_bq = 0, t_bq2 = 0, t_mq = 0, t_mq2 = 0, t_md = 0, t_md2 = 0;
        uint32_t d_bq = 0, d_bq2 = 0, d_mq = 0, d_mq2 = 0, d_md = 0, d_md2 = 0;
        uint32_t h59 = 0;            // reads with mapQ>=59: ref | alt<<8 | fwd<<16 | rev<<24 (flushed once per lane)
        int n = 0, n_other = 0;
        bool fail = false;
        for (uint32_t i = 0; i < cnt_raw; ++i) {
            const uint32_t w = s_rd[lbeg + i];
            if (w & BCFGPU_RD_SKIP) continue;
            if ((w & BCFGPU_RD_DEL) && !is_indel) continue;
            ++ori_depth;
            const int nt = (w >> 16) & 15;
            const int rev = (w >> 20) & 1;
            int q, b, baseQ, seqQ, is_diff;
            if (is_indel) {
                const uint32_t ax = P.aux[beg + i];
                b = (ax >> 16) & 0x3f;
                baseQ = q = ax & 0xff;
                if (q < P.min_baseQ) { b = 0; q = (int)(w & 0xff); }
                seqQ = (ax >> 8) & 0xff;
                is_diff = (b != 0);
            } else {
                b = nt16_int(nt ? nt : ref_base);
                baseQ = q = (int)(w & 0xff);
                if (q < P.min_baseQ) continue;
                seqQ = 99;
                is_diff = (ref4 < 4 && b == ref4) ? 0 : 1;
            }
            int mapQ = (w >> 8) & 0xff;
            if (mapQ == 255) mapQ = DEF_MAPQ;
            if (!mapQ) mq0++;
            if (q > seqQ) q = seqQ;
            mapQ = mapQ < P.capQ ? mapQ : P.capQ;
            if (q > mapQ) q = mapQ;
            if (q > 63) q = 63;
            if (q < 4) q = 4;
            if (n >= BCFGPU_MAX_DEPTH) { fail = true; break; }
            ++n;
            const uint32_t code = (uint32_t)(q << 5 | rev << 4 | b);
            if ((int)(code & 0xf) == primary) {
                const uint32_t key = (code >> 4) & 0x7f;          // q<<1 | strand
                s_cnt[(key >> 2) * WG + tid] += 1u << (8 * (key & 3));
                if (key < 64) mlo |= 1ull << key; else mhi |= 1ull << (key - 64);
            } else {
                s_rd[lbeg + n_other] = code;                      // n_other <= i: never overtakes the reader
                ++n_other;
            }
            if (want_scr && (w & BCFGPU_RD_SCLIP)) scr++;
            if (b < 4) {
                qs64 += (uint64_t)q << (16 * b);
                if (rev) adr += 1u << (8 * b); else adf += 1u << (8 * b);
            }
            if (b < 8) c64 += 1ull << (8 * b);
            cnt4 += 1u << (8 * (is_diff << 1 | rev));
            int min_dist = (int)(w >> 24);
            if (min_dist > CAP_DIST) min_dist = CAP_DIST;
            const uint32_t dm = is_diff ? ~0u : 0u;
            t_bq += baseQ;    t_bq2 += baseQ * baseQ;        d_bq += baseQ & dm;    d_bq2 += (baseQ * baseQ) & dm;
            t_mq += mapQ;     t_mq2 += mapQ * mapQ;          d_mq += mapQ & dm;     d_mq2 += (mapQ * mapQ) & dm;
            t_md += min_dist; t_md2 += min_dist * min_dist;  d_md += min_dist & dm; d_md2 += (min_dist * min_dist) & dm;
            // bias-test histograms: ibq = (int)(baseQ/60.*60) is the identity on 0..59 (checked in tests)
            const int ibq = baseQ > 59 ? 59 : baseQ;
            const int imq = mapQ > 59 ? 59 : mapQ;
            const int ep = want_epos ? s_ep[ebeg + i] : 0;
            const bool isref = (nt == ref_base);
            if (imq == 59) h59 += (isref ? 1u : 1u << 8) + (rev ? 1u << 24 : 1u << 16);
            else {
                atomicAdd(&hist[(rev ? H_REV_MQS : H_FWD_MQS) + imq], 1);
                atomicAdd(&hist[(isref ? H_REF_MQ : H_ALT_MQ) + imq], 1);
            }
            atomicAdd(&hist[(isref ? H_REF_POS : H_ALT_POS) + ep], 1);
            atomicAdd(&hist[(isref ? H_REF_BQ : H_ALT_BQ) + ibq], 1);
        }
        if (h59) {
            if (h59 & 0xff)         atomicAdd(&hist[H_REF_MQ + 59], (int)(h59 & 0xff));
            if ((h59 >> 8) & 0xff)  atomicAdd(&hist[H_ALT_MQ + 59], (int)((h59 >> 8) & 0xff));
            if ((h59 >> 16) & 0xff) atomicAdd(&hist[H_FWD_MQS + 59], (int)((h59 >> 16) & 0xff));
            if (h59 >> 24)          atomicAdd(&hist[H_REV_MQS + 59], (int)(h59 >> 24));
        }
        if (fail || ori_depth > 0xffff) { atomicExch(P.err, BCFGPU_E_DEPTH); n = 0; c64 = 0; n_other = 0; mlo = mhi = 0; }

        // ---- errmod_cal: descending walk per base ----
        double bsum[5] = {0, 0, 0, 0, 0};
        // (a) the primary base, already counted
        {
            const int cb = primary < 8 ? (int)((c64 >> (8 * primary)) & 0xff) : 0;
            const double bs = walk_base(s_cnt, s_fk, P.beta, tid, n, cb, mlo, mhi);
            #pragma unroll
            for (int b = 0; b < 5; ++b) if (b == primary) bsum[b] = bs;
        }
        // (b) every other base present in some lane of the wave: count its codes, then walk
        #pragma unroll
        for (int b = 0; b < 5; ++b) {
            const int cb = (b != primary) ? (int)((c64 >> (8 * b)) & 0xff) : 0;
            if (!__any(cb > 0)) continue;
            #pragma unroll
            for (int k = 0; k < 32; ++k) s_cnt[k * WG + tid] = 0;
            uint64_t lo = 0, hi = 0;
            if (cb > 0) {
                for (int i = 0; i < n_other; ++i) {
                    const uint32_t code = s_rd[lbeg + i];
                    if ((int)(code & 0xf) != b) continue;
                    const uint32_t key = (code >> 4) & 0x7f;
                    s_cnt[(key >> 2) * WG + tid] += 1u << (8 * (key & 3));
                    if (key < 64) lo |= 1ull << key; else hi |= 1ull << (key - 64);
                }
            }
            const double bs = walk_base(s_cnt, s_fk, P.beta, tid, n, cb, lo, hi);
            if (b != primary) bsum[b] = bs;      // lanes of one wave may belong to sites with different reference bases
        }

        // ---- epilogue of errmod_cal (m=5): float accumulators as in the reference ----
        if (part) {
            int c[5];
            #pragma unroll
            for (int b = 0; b < 5; ++b) c[b] = (int)((c64 >> (8 * b)) & 0xff);
            #pragma unroll
            for (int j = 0; j < 5; ++j) {
                float tmp1 = 0.0f; int tmp2 = 0;
                #pragma unroll
                for (int k = 0; k < 5; ++k) { if (k == j) continue; tmp1 = (float)((double)tmp1 + bsum[k]); tmp2 += c[k]; }
                float v = 0.0f;
                if (n > 0 && tmp2) v = tmp1;
                if (v < 0.0f) v = 0.0f;
                P.cr.p15[(size_t)tri(j, j) * ncells + cell] = v;
                #pragma unroll
                for (int k = j + 1; k < 5; ++k) {
                    const int cjk = c[j] + c[k];
                    float t1 = 0.0f; int t2 = 0;
                    #pragma unroll
                    for (int i = 0; i < 5; ++i) { if (i == j || i == k) continue; t1 = (float)((double)t1 + bsum[i]); t2 += c[i]; }
                    float h = 0.0f;
                    if (n > 0) {
                        const double lh = -4.343 * P.lhet[cjk << 8 | c[k]];
                        h = t2 ? (float)(lh + (double)t1) : (float)lh;
                        if (h < 0.0f) h = 0.0f;
                    }
                    P.cr.p15[(size_t)tri(j, k) * ncells + cell] = h;
                }
            }
            P.cr.qs64[cell] = qs64;
            P.cr.adf[cell] = adf; P.cr.adr[cell] = adr; P.cr.cnt4[cell] = cnt4;
            P.cr.misc[cell] = (mq0 & 0xff) | (scr & 0xff) << 8 | ori_depth << 16;
            uint32_t *sm = P.cr.sums + cell;
            sm[0 * ncells] = t_bq - d_bq; sm[1 * ncells] = t_bq2 - d_bq2; sm[2 * ncells] = d_bq; sm[3 * ncells] = d_bq2;
            sm[4 * ncells] = t_mq - d_mq; sm[5 * ncells] = t_mq2 - d_mq2; sm[6 * ncells] = d_mq; sm[7 * ncells] = d_mq2;
            sm[8 * ncells] = t_md - d_md; sm[9 * ncells] = t_md2 - d_md2; sm[10 * ncells] = d_md; sm[11 * ncells] = d_md2;
            done = true;
        }
        // ---- next round: the first cell whose reads are not resident yet (deep tiles only) ----
        if (!done) atomicMin(&s_next, beg);
        __syncthreads();
        const uint32_t nb = s_next;
        if (nb == 0xffffffffu) break;
        base = nb;
        __syncthreads();
    }

    // ---- flush the workgroup's histograms ----
    if (P.hist_slots) {
        const int nslot = min(P.hist_slots, P.n_sites - site0);
        for (int i = tid; i < nslot * H_SIZE; i += WG) {
            const int v = s_hist[i];
            if (v) atomicAdd(&P.hist[(long)site0 * H_SIZE + i], v);
        }
    }
}

size_t glfgen_lds_bytes(int cap, int hist_slots)
{
    return 2048 + 32 * WG * 4 + ((size_t)cap + 4) * 4 + (size_t)cap + 32 + (size_t)hist_slots * H_SIZE * sizeof(int);
}

void launch_glfgen(const GlfgenParams &p, hipStream_t s)
{
    const long ncells = (long)p.n_sites * p.n_smpl;
    if (ncells == 0) return;
    const int grid = (int)((ncells + WG - 1) / WG);
    const size_t lds = glfgen_lds_bytes(p.lds_cap, p.hist_slots);
    static size_t lds_attr = 0;
    if (lds > lds_attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(glfgen_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lds_attr = lds;
    }
    hipLaunchKernelGGL(glfgen_kernel, dim3(grid), dim3(WG), lds, s, p);
}

}  // namespace bcfgpu

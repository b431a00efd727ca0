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
// per lane, conflict-free) and a descending walk over the set bits of the lane's 128-bit
// key-presence mask (a key-major variant that iterates the wave's union of keys was measured
// slower: random mapQ values make the union ~10x larger than a lane's own key set).
// Reads carrying the site's reference base -- almost all of them -- are counted directly in
// the per-read loop; the others (exactly the "diff" reads of the I16 annotations) are kept, in
// place, in the lane's LDS slice as a packed word and handled by a second, short loop.
//
// Site-wide bias histograms (bam2bcf.c:228-252) are integer counts: hot bins (mapQ>=59) are
// counted in registers, the rest with LDS atomics; one flush of global atomics per workgroup.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "kernels.h"

namespace bcfgpu {

#define WG 256
#define DEF_MAPQ 20
#define CAP_DIST 25

// seq_nt16_int packed in nibbles: {4,0,1,4,2,4,4,4,3,4,4,4,4,4,4,4}
__device__ __forceinline__ int nt16_int(int c) { return (int)((0x4444444344424104ull >> (4 * (c & 15))) & 7); }
__device__ __forceinline__ int tri(int j, int k) { return k * (k + 1) / 2 + j; }   // j<=k

__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o);
    return v;
}

// LDS layout (bytes): fk[264] f64 | slots[4][WG] u32 | rd[cap+4] u32 | epos[cap+32] u8 | hist | site totals
#define LDS_FK   0
#define LDS_CNT  2112
#define NSLOT    16
#ifndef FU
#define FU       4         // source elements per trip of the slot counting
#endif
#ifndef WSTEP
#define WSTEP    3         // reads of one (quality, strand) run taken per step of the errmod walk
#endif
#define LDS_RD   (LDS_CNT + (NSLOT / 4) * WG * 4)

// packed "other" (non-primary = diff) read: baseQ:8 | mapQ(capped):6 | q:6 | b:4 | rev:1 | min_dist:5
#define OW_PACK(baseQ, mapQ, q, b, rev, md) \
    ((uint32_t)(baseQ) | (uint32_t)(mapQ) << 8 | (uint32_t)(q) << 14 | (uint32_t)(b) << 20 | (uint32_t)(rev) << 24 | (uint32_t)(md) << 25)

// Counts, as u8 in the lane's slot column, of the reads with the NSLOT/2 highest qualities of the mask `qm` (bit q = some
// read of this base has quality q): slot 2*rank(q) holds the reverse-strand reads of q, slot 2*rank(q)+1 the forward
// ones -- the descending order of errmod_cal's sorted codes q<<5|strand<<4|base (bam2bcf.c:203).  `src(j)` returns
// key7 = q<<1|strand of source element j, or -1.
template <bool FIRST, class Src>
__device__ __forceinline__ void fill_slots(uint32_t *s_slot, uint64_t qm, int tid, Src src, int nsrc)
{
    #pragma unroll
    for (int k = 0; k < NSLOT / 4; ++k) s_slot[k * WG + tid] = 0;
    // FU source elements per trip: their reads are in flight before the first count is added
    for (int j = 0; __any(j < nsrc); j += FU) {
        int k4[FU];
        #pragma unroll
        for (int u = 0; u < FU; ++u) k4[u] = j + u < nsrc ? src(j + u) : -1;
        #pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int key = k4[u];
            const int q = (key >> 1) & 63;
            if (key >= 0 && (FIRST || ((qm >> q) & 1ull))) {       // FIRST: the mask still holds every quality of the source
                const int r = 2 * __popcll((qm >> q) >> 1) + 1 - (key & 1);
                if (r < NSLOT) atomicAdd(&s_slot[(r >> 2) * WG + tid], 1u << (8 * (r & 3)));
            }
        }
    }
}

// Descending walk of errmod_cal for one base: every lane steps through its own reads from the highest (quality,
// strand) key down.  errmod_cal sorts the codes; here the lane's qualities are the set bits of a 64-bit mask and the
// number of reads per (quality, strand) sits in NSLOT rank-ordered u8 slots (a 128-entry table per lane would cost a
// third of the occupancy).  A small state machine: when the reads of the current key are used up, move to the next
// slot, taking the next set bit at every other step; a lane with more than NSLOT/2 distinct qualities refills its
// slots from the source for the remaining ones (binned base qualities give a handful).  `n` selects the beta row of
// the lane, `left` is the number of reads of this base.
// The state machine runs one step (one or two reads of a run) ahead of the summation: the beta values (gathers from a
// 32 MB table, L2 latency) and the fk factors (LDS) of the next step are requested before the current ones are added,
// in the reference's order.
template <class Src>
__device__ __forceinline__ double walk_keys(uint32_t *s_slot, uint64_t qm, const double *s_fk,
                                            const double *beta, int tid, int n, int left, Src src, int nsrc)
{
    fill_slots<true>(s_slot, qm, tid, src, nsrc);
    int rem = 0, pend = 0, pleft = left, r = 0;              // r: next slot pair (= rank of the next quality)
    uint32_t rev = 0, cc = 0, w0 = 0, w1 = 0;
    const char *bbase = reinterpret_cast<const char*>(beta);
    const uint32_t brow = (uint32_t)n << 3;                   // byte offset of beta[0][0][n] (stored q, k, n: tables.cpp); 32 MiB
    uint32_t boff = brow;
    double bs = 0;
    // Loads are issued unconditionally (inactive lanes read a valid dummy) and their results never cross a divergent
    // join, so that three gathers stay in flight; the (divergent) state update carries no memory results.
    #define SLOT_PAIR(i) (s_slot[(((i) & (NSLOT / 2 - 1)) >> 1) * WG + tid])      /* the dword holding pair i */
    uint32_t two_nx = SLOT_PAIR(0);
    #define WALK_PRODUCE(B, F, N) do { \
        if (pleft > 0 && rem == 0) { \
            if (pend > 0) { rev = 0; rem = pend; pend = 0; }         /* the forward-strand reads of the same quality */ \
            else { \
                if (r == NSLOT / 2) { fill_slots<false>(s_slot, qm, tid, src, nsrc); r = 0; two_nx = SLOT_PAIR(0); } \
                const int curq = 63 - __clzll((long long)(qm | 1ull)); \
                qm &= ~(1ull << curq); \
                const uint32_t two = (two_nx >> (16 * (r & 1))) & 0xffffu; \
                ++r; \
                const int cr = (int)(two & 0xff), cf = (int)(two >> 8); \
                rev = cr ? 1u : 0u; rem = cr ? cr : cf; pend = cr ? cf : 0; \
                if (rem == 0) rem = pleft;                           /* cannot happen: counts and mask agree */ \
                boff = brow + ((uint32_t)curq << 19); \
            } \
        } \
        two_nx = SLOT_PAIR(r);                                       /* for the next advance */ \
        { \
            /* one step takes the next read and, inside a run of equal (quality, strand), up to WSTEP - 1 more */ \
            const uint32_t na_ = pleft > 0 ? (uint32_t)min(rem, WSTEP) : 0u; \
            const uint32_t o0 = boff + (cc << 11); \
            const uint32_t wi = rev ? w1 : w0; \
            _Pragma("unroll") \
            for (int k_ = 0; k_ < WSTEP; ++k_) { \
                const bool a_ = (uint32_t)k_ < na_; \
                B[k_] = *reinterpret_cast<const double*>(bbase + (a_ ? o0 + ((uint32_t)k_ << 11) : boff)); \
                F[k_] = s_fk[a_ ? wi + (uint32_t)k_ : 0u]; \
            } \
            N = na_; \
            cc += na_; w1 += rev ? na_ : 0u; w0 += rev ? 0u : na_; rem -= (int)na_; pleft -= (int)na_; \
        } } while (0)
    #define WALK_CONSUME(B, F, N) do { \
        _Pragma("unroll") \
        for (int k_ = 0; k_ < WSTEP; ++k_) { const double t_ = F[k_] * B[k_]; bs = (uint32_t)k_ < N ? bs + t_ : bs; } \
        left -= (int)N; } while (0)
    double bx[WSTEP], fx[WSTEP], by[WSTEP], fy[WSTEP];
    uint32_t nx, ny;
    WALK_PRODUCE(bx, fx, nx);
    while (__any(left > 0)) {
        WALK_PRODUCE(by, fy, ny); WALK_CONSUME(bx, fx, nx);
        WALK_PRODUCE(bx, fx, nx); WALK_CONSUME(by, fy, ny);
    }
    #undef WALK_PRODUCE
    #undef WALK_CONSUME
    #undef SLOT_PAIR
    return bs;
}

template <bool INDEL, bool LDS_HIST>
__global__ __launch_bounds__(WG) void glfgen_kernel(const GlfgenParams P)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int cap = P.lds_cap;
    double   *s_fk  = reinterpret_cast<double*>(smem + LDS_FK);
    uint32_t *s_cnt = reinterpret_cast<uint32_t*>(smem + LDS_CNT);
    uint32_t *s_rd  = reinterpret_cast<uint32_t*>(smem + LDS_RD);
    uint8_t  *s_ep  = smem + LDS_RD + ((size_t)cap + 4) * 4;
    int      *s_hist = reinterpret_cast<int*>(smem + LDS_RD + ((size_t)cap + 4) * 4 + (size_t)cap + 32);
    unsigned long long *s_tot = reinterpret_cast<unsigned long long*>(s_hist + (size_t)P.hist_slots * H_SIZE);   // [slots][12]
    __shared__ unsigned int s_next;

    const int tid = threadIdx.x;
    const long ncells = (long)P.n_sites * P.n_smpl;
    const long cell0 = (long)blockIdx.x * WG;
    const int site0 = (int)(cell0 / P.n_smpl);

    const long cell = cell0 + tid;
    const bool active = cell < ncells;

    s_fk[tid] = P.fk[tid];
    if (tid < 8) s_fk[256 + tid] = 0.0;               // read-ahead slack of the walk (never used in a sum)
    if (LDS_HIST) {
        for (int i = tid; i < P.hist_slots * H_SIZE; i += WG) s_hist[i] = 0;
        for (int i = tid; i < P.hist_slots * 12; i += WG) s_tot[i] = 0;
    }

    int site = 0, ref_base = -1, ref4 = 4;
    uint32_t beg = 0, end = 0;
    if (active) {
        site = (int)(cell / P.n_smpl);
        beg = P.off[cell]; end = P.off[cell + 1];
        if (!INDEL) { ref_base = P.ref16[site]; ref4 = nt16_int(ref_base); }
    }
    const int primary = INDEL ? 0 : ref4;             // the base whose reads are counted on the fly
    uint32_t prim_nt = 0;                             // SNP: the nt16 codes that show it (code 0 stands for the reference base)
    if (!INDEL) {
        #pragma unroll
        for (int c = 0; c < 16; ++c) if (nt16_int(c ? c : ref_base) == primary) prim_nt |= 1u << c;
    }
    int *ghist = P.hist + (long)site * H_SIZE;
    int *lhist = s_hist + (site - site0) * H_SIZE;
    const bool want_epos = (P.fmt_flag & (BCFGPU_INFO_RPB | BCFGPU_INFO_VDB)) != 0;
    const bool want_scr = (P.fmt_flag & (BCFGPU_INFO_SCR | BCFGPU_FMT_SCR)) != 0;
    const uint32_t span_end = P.off[min(cell0 + WG, ncells)];
    const uint32_t n_reads_tot = P.n_reads;
    const int min_baseQ = P.min_baseQ, capQ = P.capQ;

    #define HIST_ADD(idx, v) do { if (LDS_HIST) atomicAdd(&lhist[idx], v); else atomicAdd(&ghist[idx], v); } while (0)

    bool done = !active;
    if (active && end - beg > (uint32_t)cap) { atomicExch(P.err, BCFGPU_E_DEPTH); done = true; }   // cannot be staged
    uint32_t base = P.off[cell0];

    for (;;) {
        // ---- stage [base, lim) of rd/epos into LDS, 16 bytes per lane per load ----
        const uint32_t abase = base & ~3u;                       // 16-byte aligned start of the u32 stream
        const uint32_t ebase = base & ~15u;                      // 16-byte aligned start of the u8 stream
        const uint32_t lim = min(abase + (uint32_t)cap, span_end);
        {
            // All loads of a batch -- eight 16-byte vectors of rd and two of epos per lane, which covers a whole span at
            // the usual LDS capacity -- are in flight before the first LDS store: one memory round trip per batch.
            const uint32_t nvec = (lim - abase + 3) >> 2;
            const uint32_t nv16 = want_epos ? (lim - ebase + 15) >> 4 : 0;
            const uint4 *src = reinterpret_cast<const uint4*>(P.rd + abase);
            const uint4 *es = reinterpret_cast<const uint4*>(P.epos + ebase);
            uint4 *dst = reinterpret_cast<uint4*>(s_rd);
            uint4 *ed = reinterpret_cast<uint4*>(s_ep);
            for (uint32_t v0 = tid, w0 = tid; v0 < nvec || w0 < nv16; v0 += 8 * WG, w0 += 2 * WG) {
                uint4 r[8], e[2];
                #pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t v = v0 + k * WG;
                    r[k] = make_uint4(0, 0, 0, 0);
                    if (v < nvec) {
                        if (abase + 4 * v + 3 < n_reads_tot) r[k] = src[v];
                        else {
                            uint32_t t4[4] = {0, 0, 0, 0};
                            for (int j = 0; j < 4; ++j) if (abase + 4 * v + j < n_reads_tot) t4[j] = P.rd[abase + 4 * v + j];
                            r[k] = make_uint4(t4[0], t4[1], t4[2], t4[3]);
                        }
                    }
                }
                #pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const uint32_t v = w0 + k * WG;
                    e[k] = make_uint4(0, 0, 0, 0);
                    if (v < nv16) {
                        if (ebase + 16 * v + 15 < n_reads_tot) e[k] = es[v];
                        else {
                            uint32_t t4[4] = {0, 0, 0, 0};
                            for (int j = 0; j < 16; ++j)
                                if (ebase + 16 * v + j < n_reads_tot) t4[j >> 2] |= (uint32_t)P.epos[ebase + 16 * v + j] << (8 * (j & 3));
                            e[k] = make_uint4(t4[0], t4[1], t4[2], t4[3]);
                        }
                    }
                }
                #pragma unroll
                for (int k = 0; k < 8; ++k) { const uint32_t v = v0 + k * WG; if (v < nvec) dst[v] = r[k]; }
                #pragma unroll
                for (int k = 0; k < 2; ++k) { const uint32_t v = w0 + k * WG; if (v < nv16) ed[v] = e[k]; }
            }
        }
        if (tid == 0) s_next = 0xffffffffu;
        __syncthreads();

        const bool part = !done && beg >= base && end <= lim;    // this lane's slice is resident
        const uint32_t lbeg = part ? beg - abase : 0;            // slice start in s_rd
        const uint32_t ebeg = part ? beg - ebase : 0;            // slice start in s_ep
        const uint32_t cnt_raw = (part && !(P.ablate & 4)) ? end - beg : 0;
        const bool skip_hist = (P.ablate & 1) != 0;

        // ---- pass 0: the per-read loop of bcf_call_glfgen (bam2bcf.c:170-253) ----
        uint64_t qmask = 0;          // qualities seen among the reads of the primary base
        uint32_t qs_prim = 0, prim_rev = 0;   // their quality sum and reverse-strand count
        uint32_t mq0 = 0, scr = 0, ori_depth = 0, n_rev = 0;
        uint32_t t_bqmd = 0, t_mq = 0, t_bq2 = 0, t_mq2 = 0, t_md2 = 0;   // totals: baseQ | min_dist<<16, mapQ, squares
        uint32_t h59 = 0;            // reads with mapQ>=59: ref | alt<<8 | fwd<<16 | rev<<24
        int n = 0, n_other = 0, n_prim = 0;
        bool fail = false;
        // The loop body is written without early exits: one predicate (`ok`) guards a single divergent region, and the
        // per-base updates are selects, because a wavefront pays for every branch any of its lanes takes.
        // The next record is fetched while the current one is processed (the in-place writes below stay behind index i;
        // the read one past the slice stays inside the staged span's slack).
        uint32_t w_nx = s_rd[lbeg];
        int ep_nx = want_epos ? s_ep[ebeg] : 0;
        for (uint32_t i = 0; i < cnt_raw; ++i) {
            const uint32_t w = w_nx;
            const int ep = ep_nx;
            w_nx = s_rd[lbeg + i + 1];
            if (want_epos) ep_nx = s_ep[ebeg + i + 1];
            const int nt = (w >> 16) & 15;
            int q, b, baseQ, seqQ;
            bool ok;
            if (INDEL) {
                const uint32_t ax = P.aux[beg + i];
                b = (ax >> 16) & 0xf;                 // 0..4 after bcf_call_gap_prep (bam2bcf_indel.c:449-456)
                baseQ = q = ax & 0xff;
                if (q < min_baseQ) { b = 0; q = (int)(w & 0xff); }
                seqQ = (ax >> 8) & 0xff;
                ok = !(w & BCFGPU_RD_SKIP);
                ori_depth += ok;
            } else {
                b = 0;                                // SNP: looked up below, only for the reads that need it
                baseQ = q = (int)(w & 0xff);
                seqQ = 99;
                const bool seen = !(w & (BCFGPU_RD_SKIP | BCFGPU_RD_DEL));
                ori_depth += seen;
                ok = seen && q >= min_baseQ;
            }
            if (ok && n >= BCFGPU_MAX_DEPTH) { fail = true; ok = false; }
            // register accumulators: unconditional arithmetic; a rejected read is a zero record and contributes zeros
            const uint32_t okm = ok ? 1u : 0u;
            const uint32_t wz = ok ? w : 0u;
            if (!ok) baseQ = 0;
            const uint32_t rev = (wz >> 20) & 1;
            int mapQ = (wz >> 8) & 0xff;
            if (mapQ == 255) mapQ = DEF_MAPQ;
            mq0 += (mapQ == 0) & okm;
            q = min(q, seqQ);
            mapQ = min(mapQ, capQ);
            q = max(min(min(q, mapQ), 63), 4);
            n += okm;
            n_rev += rev;
            const int min_dist = min((int)(wz >> 24), CAP_DIST);
            const uint32_t key = (uint32_t)(q << 1) | rev;       // (code>>4)&0x7f of bam2bcf.c:203
            // SNP: bit c of prim_nt says whether a read showing nt16 code c carries the primary base (one bit-field
            // extract instead of the nt16 -> 0..4 lookup; the lookup itself is left to the few non-primary reads)
            const bool prim = ok && (INDEL ? b == primary : ((prim_nt >> nt) & 1u) != 0);
            // reads of the primary base: quality mask, QS and strand count (the other bases' come from their stored words)
            qmask |= (uint64_t)(prim ? 1u : 0u) << q;
            qs_prim += prim ? (uint32_t)q : 0u;
            prim_rev += prim ? rev : 0u;
            if (want_scr) scr += (wz >> 21) & 1;
            t_bqmd += (uint32_t)baseQ | (uint32_t)min_dist << 16;
            t_mq += mapQ;
            t_bq2 += __umul24(baseQ, baseQ); t_mq2 += __umul24(mapQ, mapQ); t_md2 += __umul24(min_dist, min_dist);   // all < 256
            const int ibq = min(baseQ, 59);
            const int imq = min(mapQ, 59);
            const bool isref = (nt == ref_base);
            // All LDS updates of the read come last, after the prefetched next record has arrived: LDS operations
            // complete in order, so waiting for that record any later would also wait for these atomics.
            asm volatile("" : "+v"(qs_prim), "+v"(prim_rev), "+v"(t_bq2), "+v"(t_mq2), "+v"(t_md2), "+v"(qmask)
                            : "v"(w_nx), "v"(ep_nx) : "memory");
            if (!ok) continue;
            // the key of a primary read goes, compacted, into the lane's epos slice (byte n_prim <= i is behind the reader)
            // (written for every accepted read: a non-primary one is overwritten by the next primary key or lies past the list)
            s_ep[ebeg + n_prim] = (uint8_t)key;
            n_prim += prim ? 1 : 0;
            if (!prim) {
                if (!INDEL) b = nt16_int(nt ? nt : ref_base);
                s_rd[lbeg + n_other] = OW_PACK(baseQ, mapQ, q, b, rev, min_dist);    // n_other <= i: behind the reader
                ++n_other;
            }
            // bias-test histograms: ibq = (int)(baseQ/60.*60) is the identity on 0..59 (checked in tests)
            if (skip_hist) continue;
            if (imq == 59) h59 += (isref ? 1u : 1u << 8) + (rev ? 1u << 24 : 1u << 16);
            else {
                HIST_ADD((rev ? H_REV_MQS : H_FWD_MQS) + imq, 1);
                HIST_ADD((isref ? H_REF_MQ : H_ALT_MQ) + imq, 1);
            }
            HIST_ADD((isref ? H_REF_POS : H_ALT_POS) + ep, 1);
            HIST_ADD((isref ? H_REF_BQ : H_ALT_BQ) + ibq, 1);
        }
        if (h59) {
            if (h59 & 0xff)         HIST_ADD(H_REF_MQ + 59, (int)(h59 & 0xff));
            if ((h59 >> 8) & 0xff)  HIST_ADD(H_ALT_MQ + 59, (int)((h59 >> 8) & 0xff));
            if ((h59 >> 16) & 0xff) HIST_ADD(H_FWD_MQS + 59, (int)((h59 >> 16) & 0xff));
            if (h59 >> 24)          HIST_ADD(H_REV_MQS + 59, (int)(h59 >> 24));
        }
        if (fail || ori_depth > 0xffff) {
            atomicExch(P.err, BCFGPU_E_DEPTH);
            n = 0; n_other = 0; n_prim = 0; n_rev = 0; qmask = 0; qs_prim = 0; prim_rev = 0;
        }
        // the other reads (exactly the "diff" reads of the I16 annotations): their annotation sums and their share of
        // QS / ADF / ADR (bam2bcf.c:208-215), from the stored words
        uint64_t qs64 = 0;           // QS[0..3], 16 bits each
        uint64_t ad64 = 0;           // ADF[0..3] | ADR[0..3]<<32, 8 bits each
        uint32_t n_b4 = 0;           // reads showing neither A, C, G nor T
        uint32_t d_bqmd = 0, d_mq = 0, d_bq2 = 0, d_mq2 = 0, d_md2 = 0, d_fwd = 0, d_rev = 0;
        if (__any(n_other > 0)) {
            for (int i = 0; i < n_other; ++i) {
                const uint32_t ow = s_rd[lbeg + i];
                const uint32_t baseQ = ow & 0xff, mapQ = (ow >> 8) & 0x3f, md = ow >> 25;
                d_bqmd += baseQ | md << 16;
                d_mq += mapQ;
                d_bq2 += __umul24(baseQ, baseQ); d_mq2 += __umul24(mapQ, mapQ); d_md2 += __umul24(md, md);
                const uint32_t rev = (ow >> 24) & 1;
                d_rev += rev; d_fwd += 1 - rev;
                const uint32_t ob = (ow >> 20) & 0xf, oq = (ow >> 14) & 0x3f;
                if (ob < 4) {
                    qs64 += (uint64_t)oq << (16 * ob);
                    ad64 += 1ull << (8 * ob + 32 * rev);
                } else ++n_b4;
            }
        }
        if (primary < 4) {
            qs64 += (uint64_t)qs_prim << (16 * primary);
            ad64 += (uint64_t)((uint32_t)n_prim - prim_rev) << (8 * primary) | (uint64_t)prim_rev << (8 * primary + 32);
        } else n_b4 += (uint32_t)n_prim;
        // per-base counts c[0..4] (errmod_cal's aux.c)
        int c[5];
        #pragma unroll
        for (int b = 0; b < 4; ++b) c[b] = (int)((ad64 >> (8 * b)) & 0xff) + (int)((ad64 >> (8 * b + 32)) & 0xff);
        c[4] = (int)n_b4;
        const bool skip_walk = (P.ablate & 2) != 0;

        // ---- errmod_cal: descending walk per base ----
        double bsum[5] = {0, 0, 0, 0, 0};
        // (a) the primary base
        if (!skip_walk && !(P.ablate & 16384)) {
            const uint8_t *kb = s_ep + ebeg;                 // the lane's n_prim key bytes
            const double bs = walk_keys(s_cnt, qmask, s_fk, P.beta, tid, n, n_prim,
                                        [kb](int j) { return (int)kb[j]; }, n_prim);
            #pragma unroll
            for (int b = 0; b < 5; ++b) if (b == primary) bsum[b] = bs;
        }
        // (b) the other bases present in the wave
        if (!skip_walk && !(P.ablate & 128) && __any(n_other > 0)) {
            #pragma unroll
            for (int b = 0; b < 5; ++b) {
                const int cb = (b != primary) ? c[b] : 0;
                if (!__any(cb > 0)) continue;
                const uint32_t *ow_p = s_rd + lbeg;
                // key7 of the lane's i-th other read if it shows base b, else -1
                auto src = [ow_p, b](int i) {
                    const uint32_t ow = ow_p[i];
                    int bb = (ow >> 20) & 0xf;
                    if (bb > 4) bb = 4;
                    return bb == b ? (int)(((ow >> 14) & 0x3f) << 1 | ((ow >> 24) & 1)) : -1;
                };
                uint64_t qm = 0;
                if (cb > 0)
                    for (int i = 0; i < n_other; ++i) { const int key = src(i); if (key >= 0) qm |= 1ull << (key >> 1); }
                const double bs = walk_keys(s_cnt, qm, s_fk, P.beta, tid, n, cb, src, cb > 0 ? n_other : 0);
                if (b != primary) bsum[b] = bs;      // lanes of one wave may belong to sites with different reference bases
            }
        }

        // ---- epilogue of errmod_cal (m=5): float accumulators as in the reference ----
        if (part) {
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
            // anno[0..3]: ref/alt x fwd/rev counts.  "diff" reads are exactly the non-primary ones when the
            // reference base is A/C/G/T (or at indel sites); with an N reference every read is a diff read.
            const bool all_diff = (!INDEL && ref4 >= 4);
            const uint32_t n_fwd = (uint32_t)n - n_rev;
            if (all_diff) { d_fwd = n_fwd; d_rev = n_rev; d_bqmd = t_bqmd; d_mq = t_mq; d_bq2 = t_bq2; d_mq2 = t_mq2; d_md2 = t_md2; }
            const uint32_t cnt4 = (n_fwd - d_fwd) | (n_rev - d_rev) << 8 | d_fwd << 16 | d_rev << 24;
            P.cr.qs64[cell] = qs64;
            P.cr.adf[cell] = (uint32_t)ad64; P.cr.adr[cell] = (uint32_t)(ad64 >> 32); P.cr.cnt4[cell] = cnt4;
            P.cr.misc[cell] = (mq0 & 0xff) | (scr & 0xff) << 8 | ori_depth << 16;
            // anno[4..15] only feed the site totals of bcf_call_combine (bam2bcf.c:718-727): exact integers, so they are
            // summed here (LDS per workgroup, one global atomic per workgroup and site) instead of crossing HBM per cell
            const uint32_t t_bq = t_bqmd & 0xffff, t_md = t_bqmd >> 16, d_bq = d_bqmd & 0xffff, d_md = d_bqmd >> 16;
            const uint32_t v12[12] = { t_bq - d_bq, t_bq2 - d_bq2, d_bq, d_bq2, t_mq - d_mq, t_mq2 - d_mq2, d_mq, d_mq2,
                                       t_md - d_md, t_md2 - d_md2, d_md, d_md2 };
            unsigned long long *tot = LDS_HIST ? s_tot + (site - site0) * 12 : P.site_sums + (size_t)site * 12;
            #pragma unroll
            for (int j = 0; j < 12; ++j) if (v12[j]) atomicAdd(&tot[j], (unsigned long long)v12[j]);
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
    if (LDS_HIST) {
        const int nslot = min(P.hist_slots, P.n_sites - site0);
        for (int i = tid; i < nslot * H_SIZE; i += WG) {
            const int v = s_hist[i];
            if (v) atomicAdd(&P.hist[(long)site0 * H_SIZE + i], v);
        }
        for (int i = tid; i < nslot * 12; i += WG) {
            const unsigned long long v = s_tot[i];
            if (v) atomicAdd(&P.site_sums[(size_t)site0 * 12 + i], v);
        }
    }
    #undef HIST_ADD
}

size_t glfgen_lds_bytes(int cap, int hist_slots)
{
    return LDS_RD + ((size_t)cap + 4) * 4 + (size_t)cap + 32 + (size_t)hist_slots * (H_SIZE * sizeof(int) + 12 * 8);
}

template <bool INDEL, bool LDS_HIST>
static void launch_one(const GlfgenParams &p, hipStream_t s, int grid, size_t lds)
{
    static size_t lds_attr = 0;
    if (lds > lds_attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(glfgen_kernel<INDEL, LDS_HIST>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lds_attr = lds;
    }
    hipLaunchKernelGGL((glfgen_kernel<INDEL, LDS_HIST>), dim3(grid), dim3(WG), lds, s, p);
}

void launch_glfgen(const GlfgenParams &p, hipStream_t s)
{
    const long ncells = (long)p.n_sites * p.n_smpl;
    if (ncells == 0) return;
    const int grid = (int)((ncells + WG - 1) / WG);
    size_t lds = glfgen_lds_bytes(p.lds_cap, p.hist_slots);
    { const char *e = getenv("BCFGPU_LDS_PAD"); if (e) lds += (size_t)atoi(e); }   // diagnostics: lower the occupancy
    if (p.is_indel) { if (p.hist_slots) launch_one<true, true>(p, s, grid, lds); else launch_one<true, false>(p, s, grid, lds); }
    else            { if (p.hist_slots) launch_one<false, true>(p, s, grid, lds); else launch_one<false, false>(p, s, grid, lds); }
}

}  // namespace bcfgpu

// glfgen.hip -- bcf_call_glfgen + errmod_cal for every (site,sample) cell of a tile.
//
// Replaces bam2bcf.c:147-258 and the errmod_cal() it calls (htslib errmod.c).
//
// A workgroup owns 256 consecutive (site,sample) cells, whose reads are one contiguous span of the `rd`/`epos`
// arrays (CSR order).  The work of bcf_call_glfgen's per-read loop splits by what it feeds:
//
//   phase A, one lane per READ (read-parallel, every lane busy whatever the depths of the cells):
//     the span is read straight from HBM, four consecutive reads per lane and trip (one 16-byte load of `rd`, one
//     4-byte load of `epos`), each input byte exactly once.  Per read: the filters (bam2bcf.c:173-194), the quality
//     arithmetic (:196-203), and everything that only feeds SITE totals -- the I16 sums anno[4..15] (:221-226),
//     ori_depth, mq0 and the bias-test histograms (:228-252) -- which therefore never needs to know the read's cell:
//     per-lane partial sums, one reduction per workgroup and site.  What the cell needs of the read is 12 bits, left in
//     LDS as a u16 key at the read's position in the span:  strand | q<<1 | base<<7 | softclip<<10 | primary base<<11
//     (0 = read rejected).
//   phase B, one lane per CELL: the lane walks its own slice of keys: per-base counts, QS, ADF/ADR, DP4 counts and
//     errmod_cal, whose order-sensitive double sums are replayed in the reference's order (bit-identical results).
//
// LDS holds 2 bytes per read instead of the 5 of the raw tile, which is what lets four workgroups share a CU.
//
// errmod_cal() sorts the n 16-bit codes and walks them from the largest down.  Only the relative order of codes with
// the same base matters (per-base accumulators), and within a base the code order is the order of key7 = q<<1|strand:
// a counting pass per lane (count_runs) and a branch-free walk over the (quality, strand) runs (walk_runs).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "kernels.h"

namespace bcfgpu {

// Diagnostics build: cycles (s_memtime) every wavefront spends between the kernel's phase boundaries, summed into
// P.stamps[0..8] (0: prologue up to phase A, 1: phase A, 2: barrier, 3: partial sums + slice set-up, 4: pass 1,
// 5: walk of the primary base, 6: other bases, 7: epilogue, 8: flush, 9: slot fill of the primary base); tools/stamps.sh prints them.
#ifdef BCFGPU_DIAG
#define GLF_STAMP_DECL unsigned long long stamp_t_ = __builtin_amdgcn_s_memtime();
#define GLF_STAMP(i_) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0 && P.stamps) atomicAdd(&P.stamps[i_], now_ - stamp_t_); stamp_t_ = now_; }
#else
#define GLF_STAMP_DECL
#define GLF_STAMP(i_)
#endif

#define WG 256
#ifndef GLF_WAVES
#define GLF_WAVES 5          // wavefronts per SIMD the register budget is held to
#endif
#define DEF_MAPQ 20
#define CAP_DIST 25

// seq_nt16_int packed in nibbles: {4,0,1,4,2,4,4,4,3,4,4,4,4,4,4,4}
#define NT16_INT_TBL 0x4444444344424104ull
__device__ __forceinline__ int nt16_int(int c) { return (int)((NT16_INT_TBL >> (4 * (c & 15))) & 7); }
__device__ __forceinline__ int tri(int j, int k) { return k * (k + 1) / 2 + j; }   // j<=k

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// LDS layout (bytes): fk[264] f64 | slots[NSLOT][WG] u32 (a dword per quality rank) | hist [slots][HP_SIZE] 2 x u16 | site totals [slots][SITE_NSUM] u64 | keys u16[cap+8]
#define LDS_FK   0
#define LDS_CNT  2112
#ifndef NRANK
#define NRANK 10         // quality ranks a lane counts per round (count_runs): a dword each in LDS (9: 2.75, 10: 2.63, 11: 2.70 ms)
#endif
#define NSLOT    NRANK
#ifndef FU
#define FU       4         // source elements per trip of the slot counting
#endif
#define LDS_HIST_OFF (LDS_CNT + NSLOT * WG * 4)
// The workgroup's copy of a site's bias-test histograms, two 16-bit counters per dword (a workgroup has fewer than 2^16
// reads): dword i < 220 = bin i of the REF arrays (kernels.h: POS, MQ, BQ) in the low half and of the ALT arrays in the high
// half; dword 220 + mq = the forward- and the reverse-strand mapQ histogram.  Half the LDS of plain counters, and the
// REF / ALT choice is the increment instead of an address.
#define HP_SIZE 280
#define HP_MQS  220
#define NPART 12           // per-lane partial sums of phase A: the I16 site totals anno[4..15]

// the u16 key phase A leaves for phase B
#define KEY_PACK(rev, q, b, sc) ((uint32_t)(rev) | (uint32_t)(q) << 1 | (uint32_t)(b) << 7 | (uint32_t)(sc) << 10)
#define KEY_REV(k)  ((k) & 1u)
#define KEY_Q(k)    (((k) >> 1) & 63u)
#define KEY_B(k)    (((k) >> 7) & 7u)
#define KEY_SC(k)   (((k) >> 10) & 1u)
#define KEY_PRIM    0x800u     // the read shows the cell's primary base (the reference base; type 0 at indel sites)

// errmod_cal() sorts the cell's codes q<<5|strand<<4|base and walks them from the largest down; only the order among the
// reads of one base matters (per-base accumulators), and there the order is that of key7 = q<<1|reverse: quality by quality
// from the highest, the reverse-strand reads of a quality before its forward-strand ones.  The sort is replaced by counts:
// count_runs() leaves, in the lane's column of `s_slot`, one dword per distinct quality of the lane in descending order
// (rank r = qualities above it in the lane's mask):  reverse-strand reads | forward-strand reads << 8 | quality << 16.
// NRANK qualities per round (binned base qualities give a handful; a lane with more goes round again).
//   qm      bit q = some read of the source has quality q
//   src(j)  key7 of source element j, or -1 when the element is not of this base / was rejected
// Returns sum of the qualities of the reads counted (QS).
#ifndef WR
#define WR 2            // reads of a run taken per step of the errmod walk
#endif
#ifndef GLF_WIDE_KEYS
#define GLF_WIDE_KEYS 0      // the counting pass and pass 1 read four keys as one (unaligned) 8-byte LDS read instead of four 2-byte ones
#endif
// the sources of count_runs(): key7 of the j-th element or -1 -- the reads of the primary base (rejected reads are zeros), or those
// of the lane's other reads that show base b
struct PrimSrc {
    const uint16_t *kpp;
    __device__ __forceinline__ int operator()(int j) const { const uint32_t k = kpp[j]; return k ? (int)(k & 0x7f) : -1; }
    __device__ __forceinline__ void quad(int j, int n, int (&k4)[4]) const
    {
        uint64_t w; __builtin_memcpy(&w, kpp + j, 8);
        #pragma unroll
        for (int u = 0; u < 4; ++u) { const uint32_t k = (uint32_t)(w >> (16 * u)) & 0xffffu; k4[u] = (j + u < n && k) ? (int)(k & 0x7f) : -1; }
    }
};
struct BaseSrc {
    const uint16_t *kp; int b;
    __device__ __forceinline__ int operator()(int i) const { const uint32_t k = kp[i]; return (int)KEY_B(k) == b ? (int)(k & 0x7f) : -1; }
    __device__ __forceinline__ void quad(int j, int n, int (&k4)[4]) const
    {
        uint64_t w; __builtin_memcpy(&w, kp + j, 8);
        #pragma unroll
        for (int u = 0; u < 4; ++u) { const uint32_t k = (uint32_t)(w >> (16 * u)) & 0xffffu; k4[u] = (j + u < n && (int)KEY_B(k) == b) ? (int)(k & 0x7f) : -1; }
    }
};
template <bool FIRST, class Src>
__device__ __forceinline__ uint32_t count_runs(uint32_t *s_slot, uint64_t qm, int tid, Src src, int nsrc)
{
    #pragma unroll
    for (int k = 0; k < NRANK; ++k) s_slot[k * WG + tid] = 0;
    const uint64_t qm1 = qm >> 1;                                   // rank of q = qualities above it = popcount(qm >> (q + 1))
    // FU source elements per trip: their reads are in flight before the first count is added
    for (int j = 0; __any(j < nsrc); j += FU) {
        int k4[FU];
#if GLF_WIDE_KEYS
        static_assert(FU == 4, "the four keys of a trip are one 8-byte read");
        src.quad(j, nsrc, k4);
#else
        #pragma unroll
        for (int u = 0; u < FU; ++u) k4[u] = j + u < nsrc ? src(j + u) : -1;
#endif
        #pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int key = k4[u], q = (key >> 1) & 63;
            if (key >= 0 && (FIRST || ((qm >> q) & 1ull))) {       // first round: the mask holds every quality of the source
                const int r = __popcll(qm1 >> q);
                if (r < NRANK) atomicAdd(&s_slot[r * WG + tid], (key & 1) ? 1u : 0x100u);
            }
        }
    }
    // the quality of every rank joins its counts
    uint32_t qs = 0;
    uint64_t m = qm;
    for (int r = 0; r < NRANK && __any(m != 0); ++r) {
        const int q = 63 - __clzll((long long)(m | 1ull));
        const uint32_t c = s_slot[r * WG + tid];
        if (m != 0) { s_slot[r * WG + tid] = c | (uint32_t)q << 16; qs += (uint32_t)q * ((c & 0xffu) + (c >> 8)); }
        m &= ~(1ull << q);
    }
    return qs;
}

// The descending walk of errmod_cal for one base over the runs count_runs() left: the t-th read of the lane adds
//     fk[reads of its strand so far] * beta[q][t][n]
// to the double sum, in the reference's order.  Every lane steps through its own runs, one read per step, in straight-line
// code (selects only: 64 lanes are at 64 different places of their runs); the loop runs for as many steps as the deepest
// cell of the wavefront has reads of the base, and a lane that is through adds fk = +0 times a finite table entry, which
// leaves its (non-negative) sum as it is.  The loads of a step are in flight while the step before is added; they are
// issued whether or not a lane still has a read (no branch around them: the compiler can then count the loads in flight
// and wait for the older one only).
// `brow`: byte offset of beta[0][0][n] (stored q, k, n: tables.cpp; the q = 0 row is all zeros).
template <class Src>
__device__ __forceinline__ double walk_runs(uint32_t *s_slot, uint64_t qm, const double *s_fk, const char *bbase, int tid,
                                            uint32_t brow, Src src, int nsrc, uint32_t &rev_out, uint32_t &qs_out, int ab = 0)
{
    double bs = 0;
    uint32_t qs = 0;
    uint32_t wpack = 0;      // reads walked so far: reverse strand in the low half, forward strand in the high half
    uint32_t cnt = 0;        // reads of the current quality still to walk, the same halves
    uint32_t koff = brow;    // brow + (reads walked so far) << 11: the k row of the next read
    uint32_t qoff = 0;       // current quality << 19
    uint32_t r = 0;          // ranks the lane has taken since the slots were filled
    uint32_t d_nx = 0;       // the dword of rank r, read ahead
    auto slot_of = [&](uint32_t rk) -> uint32_t { return s_slot[min(rk, (uint32_t)NRANK - 1u) * WG + tid]; };
    // One step = up to WR reads of the lane's current (quality, strand) run (binned base qualities make runs of several reads
    // the rule): the run bookkeeping is paid once for all of them.
    auto step = [&](uint32_t (&off)[WR], uint32_t (&wi)[WR]) -> bool {
        // the current quality is used up: the next rank (an empty dword past the lane's last one: the lane stays through)
        const bool pop = cnt == 0 && r < NRANK;
        cnt = pop ? (d_nx & 0xffu) | (d_nx & 0xff00u) << 8 : cnt;
        qoff = pop ? (d_nx & 0x3f0000u) << 3 : qoff;
        r += pop ? 1u : 0u;
        d_nx = slot_of(r);                                                // (the same dword again for a lane that did not pop)
        // reverse strand first
        const uint32_t sh = (cnt & 0xffffu) ? 0u : 16u;
        const uint32_t left = (cnt >> sh) & 0xffffu;                      // reads left in the run (0: the lane is through)
        const uint32_t w = (wpack >> sh) & 0xffffu;
        const uint32_t o1 = qoff + koff;
        #pragma unroll
        for (uint32_t u = 0; u < WR; ++u) {
            const bool act = left > u;
            wi[u] = act ? w + u : 256u;
            off[u] = act ? o1 + (u << 11) : brow;
        }
        const uint32_t took = min(left, (uint32_t)WR);
        wpack += took << sh; cnt -= took << sh;
        koff += took << 11;
        return __any(left != 0);
    };
    // (diagnostics build: `ab` switches the table gather to one line (512), off (1024), the fk read off (2048))
    #define WALK_LOAD(B, F, off_, wi_) do { _Pragma("unroll") for (int u_ = 0; u_ < WR; ++u_) { \
        B[u_] = (ab & 1024) ? 1.0 : *reinterpret_cast<const double*>(bbase + ((ab & 512) ? (off_)[u_] & 0x1f8u : (off_)[u_])); \
        F[u_] = (ab & 2048) ? (double)(wi_)[u_] : s_fk[(wi_)[u_]]; } } while (0)
    #define WALK_ADD(B, F) do { _Pragma("unroll") for (int u_ = 0; u_ < WR; ++u_) bs += F[u_] * B[u_]; } while (0)
    bool first = true;
    while (__any(qm != 0)) {                                              // a round: the next NRANK qualities of every lane
        qs += first ? count_runs<true>(s_slot, qm, tid, src, nsrc) : count_runs<false>(s_slot, qm, tid, src, nsrc);
        first = false; r = 0;
        d_nx = slot_of(0);
        // Two steps in flight, each in its own registers: a trip adds the older one and puts the step after next in its place, so
        // no step's operands are copied and the wait before an addition is for that step's loads alone.  (Once a step finds
        // no lane with a read, every later one adds +0: the two left over at the end are added in any order.)
        double bx[WR], fx[WR], by[WR], fy[WR];
        uint32_t off[WR], wi[WR];
        bool ax = step(off, wi);
        WALK_LOAD(bx, fx, off, wi);
        bool ay = step(off, wi);
        WALK_LOAD(by, fy, off, wi);
        while (ax) {
            WALK_ADD(bx, fx);
            ax = step(off, wi);
            WALK_LOAD(bx, fx, off, wi);
            if (!ay) break;
            WALK_ADD(by, fy);
            ay = step(off, wi);
            WALK_LOAD(by, fy, off, wi);
        }
        WALK_ADD(bx, fx);
        WALK_ADD(by, fy);
        // the qualities of this round leave the mask
        if (__any(__popcll(qm) > NRANK)) {
            uint64_t m = qm;
            for (int k = 0; k < NRANK && m; ++k) m &= ~(1ull << (63 - __clzll((long long)m)));
            qm = m;
        } else qm = 0;
    }
    #undef WALK_ADD
    #undef WALK_LOAD
    rev_out = wpack & 0xffffu; qs_out = qs;
    return bs;
}

// ---- the walk with one read per step and the step number in a scalar register (GLF_WALK == 2) ----
// count_ends(): count_runs() whose rank dwords end up as  quality << 19 | end_fwd << 8 | end_rev : the lane's reads of the round in
// walking order are numbered 0 .. nr - 1; those of rank r are [end_fwd of rank r - 1, end_fwd), its reverse-strand reads first, up
// to end_rev.  walk_ends() then takes step t in every lane at once: the k row of the table is (reads of earlier rounds + t), the
// same for the whole wavefront but for a per-lane constant, the descriptor is used up when t reaches its end_fwd, the read is a
// reverse-strand one while t < end_rev, and the forward strand's running count is t minus the reverse strand's.  A step is then
// two compares and a handful of selects instead of the run bookkeeping of walk_runs() (which lets a lane take up to WR reads
// of its current run and so needs a step count of its own per lane).
template <bool FIRST, class Src>
__device__ __forceinline__ uint32_t count_ends(uint32_t *s_slot, uint64_t qm, int tid, Src src, int nsrc, uint32_t &nr)
{
    #pragma unroll
    for (int k = 0; k < NRANK; ++k) s_slot[k * WG + tid] = 0;
    const uint64_t qm1 = qm >> 1;
    for (int j = 0; __any(j < nsrc); j += FU) {
        int k4[FU];
#if GLF_WIDE_KEYS
        static_assert(FU == 4, "the four keys of a trip are one 8-byte read");
        src.quad(j, nsrc, k4);
#else
        #pragma unroll
        for (int u = 0; u < FU; ++u) k4[u] = j + u < nsrc ? src(j + u) : -1;
#endif
        #pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int key = k4[u], q = (key >> 1) & 63;
            if (key >= 0 && (FIRST || ((qm >> q) & 1ull))) {
                const int r = __popcll(qm1 >> q);
                if (r < NRANK) atomicAdd(&s_slot[r * WG + tid], (key & 1) ? 1u : 0x100u);
            }
        }
    }
    uint32_t qs = 0, acc = 0;
    uint64_t m = qm;
    for (int r = 0; r < NRANK && __any(m != 0); ++r) {
        const int q = 63 - __clzll((long long)(m | 1ull));
        const uint32_t c = s_slot[r * WG + tid];
        if (m != 0) {
            const uint32_t er = acc + (c & 0xffu);
            acc = er + (c >> 8);
            qs += (uint32_t)q * (acc - (er - (c & 0xffu)));
            s_slot[r * WG + tid] = (uint32_t)q << 19 | acc << 8 | er;
        }
        m &= ~(1ull << q);
    }
    nr = acc;
    return qs;
}
#ifndef GLF_PD
#define GLF_PD 2            // steps of the walk in flight
#endif
template <class Src>
__device__ __forceinline__ double walk_ends(uint32_t *s_slot, uint64_t qm, const double *s_fk, const char *bbase, int tid,
                                            uint32_t brow, Src src, int nsrc, uint32_t &rev_out, uint32_t &qs_out, int ab = 0)
{
    double bs = 0;
    uint32_t qs = 0;
    uint32_t wr = 0;          // reverse-strand reads walked so far
    uint32_t tl = 0;          // reads walked in earlier rounds
    uint32_t vrow = brow;     // brow + tl << 11: the lane's part of the k row
    bool first = true;
    while (__any(qm != 0)) {
        uint32_t nr;
        qs += first ? count_ends<true>(s_slot, qm, tid, src, nsrc, nr) : count_ends<false>(s_slot, qm, tid, src, nsrc, nr);
        first = false;
        // the deepest lane's reads: the steps of the round
        uint32_t T = nr;
        #pragma unroll
        for (int o = 32; o > 0; o >>= 1) T = max(T, (uint32_t)__shfl_xor((int)T, o));
        T = (uint32_t)__builtin_amdgcn_readfirstlane((int)T);
        uint32_t d = s_slot[tid], ridx = 1, d_nx = s_slot[WG + tid];
        const uint32_t tb = tl;
        auto step = [&](const uint32_t t, uint32_t &voff, uint32_t &wi) {
            const bool pop = t == ((d >> 8) & 0xffu);                  // the rank's reads are walked: the next one's descriptor
            d = pop ? d_nx : d;
            ridx += pop ? 1u : 0u;
            d_nx = s_slot[min(ridx, (uint32_t)NRANK - 1u) * WG + tid];
            const bool act = t < nr, rev = t < (d & 0xffu);
            const uint32_t w = rev ? wr : t + tb - wr;
            wr += (act && rev) ? 1u : 0u;
            wi = act ? w : 256u;
            voff = act ? (d & 0x1f80000u) + vrow : brow;
        };
        #define WALK2_LOAD(B, F, t_, voff_, wi_) do { \
            B = (ab & 1024) ? 1.0 : *reinterpret_cast<const double*>(bbase + ((size_t)(t_) << 11) + (voff_)); \
            F = (ab & 2048) ? (double)(wi_) : s_fk[(wi_)]; } while (0)
        double bx[GLF_PD], fx[GLF_PD];
        #pragma unroll
        for (int u = 0; u < GLF_PD; ++u) { uint32_t voff, wi; step((uint32_t)u, voff, wi); WALK2_LOAD(bx[u], fx[u], (uint32_t)u, voff, wi); }
        for (uint32_t t = GLF_PD; t < T + GLF_PD; t += GLF_PD) {        // (steps past every lane's reads add +0 times a finite entry of the q = 0 rows)
            #pragma unroll
            for (int u = 0; u < GLF_PD; ++u) {
                bs += fx[u] * bx[u];
                uint32_t voff, wi;
                step(t + (uint32_t)u, voff, wi);
                WALK2_LOAD(bx[u], fx[u], t + (uint32_t)u, voff, wi);
            }
        }
        #pragma unroll
        for (int u = 0; u < GLF_PD; ++u) bs += fx[u] * bx[u];
        #undef WALK2_LOAD
        tl += nr; vrow += nr << 11;
        if (__any(__popcll(qm) > NRANK)) {
            uint64_t m = qm;
            for (int k = 0; k < NRANK && m; ++k) m &= ~(1ull << (63 - __clzll((long long)m)));
            qm = m;
        } else qm = 0;
    }
    rev_out = wr; qs_out = qs;
    return bs;
}
#ifndef GLF_FETCH_GLOBAL
#define GLF_FETCH_GLOBAL 0
#endif
#ifndef GLF_WALK
#define GLF_WALK 1
#endif
#if GLF_WALK == 2
#define walk_runs walk_ends
#endif

// per-lane partial sums of phase A (one site segment of one staging round)
struct ReadSums {
    uint32_t t_bq, t_bq2, t_mq, t_mq2, t_md, t_md2;    // all accepted reads: baseQ, mapQ, min_dist and their squares
    uint32_t d_bq, d_bq2, d_mq, d_mq2, d_md, d_md2;    // the "diff" reads among them (is_diff of bam2bcf.c:186,195)
    __device__ __forceinline__ void clear() { t_bq = t_bq2 = t_mq = t_mq2 = t_md = t_md2 = d_bq = d_bq2 = d_mq = d_mq2 = d_md = d_md2 = 0; }
};
typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
// wave-uniform counts of phase A (ballots: scalar registers)
struct WaveCounts {
    uint32_t ori, mq0, ref59, alt59, fwd59, rev59;
    __device__ __forceinline__ void clear() { ori = mq0 = ref59 = alt59 = fwd59 = rev59 = 0; }
};

// DEEP = false: the tile, 256 consecutive cells per workgroup.  A cell with more pileup entries than the LDS key window holds is
// not worked on here: it is listed (P.deep_*), and the launch that follows (DEEP = true) gives each listed cell a workgroup
// of its own -- phase A over the cell's reads with all 256 lanes, the keys in a global scratch array instead of LDS, phase B
// by the one lane that owns the cell.
// PHASE = 0: both phases in one kernel (the keys never leave LDS).  PHASE = 1 / 2: the two phases as kernels of their own --
// phase A streams the reads once and leaves the keys in HBM (P.keys, 2 bytes per read) with nothing but the site
// histograms in LDS, at the occupancy its registers allow; phase B copies its span's keys into LDS with coalesced loads
// and keeps keys + fk + run counters there (no histograms: the rare take-back of a truncated cell goes to global memory).
#ifndef GLF_WAVES_A
#define GLF_WAVES_A 6
#endif
template <bool INDEL, bool LDS_HIST, bool DEEP, int PHASE>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(PHASE == 1 ? GLF_WAVES_A : GLF_WAVES, PHASE == 1 ? GLF_WAVES_A : GLF_WAVES))) void glfgen_kernel(const GlfgenParams P)
{
    extern __shared__ __align__(16) unsigned char smem[];
    if (DEEP && (blockIdx.x >= P.deep_ctr[0] || P.deep_list[2 * blockIdx.x] == 0xffffffffu)) return;
    constexpr bool KEYS_GLOBAL = PHASE == 1 || (PHASE == 2 && DEEP);       // the key of read i at P.keys[i]
    const int cap = (DEEP || PHASE == 1) ? 0x7ffffff0 : P.lds_cap;
    double   *s_fk  = reinterpret_cast<double*>(smem + LDS_FK);
    uint32_t *s_cnt = reinterpret_cast<uint32_t*>(smem + LDS_CNT);
    int      *s_hist = reinterpret_cast<int*>(smem + LDS_HIST_OFF);
    unsigned long long *s_tot = reinterpret_cast<unsigned long long*>(s_hist + (size_t)P.hist_slots * HP_SIZE);   // [slots][SITE_NSUM]
    // phase A's per-lane partial sums [slots][NPART][pcol] share the slot counters' LDS: they are added up and cleared
    // before phase B takes the region
    uint32_t *s_part = s_cnt;
    const int pcol = P.part_cols;                    // columns per value: a power of two, NPART * slots * pcol <= 2048
    uint16_t *s_key = KEYS_GLOBAL ? P.keys : DEEP ? P.deep_keys + P.deep_list[2 * blockIdx.x + 1]
                                                  : reinterpret_cast<uint16_t*>(s_tot + (size_t)P.hist_slots * SITE_NSUM);
    __shared__ unsigned int s_next, s_skip;

    const int tid = threadIdx.x;
    GLF_STAMP_DECL
    const int S = P.n_smpl;
    const long ncells = (long)P.n_sites * S;
    const long cell0 = DEEP ? (long)P.deep_list[2 * blockIdx.x] : (long)blockIdx.x * WG;
    const long cell_end = DEEP ? cell0 + 1 : min(cell0 + WG, ncells);
    const int site0 = (int)(cell0 / S);
    const int site_last = (int)((cell_end - 1) / S);

    const long cell = cell0 + tid;
    const bool active = cell < cell_end;

    if (PHASE != 1) {
        s_fk[tid] = P.fk[tid];
        if (tid < 8) s_fk[256 + tid] = 0.0;           // [256]: the factor of a lane that sits a chunk element out
    }
    if (LDS_HIST) {
        for (int i = tid; i < P.hist_slots * HP_SIZE; i += WG) s_hist[i] = 0;
        for (int i = tid; i < P.hist_slots * SITE_NSUM; i += WG) s_tot[i] = 0;
    }

    int site = 0, ref4c = 4;
    uint32_t beg = 0, end = 0;
    if (active) {
        site = (int)(cell / S);
        beg = P.off[cell]; end = P.off[cell + 1];
        if (!INDEL) ref4c = nt16_int(P.ref16[site]);
    }
    const int primary = INDEL ? 0 : ref4c;            // the base most reads of the cell show
    uint32_t fmt_flag = (uint32_t)P.fmt_flag;
    asm volatile("" : "+s"(fmt_flag));
    const bool want_epos = (fmt_flag & (BCFGPU_INFO_RPB | BCFGPU_INFO_VDB)) != 0;
    const bool want_scr = (fmt_flag & (BCFGPU_INFO_SCR | BCFGPU_FMT_SCR)) != 0;
    const uint32_t span_end = P.off[cell_end];
    // The scalar arguments the inner loops use, as values of their own: read straight from the argument block they are
    // elements of one eight-register tuple, and when the scalar registers run short the whole tuple is spilled and brought
    // back, all eight, at every use inside phase A's loop.
    uint32_t n_reads_tot = P.n_reads, min_baseQ = (uint32_t)P.min_baseQ, capQ = (uint32_t)P.capQ;
    asm volatile("" : "+s"(n_reads_tot), "+s"(min_baseQ), "+s"(capQ));
    const uint32_t *p_rd = P.rd, *p_aux = P.aux, *p_off = P.off;          // (the same for the pointers of the inner loops)
    const uint8_t *p_epos = P.epos;
    asm volatile("" : "+s"(p_rd), "+s"(p_aux), "+s"(p_off), "+s"(p_epos));

    bool done = !active;
    // A cell must fit one staging round whatever its alignment (the window starts at a multiple of 4 reads).  One that does
    // not is listed for the launch that follows, with room for its keys in the scratch array; the rounds here step over its
    // reads.  Only when the list or the scratch array is full is the tile refused.
    bool deep = false;
    if (!DEEP && PHASE != 1 && active && end - beg > (uint32_t)cap - 3u) {
        deep = true;
        const uint32_t need = (end - beg + 16u) & ~7u;           // keys of the cell, the slack of the 8-byte stores, a multiple of 8
        const uint32_t slot = atomicAdd(&P.deep_ctr[0], 1u), at = atomicAdd(&P.deep_ctr[1], need);
        const bool room = at + need <= P.deep_key_cap;
        if (slot < P.deep_cap) { P.deep_list[2 * slot] = room ? (uint32_t)cell : 0xffffffffu; P.deep_list[2 * slot + 1] = room ? at : 0u; }
        if (slot >= P.deep_cap || !room) { atomicExch(P.err, BCFGPU_E_DEPTH); atomicExch(&P.deep_ctr[2], 1u); }
    }
    uint32_t base = p_off[cell0];

    for (;;) {
        const uint32_t abase = KEYS_GLOBAL ? 0u : base & ~3u;    // key index 0 of this round
        const uint32_t lim = (DEEP || PHASE == 1) ? span_end : min(abase + (uint32_t)cap, span_end);
        if (tid == 0) { s_next = 0xffffffffu; s_skip = 0; }
        if (LDS_HIST) for (int i = tid; i < P.hist_slots * NPART * pcol; i += WG) s_part[i] = 0;
        __syncthreads();
        const bool cand = !done && !deep && beg >= base && end <= lim;
        if (!done && !cand) atomicMin(&s_next, beg);             // the first cell left for a later round (deep tiles only): one that
        __syncthreads();                                         // does not fit the window any more, or a listed cell
        const uint32_t nb = s_next;
        const uint32_t rlim = min(nb, lim);                      // reads [base, rlim) belong to this round's cells
        const bool part = cand && end <= rlim;                   // this lane's cell is handled in this round (cells behind a listed cell wait)

        GLF_STAMP(0)
        if constexpr (PHASE == 2) {
            // the keys phase A left in HBM: this round's window into LDS, 16 bytes per lane and trip (a listed cell reads its own
            // from HBM: KEYS_GLOBAL)
            if (!KEYS_GLOBAL) {
                const uint32_t nk = (rlim > abase ? rlim - abase : 0u) + 8u;        // (one key past the last cell: phase B reads ahead)
                for (uint32_t i = 8u * tid; i < nk; i += 8u * WG)
                    *reinterpret_cast<uint4*>(s_key + i) = *reinterpret_cast<const uint4*>(P.keys + abase + i);
            }
        } else
        // ================= phase A: one lane per read =================
        for (int sg = site0; sg <= site_last; ++sg) {            // uniform: the site segments of the workgroup's span
            const long c_lo = max(cell0, (long)sg * S), c_hi = min(cell_end, (long)(sg + 1) * S);
            const uint32_t rb = max(p_off[c_lo], base), re = min(p_off[c_hi], rlim);
            if (rb >= re) continue;
            const int ref_base = INDEL ? -1 : (int)P.ref16[sg];
            const uint32_t ref4 = INDEL ? 4u : (uint32_t)nt16_int(ref_base);
            const uint32_t primq = INDEL ? 0u : ref4;            // `primary` of the segment's cells
            // nt16 code -> base 0..4 with code 0 ('=') standing for the reference base (bam2bcf.c:189-190)
            const unsigned long long tbl = (NT16_INT_TBL & ~0xfull) | (unsigned long long)ref4;
            int *hist = LDS_HIST ? s_hist + (sg - site0) * HP_SIZE : P.hist + (long)sg * H_SIZE;
            ReadSums A; A.clear();
            WaveCounts C; C.clear();

            // Four consecutive reads of every lane.  Called in wave-uniform control flow only (the ballots count whole waves
            // into scalar registers).  CHECK: reads outside [rb, re) are masked (`vm`, one bit per read).
            // The I16 sums take the reads in pairs: the fields of two reads side by side as u16, one v_dot2_u32_u16 per sum
            // (a rejected read is a zero record and contributes zeros).
            auto quad = [&](const uint32_t (&wv)[4], const uint32_t (&av)[4], uint32_t e4, uint32_t vm, uint32_t (&kv)[4]) {
                uint32_t bqz[4], mqz[4], mdz[4];
                bool dif[4];
                #pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t w = wv[u];
                    const bool valid = (vm >> u) & 1;
                    uint32_t q, b, bq, mapQ, md;
                    bool ok, seen;
                    const uint32_t nt = (w >> 16) & 15;
                    if (INDEL) {
                        const uint32_t ax = av[u];
                        b = (ax >> 16) & 0xf;                     // 0..4 after bcf_call_gap_prep (bam2bcf_indel.c:449-456)
                        bq = q = ax & 0xff;
                        if (q < min_baseQ) { b = 0; q = w & 0xff; }
                        b = min(b, 4u);
                        q = min(q, (ax >> 8) & 0xff);             // seqQ
                        seen = valid && !(w & BCFGPU_RD_SKIP);
                        ok = seen;
                        mapQ = (w >> 8) & 0xff; md = w >> 24;
                        if (!ok) { bq = 0; mapQ = 0; md = 0; }
                    } else {
                        seen = valid && !(w & (BCFGPU_RD_SKIP | BCFGPU_RD_DEL));
                        ok = seen && (w & 0xff) >= min_baseQ;
                        const uint32_t wz = ok ? w : 0u;          // a rejected read: a zero record
                        bq = q = wz & 0xff;                       // seqQ = 99 never binds: q is capped by capQ <= 63 below
                        mapQ = (wz >> 8) & 0xff; md = wz >> 24;
                        b = (uint32_t)((tbl >> (4 * nt)) & 7);
                    }
                    const unsigned long long b_ok = __ballot(ok);
                    C.ori += (uint32_t)__popcll(__ballot(seen));
                    if (mapQ == 255) mapQ = DEF_MAPQ;
                    C.mq0 += (uint32_t)__popcll(__ballot(mapQ == 0) & b_ok);
                    mapQ = min(mapQ, capQ);
                    q = max(min(min(q, mapQ), 63u), 4u);
                    md = min(md, (uint32_t)CAP_DIST);
                    const uint32_t rev = (w >> 20) & 1;
                    uint32_t key = KEY_PACK(rev, q, b, 0) | (b == primq ? KEY_PRIM : 0u);
                    if (want_scr) key |= ((w >> 21) & 1) << 10;
                    kv[u] = ok ? key : 0u;
                    bqz[u] = bq; mqz[u] = mapQ; mdz[u] = md;
                    dif[u] = ok && (INDEL ? b != 0 : !(ref4 < 4 && b == ref4));
                    // bias-test histograms: ibq = (int)(baseQ/60.*60) is the identity on 0..59 (checked in tests).
                    // The mapQ >= 59 bins of the four mapQ histograms (most reads) are counted by ballot; every ALT array sits
                    // H_ALT_OFF after its REF array.
                    const bool isref = (int)nt == ref_base;
                    const bool m59 = mapQ >= 59;
                    {
                        const unsigned long long b59 = __ballot(m59) & b_ok, bref = __ballot(isref), brev = __ballot(rev != 0);
                        C.ref59 += (uint32_t)__popcll(b59 & bref); C.alt59 += (uint32_t)__popcll(b59 & ~bref);
                        C.rev59 += (uint32_t)__popcll(b59 & brev); C.fwd59 += (uint32_t)__popcll(b59 & ~brev);
                    }
#ifdef GLF_AGG_BQ
                    if (LDS_HIST) {
                        // base-quality bins: binned qualities put the 64 reads of a wave instruction on a handful of counters; one
                        // add per distinct value present, with the REF / ALT counts of its lanes (ballots), instead of 64 adds on
                        // <= 8 addresses
                        const uint32_t bin = min(bq, 59u);
                        const unsigned long long bref2 = __ballot(isref);
                        unsigned long long todo = b_ok;
                        while (todo) {
                            const int ldr = __builtin_ctzll(todo);
                            const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)bin, ldr);
                            const unsigned long long m = __ballot(bin == v) & todo;
                            if ((int)(tid & 63) == ldr) atomicAdd(&hist[H_REF_BQ + v], (int)((uint32_t)__popcll(m & bref2) | (uint32_t)__popcll(m & ~bref2) << 16));
                            todo &= ~m;
                        }
                    }
#endif
                    if (ok && !BCFGPU_ABL(P, 1)) {
                        if (LDS_HIST) {
                            const int inc = isref ? 1 : 0x10000;
                            atomicAdd(&hist[H_REF_POS + ((e4 >> (8 * u)) & 0xff)], inc);
#ifndef GLF_AGG_BQ
                            atomicAdd(&hist[H_REF_BQ + min(bq, 59u)], inc);
#endif
                            if (!m59) {
                                atomicAdd(&hist[H_REF_MQ + mapQ], inc);
                                atomicAdd(&hist[HP_MQS + mapQ], rev ? 0x10000 : 1);
                            }
                        } else {
                            const uint32_t aoff = isref ? 0u : (uint32_t)H_ALT_OFF;
                            atomicAdd(&hist[aoff + H_REF_POS + ((e4 >> (8 * u)) & 0xff)], 1);
                            atomicAdd(&hist[aoff + H_REF_BQ + min(bq, 59u)], 1);
                            if (!m59) {
                                atomicAdd(&hist[aoff + H_REF_MQ + mapQ], 1);
                                atomicAdd(&hist[(rev ? H_REV_MQS : H_FWD_MQS) + mapQ], 1);
                            }
                        }
                    }
#ifdef GLF_SCHED
                    if ((u & (GLF_SCHED - 1)) == GLF_SCHED - 1) __builtin_amdgcn_sched_barrier(0);   // the lane masks of these reads end here
#endif
                }
                #pragma unroll
                for (int h = 0; h < 4; h += 2) {
                    const v2u16 one2 = { 1, 1 };
                    const v2u16 bq2 = __builtin_bit_cast(v2u16, bqz[h] | bqz[h + 1] << 16);
                    const v2u16 mq2 = __builtin_bit_cast(v2u16, mqz[h] | mqz[h + 1] << 16);
                    const v2u16 md2 = __builtin_bit_cast(v2u16, mdz[h] | mdz[h + 1] << 16);
                    A.t_bq = __builtin_amdgcn_udot2(bq2, one2, A.t_bq, false); A.t_bq2 = __builtin_amdgcn_udot2(bq2, bq2, A.t_bq2, false);
                    A.t_mq = __builtin_amdgcn_udot2(mq2, one2, A.t_mq, false); A.t_mq2 = __builtin_amdgcn_udot2(mq2, mq2, A.t_mq2, false);
                    A.t_md = __builtin_amdgcn_udot2(md2, one2, A.t_md, false); A.t_md2 = __builtin_amdgcn_udot2(md2, md2, A.t_md2, false);
                    if (__any(dif[h] || dif[h + 1])) {           // rare: sequencing errors and the ALT reads of variant sites
                        const uint32_t dm = (dif[h] ? 0xffffu : 0u) | (dif[h + 1] ? 0xffff0000u : 0u);
                        const v2u16 bd = __builtin_bit_cast(v2u16, __builtin_bit_cast(uint32_t, bq2) & dm);
                        const v2u16 qd = __builtin_bit_cast(v2u16, __builtin_bit_cast(uint32_t, mq2) & dm);
                        const v2u16 dd = __builtin_bit_cast(v2u16, __builtin_bit_cast(uint32_t, md2) & dm);
                        A.d_bq = __builtin_amdgcn_udot2(bd, one2, A.d_bq, false); A.d_bq2 = __builtin_amdgcn_udot2(bd, bd, A.d_bq2, false);
                        A.d_mq = __builtin_amdgcn_udot2(qd, one2, A.d_mq, false); A.d_mq2 = __builtin_amdgcn_udot2(qd, qd, A.d_mq2, false);
                        A.d_md = __builtin_amdgcn_udot2(dd, one2, A.d_md, false); A.d_md2 = __builtin_amdgcn_udot2(dd, dd, A.d_md2, false);
                    }
                }
            };

            const uint32_t g0 = rb & ~3u;
            uint4 w4n = make_uint4(0, 0, 0, 0), a4n = make_uint4(0, 0, 0, 0);
            uint32_t e4n = 0;
            // (the pointers went through scalar registers as opaque values: said to be global memory again, the loads are global_load
            // with a scalar base, not flat_load -- which also counts on the LDS counter and needs a 64-bit address per lane)
#if GLF_FETCH_GLOBAL
            typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
            typedef const __attribute__((address_space(1))) u32x4_t *g_u4;
            typedef const __attribute__((address_space(1))) uint32_t *g_u32;
#endif
            auto fetch = [&](uint32_t i4) {
                if (i4 >= re) return;
#if GLF_FETCH_GLOBAL
                if (__all(i4 >= re || i4 + 3 < n_reads_tot)) {   // (wave-uniform: but for the wavefront that holds the tile's last reads)
                    const u32x4_t wq = *(g_u4)(const void*)(p_rd + i4);
                    w4n = make_uint4(wq.x, wq.y, wq.z, wq.w);
                    if (want_epos) e4n = *(g_u32)(const void*)(p_epos + i4);
                    if (INDEL) { const u32x4_t aq = *(g_u4)(const void*)(p_aux + i4); a4n = make_uint4(aq.x, aq.y, aq.z, aq.w); }
                } else
#endif
                if (i4 + 3 < n_reads_tot) {
                    w4n = *reinterpret_cast<const uint4*>(p_rd + i4);
                    if (want_epos) e4n = *reinterpret_cast<const uint32_t*>(p_epos + i4);
                    if (INDEL) a4n = *reinterpret_cast<const uint4*>(p_aux + i4);
                } else {                                         // the last reads of the tile
                    uint32_t t4[4] = {0, 0, 0, 0}, x4[4] = {0, 0, 0, 0};
                    e4n = 0;
                    for (int j = 0; j < 4; ++j) if (i4 + j < n_reads_tot) {
                        t4[j] = p_rd[i4 + j];
                        if (want_epos) e4n |= (uint32_t)p_epos[i4 + j] << (8 * j);
                        if (INDEL) x4[j] = p_aux[i4 + j];
                    }
                    w4n = make_uint4(t4[0], t4[1], t4[2], t4[3]); a4n = make_uint4(x4[0], x4[1], x4[2], x4[3]);
                }
            };
            fetch(g0 + 4u * tid);
            const uint32_t ntrip = (re - g0 + 4u * WG - 1) / (4u * WG);       // uniform trip count: the ballots need whole waves
            for (uint32_t it = 0; it < (BCFGPU_ABL(P, 32) ? 0u : ntrip); ++it) {
                const uint32_t i4 = g0 + 4u * tid + it * (4u * WG);
                if (__all(i4 >= re)) break;                      // the wavefront is past the segment's last read (the last trip's upper waves)
                const uint32_t wv[4] = { w4n.x, w4n.y, w4n.z, w4n.w };
                const uint32_t av[4] = { a4n.x, a4n.y, a4n.z, a4n.w };
                const uint32_t e4 = e4n;
                fetch(i4 + 4u * WG);                             // the next trip's loads fly while this one is worked on
                const uint32_t ko = i4 - abase;
                if (__all(i4 >= rb && i4 + 3 < re)) {            // the whole wave inside the segment: no range checks, 8-byte LDS stores
                    uint32_t kv[4];
                    quad(wv, av, e4, 15u, kv);
                    *reinterpret_cast<uint2*>(s_key + ko) = make_uint2(kv[0] | kv[1] << 16, kv[2] | kv[3] << 16);
                } else {                                         // a wave at a ragged end of the segment
                    uint32_t kv[4], vm = 0;
                    #pragma unroll
                    for (int u = 0; u < 4; ++u) vm |= (i4 + u >= rb && i4 + u < re) ? 1u << u : 0u;
                    quad(wv, av, e4, vm, kv);
                    #pragma unroll
                    for (int u = 0; u < 4; ++u) if ((vm >> u) & 1) s_key[ko + u] = (uint16_t)kv[u];
                }
            }
            // ---- the segment's site totals ----
            // LDS mode: every lane adds its partial sums to its own column of the slot's [value][64] table (no conflicts
            // inside a wave); the 64 columns are added up once, when the workgroup is through (below).
            const uint32_t v[NPART] = { A.t_bq - A.d_bq, A.t_bq2 - A.d_bq2, A.d_bq, A.d_bq2, A.t_mq - A.d_mq, A.t_mq2 - A.d_mq2, A.d_mq, A.d_mq2,
                                        A.t_md - A.d_md, A.t_md2 - A.d_md2, A.d_md, A.d_md2 };
            unsigned long long *tot = LDS_HIST ? s_tot + (sg - site0) * SITE_NSUM : P.site_sums + (size_t)sg * SITE_NSUM;
            if (LDS_HIST) {
                uint32_t *pt = s_part + (sg - site0) * (NPART * pcol) + (tid & (pcol - 1));
                #pragma unroll
                for (int j = 0; j < NPART; ++j) atomicAdd(&pt[j * pcol], v[j]);
            } else {
                // global mode (many sites per workgroup, i.e. very few samples): wave sums, one atomic per wave and value
                #pragma unroll
                for (int j = 0; j < NPART; ++j) {
                    const uint32_t x = wave_sum_u32(v[j]);
                    if ((tid & 63) == 0 && x) atomicAdd(&tot[j], (unsigned long long)x);
                }
            }
            if ((tid & 63) == 0) {                               // the wave-uniform counts
                if (C.ori) atomicAdd(&tot[12], (unsigned long long)C.ori);
                if (C.mq0) atomicAdd(&tot[13], (unsigned long long)C.mq0);
                if (LDS_HIST) {
                    if (C.ref59 | C.alt59) atomicAdd(&hist[H_REF_MQ + 59], (int)(C.ref59 | C.alt59 << 16));
                    if (C.fwd59 | C.rev59) atomicAdd(&hist[HP_MQS + 59], (int)(C.fwd59 | C.rev59 << 16));
                } else {
                    if (C.ref59) atomicAdd(&hist[H_REF_MQ + 59], (int)C.ref59);
                    if (C.alt59) atomicAdd(&hist[H_ALT_MQ + 59], (int)C.alt59);
                    if (C.fwd59) atomicAdd(&hist[H_FWD_MQS + 59], (int)C.fwd59);
                    if (C.rev59) atomicAdd(&hist[H_REV_MQS + 59], (int)C.rev59);
                }
            }
        }
        GLF_STAMP(1)
        if (!BCFGPU_ABL(P, 16384)) __syncthreads();          // (diagnostics: 16384 times the kernel without the barriers between the phases; results are then wrong)
        GLF_STAMP(2)
        if (LDS_HIST) {
            // the columns of every partial sum: four lanes per value
            const int nslot = min(P.hist_slots, P.n_sites - site0), q4 = pcol >> 2;
            for (int i = tid; i < nslot * NPART * 4; i += WG) {
                const uint32_t *pt = s_part + (i >> 2) * pcol + (i & 3) * q4;
                unsigned long long x = 0;
                for (int k = 0; k < q4; ++k) x += pt[k];
                x += __shfl_xor(x, 1); x += __shfl_xor(x, 2);
                if ((i & 3) == 0 && x) { const int vi = i >> 2; s_tot[(vi / NPART) * SITE_NSUM + vi % NPART] += x; }
            }
            if (!BCFGPU_ABL(P, 16384)) __syncthreads();
        }

        if constexpr (PHASE == 1) break;                         // the keys are in HBM: phase B is the next launch

        // ================= phase B: one lane per cell =================
        uint16_t *kp_w = s_key + (part ? beg - abase : 0);       // the lane's keys
        const uint16_t *kp = kp_w;
        const int cnt_raw = part ? (int)(end - beg) : 0;
        // A cell with more than 255 usable reads.  bcf_call_glfgen counts every read (QS, ADF/ADR, anno[], SCR, the site's I16
        // sums and histograms: bam2bcf.c:203-252, all done by phase A or below over all keys); only errmod_cal cuts its input to
        // 255 (bam2bcf.c:256; htslib errmod.c draws them with hts_drand48, one generator for the whole process).
        // Here the cell's counts over ALL reads go to a WideRec (kernels.h), and the keys errmod_cal would not take are cleared
        // so that everything below -- which only feeds the likelihoods of such a cell -- sees 255 reads: the ones marked by
        // bcfgpu_errmod_plan (draw.hip: the draw replayed in mpileup's visit order), else the first 255, which is counted in
        // P.trunc (bcfgpu_truncated_cells: cells whose PLs may deviate from a reference run).
        if (__any(cnt_raw > BCFGPU_MAX_DEPTH)) {
            if (cnt_raw > BCFGPU_MAX_DEPTH) {
                uint32_t nus = 0;
                for (int i = 0; i < cnt_raw; ++i) nus += kp_w[i] != 0 ? 1u : 0u;
                const volatile unsigned long long *t = reinterpret_cast<const volatile unsigned long long*>(P.crp);
                uint64_t *qs64_p = reinterpret_cast<uint64_t*>(t[2]); uint32_t *misc_p = reinterpret_cast<uint32_t*>(t[6]);
                uint32_t mark = 0;
                if (nus > BCFGPU_MAX_DEPTH) {
                    uint32_t qs[4] = {0, 0, 0, 0}, af[4] = {0, 0, 0, 0}, ar[4] = {0, 0, 0, 0}, cn[4] = {0, 0, 0, 0}, sc = 0, acc = 0;
                    const bool all_diff = !INDEL && ref4c >= 4;
                    // which 255 feed the likelihoods: the ones bcfgpu_errmod_plan drew for this cell (draw.hip: its bitmap has
                    // exactly 255 of the cell's usable reads marked), else the first 255
                    bool planned = false;
                    if (P.draw_bits) {
                        uint32_t nm = 0;
                        for (int i = 0; i < cnt_raw; ++i) { const uint32_t ri = beg + (uint32_t)i; nm += (kp_w[i] != 0 && ((P.draw_bits[ri >> 5] >> (ri & 31)) & 1u)) ? 1u : 0u; }
                        planned = nm == BCFGPU_MAX_DEPTH;
                    }
                    for (int i = 0; i < cnt_raw; ++i) {
                        const uint32_t k = kp_w[i];
                        if (k == 0) continue;
                        const uint32_t rev = KEY_REV(k), q = KEY_Q(k), b = KEY_B(k);
                        const uint32_t dr = ((all_diff || !(k & KEY_PRIM)) ? 2u : 0u) | rev;     // anno[0<<2 | is_diff<<1 | is_rev]
                        #pragma unroll
                        for (uint32_t x = 0; x < 4; ++x) {
                            qs[x] += b == x ? q : 0u;
                            af[x] += (b == x && !rev) ? 1u : 0u; ar[x] += (b == x && rev) ? 1u : 0u;
                            cn[x] += dr == x ? 1u : 0u;
                        }
                        sc += KEY_SC(k);
                        bool stays = ++acc <= BCFGPU_MAX_DEPTH;
                        if (planned) { const uint32_t ri = beg + (uint32_t)i; stays = ((P.draw_bits[ri >> 5] >> (ri & 31)) & 1u) != 0; }
                        if (!stays) kp_w[i] = 0;
                    }
                    const uint32_t slot = atomicAdd(P.wide_ctr, 1u);
                    const uint32_t big = af[0] | af[1] | af[2] | af[3] | ar[0] | ar[1] | ar[2] | ar[3] | cn[0] | cn[1] | cn[2] | cn[3] | sc;
                    if (slot >= P.wide_cap || big > 0xffffu) atomicExch(P.err, BCFGPU_E_DEPTH);     // (16-bit count planes; the list is sized n_reads / 256)
                    else {
                        WideRec *w = reinterpret_cast<WideRec*>(t[7]) + slot;
                        #pragma unroll
                        for (int x = 0; x < 4; ++x) { w->qs[x] = qs[x]; w->ad[x] = af[x] | ar[x] << 16; }
                        w->cnt[0] = cn[0] | cn[1] << 16; w->cnt[1] = cn[2] | cn[3] << 16;
                        w->scr = sc; w->n = nus; w->cell = (uint32_t)cell; w->pad[0] = w->pad[1] = w->pad[2] = 0;
                        qs64_p[cell] = WIDE_QS_MARK | slot;      // where combine_kernel's frequency pass finds the record
                        mark = CR_WIDE;
                    }
                    if (!planned) atomicAdd(P.trunc, 1u);
                }
                misc_p[cell] = mark;             // read back where the cell's planes are stored (below)
            }
        }
        GLF_STAMP(3)
        // pass 1: the quality mask of the primary base; the few other reads are gathered at the front of the slice
        uint64_t qmask = 0;          // qualities seen among the reads of the primary base
        uint32_t n_prim = 0, scr = 0;
        uint64_t qs64 = 0;           // QS[0..3], 16 bits each
        uint64_t ad64 = 0;           // ADF[0..3] | ADR[0..3]<<32, 8 bits each
        uint32_t n_b4 = 0;           // reads showing neither A, C, G nor T
        uint32_t o_rev = 0, n_other = 0;
        {
#if GLF_WIDE_KEYS
            uint64_t kw = 0, kw_nx;                             // four keys per 8-byte read, the next four requested a group ahead
            __builtin_memcpy(&kw_nx, kp, 8);                    // (up to seven keys past the slice: inside the key array's slack)
#else
            uint32_t k_nx = kp[0];
#endif
            for (int i = 0; i < (BCFGPU_ABL(P, 4) ? 0 : cnt_raw); ++i) {
#if GLF_WIDE_KEYS
                if ((i & 3) == 0) { kw = kw_nx; __builtin_memcpy(&kw_nx, kp + i + 4, 8); } else kw >>= 16;
                const uint32_t k = (uint32_t)kw & 0xffffu;
#else
                const uint32_t k = k_nx;
                k_nx = kp[i + 1];                               // one past the slice stays inside the key array's slack
#endif
                const uint32_t pb = (k >> 11) & 1u;             // KEY_PRIM
                qmask |= (uint64_t)pb << KEY_Q(k);
                n_prim += pb;
                if (want_scr) scr += KEY_SC(k);
                if (k != 0 && !pb) {                            // rare
                    // swapped to position n_other <= i (behind the reader): the walks only count reads per (base, quality,
                    // strand), so the order inside a cell is free
                    const uint32_t t = kp_w[n_other], rev = KEY_REV(k), q = KEY_Q(k), b = KEY_B(k);
                    kp_w[n_other] = (uint16_t)k; kp_w[i] = (uint16_t)t;
                    ++n_other; o_rev += rev;
                    if (b < 4) { qs64 += (uint64_t)q << (16 * b); ad64 += 1ull << (8 * b + 32 * rev); }
                    else ++n_b4;
                }
            }
        }
        const uint32_t n = n_prim + n_other;                 // <= 255
        const bool dead_cell = n == 0;                       // nothing to walk (also the refused cells)
        const char *bbase = reinterpret_cast<const char*>(P.beta);
        const uint32_t brow = n << 3;

        GLF_STAMP(4)
        // ---- errmod_cal: descending walk per base ----
        double bsum[5] = {0, 0, 0, 0, 0};
        uint32_t prim_rev = 0, qs_prim = 0;
        // (a) the primary base
        if (!BCFGPU_ABL(P, 2)) {
            const uint16_t *kpp = kp + n_other;               // primary-base keys and zeros (rejected reads)
            const PrimSrc psrc{kpp};
            const double bs = walk_runs(s_cnt, qmask, s_fk, bbase, tid, brow, psrc, dead_cell ? 0 : cnt_raw - (int)n_other, prim_rev, qs_prim,
                                        BCFGPU_ABL_MASK(P));
            #pragma unroll
            for (int b = 0; b < 5; ++b) if (b == primary) bsum[b] = bs;
        }
        if (primary < 4) {
            qs64 += (uint64_t)qs_prim << (16 * primary);
            ad64 += (uint64_t)(n_prim - prim_rev) << (8 * primary) | (uint64_t)prim_rev << (8 * primary + 32);
        } else n_b4 += n_prim;
        // per-base counts c[0..4] (errmod_cal's aux.c)
        int c[5];
        #pragma unroll
        for (int b = 0; b < 4; ++b) c[b] = (int)((ad64 >> (8 * b)) & 0xff) + (int)((ad64 >> (8 * b + 32)) & 0xff);
        c[4] = (int)n_b4;
        GLF_STAMP(5)
        // (b) the other bases present in the wave
        if (!BCFGPU_ABL(P, 2) && !BCFGPU_ABL(P, 32768) && __any(n_other > 0)) {    // (diagnostics: 32768 times the kernel without the other bases' walks)
            #pragma unroll 1
            for (int b = 0; b < 5; ++b) {
                int cb = b == 0 ? c[0] : b == 1 ? c[1] : b == 2 ? c[2] : b == 3 ? c[3] : c[4];
                if (b == primary || dead_cell) cb = 0;
                if (!__any(cb > 0)) continue;
                // key7 of the lane's i-th other read if it shows base b, else -1
                const BaseSrc src{kp, b};
                const int no = cb > 0 ? (int)n_other : 0;
                double bs;
                if (!__any(cb > 1)) {
                    // No cell of the wavefront has two reads of this base (sequencing errors: three wavefronts in four): a lane's walk
                    // would be its one read -- the first of its strand and of its base, fk[0] * beta[q][0][n] added to +0 -- without the
                    // slot counting and the run bookkeeping around it.
                    int key = -1;
                    for (int i = 0; __any(i < no); ++i) { const int k7 = i < no ? src(i) : -1; if (k7 >= 0) key = k7; }
                    const uint32_t off1 = key >= 0 ? ((uint32_t)(key >> 1) << 19) + brow : brow;
                    const double B1 = *reinterpret_cast<const double*>(bbase + off1);
                    const double F1 = key >= 0 ? s_fk[0] : s_fk[256];
                    bs = 0.;
                    bs += F1 * B1;
                } else {
                uint64_t qm = 0;
                for (int i = 0; i < no; ++i) { const int key = src(i); if (key >= 0) qm |= 1ull << (key >> 1); }
                uint32_t r_, q_;
                bs = walk_runs(s_cnt, qm, s_fk, bbase, tid, brow, src, no, r_, q_);
                }
                if (cb > 0) {
                    #pragma unroll
                    for (int bb = 0; bb < 5; ++bb) if (bb == b) bsum[bb] = bs;
                }
            }
        }
        const uint32_t n_rev = prim_rev + o_rev;

        GLF_STAMP(6)
        // ---- epilogue of errmod_cal (m=5): float accumulators as in the reference ----
        uint32_t code = 0;
        // The planes' addresses are read here, from device memory, with loads the optimiser must leave in place: as kernel
        // arguments they are invariant over the staging rounds, the address of every (plane, cell) would be formed at the
        // top of the kernel and live in registers (or in scratch) through both phases.
        CallretPlanes cr;
        {
            const volatile unsigned long long *t = reinterpret_cast<const volatile unsigned long long*>(P.crp);
            cr.p15 = reinterpret_cast<float*>(t[0]); cr.pa = reinterpret_cast<float*>(t[1]);
            cr.qs64 = reinterpret_cast<uint64_t*>(t[2]); cr.adf = reinterpret_cast<uint32_t*>(t[3]);
            cr.adr = reinterpret_cast<uint32_t*>(t[4]); cr.cnt4 = reinterpret_cast<uint32_t*>(t[5]);
            cr.misc = reinterpret_cast<uint32_t*>(t[6]);
        }
        if (part && !BCFGPU_ABL(P, 64)) {
            const int nbases = (c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0) + (c[4] > 0);
            if (nbases <= 1) {
                // one base b (or no read at all): every sum over "the other bases" is bsum[b] or nothing, see CallretPlanes
                const int b = c[1] > 0 ? 1 : c[2] > 0 ? 2 : c[3] > 0 ? 3 : c[4] > 0 ? 4 : 0;
                double bsb = bsum[0];
                #pragma unroll
                for (int k = 1; k < 5; ++k) if (k == b) bsb = bsum[k];
                float A = n > 0 ? (float)((double)0.0f + bsb) : 0.0f;
                if (A < 0.0f) A = 0.0f;
                cr.pa[cell] = A;
                code = (uint32_t)b;
            } else {
                code = CR_FULL;
                float *p15c = cr.p15 + (size_t)cell * 16;        // the cell's record: 15 likelihoods in 64 bytes
                #pragma unroll
                for (int j = 0; j < 5; ++j) {
                    float tmp1 = 0.0f; int tmp2 = 0;
                    #pragma unroll
                    for (int k = 0; k < 5; ++k) { if (k == j) continue; tmp1 = (float)((double)tmp1 + bsum[k]); tmp2 += c[k]; }
                    float v = 0.0f;
                    if (n > 0 && tmp2) v = tmp1;
                    if (v < 0.0f) v = 0.0f;
                    p15c[tri(j, j)] = v;
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
                        p15c[tri(j, k)] = h;
                    }
                }
            }
        }
        if (part) {
            // anno[0..3]: ref/alt x fwd/rev counts.  "diff" reads are exactly the non-primary ones when the
            // reference base is A/C/G/T (or at indel sites); with an N reference every read is a diff read.
            const bool all_diff = (!INDEL && ref4c >= 4);
            const uint32_t n_fwd = n - n_rev;
            const uint32_t d_rev = all_diff ? n_rev : o_rev, d_fwd = all_diff ? n_fwd : n_other - o_rev;
            const uint32_t cnt4 = (n_fwd - d_fwd) | (n_rev - d_rev) << 8 | d_fwd << 16 | d_rev << 24;
            // (an over-deep cell: its mark and its WideRec index were left above; qs64 keeps the index, the packed counts below are
            // those of the 255 reads the likelihoods were made of -- their sum is the n of errmod_cal)
            uint32_t wm = 0;
            if (cnt_raw > BCFGPU_MAX_DEPTH) wm = *reinterpret_cast<volatile uint32_t*>(&cr.misc[cell]) & CR_WIDE;
            if (!wm) cr.qs64[cell] = qs64;
            cr.adf[cell] = (uint32_t)ad64; cr.adr[cell] = (uint32_t)(ad64 >> 32); cr.cnt4[cell] = cnt4;
            cr.misc[cell] = code | wm | (scr & 0xff) << 8;  // mq0 and ori_depth only feed site totals: site_sums[12..13]
            done = true;
        }
        if (nb == 0xffffffffu) break;
        base = nb;
        if (!DEEP && !done && deep && beg == base) s_skip = end; // a listed cell at the head of the line: the rounds step over its reads
        __syncthreads();                                         // the slot counters are phase A's partial sums again
        if (!DEEP) {
            const uint32_t sk = s_skip;
            if (sk) { if (deep && beg == base) done = true; base = sk; }
            __syncthreads();                                     // (s_skip is cleared at the top of the round)
        }
    }

    GLF_STAMP(7)
    // ---- flush the workgroup's histograms and site totals ----
    if (LDS_HIST) {
        __syncthreads();
        const int nslot = min(P.hist_slots, P.n_sites - site0);
        for (int i = tid; i < nslot * HP_SIZE; i += WG) {
            const uint32_t v = (uint32_t)s_hist[i];
            const int sl = i / HP_SIZE, j = i - sl * HP_SIZE;
            int *g = P.hist + (long)(site0 + sl) * H_SIZE;
            const int lo = j < HP_MQS ? j : H_FWD_MQS + (j - HP_MQS), hi = j < HP_MQS ? j + H_ALT_OFF : H_REV_MQS + (j - HP_MQS);
            if (v & 0xffffu) atomicAdd(&g[lo], (int)(v & 0xffffu));
            if (v >> 16) atomicAdd(&g[hi], (int)(v >> 16));
        }
        for (int i = tid; i < nslot * SITE_NSUM; i += WG) {
            const unsigned long long v = s_tot[i];
            if (v) atomicAdd(&P.site_sums[(size_t)site0 * SITE_NSUM + i], v);
        }
    }
    GLF_STAMP(8)
}

size_t glfgen_lds_bytes(int cap, int hist_slots)
{
#ifndef GLF_LDS_PAD
#define GLF_LDS_PAD 0        // (occupancy experiments: bytes of LDS a workgroup asks for and does not use)
#endif
    return LDS_HIST_OFF + (size_t)hist_slots * (HP_SIZE * sizeof(int) + SITE_NSUM * 8) + ((size_t)cap + 8) * 2 + GLF_LDS_PAD;
}

template <bool INDEL, bool LDS_HIST>
static void launch_one(const GlfgenParams &p, hipStream_t s, int grid, size_t lds)
{
    if (p.keys) {
        // the two phases as launches of their own: A with the histograms in LDS, B with the key window (and listed cells after it)
        const size_t lds_a = glfgen_lds_bytes(0, p.hist_slots), lds_b = glfgen_lds_bytes(p.lds_cap, 0);
        GlfgenParams pb = p;
        pb.hist_slots = 0;                                  // phase B keeps no histograms: its LDS is fk + run counters + keys
        if (lds_b > 48 * 1024)
            hipFuncSetAttribute(reinterpret_cast<const void*>(glfgen_kernel<INDEL, false, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
        hipLaunchKernelGGL((glfgen_kernel<INDEL, LDS_HIST, false, 1>), dim3(grid), dim3(WG), lds_a, s, p);
        hipLaunchKernelGGL((glfgen_kernel<INDEL, false, false, 2>), dim3(grid), dim3(WG), lds_b, s, pb);
        if (p.deep_cap) hipLaunchKernelGGL((glfgen_kernel<INDEL, false, true, 2>), dim3(p.deep_cap), dim3(WG), glfgen_lds_bytes(0, 0), s, pb);
        return;
    }
    if (lds > 48 * 1024)    // per launch, on the device the caller has bound: no process-wide state
        hipFuncSetAttribute(reinterpret_cast<const void*>(glfgen_kernel<INDEL, LDS_HIST, false, 0>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((glfgen_kernel<INDEL, LDS_HIST, false, 0>), dim3(grid), dim3(WG), lds, s, p);
    // the cells the launch above listed (none, as a rule: the workgroups leave at once); no key window in LDS
    if (p.deep_cap) hipLaunchKernelGGL((glfgen_kernel<INDEL, LDS_HIST, true, 0>), dim3(p.deep_cap), dim3(WG), glfgen_lds_bytes(0, p.hist_slots), s, p);
}

void launch_glfgen(const GlfgenParams &p, hipStream_t s)
{
    const long ncells = (long)p.n_sites * p.n_smpl;
    if (ncells == 0) return;
    const int grid = (int)((ncells + WG - 1) / WG);
    const size_t lds = glfgen_lds_bytes(p.lds_cap, p.hist_slots);
    if (p.is_indel) { if (p.hist_slots) launch_one<true, true>(p, s, grid, lds); else launch_one<true, false>(p, s, grid, lds); }
    else            { if (p.hist_slots) launch_one<false, true>(p, s, grid, lds); else launch_one<false, false>(p, s, grid, lds); }
}

}  // namespace bcfgpu

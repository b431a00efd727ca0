// pileup.hip -- the pileup itself on the device: from the reads of a region (a flat pool, as bcfgpu_baq and
// bcfgpu_gap_prep take it) to the site x sample x read tile that bcfgpu_mpileup / bcfgpu_pipeline run on.
//
// In the reference this is htslib's bam_mplp iterator plus the per-read accessors of bcf_call_glfgen (mpileup.c:320-347,
// bam2bcf.c:170-236): every pileup column lists, per sample, the reads covering the position with their query offset,
// is_del / is_refskip and the following indel.  A host-built tile costs 5 bytes per (read, position) over PCIe; the read
// pool it is built from is ~30x smaller (a 100-bp read is in 100 columns), so building the tile in HBM takes the PCIe
// link out of the way of the kernels.
//
// Mapping: one lane per (site, sample) cell -- the same gather-by-cell shape as glfgen_kernel, no atomics.  The reads of
// a sample are in ascending position order (a sorted BAM), so the reads that can cover position x are the window
// [first read with pos > x - max_span, first read with pos > x): two binary searches, then a scan of ~depth
// candidates.  Pass 1 counts the covering reads of every cell, a prefix sum gives plp_off, pass 2 resolves every
// covering read's CIGAR at x (query offset, deletion / reference skip, indel after the position: htslib's
// resolve_cigar) and writes the packed records of bcfgpu_pack_read into the cell's slice, in read order.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <climits>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>
#include "kernels.h"

extern "C" int bcfgpu_internal_device(bcfgpu_ctx *ctx, hipStream_t *stream, const float **q2p);
extern "C" void *bcfgpu_internal_ws(bcfgpu_ctx *ctx, int slot, size_t bytes);
extern "C" void *bcfgpu_internal_pinned(bcfgpu_ctx *ctx, int slot, size_t bytes);
extern "C" const bcfgpu_cfg *bcfgpu_internal_cfg(const bcfgpu_ctx *ctx);
extern "C" void *bcfgpu_internal_pileup_state(bcfgpu_ctx *ctx);
int bcfgpu_set_error(int code, const char *what);

namespace bcfgpu {

// what a pileup entry needs from its read, in one 32-byte record (two 16-byte gathers per entry instead of nine scalar
// ones); a read whose CIGAR is a single operation -- the common case -- needs no CIGAR access at all
struct ReadMeta {
    int32_t pos, end;               // reference span [pos, end)
    uint32_t seq_off, cig_off;
    uint16_t lq, ntot;              // query length; M/=/X/I bases (get_position's denominator)
    uint8_t ncig, bits, mapq, pad;  // bits: 1 reverse strand, 2 has a soft clip, 4 unmapped
    uint32_t cig0;                  // the first CIGAR operation
    uint32_t pad2;
};
static_assert(sizeof(ReadMeta) == 32, "one read = two 16-byte loads");

struct PileupParams {
    int n_sites, n_smpl, beg, want_epos, max_span;
    // reads in (sample, position) order: index k of the sorted list
    const int32_t *smpl_off;        // [n_smpl+1] into the sorted list
    const int32_t *s_pos;           // [n_reads] reference start of sorted read k (the binary searches)
    const int32_t *s_read;          // [n_reads] pool index of sorted read k
    const ReadMeta *meta;           // [n_reads] everything else about sorted read k, one 32-byte record
    const uint32_t *cig;
    const uint8_t *seq16, *qual;
    // out
    uint32_t *cnt;                  // [n_cells + 1] pass 1: reads per cell; after the scan: plp_off
    uint32_t *rd; uint8_t *epos;    // pass 2
    uint32_t *col_indel;            // [n_sites] entries of the column that are followed by an indel (n_alt of bam2bcf_indel.c:117-140)
    int n_reads;                    // reads of the pool (bcfgpu_gap_prep_tile)
    uint32_t n_bases;               // bases of the pool's seq16 / qual
    const int *d_span;              // the longest reference span of a read, as pileup_meta_kernel left it (max_span once the host has read it)
    int ref_len;                    // length of the contig handed to bcfgpu_pileup (bcfgpu_gap_prep_tile: the end of par->ref)
};

// first k in [lo, hi) with a[k] > x
__device__ __forceinline__ int upper_bound(const int32_t *a, int lo, int hi, int x)
{
    while (lo < hi) { const int m = (lo + hi) >> 1; if (a[m] > x) hi = m; else lo = m + 1; }
    return lo;
}

// A workgroup owns PT_COLS consecutive columns x PT_SMPL consecutive samples.  Lanes that are neighbours in the column index
// walk the same reads of one sample one base apart: their record loads fall on the same cache lines and their base / quality
// bytes are consecutive (a lane per cell in cell order -- neighbours = different samples -- made every load of every lane a
// cache line of its own: 14 ms for the 16 384-column x 1000-sample region of bench.py --mode pileup, this mapping: see DESIGN 5).
// In the tile a column's PT_SMPL cells are one contiguous chunk (site-major plp_off): the records are collected in LDS chunk
// by chunk and written out as PT_COLS contiguous spans.
#ifndef PT_COLS
#define PT_COLS 16
#endif
#define PT_SMPL (256 / PT_COLS)
// entries a workgroup's 256 cells may hold for that (a lane's own stores would be 4 + 1 bytes per entry, 120 bytes apart
// from its neighbour's); beyond it the lanes store directly
#define PILEUP_LDS_CAP 9216

template <bool FILL>
__global__ __launch_bounds__(256) void pileup_kernel(const PileupParams P)
{
    __shared__ uint32_t s_rd[FILL ? PILEUP_LDS_CAP : 1];
    __shared__ uint8_t s_ep[FILL ? PILEUP_LDS_CAP : 4];
    __shared__ uint32_t s_cb[PT_COLS], s_lb[PT_COLS + 1];        // chunk begin in the tile; chunk begin in LDS (+ total)
    // Workgroups are dealt to the 8 XCDs round-robin, and each XCD has its own L2: every XCD gets a contiguous range of
    // (sample tile, column tile) pairs with the column tile running fastest, so that the workgroups in flight on one XCD
    // walk the same samples' reads a few columns apart and find them in that L2.
    const unsigned n_bt = (unsigned)((P.n_sites + PT_COLS - 1) / PT_COLS), n_work = n_bt * (unsigned)((P.n_smpl + PT_SMPL - 1) / PT_SMPL);
    const unsigned per_xcd = gridDim.x >> 3;             // (the grid is a multiple of 8)
    const unsigned w = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    const bool in_grid = w < n_work;
    const int st = in_grid ? (int)(w / n_bt) : 0, bt = in_grid ? (int)(w - (unsigned)st * n_bt) : 0;
    const int c_local = threadIdx.x & (PT_COLS - 1), s_local = threadIdx.x / PT_COLS;
    const int site = bt * PT_COLS + c_local, s = st * PT_SMPL + s_local;
    const bool live = in_grid && site < P.n_sites && s < P.n_smpl;
    const long cell = (long)site * P.n_smpl + s;
    bool staged = false;
    if (FILL) {
        if (threadIdx.x < PT_COLS) {
            const int st_site = bt * PT_COLS + (int)threadIdx.x;
            uint32_t b = 0, e = 0;
            if (in_grid && st_site < P.n_sites) {
                const long c0 = (long)st_site * P.n_smpl + st * PT_SMPL;
                const int s1 = min(st * PT_SMPL + PT_SMPL, P.n_smpl);
                b = P.cnt[c0]; e = P.cnt[(long)st_site * P.n_smpl + s1];
            }
            s_cb[threadIdx.x] = b;
            // exclusive prefix sum of the chunk sizes over the 16 lanes
            uint32_t v = e - b, inc = v;
            #pragma unroll
            for (int d = 1; d < PT_COLS; d <<= 1) { const uint32_t u = __shfl_up(inc, d, PT_COLS); if ((int)threadIdx.x >= d) inc += u; }
            s_lb[threadIdx.x] = inc - v;
            if (threadIdx.x == PT_COLS - 1) s_lb[PT_COLS] = inc;
        }
        __syncthreads();
        staged = s_lb[PT_COLS] <= PILEUP_LDS_CAP;
    }
    // (the PT_COLS lanes of a sample act together below: columns past the tile's edge walk along with the last column and
    // produce nothing)
    if (in_grid && s < P.n_smpl) {
    const int x = P.beg + min(site, P.n_sites - 1);
    const int lo0 = P.smpl_off[s], hi0 = P.smpl_off[s + 1];
    const int hi = upper_bound(P.s_pos, lo0, hi0, x);                    // reads starting at or before x
    const int lo = upper_bound(P.s_pos, lo0, hi, x - (FILL ? P.max_span : *P.d_span));   // ... that can still reach x
    if (!FILL) {
        uint32_t n = 0;
        for (int k = lo; k < hi; ++k) n += P.meta[k].end > x ? 1u : 0u;
        if (live) P.cnt[cell] = n;
        return;
    }
    // One walk for the sample's PT_COLS columns: from the first column's first candidate to the last column's last, every
    // lane testing the read against its own column.  The lanes then ask for the same record in the same trip (one cache line
    // for the group instead of one per lane whose own range starts a read or two later) and for neighbouring bytes.
    int lo_c = lo, hi_c = hi;
    #pragma unroll
    for (int d = 1; d < PT_COLS; d <<= 1) { lo_c = min(lo_c, __shfl_xor(lo_c, d)); hi_c = max(hi_c, __shfl_xor(hi_c, d)); }
    uint32_t o = live ? P.cnt[cell] : 0u;                                // plp_off after the scan
    if (staged) o = o - s_cb[c_local] + s_lb[c_local];

    // The walk over the sample's reads that can reach x, software-pipelined by hand: a read's record (two 16-byte loads) is
    // fetched one trip ahead, its base and quality bytes are requested in the trip that resolves its CIGAR and used two
    // trips later, so that a trip waits for loads issued one and two trips ago instead of three dependent ones in a row
    // (at three wavefronts per SIMD -- the LDS staging -- the wavefronts sat in s_waitcnt 70 % of their time).
    uint32_t any_indel = 0;                                              // entries of the cell that are followed by an indel
    const uint4 *meta4 = reinterpret_cast<const uint4*>(P.meta);
    uint4 n0 = make_uint4(0, 0, 0, 0), n1 = n0;                          // the record of read k, fetched ahead
    const int k_last = P.n_reads - 1;
    if (lo_c < hi_c) { n0 = meta4[2 * (size_t)lo_c]; n1 = meta4[2 * (size_t)lo_c + 1]; }
    // two entries in flight: what is known of the entry without the bytes (w, e), whether a base exists (h), the bytes.
    // (Two register sets for the records and two for the entries, the trips written out in pairs: a value that is moved
    // from one variable to the next at the end of a trip counts as used there, and the compiler waits for every load.)
    struct Pend { bool v, h; uint32_t w, e, nt, bq; };
    Pend qa = {false, false, 0, 0, 0, 0}, qb = qa;
    uint4 na0 = make_uint4(0, 0, 0, 0), na1 = na0;
    auto trip = [&](const int k, const uint4 &m0, const uint4 &m1, uint4 &f0, uint4 &f1, Pend &q) __attribute__((always_inline)) {
        // (fetched whether or not there is a next read, from a clamped index: a load under a condition makes the compiler
        // wait for all outstanding loads at the next use instead of counting them)
        const size_t kf = (size_t)min(k + 1, k_last);
        f0 = meta4[2 * kf]; f1 = meta4[2 * kf + 1];
        // Straight-line code for the common read (one aligned block), for every lane whether its read covers x or not (v0 says
        // so at the end); the walk over a longer CIGAR only when some lane of the wavefront needs it.  Nested per-lane
        // branches around both cases cost as many scalar instructions (exec masks) as the entry did vector ones.
        bool v0 = live && k < hi_c && (int)m0.x <= x && (int)m0.y > x, h0 = false;
        uint32_t w0 = 0, e0 = 0, idx = 0;
        {
            const int rpos = (int)m0.x, lq = (int)(m1.x & 0xffff), ntot = (int)(m1.x >> 16);
            const int ncig = (int)(m1.y & 0xff);
            const uint32_t bits = (m1.y >> 8) & 0xff, mapq = (m1.y >> 16) & 0xff, so = m0.z;
            int qpos = x - rpos, is_del = 0, is_skip = 0, indel = 0, edist = qpos + 1;
            const int op0 = m1.z & 0xf;
            const bool walk = v0 && !(ncig == 1 && (op0 == 0 || op0 == 7 || op0 == 8));
            if (__any(walk)) if (walk) {
                qpos = 0;
                // htslib's resolve_cigar at reference position x
                const uint32_t *cg = P.cig + m0.w;
                int rx = rpos, y = 0;
                for (int c = 0; c < ncig; ++c) {
                    const int op = cg[c] & 0xf, l = (int)(cg[c] >> 4);
                    if (op == 0 || op == 7 || op == 8) {
                        if (x < rx + l) {
                            qpos = y + (x - rx);
                            if (x == rx + l - 1) {                       // last base of the block: what follows?
                                int cc = c + 1;
                                while (cc < ncig && (cg[cc] & 0xf) == 6) ++cc;   // pads
                                if (cc < ncig) {
                                    const int nop = cg[cc] & 0xf;
                                    if (nop == 1) {
                                        indel = (int)(cg[cc] >> 4);
                                        for (++cc; cc < ncig && ((cg[cc] & 0xf) == 1 || (cg[cc] & 0xf) == 6); ++cc)
                                            if ((cg[cc] & 0xf) == 1) indel += (int)(cg[cc] >> 4);
                                    } else if (nop == 2) indel = -(int)(cg[cc] >> 4);
                                }
                            }
                            break;
                        }
                        rx += l; y += l;
                    } else if (op == 2 || op == 3) {
                        if (x < rx + l) { qpos = y; is_del = 1; is_skip = op == 3; break; }
                        rx += l;
                    } else if (op == 1 || op == 4) y += l;
                }
                any_indel += indel != 0 ? 1u : 0u;
                // get_position, bam2bcf.c:80-114
                int iread = 0;
                edist = qpos + 1;
                if (P.want_epos && (bits & 2))
                    for (int c = 0; c < ncig; ++c) {
                        const int op = cg[c] & 0xf, l = (int)(cg[c] >> 4);
                        if (op == 0 || op == 7 || op == 8 || op == 1) iread += l;
                        else if (op == 4) { iread += l; if (iread <= qpos) edist -= l; }
                    }
            }
            // the record of bcfgpu_pack_read, but for the base and its quality
            int tail = lq - 1 - qpos;
            if (tail > qpos) tail = qpos;
            tail = tail < 0 ? 0 : tail > 255 ? 255 : tail;
            w0 = mapq << 8 | ((bits & 1) ? BCFGPU_RD_REV : 0) | ((bits & 2) ? BCFGPU_RD_SCLIP : 0) | (is_del ? BCFGPU_RD_DEL : 0)
                    | ((is_skip || (bits & 4)) ? BCFGPU_RD_SKIP : 0) | (uint32_t)tail << 24;
            if (P.want_epos) {
                int e = (int)((double)edist / (ntot + 1) * BCFGPU_NPOS);
                e0 = (uint32_t)(e < 0 ? 0 : e > BCFGPU_NPOS - 1 ? BCFGPU_NPOS - 1 : e);
            }
            h0 = v0 && qpos >= 0 && qpos < lq;
            idx = h0 ? so + (uint32_t)qpos : 0u;
        }
        // the entry whose bytes were requested two trips ago
        if (q.v) {
            const uint32_t word = q.w | (q.h ? q.bq : 0u) | (q.h ? (q.nt & 15u) : 15u) << 16;
            if (staged) { s_rd[o] = word; s_ep[o] = (uint8_t)q.e; }
            else { P.rd[o] = word; P.epos[o] = (uint8_t)q.e; }
            ++o;
        }
        q.v = v0; q.h = h0; q.w = w0; q.e = e0;
        q.nt = P.seq16[idx]; q.bq = P.qual[idx];                          // (unconditional: the pool is never empty, see bcfgpu_pileup)
    };
    if (lo_c < hi_c)
    for (int k = lo_c; k < hi_c + 2; k += 2) {
        trip(k, n0, n1, na0, na1, qa);
        trip(k + 1, na0, na1, n0, n1, qb);
    }
    if (any_indel && live && P.col_indel) atomicAdd(&P.col_indel[site], any_indel);                // (the fill pass walks the CIGARs: once per entry)
    }
    if (FILL && staged) {
        __syncthreads();
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int c = wave; c < PT_COLS; c += 4) {
            const uint32_t gb = s_cb[c], lb = s_lb[c], n = s_lb[c + 1] - lb;
            for (uint32_t i = lane; i < n; i += 64) P.rd[gb + i] = s_rd[lb + i];
            // the bytes: single ones up to a 4-byte boundary of the output, then words, then the rest
            const uint32_t head = min((4u - (gb & 3u)) & 3u, n), nw = (n - head) >> 2, tail0 = head + 4 * nw;
            if ((uint32_t)lane < head) P.epos[gb + lane] = s_ep[lb + lane];
            for (uint32_t i = lane; i < nw; i += 64) {
                const uint8_t *b = s_ep + lb + head + 4 * i;
                *reinterpret_cast<uint32_t*>(P.epos + gb + head + 4 * i) = (uint32_t)b[0] | (uint32_t)b[1] << 8 | (uint32_t)b[2] << 16 | (uint32_t)b[3] << 24;
            }
            if ((uint32_t)lane < n - tail0) P.epos[gb + tail0 + lane] = s_ep[lb + tail0 + lane];
        }
    }
}

// The pileup entries of selected columns as bcf_call_gap_prep wants them (read, query offset, indel): the same walk as
// the fill pass over the cells of those columns only.  sel_off = exclusive prefix sum of the cells' counts.
struct EntriesParams {
    PileupParams P;
    int n_cols;
    const int32_t *cols;            // [n_cols] column indices
    uint32_t *sel_cnt;              // [n_cols*n_smpl + 1] counts, then offsets
    int32_t *e_read, *e_qpos, *e_indel;
    const uint8_t *col_keep;        // NULL, or [n_cols]: 0 = the column has no entries as far as the stage is concerned (gap_support_kernel)
};

// bcf_call_gap_prep gives up on a column whose indel reads, pooled over the samples, are fewer than min_support or a smaller
// share of the column's reads than min_frac (bam2bcf_indel.c:150-154, the default: no -p / per_sample_flt) -- in a large
// cohort nearly every column has some read with an indel and nearly none passes.  gap_keep_kernel (below) decides it from two
// numbers the pileup left per column, before any of the column's entries is listed.

// what the pileup says about read record m at reference position x (htslib resolve_cigar): query offset, indel after x
__device__ __forceinline__ void entry_of_read(const ReadMeta &m, const uint32_t *cig, int x, int &qpos, int &indel)
{
    const uint32_t *cg = cig + m.cig_off;
    int rx = m.pos, y = 0;
    qpos = 0; indel = 0;
    for (int c = 0; c < m.ncig; ++c) {
        const int op = cg[c] & 0xf, l = (int)(cg[c] >> 4);
        if (op == 0 || op == 7 || op == 8) {
            if (x < rx + l) {
                qpos = y + (x - rx);
                if (x == rx + l - 1) {
                    int cc = c + 1;
                    while (cc < m.ncig && (cg[cc] & 0xf) == 6) ++cc;
                    if (cc < m.ncig) {
                        const int nop = cg[cc] & 0xf;
                        if (nop == 1) {
                            indel = (int)(cg[cc] >> 4);
                            for (++cc; cc < m.ncig && ((cg[cc] & 0xf) == 1 || (cg[cc] & 0xf) == 6); ++cc)
                                if ((cg[cc] & 0xf) == 1) indel += (int)(cg[cc] >> 4);
                        } else if (nop == 2) indel = -(int)(cg[cc] >> 4);
                    }
                }
                break;
            }
            rx += l; y += l;
        } else if (op == 2 || op == 3) {
            if (x < rx + l) { qpos = y; break; }
            rx += l;
        } else if (op == 1 || op == 4) y += l;
    }
}

// FILL = false: a lane per cell (the counts are already in plp_off).  FILL = true: a wavefront per cell, a lane per
// candidate read; the covering reads keep their order through a ballot prefix count.
template <bool FILL>
__global__ __launch_bounds__(256) void entries_kernel(const EntriesParams E)
{
    const PileupParams &P = E.P;
    if (!FILL) {
        const long i = (long)blockIdx.x * 256 + threadIdx.x;
        if (i >= (long)E.n_cols * P.n_smpl) return;
        const int ci = (int)(i / P.n_smpl), s = (int)(i - (long)ci * P.n_smpl);
        const long cell = (long)E.cols[ci] * P.n_smpl + s;
        E.sel_cnt[i] = (E.col_keep && !E.col_keep[ci]) ? 0u : P.cnt[cell + 1] - P.cnt[cell];
        return;
    }
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= (long)E.n_cols * P.n_smpl) return;
    const int ci = (int)(i / P.n_smpl), s = (int)(i - (long)ci * P.n_smpl), site = E.cols[ci];
    const int x = P.beg + site;
    const int lo0 = P.smpl_off[s], hi0 = P.smpl_off[s + 1];
    const int hi = upper_bound(P.s_pos, lo0, hi0, x);
    const int lo = upper_bound(P.s_pos, lo0, hi, x - P.max_span);
    uint32_t o = E.sel_cnt[i];
    if (E.col_keep && !E.col_keep[ci]) return;
    for (int kb = lo; kb < hi; kb += 64) {
        const int k = kb + lane;
        bool covers = false;
        int qpos = 0, indel = 0;
        if (k < hi) {
            const ReadMeta m = P.meta[k];
            covers = m.end > x;
            if (covers) entry_of_read(m, P.cig, x, qpos, indel);
        }
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(covers);
        if (covers) {
            const uint32_t at = o + (uint32_t)__popcll(mask & ((1ull << lane) - 1));
            E.e_read[at] = P.s_read ? P.s_read[k] : k; E.e_qpos[at] = qpos; E.e_indel[at] = indel;
        }
        o += (uint32_t)__popcll(mask);
    }
}

// The records of selected columns copied into a tile of their own (the indel pass runs on the candidate columns only)
template <bool FILL>
__global__ __launch_bounds__(256) void subtile_kernel(const EntriesParams E, uint32_t *rd_out, uint8_t *ep_out)
{
    const PileupParams &P = E.P;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)E.n_cols * P.n_smpl) return;
    const int ci = (int)(i / P.n_smpl), s = (int)(i - (long)ci * P.n_smpl);
    const long cell = (long)E.cols[ci] * P.n_smpl + s;
    const uint32_t b = P.cnt[cell], e = P.cnt[cell + 1];
    if (E.col_keep && !E.col_keep[ci]) { if (!FILL) E.sel_cnt[i] = 0; return; }
    if (!FILL) { E.sel_cnt[i] = e - b; return; }
    uint32_t o = E.sel_cnt[i];
    for (uint32_t k = b; k < e; ++k, ++o) { rd_out[o] = P.rd[k]; ep_out[o] = P.epos[k]; }
}

// The candidate columns bcf_call_gap_prep goes on with: the pooled support filter above and the compaction of the columns that
// pass it, in one workgroup (a tile has some ten thousand candidates; in a large cohort a hundredth of them passes).
// kcols[j] = the column, kidx[j] = its place in `cols`; n_keep[0] = how many.  use_filter = 0 (per_sample_flt): every column.
__global__ __launch_bounds__(1024) void gap_keep_kernel(const PileupParams P, const int32_t *cols, int n_cols, int use_filter, int min_support,
                                                        double min_frac, int32_t *kcols, int32_t *kidx, int32_t *n_keep)
{
    __shared__ int s_wave[16], s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n_cols; i0 += 1024) {
        const int i = i0 + tid;
        bool keep = false;
        long c = 0;
        if (i < n_cols) {
            c = cols[i];
            keep = true;
            if (use_filter) {
                const uint32_t n_alt = P.col_indel[c], n_tot = P.cnt[(c + 1) * P.n_smpl] - P.cnt[c * P.n_smpl];
                keep = !(n_tot == 0 || (double)n_alt / n_tot < min_frac || (int)n_alt < min_support);
            }
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        if (lane == 0) s_wave[wave] = (int)__popcll(m);
        __syncthreads();
        int before = s_base;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        if (keep) { const int at = before + (int)__popcll(m & ((1ull << lane) - 1)); kcols[at] = (int32_t)c; kidx[at] = i; }
        __syncthreads();
        if (tid == 0) { int t = s_base; for (int w = 0; w < 16; ++w) t += s_wave[w]; s_base = t; }
        __syncthreads();
    }
    if (tid == 0) n_keep[0] = s_base;
}

// The indel pass's tile: the columns where bcf_call_gap_prep returned 0 (mpileup.c:354-360), picked from the columns the stage
// ran on.  lk[j] = the place of tile column j among those columns, lcols[j] = its column of the pileup; ksel = the entry offsets
// of the stage's columns.  COUNT: lsel[j*S + s] = entries of the cell (scanned by the caller); else a lane per cell copies the
// cell's read records from the pileup and its p->aux words from the stage's entry-ordered array.
template <bool COUNT>
__global__ __launch_bounds__(256) void live_tile_kernel(const PileupParams P, int n_live, const int32_t *lcols, const int32_t *lk, const uint32_t *ksel,
                                                        uint32_t *lsel, const uint32_t *aux_in, uint32_t *rd_out, uint8_t *ep_out, uint32_t *aux_out)
{
    if (COUNT) {                                             // a lane per cell
        const long i = (long)blockIdx.x * 256 + threadIdx.x;
        if (i >= (long)n_live * P.n_smpl) return;
        const int j = (int)(i / P.n_smpl), s = (int)(i - (long)j * P.n_smpl);
        const size_t kc = (size_t)lk[j] * P.n_smpl + s;
        lsel[i] = ksel[kc + 1] - ksel[kc];
        return;
    }
    // a lane per (tile column, entry of the column): a column's entries are one stretch of the pileup, of the stage's array and of the tile
    for (int j = blockIdx.y; j < n_live; j += gridDim.y) {
        const size_t kc0 = (size_t)lk[j] * P.n_smpl;
        const uint32_t kb = ksel[kc0], n = ksel[kc0 + P.n_smpl] - kb;
        const uint32_t b = P.cnt[(size_t)lcols[j] * P.n_smpl], o = lsel[(size_t)j * P.n_smpl];
        for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < n; k += gridDim.x * 256u) {
            rd_out[o + k] = P.rd[b + k]; ep_out[o + k] = P.epos[b + k]; aux_out[o + k] = aux_in[kb + k];
        }
    }
}

// One record per read (reference span, constants) from the caller's per-read arrays: a pass over the CIGARs, one lane per read
// of the (sample, position)-ordered list.  status[0] = the longest reference span, status[1] = 1: a read too long for the
// record's fields, 2: a sample's reads out of position order.
struct MetaParams {
    int n, n_smpl;
    const int32_t *r_pos, *r_lq, *r_flag, *r_ncig, *r_cig_off, *r_seq_off;
    const uint8_t *r_mapq;
    const uint8_t *keep;            // NULL, or [n] by pool index: 0 = the read does not enter the pileup
    const int32_t *s_read;          // pool index of sorted read k, or NULL: the pool is in that order already
    const int32_t *smpl_off;
    const uint32_t *cig;
    ReadMeta *meta; int32_t *s_pos; int *status;
};
__global__ __launch_bounds__(256) void pileup_meta_kernel(const MetaParams M)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    int span = 1, bad = 0;
    if (k < M.n) {
        const int r = M.s_read ? M.s_read[k] : k;
        const int pos = M.r_pos[r], ncig = M.r_ncig[r], lq = M.r_lq[r], flag = M.r_flag[r];
        const uint32_t coff = (uint32_t)M.r_cig_off[r];
        const uint32_t *cg = M.cig + coff;
        int x = pos, ntot = 0, scl = 0;
        for (int c = 0; c < ncig; ++c) {
            const int op = cg[c] & 0xf, l = (int)(cg[c] >> 4);
            if (op == 0 || op == 7 || op == 8) { x += l; ntot += l; }
            else if (op == 2 || op == 3) x += l;
            else if (op == 1) ntot += l;
            else if (op == 4) scl = 1;
        }
        if (lq > 65535 || lq < 0 || ntot > 65535 || ncig > 255 || ncig < 0) bad = 1;
        const bool out = M.keep && !M.keep[r];                      // dropped by the caller's filters: covers no column
        ReadMeta m;
        m.pos = pos; m.end = (bad || out) ? pos : x; m.seq_off = (uint32_t)M.r_seq_off[r]; m.cig_off = coff;
        m.lq = (uint16_t)lq; m.ntot = (uint16_t)ntot; m.ncig = (uint8_t)ncig;
        m.bits = (uint8_t)(((flag & 16) ? 1 : 0) | (scl ? 2 : 0) | ((flag & 4) ? 4 : 0));
        m.mapq = M.r_mapq[r]; m.pad = 0; m.cig0 = ncig > 0 ? cg[0] : 0; m.pad2 = 0;
        uint4 *o = reinterpret_cast<uint4*>(M.meta + k);
        const uint4 *mi = reinterpret_cast<const uint4*>(&m);
        o[0] = mi[0]; o[1] = mi[1];
        M.s_pos[k] = pos;
        if (!bad && !out) span = x - pos;
        if (k > 0) {
            const int rp = M.s_read ? M.s_read[k - 1] : k - 1;
            if (pos < M.r_pos[rp]) {                                     // fine only across a sample boundary
                const int idx = upper_bound(M.smpl_off, 0, M.n_smpl + 1, k);
                if (M.smpl_off[idx - 1] != k) bad |= 2;
            }
        }
    }
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) { span = max(span, __shfl_xor(span, o)); bad |= __shfl_xor(bad, o); }
    if ((threadIdx.x & 63) == 0) {
        // (the maximum only grows: a wavefront that does not raise what is there already stays away from the word -- 77 000
        // wavefronts queueing on one address were most of this kernel's time)
        if (span > 1 && span > *reinterpret_cast<volatile int*>(&M.status[0])) atomicMax(&M.status[0], span);
        if (bad) atomicOr(&M.status[1], bad);
    }
}

// [lowest start, highest end) of the pool's reads on the reference
__global__ __launch_bounds__(256) void pool_extent_kernel(const DevPool D, int *out)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    int lo = INT32_MAX, hi = 0;
    if (r < D.n_reads) {
        int x = D.r_pos[r];
        lo = x;
        const uint32_t *cg = D.cig + D.r_cig_off[r];
        for (int k = 0; k < D.r_ncig[r]; ++k) { const int op = cg[k] & 15; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) x += (int)(cg[k] >> 4); }
        hi = x;
    }
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
    if ((threadIdx.x & 63) == 0 && lo != INT32_MAX) { atomicMin(&out[0], lo); atomicMax(&out[1], hi); }
}

// bcfgpu_read12 records to the per-read arrays the stages index; the lengths (rounded up to four bases) and operation counts
// in the offset arrays, which an exclusive prefix sum then turns into r_seq_off / r_cig_off
__global__ __launch_bounds__(256) void pool_expand_kernel(const bcfgpu_read12 *rec, int n, int32_t *r_pos, int32_t *r_lq, int32_t *r_flag,
                                                         int32_t *r_ncig, int32_t *r_cig_off, int32_t *r_seq_off, uint8_t *r_mapq)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r > n) return;
    if (r == n) { r_cig_off[n] = 0; r_seq_off[n] = 0; return; }           // (the scans run over n + 1 elements)
    const uint32_t *w = reinterpret_cast<const uint32_t*>(rec + r);
    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
    const int lq = (int)(w1 & 0xffff), ncig = (int)((w1 >> 16) & 0xff), f8 = (int)(w1 >> 24);
    r_pos[r] = (int32_t)w0; r_lq[r] = lq; r_ncig[r] = ncig; r_flag[r] = ((f8 & 1) ? 16 : 0) | ((f8 & 2) ? 4 : 0);
    r_mapq[r] = (uint8_t)(w2 & 0xff);
    r_seq_off[r] = (lq + 3) & ~3; r_cig_off[r] = ncig;
}

// The pool as BAM records hold it (two 4-bit base codes per byte, high nibble first) and qualities as palette indices, to the
// one-byte-per-base arrays every kernel of the host-fed stages reads.  A lane per four input bytes = eight bases.
__global__ __launch_bounds__(256) void pileup_unpack_kernel(const uint8_t *seq4, const uint8_t *qual4, unsigned long long pal_lo,
                                                            unsigned long long pal_hi, size_t n_in, uint8_t *seq16, uint8_t *qual, int qual_bits)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (4 * t >= n_in) return;
    if (qual4 && qual_bits == 2) {               // four indices per byte: this lane's eight bases are two input bytes
        const uint32_t v = *reinterpret_cast<const uint16_t*>(qual4 + 2 * t);
        uint32_t lo = 0, hi = 0;
        #pragma unroll
        for (int b = 0; b < 4; ++b) {
            lo |= (uint32_t)((pal_lo >> (8 * ((v >> (6 - 2 * b)) & 3))) & 0xff) << (8 * b);
            hi |= (uint32_t)((pal_lo >> (8 * ((v >> (14 - 2 * b)) & 3))) & 0xff) << (8 * b);
        }
        *reinterpret_cast<uint2*>(qual + 8 * t) = make_uint2(lo, hi);
        qual4 = nullptr;
    }
    if (seq4) {
        const uint32_t v = *reinterpret_cast<const uint32_t*>(seq4 + 4 * t);
        uint32_t lo = 0, hi = 0;
        #pragma unroll
        for (int b = 0; b < 2; ++b) {
            const uint32_t by = (v >> (8 * b)) & 0xff, by2 = (v >> (8 * b + 16)) & 0xff;
            lo |= ((by >> 4) | (by & 15) << 8) << (16 * b);
            hi |= ((by2 >> 4) | (by2 & 15) << 8) << (16 * b);
        }
        *reinterpret_cast<uint2*>(seq16 + 8 * t) = make_uint2(lo, hi);
    }
    if (qual4) {
        const uint32_t v = *reinterpret_cast<const uint32_t*>(qual4 + 4 * t);
        auto pal = [&](uint32_t j) -> uint32_t { return (uint32_t)(((j & 8) ? pal_hi : pal_lo) >> (8 * (j & 7))) & 0xff; };
        uint32_t lo = 0, hi = 0;
        #pragma unroll
        for (int b = 0; b < 2; ++b) {
            const uint32_t by = (v >> (8 * b)) & 0xff, by2 = (v >> (8 * b + 16)) & 0xff;
            lo |= (pal(by >> 4) | pal(by & 15) << 8) << (16 * b);
            hi |= (pal(by2 >> 4) | pal(by2 & 15) << 8) << (16 * b);
        }
        *reinterpret_cast<uint2*>(qual + 8 * t) = make_uint2(lo, hi);
    }
}

}  // namespace bcfgpu

using namespace bcfgpu;

// seq_nt16_table restricted to what a reference sequence holds (IUPAC codes, case-insensitive; anything else is N)
// sum of the per-cell counts in 64 bits (grid-stride, one atomic per workgroup)
__global__ __launch_bounds__(256) void count_total_kernel(const uint32_t *cnt, size_t n, unsigned long long *tot)
{
    unsigned long long v = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) v += cnt[i];
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(tot, v);
}

static int ref_nt16(char c)
{
    static const char *codes = "=ACMGRSVTWYHKDBN";
    if (c >= 'a' && c <= 'z') c = (char)(c - 32);
    for (int i = 1; i < 16; ++i) if (codes[i] == c) return i;
    return 15;
}

// ---- the read pool in HBM, and the tile built from it ----------------------------------------------------------------
// bcfgpu_pileup / bcfgpu_pileup_packed = pool_upload + pool_pileup.  The host part is argument checks, the read -> sample
// bookkeeping (one pass over r_smpl, none when the caller hands over smpl_off), uploads of the caller's arrays as they are,
// and launches.
extern "C" void *bcfgpu_internal_pool_state(bcfgpu_ctx *ctx);

static int host_threads(int n)
{
    int nthr = (int)std::thread::hardware_concurrency();
    if (const char *e = getenv("BCFGPU_HOST_THREADS")) nthr = atoi(e);
    return std::max(1, std::min(std::min(nthr, 16), n / 262144 + 1));
}
template <class F> static void on_threads(int nthr, F &&fn)
{
    std::vector<std::thread> thr;
    for (int t = 1; t < nthr; ++t) thr.emplace_back(fn, t);
    fn(0);
    for (auto &th : thr) th.join();
}

static int pool_upload_impl(const char *who, bcfgpu_ctx *ctx, const bcfgpu_reads *rd, const bcfgpu_packed *pk, const uint8_t *r_mapq)
{
    auto fail = [&](int code, const char *what) { char msg[160]; snprintf(msg, sizeof msg, "%s: %s", who, what); return bcfgpu_set_error(code, msg); };
    const bool recs = pk && pk->recs;
    if (!ctx || !rd || rd->n_reads < 0 || (rd->n_reads && !recs && !r_mapq)) return fail(BCFGPU_E_ARG, "bad arguments");
    if (rd->n_reads && ((!recs && (!rd->r_pos || !rd->r_lq || !rd->r_flag || !rd->r_ncig || !rd->r_cig_off || !rd->r_seq_off)) ||
                        (!pk && !rd->seq16) || (!(pk && pk->qual4) && !rd->qual)))
        return fail(BCFGPU_E_ARG, "a read array is missing");
    if (pk && (!pk->seq4 || pk->n_bases < 0 || pk->n_cig < 0 || (pk->n_bases >> 32) || (pk->qual_bits != 0 && pk->qual_bits != 2 && pk->qual_bits != 4)))
        return fail(BCFGPU_E_ARG, "bad packed pool");
    hipStream_t stream = nullptr;
    if (bcfgpu_internal_device(ctx, &stream, nullptr)) return fail(BCFGPU_E_ARG, "bad context");
    DevPool &D = *static_cast<DevPool*>(bcfgpu_internal_pool_state(ctx));
    static_assert(sizeof(DevPool) <= 256, "fits the context's pool_state");
    D = DevPool{};
    const int n = rd->n_reads;
    // the extent of the pools (the packed form states it)
    size_t nbase = 0, ncig = 0;
    if (pk) { nbase = (size_t)pk->n_bases; ncig = (size_t)pk->n_cig; }
    else {
        const int nthr = host_threads(n);
        std::vector<size_t> t_nbase(nthr, 0), t_ncig(nthr, 0);
        on_threads(nthr, [&](int t) {
            const int k0 = (int)((long)n * t / nthr), k1 = (int)((long)n * (t + 1) / nthr);
            size_t nb = 0, nc = 0;
            for (int r = k0; r < k1; ++r) {
                const size_t e = (size_t)rd->r_seq_off[r] + (size_t)std::max(rd->r_lq[r], 0), c = (size_t)rd->r_cig_off[r] + (size_t)std::max(rd->r_ncig[r], 0);
                if (e > nb) nb = e;
                if (c > nc) nc = c;
            }
            t_nbase[t] = nb; t_ncig[t] = nc;
        });
        for (int t = 0; t < nthr; ++t) { nbase = std::max(nbase, t_nbase[t]); ncig = std::max(ncig, t_ncig[t]); }
        if (nbase >> 32) return fail(BCFGPU_E_RANGE, "the pool holds 2^32 or more bases");
    }
    auto up = [&](int slot, const void *src, size_t bytes) -> void* {
        void *d = bcfgpu_internal_ws(ctx, slot, bytes + 64);
        if (d && bytes && hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return nullptr;
        return d;
    };
    D.n_reads = n; D.n_bases = (uint32_t)nbase; D.n_cig = (uint32_t)ncig;
    if (pk && pk->recs) {
        // the 12-byte records: the arrays are formed here, the offsets by two prefix sums
        const bcfgpu_read12 *d_rec = (const bcfgpu_read12*)up(128, pk->recs, (size_t)n * sizeof(bcfgpu_read12));
        int32_t *a_pos = (int32_t*)bcfgpu_internal_ws(ctx, 104, (size_t)n * 4 + 64), *a_lq = (int32_t*)bcfgpu_internal_ws(ctx, 105, (size_t)n * 4 + 64);
        int32_t *a_flag = (int32_t*)bcfgpu_internal_ws(ctx, 106, (size_t)n * 4 + 64), *a_ncig = (int32_t*)bcfgpu_internal_ws(ctx, 107, (size_t)n * 4 + 64);
        int32_t *a_coff = (int32_t*)bcfgpu_internal_ws(ctx, 108, (size_t)(n + 1) * 4 + 64), *a_soff = (int32_t*)bcfgpu_internal_ws(ctx, 109, (size_t)(n + 1) * 4 + 64);
        uint8_t *a_mapq = (uint8_t*)bcfgpu_internal_ws(ctx, 110, (size_t)n + 64);
        if (!d_rec || !a_pos || !a_lq || !a_flag || !a_ncig || !a_coff || !a_soff || !a_mapq) return fail(BCFGPU_E_NOMEM, "device workspace");
        hipLaunchKernelGGL(pool_expand_kernel, dim3((n + 256) / 256), dim3(256), 0, stream, d_rec, n, a_pos, a_lq, a_flag, a_ncig, a_coff, a_soff, a_mapq);
        size_t tmp_bytes = 0;
        if (hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, a_soff, a_soff, n + 1, stream) != hipSuccess) return fail(BCFGPU_E_HIP, "scan");
        void *d_tmp = bcfgpu_internal_ws(ctx, 129, tmp_bytes + 64);
        if (!d_tmp) return fail(BCFGPU_E_NOMEM, "device workspace");
        if (hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, a_soff, a_soff, n + 1, stream) != hipSuccess ||
            hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, a_coff, a_coff, n + 1, stream) != hipSuccess) return fail(BCFGPU_E_HIP, "scan");
        D.r_pos = a_pos; D.r_lq = a_lq; D.r_flag = a_flag; D.r_ncig = a_ncig; D.r_cig_off = a_coff; D.r_seq_off = a_soff; D.r_mapq = a_mapq;
    } else {
    D.r_pos = (const int32_t*)up(104, rd->r_pos, (size_t)n * 4);
    D.r_lq = (const int32_t*)up(105, rd->r_lq, (size_t)n * 4);
    D.r_flag = (const int32_t*)up(106, rd->r_flag, (size_t)n * 4);
    D.r_ncig = (const int32_t*)up(107, rd->r_ncig, (size_t)n * 4);
    D.r_cig_off = (const int32_t*)up(108, rd->r_cig_off, (size_t)n * 4);
    D.r_seq_off = (const int32_t*)up(109, rd->r_seq_off, (size_t)n * 4);
    D.r_mapq = (uint8_t*)up(110, r_mapq, (size_t)n);
    }
    D.cig = (const uint32_t*)up(27, rd->cig, ncig * 4);
    uint8_t *d_seq16 = nullptr, *d_qual = nullptr;
    if (pk) {
        const size_t n_in = (nbase + 1) / 2;
        d_seq16 = (uint8_t*)bcfgpu_internal_ws(ctx, 28, nbase + 64);
        const uint8_t *d_seq4 = (const uint8_t*)up(111, pk->seq4, n_in), *d_qual4 = nullptr;
        if (pk->qual4) { d_qual = (uint8_t*)bcfgpu_internal_ws(ctx, 29, nbase + 64); d_qual4 = (const uint8_t*)up(112, pk->qual4, pk->qual_bits == 2 ? (nbase + 3) / 4 : n_in); }
        else d_qual = (uint8_t*)up(29, rd->qual, nbase);
        if (!d_seq4 || (pk->qual4 && !d_qual4) || !d_seq16 || !d_qual) return fail(BCFGPU_E_NOMEM, "device workspace");
        unsigned long long pal[2] = {0, 0};
        std::memcpy(pal, pk->palette, 16);
        const size_t nthreads = (n_in + 3) / 4;
        if (nthreads) hipLaunchKernelGGL(pileup_unpack_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, stream,
                                         d_seq4, d_qual4, pal[0], pal[1], n_in, d_seq16, d_qual, (int)pk->qual_bits);
    } else {
        d_seq16 = (uint8_t*)up(28, rd->seq16, nbase);
        d_qual = (uint8_t*)up(29, rd->qual, nbase);
    }
    D.seq16 = d_seq16; D.qual = d_qual; D.qual_slot = 29;
    if (!D.r_pos || !D.r_lq || !D.r_flag || !D.r_ncig || !D.r_cig_off || !D.r_seq_off || !D.r_mapq || !D.cig || !D.seq16 || !D.qual)
        return fail(BCFGPU_E_NOMEM, "device workspace");
    if (hipGetLastError() != hipSuccess) return fail(BCFGPU_E_HIP, "launch");
    D.valid = 1;
    return BCFGPU_OK;
}

namespace bcfgpu {
__global__ __launch_bounds__(256) void col_counts_kernel(const uint32_t *off, const uint32_t *col_indel, int n_sites, int S, uint32_t *out)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_sites) return;
    out[2 * k] = off[(size_t)(k + 1) * S] - off[(size_t)k * S];
    out[2 * k + 1] = col_indel[k];
}
}  // namespace bcfgpu

static int pool_pileup_impl(const char *who, bcfgpu_ctx *ctx, const int32_t *r_smpl, const int32_t *given_off, int32_t beg, int32_t end,
                            const char *ref, int32_t ref_len, bcfgpu_tile *tile, int32_t *col_n, uint8_t *col_indel)
{
    auto fail = [&](int code, const char *what) { char msg[160]; snprintf(msg, sizeof msg, "%s: %s", who, what); return bcfgpu_set_error(code, msg); };
    if (!ctx || !tile || end < beg || (ref_len > 0 && !ref)) return fail(BCFGPU_E_ARG, "bad arguments");
    hipStream_t stream = nullptr;
    if (bcfgpu_internal_device(ctx, &stream, nullptr)) return fail(BCFGPU_E_ARG, "bad context");
    const DevPool &D = *static_cast<const DevPool*>(bcfgpu_internal_pool_state(ctx));
    if (!D.valid) return fail(BCFGPU_E_ARG, "no read pool on this context (bcfgpu_pool_upload)");
    if (D.n_reads && !r_smpl && !given_off) return fail(BCFGPU_E_ARG, "bad arguments");
    const bcfgpu_cfg *cfg = bcfgpu_internal_cfg(ctx);
    const int n = D.n_reads, n_sites = end - beg, S = cfg->n_smpl;
    const bool trace = getenv("BCFGPU_TRACE") != nullptr;          // diagnostics: host timeline on stderr
    const auto t_begin = std::chrono::steady_clock::now();
    auto ms_now = [&]() { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    const size_t ncells = (size_t)n_sites * S;
    std::memset(tile, 0, sizeof *tile);
    if (ncells >> 31) return fail(BCFGPU_E_RANGE, "region x samples too large for one tile");
    const int nthr = host_threads(n);
    // ---- host: which reads belong to which sample (usually they come grouped: one file per sample) ----
    std::vector<int32_t> smpl_off(S + 1, 0);
    bool grouped = true;
    if (given_off) {
        for (int s = 0; s <= S; ++s) smpl_off[s] = given_off[s];
        if (smpl_off[0] != 0 || smpl_off[S] != n) return fail(BCFGPU_E_ARG, "smpl_off does not cover the pool");
        for (int s = 0; s < S; ++s) if (smpl_off[s + 1] < smpl_off[s]) return fail(BCFGPU_E_ARG, "smpl_off is not ascending");
    } else {
        std::vector<std::vector<int32_t>> t_cnt(nthr);
        std::vector<int> t_flag(nthr, 0);
        on_threads(nthr, [&](int t) {
            const int k0 = (int)((long)n * t / nthr), k1 = (int)((long)n * (t + 1) / nthr);
            std::vector<int32_t> &c = t_cnt[t];
            c.assign(S, 0);
            int flag = 0;
            for (int r = k0; r < k1; ++r) {
                const int sm = r_smpl[r];
                if (sm < 0 || sm >= S) { flag |= 1; continue; }
                if (r && sm < r_smpl[r - 1]) flag |= 2;
                ++c[sm];
            }
            t_flag[t] = flag;
        });
        for (int t = 0; t < nthr; ++t) {
            if (t_flag[t] & 1) return fail(BCFGPU_E_ARG, "sample index out of range");
            if (t_flag[t] & 2) grouped = false;
            for (int s = 0; s < S; ++s) smpl_off[s + 1] += t_cnt[t][s];
        }
        for (int s = 0; s < S; ++s) smpl_off[s + 1] += smpl_off[s];
    }
    // (grow-only scratch kept per calling thread and page-locked: the upload below is then a DMA transfer the call does not wait for)
    static thread_local struct Scratch {
        void *p = nullptr; size_t bytes = 0; bool pinned = false;
        void drop() { if (p) { if (pinned) hipHostFree(p); else free(p); } p = nullptr; bytes = 0; }
        ~Scratch() { if (p && !pinned) free(p); }                 // (a pinned block outlives the runtime's teardown order: left to the process exit)
    } scratch;
    int32_t *s_read = nullptr;
    if (!grouped) {   // counting sort by sample, the given order kept inside a sample
        const size_t want = (size_t)n * 4 + 64;
        if (scratch.bytes < want) {
            hipStreamSynchronize(stream);                          // an earlier call's upload may still read the old block
            scratch.drop();
            const size_t sz = want + want / 8;
            if (hipHostMalloc(&scratch.p, sz, hipHostMallocDefault) == hipSuccess) scratch.pinned = true;
            else { (void)hipGetLastError(); scratch.p = malloc(sz); scratch.pinned = false; }
            scratch.bytes = scratch.p ? sz : 0;
            if (!scratch.p) return fail(BCFGPU_E_NOMEM, "host scratch");
        }
        s_read = static_cast<int32_t*>(scratch.p);
        std::vector<int32_t> cur(smpl_off.begin(), smpl_off.end() - 1);
        for (int r = 0; r < n; ++r) s_read[cur[r_smpl[r]]++] = r;
    }
    std::vector<int8_t> ref16(n_sites);
    for (int k = 0; k < n_sites; ++k) ref16[k] = (int8_t)(beg + k < ref_len ? ref_nt16(ref[beg + k]) : 15);

    if (trace) fprintf(stderr, "[pileup] host preparation done at %.2f ms\n", ms_now());
    // ---- device ----
    #define PL_CHK(call) do { if ((call) != hipSuccess) return fail(BCFGPU_E_HIP, #call); } while (0)
    auto up = [&](int slot, const void *src, size_t bytes) -> void* {
        void *d = bcfgpu_internal_ws(ctx, slot, bytes + 64);
        if (d && bytes && hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return nullptr;
        return d;
    };
    PileupParams P{};
    P.n_sites = n_sites; P.n_smpl = S; P.beg = beg; P.max_span = 1; P.n_reads = n; P.n_bases = D.n_bases;
    P.want_epos = (cfg->fmt_flag & (BCFGPU_INFO_RPB | BCFGPU_INFO_VDB)) ? 1 : 0;
    void *d_ref16 = up(16, ref16.data(), (size_t)n_sites);
    P.smpl_off = (const int32_t*)up(17, smpl_off.data(), (size_t)(S + 1) * 4);
    MetaParams M{};
    M.n = n; M.n_smpl = S; M.smpl_off = P.smpl_off;
    M.r_pos = D.r_pos; M.r_lq = D.r_lq; M.r_flag = D.r_flag; M.r_ncig = D.r_ncig; M.r_cig_off = D.r_cig_off; M.r_seq_off = D.r_seq_off;
    M.r_mapq = D.r_mapq; M.keep = D.keep; M.cig = D.cig;
    M.s_read = s_read ? (const int32_t*)up(20, s_read, (size_t)n * 4) : nullptr;
    M.meta = (ReadMeta*)bcfgpu_internal_ws(ctx, 19, (size_t)n * sizeof(ReadMeta) + 64);
    M.s_pos = (int32_t*)bcfgpu_internal_ws(ctx, 18, (size_t)n * 4 + 64);
    M.status = (int*)bcfgpu_internal_ws(ctx, 113, 64);
    P.cig = D.cig; P.seq16 = D.seq16; P.qual = D.qual; P.s_read = M.s_read; P.meta = M.meta; P.s_pos = M.s_pos; P.d_span = M.status;
    uint32_t *d_cnt = (uint32_t*)bcfgpu_internal_ws(ctx, 30, (ncells + 1) * 4 + (size_t)n_sites * 4 + 64);
    if (!d_ref16 || !P.smpl_off || (s_read && !M.s_read) || !M.meta || !M.s_pos || !M.status || !d_cnt) return fail(BCFGPU_E_NOMEM, "device workspace");
    P.cnt = d_cnt;
    P.col_indel = d_cnt + ncells + 1;
    PL_CHK(hipMemsetAsync(d_cnt, 0, (ncells + 1) * 4 + (size_t)n_sites * 4, stream));
    const int init_status[2] = {1, 0};
    PL_CHK(hipMemcpyAsync(M.status, init_status, 8, hipMemcpyHostToDevice, stream));
    if (n) hipLaunchKernelGGL(pileup_meta_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, M);
    const int n_work = ((n_sites + PT_COLS - 1) / PT_COLS) * ((S + PT_SMPL - 1) / PT_SMPL);
    const int grid = (n_work + 7) / 8 * 8;           // (a multiple of the XCD count: see the kernel's work mapping)
    uint32_t total = 0;
    int status[2] = {1, 0};
    if (trace) { hipStreamSynchronize(stream); fprintf(stderr, "[pileup] pool on the device at %.2f ms\n", ms_now()); }
    if (ncells) {
        hipLaunchKernelGGL(pileup_kernel<false>, dim3(grid), dim3(256), 0, stream, P);
        // the total in 64 bits first: plp_off is 32-bit, a region whose pileup reaches 2^32 entries must be refused, not wrapped
        unsigned long long *d_tot64 = (unsigned long long*)bcfgpu_internal_ws(ctx, 34, 64);
        if (!d_tot64) return fail(BCFGPU_E_NOMEM, "device workspace");
        PL_CHK(hipMemsetAsync(d_tot64, 0, 8, stream));
        hipLaunchKernelGGL(count_total_kernel, dim3((unsigned)std::min<size_t>((ncells + 4095) / 4096, 1024)), dim3(256), 0, stream, d_cnt, ncells, d_tot64);   // (one atomic per wavefront on one word: few, long-running workgroups)
        // plp_off = exclusive prefix sum of the counts (in place, one element past the end for the total)
        size_t tmp_bytes = 0;
        PL_CHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_cnt, d_cnt, (int)(ncells + 1), stream));
        void *d_tmp = bcfgpu_internal_ws(ctx, 31, tmp_bytes + 16);
        if (!d_tmp) return fail(BCFGPU_E_NOMEM, "device workspace");
        PL_CHK(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_cnt, d_cnt, (int)(ncells + 1), stream));
        unsigned long long tot64 = 0;
        PL_CHK(hipMemcpyAsync(&total, d_cnt + ncells, 4, hipMemcpyDeviceToHost, stream));
        PL_CHK(hipMemcpyAsync(&tot64, d_tot64, 8, hipMemcpyDeviceToHost, stream));
        PL_CHK(hipMemcpyAsync(status, M.status, 8, hipMemcpyDeviceToHost, stream));
        PL_CHK(hipStreamSynchronize(stream));
        if (tot64 >> 32) return fail(BCFGPU_E_RANGE, "the region's pileup has 2^32 or more entries; use a smaller region per call");
    } else {
        PL_CHK(hipMemcpyAsync(status, M.status, 8, hipMemcpyDeviceToHost, stream));
        PL_CHK(hipStreamSynchronize(stream));
    }
    if (status[1] & 1) return fail(BCFGPU_E_RANGE, "a read is longer than 65535 bases or has more than 255 CIGAR operations");
    if (status[1] & 2) return fail(BCFGPU_E_ARG, "the reads of a sample are not in position order");
    P.max_span = status[0];
    if (trace) fprintf(stderr, "[pileup] counted and scanned at %.2f ms (%u entries)\n", ms_now(), total);
    // the read records: the scan's temporary storage is done with, its slot is reused (grow-only) for rd + epos
    const size_t epos_at = (((size_t)total + 4) * 4 + 255) & ~(size_t)255;      // both arrays aligned for 16-byte staging loads
    uint8_t *d_out = (uint8_t*)bcfgpu_internal_ws(ctx, 31, epos_at + (size_t)total + 64);
    if (!d_out) return fail(BCFGPU_E_NOMEM, "device workspace");
    P.rd = (uint32_t*)d_out; P.epos = d_out + epos_at;
    if (total) hipLaunchKernelGGL(pileup_kernel<true>, dim3(grid), dim3(256), 0, stream, P);
    PL_CHK(hipGetLastError());
    if (trace) { hipStreamSynchronize(stream); fprintf(stderr, "[pileup] tile filled at %.2f ms\n", ms_now()); }
    if ((col_n || col_indel) && n_sites) {
        // per column: its entries (the difference of two offsets, formed on the device: the offsets themselves are 4 bytes a cell and
        // stay in HBM) and whether any is followed by an indel; two words a column come back, through page-locked memory
        uint32_t *d_cc = (uint32_t*)bcfgpu_internal_ws(ctx, 134, (size_t)n_sites * 8 + 64);
        uint32_t *h_cc = (uint32_t*)bcfgpu_internal_pinned(ctx, 3, (size_t)n_sites * 8 + 64);
        if (!d_cc || !h_cc) return fail(BCFGPU_E_NOMEM, "device workspace");
        hipLaunchKernelGGL(col_counts_kernel, dim3((n_sites + 255) / 256), dim3(256), 0, stream, d_cnt, P.col_indel, n_sites, S, d_cc);
        PL_CHK(hipMemcpyAsync(h_cc, d_cc, (size_t)n_sites * 8, hipMemcpyDeviceToHost, stream));
        PL_CHK(hipStreamSynchronize(stream));
        for (int k = 0; k < n_sites; ++k) {
            if (col_n) col_n[k] = (int32_t)h_cc[2 * k];
            if (col_indel) col_indel[k] = h_cc[2 * k + 1] ? 1 : 0;
        }
    }
    #undef PL_CHK
    static_assert(sizeof(PileupParams) <= 256, "fits the context's pileup_state");
    P.ref_len = ref_len;
    std::memcpy(bcfgpu_internal_pileup_state(ctx), &P, sizeof P);          // for bcfgpu_pileup_entries
    tile->n_sites = n_sites; tile->is_indel = 0; tile->n_reads = total;
    tile->ref16 = (const int8_t*)d_ref16; tile->plp_off = d_cnt; tile->rd = P.rd; tile->epos = P.epos;
    return BCFGPU_OK;
}

extern "C" int bcfgpu_pileup(bcfgpu_ctx *ctx, const bcfgpu_reads *rd, const uint8_t *r_mapq, const int32_t *r_smpl,
                             int32_t beg, int32_t end, const char *ref, int32_t ref_len,
                             bcfgpu_tile *tile, int32_t *col_n, uint8_t *col_indel)
{
    if (!rd || !tile || (rd->n_reads > 0 && !r_smpl)) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pileup: bad arguments");
    const int rc = pool_upload_impl("bcfgpu_pileup", ctx, rd, nullptr, r_mapq);
    return rc ? rc : pool_pileup_impl("bcfgpu_pileup", ctx, r_smpl, nullptr, beg, end, ref, ref_len, tile, col_n, col_indel);
}

extern "C" int bcfgpu_pileup_packed(bcfgpu_ctx *ctx, const bcfgpu_reads *rd, const bcfgpu_packed *pk, const uint8_t *r_mapq,
                                    const int32_t *r_smpl, int32_t beg, int32_t end, const char *ref, int32_t ref_len,
                                    bcfgpu_tile *tile, int32_t *col_n, uint8_t *col_indel)
{
    if (!rd || !pk || !tile || (rd->n_reads > 0 && !r_smpl && !pk->smpl_off)) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pileup_packed: bad arguments");
    const int rc = pool_upload_impl("bcfgpu_pileup_packed", ctx, rd, pk, r_mapq);
    return rc ? rc : pool_pileup_impl("bcfgpu_pileup_packed", ctx, r_smpl, pk->smpl_off, beg, end, ref, ref_len, tile, col_n, col_indel);
}

// ---- the pool as an object of its own: upload once, run the stages on it, build the tile ----
// [lowest start, highest end) of the pool's reads (one small kernel and a wait, once per pool)
int bcfgpu_internal_pool_extent(bcfgpu_ctx *ctx, int *lo, int *hi)
{
    hipStream_t stream = nullptr;
    if (bcfgpu_internal_device(ctx, &stream, nullptr)) return BCFGPU_E_ARG;
    DevPool &D = *static_cast<DevPool*>(bcfgpu_internal_pool_state(ctx));
    if (!D.valid) return BCFGPU_E_ARG;
    if (!D.ext_valid) {
        int *d = (int*)bcfgpu_internal_ws(ctx, 122, 64);
        if (!d) return BCFGPU_E_NOMEM;
        int v[2] = {INT32_MAX, 0};
        if (hipMemcpyAsync(d, v, 8, hipMemcpyHostToDevice, stream) != hipSuccess) return BCFGPU_E_HIP;
        if (D.n_reads) hipLaunchKernelGGL(pool_extent_kernel, dim3((D.n_reads + 255) / 256), dim3(256), 0, stream, D, d);
        if (hipMemcpyAsync(v, d, 8, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return BCFGPU_E_HIP;
        if (v[0] == INT32_MAX) v[0] = 0;
        D.ext_lo = v[0]; D.ext_hi = v[1] > v[0] ? v[1] : v[0]; D.ext_valid = 1;
    }
    *lo = D.ext_lo; *hi = D.ext_hi;
    return BCFGPU_OK;
}

extern "C" int bcfgpu_pool_upload(bcfgpu_ctx *ctx, const bcfgpu_reads *rd, const bcfgpu_packed *pk, const uint8_t *r_mapq)
{
    const int rc = pool_upload_impl("bcfgpu_pool_upload", ctx, rd, pk, r_mapq);
    if (rc) return rc;
    // the arrays are the caller's: they are free again when this call returns (bcfgpu_pileup[_packed] wait later, with the counts)
    hipStream_t stream = nullptr;
    bcfgpu_internal_device(ctx, &stream, nullptr);
    if (hipStreamSynchronize(stream) != hipSuccess) return bcfgpu_set_error(BCFGPU_E_HIP, "bcfgpu_pool_upload: upload");
    return BCFGPU_OK;
}

extern "C" int bcfgpu_pool_keep(bcfgpu_ctx *ctx, const uint8_t *keep)
{
    hipStream_t stream = nullptr;
    if (!ctx || bcfgpu_internal_device(ctx, &stream, nullptr)) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pool_keep: bad context");
    DevPool &D = *static_cast<DevPool*>(bcfgpu_internal_pool_state(ctx));
    if (!D.valid) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pool_keep: no read pool on this context (bcfgpu_pool_upload)");
    if (!keep) { D.keep = nullptr; return BCFGPU_OK; }
    uint8_t *d = (uint8_t*)bcfgpu_internal_ws(ctx, 114, (size_t)D.n_reads + 64);
    if (!d) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_pool_keep: device workspace");
    if (D.n_reads && hipMemcpyAsync(d, keep, (size_t)D.n_reads, hipMemcpyHostToDevice, stream) != hipSuccess)
        return bcfgpu_set_error(BCFGPU_E_HIP, "bcfgpu_pool_keep: upload");
    if (hipStreamSynchronize(stream) != hipSuccess) return bcfgpu_set_error(BCFGPU_E_HIP, "bcfgpu_pool_keep: upload");   // (keep may be the caller's stack)
    D.keep = d;
    return BCFGPU_OK;
}

extern "C" int bcfgpu_pool_pileup(bcfgpu_ctx *ctx, const int32_t *r_smpl, const int32_t *smpl_off, int32_t beg, int32_t end,
                                  const char *ref, int32_t ref_len, bcfgpu_tile *tile, int32_t *col_n, uint8_t *col_indel)
{
    return pool_pileup_impl("bcfgpu_pool_pileup", ctx, r_smpl, smpl_off, beg, end, ref, ref_len, tile, col_n, col_indel);
}

extern "C" int bcfgpu_pool_download(bcfgpu_ctx *ctx, uint8_t *qual, uint8_t *zq, uint8_t *r_mapq)
{
    hipStream_t stream = nullptr;
    if (!ctx || bcfgpu_internal_device(ctx, &stream, nullptr)) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pool_download: bad context");
    const DevPool &D = *static_cast<const DevPool*>(bcfgpu_internal_pool_state(ctx));
    if (!D.valid) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pool_download: no read pool on this context (bcfgpu_pool_upload)");
    if (qual && D.n_bases && hipMemcpyAsync(qual, D.qual, D.n_bases, hipMemcpyDeviceToHost, stream) != hipSuccess) return bcfgpu_set_error(BCFGPU_E_HIP, "bcfgpu_pool_download");
    if (zq && D.n_bases) {
        if (D.zq) { if (hipMemcpyAsync(zq, D.zq, D.n_bases, hipMemcpyDeviceToHost, stream) != hipSuccess) return bcfgpu_set_error(BCFGPU_E_HIP, "bcfgpu_pool_download"); }
        else std::memset(zq, 0, D.n_bases);
    }
    if (r_mapq && D.n_reads && hipMemcpyAsync(r_mapq, D.r_mapq, (size_t)D.n_reads, hipMemcpyDeviceToHost, stream) != hipSuccess) return bcfgpu_set_error(BCFGPU_E_HIP, "bcfgpu_pool_download");
    if (hipStreamSynchronize(stream) != hipSuccess) return bcfgpu_set_error(BCFGPU_E_HIP, "bcfgpu_pool_download");
    return BCFGPU_OK;
}

extern "C" int bcfgpu_pileup_entries(bcfgpu_ctx *ctx, int32_t n_cols, const int32_t *cols, int32_t *smpl_off,
                                     int32_t *p_read, int32_t *p_qpos, int32_t *p_indel, int64_t cap)
{
    if (!ctx || n_cols < 0 || (n_cols && (!cols || !smpl_off)))
        return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pileup_entries: bad arguments");
    hipStream_t stream = nullptr;
    if (bcfgpu_internal_device(ctx, &stream, nullptr)) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pileup_entries: bad context");
    EntriesParams E{};
    std::memcpy(&E.P, bcfgpu_internal_pileup_state(ctx), sizeof E.P);
    if (!E.P.cnt) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pileup_entries: no bcfgpu_pileup on this context yet");
    if (n_cols == 0) return BCFGPU_OK;
    const int S = E.P.n_smpl;
    for (int i = 0; i < n_cols; ++i)
        if (cols[i] < 0 || cols[i] >= E.P.n_sites) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pileup_entries: column out of range");
    const size_t nsel = (size_t)n_cols * S;
    #define PE_CHK(call) do { if ((call) != hipSuccess) return bcfgpu_set_error(BCFGPU_E_HIP, #call); } while (0)
    int32_t *d_cols = (int32_t*)bcfgpu_internal_ws(ctx, 21, (size_t)n_cols * 4 + 16);
    uint32_t *d_sel = (uint32_t*)bcfgpu_internal_ws(ctx, 22, (nsel + 1) * 4 + 16);
    if (!d_cols || !d_sel) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_pileup_entries: device workspace");
    PE_CHK(hipMemcpyAsync(d_cols, cols, (size_t)n_cols * 4, hipMemcpyHostToDevice, stream));
    PE_CHK(hipMemsetAsync(d_sel, 0, (nsel + 1) * 4, stream));
    E.n_cols = n_cols; E.cols = d_cols; E.sel_cnt = d_sel;
    const int grid = (int)((nsel + 255) / 256);
    hipLaunchKernelGGL(entries_kernel<false>, dim3(grid), dim3(256), 0, stream, E);
    size_t tmp_bytes = 0;
    PE_CHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_sel, d_sel, (int)(nsel + 1), stream));
    void *d_tmp = bcfgpu_internal_ws(ctx, 23, tmp_bytes + 16);
    if (!d_tmp) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_pileup_entries: device workspace");
    PE_CHK(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_sel, d_sel, (int)(nsel + 1), stream));
    std::vector<uint32_t> off(nsel + 1);
    PE_CHK(hipMemcpyAsync(off.data(), d_sel, (nsel + 1) * 4, hipMemcpyDeviceToHost, stream));
    PE_CHK(hipStreamSynchronize(stream));
    const uint32_t total = off[nsel];
    for (size_t i = 0; i <= nsel; ++i) smpl_off[i] = (int32_t)off[i];
    if ((int64_t)total > cap || (total && (!p_read || !p_qpos || !p_indel)))
        return bcfgpu_set_error(BCFGPU_E_RANGE, "bcfgpu_pileup_entries: the output arrays are too small (sum of col_n over the columns)");
    if (total) {
        int32_t *d_e = (int32_t*)bcfgpu_internal_ws(ctx, 24, (size_t)total * 12 + 16);
        if (!d_e) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_pileup_entries: device workspace");
        E.e_read = d_e; E.e_qpos = d_e + total; E.e_indel = d_e + 2 * (size_t)total;
        hipLaunchKernelGGL(entries_kernel<true>, dim3((unsigned)((nsel + 3) / 4)), dim3(256), 0, stream, E);
        PE_CHK(hipGetLastError());
        PE_CHK(hipMemcpyAsync(p_read, E.e_read, (size_t)total * 4, hipMemcpyDeviceToHost, stream));
        PE_CHK(hipMemcpyAsync(p_qpos, E.e_qpos, (size_t)total * 4, hipMemcpyDeviceToHost, stream));
        PE_CHK(hipMemcpyAsync(p_indel, E.e_indel, (size_t)total * 4, hipMemcpyDeviceToHost, stream));
        PE_CHK(hipStreamSynchronize(stream));
    }
    #undef PE_CHK
    return BCFGPU_OK;
}

extern "C" int bcfgpu_pileup_indel_tile(bcfgpu_ctx *ctx, int32_t n_cols, const int32_t *cols, const uint32_t *aux, int64_t n_aux,
                                        bcfgpu_tile *tile)
{
    if (!ctx || n_cols < 0 || (n_cols && !cols) || !tile || n_aux < 0 || (n_aux && !aux))
        return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pileup_indel_tile: bad arguments");
    hipStream_t stream = nullptr;
    if (bcfgpu_internal_device(ctx, &stream, nullptr)) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pileup_indel_tile: bad context");
    EntriesParams E{};
    std::memcpy(&E.P, bcfgpu_internal_pileup_state(ctx), sizeof E.P);
    if (!E.P.cnt) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pileup_indel_tile: no bcfgpu_pileup on this context yet");
    std::memset(tile, 0, sizeof *tile);
    const int S = E.P.n_smpl;
    for (int i = 0; i < n_cols; ++i)
        if (cols[i] < 0 || cols[i] >= E.P.n_sites) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pileup_indel_tile: column out of range");
    const size_t nsel = (size_t)n_cols * S;
    #define PT_CHK(call) do { if ((call) != hipSuccess) return bcfgpu_set_error(BCFGPU_E_HIP, #call); } while (0)
    int32_t *d_cols = (int32_t*)bcfgpu_internal_ws(ctx, 21, (size_t)n_cols * 4 + 16);
    uint32_t *d_sel = (uint32_t*)bcfgpu_internal_ws(ctx, 25, (nsel + 1) * 4 + (size_t)n_cols + 64);
    if (!d_cols || !d_sel) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_pileup_indel_tile: device workspace");
    PT_CHK(hipMemcpyAsync(d_cols, cols, (size_t)n_cols * 4, hipMemcpyHostToDevice, stream));
    PT_CHK(hipMemsetAsync(d_sel, 0, (nsel + 1) * 4 + (size_t)n_cols + 64, stream));
    E.n_cols = n_cols; E.cols = d_cols; E.sel_cnt = d_sel;
    const int grid = (int)((nsel + 255) / 256);
    uint32_t total = 0;
    if (nsel) {
        hipLaunchKernelGGL(subtile_kernel<false>, dim3(grid), dim3(256), 0, stream, E, (uint32_t*)nullptr, (uint8_t*)nullptr);
        size_t tmp_bytes = 0;
        PT_CHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_sel, d_sel, (int)(nsel + 1), stream));
        void *d_tmp = bcfgpu_internal_ws(ctx, 23, tmp_bytes + 16);
        if (!d_tmp) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_pileup_indel_tile: device workspace");
        PT_CHK(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_sel, d_sel, (int)(nsel + 1), stream));
        PT_CHK(hipMemcpyAsync(&total, d_sel + nsel, 4, hipMemcpyDeviceToHost, stream));
        PT_CHK(hipStreamSynchronize(stream));
    }
    if ((int64_t)total != n_aux) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_pileup_indel_tile: aux must hold one word per entry of the columns");
    const size_t ep_at = (((size_t)total + 4) * 4 + 255) & ~(size_t)255, aux_at = (ep_at + total + 64 + 255) & ~(size_t)255;
    uint8_t *d_out = (uint8_t*)bcfgpu_internal_ws(ctx, 26, aux_at + ((size_t)total + 4) * 4);
    if (!d_out) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_pileup_indel_tile: device workspace");
    if (total) {
        hipLaunchKernelGGL(subtile_kernel<true>, dim3(grid), dim3(256), 0, stream, E, (uint32_t*)d_out, d_out + ep_at);
        PT_CHK(hipGetLastError());
        PT_CHK(hipMemcpyAsync(d_out + aux_at, aux, (size_t)total * 4, hipMemcpyHostToDevice, stream));
        PT_CHK(hipStreamSynchronize(stream));
    }
    #undef PT_CHK
    tile->n_sites = n_cols; tile->is_indel = 1; tile->n_reads = total;
    tile->ref16 = reinterpret_cast<const int8_t*>(d_sel + nsel + 1);       // (zeros: the indel pass does not read it)
    tile->plp_off = d_sel; tile->rd = (const uint32_t*)d_out; tile->epos = d_out + ep_at; tile->aux = (const uint32_t*)(d_out + aux_at);
    return BCFGPU_OK;
}

// ---- bcf_call_gap_prep for candidate columns of the last bcfgpu_pileup, everything staying in HBM -------------------
namespace bcfgpu {
// the per-read arrays bcfgpu_gap_prep takes (bcfgpu_reads), rebuilt from the pileup's read records; by pool index
__global__ __launch_bounds__(256) void gap_unpack_reads_kernel(const PileupParams P, int32_t *r_pos, int32_t *r_lq, int32_t *r_flag,
                                                               int32_t *r_ncig, int32_t *r_cig_off, int32_t *r_seq_off)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= P.n_reads) return;
    const ReadMeta m = P.meta[k];
    const int r = P.s_read ? P.s_read[k] : k;
    r_pos[r] = m.pos; r_lq[r] = m.lq; r_flag[r] = ((m.bits & 1) ? 16 : 0) | ((m.bits & 4) ? 4 : 0);
    r_ncig[r] = m.ncig; r_cig_off[r] = (int32_t)m.cig_off; r_seq_off[r] = (int32_t)m.seq_off;
}
__global__ __launch_bounds__(256) void gap_col_pos_kernel(const int32_t *cols, int n_cols, int beg, int32_t *pos)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_cols) pos[i] = beg + cols[i];
}
}  // namespace bcfgpu

int bcfgpu_internal_gap_core(bcfgpu_ctx *ctx, const GapIn &g, size_t n_ent, uint32_t *d_aux, const bcfgpu_indel_out *out, int inscns_cap);
extern "C" bcfgpu_gap_stats *bcfgpu_internal_gap_stats(bcfgpu_ctx *ctx);

extern "C" int bcfgpu_gap_prep_tile(bcfgpu_ctx *ctx, int32_t n_cols, const int32_t *cols, const bcfgpu_reads *reads,
                                    const bcfgpu_indel_in *par, const bcfgpu_indel_out *out, int inscns_cap, bcfgpu_tile *tile)
{
    if (!ctx || n_cols < 0 || (n_cols && !cols) || !par || !par->ref || !out || !out->ret || !out->indel_types || !tile || inscns_cap < 0)
        return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_gap_prep_tile: bad arguments");
    hipStream_t stream = nullptr;
    if (bcfgpu_internal_device(ctx, &stream, nullptr)) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_gap_prep_tile: bad context");
    EntriesParams E{};
    std::memcpy(&E.P, bcfgpu_internal_pileup_state(ctx), sizeof E.P);
    const PileupParams &P = E.P;
    if (!P.cnt) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_gap_prep_tile: no bcfgpu_pileup on this context yet");
    std::memset(tile, 0, sizeof *tile);
    tile->is_indel = 1;
    bcfgpu_gap_stats &gs = *bcfgpu_internal_gap_stats(ctx);
    gs = bcfgpu_gap_stats{};
    if (n_cols == 0) return BCFGPU_OK;
    const auto t_begin = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t0) {
        return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    const int S = P.n_smpl, nr = P.n_reads;
    for (int i = 0; i < n_cols; ++i)
        if (cols[i] < 0 || cols[i] >= P.n_sites) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_gap_prep_tile: column out of range");
    // every column starts as "bcf_call_gap_prep returned -1": the columns the stage goes on with overwrite their rows below
    for (int i = 0; i < n_cols; ++i) { out->ret[i] = -1; for (int t = 0; t < 4; ++t) out->indel_types[(size_t)i * 4 + t] = 10000; }
    if (out->inscns) std::memset(out->inscns, 0, (size_t)n_cols * 4 * inscns_cap);
    if (out->maxins) std::memset(out->maxins, 0, (size_t)n_cols * 4);
    if (out->indelreg) std::memset(out->indelreg, 0, (size_t)n_cols * 4);
    if (out->max_support) std::memset(out->max_support, 0, (size_t)n_cols * 4);
    if (out->max_frac) std::memset(out->max_frac, 0, (size_t)n_cols * 4);
    #define GT_CHK(call) do { if ((call) != hipSuccess) return bcfgpu_set_error(BCFGPU_E_HIP, #call); } while (0)
    #define GWS(slot, bytes) bcfgpu_internal_ws(ctx, 40 + (slot), (bytes) + 64)      /* the slots bcfgpu_gap_prep uses for its uploads */
    // ---- the columns the stage goes on with: those that pass the pooled support filter (bam2bcf_indel.c:150-154; every column with
    // per_sample_flt), compacted on the device.  Nothing below touches a column that fails: no entry of it is listed, no workgroup
    // of the stage is started for it. ----
    int32_t *d_cols = (int32_t*)bcfgpu_internal_ws(ctx, 21, (size_t)n_cols * 12 + 64);     // cols, then the kept columns, then their places in cols
    int32_t *d_nk = (int32_t*)bcfgpu_internal_ws(ctx, 143, 64);
    int32_t *h_k = (int32_t*)bcfgpu_internal_pinned(ctx, 2, (size_t)n_cols * 4 + 64);    // [0] the count, [16..] the places
    if (!d_cols || !d_nk || !h_k) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_gap_prep_tile: device workspace");
    int32_t *d_kcols = d_cols + n_cols, *d_kidx = d_cols + 2 * (size_t)n_cols;
    GT_CHK(hipMemcpyAsync(d_cols, cols, (size_t)n_cols * 4, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(gap_keep_kernel, dim3(1), dim3(1024), 0, stream, P, d_cols, n_cols, (!par->per_sample_flt && P.col_indel) ? 1 : 0,
                       par->min_support, par->min_frac, d_kcols, d_kidx, d_nk);
    GT_CHK(hipMemcpyAsync(h_k, d_nk, 4, hipMemcpyDeviceToHost, stream));
    GT_CHK(hipMemcpyAsync(h_k + 16, d_kidx, (size_t)n_cols * 4, hipMemcpyDeviceToHost, stream));
    // meanwhile: the reads as bcfgpu_gap_prep's kernels index them
    int32_t *d_rpos = (int32_t*)GWS(0, (size_t)nr * 4), *d_rlq = (int32_t*)GWS(1, (size_t)nr * 4), *d_rflag = (int32_t*)GWS(2, (size_t)nr * 4);
    int32_t *d_rncig = (int32_t*)GWS(3, (size_t)nr * 4), *d_rcoff = (int32_t*)GWS(4, (size_t)nr * 4), *d_rsoff = (int32_t*)GWS(5, (size_t)nr * 4);
    if (!d_rpos || !d_rlq || !d_rflag || !d_rncig || !d_rcoff || !d_rsoff) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_gap_prep_tile: device workspace");
    if (nr) hipLaunchKernelGGL(gap_unpack_reads_kernel, dim3((nr + 255) / 256), dim3(256), 0, stream, P, d_rpos, d_rlq, d_rflag, d_rncig, d_rcoff, d_rsoff);
    GT_CHK(hipStreamSynchronize(stream));                       // how many columns go on
    const int nk = h_k[0];
    std::vector<int32_t> kidx(h_k + 16, h_k + 16 + nk);
    gs.prepare_ms = ms_since(t_begin);
    if (nk == 0) { gs.total_ms = ms_since(t_begin); return BCFGPU_OK; }
    const size_t nsel = (size_t)nk * S;
    // ---- their pileup entries (read, query offset, indel after the position), on the device ----
    uint32_t *d_sel = (uint32_t*)bcfgpu_internal_ws(ctx, 25, (nsel + 1) * 4 + 64);
    if (!d_sel) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_gap_prep_tile: device workspace");
    GT_CHK(hipMemsetAsync(d_sel, 0, (nsel + 1) * 4, stream));
    E.n_cols = nk; E.cols = d_kcols; E.sel_cnt = d_sel; E.col_keep = nullptr;
    const int grid = (int)((nsel + 255) / 256);
    hipLaunchKernelGGL(entries_kernel<false>, dim3(grid), dim3(256), 0, stream, E);
    size_t tmp_bytes = 0;
    GT_CHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_sel, d_sel, (int)(nsel + 1), stream));
    void *d_tmp = bcfgpu_internal_ws(ctx, 23, tmp_bytes + 16);
    if (!d_tmp) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_gap_prep_tile: device workspace");
    GT_CHK(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_sel, d_sel, (int)(nsel + 1), stream));
    uint32_t total = 0;
    GT_CHK(hipMemcpyAsync(&total, d_sel + nsel, 4, hipMemcpyDeviceToHost, stream));
    // meanwhile: the columns' positions, the reference slice, ZQ
    int cmin = INT32_MAX, cmax = 0;
    for (int j = 0; j < nk; ++j) { cmin = std::min(cmin, cols[kidx[j]]); cmax = std::max(cmax, cols[kidx[j]]); }
    int32_t *d_pos = (int32_t*)GWS(11, (size_t)nk * 4);
    const int pmin = P.beg + cmin, pmax = P.beg + cmax;
    // The slice of the contig the batch touches.  The contig ends where bcfgpu_pileup was told it ends (ref_len): a candidate
    // column at or past that end reads nothing of par->ref (bcfgpu_indel_in carries no length of its own).
    const long ref_end = P.ref_len > 0 ? (long)P.ref_len : 0;
    const long ref_lo = std::min<long>(pmin > 65536 ? pmin - 65536 : 0, ref_end);
    const long ref_from = std::min<long>((long)pmax + 1, ref_end);
    const long ref_hi = ref_from + (long)strnlen(par->ref + ref_from, (size_t)std::min<long>(65536 + 4096, ref_end - ref_from));
    char *d_ref = (char*)GWS(24, (size_t)(ref_hi - ref_lo));
    const bool any_zq = reads && reads->zq && reads->r_has_zq;
    uint8_t *d_zq = any_zq ? (uint8_t*)GWS(9, (size_t)P.n_bases) : nullptr, *d_haszq = any_zq ? (uint8_t*)GWS(10, (size_t)nr) : nullptr;
    {   // no ZQ bytes from the host: those bcfgpu_pool_baq left in HBM, if the pool of the pileup is still the context's pool
        const DevPool &D = *static_cast<const DevPool*>(bcfgpu_internal_pool_state(ctx));
        if (!any_zq && D.valid && D.zq && D.r_has_zq && D.n_reads == nr && D.seq16 == P.seq16) { d_zq = D.zq; d_haszq = D.r_has_zq; }
    }
    if (!d_pos || !d_ref || (any_zq && (!d_zq || !d_haszq))) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_gap_prep_tile: device workspace");
    if (any_zq && reads->n_reads != nr) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_gap_prep_tile: `reads` is not the pool of the last bcfgpu_pileup");
    hipLaunchKernelGGL(gap_col_pos_kernel, dim3((nk + 255) / 256), dim3(256), 0, stream, d_kcols, nk, P.beg, d_pos);
    if (ref_hi > ref_lo) GT_CHK(hipMemcpyAsync(d_ref, par->ref + ref_lo, (size_t)(ref_hi - ref_lo), hipMemcpyHostToDevice, stream));
    if (any_zq) {
        GT_CHK(hipMemcpyAsync(d_zq, reads->zq, (size_t)P.n_bases, hipMemcpyHostToDevice, stream));
        GT_CHK(hipMemcpyAsync(d_haszq, reads->r_has_zq, (size_t)nr, hipMemcpyHostToDevice, stream));
    }
    GT_CHK(hipStreamSynchronize(stream));                       // the entry count sizes everything that follows
    if ((total >> 31) != 0) return bcfgpu_set_error(BCFGPU_E_RANGE, "bcfgpu_gap_prep_tile: too many pileup entries in one call, use fewer columns");
    int32_t *d_e = (int32_t*)bcfgpu_internal_ws(ctx, 24, (size_t)total * 12 + 16);
    uint32_t *d_aux = (uint32_t*)GWS(27, ((size_t)total + 4) * 4);
    if (!d_e || !d_aux) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_gap_prep_tile: device workspace");
    E.e_read = d_e; E.e_qpos = d_e + total; E.e_indel = d_e + 2 * (size_t)total;
    if (total) hipLaunchKernelGGL(entries_kernel<true>, dim3((unsigned)((nsel + 3) / 4)), dim3(256), 0, stream, E);
    GT_CHK(hipGetLastError());
    GapIn g{};
    g.n_sites = nk; g.n_smpl = S; g.n_reads = nr;
    g.pos = d_pos; g.smpl_off = reinterpret_cast<const int32_t*>(d_sel); g.p_read = E.e_read; g.p_qpos = E.e_qpos; g.p_indel = E.e_indel;
    g.r_pos = d_rpos; g.r_lq = d_rlq; g.r_flag = d_rflag; g.r_ncig = d_rncig; g.r_cig_off = d_rcoff; g.r_seq_off = d_rsoff;
    g.cig = P.cig; g.seq16 = P.seq16; g.qual = P.qual; g.zq = d_zq; g.r_has_zq = d_haszq;
    g.ref = d_ref; g.ref_lo = ref_lo; g.ref_hi = ref_hi;
    g.openQ = par->openQ; g.extQ = par->extQ; g.tandemQ = par->tandemQ; g.min_support = par->min_support; g.per_sample_flt = par->per_sample_flt;
    g.min_frac = par->min_frac;
    gs.prepare_ms = ms_since(t_begin);
    // the stage's per-column results, by kept column; scattered to the caller's rows afterwards
    std::vector<int32_t> k_ret(nk), k_types((size_t)nk * 4), k_maxins(nk), k_ireg(nk), k_msup(nk);
    std::vector<float> k_mfrac(nk);
    std::vector<int8_t> k_inscns(out->inscns ? (size_t)nk * 4 * inscns_cap : 0);
    bcfgpu_indel_out ko{};
    ko.ret = k_ret.data(); ko.indel_types = k_types.data(); ko.maxins = k_maxins.data(); ko.indelreg = k_ireg.data();
    ko.max_support = k_msup.data(); ko.max_frac = k_mfrac.data(); ko.inscns = out->inscns ? k_inscns.data() : nullptr;
    const int rc = bcfgpu_internal_gap_core(ctx, g, (size_t)total, d_aux, &ko, inscns_cap);
    if (rc) return rc;
    std::vector<int32_t> lk, lcols;
    for (int j = 0; j < nk; ++j) {
        const size_t i = (size_t)kidx[j];
        out->ret[i] = k_ret[j];
        std::memcpy(out->indel_types + i * 4, &k_types[(size_t)j * 4], 16);
        if (out->inscns && inscns_cap) std::memcpy(out->inscns + i * 4 * inscns_cap, &k_inscns[(size_t)j * 4 * inscns_cap], (size_t)4 * inscns_cap);
        if (out->maxins) out->maxins[i] = k_maxins[j];
        if (out->indelreg) out->indelreg[i] = k_ireg[j];
        if (out->max_support) out->max_support[i] = k_msup[j];
        if (out->max_frac) out->max_frac[i] = k_mfrac[j];
        if (k_ret[j] == 0) { lk.push_back(j); lcols.push_back(cols[i]); }
    }
    // ---- the indel pass's tile: the columns with ret == 0 (mpileup.c:354-360), their records from the pileup, p->aux from the stage ----
    const int nl = (int)lk.size();
    if (nl == 0) { gs.total_ms = ms_since(t_begin); return BCFGPU_OK; }
    const size_t nlsel = (size_t)nl * S;
    int32_t *d_l = (int32_t*)bcfgpu_internal_ws(ctx, 130, (size_t)nl * 8 + 64);
    uint32_t *d_lsel = (uint32_t*)bcfgpu_internal_ws(ctx, 131, (nlsel + 1) * 4 + (size_t)nl + 64);
    if (!d_l || !d_lsel) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_gap_prep_tile: device workspace");
    GT_CHK(hipMemcpyAsync(d_l, lk.data(), (size_t)nl * 4, hipMemcpyHostToDevice, stream));
    GT_CHK(hipMemcpyAsync(d_l + nl, lcols.data(), (size_t)nl * 4, hipMemcpyHostToDevice, stream));
    GT_CHK(hipMemsetAsync(d_lsel, 0, (nlsel + 1) * 4 + (size_t)nl + 64, stream));
    const int lgrid = (int)((nlsel + 255) / 256);
    hipLaunchKernelGGL(live_tile_kernel<true>, dim3(lgrid), dim3(256), 0, stream, P, nl, d_l + nl, d_l, d_sel, d_lsel,
                       (const uint32_t*)nullptr, (uint32_t*)nullptr, (uint8_t*)nullptr, (uint32_t*)nullptr);
    GT_CHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_lsel, d_lsel, (int)(nlsel + 1), stream));
    d_tmp = bcfgpu_internal_ws(ctx, 23, tmp_bytes + 16);
    if (!d_tmp) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_gap_prep_tile: device workspace");
    GT_CHK(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_lsel, d_lsel, (int)(nlsel + 1), stream));
    uint32_t ltotal = 0;
    GT_CHK(hipMemcpyAsync(&ltotal, d_lsel + nlsel, 4, hipMemcpyDeviceToHost, stream));
    GT_CHK(hipStreamSynchronize(stream));                       // (lk / lcols are this call's host vectors)
    const size_t ep_at = (((size_t)ltotal + 4) * 4 + 255) & ~(size_t)255, aux_at = (ep_at + ltotal + 64 + 255) & ~(size_t)255;
    uint8_t *d_out = (uint8_t*)bcfgpu_internal_ws(ctx, 26, aux_at + ((size_t)ltotal + 4) * 4);
    if (!d_out) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_gap_prep_tile: device workspace");
    if (ltotal) hipLaunchKernelGGL(live_tile_kernel<false>, dim3(32, nl < 65535 ? nl : 65535), dim3(256), 0, stream, P, nl, d_l + nl, d_l, d_sel, d_lsel,
                                   (const uint32_t*)d_aux, (uint32_t*)d_out, d_out + ep_at, (uint32_t*)(d_out + aux_at));
    GT_CHK(hipGetLastError());
    if (out->p_aux && ltotal) {                                 // optional: the tile's p->aux words for the caller as well
        GT_CHK(hipMemcpyAsync(out->p_aux, d_out + aux_at, (size_t)ltotal * 4, hipMemcpyDeviceToHost, stream));
        GT_CHK(hipStreamSynchronize(stream));
    }
    gs.total_ms = ms_since(t_begin);
    #undef GT_CHK
    #undef GWS
    tile->n_sites = nl; tile->is_indel = 1; tile->n_reads = ltotal;
    tile->ref16 = reinterpret_cast<const int8_t*>(d_lsel + nlsel + 1);     // (zeros: the indel pass does not read it)
    tile->plp_off = d_lsel; tile->rd = (const uint32_t*)d_out; tile->epos = d_out + ep_at; tile->aux = (const uint32_t*)(d_out + aux_at);
    return BCFGPU_OK;
}

// gap_prep.hip -- bcfgpu_gap_prep: bcf_call_gap_prep (bam2bcf_indel.c:99-470) for a batch of candidate columns, every
// stage on the device.  The host side of this file only moves the caller's arrays to HBM, launches and brings the
// results back; it holds no part of the computation.
//
// The reference works one position at a time; here a batch of positions ("sites") goes through tile kernels:
//
//   gap_read_info_kernel   lane per read      query length of the CIGAR, "has a reference skip" (bam2bcf_indel.c:130,321)
//   gap_type_kernel        workgroup per site the candidate indel types of the column: a hash set of the distinct p->indel
//                                             values in LDS, ranked into ascending order; the per-sample support filter,
//                                             the N filter, the window, the homopolymer run          (:106-181,236-247)
//   gap_scan_kernel        one workgroup      prefix sums of what each site needs (jobs, consensus windows, insertion rows)
//   gap_inscns_kernel      workgroup per site base counts of every insertion type (global atomics, few reads carry one),
//                                             majority consensus, insertions with an N dropped, est_indelreg  (:249-299)
//   gap_cons_kernel        wave per (site, sample): mismatch counters over the +-50 bp window (LDS atomics, a lane per read
//                                             walking its CIGAR), the two worst columns masked when fewer than 70 % of the
//                                             reads agree with the reference, then the realignment targets `ref2` of every
//                                             candidate type written out                             (:190-235,302-311)
//   probaln_kernel         lane per (site, type, read): the banded pair-HMM forward pass, indel.hip   (:313-357)
//   gap_finalize_kernel    workgroup per site per read the two smallest scores and the reference type's -> indelQ, seqQ;
//                                             LDS sums per type, the <= 4 output types, p->aux remapped (:372-459)
//
// Two small device->host reads size the buffers of the following stage (job / window totals after the typing; the count
// of wide-band jobs after the register-resident realignment pass); everything else is queued on the context's stream.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <chrono>
#include <limits.h>
#include "kernels.h"

using namespace bcfgpu;

extern "C" bcfgpu_gap_stats *bcfgpu_internal_gap_stats(bcfgpu_ctx *ctx);
extern "C" void *bcfgpu_internal_ws(bcfgpu_ctx *c, int slot, size_t bytes);
extern "C" int bcfgpu_internal_device(bcfgpu_ctx *c, hipStream_t *stream, const float **q2p);
extern "C" void *bcfgpu_internal_pinned(bcfgpu_ctx *c, int slot, size_t bytes);
int bcfgpu_set_error(int code, const char *what);

namespace bcfgpu {

#define GAP_EMPTY INT_MIN
#define INDEL_WINDOW_SIZE 50
#define INDEL_NULL 10000

// seq_nt16_table of htslib restricted to what a reference sequence holds: IUPAC letter -> 4-bit code
__device__ __forceinline__ int gap_nt16_of(char c)
{
    switch (c) {
        case 'A': case 'a': return 1;  case 'C': case 'c': return 2;  case 'G': case 'g': return 4;  case 'T': case 't': return 8;
        case '=': return 0;
        case 'M': case 'm': return 3;  case 'R': case 'r': return 5;  case 'S': case 's': return 6;  case 'V': case 'v': return 7;
        case 'W': case 'w': return 9;  case 'Y': case 'y': return 10; case 'H': case 'h': return 11; case 'K': case 'k': return 12;
        case 'D': case 'd': return 13; case 'B': case 'b': return 14;
        default: return 15;
    }
}
__device__ __forceinline__ int gap_nt16_int(int c) { return (int)((0x4444444344424104ull >> (4 * (c & 15))) & 7); }
__device__ __forceinline__ char gap_toupper(char c) { return (c >= 'a' && c <= 'z') ? (char)(c - 32) : c; }

__device__ __forceinline__ char gap_ref(const GapIn &in, long i)
{
    return (i >= in.ref_lo && i < in.ref_hi) ? in.ref[i - in.ref_lo] : (char)0;
}

// ---- per read: query length of the CIGAR (bam_cigar2qlen) and whether it holds a reference skip ----
__global__ __launch_bounds__(256) void gap_read_info_kernel(const GapIn in, uint32_t *r_info)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= in.n_reads) return;
    const uint32_t *cg = in.cig + in.r_cig_off[r];
    uint32_t ql = 0, hasn = 0;
    for (int k = 0; k < in.r_ncig[r]; ++k) {
        const uint32_t op = cg[k] & 0xf;
        if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) ql += cg[k] >> 4;
        if (op == 3) hasn = 1;
    }
    r_info[r] = (ql & 0x7fffffffu) | hasn << 31;
}

// ---- candidate types of every site ----
__global__ __launch_bounds__(256) void gap_type_kernel(const GapIn in, GapSite *sites, const uint32_t *r_info,
                                                       int32_t *o_ret, int32_t *o_types, int32_t *o_max_support, float *o_max_frac)
{
    __shared__ int s_keys[128], s_sorted[64];
    __shared__ int s_nkeys, s_over, s_nalt, s_ntot, s_ok, s_maxlen, s_nN, s_end, s_end2;
    __shared__ unsigned long long s_best;
    const int is = blockIdx.x, tid = threadIdx.x, n = in.n_smpl;
    const int32_t *soff = in.smpl_off + (size_t)is * n;
    const int e0 = soff[0], e1 = soff[n], N = e1 - e0;
    const int pos = in.pos[is];
    GapSite &S = sites[is];
    if (tid < 128) s_keys[tid] = GAP_EMPTY;
    if (tid == 0) { s_nkeys = 0; s_over = 0; s_nalt = 0; s_ntot = 0; s_ok = 0; s_maxlen = 0; s_nN = 0; s_best = 0; }
    __syncthreads();
    auto insert = [&](int v) {
        if (s_over) return;
        uint32_t h = ((uint32_t)v * 2654435761u) >> 25;
        for (int probe = 0; probe < 128; ++probe, h = (h + 1) & 127) {
            const int old = atomicCAS(&s_keys[h], GAP_EMPTY, v);
            if (old == GAP_EMPTY) { if (atomicAdd(&s_nkeys, 1) >= 64) s_over = 1; return; }
            if (old == v) return;
        }
        s_over = 1;
    };
    if (tid == 0) insert(0);                                   // zero indel is always a type (:120)
    // per sample: reads with an indel, support filter (:121-139)
    int my_nalt = 0, my_ntot = 0, my_ok = 0, my_maxlen = 0;
    unsigned long long my_best = 0;
    for (int s = tid; s < n; s += 256) {
        int na = 0;
        const int nt = soff[s + 1] - soff[s];
        for (int e = soff[s]; e < soff[s + 1]; ++e) {
            const int ind = in.p_indel[e];
            if (ind != 0) { ++na; insert(ind); }
            const int ql = (int)(r_info[in.p_read[e]] & 0x7fffffffu);
            my_maxlen = max(my_maxlen, ql);
        }
        if (nt > 0) {
            const double frac = (double)na / nt;
            if (na >= in.min_support && frac >= in.min_frac) my_ok = 1;
        }
        // "na > max_support && frac > 0": the largest na, the first sample on ties
        if (na > 0) { const unsigned long long cand = (unsigned long long)na << 32 | (0xffffffffu - (uint32_t)s); if (cand > my_best) my_best = cand; }
        my_nalt += na; my_ntot += nt;
    }
    atomicAdd(&s_nalt, my_nalt); atomicAdd(&s_ntot, my_ntot);
    if (my_ok) atomicOr(&s_ok, 1);
    atomicMax(&s_maxlen, my_maxlen);
    atomicMax(&s_best, my_best);
    __syncthreads();
    const int max_rd_len = s_maxlen, n_alt = s_nalt, n_tot = s_ntot;
    if (tid == 0) {
        S.live = 0; S.N = N; S.e0 = e0; S.pos = pos; S.n_types = 0; S.max_ins = 0; S.indelreg = 0;
        S.job0 = 0; S.ref2_0 = 0; S.ins0 = 0; S.max_ref2 = 0; S.left = S.right = 0; S.max_rd_len = max_rd_len;
        o_ret[is] = -1;
        for (int t = 0; t < 4; ++t) o_types[is * 4 + t] = INDEL_NULL;
        int msup = 0; float mfrac = 0;
        if (n_alt > 0 && s_best) {
            msup = (int)(s_best >> 32);
            const int sb = (int)(0xffffffffu - (uint32_t)(s_best & 0xffffffffu));
            mfrac = (float)((double)msup / (soff[sb + 1] - soff[sb]));
        }
        if (o_max_support) o_max_support[is] = msup;
        if (o_max_frac) o_max_frac[is] = mfrac;
        s_end = pos + max_rd_len;
    }
    __syncthreads();
    if (n_alt == 0) return;                                     // no indel at this position (:112)
    // N filter (:142-143): positions pos .. pos+max_rd_len-1, up to the end of the sequence
    for (int i = pos + tid; i < pos + max_rd_len; i += 256) if (gap_ref(in, i) == 0) atomicMin(&s_end, i);
    __syncthreads();
    {
        int c = 0;
        for (int i = pos + tid; i < s_end; i += 256) c += gap_ref(in, i) == 'N';
        if (c) atomicAdd(&s_nN, c);
    }
    __syncthreads();
    if (s_nN * 2 > s_end - pos) return;
    const int n_types = s_nkeys;
    int ok = s_ok;
    if (!in.per_sample_flt) ok = ((double)n_alt / n_tot < in.min_frac || n_alt < in.min_support) ? 0 : 1;
    if (n_types == 1 || !ok || s_over || n_types >= 64) return;   // (:152-161)
    // ascending order of the distinct types: rank = number of smaller keys
    if (tid < 128 && s_keys[tid] != GAP_EMPTY) {
        const int v = s_keys[tid];
        int rank = 0;
        for (int k = 0; k < 128; ++k) { const int o = s_keys[k]; rank += (o != GAP_EMPTY && o < v) ? 1 : 0; }
        s_sorted[rank] = v;
        if (v == 0) S.ref_type = rank;
    }
    __syncthreads();
    if (tid < n_types) S.types[tid] = s_sorted[tid];
    // window (:173-181)
    const int left = pos > INDEL_WINDOW_SIZE ? pos - INDEL_WINDOW_SIZE : 0;
    int right = pos + INDEL_WINDOW_SIZE;
    const int t0 = s_sorted[0], tl = s_sorted[n_types - 1];
    if (t0 < 0) right -= t0;
    if (tid == 0) s_end2 = right;
    __syncthreads();
    for (int i = pos + tid; i < right; i += 256) if (gap_ref(in, i) == 0) atomicMin(&s_end2, i);
    __syncthreads();
    right = s_end2;
    if (tid == 0) {
        // the length of the homopolymer run around the current position (:236-247)
        int l_run;
        const int c = gap_nt16_of(gap_ref(in, pos + 1));
        if (c == 15) l_run = 1;
        else {
            int i;
            for (i = pos + 2; gap_ref(in, i); ++i) if (gap_nt16_of(gap_ref(in, i)) != c) break;
            l_run = i;
            for (i = pos; i >= 0; --i) if (gap_nt16_of(gap_ref(in, i)) != c) break;
            l_run -= i + 1;
        }
        S.l_run = l_run;
        S.n_types = n_types; S.left = left; S.right = right;
        S.max_ins = tl;                                         // max_ins is at least 0 (:249)
        S.max_ref2 = (right - left + 2 + 2 * (tl > -t0 ? tl : -t0) + 3) & ~3;     // (a multiple of four: gap_cons_kernel stores four codes at a time)
        S.live = 1;
    }
}

// ---- offsets of each live site's share of the job / ref2 / insertion-consensus / packed-query pools ----
// One workgroup: every thread sums a contiguous slice of the sites, the 256 slice totals are scanned in LDS, and each
// thread goes through its slice again handing out the offsets.
__global__ __launch_bounds__(256) void gap_scan_kernel(GapSite *sites, int n_sites, int n_smpl, GapTotals *tot)
{
    __shared__ uint64_t s_sum[4][256];
    __shared__ int s_max[6][256];
    const int tid = threadIdx.x;
    const int per = (n_sites + 255) / 256, i0 = tid * per, i1 = min(n_sites, i0 + per);
    uint64_t jobs = 0, ref2 = 0, ins = 0, q8 = 0;
    int maxL = 0, max_bw = 0, n_live = 0, max_ref2 = 0, max_qstride = 0, max_N = 0;
    for (int is = i0; is < i1; ++is) {
        const GapSite &S = sites[is];
        if (!S.live) continue;
        ++n_live;
        jobs += (uint64_t)S.N * S.n_types;
        ref2 += (uint64_t)S.n_types * n_smpl * S.max_ref2;
        ins += (uint64_t)S.n_types * (S.max_ins > 0 ? S.max_ins : 0);
        const int qstride = (S.max_rd_len + 7) & ~7;           // qend - qbeg of an entry never exceeds its read's length
        q8 += (uint64_t)S.N * (uint32_t)(qstride >> 3);
        maxL = max(maxL, S.right - S.left + 1);
        max_ref2 = max(max_ref2, S.max_ref2);
        max_qstride = max(max_qstride, qstride);
        max_N = max(max_N, S.N);
        const int a = abs(S.types[0]), b = abs(S.types[S.n_types - 1]);
        max_bw = max(max_bw, max(a, b) + 3);
    }
    s_sum[0][tid] = jobs; s_sum[1][tid] = ref2; s_sum[2][tid] = ins; s_sum[3][tid] = q8;
    s_max[0][tid] = maxL; s_max[1][tid] = max_bw; s_max[2][tid] = n_live; s_max[3][tid] = max_ref2; s_max[4][tid] = max_qstride; s_max[5][tid] = max_N;
    __syncthreads();
    if (tid < 4) {                                             // exclusive scans of the four running totals
        uint64_t run = 0;
        for (int k = 0; k < 256; ++k) { const uint64_t v = s_sum[tid][k]; s_sum[tid][k] = run; run += v; }
        if (tid == 0) tot->n_jobs = run; else if (tid == 1) tot->ref2_bytes = run; else if (tid == 2) tot->ins_bytes = run; else tot->qpack8 = run;
    } else if (tid < 10) {
        const int w = tid - 4;
        int m = 0;
        for (int k = 0; k < 256; ++k) m = w == 2 ? m + s_max[w][k] : max(m, s_max[w][k]);
        if (w == 0) tot->max_L = m; else if (w == 1) tot->max_bw = m; else if (w == 2) tot->n_live = m; else if (w == 3) tot->max_ref2 = m; else if (w == 4) tot->max_qstride = m; else tot->max_N = m;
    } else if (tid == 10) { tot->n_wide = 0; tot->max_eff = 0; tot->n_passes = 0; tot->dp_cells = 0; tot->n_lds = 0; }
    __syncthreads();
    jobs = s_sum[0][tid]; ref2 = s_sum[1][tid]; ins = s_sum[2][tid]; q8 = s_sum[3][tid];
    for (int is = i0; is < i1; ++is) {
        GapSite &S = sites[is];
        S.job_end = (uint32_t)jobs;
        if (!S.live) continue;
        S.job0 = (uint32_t)jobs; S.ref2_0 = (uint32_t)ref2; S.ins0 = (uint32_t)ins;
        jobs += (uint64_t)S.N * S.n_types;
        S.job_end = (uint32_t)jobs;
        ref2 += (uint64_t)S.n_types * n_smpl * S.max_ref2;
        ins += (uint64_t)S.n_types * (S.max_ins > 0 ? S.max_ins : 0);
        S.qstride = (S.max_rd_len + 7) & ~7;
        S.q8_0 = (uint32_t)q8;
        q8 += (uint64_t)S.N * (uint32_t)(S.qstride >> 3);
    }
}

// the site of pileup entry e: the last one with smpl_off[site*n] <= e
__device__ __forceinline__ int gap_site_of_entry(const GapIn &in, int e)
{
    int lo = 0, hi = in.n_sites - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (in.smpl_off[(size_t)mid * in.n_smpl] <= e) lo = mid; else hi = mid - 1; }
    return lo;
}

// ---- insertion consensus and est_indelreg ----
// ins_cnt: [ins_bytes][5] counters (zeroed), inscns: [ins_bytes] (zeroed: the positions after a dropped one stay 0 as in the
// reference's calloc'ed array)
// the occurrences of each base at each position of each type of insertion (:253-269): a lane per pileup entry, the few
// entries that carry an insertion add their bases
__global__ __launch_bounds__(256) void gap_inscnt_kernel(const GapIn in, const GapSite *sites, int n_ent, int32_t *ins_cnt)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_ent) return;
    const int ind = in.p_indel[e];
    if (ind <= 0) return;
    const GapSite &S = sites[gap_site_of_entry(in, e)];
    if (!S.live || S.max_ins <= 0) return;
    int t = 0;
    for (; t < S.n_types; ++t) if (S.types[t] == ind) break;
    const uint8_t *seq = in.seq16 + in.r_seq_off[in.p_read[e]];
    const int qpos = in.p_qpos[e];
    // lanes of a wavefront adding to the same counter (the carriers of one insertion, one after another in the pileup) add once
    const unsigned lane = threadIdx.x & 63;
    for (int j = 1; j <= ind; ++j) {
        const int c = gap_nt16_int(seq[qpos + j] & 15);
        const unsigned long long key = ((unsigned long long)S.ins0 + (size_t)t * S.max_ins + (j - 1)) * 5 + c;
        bool done = false;
        while (!done) {                                        // (lanes leave as their counter is served)
            const int leader = __builtin_ctzll(__builtin_amdgcn_ballot_w64(true));
            const unsigned long long lk = __shfl(key, leader);
            const unsigned long long same = __builtin_amdgcn_ballot_w64(key == lk);
            if (key == lk) { if (lane == (unsigned)leader) atomicAdd(&ins_cnt[lk], (int)__popcll(same)); done = true; }
        }
    }
}

__global__ __launch_bounds__(64) void gap_inscns_kernel(const GapIn in, GapSite *sites, const int32_t *ins_cnt, int8_t *inscns)
{
    __shared__ int s_ir;
    const int is = blockIdx.x, tid = threadIdx.x;
    GapSite &S = sites[is];
    if (!S.live) return;
    const int n_types = S.n_types, max_ins = S.max_ins, pos = S.pos;
    if (tid == 0) s_ir = 0;
    if (max_ins > 0) {
        // majority rule (:271-281); an insertion whose consensus holds an N is dropped: its type becomes 0
        for (int t = tid; t < n_types; t += 64) {
            const int len = S.types[t];
            for (int j = 0; j < len; ++j) {
                const int32_t *ia = &ins_cnt[((size_t)S.ins0 + (size_t)t * max_ins + j) * 5];
                int mx = 0, mk = -1;
                for (int k = 0; k < 5; ++k) { const int v = ia[k]; if (v > mx) { mx = v; mk = k; } }
                inscns[(size_t)S.ins0 + (size_t)t * max_ins + j] = mx ? (int8_t)mk : (int8_t)4;
                if (mk == 4) { S.types[t] = 0; break; }
            }
        }
    }
    __syncthreads();
    // est_indelreg of every type (:77-88,296-299), the site's value is the largest
    for (int t = tid; t < n_types; t += 64) {
        const int ty = S.types[t];
        int ir = 0;
        if (ty != 0) {
            const int l = abs(ty);
            const int8_t *ins4 = ty > 0 ? &inscns[(size_t)S.ins0 + (size_t)t * max_ins] : nullptr;
            int mx = 0, max_i = pos, score = 0;
            for (int i = pos + 1, j = 0; gap_ref(in, i); ++i, ++j) {
                const char rc = gap_toupper(gap_ref(in, i));
                if (ins4) score += (rc != "ACGTN"[(int)ins4[j % l]]) ? -10 : 1;
                else score += (rc != gap_toupper(gap_ref(in, pos + 1 + j % l))) ? -10 : 1;
                if (score < 0) break;
                if (mx < score) { mx = score; max_i = i; }
            }
            ir = max_i - pos;
        }
        if (ir > 0) atomicMax(&s_ir, ir);
    }
    __syncthreads();
    if (tid == 0) S.indelreg = s_ir;
}

// ---- per-sample consensus and the realignment targets ----
// One wavefront per (site, sample), a lane per window column (two columns per lane: the window is ~100 wide): the lane
// keeps its columns' reference codes and match / mismatch counts in registers while the wavefront goes through the
// sample's reads together.  Each lane fetches one read's record (four dependent loads that the whole wavefront would
// otherwise wait out read after read); the record travels to every lane through v_readlane into scalar registers, so the
// CIGAR walk is scalar code and only the base under the lane's column is a vector load.  Reads whose CIGAR is one match
// operation -- the common case -- go four at a time, their base loads in flight together.  No atomics, no LDS traffic,
// no workgroup barrier in this phase, and every lane is busy whatever the sample's depth.
#define GAP_CONS_THREADS 64
__global__ __launch_bounds__(GAP_CONS_THREADS) void gap_cons_kernel(const GapIn in, const GapSite *sites, const int8_t *inscns, uint8_t *ref2pool)
{
    extern __shared__ uint32_t s_cns[];            // [max_L] counts, then [max_L] bytes: the sample's consensus codes
    __shared__ int s_m[2];
    const int is = blockIdx.x / in.n_smpl, s = blockIdx.x % in.n_smpl, tid = threadIdx.x;
    const GapSite &S = sites[is];
    if (!S.live) return;
    const int left = S.left, right = S.right, W = right - left, pos = S.pos;
    const int32_t *soff = in.smpl_off + (size_t)is * in.n_smpl;
    const int e0 = soff[s], e1 = soff[s + 1];
    uint32_t cnt0 = 0, cnt1 = 0;                              // this lane's columns tid and tid + 64
    const int colA = left + tid, colB = colA + GAP_CONS_THREADS;
    const bool mineA = tid < W, mineB = tid + GAP_CONS_THREADS < W;
    const int rcA = mineA ? gap_nt16_of(gap_ref(in, colA)) : -1, rcB = mineB ? gap_nt16_of(gap_ref(in, colB)) : -1;
    for (int eb = e0; eb < e1; eb += GAP_CONS_THREADS) {
        const int nb = min(GAP_CONS_THREADS, e1 - eb);
        int my_pos = 0, my_soff = 0, my_ncig = 0, my_coff = 0; uint32_t my_c0 = 0;
        if (tid < nb) {
            const int r = in.p_read[eb + tid];
            my_pos = in.r_pos[r]; my_soff = in.r_seq_off[r]; my_ncig = in.r_ncig[r]; my_coff = in.r_cig_off[r];
            my_c0 = my_ncig > 0 ? in.cig[my_coff] : 0u;
        }
        auto general = [&](int k) {                           // read k of this round, any CIGAR
            const int x0 = __builtin_amdgcn_readlane(my_pos, k), ncig = __builtin_amdgcn_readlane(my_ncig, k);
            const int coff = __builtin_amdgcn_readlane(my_coff, k);
            const uint8_t *seq = in.seq16 + __builtin_amdgcn_readlane(my_soff, k);
            uint32_t cg = (uint32_t)__builtin_amdgcn_readlane((int)my_c0, k);
            int x = x0, y = 0;
            for (int c = 0; c < ncig; ++c) {
                if (c) cg = in.cig[coff + c];
                const int op = cg & 0xf, l = (int)(cg >> 4);
                if (op == 0 || op == 7 || op == 8) {
                    if (mineA && colA >= x && colA < x + l) cnt0 += (int)(seq[y + (colA - x)] & 15) == rcA ? 1u : 0x10000u;
                    if (mineB && colB >= x && colB < x + l) cnt1 += (int)(seq[y + (colB - x)] & 15) == rcB ? 1u : 0x10000u;
                    x += l; y += l;
                } else if (op == 2 || op == 3) x += l;
                else if (op == 1 || op == 4) y += l;
            }
        };
        int k0 = 0;
        for (; k0 + 4 <= nb; k0 += 4) {
            int x[4], l[4], so[4];
            bool simple = true;
            #pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t cg = (uint32_t)__builtin_amdgcn_readlane((int)my_c0, k0 + u);
                x[u] = __builtin_amdgcn_readlane(my_pos, k0 + u); so[u] = __builtin_amdgcn_readlane(my_soff, k0 + u); l[u] = (int)(cg >> 4);
                simple = simple && __builtin_amdgcn_readlane(my_ncig, k0 + u) == 1 && (cg & 0xf) == 0;
            }
            if (simple) {
                int bA[4], bB[4];
                #pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint8_t *seq = in.seq16 + so[u];
                    bA[u] = (mineA && colA >= x[u] && colA < x[u] + l[u]) ? (int)(seq[colA - x[u]] & 15) : -2;
                    bB[u] = (mineB && colB >= x[u] && colB < x[u] + l[u]) ? (int)(seq[colB - x[u]] & 15) : -2;
                }
                #pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (bA[u] != -2) cnt0 += bA[u] == rcA ? 1u : 0x10000u;
                    if (bB[u] != -2) cnt1 += bB[u] == rcB ? 1u : 0x10000u;
                }
            } else {
                #pragma unroll 1
                for (int u = 0; u < 4; ++u) general(k0 + u);
            }
        }
        for (; k0 < nb; ++k0) general(k0);
    }
    if (mineA) s_cns[tid] = cnt0;
    if (mineB) s_cns[tid + GAP_CONS_THREADS] = cnt1;
    // windows wider than two columns per lane (a deletion longer than ~150 bases among the types): the plain loop
    for (int c0 = 2 * GAP_CONS_THREADS; c0 < W; c0 += GAP_CONS_THREADS) {
        const int col = left + c0 + tid;
        const bool mine = c0 + tid < W;
        const int rc = mine ? gap_nt16_of(gap_ref(in, col)) : -1;
        uint32_t cnt = 0;
        for (int e = e0; e < e1; ++e) {
            const int r = in.p_read[e];
            const uint32_t *cigar = in.cig + in.r_cig_off[r];
            const uint8_t *seq = in.seq16 + in.r_seq_off[r];
            const int ncig = in.r_ncig[r];
            int x = in.r_pos[r], y = 0;
            for (int k = 0; k < ncig; ++k) {
                const int op = cigar[k] & 0xf, l = (int)(cigar[k] >> 4);
                if (op == 0 || op == 7 || op == 8) {
                    if (mine && col >= x && col < x + l) cnt += (int)(seq[y + (col - x)] & 15) == rc ? 1u : 0x10000u;
                    x += l; y += l;
                } else if (op == 2 || op == 3) x += l;
                else if (op == 1 || op == 4) y += l;
            }
        }
        if (mine) s_cns[c0 + tid] = cnt;
    }
    __syncthreads();
    // The two columns with most mismatches (:223-231).  The reference scans the columns in order: a count >= the running
    // maximum becomes the maximum (the old one the runner-up), else a count >= the runner-up becomes the runner-up.  At the end
    // the maximum is the LAST column holding the largest count and the runner-up the last column holding the largest count
    // among the others: two reductions over (count, column) pairs by the first wavefront.
    if (tid < 64) {
        auto last_max = [&](int skip) {
            unsigned long long best = 0;                      // (mismatches + 1) << 32 | column: 0 = none
            for (int i = tid; i < W; i += 64)
                if (i != skip) { const unsigned long long k = (unsigned long long)((s_cns[i] >> 16) + 1) << 32 | (unsigned)i; if (k > best) best = k; }
            #pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const unsigned long long k = __shfl_xor(best, o); if (k > best) best = k; }
            return best ? (int)(best & 0xffffffffu) : -1;
        };
        int max_i = last_max(-1);
        int max2_i = last_max(max_i);
        if (tid == 0) {
            const uint32_t mx = max_i >= 0 ? s_cns[max_i] : 0u, mx2 = max2_i >= 0 ? s_cns[max2_i] : 0u;
            if ((double)(mx & 0xffff) / ((mx & 0xffff) + (mx >> 16)) >= 0.7) max_i = -1;
            if ((double)(mx2 & 0xffff) / ((mx2 & 0xffff) + (mx2 >> 16)) >= 0.7) max2_i = -1;
            s_m[0] = max_i; s_m[1] = max2_i;
        }
    }
    __syncthreads();
    const int m1 = s_m[0], m2 = s_m[1];
    // the sample's consensus over the window as base codes 0..4, once (the masked columns read as N)
    uint8_t *s_code = reinterpret_cast<uint8_t*>(s_cns + W);
    for (int c = tid; c < W; c += GAP_CONS_THREADS)
        s_code[c] = (uint8_t)gap_nt16_int((c == m1 || c == m2) ? 15 : gap_nt16_of(gap_ref(in, left + c)));
    __syncthreads();
    // ref2 of every type (:302-311): the sample's consensus left of the indel, the inserted consensus or the deletion,
    // the consensus right of it, padded with N.  max_ref2 is a multiple of four and so is every row's offset: four codes
    // per store.
    const int n1 = pos - left + 1, max_ref2 = S.max_ref2;
    for (int t = 0; t < S.n_types; ++t) {
        const int ty = S.types[t];
        uint32_t *dst = reinterpret_cast<uint32_t*>(ref2pool + (size_t)S.ref2_0 + ((size_t)t * in.n_smpl + s) * max_ref2);
        const int8_t *ic = inscns + (size_t)S.ins0 + (size_t)t * (S.max_ins > 0 ? S.max_ins : 0);
        for (int k4 = tid; k4 * 4 < max_ref2; k4 += GAP_CONS_THREADS) {
            uint32_t w = 0;
            #pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k4 * 4 + u;
                uint32_t v = 4;
                if (k < n1) v = s_code[k];
                else if (ty > 0) {
                    if (k < n1 + ty) v = (uint8_t)ic[k - n1];
                    else { const int j = pos + 1 + (k - n1 - ty); if (j < right) v = s_code[j - left]; }
                } else { const int j = pos + 1 - ty + (k - n1); if (j < right) v = s_code[j - left]; }
                w |= v << (8 * u);
            }
            dst[k4] = w;
        }
    }
}

// bam2bcf_indel.c:69-75
__device__ __forceinline__ int gap_est_seqQ(const GapIn &in, int l, int l_run)
{
    const int q = in.openQ + in.extQ * (abs(l) - 1);
    const int qh = l_run >= 3 ? (int)(in.tandemQ * (double)abs(l) / l_run + .499) : 1000;
    return q < qh ? q : qh;
}

// ---- indelQ / seqQ of every read, the output types, p->aux (:372-459) ----
// Three launches: (1) per entry the best type, indelQ and seqQ, and per site the types' quality sums (a workgroup works
// on up to GAP_FIN_SEG entries of ONE site: LDS sums, one global atomic per type and workgroup); (2) per site the <= 4 output
// types; (3) per entry p->aux with the index among the output types.
#define GAP_FIN_SEG 1024
__global__ __launch_bounds__(256) void gap_fin_entries_kernel(const GapIn in, const GapSite *sites, const int32_t *score1, const int32_t *score2,
                                                              uint32_t *p_aux, int32_t *sumq)
{
    __shared__ int s_types[64], s_sumq[64];
    const int is = blockIdx.x, tid = threadIdx.x;
    const GapSite &S = sites[is];
    if (!S.live) return;
    const int n_types = S.n_types, ref_type = S.ref_type, N = S.N, l_run = S.l_run;
    const int k0 = blockIdx.y * GAP_FIN_SEG, k1 = min(N, k0 + GAP_FIN_SEG);
    if (k0 >= N) return;
    if (tid < 64) { s_types[tid] = tid < n_types ? S.types[tid] : 0; s_sumq[tid] = 0; }
    __syncthreads();
    // The reference sorts sc[t] = score<<6 | t and uses the smallest, the second smallest and the reference type's value.
    auto pick = [&](const int32_t *sc, int K, int &v0, int &v1, int &vref) {
        v0 = INT_MAX; v1 = INT_MAX; vref = 0;
        for (int t = 0; t < n_types; ++t) {
            const int v = sc[(size_t)S.job0 + (size_t)t * N + K] << 6 | t;
            if (t == ref_type) vref = v;
            if (v < v0) { v1 = v0; v0 = v; } else if (v < v1) v1 = v;
        }
    };
    for (int K = k0 + tid; K < k1; K += 256) {
        int a0, a1, aref, indelQ1, indelQ2, seqQ;
        pick(score1, K, a0, a1, aref);
        if ((a0 & 0x3f) == ref_type) { indelQ1 = (a1 >> 14) - (a0 >> 14); seqQ = gap_est_seqQ(in, s_types[a1 & 0x3f], l_run); }
        else { indelQ1 = (aref >> 14) - (a0 >> 14); seqQ = gap_est_seqQ(in, s_types[a0 & 0x3f], l_run); }
        int tmp = a0 >> 6 & 0xff;
        indelQ1 = tmp > 111 ? 0 : (int)((1. - tmp / 111.) * indelQ1 + .499);
        pick(score2, K, a0, a1, aref);
        if ((a0 & 0x3f) == ref_type) indelQ2 = (a1 >> 14) - (a0 >> 14);
        else indelQ2 = (aref >> 14) - (a0 >> 14);
        tmp = a0 >> 6 & 0xff;
        indelQ2 = tmp > 111 ? 0 : (int)((1. - tmp / 111.) * indelQ2 + .499);
        int indelQ = indelQ1 < indelQ2 ? indelQ1 : indelQ2;
        if (indelQ > 255) indelQ = 255;
        if (seqQ > 255) seqQ = 255;
        p_aux[S.e0 + K] = (uint32_t)((a0 & 0x3f) << 16 | seqQ << 8 | indelQ);
        atomicAdd(&s_sumq[a0 & 0x3f], indelQ < seqQ ? indelQ : seqQ);
    }
    __syncthreads();
    if (tid < n_types && s_sumq[tid]) atomicAdd(&sumq[(size_t)is * 64 + tid], s_sumq[tid]);
}

__global__ __launch_bounds__(64) void gap_fin_types_kernel(const GapSite *sites, const int32_t *sumq, const int8_t *inscns, int32_t *o_types,
                                                           int8_t *o_inscns, int inscns_cap, int32_t *o_maxins, int32_t *o_indelreg, uint8_t *otype_of)
{
    const int is = blockIdx.x, tid = threadIdx.x;
    const GapSite &S = sites[is];
    if (!S.live) return;
    __shared__ int s_otypes[4];
    const int n_types = S.n_types, ref_type = S.ref_type;
    if (tid == 0) {
        // the types with the largest quality sums, the reference type first (:431-447)
        int sq[64];
        for (int t = 0; t < n_types; ++t) sq[t] = sumq[(size_t)is * 64 + t] << 6 | t;
        for (int t = 1; t < n_types; ++t)
            for (int j = t; j > 0 && sq[j] > sq[j - 1]; --j) { const int x = sq[j]; sq[j] = sq[j - 1]; sq[j - 1] = x; }
        int t;
        for (t = 0; t < n_types; ++t) if ((sq[t] & 0x3f) == ref_type) break;
        if (t) { const int x = sq[t]; for (; t > 0; --t) sq[t] = sq[t - 1]; sq[0] = x; }
        for (t = 0; t < 4; ++t) s_otypes[t] = INDEL_NULL;
        const int max_ins = S.max_ins;
        for (t = 0; t < 4 && t < n_types; ++t) {
            s_otypes[t] = S.types[sq[t] & 0x3f];
            if (o_inscns && max_ins > 0 && (t + 1) * max_ins <= 4 * inscns_cap)
                for (int k = 0; k < max_ins; ++k)
                    o_inscns[(size_t)is * 4 * inscns_cap + (size_t)t * max_ins + k] = inscns[(size_t)S.ins0 + (size_t)(sq[t] & 0x3f) * max_ins + k];
        }
        for (t = 0; t < 4; ++t) o_types[is * 4 + t] = s_otypes[t];
        if (o_maxins) o_maxins[is] = max_ins;
        if (o_indelreg) o_indelreg[is] = S.indelreg;
    }
    __syncthreads();
    // type index -> index among the output types (4: not an output type), by VALUE as the reference compares them (:449-458)
    if (tid < n_types) {
        const int x = S.types[tid];
        int j;
        for (j = 0; j < 4; ++j) if (x == s_otypes[j]) break;
        otype_of[(size_t)is * 64 + tid] = (uint8_t)j;
    }
}

__global__ __launch_bounds__(256) void gap_fin_aux_kernel(const GapSite *sites, const uint8_t *otype_of, uint32_t *p_aux, int32_t *o_ret)
{
    const int is = blockIdx.x, tid = threadIdx.x;
    const GapSite &S = sites[is];
    if (!S.live) return;
    const int N = S.N, k0 = blockIdx.y * GAP_FIN_SEG, k1 = min(N, k0 + GAP_FIN_SEG);
    if (k0 >= N) return;
    bool alt = false;
    for (int K = k0 + tid; K < k1; K += 256) {
        const uint32_t a = p_aux[S.e0 + K];
        const int j = otype_of[(size_t)is * 64 + (a >> 16 & 0x3f)];
        const uint32_t v = (uint32_t)(j << 16) | (j == 4 ? 0u : (a & 0xffff));
        p_aux[S.e0 + K] = v;
        alt |= (v >> 16 & 0x3f) > 0;
    }
    if (__any(alt) && (tid & 63) == 0) o_ret[is] = 0;         // (set to -1 by gap_type_kernel; every writer stores the same 0)
}

}  // namespace bcfgpu

// =====================================================================================================================
// host: transfers and launch sequencing only
#define GP_CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return bcfgpu_set_error(BCFGPU_E_HIP, #call); } while (0)
#define WS(slot, bytes) bcfgpu_internal_ws(ctx, 40 + (slot), (bytes) + 64)      /* slots 40..: this stage's own */

extern "C" int bcfgpu_internal_n_cu(const bcfgpu_ctx *c);
extern "C" int bcfgpu_internal_side(bcfgpu_ctx *c, hipStream_t **streams, hipEvent_t **events);

// The stage on arrays that are already in HBM (`g`: device pointers throughout; n_ent pileup entries).  d_aux [n_ent]
// (device) receives p->aux; the per-site outputs go to the host arrays of `out` (out->p_aux is not touched: the callers
// decide whether the entries' words leave the device).  Three waits on the stream: the totals that size the second
// half's buffers, the count of wide-band jobs, the per-site results.
int bcfgpu_internal_gap_core(bcfgpu_ctx *ctx, const GapIn &g, size_t n_ent, uint32_t *d_aux, const bcfgpu_indel_out *out, int inscns_cap)
{
    hipStream_t st;
    const float *q2p;
    if (bcfgpu_internal_device(ctx, &st, &q2p)) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_gap_prep: bad context");
    bcfgpu_gap_stats &gs = *bcfgpu_internal_gap_stats(ctx);
    const auto t_begin = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t0) {
        return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    const int ns = g.n_sites, n = g.n_smpl, nr = g.n_reads;
    GapSite *d_sites = (GapSite*)WS(25, (size_t)ns * sizeof(GapSite));
    uint32_t *d_rinfo = (uint32_t*)WS(26, (size_t)nr * 4);
    // small outputs share one block: ret, types[4], maxins, indelreg, max_support, max_frac per site, then the totals
    const size_t so_ret = 0, so_types = (size_t)ns * 4, so_maxins = so_types + (size_t)ns * 16, so_ireg = so_maxins + (size_t)ns * 4,
                 so_msup = so_ireg + (size_t)ns * 4, so_mfrac = so_msup + (size_t)ns * 4, so_tot = (so_mfrac + (size_t)ns * 4 + 15) & ~(size_t)15,
                 so_bytes = so_tot + sizeof(GapTotals) + sizeof(ProbalnQueue);
    uint8_t *d_small = (uint8_t*)WS(28, so_bytes);
    uint8_t *h_small = (uint8_t*)bcfgpu_internal_pinned(ctx, 0, so_bytes);
    if (!d_sites || !d_rinfo || !d_small || !h_small) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_gap_prep: device workspace");
    int32_t *o_ret = (int32_t*)(d_small + so_ret), *o_types = (int32_t*)(d_small + so_types), *o_maxins = (int32_t*)(d_small + so_maxins);
    int32_t *o_ireg = (int32_t*)(d_small + so_ireg), *o_msup = (int32_t*)(d_small + so_msup);
    float *o_mfrac = (float*)(d_small + so_mfrac);
    GapTotals *d_tot = (GapTotals*)(d_small + so_tot);
    GP_CHK(hipMemsetAsync(d_small, 0, so_bytes, st));
    if (n_ent) GP_CHK(hipMemsetAsync(d_aux, 0, n_ent * 4, st));

    // ---- typing ----
    if (nr) hipLaunchKernelGGL(gap_read_info_kernel, dim3((nr + 255) / 256), dim3(256), 0, st, g, d_rinfo);
    hipLaunchKernelGGL(gap_type_kernel, dim3(ns), dim3(256), 0, st, g, d_sites, d_rinfo, o_ret, o_types, o_msup, o_mfrac);
    hipLaunchKernelGGL(gap_scan_kernel, dim3(1), dim3(256), 0, st, d_sites, ns, n, d_tot);
    GapTotals tot{};
    GP_CHK(hipMemcpyAsync(h_small, d_tot, sizeof(GapTotals), hipMemcpyDeviceToHost, st));
    GP_CHK(hipStreamSynchronize(st));                           // sizes of the next stage's buffers
    memcpy(&tot, h_small, sizeof tot);
    gs.prepare_ms += ms_since(t_begin);
    if (tot.n_jobs >> 31 || tot.ref2_bytes >> 32 || tot.qpack8 >> 32)
        return bcfgpu_set_error(BCFGPU_E_RANGE, "bcfgpu_gap_prep: batch too large (pool offsets are 32-bit), use fewer sites per call");
    gs.n_jobs = tot.n_jobs;
    int8_t *d_oinscns = nullptr;
    if (tot.n_live) {
        const size_t nj = (size_t)tot.n_jobs;
        int32_t *d_inscnt = (int32_t*)WS(16, (size_t)tot.ins_bytes * 5 * 4);
        int8_t *d_inscns = (int8_t*)WS(17, (size_t)tot.ins_bytes);
        // (the realignment reads 8 bytes at a time, up to two groups ahead; the LDS class reads the bases under its whole band, which
        // hangs over a row's ends by up to PROBALN_LDS16_MAX + 16 positions: 512 bytes of padding on both sides)
        uint8_t *d_ref2 = (uint8_t*)WS(18, (size_t)tot.ref2_bytes + 1024);
        if (d_ref2) d_ref2 += 512;
        int32_t *d_s1 = (int32_t*)WS(19, nj * 4), *d_s2 = (int32_t*)WS(20, nj * 4);
        uint32_t *d_wide = (uint32_t*)WS(21, nj * 4);
        GapEntry *d_ent = (GapEntry*)WS(29, n_ent * sizeof(GapEntry));
        uint8_t *d_qpack = (uint8_t*)WS(30, (size_t)tot.qpack8 * 8 + 64);
        PJob *d_pjob = (PJob*)WS(31, nj * sizeof(PJob));
        uint32_t *d_k0 = (uint32_t*)WS(32, nj * 4), *d_v0 = (uint32_t*)WS(33, nj * 4), *d_k1 = (uint32_t*)WS(34, nj * 4), *d_v1 = (uint32_t*)WS(35, nj * 4);
        uint32_t *d_list2 = (uint32_t*)WS(36, nj * 4);
        ProbalnQueue *d_queue = (ProbalnQueue*)WS(38, sizeof(ProbalnQueue));
        double2 *d_emt = (double2*)bcfgpu_internal_ws(ctx, 133, 256 * sizeof(double2));
        int32_t *d_sumq = (int32_t*)WS(39, (size_t)ns * 64 * 4);
        uint8_t *d_otype = (uint8_t*)WS(40, (size_t)ns * 64);
        size_t sort_bytes = 0;
        GP_CHK(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, d_k0, d_k1, d_v0, d_v1, (int)nj, 0, 17, st));
        void *d_sort = WS(37, sort_bytes);
        if (out->inscns) d_oinscns = (int8_t*)WS(22, (size_t)ns * 4 * inscns_cap);
        if (!d_inscnt || !d_inscns || !d_ref2 || !d_s1 || !d_s2 || !d_wide || !d_ent || !d_qpack || !d_pjob || !d_k0 || !d_v0 || !d_k1 || !d_v1 ||
            !d_list2 || !d_queue || !d_emt || !d_sort || !d_sumq || !d_otype || (out->inscns && !d_oinscns))
            return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_gap_prep: device workspace");
        if (tot.ins_bytes) { GP_CHK(hipMemsetAsync(d_inscnt, 0, (size_t)tot.ins_bytes * 5 * 4, st)); GP_CHK(hipMemsetAsync(d_inscns, 0, (size_t)tot.ins_bytes, st)); }
        GP_CHK(hipMemsetAsync(d_s1, 0, nj * 4, st)); GP_CHK(hipMemsetAsync(d_s2, 0, nj * 4, st));
        GP_CHK(hipMemsetAsync(d_sumq, 0, (size_t)ns * 64 * 4, st));
        if (d_oinscns) GP_CHK(hipMemsetAsync(d_oinscns, 0, (size_t)ns * 4 * inscns_cap, st));
        if (tot.ins_bytes && n_ent) hipLaunchKernelGGL(gap_inscnt_kernel, dim3((unsigned)((n_ent + 255) / 256)), dim3(256), 0, st, g, d_sites, (int)n_ent, d_inscnt);
        hipLaunchKernelGGL(gap_inscns_kernel, dim3(ns), dim3(64), 0, st, g, d_sites, d_inscnt, d_inscns);
        hipLaunchKernelGGL(gap_cons_kernel, dim3((unsigned)((size_t)ns * n)), dim3(GAP_CONS_THREADS), (size_t)tot.max_L * 5 + 16, st, g, d_sites, d_inscns, d_ref2);
        // ---- realignment: the jobs decoded and sorted by band, register-resident passes per band width; the jobs with
        // wider bands are listed and run from scratch rows ----
        launch_gap_entries(g, d_sites, (int)n_ent, d_ent, tot.max_qstride, d_qpack, st);
        ProbalnParams p{};
        p.ent = d_ent;
        p.gin = g; p.sites = d_sites; p.n_sites = ns; p.n_jobs = (int)nj;
        p.ref2 = d_ref2; p.qpack = d_qpack; p.q2p = q2p; p.emt = d_emt; p.score1 = d_s1; p.score2 = d_s2;
        p.pjob = d_pjob; p.key_in = d_k0; p.val_in = d_v0; p.key_sorted = d_k1; p.val_sorted = d_v1; p.list2 = d_list2; p.queue = d_queue;
        p.wide = d_wide; p.tot = d_tot;
#ifdef BCFGPU_DIAG
        p.force_wide = getenv("BCFGPU_FORCE_WIDE") != nullptr;      // diagnostics build only: every job through the rolling-row kernel
#endif
        hipEvent_t e0 = nullptr, e1 = nullptr;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, st);
        launch_probaln_jobs(p, st);
        GP_CHK(hipcub::DeviceRadixSort::SortPairs(d_sort, sort_bytes, d_k0, d_k1, d_v0, d_v1, (int)nj, 0, 17, st));
        launch_probaln_bounds(p, st);
        {
            hipStream_t *side = nullptr; hipEvent_t *sev = nullptr;
            if (bcfgpu_internal_side(ctx, &side, &sev) || launch_probaln_exact(p, st, bcfgpu_internal_n_cu(ctx), side, sev))
                return bcfgpu_set_error(BCFGPU_E_HIP, "bcfgpu_gap_prep: side streams");
        }
        GP_CHK(hipMemcpyAsync(h_small, d_tot, sizeof(GapTotals), hipMemcpyDeviceToHost, st));
        GP_CHK(hipMemcpyAsync(h_small + sizeof(GapTotals), d_queue, sizeof(ProbalnQueue), hipMemcpyDeviceToHost, st));
        GP_CHK(hipStreamSynchronize(st));                       // how many jobs need the wide-band version, and how wide
        memcpy(&tot, h_small, sizeof tot);
        {   // jobs by band width (statistics): the register classes, the five groups of the LDS class, sixteen-a-wavefront, scratch
            ProbalnQueue hq;
            memcpy(&hq, h_small + sizeof(GapTotals), sizeof hq);
            gs.band_jobs[0] = hq.cls_begin[PROBALN_CLS_LDS] - hq.cls_begin[PROBALN_BW_MIN];
            for (int g = 0; g < PROBALN_LDS_GROUPS; ++g) gs.band_jobs[1 + g] = hq.lds_begin[g + 1] - hq.lds_begin[g];
            gs.band_jobs[1 + PROBALN_LDS_GROUPS] = hq.lds_begin[PROBALN_LDS_GROUPS + 2] - hq.lds_begin[PROBALN_LDS_GROUPS];
            gs.band_jobs[7] = tot.n_wide;
        }
        gs.n_wide = (uint64_t)tot.n_wide + tot.n_lds; gs.n_scratch = tot.n_wide;
        if (tot.n_wide) {
            p.ncell = 3 * (2 * tot.max_eff + 1) + 6;
            size_t chunk = ((size_t)1 << 30) / (2 * (size_t)p.ncell * sizeof(double));
            chunk = chunk < 64 ? 64 : (chunk & ~(size_t)63);
            if (chunk > tot.n_wide) chunk = ((size_t)tot.n_wide + 63) & ~(size_t)63;
            p.scratch_stride = chunk;
            p.scratch = (double*)WS(23, 2 * (size_t)p.ncell * chunk * sizeof(double));
            if (!p.scratch) return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_gap_prep: device workspace");
            for (size_t j0 = 0; j0 < tot.n_wide; j0 += chunk) {
                p.wide_first = (uint32_t)j0;
                p.wide_count = (int)(tot.n_wide - j0 < chunk ? tot.n_wide - j0 : chunk);
                launch_probaln_wide(p, st);
            }
        }
        hipEventRecord(e1, st);
        {
            const dim3 fgrid((unsigned)ns, (unsigned)((tot.max_N + GAP_FIN_SEG - 1) / GAP_FIN_SEG));
            hipLaunchKernelGGL(gap_fin_entries_kernel, fgrid, dim3(256), 0, st, g, d_sites, d_s1, d_s2, d_aux, d_sumq);
            hipLaunchKernelGGL(gap_fin_types_kernel, dim3(ns), dim3(64), 0, st, d_sites, d_sumq, d_inscns, o_types, d_oinscns, inscns_cap, o_maxins, o_ireg, d_otype);
            hipLaunchKernelGGL(gap_fin_aux_kernel, fgrid, dim3(256), 0, st, d_sites, d_otype, d_aux, o_ret);
        }
        GP_CHK(hipGetLastError());
        GP_CHK(hipMemcpyAsync(h_small, d_small, so_bytes, hipMemcpyDeviceToHost, st));
        if (out->inscns) GP_CHK(hipMemcpyAsync(out->inscns, d_oinscns, (size_t)ns * 4 * inscns_cap, hipMemcpyDeviceToHost, st));
        GP_CHK(hipStreamSynchronize(st));
        hipEventElapsedTime(&gs.kernel_ms, e0, e1);
        hipEventDestroy(e0); hipEventDestroy(e1);
        memcpy(&tot, h_small + so_tot, sizeof tot);
        gs.n_passes = tot.n_passes; gs.dp_cells = tot.dp_cells;
    } else {
        GP_CHK(hipMemcpyAsync(h_small, d_small, so_bytes, hipMemcpyDeviceToHost, st));
        GP_CHK(hipStreamSynchronize(st));
        if (out->inscns) memset(out->inscns, 0, (size_t)ns * 4 * inscns_cap);
    }
    const auto t_fin = std::chrono::steady_clock::now();
    memcpy(out->ret, h_small + so_ret, (size_t)ns * 4);
    memcpy(out->indel_types, h_small + so_types, (size_t)ns * 16);
    if (out->maxins) memcpy(out->maxins, h_small + so_maxins, (size_t)ns * 4);
    if (out->indelreg) memcpy(out->indelreg, h_small + so_ireg, (size_t)ns * 4);
    if (out->max_support) memcpy(out->max_support, h_small + so_msup, (size_t)ns * 4);
    if (out->max_frac) memcpy(out->max_frac, h_small + so_mfrac, (size_t)ns * 4);
    gs.finalize_ms += ms_since(t_fin);
    return BCFGPU_OK;
}

extern "C" int bcfgpu_gap_prep(bcfgpu_ctx *ctx, const bcfgpu_reads *rd, const bcfgpu_indel_in *in, const bcfgpu_indel_out *out,
                               int inscns_cap)
{
    if (!ctx || !rd || !in || !out || !out->ret || !out->p_aux || !out->indel_types || !in->ref)
        return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_gap_prep: bad arguments");
    hipStream_t st;
    if (bcfgpu_internal_device(ctx, &st, nullptr)) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_gap_prep: bad context");
    bcfgpu_gap_stats &gs = *bcfgpu_internal_gap_stats(ctx);
    gs = bcfgpu_gap_stats{};
    const auto t_begin = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t0) {
        return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    const int ns = in->n_sites, n = in->n_smpl, nr = rd->n_reads;
    if (ns <= 0) return BCFGPU_OK;
    if (n <= 0) return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_gap_prep: n_smpl");
    const size_t n_ent = (size_t)in->smpl_off[(size_t)ns * n];
    // extents of the pools (the caller hands over pointers, not lengths)
    size_t nbase = 0, ncig = 0;
    bool any_zq = false;
    for (int r = 0; r < nr; ++r) {
        const size_t e = (size_t)rd->r_seq_off[r] + rd->r_lq[r], c = (size_t)rd->r_cig_off[r] + rd->r_ncig[r];
        if (e > nbase) nbase = e;
        if (c > ncig) ncig = c;
        if (rd->r_has_zq && rd->r_has_zq[r] && rd->zq) any_zq = true;
    }
    if (nbase >> 31 || n_ent >> 31) return bcfgpu_set_error(BCFGPU_E_RANGE, "bcfgpu_gap_prep: batch too large (32-bit offsets), use fewer sites per call");
    // the slice of the reference the batch can touch: from the leftmost window to 64 kb past the rightmost position (the
    // scans of est_indelreg and of the homopolymer run stop at the end of the slice: exact unless a repeat is longer)
    int pmin = INT_MAX, pmax = 0;
    for (int i = 0; i < ns; ++i) { if (in->pos[i] < pmin) pmin = in->pos[i]; if (in->pos[i] > pmax) pmax = in->pos[i]; }
    const long ref_lo = pmin > 65536 ? pmin - 65536 : 0;
    const long ref_hi = pmax + 1 + (long)strnlen(in->ref + pmax + 1, 65536 + 4096);   // the caller guarantees ref[pos+1] exists (mpileup.c:341)

    // ---- inputs to HBM (queued on the stream; the kernels follow in order) ----
    int32_t *d_rpos = (int32_t*)WS(0, (size_t)nr * 4), *d_rlq = (int32_t*)WS(1, (size_t)nr * 4), *d_rflag = (int32_t*)WS(2, (size_t)nr * 4);
    int32_t *d_rncig = (int32_t*)WS(3, (size_t)nr * 4), *d_rcoff = (int32_t*)WS(4, (size_t)nr * 4), *d_rsoff = (int32_t*)WS(5, (size_t)nr * 4);
    uint32_t *d_cig = (uint32_t*)WS(6, ncig * 4);
    uint8_t *d_seq = (uint8_t*)WS(7, nbase), *d_qual = (uint8_t*)WS(8, nbase), *d_zq = any_zq ? (uint8_t*)WS(9, nbase) : nullptr;
    uint8_t *d_haszq = any_zq ? (uint8_t*)WS(10, (size_t)nr) : nullptr;
    int32_t *d_pos = (int32_t*)WS(11, (size_t)ns * 4), *d_soff = (int32_t*)WS(12, ((size_t)ns * n + 1) * 4);
    int32_t *d_pread = (int32_t*)WS(13, n_ent * 4), *d_pqpos = (int32_t*)WS(14, n_ent * 4), *d_pindel = (int32_t*)WS(15, n_ent * 4);
    char *d_ref = (char*)WS(24, (size_t)(ref_hi - ref_lo));
    uint32_t *d_aux = (uint32_t*)WS(27, n_ent * 4);
    if (!d_rpos || !d_rlq || !d_rflag || !d_rncig || !d_rcoff || !d_rsoff || !d_cig || !d_seq || !d_qual || (any_zq && (!d_zq || !d_haszq)) ||
        !d_pos || !d_soff || !d_pread || !d_pqpos || !d_pindel || !d_ref || !d_aux)
        return bcfgpu_set_error(BCFGPU_E_NOMEM, "bcfgpu_gap_prep: device workspace");
    #define UP(dst, src, bytes) GP_CHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st))
    UP(d_rpos, rd->r_pos, (size_t)nr * 4); UP(d_rlq, rd->r_lq, (size_t)nr * 4); UP(d_rflag, rd->r_flag, (size_t)nr * 4);
    UP(d_rncig, rd->r_ncig, (size_t)nr * 4); UP(d_rcoff, rd->r_cig_off, (size_t)nr * 4); UP(d_rsoff, rd->r_seq_off, (size_t)nr * 4);
    if (ncig) UP(d_cig, rd->cig, ncig * 4);
    if (nbase) { UP(d_seq, rd->seq16, nbase); UP(d_qual, rd->qual, nbase); }
    if (any_zq) { UP(d_zq, rd->zq, nbase); UP(d_haszq, rd->r_has_zq, (size_t)nr); }
    UP(d_pos, in->pos, (size_t)ns * 4); UP(d_soff, in->smpl_off, ((size_t)ns * n + 1) * 4);
    if (n_ent) { UP(d_pread, in->p_read, n_ent * 4); UP(d_pqpos, in->p_qpos, n_ent * 4); UP(d_pindel, in->p_indel, n_ent * 4); }
    UP(d_ref, in->ref + ref_lo, (size_t)(ref_hi - ref_lo));
    #undef UP
    GapIn g{};
    g.n_sites = ns; g.n_smpl = n; g.n_reads = nr;
    g.pos = d_pos; g.smpl_off = d_soff; g.p_read = d_pread; g.p_qpos = d_pqpos; g.p_indel = d_pindel;
    g.r_pos = d_rpos; g.r_lq = d_rlq; g.r_flag = d_rflag; g.r_ncig = d_rncig; g.r_cig_off = d_rcoff; g.r_seq_off = d_rsoff;
    g.cig = d_cig; g.seq16 = d_seq; g.qual = d_qual; g.zq = d_zq; g.r_has_zq = d_haszq;
    g.ref = d_ref; g.ref_lo = ref_lo; g.ref_hi = ref_hi;
    g.openQ = in->openQ; g.extQ = in->extQ; g.tandemQ = in->tandemQ; g.min_support = in->min_support; g.per_sample_flt = in->per_sample_flt;
    g.min_frac = in->min_frac;
    gs.prepare_ms = ms_since(t_begin);
    const int rc = bcfgpu_internal_gap_core(ctx, g, n_ent, d_aux, out, inscns_cap);
    if (rc) return rc;
    const auto t_dl = std::chrono::steady_clock::now();
    if (n_ent) {
        GP_CHK(hipMemcpyAsync(out->p_aux, d_aux, n_ent * 4, hipMemcpyDeviceToHost, st));
        GP_CHK(hipStreamSynchronize(st));
    }
    gs.finalize_ms += ms_since(t_dl);
    gs.total_ms = ms_since(t_begin);
    return BCFGPU_OK;
}
#undef GP_CHK
#undef WS

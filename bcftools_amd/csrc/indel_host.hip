// indel_host.hip -- bcfgpu_gap_prep: the host half of bcf_call_gap_prep (bam2bcf_indel.c:99-470) around the
// device realignment kernel of indel.hip.
//
//   prepare(site)  : :106-283  any indel? -> candidate types (sorted, unique, support filter, N filter), window,
//                    per-sample consensus with the two worst mismatch columns masked, homopolymer run, insertion
//                    consensus; then :291-345 one realignment job per (type, read): ref2 window, clipped query, capped quals
//   device         : :346-357  probaln_glocal forward with {1e-4,1e-2} and, if score>5, {1e-6,1e-3}
//   finalize(site) : :372-469  per-read indelQ/seqQ from the score gaps, the <=4 output types, remapped p->aux
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <cctype>
#include <cmath>
#include <vector>
#include <algorithm>
#include <chrono>
#include <thread>
#include <functional>
#include "kernels.h"

using namespace bcfgpu;

// provided by api.hip
extern "C" int bcfgpu_internal_upload_reads(bcfgpu_ctx *ctx, const bcfgpu_reads *rd, size_t nbase, bool any_zq);
extern "C" int bcfgpu_internal_run_probaln(bcfgpu_ctx *ctx, const std::vector<ProbalnPools> &pools, int max_bw,
                                           size_t nbase, bool any_zq,
                                           std::vector<int32_t> &score1, std::vector<int32_t> &score2);
int bcfgpu_set_error(int code, const char *what);
extern "C" bcfgpu_gap_stats *bcfgpu_internal_gap_stats(bcfgpu_ctx *ctx);

namespace {

const int MINUS_CONST = 0x10000000, INDEL_WINDOW_SIZE = 50, INDEL_NULL = 10000;

inline int nt16_of(char c)
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
const int nt16_int[16] = { 4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4 };

// bam2bcf_indel.c:40-66
int tpos2qpos(int cpos, int n_cigar, const uint32_t *cigar, int tpos, int is_left, int *_tpos)
{
    int x = cpos, y = 0, last_y = 0;
    *_tpos = cpos;
    for (int k = 0; k < n_cigar; ++k) {
        const int op = cigar[k] & 0xf, l = cigar[k] >> 4;
        if (op == 0 || op == 7 || op == 8) {
            if (cpos > tpos) return y;
            if (x + l > tpos) { *_tpos = tpos; return y + (tpos - x); }
            x += l; y += l; last_y = y;
        } else if (op == 1 || op == 4) y += l;
        else if (op == 2 || op == 3) {
            if (x + l > tpos) { *_tpos = is_left ? x : x + l; return y; }
            x += l;
        }
    }
    *_tpos = x;
    return last_y;
}

// bam2bcf_indel.c:69-75
inline int est_seqQ(const bcfgpu_indel_in *in, int l, int l_run)
{
    const int q = in->openQ + in->extQ * (std::abs(l) - 1);
    const int qh = l_run >= 3 ? (int)(in->tandemQ * (double)std::abs(l) / l_run + .499) : 1000;
    return q < qh ? q : qh;
}

// bam2bcf_indel.c:77-88
int est_indelreg(int pos, const char *ref, int l, const char *ins4)
{
    int max = 0, max_i = pos, score = 0;
    l = std::abs(l);
    for (int i = pos + 1, j = 0; ref[i]; ++i, ++j) {
        if (ins4) score += (toupper(ref[i]) != "ACGTN"[(int)ins4[j % l]]) ? -10 : 1;
        else score += (toupper(ref[i]) != toupper(ref[pos + 1 + j % l])) ? -10 : 1;
        if (score < 0) break;
        if (max < score) { max = score; max_i = i; }
    }
    return max_i - pos;
}

struct SiteState {
    bool live = false;
    int n_types = 0, ref_type = 0, l_run = 0, max_ins = 0, N = 0, indelreg = 0;
    std::vector<int> types;
    std::vector<char> inscns;
    size_t job0 = 0;                 // first job of this site
    std::vector<int32_t> jobidx;     // [K*n_types + t] -> job index or -1 (read skipped)
};

}  // namespace

extern "C" int bcfgpu_gap_prep(bcfgpu_ctx *ctx, const bcfgpu_reads *rd, const bcfgpu_indel_in *in, const bcfgpu_indel_out *out,
                               int inscns_cap)
{
    if (!ctx || !rd || !in || !out || !out->ret || !out->p_aux || !out->indel_types || !in->ref)
        return bcfgpu_set_error(BCFGPU_E_ARG, "bcfgpu_gap_prep: bad arguments");
    const int n = in->n_smpl;
    const char *ref = in->ref;
    bcfgpu_gap_stats &gs = *bcfgpu_internal_gap_stats(ctx);
    gs = bcfgpu_gap_stats{};
    const auto t_begin = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t0) {
        return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    std::vector<SiteState> st(in->n_sites);
    // The reads' base/quality pools do not depend on the typing below: a helper thread sends them to the device
    // meanwhile (pageable host memory: the copies occupy the calling thread).
    size_t nbase = 0;
    bool any_zq = false;
    for (int r = 0; r < rd->n_reads; ++r) {
        const size_t e = (size_t)rd->r_seq_off[r] + rd->r_lq[r];
        if (e > nbase) nbase = e;
        if (rd->r_has_zq && rd->r_has_zq[r] && rd->zq) any_zq = true;
    }
    if (nbase >> 32) return bcfgpu_set_error(BCFGPU_E_RANGE, "bcfgpu_gap_prep: read pool too large (32-bit offsets)");
    int upload_rc = 0;
    std::thread uploader([&]() { if (nbase) upload_rc = bcfgpu_internal_upload_reads(ctx, rd, nbase, any_zq); });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{uploader};
    // Sites are independent: contiguous chunks of sites are prepared by host threads into their own job pools, which
    // are then concatenated in site order (job and pool offsets rebased).
    typedef ProbalnPools Pools;
    auto prepare_range = [&](int is_begin, int is_end, Pools &PL) {
    std::vector<ProbalnJob> &jobs = PL.jobs;
    std::vector<uint8_t> &ref2pool = PL.ref2pool;
    int &max_bw = PL.max_bw;
    {   // one allocation instead of geometric regrowth: ~3 candidate types per read
        const size_t entries = (size_t)(in->smpl_off[(size_t)is_end * n] - in->smpl_off[(size_t)is_begin * n]);
        jobs.reserve(entries * 3);
    }
    for (int is = is_begin; is < is_end; ++is) {
        SiteState &S = st[is];
        const int32_t *soff = in->smpl_off + (size_t)is * n;
        const int pos = in->pos[is];
        out->ret[is] = -1;
        if (out->maxins) out->maxins[is] = 0;
        if (out->indelreg) out->indelreg[is] = 0;
        if (out->max_support) out->max_support[is] = 0;
        if (out->max_frac) out->max_frac[is] = 0;
        for (int t = 0; t < 4; ++t) out->indel_types[is * 4 + t] = INDEL_NULL;
        #define NPLP(s) (soff[(s) + 1] - soff[s])
        #define PE(s, i) (soff[s] + (i))
        // ---- is there a gap at all? (:106-113)
        bool any = false;
        for (int e = soff[0]; e < soff[n] && !any; ++e) any = in->p_indel[e] != 0;
        if (!any) continue;
        int N = soff[n] - soff[0];
        // ---- candidate types (:114-172)
        int max_rd_len = 0;
        {
            std::vector<uint32_t> aux; aux.reserve(N + 1);
            aux.push_back(MINUS_CONST);
            int n_alt = 0, n_tot = 0, indel_support_ok = 0;
            uint32_t max_support = 0; float max_frac = 0;
            for (int s = 0; s < n; ++s) {
                int na = 0, nt = 0;
                for (int i = 0; i < NPLP(s); ++i) {
                    const int e = PE(s, i), r = in->p_read[e];
                    ++nt;
                    if (in->p_indel[e] != 0) { ++na; aux.push_back(MINUS_CONST + in->p_indel[e]); }
                    int ql = 0;
                    const uint32_t *cg = rd->cig + rd->r_cig_off[r];
                    for (int k = 0; k < rd->r_ncig[r]; ++k) { const int op = cg[k] & 0xf; if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) ql += cg[k] >> 4; }
                    if (ql > max_rd_len) max_rd_len = ql;
                }
                const double frac = (double)na / nt;
                if (!indel_support_ok && na >= in->min_support && frac >= in->min_frac) indel_support_ok = 1;
                if (na > (int)max_support && frac > 0) { max_support = na; max_frac = frac; }
                n_alt += na; n_tot += nt;
            }
            if (out->max_support) out->max_support[is] = max_support;
            if (out->max_frac) out->max_frac[is] = max_frac;
            int nN = 0, i;
            for (i = pos; i - pos < max_rd_len && ref[i]; i++) if (ref[i] == 'N') nN++;
            if (nN * 2 > (i - pos)) continue;
            std::sort(aux.begin(), aux.end());
            int n_types = 1;
            for (size_t k = 1; k < aux.size(); ++k) if (aux[k] != aux[k - 1]) ++n_types;
            if (!in->per_sample_flt)
                indel_support_ok = ((double)n_alt / n_tot < in->min_frac || n_alt < in->min_support) ? 0 : 1;
            if (n_types == 1 || !indel_support_ok || n_types >= 64) continue;
            S.types.push_back((int)(aux[0] - MINUS_CONST));
            for (size_t k = 1; k < aux.size(); ++k) if (aux[k] != aux[k - 1]) S.types.push_back((int)(aux[k] - MINUS_CONST));
            S.n_types = n_types;
            for (S.ref_type = 0; S.ref_type < n_types; ++S.ref_type) if (S.types[S.ref_type] == 0) break;
        }
        std::vector<int> &types = S.types;
        const int n_types = S.n_types;
        // ---- window (:173-181)
        int left = pos > INDEL_WINDOW_SIZE ? pos - INDEL_WINDOW_SIZE : 0;
        int right = pos + INDEL_WINDOW_SIZE;
        if (types[0] < 0) right -= types[0];
        { int i; for (i = pos; i < right; ++i) if (ref[i] == 0) break; right = i; }
        // ---- per-sample consensus (:190-235)
        const int L = right - left + 1;
        std::vector<std::vector<char>> ref_sample(n, std::vector<char>(L, 0));
        {
            std::vector<uint32_t> cns(L);
            std::vector<char> ref0(L, 0);
            for (int i = 0; i < right - left; ++i) ref0[i] = (char)nt16_of(ref[i + left]);
            for (int s = 0; s < n; ++s) {
                std::vector<char> &r = ref_sample[s];
                std::fill(cns.begin(), cns.end(), 0u);
                for (int i = 0; i < NPLP(s); ++i) {
                    const int rdx = in->p_read[PE(s, i)];
                    const uint32_t *cigar = rd->cig + rd->r_cig_off[rdx];
                    const uint8_t *seq = rd->seq16 + rd->r_seq_off[rdx];
                    int x = rd->r_pos[rdx], y = 0;
                    for (int k = 0; k < rd->r_ncig[rdx]; ++k) {
                        const int op = cigar[k] & 0xf, l = cigar[k] >> 4;
                        if (op == 0 || op == 7 || op == 8) {
                            for (int j = 0; j < l; ++j)
                                if (x + j >= left && x + j < right)
                                    cns[x + j - left] += (seq[y + j] == (uint8_t)ref0[x + j - left]) ? 1 : 0x10000;
                            x += l; y += l;
                        } else if (op == 2 || op == 3) x += l;
                        else if (op == 1 || op == 4) y += l;
                    }
                }
                for (int i = 0; i < right - left; ++i) r[i] = ref0[i];
                uint32_t max = 0, max2 = 0; int max_i = -1, max2_i = -1;
                for (int i = 0; i < right - left; ++i) {
                    if (cns[i] >> 16 >= max >> 16) { max2 = max; max2_i = max_i; max = cns[i]; max_i = i; }
                    else if (cns[i] >> 16 >= max2 >> 16) { max2 = cns[i]; max2_i = i; }
                }
                if ((double)(max & 0xffff) / ((max & 0xffff) + (max >> 16)) >= 0.7) max_i = -1;
                if ((double)(max2 & 0xffff) / ((max2 & 0xffff) + (max2 >> 16)) >= 0.7) max2_i = -1;
                if (max_i >= 0) r[max_i] = 15;
                if (max2_i >= 0) r[max2_i] = 15;
            }
        }
        // ---- homopolymer run (:236-247)
        {
            const int c = nt16_of(ref[pos + 1]);
            if (c == 15) S.l_run = 1;
            else {
                int i;
                for (i = pos + 2; ref[i]; ++i) if (nt16_of(ref[i]) != c) break;
                S.l_run = i;
                for (i = pos; i >= 0; --i) if (nt16_of(ref[i]) != c) break;
                S.l_run -= i + 1;
            }
        }
        // ---- insertion consensus (:249-283)
        const int max_ins = types[n_types - 1];
        S.max_ins = max_ins;
        if (max_ins > 0) {
            std::vector<int> ia_all((size_t)5 * n_types * max_ins, 0);
            for (int t = 0; t < n_types; ++t) {
                if (types[t] <= 0) continue;
                for (int s = 0; s < n; ++s)
                    for (int i = 0; i < NPLP(s); ++i) {
                        const int e = PE(s, i), rdx = in->p_read[e];
                        if (in->p_indel[e] != types[t]) continue;
                        const uint8_t *seq = rd->seq16 + rd->r_seq_off[rdx];
                        for (int k = 1; k <= in->p_indel[e]; ++k) {
                            const int c = nt16_int[seq[in->p_qpos[e] + k] & 15];
                            ++ia_all[((size_t)t * max_ins + (k - 1)) * 5 + c];
                        }
                    }
            }
            S.inscns.assign((size_t)n_types * max_ins, 0);
            for (int t = 0; t < n_types; ++t)
                for (int j = 0; j < types[t]; ++j) {
                    int max = 0, max_k = -1; const int *ia = &ia_all[((size_t)t * max_ins + j) * 5];
                    for (int k = 0; k < 5; ++k) if (ia[k] > max) { max = ia[k]; max_k = k; }
                    S.inscns[(size_t)t * max_ins + j] = max ? (char)max_k : 4;
                    if (max_k == 4) { types[t] = 0; break; }      // discard insertions which contain N's
                }
        }
        // ---- realignment jobs (:284-345)
        const int max_ref2 = right - left + 2 + 2 * (max_ins > -types[0] ? max_ins : -types[0]);
        std::vector<char> ref2(max_ref2);
        S.N = N;
        struct QSeg { int left = -1, right = -1, qbeg_w = 0, qend_w = 0, tbeg = 0, tend = 0; };
        std::vector<QSeg> qseg(N);
        S.jobidx.assign((size_t)N * n_types, -1);
        S.job0 = jobs.size();
        S.indelreg = 0;
        for (int t = 0; t < n_types; ++t) {
            const int bw = std::abs(types[t]) + 3;
            int ir;
            if (types[t] == 0) ir = 0;
            else if (types[t] > 0) ir = est_indelreg(pos, ref, types[t], &S.inscns[(size_t)t * max_ins]);
            else ir = est_indelreg(pos, ref, -types[t], 0);
            if (ir > S.indelreg) S.indelreg = ir;
            int K = 0;
            for (int s = 0; s < n; ++s) {
                int k = 0, j;
                for (j = left; j <= pos; ++j) ref2[k++] = (char)nt16_int[(int)ref_sample[s][j - left]];
                if (types[t] <= 0) j += -types[t];
                else for (int l = 0; l < types[t]; ++l) ref2[k++] = S.inscns[(size_t)t * max_ins + l];
                for (; j < right && ref[j]; ++j) ref2[k++] = (char)nt16_int[(int)ref_sample[s][j - left]];
                for (; k < max_ref2; ++k) ref2[k] = 4;
                if (j < right) right = j;
                const uint32_t ref2_off = (uint32_t)ref2pool.size();
                ref2pool.insert(ref2pool.end(), ref2.begin(), ref2.end());
                for (int i = 0; i < NPLP(s); ++i, ++K) {
                    const int rdx = in->p_read[PE(s, i)];
                    const uint32_t *cigar = rd->cig + rd->r_cig_off[rdx];
                    const uint8_t *seq = rd->seq16 + rd->r_seq_off[rdx];
                    if (rd->r_flag[rdx] & 4) continue;
                    bool skipN = false;
                    for (int kk = 0; kk < rd->r_ncig[rdx]; ++kk) if ((cigar[kk] & 0xf) == 3) { skipN = true; break; }
                    if (skipN) continue;
                    // the CIGAR walks depend on the window only, which changes (shrinks) rarely: cached per read across types
                    QSeg &qc = qseg[K];
                    if (qc.left != left || qc.right != right) {
                        qc.left = left; qc.right = right;
                        qc.qbeg_w = tpos2qpos(rd->r_pos[rdx], rd->r_ncig[rdx], cigar, left, 0, &qc.tbeg);
                        qc.qend_w = tpos2qpos(rd->r_pos[rdx], rd->r_ncig[rdx], cigar, right, 1, &qc.tend);
                    }
                    int tbeg = qc.tbeg;
                    const int tend = qc.tend, qbeg = qc.qbeg_w, qend = qc.qend_w;
                    if (types[t] < 0) { const int l = -types[t]; tbeg = tbeg - l > left ? tbeg - l : left; }
                    ProbalnJob jb;
                    jb.ref_off = ref2_off + (uint32_t)(tbeg - left);
                    jb.l_ref = tend - tbeg + std::abs(types[t]);
                    jb.l_query = qend - qbeg;
                    jb.bw = bw;
                    {   // the band probaln_glocal will really use (probaln.c): min(bw, max(l_ref,l_query)), at least |l_ref-l_query|
                        int eff = jb.l_ref > jb.l_query ? jb.l_ref : jb.l_query;
                        if (eff > bw) eff = bw;
                        if (eff < std::abs(jb.l_ref - jb.l_query)) eff = std::abs(jb.l_ref - jb.l_query);
                        if (eff > max_bw) max_bw = eff;
                    }
                    // the query is read from the caller's seq16/qual(/ZQ) pools on the device, converted there
                    jb.query_off = (uint32_t)(rd->r_seq_off[rdx] + qbeg);
                    jb.flags = (rd->r_has_zq && rd->r_has_zq[rdx] && rd->zq) ? 1 : 0;
                    S.jobidx[(size_t)K * n_types + t] = (int32_t)jobs.size();
                    jobs.push_back(jb);
                }
            }
        }
        S.live = true;
        #undef NPLP
        #undef PE
    }
    };  // prepare_range

    int max_bw = 0;
    std::vector<Pools> pools;
    size_t n_jobs_total = 0;
    {
        int nthr = (int)std::thread::hardware_concurrency();
        if (const char *e = getenv("BCFGPU_HOST_THREADS")) nthr = atoi(e);
        nthr = std::max(1, std::min(std::min(nthr, 16), in->n_sites));
        pools.resize(nthr);
        auto cut = [&](int t) { return (int)((long)in->n_sites * t / nthr); };
        auto run_all = [&](auto fn) {
            std::vector<std::thread> thr;
            for (int t = 1; t < nthr; ++t) thr.emplace_back(fn, t);
            fn(0);
            for (auto &th : thr) th.join();
        };
        run_all([&](int t) { prepare_range(cut(t), cut(t + 1), pools[t]); });
        // pool bases in site order, then every thread rebases its own jobs and sites (no host-side concatenation:
        // the pools are uploaded segment by segment)
        std::vector<size_t> jb0(nthr + 1, 0), r0(nthr + 1, 0);
        for (int t = 0; t < nthr; ++t) {
            jb0[t + 1] = jb0[t] + pools[t].jobs.size(); r0[t + 1] = r0[t] + pools[t].ref2pool.size();
            if (pools[t].max_bw > max_bw) max_bw = pools[t].max_bw;
        }
        if (r0[nthr] >> 32 || jb0[nthr] >> 31)
            return bcfgpu_set_error(BCFGPU_E_RANGE, "bcfgpu_gap_prep: batch too large (pool offsets are 32-bit), use fewer sites per call");
        n_jobs_total = jb0[nthr];
        run_all([&](int t) {
            for (ProbalnJob &j : pools[t].jobs) j.ref_off += (uint32_t)r0[t];
            for (int is = cut(t); is < cut(t + 1); ++is) {
                SiteState &S = st[is];
                if (!S.live) continue;
                S.job0 += jb0[t];
                for (int32_t &j : S.jobidx) if (j >= 0) j += (int32_t)jb0[t];
            }
        });
    }

    gs.prepare_ms = ms_since(t_begin);
    // ---- device: forward scores of every job
    std::vector<int32_t> sc1, sc2;
    if (n_jobs_total) {
        const bool trace = getenv("BCFGPU_TRACE") != nullptr;               // diagnostics: host timeline on stderr
        uploader.join();
        if (upload_rc) return bcfgpu_set_error(upload_rc, "bcfgpu_gap_prep: read pool upload failed");
        if (trace) fprintf(stderr, "[gap_prep] prepare %.2f ms, read pools on device at %.2f ms\n", gs.prepare_ms, ms_since(t_begin));
        const int rc = bcfgpu_internal_run_probaln(ctx, pools, max_bw, nbase, any_zq, sc1, sc2);
        if (rc) return rc;
        gs.n_jobs = n_jobs_total;
        if (trace) fprintf(stderr, "[gap_prep] scores back at %.2f ms (kernel %.2f ms)\n", ms_since(t_begin), gs.kernel_ms);
        size_t j = 0;
        for (const Pools &pl : pools)
            for (const ProbalnJob &jb : pl.jobs) {
                const size_t jj = j++;
                if (jb.l_ref <= 0 || jb.l_query <= 0) continue;
                int bw = jb.l_ref > jb.l_query ? jb.l_ref : jb.l_query;
                if (bw > jb.bw) bw = jb.bw;
                if (bw < std::abs(jb.l_ref - jb.l_query)) bw = std::abs(jb.l_ref - jb.l_query);
                const uint64_t cells = (uint64_t)jb.l_query * (2 * bw + 1) * 3;
                const int passes = (sc1[jj] >> 8) > 5 ? 2 : 1;         // bam2bcf_indel.c:351
                gs.n_passes += passes; gs.dp_cells += cells * passes;
            }
    }
    pools.clear();
    if (getenv("BCFGPU_TRACE")) fprintf(stderr, "[gap_prep] finalize starts at %.2f ms\n", ms_since(t_begin));
    const auto t_fin = std::chrono::steady_clock::now();

    // ---- finalize (:372-469): per site, independent -> the same host threads
    auto finalize_range = [&](int is_begin, int is_end) {
    for (int is = is_begin; is < is_end; ++is) {
        SiteState &S = st[is];
        if (!S.live) continue;
        const int32_t *soff = in->smpl_off + (size_t)is * n;
        const int n_types = S.n_types, ref_type = S.ref_type;
        const std::vector<int> &types = S.types;
        std::vector<int> sc(n_types), sumq(n_types, 0);
        int K = 0;
        for (int e = soff[0]; e < soff[n]; ++e, ++K) {
            auto score = [&](const std::vector<int32_t> &tab, int t) { const int32_t j = S.jobidx[(size_t)K * n_types + t]; return j < 0 ? 0 : tab[j]; };
            int indelQ1, indelQ2, seqQ, tmp, t;
            for (t = 0; t < n_types; ++t) sc[t] = score(sc1, t) << 6 | t;
            for (t = 1; t < n_types; ++t) for (int j = t; j > 0 && sc[j] < sc[j - 1]; --j) std::swap(sc[j], sc[j - 1]);
            if ((sc[0] & 0x3f) == ref_type) {
                indelQ1 = (sc[1] >> 14) - (sc[0] >> 14);
                seqQ = est_seqQ(in, types[sc[1] & 0x3f], S.l_run);
            } else {
                for (t = 0; t < n_types; ++t) if ((sc[t] & 0x3f) == ref_type) break;
                indelQ1 = (sc[t] >> 14) - (sc[0] >> 14);
                seqQ = est_seqQ(in, types[sc[0] & 0x3f], S.l_run);
            }
            tmp = sc[0] >> 6 & 0xff;
            indelQ1 = tmp > 111 ? 0 : (int)((1. - tmp / 111.) * indelQ1 + .499);
            for (t = 0; t < n_types; ++t) sc[t] = score(sc2, t) << 6 | t;
            for (t = 1; t < n_types; ++t) for (int j = t; j > 0 && sc[j] < sc[j - 1]; --j) std::swap(sc[j], sc[j - 1]);
            if ((sc[0] & 0x3f) == ref_type) indelQ2 = (sc[1] >> 14) - (sc[0] >> 14);
            else {
                for (t = 0; t < n_types; ++t) if ((sc[t] & 0x3f) == ref_type) break;
                indelQ2 = (sc[t] >> 14) - (sc[0] >> 14);
            }
            tmp = sc[0] >> 6 & 0xff;
            indelQ2 = tmp > 111 ? 0 : (int)((1. - tmp / 111.) * indelQ2 + .499);
            int indelQ = indelQ1 < indelQ2 ? indelQ1 : indelQ2;
            if (indelQ > 255) indelQ = 255;
            if (seqQ > 255) seqQ = 255;
            out->p_aux[e] = (uint32_t)((sc[0] & 0x3f) << 16 | seqQ << 8 | indelQ);
            sumq[sc[0] & 0x3f] += indelQ < seqQ ? indelQ : seqQ;
        }
        if (out->maxins) out->maxins[is] = S.max_ins;
        for (int t = 0; t < n_types; ++t) sumq[t] = sumq[t] << 6 | t;
        for (int t = 1; t < n_types; ++t) for (int j = t; j > 0 && sumq[j] > sumq[j - 1]; --j) std::swap(sumq[j], sumq[j - 1]);
        int t;
        for (t = 0; t < n_types; ++t) if ((sumq[t] & 0x3f) == ref_type) break;
        if (t) { const int tmp = sumq[t]; for (; t > 0; --t) sumq[t] = sumq[t - 1]; sumq[0] = tmp; }
        int32_t *otypes = out->indel_types + (size_t)is * 4;
        for (t = 0; t < 4; ++t) otypes[t] = INDEL_NULL;
        for (t = 0; t < 4 && t < n_types; ++t) {
            otypes[t] = types[sumq[t] & 0x3f];
            if (out->inscns && S.max_ins > 0 && (t + 1) * S.max_ins <= 4 * inscns_cap)
                memcpy(out->inscns + (size_t)is * 4 * inscns_cap + (size_t)t * S.max_ins,
                       &S.inscns[(size_t)(sumq[t] & 0x3f) * S.max_ins], S.max_ins);
        }
        int n_alt = 0;
        for (int e = soff[0]; e < soff[n]; ++e) {
            const int x = types[out->p_aux[e] >> 16 & 0x3f];
            int j;
            for (j = 0; j < 4; ++j) if (x == otypes[j]) break;
            out->p_aux[e] = (uint32_t)(j << 16) | (j == 4 ? 0u : (out->p_aux[e] & 0xffff));
            if ((out->p_aux[e] >> 16 & 0x3f) > 0) ++n_alt;
        }
        if (out->indelreg) out->indelreg[is] = S.indelreg;
        out->ret[is] = n_alt > 0 ? 0 : -1;
    }
    };  // finalize_range
    {
        int nthr = (int)std::thread::hardware_concurrency();
        if (const char *e = getenv("BCFGPU_HOST_THREADS")) nthr = atoi(e);
        nthr = std::max(1, std::min(std::min(nthr, 16), in->n_sites));
        std::vector<std::thread> thr;
        auto cut = [&](int t) { return (int)((long)in->n_sites * t / nthr); };
        for (int t = 1; t < nthr; ++t) thr.emplace_back(finalize_range, cut(t), cut(t + 1));
        finalize_range(cut(0), cut(1));
        for (auto &th : thr) th.join();
    }
    gs.finalize_ms = ms_since(t_fin);
    gs.total_ms = ms_since(t_begin);
    return BCFGPU_OK;
}

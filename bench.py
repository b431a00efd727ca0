#!/usr/bin/env python3
"""bench.py -- throughput of the mpileup -> call -m hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path (glfgen+errmod -> combine -> mcall, PL/QS kept in
HBM) over one tile of synthetic pileup columns that is already resident in HBM.  The default
workload is the one BASELINE.json's metric is quoted on: 1000 samples x 30x, a tile of
--sites columns per rank (weak scaling: every rank owns its own genomic region shard; the only
collective is the ordered gather of the per-site call records to rank 0).

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for the roofline accounting).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# (bcftools_amd/csrc/api.hip: the stages' side streams need hardware queues of their own; the library sets this when it is loaded, but the
# runtime reads it at the process's first HIP call, which here may be torch's)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--samples", type=int, default=1000)
    ap.add_argument("--depth", type=float, default=30.0)
    ap.add_argument("--sites", type=int, default=None, help="pileup columns per tile (= per step, per rank); default 32768 (snp), 128 (indel), 32 (baq)")
    ap.add_argument("--var-rate", type=float, default=0.01)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline budget on rank 0 at N=1 (0: skip)")
    ap.add_argument("--seed", type=int, default=20260104)
    ap.add_argument("--packed", type=int, default=1, help="--mode pileup: 1 = the pool as bcfgpu_pileup_packed takes it (4-bit bases, palette qualities), 0 = a byte per base")
    ap.add_argument("--baq", type=int, default=0, help="--mode pileup: 1 = BAQ on the pool in HBM between the upload and the pileup (bcfgpu_pool_upload -> bcfgpu_pool_baq -> bcfgpu_pool_pileup)")
    ap.add_argument("--pageable", action="store_true", help="--mode pileup: keep the read pool in ordinary (pageable) host memory")
    ap.add_argument("--cpu-all-cores", type=int, default=1, help="also time the CPU baseline on all host cores (0: skip)")
    ap.add_argument("--groups", type=int, default=1, help="snp mode: call -G with this many sample groups (frequencies from FORMAT/AD); "
                                                          ">1 takes the general caller path (BASELINE configs[4] shape)")
    ap.add_argument("--haploid-frac", type=float, default=0.0, help="snp mode: fraction of haploid samples (ploidy array; general caller path)")
    ap.add_argument("--extras", type=int, default=1, help="snp mode at N=1: also measure the BASELINE configs[4] shape (sample groups + ploidy array) and "
                                                          "the indel stage (configs[2] shape) in child processes and embed their results under \"extra\" (0: skip)")
    ap.add_argument("--indel-read-rate", type=float, default=0.005, help="wgs mode: fraction of reads that carry a noise indel (SURVEY 8d: 0.5 %%)")
    ap.add_argument("--true-indel-rate", type=float, default=0.01, help="wgs mode: true indel sites per column")
    ap.add_argument("--fixed-depth", type=int, default=0, help="snp mode (experiment): 1 = every cell exactly --depth reads deep instead of Poisson(--depth)")
    ap.add_argument("--long-indel-frac", type=float, default=0.05, help="wgs mode: fraction of the indels (noise and true) that are 8-40 bases long")
    ap.add_argument("--indel-callers", type=int, default=1, help="indel mode: also time the host-pointer form of the stage on one 32-column batch (0: skip)")
    ap.add_argument("--mode", choices=["snp", "indel", "baq", "pileup", "gvcf", "mixed", "wgs"], default="snp",
                    help="snp: the headline pipeline (default).  indel: bcf_call_gap_prep on synthetic indel-candidate columns "
                         "(BASELINE configs[2] shape, 500 samples), reports DP cells/s of the realignment kernel.  "
                         "baq: bcfgpu_baq (sam_prob_realn) over the reads of the same synthetic columns.  "
                         "gvcf: bcfgpu_gvcf_blocks (gvcf_write) over the mpileup-stage planes of a tile resident in HBM")
    return ap.parse_args()


def host_cores():
    """Cores this process may really use: the cgroup CPU quota if there is one, else the affinity mask, and never more
    than 16 per GPU (the share of a one-GPU box)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, 16))


def algorithmic_bytes(n_sites, n_smpl, n_reads, A=2):
    """B_in = S*D*6 + S*6, B_out = S*(4G + 12A + 16) + 128 per site, with the tile's
    actual read count in place of S*D and A=2 (REF + <*>), G=3."""
    G = A * (A + 1) // 2
    b_in = n_reads * 6 + n_sites * n_smpl * 6
    b_out = n_sites * (n_smpl * (4 * G + 12 * A + 16) + 128)
    return b_in + b_out


def main_indel(a):
    """Secondary measurement (indel stage unit; BASELINE configs[2] shape): bcf_call_gap_prep over the candidate
    columns of a region whose reads are already in HBM -- bcfgpu_pileup has run (untimed, like the SNP bench's resident
    tile), then every step is one bcfgpu_gap_prep_tile over all candidate columns plus the indel pass (bcfgpu_mpileup on the
    tile it returns).  `value` counts the gap_prep_tile calls only (wall clock, host side included)."""
    import torch
    from bcftools_amd import abi, synth, engine
    from bcftools_amd.lib import check
    from tests.helpers import indeldrv
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    S = 500 if a.samples == 1000 else a.samples
    n_sites = 128 if a.sites is None else a.sites
    b = synth.indel_batch(a.seed, n_sites, S, depth=a.depth)
    E = len(b["p_read"])
    ctx = engine.Context(abi.default_cfg(S, max_sites=n_sites, max_reads=int(E * 1.05) + 64))
    t0 = time.perf_counter()
    pool = indeldrv.DevicePool(ctx, b)
    ctx.sync()
    t_pileup = time.perf_counter() - t0
    steps = max(1, a.steps)
    pool.gap_prep_tile()                                                            # warm-up: module load, workspaces at full size
    tot = dict(jobs=0, passes=0, cells=0, kernel=0.0, prepare=0.0, finalize=0.0, total=0.0, wall=0.0, pass_glfgen=0.0, pass_combine=0.0)
    o, ob, res = ctx.alloc_mplp_out(n_sites)
    ctx.timing(True)
    got = None
    for _ in range(steps):
        ctx.sync()
        t0 = time.perf_counter()
        got, st, tile = pool.gap_prep_tile()
        tot["wall"] += time.perf_counter() - t0
        tot["jobs"] += st.n_jobs; tot["passes"] += st.n_passes; tot["cells"] += st.dp_cells
        tot["kernel"] += st.kernel_ms; tot["prepare"] += st.prepare_ms; tot["finalize"] += st.finalize_ms; tot["total"] += st.total_ms
        # the indel records themselves (mpileup.c:357-365): the tile gap_prep_tile left in HBM through glfgen (ref_base = -1)
        # and combine; kernel times from the library's HIP events
        check(ctx.L.bcfgpu_mpileup(ctx.h, C.byref(tile), C.byref(o)))
        ctx.sync()
        tmg = ctx.last_timing()
        tot["pass_glfgen"] += tmg["glfgen_ms"]; tot["pass_combine"] += tmg["combine_ms"]
    ctx._download(ob, res)
    n_live = int((got["ret"] == 0).sum())                      # the tile holds the columns with ret == 0, in order
    records = int((res.site["ret"][:n_live] == 0).sum())
    per = lambda k: tot[k] / steps
    cells_per_s = tot["cells"] / (tot["kernel"] * 1e-3)
    out = {"metric": "indel-candidate columns/sec through bcf_call_gap_prep (typing, consensus, realignment, indelQ: all device kernels), %d samples x %.0fx" % (S, a.depth),
           "value": n_sites / (per("wall")), "unit": "sites/s", "n_gpus": 1, "higher_is_better": True, "steps": steps,
           "dtype": "f64 pair-HMM forward", "data": "synthetic",
           "config": {"workload": "synthetic indel-candidate columns (BASELINE configs[2] shape): one bcfgpu_gap_prep_tile over %d columns of a "
                                  "region whose read pool bcfgpu_pileup left in HBM" % n_sites,
                      "samples": S, "depth": a.depth, "sites": n_sites, "pileup_entries": E, "region_columns": len(b["ref"]),
                      "pileup_ms_untimed": t_pileup * 1e3},
           "kernel": {"name": "probaln_exact_kernel<band, pass> (+ job decode, radix sort)", "jobs": tot["jobs"] // steps,
                      "forward_passes": tot["passes"] // steps, "dp_cells": tot["cells"] // steps,
                      "kernel_ms": per("kernel"), "dp_cells_per_s": cells_per_s,
                      "fp64_flop_per_s": cells_per_s * 6.0,
                      "fp64_frac_of_vector_peak": cells_per_s * 6.0 / 78.6e12, "fp64_frac_of_mul_add_peak": cells_per_s * 6.0 / 39.3e12,
                      "flop_note": "18 fp64 multiplies/adds per band position (3 cells) in the reference's operation order, no FMA "
                                   "(contraction would change the integer scores): the mul/add-only peak is half the 78.6 TFLOP/s vector FMA peak"},
           "host_ms": {"prepare": per("prepare"), "finalize": per("finalize"), "whole_call": per("wall") * 1e3, "library_total": per("total")},
           "indel_pass": {"records": records, "glfgen_indel_ms": per("pass_glfgen"), "combine_ms": per("pass_combine"),
                          "note": "glfgen_kernel<INDEL> + combine_kernel on the tile gap_prep_tile left in HBM (the columns with ret == 0), kernel times"}}
    # the host-pointer form of the same stage (every array over PCIe both ways), for comparison: 32-column batches
    if a.indel_callers > 0:
        hb = synth.indel_batch(a.seed, min(32, n_sites), S, depth=a.depth)
        hctx = engine.Context(abi.default_cfg(S, max_sites=32, max_reads=64))
        indeldrv.gap_prep_gpu(hctx, hb)
        t0 = time.perf_counter()
        _, hst = indeldrv.gap_prep_gpu(hctx, hb)
        th = time.perf_counter() - t0
        out["host_pointer_form"] = {"value": hb["n_sites"] / th, "unit": "sites/s", "whole_call_ms": th * 1e3, "prepare_ms": hst.prepare_ms,
                                    "note": "bcfgpu_gap_prep with host pointers, one 32-column batch: uploads of the reads and entries, download of p->aux"}
        hctx.close()
    if a.cpu_seconds > 0:
        # the oracle on one host core, first columns; results compared with the device path (entries matched through the pool order)
        cell = np.repeat(np.arange(n_sites * S), np.diff(b["smpl_off"]))
        dev2batch = pool.order[np.argsort(cell[pool.order], kind="stable")]
        gaux, _, gt = pool.gap_prep_tile(want_aux=True)
        keep = np.repeat(np.repeat(gaux["ret"] == 0, S), np.diff(b["smpl_off"]))      # the tile's entries: the columns with ret == 0
        aux_batch = np.zeros(E, np.uint32)
        aux_batch[dev2batch[keep]] = gaux["aux"][:gt.n_reads]
        chk = dict(got, aux=aux_batch)
        t0 = time.perf_counter()
        k = 0
        while k < n_sites and (k < 1 or time.perf_counter() - t0 < a.cpu_seconds):
            indeldrv.assert_site_equal(chk, k, indeldrv.gap_prep_oracle_site(b, k))
            k += 1
        tc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": k / tc, "unit": "sites/s", "cores": 1, "kind": "port",
                               "sample": "first %d columns, oracle orc_gap_prep on one host core, %.1f s "
                                         "(results compared with the device path)" % (k, tc)}
    print(json.dumps(out), flush=True)
    ctx.close()


def main_mixed(a):
    """BASELINE configs[2] end to end (100 k sites, 10 % of them indel sites, 500 samples), scaled to one step: the SNP path over
    a tile of T pileup columns (glfgen + combine + call -m + compaction of the variant records) AND the indel path over T/10
    candidate columns of a read pool resident in HBM (bcfgpu_gap_prep_tile and the indel pass of bcfgpu_mpileup on the tile it
    returns: the indel records as `mpileup` writes them).  value = T / wall time of a step: pileup columns per second with their indel
    records.  The CPU baseline runs the oracle on the first sites / columns of the same inputs, one core."""
    import torch
    from bcftools_amd import abi, synth, engine, host
    from bcftools_amd.lib import check
    from tests.helpers import indeldrv
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    dev = torch.device("cuda", 0)
    S = 500 if a.samples == 1000 else a.samples
    n_ind = 128 if a.sites is None else max(1, a.sites // 10)
    T = 10 * n_ind
    # ---- SNP side: a synthetic tile on the device ----
    tile = synth.torch_tile(a.seed, T, S, dev, depth=a.depth, var_rate=a.var_rate)
    R = tile["n_reads"]
    b = synth.indel_batch(a.seed, n_ind, S, depth=a.depth)
    E = len(b["p_read"])
    cfg = abi.default_cfg(S, max_sites=T, max_reads=max(R, int(E * 1.05)) + 64)
    ctx = engine.Context(cfg)
    L = ctx.L
    dt = abi.Tile()
    dt.n_sites, dt.is_indel, dt.n_reads = T, 0, R
    dt.ref16, dt.plp_off, dt.rd, dt.epos = (tile["ref16"].data_ptr(), tile["plp_off"].data_ptr(), tile["rd"].data_ptr(), tile["epos"].data_ptr())
    mo, mbufs, _ = ctx.alloc_mplp_out(T, ctx.flagged_planes())
    co, cbufs, _ = ctx.alloc_call_out(T, abi.MAX_PL)
    rec_cap = 64 << 20
    recbuf = torch.empty(rec_cap, dtype=torch.uint8, device=dev)
    counts = torch.zeros(4, dtype=torch.int64, device=dev)
    # ---- indel side: the pool in HBM (bcfgpu_pileup, untimed like the resident SNP tile), outputs of the indel pass and its calls ----
    pool = indeldrv.DevicePool(ctx, b)
    imo, imb, ires = ctx.alloc_mplp_out(n_ind, ctx.flagged_planes())
    ctx.sync()

    def step():
        check(L.bcfgpu_pipeline(ctx.h, C.byref(dt), None, None, C.byref(mo), C.byref(co)))
        check(L.bcfgpu_compact_calls_async(ctx.h, T, 0, mo.site, C.byref(co), abi.MAX_PL, 2, recbuf.data_ptr(), rec_cap, counts.data_ptr()))
        got, st, itile = pool.gap_prep_tile()
        check(L.bcfgpu_mpileup(ctx.h, C.byref(itile), C.byref(imo)))
        return got, st
    for _ in range(max(1, a.warmup)):
        step()
    ctx.sync()
    steps = max(1, a.steps)
    t0 = time.perf_counter()
    prep = kern = 0.0
    for _ in range(steps):
        got, st = step()
        prep += st.prepare_ms; kern += st.kernel_ms
    ctx.sync()
    t1 = time.perf_counter()
    nb_, nr_ = C.c_uint64(), C.c_uint32()
    check(L.bcfgpu_compact_counts(ctx.h, counts.data_ptr(), C.byref(nb_), C.byref(nr_)))
    ctx._download(imb, ires)
    n_live = int((got["ret"] == 0).sum())                      # the indel tile holds the columns with ret == 0, in order
    per_step = (t1 - t0) / steps
    out = {"metric": "pileup columns/sec with their indel records (mpileup | call -m, 10 %% indel sites), %d samples x %.0fx" % (S, a.depth),
           "value": T / per_step, "unit": "sites/s", "n_gpus": 1, "steps": steps, "ms_per_step": per_step * 1e3, "higher_is_better": True,
           "dtype": "u8/i32 + f64 likelihood sums; f64 pair-HMM", "data": "synthetic",
           "config": {"workload": "BASELINE configs[2] shape scaled to one step: SNP path over %d columns + bcf_call_gap_prep and the indel pass over %d "
                                  "candidate columns, inputs resident in HBM" % (T, n_ind),
                      "samples": S, "depth": a.depth, "snp_columns": T, "indel_columns": n_ind, "snp_reads": R, "indel_pileup_entries": E,
                      "variant_records": int(nr_.value), "indel_records": int((ires.site["ret"][:n_live] == 0).sum())},
           "split_ms": {"gap_prep_realignment_kernels": kern / steps, "gap_prep_host_prepare": prep / steps},
           "note": "the indel path is the Amdahl term: %d candidate columns cost far more than %d SNP columns" % (n_ind, T)}
    if a.cpu_seconds > 0:
        from tests.helpers import orc
        ht = synth.tile_from_torch(tile)
        ns = min(T, 64)
        sub = host.HostTile(S, ht.ref16[:ns], ht.plp_off[: ns * S + 1], ht.rd[: ht.plp_off[ns * S]], ht.epos[: ht.plp_off[ns * S]])
        orc.mpileup(cfg, sub)
        c0 = time.perf_counter()
        m = orc.mpileup(cfg, sub)
        orc.mcall(cfg, host.CallInput(S, m.site["n_alleles"], np.maximum(m.site["unseen"], 0), m.pl.astype(np.int32), m.site["qsum"]))
        t_snp = (time.perf_counter() - c0) / ns
        c0 = time.perf_counter()
        k = 0
        while k < n_ind and (k < 1 or time.perf_counter() - c0 < a.cpu_seconds):
            indeldrv.gap_prep_oracle_site(b, k)
            k += 1
        t_ind = (time.perf_counter() - c0) / k
        out["cpu_baseline"] = {"value": 1.0 / (t_snp + 0.1 * t_ind), "unit": "sites/s", "cores": 1, "kind": "port",
                               "sample": "oracle on one host core: mpileup+mcall on the first %d columns (%.2f ms each), orc_gap_prep on the first %d "
                                         "candidate columns (%.1f ms each); per pileup column = SNP cost + a tenth of the indel cost" % (ns, t_snp * 1e3, k, t_ind * 1e3)}
    print(json.dumps(out), flush=True)
    ctx.close()


def main_wgs(a):
    """BASELINE configs[3] end to end on one GPU, scaled to one region: 1000 samples x 30x of READS (not a ready-made tile) over
    --sites columns, with sequencing / alignment indel noise on --indel-read-rate of the reads (0.5 %) and true indel
    sites every 1 / --true-indel-rate columns.  The read pool is resident in HBM (bcfgpu_pool_upload) and BAQ'd (mpileup's
    default), the columns are built there (bcfgpu_pool_pileup): `front_ms` -- then every timed step is what `mpileup | call -m`
    does per column: the SNP path over all columns (bcfgpu_pipeline + compaction of the variant records), candidate typing on
    every column where some read carries an indel (mpileup.c:354, bam2bcf_indel.c:106-188), realignment of all reads of the
    columns that pass -m / -F (bcfgpu_gap_prep_tile) and the indel pass (bcfgpu_mpileup on the tile it returns).
    value = columns / wall time of a step.  cpu_baseline: the oracle on the first columns / candidate columns of the same
    region, one host core; the candidates' p->aux is compared with the device's on the way."""
    import torch
    from bcftools_amd import abi, engine, host, synth
    from bcftools_amd.lib import check
    from tests.helpers import indeldrv, mplpdrv, orc
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    S = a.samples
    n_sites = 16384 if a.sites is None else a.sites            # host/bcfgpu_sam's default tile (--tile)
    W = synth.wgs_reads(a.seed, n_sites, S, a.depth, indel_read_rate=a.indel_read_rate, true_indel_rate=a.true_indel_rate,
                        long_indel_frac=a.long_indel_frac)
    L, beg, end, n, n_true = W["read_len"], W["beg"], W["end"], W["n_reads"], W["n_true"]
    arrs, mapq, smpl, refseq = W["reads"], W["mapq"], W["smpl"], W["refseq"]
    rd = abi.Reads()
    rd.n_reads = n
    for k_, v in arrs.items():
        setattr(rd, k_, v.ctypes.data)
    ref_b = refseq.encode()
    # ---- the front of the chain, once: pool -> HBM, BAQ, the columns ----
    ctx0 = engine.Context(abi.default_cfg(S, max_sites=1, max_reads=64))
    t = abi.Tile()
    check(ctx0.L.bcfgpu_pileup(ctx0.h, C.byref(rd), mapq.ctypes.data, smpl.ctypes.data, beg, end, ref_b, len(refseq), C.byref(t), None, None))
    entries = int(t.n_reads)
    ctx0.close()
    cfg = abi.default_cfg(S, max_sites=n_sites, max_reads=entries + 64)
    ctx = engine.Context(cfg)
    Lb = ctx.L
    tile = abi.Tile()
    col_n, col_indel = np.zeros(n_sites, np.int32), np.zeros(n_sites, np.uint8)

    def front():
        t0 = time.perf_counter()
        check(Lb.bcfgpu_pool_upload(ctx.h, C.byref(rd), None, mapq.ctypes.data))
        ctx.sync(); t1 = time.perf_counter()
        if a.baq:
            check(Lb.bcfgpu_pool_baq(ctx.h, ref_b, len(refseq), 3, None))
        ctx.sync(); t2 = time.perf_counter()
        check(Lb.bcfgpu_pool_pileup(ctx.h, smpl.ctypes.data, None, beg, end, ref_b, len(refseq), C.byref(tile), col_n.ctypes.data, col_indel.ctypes.data))
        ctx.sync(); t3 = time.perf_counter()
        return (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3
    front()
    f_up, f_baq, f_plp = front()
    cand = np.ascontiguousarray(np.nonzero((col_indel != 0) & (col_n < 250 * S))[0], dtype=np.int32)      # mpileup.c:354, -L 250
    nc = len(cand)
    mo, mbufs, _ = ctx.alloc_mplp_out(n_sites, ctx.flagged_planes())
    co, cbufs, _ = ctx.alloc_call_out(n_sites, abi.MAX_PL)
    rec_cap = 256 << 20
    recbuf = torch.empty(rec_cap, dtype=torch.uint8, device="cuda")
    irecbuf = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    counts = torch.zeros(4, dtype=torch.int64, device="cuda")
    icounts = torch.zeros(4, dtype=torch.int64, device="cuda")
    par = abi.IndelIn()
    par.ref = ref_b
    for k_, v in indeldrv.DEFAULTS.items():
        setattr(par, k_, v)
    CAP = indeldrv.CAP
    gout = dict(ret=np.zeros(max(nc, 1), np.int32), indel_types=np.zeros((max(nc, 1), 4), np.int32), inscns=np.zeros((max(nc, 1), 4 * CAP), np.int8),
                maxins=np.zeros(max(nc, 1), np.int32), indelreg=np.zeros(max(nc, 1), np.int32), max_support=np.zeros(max(nc, 1), np.int32),
                max_frac=np.zeros(max(nc, 1), np.float32))
    oo = abi.IndelOut()
    oo.ret, oo.indel_types, oo.inscns = gout["ret"].ctypes.data, gout["indel_types"].ctypes.data, gout["inscns"].ctypes.data
    oo.maxins, oo.indelreg, oo.max_support, oo.max_frac = (gout["maxins"].ctypes.data, gout["indelreg"].ctypes.data, gout["max_support"].ctypes.data,
                                                           gout["max_frac"].ctypes.data)
    itile = abi.Tile()
    st = abi.GapStats()
    split = dict(snp=0.0, gap=0.0, ipass=0.0)
    # the indel pass's outputs hold the columns bcf_call_gap_prep accepts (the indel tile's sites): counted by a first call
    n_acc = 1
    if nc:
        check(Lb.bcfgpu_gap_prep_tile(ctx.h, nc, cand.ctypes.data, None, C.byref(par), C.byref(oo), CAP, C.byref(itile)))
        n_acc = max(1, int(itile.n_sites))
    imo, imb, ires = ctx.alloc_mplp_out(n_acc, ctx.flagged_planes())
    ico, icb, icres = ctx.alloc_call_out(n_acc, abi.MAX_PL)

    def step(timed=False):
        t0 = time.perf_counter()
        check(Lb.bcfgpu_pipeline(ctx.h, C.byref(tile), None, None, C.byref(mo), C.byref(co)))
        check(Lb.bcfgpu_compact_calls_async(ctx.h, n_sites, 0, mo.site, C.byref(co), abi.MAX_PL, 2, recbuf.data_ptr(), rec_cap, counts.data_ptr()))
        if timed:
            ctx.sync()
        t1 = time.perf_counter()
        if nc:
            check(Lb.bcfgpu_gap_prep_tile(ctx.h, nc, cand.ctypes.data, None, C.byref(par), C.byref(oo), CAP, C.byref(itile)))
            if timed:
                ctx.sync()
            t2 = time.perf_counter()
            if itile.n_sites:
                # the indel records through call -m as well (mpileup.c:357-364 -> vcfcall.c:1137), the variant ones compacted
                assert itile.n_sites <= n_acc
                check(Lb.bcfgpu_pipeline(ctx.h, C.byref(itile), None, None, C.byref(imo), C.byref(ico)))
                check(Lb.bcfgpu_compact_calls_async(ctx.h, itile.n_sites, 0, imo.site, C.byref(ico), abi.MAX_PL, 2, irecbuf.data_ptr(), 64 << 20, icounts.data_ptr()))
        else:
            t2 = t1
        if timed:
            ctx.sync()
            split["snp"] += t1 - t0; split["gap"] += t2 - t1; split["ipass"] += time.perf_counter() - t2
    for _ in range(max(1, a.warmup)):
        step()
    ctx.sync()
    steps = max(1, a.steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.sync()
    per_step = (time.perf_counter() - t0) / steps
    for _ in range(2):
        step(True)
    check(Lb.bcfgpu_gap_prep_stats(ctx.h, C.byref(st)))
    nb_, nr_ = C.c_uint64(), C.c_uint32()
    check(Lb.bcfgpu_compact_counts(ctx.h, counts.data_ptr(), C.byref(nb_), C.byref(nr_)))
    inb_, inr_ = C.c_uint64(), C.c_uint32()
    if nc and itile.n_sites:
        check(Lb.bcfgpu_compact_counts(ctx.h, icounts.data_ptr(), C.byref(inb_), C.byref(inr_)))
    n_live = int((gout["ret"][:nc] == 0).sum()) if nc else 0
    assert n_live == (int(itile.n_sites) if nc else 0)
    if nc:
        ctx._download(imb, ires)
    n_irec = int((ires.site["ret"][:n_live] == 0).sum()) if nc else 0      # (the indel tile holds the columns with ret == 0, in order)
    out = {"metric": "pileup columns/sec with their indel records (mpileup | call -m), %d samples x %.0fx of reads with indel noise" % (S, a.depth),
           "value": n_sites / per_step, "unit": "sites/s", "n_gpus": 1, "steps": steps, "ms_per_step": per_step * 1e3, "higher_is_better": True,
           "dtype": "u8/i32 + f64 likelihood sums; f64 pair-HMM", "data": "synthetic",
           "config": {"workload": "BASELINE configs[3] shape on one GPU, one region: SNP path over every column + candidate typing on every column with an "
                                  "indel read + realignment of the columns that pass -m 1 -F 0.002 + the indel pass; read pool and columns resident in HBM",
                      "samples": S, "depth": a.depth, "columns": n_sites, "reads": n, "read_length": L, "pileup_entries": entries,
                      "indel_read_rate": a.indel_read_rate, "true_indel_sites": n_true, "baq": bool(a.baq),
                      "long_indel_frac": a.long_indel_frac,
                      "candidate_columns": nc, "realigned_columns": n_live, "indel_records": n_irec, "variant_records": int(nr_.value),
                      "indel_variant_records": int(inr_.value),
                      "realignment_jobs": int(st.n_jobs), "realignment_wide_band_jobs": int(st.n_wide), "realignment_passes": int(st.n_passes),
                      "realignment_jobs_by_band": dict(zip(["<=10", "11-15", "16-31", "32-43", "44-58", "59-73", "74-300", ">300"], [int(x) for x in st.band_jobs]))},
           "split_ms": {"snp_pipeline_and_compaction": split["snp"] / 2 * 1e3, "gap_prep_tile": split["gap"] / 2 * 1e3, "indel_pass_and_call": split["ipass"] / 2 * 1e3,
                        "realignment_kernels": float(st.kernel_ms)},
           "us_per_column": {"snp_pipeline_and_compaction": split["snp"] / 2 * 1e6 / n_sites, "gap_prep_tile": split["gap"] / 2 * 1e6 / n_sites,
                             "indel_pass_and_call": split["ipass"] / 2 * 1e6 / n_sites, "pool_upload_pcie": f_up * 1e3 / n_sites,
                             "pool_baq": f_baq * 1e3 / n_sites, "pool_pileup": f_plp * 1e3 / n_sites},
           "realignment": {"dp_cells": int(st.dp_cells), "kernel_ms": float(st.kernel_ms),
                           "fp64_frac_of_mul_add_peak": (float(st.dp_cells) * 6.0 / (float(st.kernel_ms) * 1e-3) / 39.3e12) if st.kernel_ms > 0 else None,
                           "note": "18 fp64 multiplies/adds per band position (3 cells), no FMA: peak 39.3 TFLOP/s; kernel_ms covers job decode, sort and all band classes"},
           "front_ms": {"pool_upload_pcie": f_up, "pool_baq": f_baq, "pool_pileup": f_plp,
                        "note": "once per region, before the timed steps (the steps start from columns resident in HBM, like the headline)"},
           "with_front": {"sites_per_s": n_sites / (per_step + (f_up + f_baq + f_plp) * 1e-3), "note": "columns / (step + upload + BAQ + pileup): a region from host reads to records"},
           "note": "the indel path is the Amdahl term of a large cohort: with %d samples nearly every column has some read with an indel" % S}
    if a.cpu_seconds > 0:
        # ---- the oracle on one core: the SNP path on the first columns (their tile downloaded), bcf_call_gap_prep on the first candidates ----
        ns = min(n_sites, 8)
        off = np.zeros(ns * S + 1, np.uint32)
        check(Lb.bcfgpu_memcpy_d2h(ctx.h, off.ctypes.data, tile.plp_off, off.nbytes))
        nr_h = int(off[-1])
        rdh, eph, r16 = np.zeros(nr_h, np.uint32), np.zeros(nr_h, np.uint8), np.zeros(ns, np.int8)
        check(Lb.bcfgpu_memcpy_d2h(ctx.h, rdh.ctypes.data, tile.rd, rdh.nbytes)); check(Lb.bcfgpu_memcpy_d2h(ctx.h, eph.ctypes.data, tile.epos, eph.nbytes))
        check(Lb.bcfgpu_memcpy_d2h(ctx.h, r16.ctypes.data, tile.ref16, r16.nbytes))
        sub = host.HostTile(S, r16, off, rdh, eph)
        orc.mpileup(cfg, sub)
        c0 = time.perf_counter()
        m = orc.mpileup(cfg, sub)
        orc.mcall(cfg, host.CallInput(S, m.site["n_alleles"], np.maximum(m.site["unseen"], 0), m.pl.astype(np.int32), m.site["qsum"]))
        t_snp = (time.perf_counter() - c0) / ns
        t_ind, k, t_rej, t_live = 0.0, 0, 0.0, 0.0
        if nc:
            # the pool as the device sees it after BAQ, and the candidates' pileup entries
            q_h, z_h, mq_h = np.zeros(n * L, np.uint8), np.zeros(n * L, np.uint8), np.zeros(n, np.uint8)
            check(Lb.bcfgpu_pool_download(ctx.h, q_h.ctypes.data, z_h.ctypes.data, mq_h.ctypes.data))
            # a bounded sample of both kinds of candidate column: ones the support filter turns away (cheap) and ones that are realigned
            rej_i = [int(x) for x in np.nonzero(gout["ret"][:nc] != 0)[0][:2]]
            live_i = [int(x) for x in np.nonzero(gout["ret"][:nc] == 0)[0][:2]]
            pick = sorted(rej_i + live_i)
            kmax = len(pick)
            cols = np.ascontiguousarray(cand[pick])
            tot_e = int(col_n[cols].sum())
            so = np.zeros(kmax * S + 1, np.int32); pr = np.zeros(tot_e, np.int32); pq = np.zeros(tot_e, np.int32); pi = np.zeros(tot_e, np.int32)
            check(Lb.bcfgpu_pileup_entries(ctx.h, kmax, cols.ctypes.data, so.ctypes.data, pr.ctypes.data, pq.ctypes.data, pi.ctypes.data, tot_e))
            has_zq = np.full(n, 1 if a.baq else 0, np.uint8)
            b = dict(n_sites=kmax, n_smpl=S, ref=ref_b, pos=(cols + beg).astype(np.int32), smpl_off=so, p_read=pr, p_qpos=pq, p_indel=pi,
                     reads=dict(arrs, n_reads=n, qual=q_h, zq=z_h, r_has_zq=has_zq))
            t_cls = {True: [], False: []}
            for k in range(kmax):
                c0 = time.perf_counter()
                want = indeldrv.gap_prep_oracle_site(b, k)
                t_cls[want is not None].append(time.perf_counter() - c0)
                gi_ = pick[k]
                assert (want is None) == (gout["ret"][gi_] != 0), "bcf_call_gap_prep: device and oracle disagree on candidate column %d" % gi_
                if want is not None:
                    assert np.array_equal(gout["indel_types"][gi_], want["indel_types"]), "indel types of candidate column %d" % gi_
            t_rej = float(np.mean(t_cls[False])) if t_cls[False] else 0.0
            t_live = float(np.mean(t_cls[True])) if t_cls[True] else 0.0
            t_ind = ((nc - n_live) * t_rej + n_live * t_live) / max(nc, 1)
            k = kmax
        frac = nc / n_sites
        out["cpu_baseline"] = {"value": 1.0 / (t_snp + frac * t_ind), "unit": "sites/s", "cores": 1, "kind": "port",
                               "sample": "oracle on one host core: mpileup+mcall on the first %d columns (%.1f ms each), orc_gap_prep on %d candidate columns -- "
                                         "turned away by the support filter %.1f ms each, realigned %.0f ms each (results compared with the device's); per column = "
                                         "SNP cost + %.3f x the mean candidate cost (%d of %d candidates are realigned)"
                                         % (ns, t_snp * 1e3, k, (t_rej if nc else 0.0) * 1e3, (t_live if nc else 0.0) * 1e3, frac, n_live, nc)}
    print(json.dumps(out), flush=True)
    ctx.close()


def main_baq(a):
    """Secondary measurement: BAQ (upstream of the pileup) over a pool of synthetic 100-bp reads."""
    import torch
    from bcftools_amd import abi, synth, engine
    from tests.helpers import mplpdrv
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    from bcftools_amd.lib import check
    ctx = engine.Context(abi.default_cfg(1, max_sites=1, max_reads=64))
    S = 500 if a.samples == 1000 else a.samples
    n_sites = 32 if a.sites is None else a.sites
    b = synth.indel_batch(a.seed, n_sites, S, depth=a.depth)
    R = b["reads"]
    rd = abi.Reads()
    rd.n_reads = R["n_reads"]
    for k in ("r_pos", "r_lq", "r_flag", "r_ncig", "r_cig_off", "r_seq_off", "cig", "seq16", "qual", "zq", "r_has_zq"):
        setattr(rd, k, R[k].ctypes.data)
    nb = len(R["qual"])
    qo, zo, ret = np.zeros(nb, np.uint8), np.zeros(nb, np.uint8), np.zeros(R["n_reads"], np.int32)

    def run():
        t0 = time.perf_counter()
        check(ctx.L.bcfgpu_baq(ctx.h, C.byref(rd), b["ref"], len(b["ref"]), 3, qo.ctypes.data, zo.ctypes.data, ret.ctypes.data))
        return time.perf_counter() - t0
    run()
    t = min(run() for _ in range(max(1, a.steps // 3)))
    out = {"metric": "reads/sec through BAQ (sam_prob_realn, extended, applied), %d-bp reads" % int(R["r_lq"][0]),
           "value": R["n_reads"] / t, "unit": "reads/s", "n_gpus": 1, "higher_is_better": True, "dtype": "f64 pair-HMM forward-backward",
           "data": "synthetic", "config": {"workload": "reads of %d synthetic indel-candidate columns x %d samples x %.0fx" % (n_sites, S, a.depth),
                                           "reads": int(R["n_reads"]), "bases": int(nb)},
           "whole_call_ms": t * 1e3, "note": "host pointers in and out: the time includes the window preparation on the host, "
                                             "uploads, baq_kernel and the download of the new qualities"}
    # the same stage on the pool kept in HBM (bcfgpu_pool_upload once, then bcfgpu_pool_baq on that copy): what a caller that
    # chains BAQ -> overlaps -> pileup on the device pays for BAQ
    mapq = np.full(R["n_reads"], 60, np.uint8)

    def run_pool():
        check(ctx.L.bcfgpu_pool_upload(ctx.h, C.byref(rd), None, mapq.ctypes.data))
        ctx.sync()
        t0 = time.perf_counter()
        check(ctx.L.bcfgpu_pool_baq(ctx.h, b["ref"], len(b["ref"]), 3, None))
        ctx.sync()
        return time.perf_counter() - t0
    run_pool()
    tpool = min(run_pool() for _ in range(max(1, a.steps // 3)))
    qp, zp = np.zeros(nb, np.uint8), np.zeros(nb, np.uint8)
    check(ctx.L.bcfgpu_pool_download(ctx.h, qp.ctypes.data, zp.ctypes.data, None))
    assert np.array_equal(qp, qo) and np.array_equal(zp, zo)
    lq = int(R["r_lq"][0])
    rows_kept = (lq + 1) // 2 + 1                                   # the odd forward rows (the even ones are re-formed by the backward pass)
    scratch = 2 * rows_kept * 32 * 8                                # M' and I' of 16 band positions per kept row: written once, read once
    # fp64 work per read, counted as the reference's multiplies and adds (no FMA: contraction would change the rounding the ZQ bytes and
    # the new qualities are compared on): forward 19 per band cell (M 6, I 4, D 3, the row sum 3, the scale 3), backward 20 (M 6, I 3, D 4,
    # the scale 3, the posterior 4), an even row re-formed 24 per cell: 16 cells x (19 + 20 + 12) per row
    flops = lq * 16 * (19 + 20 + 12)
    out["pool_form"] = {"stage_ms": tpool * 1e3, "value": R["n_reads"] / tpool, "unit": "reads/s",
                        "roofline": {"bound": "hbm", "bytes_per_read": scratch + 4 * lq, "achieved": R["n_reads"] * (scratch + 4 * lq) / tpool / 1e9,
                                     "peak": 8000.0, "unit": "GB/s", "frac": R["n_reads"] * (scratch + 4 * lq) / tpool / 8e12,
                                     "algorithmic_bytes_per_read": 4 * lq, "scratch_over_algorithmic": scratch / (4.0 * lq),
                                     "fp64_flop_per_s": R["n_reads"] * flops / tpool,
                                     "fp64_frac_of_vector_peak": R["n_reads"] * flops / tpool / 78.6e12,
                                     "fp64_frac_of_mul_add_peak": R["n_reads"] * flops / tpool / 39.3e12},
                        "note": "bcfgpu_pool_baq on the pool in HBM (window and band per read on the device; of the forward rows only the odd ones "
                                "go through the [row][cell][lane] scratch, unscaled, the backward pass re-forms the even ones; backward rows in "
                                "registers), wall time of the call incl. its one wait; results equal to the host-pointer call.  bytes_per_read is "
                                "the kernel's OWN row traffic plus the read's input and output (algorithmic_bytes_per_read)"}
    # the mate-overlap tweak over the same pool: consecutive reads of the pool taken as mates (reads of one column overlap
    # around it), whole call with host pointers; the C oracle on one core beside it, results compared
    npair = R["n_reads"] // 2
    pa = np.arange(0, 2 * npair, 2, dtype=np.int32)
    pb = pa + 1
    ov = np.zeros(nb, np.uint8)

    def run_ov():
        t0 = time.perf_counter()
        check(ctx.L.bcfgpu_overlap_tweak(ctx.h, C.byref(rd), npair, pa.ctypes.data, pb.ctypes.data, ov.ctypes.data))
        return time.perf_counter() - t0
    run_ov()
    tov = min(run_ov() for _ in range(3))
    from tests.helpers import orc
    OL = orc.lib()
    OL.orc_overlap_tweak.restype = C.c_int
    OL.orc_overlap_tweak.argtypes = [C.POINTER(abi.Reads), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    ow = R["qual"].copy()
    t0 = time.perf_counter()
    OL.orc_overlap_tweak(C.byref(rd), npair, pa.ctypes.data, pb.ctypes.data, ow.ctypes.data)
    toc = time.perf_counter() - t0
    assert np.array_equal(ov, ow)
    out["overlap_tweak"] = {"pairs": int(npair), "value": npair / tov, "unit": "pairs/s", "whole_call_ms": tov * 1e3,
                            "cpu_baseline": {"value": npair / toc, "unit": "pairs/s", "cores": 1, "kind": "port"},
                            "note": "bcfgpu_overlap_tweak with host pointers (uploads, overlap_kernel, download) against "
                                    "oracle/overlap.c on one core; outputs identical"}
    if a.cpu_seconds > 0:
        class Rd:
            pass
        nt = "=ACMGRSVTWYHKDBN"
        refseq = b["ref"].decode()
        k, t0 = 0, time.perf_counter()
        while k < R["n_reads"] and (k < 10 or time.perf_counter() - t0 < a.cpu_seconds):
            o, n = int(R["r_seq_off"][k]), int(R["r_lq"][k])
            r = Rd()
            r.pos, r.l_qseq, r.flag = int(R["r_pos"][k]), n, int(R["r_flag"][k])
            r.bamcigar = R["cig"][R["r_cig_off"][k]:R["r_cig_off"][k] + R["r_ncig"][k]].copy()
            r.seq = "".join(nt[c] for c in R["seq16"][o:o + n])
            r.qual = R["qual"][o:o + n].astype(np.int32)
            r.zq = None
            mplpdrv.apply_baq(r, refseq, 3)
            assert ret[k] == 0 and np.array_equal(r.qual, qo[o:o + n]) and np.array_equal(r.zq, zo[o:o + n])
            k += 1
        tc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": k / tc, "unit": "reads/s", "cores": 1, "kind": "port",
                               "sample": "first %d reads, oracle orc_sam_prob_realn on one host core incl. the Python call overhead, "
                                         "%.1f s (results compared with the device path)" % (k, tc)}
    print(json.dumps(out), flush=True)
    ctx.close()


def main_gvcf(a):
    """Secondary measurement: the gVCF block merging of `mpileup --gvcf` over one tile's planes in HBM."""
    import torch
    from bcftools_amd import abi, engine, host
    from bcftools_amd.lib import check
    from tests.helpers import orc
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    n, S = (32768 if a.sites is None else a.sites), a.samples
    rng = np.random.default_rng(a.seed)
    res = host.MplpResult(n, S)
    var = rng.random(n) < a.var_rate                                  # variant sites keep their records
    res.site["n_alleles"] = np.where(var, 3, 2)
    res.site["unseen"] = res.site["n_alleles"] - 1
    # per-sample depth: Poisson around a slowly drifting coverage, so that the smallest DP of a site crosses the limits
    cov = np.clip(a.depth + np.cumsum(rng.normal(0, 0.5, n)), 4, 200)
    tot = np.minimum(rng.poisson(cov[:, None], (n, S)), 255)
    fwd = rng.binomial(tot, 0.5)
    res.dp4[:, 0, :], res.dp4[:, 1, :] = fwd, tot - fwd
    res.pl[:, 1, :] = np.minimum(3 * tot, 255)
    res.pl[:, 2, :] = np.minimum(20 * tot + rng.integers(0, 40, (n, S)), 255)
    pos = np.arange(n, dtype=np.int32)
    ranges = np.array([1, 5, 10, 15, 20, 25, 30, 40, 60], dtype=np.int32)     # a typical --gvcf list
    ctx = engine.Context(abi.default_cfg(S))
    d = {k: ctx.to_device(v) for k, v in (("pos", pos), ("site", res.site), ("pl", res.pl), ("dp4", res.dp4))}
    o = {k: ctx.buf(b) for k, b in (("blk", n * 4), ("min_dp", n * 4), ("block", n * 24), ("dp", n * S * 4), ("pl", n * 3 * S))}
    gi, go, nb = abi.GvcfIn(), abi.GvcfOut(), C.c_int32(0)
    gi.n_sites, gi.n_range, gi.dp_range = n, len(ranges), ranges.ctypes.data
    gi.pos, gi.site, gi.pl, gi.dp4 = d["pos"].ptr, d["site"].ptr, d["pl"].ptr, d["dp4"].ptr
    go.blk, go.min_dp, go.block, go.dp, go.pl = (o[k].ptr for k in ("blk", "min_dp", "block", "dp", "pl"))

    def run():
        t0 = time.perf_counter()
        check(ctx.L.bcfgpu_gvcf_blocks(ctx.h, C.byref(gi), C.byref(go), C.byref(nb)))
        ctx.sync()
        return time.perf_counter() - t0
    for _ in range(max(1, a.warmup)):
        run()
    ts = [run() for _ in range(max(1, a.steps))]
    t = sum(ts) / len(ts)
    in_blocks = None
    blk = np.zeros(n, np.int32)
    o["blk"].download(blk)
    in_blocks = int((blk >= 0).sum())
    # algorithmic bytes: FORMAT/DP of every cell once for the range (4 B), DP and PL[1], PL[2] of the cells inside blocks once
    # for the reduction (6 B), the block's DP and PL out
    alg = n * S * 4 + in_blocks * S * 6 + nb.value * S * 8
    out = {"metric": "sites/sec through gvcf_write (gVCF block merging)", "value": n / t, "unit": "sites/s", "n_gpus": 1,
           "steps": len(ts), "warmup": a.warmup, "ms_per_step": t * 1e3, "higher_is_better": True, "dtype": "u8/int32", "data": "synthetic",
           "config": {"workload": "%d sites x %d samples, %.0fx, planes of the mpileup stage resident in HBM, --gvcf %s"
                                  % (n, S, a.depth, ",".join(map(str, ranges))), "blocks": nb.value, "sites_in_blocks": in_blocks},
           "roofline": {"bound": "hbm", "achieved": alg / t / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg / t / 8e12, "traffic": None},
           "note": "whole call incl. the host synchronisation for the block count (4 kernels + a hipcub scan)"}
    if a.cpu_seconds > 0:
        m = min(n, 4096)
        sub = host.MplpResult(m, S)
        sub.site[:], sub.pl[:], sub.dp4[:] = res.site[:m], res.pl[:m], res.dp4[:m]
        t0 = time.perf_counter()
        want = orc.gvcf_blocks(sub, pos[:m], ranges)
        tc = time.perf_counter() - t0
        got = ctx.gvcf_blocks(sub, pos[:m], ranges)
        assert got.n_blocks == want.n_blocks and np.array_equal(got.blk, want.blk) and np.array_equal(got.dp, want.dp) and np.array_equal(got.pl, want.pl)
        out["cpu_baseline"] = {"value": m / tc, "unit": "sites/s", "cores": 1, "kind": "port",
                               "sample": "first %d sites, oracle/gvcf.c on one host core incl. the widening of the planes to int32 "
                                         "(results compared with the device path)" % m}
    print(json.dumps(out), flush=True)
    ctx.close()


def main_pileup(a):
    """Secondary measurement (the pileup engine): bcfgpu_pileup builds the tile in HBM from a pool of reads
    (host pointers: the pool crosses PCIe, the tile does not), then the pipeline runs on it."""
    import torch
    from bcftools_amd import abi, engine
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    from bcftools_amd.lib import check
    S = a.samples
    n_sites = 16384 if a.sites is None else a.sites
    L, beg = 100, 200
    end = beg + n_sites
    rng = np.random.default_rng(a.seed)
    per = int((n_sites + L) * a.depth / L)                      # reads per sample
    n = per * S
    pos = np.sort(rng.integers(beg - L + 1, end, size=(S, per)), axis=1).astype(np.int32).ravel()
    smpl = np.repeat(np.arange(S, dtype=np.int32), per)
    # CIGARs: 100M, or (6 %) 48M2D52M / 40M3I57M / 5S95M
    kind = rng.choice(4, n, p=[0.94, 0.02, 0.02, 0.02])
    table = {0: [L << 4], 1: [48 << 4, 2 << 4 | 2, 52 << 4], 2: [40 << 4, 3 << 4 | 1, 57 << 4], 3: [5 << 4 | 4, 95 << 4]}
    ncig = np.array([1, 3, 3, 2], np.int32)[kind]
    cig_off = np.concatenate([[0], np.cumsum(ncig)[:-1]]).astype(np.int32)
    cig = np.zeros(int(ncig.sum()), np.uint32)
    for k, ops in table.items():
        idx = np.nonzero(kind == k)[0]
        for j, op in enumerate(ops):
            cig[cig_off[idx] + j] = op
    ref_codes = rng.integers(0, 4, end + 2 * L).astype(np.uint8)
    refseq = "".join("ACGT"[i] for i in ref_codes)
    rd = abi.Reads()
    arrs = dict(r_pos=pos, r_lq=np.full(n, L, np.int32), r_flag=(rng.integers(0, 2, n) * 16).astype(np.int32), r_ncig=ncig,
                r_cig_off=cig_off, r_seq_off=(np.arange(n, dtype=np.int64) * L).astype(np.int32), cig=cig,
                seq16=None,
                qual=rng.choice(np.array([11, 25, 37, 40], np.uint8), n * L), zq=np.zeros(1, np.uint8), r_has_zq=np.zeros(n, np.uint8))
    # bases: the reference under the read (indel reads are not shifted: a few mismatches), 0.3 % errors
    seq = np.empty(n * L, np.uint8)
    step = 1 << 16
    for r0 in range(0, n, step):
        r1 = min(n, r0 + step)
        idx = np.maximum(pos[r0:r1, None], 0) + np.arange(L, dtype=np.int32)[None, :]
        b = ref_codes[idx]
        err = rng.random(b.shape) < 0.003
        b = np.where(err, (b + rng.integers(1, 4, b.shape)) & 3, b)
        seq[r0 * L:r1 * L] = (1 << b).astype(np.uint8).ravel()
    arrs["seq16"] = seq
    mapq = np.where(rng.random(n) < 0.92, 60, rng.integers(0, 60, n)).astype(np.uint8)
    pool_bytes = sum(v.nbytes for v in arrs.values()) + mapq.nbytes + smpl.nbytes
    # the same pool as BAM records hold it: two bases per byte, the (binned) qualities as palette indices, the samples' offsets
    palette = np.unique(arrs["qual"])
    qual_bits = 2 if len(palette) <= 4 else 4
    packed_arrs = dict(seq4=abi.pack_nibbles(seq), qual4=(abi.pack_crumbs if qual_bits == 2 else abi.pack_nibbles)(np.searchsorted(palette, arrs["qual"])),
                       smpl_off=(np.arange(S + 1, dtype=np.int64) * per).astype(np.int32))
    packed_arrs["recs"] = abi.read12(pos, arrs["r_lq"], ncig, arrs["r_flag"], mapq).view(np.uint8)      # the per-read fields, 12 bytes a read
    assert L % 4 == 0                                                # (records need every read at a multiple of four bases: dense here)
    packed_bytes = arrs["cig"].nbytes + sum(v.nbytes for v in packed_arrs.values())
    # the pool in page-locked memory (bcfgpu_host_alloc), as a host that parses its reads straight into such buffers has it:
    # the uploads are then DMA transfers that run beside the other context's kernels
    from bcftools_amd.lib import load
    Lib = load()
    pinned = []

    def pin(v):
        ptr = C.c_void_p()
        check(Lib.bcfgpu_host_alloc(max(v.nbytes, 1), C.byref(ptr)))
        pinned.append(ptr)
        w = np.ctypeslib.as_array((C.c_uint8 * max(v.nbytes, 1)).from_address(ptr.value))[:v.nbytes].view(v.dtype)
        w[...] = v
        return w
    if not a.pageable:
        arrs = {k: pin(v) for k, v in arrs.items()}
        packed_arrs = {k: pin(v) for k, v in packed_arrs.items()}
        mapq, smpl = pin(mapq), pin(smpl)
    rd.n_reads = n
    for k, v in arrs.items():
        setattr(rd, k, v.ctypes.data)
    pk = abi.Packed()
    pk.seq4, pk.qual4, pk.smpl_off, pk.recs = (packed_arrs[k].ctypes.data for k in ("seq4", "qual4", "smpl_off", "recs"))
    pk.n_bases, pk.n_cig, pk.qual_bits = n * L, len(cig), qual_bits
    for j, q in enumerate(palette):
        pk.palette[j] = int(q)
    # size the contexts from a first build
    ctx0 = engine.Context(abi.default_cfg(S, max_sites=1, max_reads=64))
    t = abi.Tile()
    ref_b = refseq.encode()

    use_packed = [bool(a.packed)]

    def build(ctx, tile):
        t0 = time.perf_counter()
        if a.baq:
            check(ctx.L.bcfgpu_pool_upload(ctx.h, C.byref(rd), C.byref(pk) if use_packed[0] else None, mapq.ctypes.data))
            check(ctx.L.bcfgpu_pool_baq(ctx.h, ref_b, len(refseq), 3, None))
            check(ctx.L.bcfgpu_pool_pileup(ctx.h, None if use_packed[0] else smpl.ctypes.data, pk.smpl_off if use_packed[0] else None, beg, end,
                                           ref_b, len(refseq), C.byref(tile), None, None))
        elif use_packed[0]:
            check(ctx.L.bcfgpu_pileup_packed(ctx.h, C.byref(rd), C.byref(pk), mapq.ctypes.data, None, beg, end, ref_b, len(refseq),
                                             C.byref(tile), None, None))
        else:
            check(ctx.L.bcfgpu_pileup(ctx.h, C.byref(rd), mapq.ctypes.data, smpl.ctypes.data, beg, end, ref_b, len(refseq),
                                      C.byref(tile), None, None))
        return time.perf_counter() - t0
    build(ctx0, t)
    entries = int(t.n_reads)
    ctx0.close()
    # two contexts (a stream and a workspace each): while one region's kernels run, the next region's pool is prepared and
    # uploaded on the other -- the region loop of mpileup.c:652-683 with the regions in flight two deep
    ctxs = [engine.Context(abi.default_cfg(S, max_sites=n_sites, max_reads=entries)) for _ in range(2)]
    tiles = [abi.Tile(), abi.Tile()]
    outs = []
    for c in ctxs:
        mo, mbufs, _ = c.alloc_mplp_out(n_sites, c.flagged_planes())
        co = abi.CallOut()
        csite = torch.zeros(n_sites * C.sizeof(abi.CallSite), dtype=torch.uint8, device="cuda")
        cgt = torch.zeros(n_sites * 2 * S, dtype=torch.int8, device="cuda")
        cpl = torch.zeros(n_sites * abi.MAX_PL * S, dtype=torch.int32, device="cuda")
        co.site, co.gt, co.pl, co.gq, co.gp = csite.data_ptr(), cgt.data_ptr(), cpl.data_ptr(), None, None
        outs.append((mo, mbufs, co, (csite, cgt, cpl)))

    def pipe(i):
        check(ctxs[i].L.bcfgpu_pipeline(ctxs[i].h, C.byref(tiles[i]), None, None, C.byref(outs[i][0]), C.byref(outs[i][2])))
    # one region at a time, each step waited for (the unoverlapped numbers)
    def serial():
        t0 = time.perf_counter(); build(ctxs[0], tiles[0]); ctxs[0].sync()     # (the call returns with its fill kernel enqueued)
        t1 = time.perf_counter(); pipe(0); ctxs[0].sync()
        return t1 - t0, time.perf_counter() - t1
    # the overlapped loop: K regions (the same pool stands for every region), contexts alternating
    K = max(4, a.steps)

    def overlapped():
        for i in range(2):
            build(ctxs[i], tiles[i]); pipe(i)
        for c in ctxs:
            c.sync()
        t0 = time.perf_counter()
        for k in range(K):
            i = k & 1
            ctxs[i].sync()                   # the region that used this context two steps ago is done (its records would be read here)
            build(ctxs[i], tiles[i])         # uploads, counts; returns with the fill kernel enqueued
            pipe(i)                          # enqueued behind it; runs while the next region is uploaded on the other context
        for c in ctxs:
            c.sync()
        return (time.perf_counter() - t0) / K
    other = None
    if a.packed:                             # the byte-per-base form of the same pool, for comparison
        use_packed[0] = False
        serial()
        tb_u, _ = min(serial() for _ in range(3))
        other = {"pool_bytes": int(pool_bytes), "whole_call_ms": tb_u * 1e3, "overlapped_ms_per_region": overlapped() * 1e3}
        other["sites_per_s"] = n_sites / (other["overlapped_ms_per_region"] * 1e-3)
        use_packed[0] = True
        pool_bytes = packed_bytes
    serial()
    tb, tp = min(serial() for _ in range(3))
    t_loop = overlapped()
    tile_bytes = entries * 5 + (n_sites * S + 1) * 4 + n_sites
    out = {"metric": "pileup entries/sec through bcfgpu_pileup (read pool -> site x sample x read tile in HBM), %d samples x %.0fx" % (S, a.depth),
           "value": entries / tb, "unit": "entries/s", "n_gpus": 1, "higher_is_better": True, "dtype": "u32/u8 records", "data": "synthetic",
           "config": {"workload": "%d reads of %d bp over %d columns x %d samples" % (n, L, n_sites, S), "reads": n, "entries": entries,
                      "columns": n_sites, "pool_memory": "pageable" if a.pageable else "page-locked (bcfgpu_host_alloc)",
                      "pool_form": ("bcfgpu_pileup_packed: 4-bit bases, %d-value quality palette (%d-bit), 12-byte read records, per-sample offsets" % (len(palette), qual_bits))
                                   if a.packed else "bcfgpu_pileup: one byte per base and per quality",
                      "stages": "bcfgpu_pool_upload -> bcfgpu_pool_baq (flag 3) -> bcfgpu_pool_pileup -> bcfgpu_pipeline" if a.baq else "pileup -> bcfgpu_pipeline"},
           "whole_call_ms": tb * 1e3, "tile_written_gbs": tile_bytes / tb / 1e9,
           "pcie": {"pool_bytes": int(pool_bytes), "tile_bytes": int(tile_bytes), "pool_gbs_in_call": pool_bytes / tb / 1e9,
                    "note": "the pool is what crosses PCIe; a host-packed tile of this region would be tile_bytes"},
           "host_fed_pipeline": {"pipeline_ms": tp * 1e3, "serial_sites_per_s": n_sites / (tb + tp),
                                 "overlapped_ms_per_region": t_loop * 1e3, "sites_per_s": n_sites / t_loop, "regions": K,
                                 "note": "bcfgpu_pileup (host pointers in) then bcfgpu_pipeline on the tile it left in HBM; overlapped = two "
                                         "contexts alternating, a region's kernels running while the next region is prepared and uploaded"}}
    if other:
        out["byte_per_base_form"] = other
    print(json.dumps(out), flush=True)
    for c, o in zip(ctxs, outs):
        c.release(list(o[1].values()))
        c.close()
    for ptr in pinned:
        Lib.bcfgpu_host_free(ptr)


def main():
    a = parse()
    if a.mode == "pileup":
        return main_pileup(a)
    if a.mode == "indel":
        return main_indel(a)
    if a.mode == "mixed":
        return main_mixed(a)
    if a.mode == "wgs":
        return main_wgs(a)
    if a.mode == "baq":
        return main_baq(a)
    if a.mode == "gvcf":
        return main_gvcf(a)
    import torch
    import torch.distributed as dist
    from bcftools_amd import abi, synth, engine, shard
    from bcftools_amd.lib import check

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL, one rank per GPU.  BCFGPU_BENCH_REHEARSE=1: a dry run of the N>1 code path on a box with fewer GPUs than
        # ranks (gloo, ranks share the devices) -- its number means nothing and the JSON line says so
        rehearse = os.environ.get("BCFGPU_BENCH_REHEARSE") == "1"
        dist.init_process_group("gloo" if rehearse else "nccl", rank=rank, world_size=world)
        if rehearse:
            local %= max(1, torch.cuda.device_count())
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    # 32768 columns x 1000 samples x 30x = 4.9 GB of reads per tile: twice the tile costs nothing in HBM and the two
    # one-wavefront-per-site kernels lose less to their last partial round (16384: 3.5, 32768: 3.7, 65536: 3.7 M sites/s)
    S, T = a.samples, (a.sites if a.sites is not None else 32768)
    # ---- synthetic tile, generated on the device; each rank owns a different region shard ----
    tile = synth.torch_tile(a.seed + rank, T, S, dev, depth=a.depth, var_rate=a.var_rate, fixed_depth=bool(a.fixed_depth))
    torch.cuda.synchronize()
    R = tile["n_reads"]

    cfg = abi.default_cfg(S, max_sites=T, max_reads=R, device=local, n_grp=a.groups,
                          fmt_flag=abi.INFO_VDB | abi.INFO_RPB | (abi.FMT_AD if a.groups > 1 else 0))
    ctx = engine.Context(cfg)
    d_ploidy = d_grp = None
    if a.haploid_frac > 0:
        d_ploidy = torch.where(torch.rand(S, device=dev) < a.haploid_frac, 1, 2).to(torch.uint8)
    if a.groups > 1:
        d_grp = (torch.arange(S, device=dev) * a.groups // S).to(torch.int32)
    L = ctx.L
    dt = abi.Tile()
    dt.n_sites, dt.is_indel, dt.n_reads = T, 0, R
    dt.ref16, dt.plp_off, dt.rd, dt.epos = (tile["ref16"].data_ptr(), tile["plp_off"].data_ptr(),
                                            tile["rd"].data_ptr(), tile["epos"].data_ptr())
    # outputs (device)
    mo, mbufs, _ = ctx.alloc_mplp_out(T, ctx.flagged_planes())
    co = abi.CallOut()
    csite = torch.zeros(T * C.sizeof(abi.CallSite), dtype=torch.uint8, device=dev)
    cgt = torch.zeros(T * 2 * S, dtype=torch.int8, device=dev)
    cpl = torch.zeros(T * abi.MAX_PL * S, dtype=torch.int32, device=dev)
    co.site, co.gt, co.pl, co.gq, co.gp = csite.data_ptr(), cgt.data_ptr(), cpl.data_ptr(), None, None

    # What leaves a shard is the records `call -mv` would write: compacted on the device (call record + mpileup site record +
    # the GT and PL planes of the variant sites, bcfgpu_compact_calls) and, for N > 1, gathered to rank 0 in rank order with
    # grouped send/recv -- the same two library calls host/bcfgpu_mgpu.c makes.
    # room for the records of a step: every site a variant with all PL planes would be T * (400 + 62 S) bytes; variant sites are
    # a few per cent of the tile, a quarter of that bound (at least 64 MiB) is held and an overflow is reported, never overrun
    rec_cap = max(64 << 20, (T * (512 + S * (2 + 4 * abi.MAX_PL))) // 4)
    if os.environ.get("BCFGPU_ABLATE"):                        # diagnostics build (tools/ablate_kernel.sh): a kernel with parts switched off
        rec_cap = T * (512 + S * (2 + 4 * abi.MAX_PL)) + 4096   # calls garbage, so every site may come out a variant
    n_bytes, n_rec = C.c_uint64(), C.c_uint32()
    # Two record buffers with their counters (bytes, records, overflow: on the device).  N = 1: nothing comes back to the host
    # inside the timed loop.  N > 1: the gather of step i-1 runs on a side stream beside the kernels of step i -- the only
    # host wait is for step i-1's byte count, while the device works on step i (the two library calls
    # host/bcfgpu_mgpu.c makes, bcfgpu_compact_calls* and the ordered gather).
    recbuf = [torch.empty(rec_cap, dtype=torch.uint8, device=dev) for _ in range(2 if world > 1 else 1)]
    counts = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in recbuf]
    rehearse = world > 1 and os.environ.get("BCFGPU_BENCH_REHEARSE") == "1"
    gathered = torch.empty(rec_cap * world, dtype=torch.uint8, device=dev) if (world > 1 and rank == 0 and not rehearse) else None
    gathered_host = torch.empty(rec_cap * world, dtype=torch.uint8) if (rehearse and rank == 0) else None
    work_stream = comm_stream = None
    if world > 1:
        work_stream = torch.cuda.Stream(device=dev)          # (the default stream has handle 0 = "the library's own stream")
        comm_stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(work_stream)
        check(L.bcfgpu_set_stream(ctx.h, C.c_void_p(work_stream.cuda_stream)))
        ev_done = [torch.cuda.Event() for _ in recbuf]       # compaction into buffer j queued up to here
        ev_free = [torch.cuda.Event() for _ in recbuf]       # the gather has read buffer j
    state = {"i": 0, "pending": None}

    def gather(j):
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ev_done[j])
            if rehearse:                                      # gloo moves host tensors: the dry run stages the records through the host
                nb_ = int(counts[j][0].item())
                sz = shard.gather_packed(recbuf[j][:nb_].cpu(), nb_, gathered_host, dst=0)
                if rank == 0:
                    state["gathered_bytes"] = sum(sz)
            else:
                sz = shard.gather_packed(recbuf[j], counts[j][0:1], gathered, dst=0)
                if rank == 0:
                    state["gathered_bytes"] = sum(sz)
            ev_free[j].record(comm_stream)

    def step():
        j = state["i"] % len(recbuf)
        if world > 1 and state["i"] >= len(recbuf):
            work_stream.wait_event(ev_free[j])                # the buffer's previous records have left
        check(L.bcfgpu_pipeline(ctx.h, C.byref(dt), d_ploidy.data_ptr() if d_ploidy is not None else None,
                                d_grp.data_ptr() if d_grp is not None else None, C.byref(mo), C.byref(co)))
        check(L.bcfgpu_compact_calls_async(ctx.h, T, rank * T, mo.site, C.byref(co), abi.MAX_PL, 2, recbuf[j].data_ptr(), rec_cap,
                                           counts[j].data_ptr()))
        if world > 1:
            ev_done[j].record(work_stream)
            if state["pending"] is not None:
                gather(state["pending"])
            state["pending"] = j
        state["i"] += 1

    def fence():
        if world > 1 and state["pending"] is not None:
            gather(state["pending"])
            state["pending"] = None
        check(L.bcfgpu_sync(ctx.h))
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    check(L.bcfgpu_timing_enable(ctx.h, 2))
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    t1 = time.perf_counter()
    tm = ctx.last_timing()
    check(L.bcfgpu_timing_enable(ctx.h, 0))
    check(L.bcfgpu_compact_counts(ctx.h, counts[(state["i"] - 1) % len(recbuf)].data_ptr(), C.byref(n_bytes), C.byref(n_rec)))   # (BCFGPU_E_RANGE on overflow)

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    total_sites = T * world * a.steps
    value = total_sites / elapsed

    out = None
    if rank == 0:
        alg = algorithmic_bytes(T, S, R)
        # HBM traffic of the dominant kernel: PMC-measured on a launch of this very shape (tools/profile.sh: rocprofv3 --pmc,
        # FETCH_SIZE x2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes for gfx950, separate passes), never scaled: a run
        # with another tile shape reports null
        traffic = traffic_source = None
        for tf in ("r5_traffic.json", "r4_traffic.json", "r3_traffic.json", "r2_traffic.json"):
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", tf)))
                if tj["samples"] == S and abs(tj["depth"] - a.depth) < 1e-9 and tj["sites"] == T:
                    traffic = tj["hbm_bytes_per_launch"]
                    traffic_source = "profiles/%s (builder-side tools/profile.sh run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this launch shape; not measured in this run)" % tf
                    break
            except Exception:
                pass
        kern_s = tm["glfgen_ms"] * 1e-3
        achieved = alg / kern_s / 1e9 if kern_s > 0 else 0.0
        out = {
            "metric": "variant sites/sec (mpileup|call -m), 1000-sample 30x WGS synthetic",
            "value": value, "unit": "sites/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/i32 + f64 likelihood sums", "data": "synthetic",
            **({"rehearsal": "BCFGPU_BENCH_REHEARSE=1: gloo, ranks share devices -- not a measurement"}
               if world > 1 and os.environ.get("BCFGPU_BENCH_REHEARSE") == "1" else {}),
            "config": {"workload": "1000-sample 30x synthetic WGS tile (BASELINE configs[3] shape), SNP path: "
                                   "glfgen+errmod -> combine -> call -m, inputs resident in HBM",
                       "samples": S, "depth": a.depth, "sites_per_step_per_gpu": T, "reads_per_tile": R,
                       "sharding": "contiguous region shard per GPU; the records call -mv would write are compacted on the device and "
                                   "gathered to rank 0 in rank order (grouped send/recv)",
                       "records_per_step_per_gpu": int(n_rec.value), "record_bytes_per_step_per_gpu": int(n_bytes.value),
                       **({"gathered_bytes_last_step": int(state.get("gathered_bytes", 0)),
                           "transport": "gloo (rehearsal, staged through the host)" if rehearse else "RCCL grouped send/recv (torch.distributed nccl backend), side stream"} if world > 1 else {}),
                       "groups": a.groups, "haploid_frac": a.haploid_frac},
            "roofline": {"bound": "hbm", "kernel": "glfgen_kernel", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": alg, "kernel_ms": tm["glfgen_ms"],
                         "other_kernels_ms": {"combine_kernel": tm["combine_ms"], "mcall_kernel": tm["mcall_ms"]},
                         # the two smaller kernels against the same roofline: bytes they must move per cell (DESIGN.md 3.2, 3.3;
                         # combine: 28 B of per-cell workspace + QS read twice in, PL + DP4 out)
                         "other_kernels_frac": {
                             "combine_kernel": (T * S * (28 + 8 + 15 + 4)) / (tm["combine_ms"] * 1e-3) / 8e12 if tm["combine_ms"] > 0 else None,
                             "mcall_kernel": (T * S * (15 + 2 + 12)) / (tm["mcall_ms"] * 1e-3) / 8e12 if tm["mcall_ms"] > 0 else None},
                         "other_kernels_gbs": {
                             "combine_kernel": (T * S * (28 + 8 + 15 + 4)) / (tm["combine_ms"] * 1e-3) / 1e9 if tm["combine_ms"] > 0 else None,
                             "mcall_kernel": (T * S * (15 + 2 + 12)) / (tm["mcall_ms"] * 1e-3) / 1e9 if tm["mcall_ms"] > 0 else None}},
        }
        # ---- CPU baseline: the oracle (a port of the reference's loops), 1 core, bounded sample ----
        if world == 1 and a.cpu_seconds > 0:
            from tests.helpers import orc
            from bcftools_amd import host
            ht = synth.tile_from_torch(tile)

            def cpu_run(ns):
                sub = host.HostTile(S, ht.ref16[:ns], ht.plp_off[: ns * S + 1], ht.rd[: ht.plp_off[ns * S]],
                                    ht.epos[: ht.plp_off[ns * S]])
                c0 = time.perf_counter()
                m = orc.mpileup(cfg, sub)
                cin = host.CallInput(S, m.site["n_alleles"], np.maximum(m.site["unseen"], 0), m.pl.astype(np.int32),
                                     m.site["qsum"])
                orc.mcall(cfg, cin)
                return time.perf_counter() - c0
            probe = min(T, 64)
            cpu_run(probe)                                   # first call: library load, errmod tables
            tp = cpu_run(probe)
            ns = int(max(probe, min(T, a.cpu_seconds / max(tp / probe, 1e-9))))
            tc = cpu_run(ns)
            model = ""
            try:
                model = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
            except Exception:
                pass
            out["cpu_baseline"] = {"value": ns / tc, "unit": "sites/s", "cores": 1, "kind": "port",
                                   "sample": "first %d sites of the same tile (%d samples x %.0fx), oracle/liboracle.so "
                                             "mpileup+mcall on one host core, %.1f s" % (ns, S, a.depth, tc),
                                   "cpu_model": model}
            # all host cores, region-sharded (how users scale the reference: one process per region, -r + concat)
            ncore = host_cores()
            if ncore > 1 and a.cpu_all_cores:
                import subprocess, tempfile, glob
                per = int(max(16, min(256, (ns / tc) * a.cpu_seconds / 2)))          # sites per worker tile ...
                reps = int(max(1, round((ns / tc) * a.cpu_seconds / 2 / per)))       # ... repeated to ~cpu_seconds/2 of work
                go = os.path.join(tempfile.mkdtemp(prefix="bcfgpu_cpu_"), "go")
                env = dict(os.environ, OMP_NUM_THREADS="1")
                procs = [subprocess.Popen([sys.executable, "-m", "tests.helpers.cpu_worker", str(a.seed + 1000 + i), str(per),
                                           str(S), str(a.depth), go, str(reps)], cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True)
                         for i in range(ncore)]
                t_wait = time.time()
                while len(glob.glob(go + ".ready.*")) < ncore and time.time() - t_wait < 300 and all(p.poll() is None for p in procs):
                    time.sleep(0.05)
                open(go, "w").close()
                res = [p.communicate()[0].split() for p in procs]
                if all(len(r) == 2 for r in res):
                    tmax = max(float(r[1]) for r in res)
                    # (scalar keys of cpu_baseline itself, so that a parser that keeps one level of it keeps them)
                    out["cpu_baseline"]["all_cores_value"] = sum(int(r[0]) for r in res) / tmax
                    out["cpu_baseline"]["all_cores_cores"] = ncore
                    out["cpu_baseline"]["all_cores"] = {"value": sum(int(r[0]) for r in res) / tmax, "unit": "sites/s",
                                                        "cores": ncore, "sample": "%d region shards of %d sites x %d passes, one "
                                                        "oracle process per core, %.1f s" % (ncore, per, reps, tmax)}
        # ---- the other BASELINE shapes, measured in child processes (their own contexts and tiles) ----
        if world == 1 and a.extras and a.groups == 1 and a.haploid_frac == 0.0:
            import subprocess

            def child(args, timeout=600):
                try:
                    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
                    for ln in reversed(r.stdout.strip().splitlines()):
                        if ln.startswith("{"):
                            return json.loads(ln)
                    return {"error": (r.stderr or r.stdout)[-300:]}
                except Exception as e:                       # a failed extra must not take the headline with it
                    return {"error": repr(e)}
            common = ["--cpu-seconds", "0", "--cpu-all-cores", "0", "--extras", "0", "--seed", str(a.seed)]
            c4 = child(["--groups", "4", "--haploid-frac", "0.25", "--steps", str(max(3, a.steps // 2)), "--warmup", "2",
                        "--samples", str(S), "--depth", str(a.depth), "--sites", str(T)] + common)
            ind = child(["--mode", "indel", "--steps", "4", "--cpu-seconds", "8", "--cpu-all-cores", "0", "--extras", "0", "--seed", str(a.seed)])
            mix = child(["--mode", "mixed", "--steps", "4", "--warmup", "1", "--cpu-seconds", "6", "--cpu-all-cores", "0", "--extras", "0", "--seed", str(a.seed)])
            hf = child(["--mode", "pileup", "--steps", "6", "--cpu-seconds", "0", "--cpu-all-cores", "0", "--extras", "0", "--seed", str(a.seed)])
            hfb = child(["--mode", "pileup", "--baq", "1", "--packed", "1", "--sites", "4096", "--steps", "4", "--cpu-seconds", "0", "--cpu-all-cores", "0",
                         "--extras", "0", "--seed", str(a.seed)])
            bq = child(["--mode", "baq", "--steps", "6", "--cpu-seconds", "5", "--cpu-all-cores", "0", "--extras", "0", "--seed", str(a.seed)])
            wgs = child(["--mode", "wgs", "--steps", "3", "--warmup", "1", "--cpu-seconds", "6", "--cpu-all-cores", "0", "--extras", "0", "--baq", "1",
                         "--samples", str(S), "--depth", str(a.depth), "--seed", str(a.seed)])
            # ---- HBM bytes of the dominant kernel, measured in this run: two child passes under rocprofv3 --pmc (FETCH_SIZE and
            # WRITE_SIZE apart, nothing but --kernel-trace beside them), corrected as MI355X_MICROARCH.md prescribes for gfx950 ----
            live = None
            try:
                import shutil, tempfile, csv, glob
                prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
                if prof:
                    kb = {}
                    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
                        d = tempfile.mkdtemp(prefix="bcfgpu_pmc_", dir="/tmp")
                        first, tot = None, 0.0
                        try:                                 # (a counter pass that hangs costs three minutes, not twenty, and leaves no directory behind)
                            subprocess.run([prof, "--kernel-trace", "--pmc", cname, "-d", d, "-o", "pmc", "--output-format", "csv", "--", sys.executable,
                                            os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--samples", str(S), "--depth", str(a.depth),
                                            "--sites", str(T)] + common, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=180)
                            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                                for r in csv.DictReader(open(f)):
                                    if "glfgen_kernel" not in r["Kernel_Name"] or r["Counter_Name"] != cname:
                                        continue
                                    key = (r["Kernel_Name"], r["Dispatch_Id"])
                                    if first is None:
                                        first = key
                                    if key == first:
                                        tot += float(r["Counter_Value"])
                        finally:
                            shutil.rmtree(d, ignore_errors=True)
                        if first is not None:
                            kb[cname] = tot
                    if len(kb) == 2:
                        live = int((2 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024)
                        out["roofline"]["traffic"] = live
                        out["roofline"]["traffic_source"] = ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate child passes of this "
                                                             "bench on the same launch shape, first glfgen_kernel dispatch); gfx950: 2 x FETCH_SIZE + WRITE_SIZE, KiB")
                        out["roofline"]["traffic_counters_kib"] = kb
            except Exception as e:                           # the counters are an extra: the headline does not depend on them
                out["roofline"]["traffic_live_error"] = repr(e)[:200]
            # the honest end-to-end numbers where the driver's parser keeps them (it keeps `config` whole and drops `extra`)
            out["config"]["configs3_mixed_sites_per_s"] = wgs.get("value")
            out["config"]["configs3_mixed_with_front_sites_per_s"] = (wgs.get("with_front") or {}).get("sites_per_s")
            out["config"]["configs3_mixed_us_per_column"] = wgs.get("us_per_column")
            out["config"]["configs4_shape_sites_per_s"] = c4.get("value")
            out["config"]["configs4_shape_mcall_ms"] = ((c4.get("roofline") or {}).get("other_kernels_ms") or {}).get("mcall_kernel")
            out["config"]["secondary_note"] = ("configs3_mixed: bench.py --mode wgs -- 1000 samples x 30x of READS over a 16384-column tile with 0.5 % indel-noise reads "
                                               "(5 % of the indels 8-40 bases long): SNP path + candidate typing + realignment + indel pass + call -m on both kinds of record; "
                                               "with_front adds upload, BAQ and pileup; configs4_shape: call -G (4 groups) with a ploidy array")
            out["extra"] = {
                "configs2_mixed": {k: mix.get(k) for k in ("metric", "value", "unit", "ms_per_step", "config", "split_ms", "cpu_baseline", "error") if k in mix},
                "host_fed_pileup": {k: hf.get(k) for k in ("metric", "value", "unit", "config", "whole_call_ms", "pcie", "host_fed_pipeline", "byte_per_base_form", "error") if k in hf},
                "host_fed_chain_with_baq": {k: hfb.get(k) for k in ("config", "whole_call_ms", "pcie", "host_fed_pipeline", "error") if k in hfb},
                "baq_stage": {k: bq.get(k) for k in ("metric", "value", "unit", "config", "whole_call_ms", "pool_form", "cpu_baseline", "error") if k in bq},
                "configs4_shape": {k: c4.get(k) for k in ("value", "unit", "ms_per_step", "config", "roofline", "error") if k in c4},
                "indel_stage": {k: ind.get(k) for k in ("metric", "value", "unit", "config", "kernel", "host_ms", "indel_pass", "host_pointer_form", "cpu_baseline", "error") if k in ind},
                "configs3_mixed": {k: wgs.get(k) for k in ("metric", "value", "unit", "ms_per_step", "config", "split_ms", "us_per_column", "realignment", "front_ms", "with_front",
                                                            "cpu_baseline", "note", "error") if k in wgs},
                "note": "configs3_mixed: --mode wgs, the honest end-to-end line of the headline shape -- 1000 samples x 30x of reads with 0.5 % indel noise: SNP path on every "
                        "column + candidate typing + realignment + indel pass; configs4_shape: the same tile through call -G (4 sample groups on FORMAT/AD) with a ploidy array (25 % haploid); "
                        "indel_stage: bcf_call_gap_prep on 500-sample indel-candidate columns (BASELINE configs[2] shape), bcfgpu_gap_prep_tile on a read pool resident in HBM; "
                        "configs2_mixed: SNP path + indel path per step at the configs[2] mix (10 % indel sites); host_fed_pileup: --mode pileup, the read "
                        "pool crossing PCIe every region (DESIGN.md 5); host_fed_chain_with_baq: the same with BAQ on the pool in HBM before the pileup (what a user of "
                        "host/bcfgpu_sam sees with BAQ on: BAQ is the stage that bounds it); baq_stage: --mode baq (bcfgpu_baq with host pointers, bcfgpu_pool_baq on the resident pool)"}
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

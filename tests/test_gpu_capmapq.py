"""bcfgpu_cap_mapq (`mpileup -C INT`: sam_cap_mapq of htslib, mpileup.c:235-239) against the oracle's restatement
(oracle/capmapq.c) read by read -- the reference's own fixture reads after BAQ, and random reads with mismatches, clips,
indels, N's and ends of the reference -- and the C driver with -C against the Python pipeline over the oracle.
No golden of the reference exercises -C: parity unpinned for this one function."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from bcftools_amd import abi
from bcftools_amd.lib import check
from tests.helpers import sam, vcf, orc, mplpdrv as M, mplpcmp as K

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_caps(ctx, reads, refseq, thres):
    rd, d = M.pack_reads(reads)
    cap = np.zeros(len(reads), np.int32)
    check(ctx.L.bcfgpu_cap_mapq(ctx.h, C.byref(rd), refseq.encode(), len(refseq), thres, cap.ctypes.data))
    return cap


@pytest.mark.parametrize("thres", [50, 20, -1])
def test_fixture_reads_after_baq(golden_dir, gpu_ctx_factory, thres):
    G = os.path.join(golden_dir, "mpileup")
    ref = sam.read_fasta(os.path.join(G, "mpileup.ref.fa"))["17"]
    reads = [r for f in ("mpileup.1.sam", "mpileup.2.sam", "mpileup.3.sam") for r in sam.Sam(os.path.join(G, f)).reads
             if r.rname == "17" and not (r.flag & 4)]
    for r in reads:
        r.zq = None
        M.apply_baq(r, ref)
    ctx = gpu_ctx_factory(abi.default_cfg(1, max_sites=1, max_reads=64))
    got = device_caps(ctx, reads, ref, thres)
    want = np.array([M.cap_mapq_oracle(r, ref, thres) for r in reads], np.int32)
    np.testing.assert_array_equal(got, want)
    assert len(set(want.tolist())) > 3


def test_random_reads(gpu_ctx_factory):
    rng = np.random.default_rng(2026)
    ref = "".join(rng.choice(list("ACGT"), 3000))
    ref = ref[:1500] + "NNNN" + ref[1504:]
    reads = []
    for i in range(4000):
        pos = int(rng.integers(0, 2990))
        ops = []
        if rng.random() < 0.1:
            ops.append((int(rng.integers(1, 6)), "H"))
        if rng.random() < 0.3:
            ops.append((int(rng.integers(1, 12)), "S"))
        ops.append((int(rng.integers(10, 60)), "M"))
        if rng.random() < 0.3:
            ops.append((int(rng.integers(1, 5)), "I" if rng.random() < 0.5 else "D"))
            ops.append((int(rng.integers(5, 40)), "M"))
        if rng.random() < 0.05:
            ops.append((int(rng.integers(20, 200)), "N"))
            ops.append((int(rng.integers(5, 30)), "M"))
        if rng.random() < 0.2:
            ops.append((int(rng.integers(1, 10)), "S"))
        seq, x = [], pos
        for n, op in ops:
            if op in "M":
                for j in range(n):
                    b = ref[x + j] if x + j < len(ref) else "A"
                    if rng.random() < 0.06:
                        b = rng.choice(list("ACGTN"))
                    seq.append(b)
                x += n
            elif op in "IS":
                seq.extend(rng.choice(list("ACGT"), n))
            elif op in "DN":
                x += n
        lq = len(seq)
        r = sam.Read.__new__(sam.Read)
        r.pos, r.cigar, r.seq, r.flag, r.mapq, r.l_qseq, r.zq = pos, ops, "".join(seq), 0, 60, lq, None
        r.qual = rng.choice([2, 11, 12, 13, 25, 33, 34, 41], lq).astype(np.int32)
        r.bamcigar = np.array([(n << 4) | "MIDNSHP=X".index(op) for n, op in ops], dtype=np.uint32)
        reads.append(r)
    ctx = gpu_ctx_factory(abi.default_cfg(1, max_sites=1, max_reads=64))
    for thres in (60, 35):
        got = device_caps(ctx, reads, ref, thres)
        want = np.array([M.cap_mapq_oracle(r, ref, thres) for r in reads], np.int32)
        np.testing.assert_array_equal(got, want)
        assert (want < 0).any() and (want == thres).any() and ((want > 0) & (want < thres)).any()


def test_c_driver_with_adjust_mq(golden_dir):
    """`bcfgpu_sam -C 50` (BAQ on): every SNP record equals what the Python pipeline over the oracle gives with the same option
    (BAQ -> sam_cap_mapq -> -q / orphan filters -> overlaps -> pileup), and differs from the run without -C."""
    from tests.test_c_host import build_host, SAM_EXE
    build_host()
    G = os.path.join(golden_dir, "mpileup")
    files = [os.path.join(G, "mpileup.%d.sam" % i) for i in (1, 2, 3)]
    base = [os.path.join(G, "mpileup.ref.fa"), "17", "100", "400"] + files
    out = subprocess.run([SAM_EXE, "-C", "50"] + base, check=True, capture_output=True, text=True).stdout
    plain = subprocess.run([SAM_EXE] + base, check=True, capture_output=True, text=True).stdout
    assert out != plain
    tmp = os.path.join(ROOT, "gpurun_out", "capmapq.vcf") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else "/tmp/capmapq.vcf"
    open(tmp, "w").write(out)
    got = vcf.Vcf(tmp)
    fmt_flag = abi.INFO_VDB | abi.INFO_RPB
    sams = [sam.Sam(f) for f in files]
    ref = sam.read_fasta(os.path.join(G, "mpileup.ref.fa"))
    prep = M.Prepared(sams, ref, "17", sam.MplpOpts(fmt_flag=fmt_flag, cap_thres=50))
    tile, cols, kept = M.snp_tile(prep, range(99, 400))
    res = orc.mpileup(abi.default_cfg(len(prep.samples), fmt_flag=fmt_flag), tile)
    snp = [r for r in got.recs if "INDEL" not in r.info]
    assert len(snp) == len(kept) > 250
    for i, r in enumerate(snp):
        assert r.pos == kept[i] + 1
        K.check_record(r, res.site[i], res, i, K.snp_alleles(res.site[i]), fmt_flag)

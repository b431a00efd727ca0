"""bcfgpu_baq (sam_prob_realn on the device) against the oracle, read by read: the reads of the reference's own test
SAM files and synthetic reads with indels; plain and extended BAQ, with and without applying it."""
import os

import numpy as np
import pytest

from bcftools_amd import abi, synth
from tests.helpers import sam, mplpdrv as M

pytestmark = pytest.mark.gpu


def _oracle(reads, refseq, flag):
    out = []
    for r in reads:
        q0 = r.qual.copy()
        r.zq = None
        rc = M.apply_baq(r, refseq, flag)
        out.append((rc, r.qual.copy(), None if r.zq is None else r.zq.copy()))
        r.qual, r.zq = q0, None
    return out


@pytest.mark.parametrize("samf,fa", [("mpileup.1.sam", "mpileup.ref.fa"), ("mpileup.2.sam", "mpileup.ref.fa"),
                                     ("mpileup.4.sam", "mpileup.ref.fa"), ("indel-AD.1.sam", "indel-AD.1.fa")])
@pytest.mark.parametrize("flag", [3, 1, 2, 7])        # 7: mpileup -E (apply | extended | recompute)
def test_baq_matches_oracle_on_reference_reads(golden_dir, gpu_ctx_factory, samf, fa, flag):
    G = os.path.join(golden_dir, "mpileup")
    s = sam.Sam(os.path.join(G, samf))
    ref = sam.read_fasta(os.path.join(G, fa))
    ctx = gpu_ctx_factory(abi.default_cfg(1, max_sites=1, max_reads=64))
    for contig, refseq in ref.items():
        reads = [r for r in s.reads if r.rname == contig and not (r.flag & sam.BAM_FUNMAP) and r.l_qseq > 0]
        if not reads:
            continue
        want = _oracle(reads, refseq, flag)
        q0 = [r.qual.copy() for r in reads]
        ret = M.apply_baq_hip(reads, refseq, ctx, flag)
        n_applied = 0
        for r, (rc, wq, wz), rr, q in zip(reads, want, ret, q0):
            assert (rc == 0) == (rr == 0)
            if rc == 0:
                np.testing.assert_array_equal(r.qual, wq, err_msg=r.qname)
                np.testing.assert_array_equal(r.zq, wz, err_msg=r.qname)
                n_applied += 1
            r.qual, r.zq = q, None
        assert n_applied > 0


@pytest.mark.parametrize("kw", [{}, dict(lens=(-40, -25, -12, -8, 8, 12, 25, 40), lens2=(-3, 2, 60, -90)),
                                dict(lens=(-305, -320), lens2=(-2, 3), spacing=900)])     # bands past what three LDS rows hold: the scratch matrices
def test_baq_matches_oracle_on_synthetic_reads(gpu_ctx_factory, kw):
    """... the second case: reads with indels of 8-90 bases, whose band (the indel's length + 3, realn.c) is past the two
    register-row classes: the class with its working rows in LDS (baq_fb_lds); the third: a 305-320 base deletion, whose band does
    not fit LDS (baq_fb_scratch)."""
    b = synth.indel_batch(77, 6, 20, depth=12.0, **kw)
    refseq = b["ref"].decode()
    R = b["reads"]

    class Rd:
        pass
    reads = []
    nt = "=ACMGRSVTWYHKDBN"
    for i in range(R["n_reads"]):
        r = Rd()
        o, n = int(R["r_seq_off"][i]), int(R["r_lq"][i])
        r.pos, r.l_qseq, r.flag, r.qname = int(R["r_pos"][i]), n, int(R["r_flag"][i]), "r%d" % i
        r.bamcigar = R["cig"][R["r_cig_off"][i]:R["r_cig_off"][i] + R["r_ncig"][i]].copy()
        r.seq = "".join(nt[c] for c in R["seq16"][o:o + n])
        r.qual = R["qual"][o:o + n].astype(np.int32)
        r.zq = None
        reads.append(r)
    ctx = gpu_ctx_factory(abi.default_cfg(1, max_sites=1, max_reads=64))
    want = _oracle(reads, refseq, 3)
    ret = M.apply_baq_hip(reads, refseq, ctx, 3)
    assert (ret == 0).all()
    for r, (rc, wq, wz) in zip(reads, want):
        assert rc == 0
        np.testing.assert_array_equal(r.qual, wq)
        np.testing.assert_array_equal(r.zq, wz)


def _random_reads(seed, n, L=600):
    """Reads with soft clips, insertions, deletions, N bases, low qualities, near both ends of the reference."""
    rng = np.random.default_rng(seed)
    ref = "".join("ACGT"[i] for i in rng.integers(0, 4, L))
    if seed % 2:
        ref = ref[:200] + "N" * 7 + ref[207:]

    class Rd:
        pass
    reads = []
    for i in range(n):
        ops = []
        if rng.random() < 0.3:
            ops.append((int(rng.integers(1, 12)), "S"))
        remaining = int(rng.integers(25, 120))
        while remaining > 0:
            m = int(min(remaining, rng.integers(5, 60)))
            ops.append((m, "M"))
            remaining -= m
            if remaining > 0 and rng.random() < 0.5:
                ops.append((int(rng.integers(1, 14 if rng.random() < 0.15 else 4)), "ID"[int(rng.integers(0, 2))]))
        if rng.random() < 0.3:
            ops.append((int(rng.integers(1, 12)), "S"))
        reflen = sum(l for l, o in ops if o in "MD")
        pos = int(rng.integers(0, 3)) if rng.random() < 0.1 else int(rng.integers(0, L - reflen + 1))
        if rng.random() < 0.1:
            pos = L - reflen                                     # flush with the end of the contig
        seq, x = [], pos
        for l, o in ops:
            if o == "M":
                seq += [ref[x + k] if rng.random() > 0.03 else "ACGT"[int(rng.integers(0, 4))] for k in range(l)]
                x += l
            elif o == "D":
                x += l
            else:
                seq += ["ACGT"[int(rng.integers(0, 4))] for _ in range(l)]
        seq = "".join("N" if (c != "N" and rng.random() < 0.01) else c for c in seq)
        r = Rd()
        r.pos, r.seq, r.l_qseq, r.flag, r.qname = pos, seq, len(seq), 0, "f%d" % i
        r.qual = rng.integers(2, 42, len(seq)).astype(np.int32)
        r.bamcigar = np.array([(l << 4) | "MIDNS".index(o) for l, o in ops], dtype=np.uint32)
        r.zq = None
        reads.append(r)
    return ref, reads


@pytest.mark.parametrize("seed,flag", [(101, 3), (102, 3), (103, 1), (104, 2)])
def test_baq_matches_oracle_on_random_cigars(gpu_ctx_factory, seed, flag):
    refseq, reads = _random_reads(seed, 400)
    ctx = gpu_ctx_factory(abi.default_cfg(1, max_sites=1, max_reads=64))
    want = _oracle(reads, refseq, flag)
    ret = M.apply_baq_hip(reads, refseq, ctx, flag)
    wide = 0
    for r, (rc, wq, wz), rr in zip(reads, want, ret):
        assert (rc == 0) == (rr == 0), r.qname
        if rc == 0:
            np.testing.assert_array_equal(r.qual, wq, err_msg=r.qname)
            np.testing.assert_array_equal(r.zq, wz, err_msg=r.qname)
    assert sum(rc == 0 for rc, _, _ in want) > 300


def test_baq_on_reads_of_one_to_nine_bases(gpu_ctx_factory):
    """Reads of 1 .. 9 bases, every length on both strands of the parity the row store turns on: only the odd forward rows are
    kept, row 1 is scaled by a division and row 2 is re-formed from it, a read of one base has nothing but row 1 (csrc/baq.hip)."""
    rng = np.random.default_rng(7)
    L = 300
    refseq = "".join("ACGT"[i] for i in rng.integers(0, 4, L))

    class Rd:
        pass
    reads = []
    for n in range(1, 10):
        for rep in range(24):
            pos = int(rng.integers(0, L - n + 1)) if rep else (0 if n % 2 else L - n)     # also flush with both ends of the contig
            seq = "".join(refseq[pos + k] if rng.random() > 0.1 else "ACGT"[int(rng.integers(0, 4))] for k in range(n))
            r = Rd()
            r.pos, r.seq, r.l_qseq, r.flag, r.qname = pos, seq, n, 0, "t%d_%d" % (n, rep)
            r.qual = rng.integers(2, 42, n).astype(np.int32)
            r.bamcigar = np.array([(n << 4) | 0], dtype=np.uint32)
            r.zq = None
            reads.append(r)
    order = rng.permutation(len(reads))                       # lengths mixed inside the wavefronts
    reads = [reads[i] for i in order]
    for flag in (3, 7):
        rs = []
        for r in reads:
            c = Rd(); c.__dict__.update(r.__dict__); c.qual = r.qual.copy(); rs.append(c)
        ctx = gpu_ctx_factory(abi.default_cfg(1, max_sites=1, max_reads=64))
        want = _oracle(rs, refseq, flag)
        ret = M.apply_baq_hip(rs, refseq, ctx, flag)
        n_ok = 0
        for r, (rc, wq, wz), rr in zip(rs, want, ret):
            assert (rc == 0) == (rr == 0), r.qname
            if rc == 0:
                n_ok += 1
                np.testing.assert_array_equal(r.qual, wq, err_msg=r.qname)
                np.testing.assert_array_equal(r.zq, wz, err_msg=r.qname)
        assert n_ok > 150


def test_baq_on_reads_of_one_length_over_a_reference_with_n(gpu_ctx_factory):
    """Wavefronts whose reads all have one length run the mask-free row bodies of baq_fb_reg -- until a read of the wavefront gets
    an N under its band, or its band reaches the end of the contig: from that row on the wavefront runs the masked bodies, in the
    forward pass and again (from the other side) in the backward pass.  256 reads of 100 bases, plain 100M, over a reference with
    an N, a run of N and reads flush with both ends of the contig; a few reads carry an N of their own."""
    rng = np.random.default_rng(2024)
    L, n = 2000, 100
    ref = [("ACGT"[i]) for i in rng.integers(0, 4, L)]
    ref[500] = "N"
    for k in range(1200, 1207):
        ref[k] = "N"
    refseq = "".join(ref)

    class Rd:
        pass
    reads = []
    for i in range(256):
        pos = int(rng.integers(0, L - n + 1))
        if i % 32 == 0:
            pos = 0 if i % 64 == 0 else L - n
        if i % 16 == 5:
            pos = int(rng.integers(420, 500))                  # the N comes under the band somewhere inside the read
        seq = [refseq[pos + k] if rng.random() > 0.02 else "ACGT"[int(rng.integers(0, 4))] for k in range(n)]
        if i % 23 == 0:
            seq[int(rng.integers(0, n))] = "N"
        r = Rd()
        r.pos, r.seq, r.l_qseq, r.flag, r.qname = pos, "".join(seq), n, 0, "u%d" % i
        r.qual = rng.integers(2, 42, n).astype(np.int32)
        r.bamcigar = np.array([(n << 4) | 0], dtype=np.uint32)
        r.zq = None
        reads.append(r)
    for flag in (3, 1):
        rs = []
        for r in reads:
            c = Rd(); c.__dict__.update(r.__dict__); c.qual = r.qual.copy(); rs.append(c)
        ctx = gpu_ctx_factory(abi.default_cfg(1, max_sites=1, max_reads=64))
        want = _oracle(rs, refseq, flag)
        ret = M.apply_baq_hip(rs, refseq, ctx, flag)
        for r, (rc, wq, wz), rr in zip(rs, want, ret):
            assert rc == 0 and rr == 0, r.qname
            np.testing.assert_array_equal(r.qual, wq, err_msg=r.qname)
            np.testing.assert_array_equal(r.zq, wz, err_msg=r.qname)

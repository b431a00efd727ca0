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
@pytest.mark.parametrize("flag", [3, 1, 2])
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


def test_baq_matches_oracle_on_synthetic_reads(gpu_ctx_factory):
    b = synth.indel_batch(77, 6, 20, depth=12.0)
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

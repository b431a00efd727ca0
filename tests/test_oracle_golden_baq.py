"""Pins the oracle against the reference's mpileup goldens produced with the DEFAULT read preprocessing
(test.pl:640-644,658): BAQ + mate-overlap tweak upstream, then glfgen/errmod/combine for the SNP record and
bcf_call_gap_prep (probaln realignment) + glfgen/combine for the indel record.  Every record field that
bcf_call2bcf derives from bcf_call_t is compared: alleles, DP, I16, QS, VDB, SGB, RPB, MQB, MQSB, BQB, MQ0F,
IDV/IMF, PL, DP, DV, DP4, SP (Fisher exact), AD/ADF/ADR/DPR.

This is what pins the restated htslib pieces (probaln_glocal fwd/bwd, sam_prob_realn, tweak_overlap_quality,
kt_fisher_exact, kf_erfc) that test_oracle_golden_mpileup.py cannot reach."""
import os
import pytest

from bcftools_amd import abi as A
from tests.helpers import sam, orc, vcf, mplpdrv as M, mplpcmp as K

BASE = A.INFO_VDB | A.INFO_RPB
TRIO = ["mpileup.1.sam", "mpileup.2.sam", "mpileup.3.sam"]
CASES = [
    (TRIO, "mpileup.ref.fa", "17", 99, 149, "mpileup.1.out", BASE, 51, 0),
    (TRIO, "mpileup.ref.fa", "17", 99, 599, "mpileup.2.out", BASE | A.FMT_DP | A.FMT_DV, 501, 1),
    (TRIO, "mpileup.ref.fa", "17", 99, 599, "mpileup.4.out",
     BASE | A.FMT_DP | A.FMT_DPR | A.FMT_DV | A.FMT_DP4 | A.INFO_DPR | A.FMT_SP, 501, 1),
    (TRIO, "mpileup.ref.fa", "17", 99, 599, "mpileup.5.out",
     BASE | A.FMT_DP | A.FMT_AD | A.FMT_ADF | A.FMT_ADR | A.FMT_SP | A.INFO_AD | A.INFO_ADF | A.INFO_ADR, 501, 1),
    (["indel-AD.1.sam"], "indel-AD.1.fa", "000000F", 0, 10000, "indel-AD.1.out", BASE | A.FMT_AD, 297, 6),
    # test.pl:653: one file, no region: every covered position of the contig (4001 SNP records, many of them without a
    # usable read, and one indel record)
    (["mpileup.3.sam"], "mpileup.ref.fa", "17", 0, 4100, "mpileup.11.out", BASE, 4001, 1),
    # test.pl:647-652: sample and read-group selection (bam_sample.c) regroups the same reads: -s keeps two of the three
    # one-sample files, -s ^ the third, -S renames (names only), -G maps read groups to samples (reads of unlisted read
    # groups are dropped; SAMPLE2's read group is in none of the three files, its column stays empty)
    (["mpileup.2.sam", "mpileup.3.sam"], "mpileup.ref.fa", "17", 99, 149, "mpileup.7.out", BASE, 51, 0),
    (["mpileup.1.sam"], "mpileup.ref.fa", "17", 99, 149, "mpileup.8.out", BASE, 51, 0),
    (["mpileup.2.sam", "mpileup.3.sam"], "mpileup.ref.fa", "17", 99, 149, "mpileup.9.out", BASE, 51, 0),
    # test.pl:659: soft-clip counts (the fixture exists as BAM only: tests/helpers/sam.py reads it)
    (["mpileup-SCR.bam"], "mpileup-SCR.fa", "1", 0, 200, "mpileup-SCR.out", BASE | A.INFO_SCR | A.FMT_SCR, None, None),
    (TRIO, "mpileup.ref.fa", "17", 99, 149, "mpileup.10.out", BASE, 51, 0,
     dict(rg_map={"ERR162872": "HG00100", "ERR162875": "SAMPLE1a", "ERR013140": "SAMPLE1b", "ERR229776": "SAMPLE2",
                  "ERR229775": "SAMPLE3"}, samples=["SAMPLE1b", "HG00100", "SAMPLE1a", "SAMPLE2", "SAMPLE3"])),
]


def run_case(golden_dir, case, engine, gap_ctx=None, baq_ctx=None):
    samfiles, reffa, contig, beg, end, goldf, fmt_flag, n_snp, n_indel = case[:9]
    kw = case[9] if len(case) > 9 else {}
    G = os.path.join(golden_dir, "mpileup")
    sams = [(sam.Bam if f.endswith(".bam") else sam.Sam)(os.path.join(G, f)) for f in samfiles]
    ref = sam.read_fasta(os.path.join(G, reffa))
    prep = M.Prepared(sams, ref, contig, sam.MplpOpts(fmt_flag=fmt_flag), baq_ctx=baq_ctx() if baq_ctx else None, **kw)
    tile, cols, kept = M.snp_tile(prep, range(beg, end + 1))
    cfg = A.default_cfg(len(prep.samples), fmt_flag=fmt_flag)
    res = engine(cfg, tile)
    gold = vcf.Vcf(os.path.join(G, goldf))
    if kw.get("samples"):
        assert gold.samples == kw["samples"]
    snp = {r.pos: r for r in gold.recs if "INDEL" not in r.info}
    ind = {r.pos: r for r in gold.recs if "INDEL" in r.info}
    if n_snp is None:
        n_snp, n_indel = len(snp), len(ind)
    assert (len(snp), len(ind)) == (n_snp, n_indel)
    assert [p + 1 for p in kept] == sorted(snp)
    seen_indel = 0
    for i, p in enumerate(kept):
        K.check_record(snp[p + 1], res.site[i], res, i, K.snp_alleles(res.site[i]), fmt_flag)
        # indel record at the same position (mpileup.c:354-365), max_indel_depth 250 per sample
        g = M.gap_prep(prep, cols[i], p, ctx=gap_ctx() if gap_ctx else None) if sum(len(x) for x in cols[i]) < 250 * len(prep.samples) else None
        if g is None:
            assert (p + 1) not in ind
            continue
        ir = engine(cfg, M.indel_tile(prep, cols[i], g))
        if ir.site[0]["ret"] < 0:
            assert (p + 1) not in ind
            continue
        assert (p + 1) in ind, "unexpected indel record at %d" % (p + 1)
        K.check_record(ind[p + 1], ir.site[0], ir, 0, M.indel_alleles(prep.refseq, p, ir.site[0], g), fmt_flag, extra=g)
        seen_indel += 1
    assert seen_indel == n_indel


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_oracle_reproduces_default_mpileup_golden(golden_dir, idx):
    run_case(golden_dir, CASES[idx], orc.mpileup)

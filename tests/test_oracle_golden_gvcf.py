"""Pins the oracle's gvcf_write restatement (oracle/gvcf.c) on the reference's golden test/mpileup/mpileup.6.out
(`mpileup -a DP,DV -r17:100-600 --gvcf 0,2,5`, test.pl:645-646): every line of the golden -- gVCF blocks, the records that
stay as they are, the indel record that cuts a block one position short -- from the mpileup stage's results."""
import os

import numpy as np

from bcftools_amd import abi as A
from tests.helpers import orc, sam, vcf, mplpdrv as M, mplpcmp as K
from tests.test_oracle_golden_baq import TRIO, BASE


def run_gvcf_case(golden_dir, engine, gvcf_engine, gap_ctx=None, baq_ctx=None):
    G = os.path.join(golden_dir, "mpileup")
    fmt_flag = BASE | A.FMT_DP | A.FMT_DV
    sams = [sam.Sam(os.path.join(G, f)) for f in TRIO]
    ref = sam.read_fasta(os.path.join(G, "mpileup.ref.fa"))
    prep = M.Prepared(sams, ref, "17", sam.MplpOpts(fmt_flag=fmt_flag), baq_ctx=baq_ctx() if baq_ctx else None)
    tile, cols, kept = M.snp_tile(prep, range(99, 600))
    cfg = A.default_cfg(len(prep.samples), fmt_flag=fmt_flag)
    res = engine(cfg, tile)
    # the indel records (mpileup.c:354-365): they follow the SNP record of their position and cannot join a block
    indel = {}
    for i, p in enumerate(kept):
        if sum(len(x) for x in cols[i]) >= 250 * len(prep.samples):
            continue
        g = M.gap_prep(prep, cols[i], p, ctx=gap_ctx() if gap_ctx else None)
        if g is None:
            continue
        ir = engine(cfg, M.indel_tile(prep, cols[i], g))
        if ir.site[0]["ret"] >= 0:
            indel[i] = (ir, g)
    brk = np.array([1 if i in indel else 0 for i in range(len(kept))], dtype=np.uint8)
    gv = gvcf_engine(cfg, res, np.array(kept, dtype=np.int32), [0, 2, 5], brk)

    gold = vcf.Vcf(os.path.join(G, "mpileup.6.out"))
    it = iter(gold.recs)
    S = len(prep.samples)
    n_blocks = n_plain = 0
    for i, p in enumerate(kept):
        b = int(gv.blk[i])
        if b < 0:
            K.check_record(next(it), res.site[i], res, i, K.snp_alleles(res.site[i]), fmt_flag)
            n_plain += 1
        elif int(gv.block[b]["last_site"]) == i:
            B, r = gv.block[b], next(it)
            first = int(B["first_site"])
            where = "17:%d" % r.pos
            assert r.pos == int(B["start_pos"]) + 1 == kept[first] + 1, where
            assert r.alleles == K.snp_alleles(res.site[first]) and r.qual is None, where
            assert set(r.info) == ({"END", "MinDP", "QS"} if int(B["start_pos"]) + 1 < int(B["end1"]) else {"MinDP", "QS"}), (where, r.info)
            if "END" in r.info:
                assert int(r.info["END"]) == int(B["end1"]), (where, r.info["END"], B)
            assert int(r.info["MinDP"]) == int(B["min_dp"]), (where, r.info["MinDP"], B)
            for a, q in zip(r.info_floats("QS"), res.site[first]["qsum"]):
                assert K.fclose(a, float(q)), (where, "QS")
            assert r.fmt_keys == ["PL", "DP"], where
            for s in range(S):
                assert r.fmt("PL", s) == ",".join(str(int(x)) for x in gv.pl[b, :, s]), (where, s, r.fmt("PL", s), gv.pl[b, :, s])
                assert int(r.fmt("DP", s)) == int(gv.dp[b, s]), (where, s, "DP")
            n_blocks += 1
        if i in indel:
            ir, g = indel[i]
            K.check_record(next(it), ir.site[0], ir, 0, M.indel_alleles(prep.refseq, p, ir.site[0], g), fmt_flag, extra=g)
    assert next(it, None) is None, "golden has more records"
    assert n_blocks == gv.n_blocks == 31 and n_plain == 10 and len(indel) == 1, (n_blocks, gv.n_blocks, n_plain, len(indel))


def orc_gvcf(cfg, res, pos, dp_range, brk):
    return orc.gvcf_blocks(res, pos, dp_range, brk=brk)


def test_oracle_reproduces_gvcf_golden(golden_dir):
    run_gvcf_case(golden_dir, orc.mpileup, orc_gvcf)


def run_call_gvcf_case(golden_dir, engine, blocks):
    """`bcftools call -mg0` (test.pl:277, vcfcall.c:1145-1149): the records mcall() leaves with the reference allele alone
    (ret == 1) collapse into blocks by FORMAT/DP; golden test/mpileup.2.out.  `blocks(n, S, pos, is_ref, dp, ranges)` returns
    (n_blocks, blk, min_dp, block table, block DP)."""
    from tests.helpers import calldrv as D
    G = os.path.join(golden_dir, "call")
    v, gold = vcf.Vcf(os.path.join(G, "mpileup.vcf")), vcf.Vcf(os.path.join(G, "mpileup.2.out"))
    called, names = D.run_call(v, engine)
    assert gold.samples == names
    S, n = len(names), len(called)
    pos = np.array([c.src.pos - 1 for c in called], dtype=np.int32)
    is_ref = np.array([1 if c.ret == 1 else 0 for c in called], dtype=np.uint8)
    dp = np.array([[int(c.src.fmt("DP", s)) for s in range(S)] for c in called], dtype=np.int32)
    nb, blk, min_dp, block, bdp = blocks(n, S, pos, is_ref, dp, [0])
    it = iter(gold.recs)
    n_blocks = n_plain = 0
    for i, c in enumerate(called):
        b = int(blk[i])
        if b < 0:
            r = next(it)
            assert r.pos == c.src.pos and r.alleles == c.alleles, (r.pos, c.src.pos)
            assert (c.ret == 1 and int(min_dp[i]) != 0) == ("MinDP" in r.info), r.pos
            n_plain += 1
        elif int(block[b]["last_site"]) == i:
            B, r = block[b], next(it)
            first = called[int(B["first_site"])]
            where = "17:%d" % r.pos
            assert r.pos == int(B["start_pos"]) + 1 == first.src.pos, where
            assert r.alleles == first.alleles and len(r.alleles) == 1 and r.qual is None, where
            assert set(r.info) == ({"END", "MinDP"} if int(B["start_pos"]) + 1 < int(B["end1"]) else {"MinDP"}), (where, r.info)
            if "END" in r.info:
                assert int(r.info["END"]) == int(B["end1"]), (where, r.info["END"], B)
            assert int(r.info["MinDP"]) == int(B["min_dp"]), (where, r.info["MinDP"], B)
            assert r.fmt_keys == ["GT", "DP"], where
            for s in range(S):
                assert r.fmt("GT", s) == first.gt[s], (where, s)
                assert int(r.fmt("DP", s)) == int(bdp[b, s]), (where, s, "DP")
            n_blocks += 1
    assert next(it, None) is None, "golden has more records"
    assert n_blocks == nb == 12 and n_plain == 11, (n_blocks, nb, n_plain)


def orc_call_blocks(n, S, pos, is_ref, dp, ranges):
    from bcftools_amd.host import GVCF_BLOCK_DTYPE
    rng = np.ascontiguousarray(ranges, dtype=np.int32)
    blk, min_dp = np.zeros(n, np.int32), np.zeros(n, np.int32)
    block = np.zeros(max(n, 1), GVCF_BLOCK_DTYPE)
    dpo, plo = np.zeros((max(n, 1), S), np.int32), np.zeros((max(n, 1), 3, S), np.int32)
    pl = np.zeros((n, 3, S), np.int32)
    p = orc._p
    nb = orc.lib().orc_gvcf_blocks(n, S, p(pos), None, None, p(is_ref), p(np.ascontiguousarray(dp)), p(pl), p(rng), len(rng),
                                   p(blk), p(min_dp), p(block), p(dpo), p(plo))
    return nb, blk, min_dp, block, dpo


def test_oracle_reproduces_call_gvcf_golden(golden_dir):
    run_call_gvcf_case(golden_dir, orc.mcall, orc_call_blocks)

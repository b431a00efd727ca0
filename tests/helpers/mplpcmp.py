"""Compare mpileup-stage results with a golden `bcftools mpileup` VCF, field by field (the record fields that
bcf_call2bcf, bam2bcf.c:756-906, derives from bcf_call_t)."""
import math
import numpy as np

from bcftools_amd import abi
from . import orc


def fclose(a, b, rtol=2e-5):
    return abs(a - b) <= rtol * max(abs(a), abs(b)) + 1e-9


def check_record(rec, site, res, i, alleles, fmt_flag, extra=None):
    """rec: golden vcf.Rec; site: res.site[i]; alleles: expected allele strings."""
    where = "%s:%d" % (rec.chrom, rec.pos)
    assert rec.alleles == alleles, (where, rec.alleles, alleles)
    na = len(alleles)
    assert int(site["ori_depth"]) == int(rec.info["DP"]), (where, "DP")
    assert [float(np.float32(x)) for x in site["anno"]] == rec.info_floats("I16"), (where, "I16", site["anno"], rec.info["I16"])
    q = [float(x) for x in site["qsum"][:na]]
    for a, b in zip(q, rec.info_floats("QS")):
        assert fclose(a, b), (where, "QS", q, rec.info["QS"])
    for tag, key in (("VDB", "vdb"), ("SGB", "seg_bias"), ("RPB", "mwu_pos"), ("MQB", "mwu_mq"), ("MQSB", "mwu_mqs"), ("BQB", "mwu_bq")):
        v = float(site[key])
        if tag in rec.info:
            assert math.isfinite(v) and fclose(v, float(rec.info[tag])), (where, tag, v, rec.info[tag])
        else:
            assert not math.isfinite(v), (where, tag, v)
    mq0f = float(np.float32(site["mq0"]) / np.float32(site["ori_depth"])) if site["ori_depth"] else 0.0
    assert fclose(mq0f, float(rec.info["MQ0F"])), (where, "MQ0F")
    adf_tot = [int(x) for x in site["adf_tot"][:na]]
    adr_tot = [int(x) for x in site["adr_tot"][:na]]
    if "ADF" in rec.info:
        assert rec.info_ints("ADF") == adf_tot, (where, "INFO/ADF")
    if "ADR" in rec.info:
        assert rec.info_ints("ADR") == adr_tot, (where, "INFO/ADR")
    for tag in ("AD", "DPR"):
        if tag in rec.info:
            assert rec.info_ints(tag) == [a + b for a, b in zip(adf_tot, adr_tot)], (where, "INFO/" + tag)
    if "SCR" in rec.info:
        assert fmt_flag & abi.INFO_SCR and int(rec.info["SCR"]) == int(site["scr_tot"]), (where, "INFO/SCR", site["scr_tot"])
    if extra:
        if "IDV" in rec.info:
            assert int(rec.info["IDV"]) == extra["max_support"], (where, "IDV")
            assert fclose(float(rec.info["IMF"]), extra["max_frac"]), (where, "IMF")
    S = res.pl.shape[2]
    pl = res.pl_of(i)
    for s in range(S):
        assert rec.fmt("PL", s) == ",".join(str(int(x)) for x in pl[s]), (where, s, "PL", rec.fmt("PL", s), pl[s])
        dp4 = [int(res.dp4[i, k, s]) for k in range(4)]
        if rec.fmt("DP", s) is not None:
            assert int(rec.fmt("DP", s)) == sum(dp4), (where, s, "DP")
        if rec.fmt("DV", s) is not None:
            assert int(rec.fmt("DV", s)) == dp4[2] + dp4[3], (where, s, "DV")
        if rec.fmt("DP4", s) is not None:
            assert [int(x) for x in rec.fmt("DP4", s).split(",")] == dp4, (where, s, "DP4")
        if rec.fmt("SP", s) is not None:
            assert fmt_flag & abi.FMT_SP, (where, "golden has FMT/SP but the flag is not set")
            assert int(rec.fmt("SP", s)) == int(res.sp[i, s]), (where, s, "SP", dp4, int(res.sp[i, s]))
        if rec.fmt("SCR", s) is not None:
            assert fmt_flag & abi.FMT_SCR and int(rec.fmt("SCR", s)) == int(res.scr[i, s]), (where, s, "FMT/SCR")
        adf = [int(res.adf[i, k, s]) for k in range(na)]
        adr = [int(res.adr[i, k, s]) for k in range(na)]
        if rec.fmt("ADF", s) is not None:
            assert [int(x) for x in rec.fmt("ADF", s).split(",")] == adf, (where, s, "ADF")
        if rec.fmt("ADR", s) is not None:
            assert [int(x) for x in rec.fmt("ADR", s).split(",")] == adr, (where, s, "ADR")
        for tag in ("AD", "DPR"):
            if rec.fmt(tag, s) is not None:
                assert [int(x) for x in rec.fmt(tag, s).split(",")] == [a + b for a, b in zip(adf, adr)], (where, s, tag)


def snp_alleles(site):
    na = int(site["n_alleles"])
    return ["ACGTN"[int(site["ori_ref"])]] + ["<*>" if int(site["unseen"]) == j else "ACGT"[int(site["a"][j])] for j in range(1, na)]

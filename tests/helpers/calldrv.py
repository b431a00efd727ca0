"""Test-side driver of the call stage: the record loop of `bcftools call -m` (vcfcall.c:1089-1148)
around an mcall *engine* (the CPU oracle or the HIP library), fed from the reference's VCF fixtures.

The engine is any callable  engine(cfg, CallInput) -> CallResult  (tests/helpers/orc.py types).
"""
import math
import numpy as np

from bcftools_amd import abi
from . import orc, vcf as V


def parse_ploidy_file(path):
    """ploidy.c: lines CHROM FROM TO SEX PLOIDY, '*' = default for that sex."""
    regs, dflt, sexes = [], {}, []
    with open(path) as fh:
        for line in fh:
            f = line.split()
            if len(f) < 5:
                continue
            if f[3] not in sexes:
                sexes.append(f[3])
            if f[0] == "*":
                dflt[f[3]] = int(f[4])
            else:
                regs.append((f[0], int(f[1]), int(f[2]), f[3], int(f[4])))
    return dict(regs=regs, dflt=dflt, sexes=sexes)


def ploidy_query(pl, chrom, pos1, sex):
    for c, b, e, s, p in pl["regs"]:
        if c == chrom and s == sex and b <= pos1 <= e:
            return p
    return pl["dflt"].get(sex, 2)


def parse_samples_file(path):
    """vcfcall.c:270-344 incl. PED (vcfcall.c:202-262): returns [(name, spec)] spec = '0'|'1'|'2' or a sex name."""
    lines = [l.rstrip("\n") for l in open(path) if l.strip() and not l.startswith("#")]
    out = []
    if lines and len(lines[0].split()) >= 5:       # PED: fam, sample, father, mother, sex(1=M,2=F)
        for l in lines:
            f = l.split()
            out.append((f[1], "M" if f[4] == "1" else "F"))
        return out
    for l in lines:
        f = l.split()
        out.append((f[0], f[1] if len(f) > 1 else "2"))
    return out


class CalledRec:
    __slots__ = ("src", "alleles", "qual", "qual_missing", "ac", "an", "gt", "pl", "gq", "gp", "als_map",
                 "dp4", "mq", "pl_dropped", "ploidy", "pv4", "ret")


def fmt_gt(a, b):
    def one(x):
        return "." if x == abi.GT_MISSING else str(x)
    if b == abi.GT_VECTOR_END:
        return one(a)
    return one(a) + "/" + one(b)


def run_call(vcf, engine, call_flag=0, output_tags=0, theta=1.1e-3, samples=None, ploidy=None,
             groups=None, grp_tag="AD", prior=None, recs=None):
    """Returns the list of CalledRec that `bcftools call -m` would print.

    samples: [(name, spec)] (-S); ploidy: parsed --ploidy-file; groups: {sample: group} or '-' (-G);
    prior: (AN_tag, AC_tag) (-F); recs: the records to call instead of vcf.recs (helpers/cals.py: -C alleles).
    """
    names = vcf.samples
    if samples is None:
        cols = list(range(len(names)))
        specs = ["2"] * len(cols)
        # without -S every sample gets the last sex of the ploidy definition (vcfcall.c:645-650)
        if ploidy is not None:
            specs = [ploidy["sexes"][-1]] * len(cols)
    else:
        cols = [names.index(n) for n, _ in samples if n in names]
        specs = [s for n, s in samples if n in names]
    S = len(cols)
    sub_names = [names[c] for c in cols]
    if groups == "-":
        grp, ngrp = list(range(S)), S
    elif groups:
        gids = {}
        grp = []
        for n in sub_names:
            g = groups[n]
            if g not in gids:
                gids[g] = len(gids)
        # group ids in order of first appearance in the group *file* (mcall.c:308-330)
        order = {}
        for n, g in groups.items():
            if n in sub_names and g not in order:
                order[g] = len(order)
        grp = [order[groups[n]] for n in sub_names]
        ngrp = len(order)
    else:
        grp, ngrp = None, 1

    cfg = abi.default_cfg(S, call_theta=theta, call_flag=call_flag, output_tags=output_tags, n_grp=ngrp)

    # select candidate records and their per-site ploidy vectors
    cand = []
    for rec in (vcf.recs if recs is None else recs):
        unseen = V.find_unseen(rec)
        nals = len(rec.alleles)
        is_ref = nals == 1 or (nals == 2 and unseen > 0)
        if is_ref and (call_flag & abi.CALL_VARONLY):
            continue
        pv = []
        for sp in specs:
            if sp in ("0", "1", "2"):
                pv.append(int(sp))
            elif ploidy is not None:
                pv.append(ploidy_query(ploidy, rec.chrom, rec.pos, sp))
            else:
                pv.append(2)
        cand.append((rec, unseen, tuple(pv)))

    out = []
    i = 0
    while i < len(cand):
        j = i
        while j < len(cand) and cand[j][2] == cand[i][2]:
            j += 1
        batch = cand[i:j]
        i = j
        pv = np.array(batch[0][2], dtype=np.uint8)
        n = len(batch)
        ngmax = max(len(r.alleles) * (len(r.alleles) + 1) // 2 for r, _, _ in batch)
        namax = max(len(r.alleles) for r, _, _ in batch)
        pl = np.full((n, ngmax, S), abi.INT32_VECTOR_END, dtype=np.int32)
        qs = np.zeros((n, 5), dtype=np.float32)
        ad = np.full((n, namax, S), abi.INT32_VECTOR_END, dtype=np.int32) if ngrp > 1 else None
        nals = np.zeros(n, dtype=np.int32)
        uns = np.zeros(n, dtype=np.int32)
        i16 = np.zeros((n, 16), dtype=np.float32)
        has16 = all("I16" in r.info for r, _, _ in batch)
        pan = np.full(n, abi.INT32_MISSING, dtype=np.int32) if prior else None
        pac = np.full((n, 4), abi.INT32_VECTOR_END, dtype=np.int32) if prior else None
        for k, (rec, unseen, _) in enumerate(batch):
            na = len(rec.alleles)
            ng = na * (na + 1) // 2
            nals[k], uns[k] = na, unseen
            for si, c in enumerate(cols):
                v = rec.fmt_ints("PL", c, ng)
                pl[k, :ng, si] = v
                if ad is not None:
                    a = rec.fmt_ints(grp_tag, c, na)
                    if a is not None:
                        ad[k, :na, si] = a
            if has16:
                i16[k] = rec.info_floats("I16")
            if "QS" in rec.info:
                q = rec.info_floats("QS")[:5]
                qs[k, :len(q)] = q
            if prior and prior[0] in rec.info:
                v = rec.info_ints(prior[0])
                if len(v) == 1:
                    pan[k] = v[0]
                    if prior[1] in rec.info:
                        acv = rec.info_ints(prior[1])[:4]
                        pac[k, :len(acv)] = acv
        cin = orc.CallInput(S, nals, uns, pl, qs, ad=ad, ploidy=pv, grp=grp, prior_an=pan, prior_ac=pac, i16=i16 if has16 else None)
        res = engine(cfg, cin)
        for k, (rec, unseen, _) in enumerate(batch):
            st = res.site[k]
            ret = int(st["ret"])
            if ret == -2:
                continue
            if (call_flag & abi.CALL_VARONLY) and ret == 0:
                continue
            c = CalledRec()
            c.src = rec
            c.ret = ret
            c.ploidy = pv
            na = len(rec.alleles)
            amap = [int(x) for x in st["als_map"][:na]]
            nn = int(st["nals_new"])
            al = [None] * max(nn, max(amap) + 1)
            for ia, m in enumerate(amap):
                if m >= 0:
                    al[m] = rec.alleles[ia]
            c.alleles = al[:nn]
            c.als_map = amap
            c.qual = float(st["qual"])
            c.qual_missing = bool(st["qual_missing"])
            c.ac = [int(x) for x in st["ac"][1:nn]]
            c.an = int(st["an"])
            c.gt = [fmt_gt(int(res.gt[k, 0, s]), int(res.gt[k, 1, s])) for s in range(S)]
            ngn = nn * (nn + 1) // 2
            c.pl_dropped = bool(st["pl_dropped"])
            c.pl = None if c.pl_dropped else [res.pl[k, :ngn, s].tolist() for s in range(S)]
            c.gq = [int(res.gq[k, s]) for s in range(S)]
            c.gp = [res.gp[k, :ngn, s].copy() for s in range(S)]
            # DP4 and MQ as the engine derives them from I16 (mcall.c:1659-1666)
            if int(st["has_i16"]):
                c.dp4, c.mq = [int(x) for x in st["dp4"]], int(st["mq"])
                c.pv4 = [float(x) for x in st["pv4"]] if int(st["pv4_tested"]) else None
            else:
                c.dp4, c.mq, c.pv4 = None, None, None
            out.append(c)
    return out, sub_names


def _int_list(s):
    return [abi.INT32_MISSING if x == "." else int(x) for x in s.split(",")]


def compare_with_golden(called, sub_names, gold, qual_rtol=2e-5, check_tags=()):
    """Field-by-field comparison of CalledRec list with a golden VCF (Vcf object). Raises AssertionError."""
    assert gold.samples == sub_names, (gold.samples, sub_names)
    assert len(called) == len(gold.recs), "record count %d vs golden %d" % (len(called), len(gold.recs))
    for c, g in zip(called, gold.recs):
        where = "%s:%d" % (g.chrom, g.pos)
        assert (c.src.chrom, c.src.pos) == (g.chrom, g.pos), where
        assert c.alleles == g.alleles, (where, c.alleles, g.alleles)
        if g.qual is None:
            assert c.qual_missing, where
        else:
            assert not c.qual_missing, where
            tol = max(qual_rtol * abs(g.qual), 1e-4 if abs(g.qual) > 1e-3 else 1e-6)
            assert abs(c.qual - g.qual) <= tol, (where, c.qual, g.qual)
        if "AC" in g.info:
            assert c.ac == g.info_ints("AC"), (where, c.ac, g.info["AC"])
        else:
            assert len(c.alleles) <= 1, where
        assert c.an == g.info_ints("AN")[0], (where, c.an, g.info["AN"])
        if "DP4" in g.info:
            assert c.dp4 == g.info_ints("DP4"), (where, c.dp4, g.info["DP4"])
            assert c.mq == g.info_ints("MQ")[0], (where, c.mq, g.info["MQ"])
        for s in range(len(sub_names)):
            assert c.gt[s] == g.fmt("GT", s), (where, s, c.gt[s], g.fmt("GT", s))
            gpl = g.fmt("PL", s)
            if gpl is None:
                assert c.pl_dropped, where
            else:
                want = _int_list(gpl)
                got = [x for x in c.pl[s] if x != abi.INT32_VECTOR_END]
                assert got == want, (where, s, got, want)
            if "GQ" in check_tags:
                assert c.gq[s] == int(g.fmt("GQ", s)), (where, s, c.gq[s], g.fmt("GQ", s))
            if "GP" in check_tags:
                want = [float(x) for x in g.fmt("GP", s).split(",")]
                got = [float(x) for x in c.gp[s][:len(want)]]
                for a, b in zip(got, want):
                    assert abs(a - b) <= 2e-5 * max(abs(b), 1e-30) + 1e-12, (where, s, got, want)
    return True

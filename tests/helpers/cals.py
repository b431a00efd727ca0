"""Test-side restatement of `bcftools call -C alleles -T targets.tab [-i]` around an mcall engine (test infrastructure).

Host logic only: which target line a record is paired with (next_line, vcfcall.c:501-605), the allele comparison of
vcmp.c:55-119, the record rewritten to the target alleles (mcall_constrain_alleles, mcall.c:1271-1421: alleles, FORMAT/PL
gathered through pl_map with the unseen allele standing in for alleles mpileup did not see, INFO/QS, Number=R FORMAT tags)
and the lines inserted for targets without a record (-i: tgt_flush, vcfcall.c:408-455).  The rewritten records then go
through calldrv.run_call with CALL_KEEPALT, i.e. through the same mcall engine as every other test.
"""
import math

import numpy as np

from bcftools_amd import abi
from . import vcf as V

MISSING, VEND = abi.INT32_MISSING, abi.INT32_VECTOR_END


def parse_tab(path):
    """tgt_parse (vcfcall.c:359-398): CHROM POS REF,ALT[,ALT..] -> [(chrom, pos1, [alleles])] in file order."""
    out = []
    for line in open(path):
        f = line.split()
        if len(f) >= 3:
            out.append((f[0], int(f[1]), f[2].split(",")))
    return out


class Vcmp:
    """vcmp.c:55-119."""

    def __init__(self):
        self.ndref, self.dref = 0, ""

    def set_ref(self, ref1, ref2):
        self.ndref = 0
        a, b = ref1.upper(), ref2.upper()
        i = 0
        while i < len(a) and i < len(b) and a[i] == b[i]:
            i += 1
        if i == len(a) and i == len(b):
            return 0
        if i < len(a) and i < len(b):
            return -1
        if i < len(a):                     # ref1 is longer
            self.dref, self.ndref = a[i:], len(a) - i
        else:
            self.dref, self.ndref = b[i:], -(len(b) - i)
        return 0

    def find_allele(self, als1, al2):
        b = al2.upper()
        for i, a1 in enumerate(als1):
            a = a1.upper()
            k = 0
            while k < len(a) and k < len(b) and a[k] == b[k]:
                k += 1
            if k < len(a) and k < len(b):
                continue
            if not self.ndref:
                if k == len(a) and k == len(b):
                    return i
                continue
            if k < len(a):
                if self.ndref < 0 or a[k:] != self.dref:
                    continue
                return i
            if self.ndref > 0 or b[k:] != self.dref:
                continue
            return i
        return -1


def is_indel(als):
    """vcfcall.c:456-470"""
    if len(als) > 1 and als[1][0] == "<":
        return False
    return any(a[0] != "<" and len(a) > 1 for a in als)


def gt2alleles(igt):
    k = int((math.isqrt(8 * igt + 1) - 1) // 2)
    return igt - k * (k + 1) // 2, k


def alleles2gt(a, b):
    return a * (a + 1) // 2 + b if a > b else b * (b + 1) // 2 + a


def _numberR_formats(header):
    out = set()
    for h in header:
        if h.startswith("##FORMAT=<ID=") and "Number=R" in h:
            out.add(h[len("##FORMAT=<ID="):].split(",")[0])
    return out


def constrain_record(rec, tals, unseen, nsmpl, fmtR):
    """mcall_constrain_alleles: returns (Rec, unseen) with the target alleles, or None when the site is skipped."""
    if len(tals) > 5:
        raise ValueError("Maximum accepted number of alleles is 5")
    vc = Vcmp()
    if vc.set_ref(rec.ref, tals[0]) < 0:
        raise ValueError("The reference alleles are not compatible at %s:%d" % (rec.chrom, rec.pos))
    als, amap, has_new = [tals[0]], [0], False
    nori = len(rec.alleles)
    for a in tals[1:]:
        j = vc.find_allele(rec.alts, a)
        if j + 1 == unseen:                                 # mcall.c:1294-1303 (also: not found and no unseen allele)
            return None
        if j >= 0:
            amap.append(j + 1)
        else:
            amap.append(unseen if unseen >= 0 else nori - 1)
            has_new = True
        als.append(a)
    if unseen:
        amap.append(unseen)
        als.append(rec.alleles[unseen])
    nals = len(als)
    if not has_new and nals == nori:
        return rec, unseen
    pl_map = [alleles2gt(amap[i], amap[j]) for i in range(nals) for j in range(i + 1)]
    # the widest PL vector of the record is the stride of bcf_get_format_int32
    width = max(len(rec.fmt("PL", s).split(",")) for s in range(nsmpl))
    f = rec.line.rstrip("\n").split("\t")
    f[3], f[4] = als[0], ",".join(als[1:]) if nals > 1 else "."
    keys = rec.fmt_keys
    new_smpl = []
    for s in range(nsmpl):
        ori = rec.fmt_ints("PL", s, width)
        new = []
        for k, km in enumerate(pl_map):
            v = ori[km]
            if v == MISSING and unseen >= 0:
                ia, ib = gt2alleles(km)
                ko = alleles2gt(ia, unseen)
                if ori[ko] == MISSING:
                    ko = alleles2gt(ib, unseen)
                if ori[ko] == MISSING:
                    ko = alleles2gt(unseen, unseen)
                v = ori[ko]
            if k == 0 and v == VEND:
                v = MISSING
            new.append(v)
        vals = list(rec.smpl[s]) + ["."] * (len(keys) - len(rec.smpl[s]))
        out = []
        for v in new:
            if v == VEND:
                break
            out.append("." if v == MISSING else str(v))
        vals[keys.index("PL")] = ",".join(out)
        for key in keys:                                    # Number=R FORMAT tags: new[k] = old[als_map[k]]
            if key in fmtR and key != "PL":
                o = vals[keys.index(key)].split(",")
                if o == ["."]:
                    continue
                vals[keys.index(key)] = ",".join(o[amap[k]] if amap[k] < len(o) else "." for k in range(nals))
        new_smpl.append(":".join(vals))
    info = []
    for kv in f[7].split(";"):
        if kv.startswith("QS="):
            qs = np.array([float(x) for x in kv[3:].split(",")], dtype=np.float32)
            nq = [qs[amap[i]] if amap[i] < len(qs) else np.float32(0) for i in range(nals)]
            kv = "QS=" + ",".join("%.9g" % float(x) for x in nq)
        info.append(kv)
    f[7] = ";".join(info)
    f[9:] = new_smpl
    return V.Rec("\t".join(f)), (nals - 1 if unseen else unseen)


class Missed:
    """a line written by tgt_flush_region: target alleles, QUAL '.', GT '.' for every sample"""
    def __init__(self, chrom, pos, alleles):
        self.chrom, self.pos, self.alleles = chrom, pos, alleles


def constrain(vcf, tab, insert_missed=False):
    """Returns the event list of the record loop: V.Rec (to be called, with .unseen_c set) and Missed entries in output
    order.  A Rec whose call is skipped by mcall() simply produces no line."""
    fmtR = _numberR_formats(vcf.header)
    nsmpl = len(vcf.samples)
    chroms, bychr = [], {}
    for i, (c, p, a) in enumerate(tab):
        if c not in bychr:
            bychr[c] = []
            chroms.append(c)
        bychr[c].append(i)
    for c in chroms:
        bychr[c].sort(key=lambda i: tab[i][1])             # regidx keeps the regions of a sequence sorted by start
    used = [False] * len(tab)
    events = []

    def flush_region(chrom, beg0, end0):
        for i in bychr.get(chrom, []):
            p0 = tab[i][1] - 1
            if p0 < beg0 or p0 > end0 or used[i]:
                continue
            used[i] = True
            events.append(Missed(chrom, tab[i][1], tab[i][2]))

    BIG = 1 << 40
    prev = None
    for rec in vcf.recs:
        at = [i for i in bychr.get(rec.chrom, []) if tab[i][1] == rec.pos]
        if not at:
            continue                                         # not a target position (vcfcall.c:525-535)
        best, bestn = None, 0
        rec_indel = 1 if is_indel(rec.alleles) else -1
        for i in at:
            if used[i]:
                continue
            tals = tab[i][2]
            n = 0
            vc = Vcmp()
            if vc.set_ref(rec.ref, tals[0]) == 0:
                n = 1
                if len(rec.alleles) > 1 and len(tals) > 1:
                    n += sum(1 for a in tals[1:] if vc.find_allele(rec.alts, a) >= 0)
            n *= rec_indel * (1 if is_indel(tals) else -1)
            if best is None or n > bestn:
                best, bestn = i, n
        if best is None:
            continue                                         # every target of this position is used up (vcfcall.c:1093)
        used[best] = True
        if insert_missed:                                    # tgt_flush (vcfcall.c:426-455)
            p0 = rec.pos - 1
            if prev is None:
                flush_region(rec.chrom, 0, p0 - 1)
            elif prev[0] != rec.chrom:
                flush_region(prev[0], prev[1] + 1, BIG)
                flush_region(rec.chrom, 0, p0 - 1)
            else:
                flush_region(prev[0], prev[1], p0 - 1)
            prev = (rec.chrom, p0)
        unseen = V.find_unseen(rec)
        r = constrain_record(rec, tab[best][2], unseen, nsmpl, fmtR)
        if r is None:
            continue                                         # mcall() returns -2
        events.append(r[0])
    if insert_missed:
        if prev is not None:
            flush_region(prev[0], prev[1], BIG)
        for c in chroms:
            flush_region(c, 0, BIG)
    return events


def run(vcf, tab, engine, insert_missed=False):
    """`call -mA -C alleles -T tab [-i]`: the output lines as CalledRec / Missed in order, and the sample names."""
    from . import calldrv
    events = constrain(vcf, tab, insert_missed)
    recs = [e for e in events if not isinstance(e, Missed)]
    called, names = calldrv.run_call(vcf, engine, call_flag=abi.CALL_KEEPALT, recs=recs)
    by_src = {id(c.src): c for c in called}
    out = []
    for e in events:
        if isinstance(e, Missed):
            out.append(e)
        elif id(e) in by_src:
            out.append(by_src[id(e)])
    return out, names


def compare_with_golden(out, names, gold, **kw):
    from . import calldrv
    assert len(out) == len(gold.recs), "record count %d vs golden %d" % (len(out), len(gold.recs))
    called, gcalled = [], []
    for o, g in zip(out, gold.recs):
        where = "%s:%d" % (g.chrom, g.pos)
        if isinstance(o, Missed):
            assert (o.chrom, o.pos, o.alleles) == (g.chrom, g.pos, g.alleles), (where, o.chrom, o.pos, o.alleles)
            assert g.qual is None and not g.info and g.fmt_keys == ["GT"] and all(s == ["."] for s in g.smpl), where
        else:
            assert g.fmt_keys != ["GT"], where
            called.append(o)
            gcalled.append(g)

    class _G:
        samples, recs = gold.samples, gcalled
    calldrv.compare_with_golden(called, names, _G, **kw)
    return True

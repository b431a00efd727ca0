"""Minimal VCF text reader for the reference's fixtures and goldens (test infrastructure)."""
import numpy as np

from bcftools_amd import abi


class Rec:
    __slots__ = ("chrom", "pos", "ref", "alts", "qual", "info", "fmt_keys", "smpl", "line")

    def __init__(self, line):
        f = line.rstrip("\n").split("\t")
        self.line = line
        self.chrom, self.pos, self.ref = f[0], int(f[1]), f[3]
        self.alts = [] if f[4] == "." else f[4].split(",")
        self.qual = None if f[5] == "." else float(f[5])
        self.info = {}
        if f[7] != ".":
            for kv in f[7].split(";"):
                if "=" in kv:
                    k, v = kv.split("=", 1)
                    self.info[k] = v
                else:
                    self.info[kv] = True
        self.fmt_keys = f[8].split(":") if len(f) > 8 else []
        self.smpl = [s.split(":") for s in f[9:]]

    @property
    def alleles(self):
        return [self.ref] + self.alts

    def info_floats(self, key):
        return [float(x) for x in self.info[key].split(",")]

    def info_ints(self, key):
        return [abi.INT32_MISSING if x == "." else int(x) for x in self.info[key].split(",")]

    def fmt(self, key, ismpl):
        """raw string of FORMAT/key for sample ismpl, or None if absent (trailing fields may be dropped)"""
        if key not in self.fmt_keys:
            return None
        k = self.fmt_keys.index(key)
        s = self.smpl[ismpl]
        return s[k] if k < len(s) else "."

    def fmt_ints(self, key, ismpl, width):
        """integer vector padded with vector_end like htslib's bcf_get_format_int32"""
        v = self.fmt(key, ismpl)
        out = [abi.INT32_VECTOR_END] * width
        if v is None:
            return None
        vals = v.split(",")
        for i, x in enumerate(vals[:width]):
            out[i] = abi.INT32_MISSING if x == "." else int(x)
        return out


class Vcf:
    def __init__(self, path):
        self.header, self.samples, self.recs = [], [], []
        with open(path) as fh:
            for line in fh:
                if line.startswith("##"):
                    self.header.append(line.rstrip("\n"))
                elif line.startswith("#CHROM"):
                    self.samples = line.rstrip("\n").split("\t")[9:]
                elif line.strip():
                    self.recs.append(Rec(line))


def find_unseen(rec):
    """vcfcall.c:1102-1111"""
    for i, a in enumerate(rec.alleles):
        if i == 0:
            continue
        if a[0] == "X":
            return i
        if a[0] == "<" and len(a) >= 3 and ((a[1] == "X" and a[2] == ">") or (a[1] == "*" and a[2] == ">")):
            return i
    return 0


def is_snp(rec):
    """bcf_is_snp(): every allele is a single base or symbolic <*>/<X>/X-like (htslib bcf_set_variant_types)"""
    if len(rec.ref) != 1:
        return False
    for a in rec.alts:
        if a in ("<*>", "<X>", "X", "<NON_REF>"):
            continue
        if a.startswith("<"):
            return False
        if len(a) != 1:
            return False
    return True

"""Test-side SAM reader + pileup engine + tile packer (pure Python, small inputs only).

This stands in for htslib's BAM reader and bam_mplp pileup iterator (absent here)
so that the reference's SAM fixtures can drive the oracle and the HIP path.  It follows the
driver logic of mpileup.c:183-246 (read filters) and :275-293,:320-347 (grouping and per-site order).
"""
import re
import numpy as np

from bcftools_amd import abi

NT16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
NT16.update({c.lower(): i for c, i in list(NT16.items())})
CIG_OPS = "MIDNSHP=X"

BAM_FPAIRED, BAM_FPROPER_PAIR, BAM_FUNMAP, BAM_FREVERSE = 1, 2, 4, 16
BAM_FSECONDARY, BAM_FQCFAIL, BAM_FDUP = 256, 512, 1024


def nt16(ch):
    return NT16.get(ch, 15)


class Read:
    __slots__ = ("qname", "flag", "rname", "pos", "mapq", "cigar", "seq", "qual", "rg", "end", "l_qseq", "bamcigar",
                 "rnext", "mpos", "isize", "zq")

    def __init__(self, f):
        self.qname, self.flag, self.rname = f[0], int(f[1]), f[2]
        self.pos, self.mapq = int(f[3]) - 1, int(f[4])
        self.cigar = [(int(n), op) for n, op in re.findall(r"(\d+)([MIDNSHP=X])", f[5])]
        self.rnext, self.mpos, self.isize = f[6], int(f[7]) - 1, int(f[8])
        self.zq = None
        self.seq = f[9]
        self.qual = np.frombuffer(f[10].encode(), dtype=np.uint8).astype(np.int32) - 33 if f[10] != "*" \
            else np.full(len(f[9]), 255, dtype=np.int32)
        self.rg = None
        for t in f[11:]:
            if t.startswith("RG:Z:"):
                self.rg = t[5:]
        self.l_qseq = len(self.seq)
        self.end = self.pos + sum(n for n, op in self.cigar if op in "MDN=X")
        self.bamcigar = np.array([(n << 4) | CIG_OPS.index(op) for n, op in self.cigar], dtype=np.uint32)


class Sam:
    def __init__(self, path):
        self.rg2sm, self.samples, self.reads, self.contigs = {}, [], [], {}
        with open(path) as fh:
            for line in fh:
                line = line.rstrip("\n")
                if line.startswith("@"):
                    f = line.split("\t")
                    if f[0] == "@RG":
                        d = dict(x.split(":", 1) for x in f[1:])
                        self.rg2sm[d["ID"]] = d.get("SM", d["ID"])
                        if self.rg2sm[d["ID"]] not in self.samples:
                            self.samples.append(self.rg2sm[d["ID"]])
                    elif f[0] == "@SQ":
                        d = dict(x.split(":", 1) for x in f[1:])
                        self.contigs[d["SN"]] = int(d["LN"])
                    continue
                if line:
                    self.reads.append(Read(line.split("\t")))


class Bam(Sam):
    """The same from a BAM file (the reference keeps some fixtures as BAM only): BGZF blocks are gzip members, the
    records are turned into SAM fields and go through Read."""

    def __init__(self, path):
        import struct
        import zlib
        self.rg2sm, self.samples, self.reads, self.contigs = {}, [], [], {}
        raw, data = open(path, "rb").read(), b""
        while raw:
            d = zlib.decompressobj(31)
            data += d.decompress(raw)
            raw = d.unused_data
        assert data[:4] == b"BAM\1"
        l_text, = struct.unpack_from("<i", data, 4)
        text = data[8:8 + l_text].split(b"\0")[0].decode()
        o = 8 + l_text
        n_ref, = struct.unpack_from("<i", data, o)
        o += 4
        names = []
        for _ in range(n_ref):
            l_name, = struct.unpack_from("<i", data, o)
            names.append(data[o + 4:o + 4 + l_name - 1].decode())
            l_ref, = struct.unpack_from("<i", data, o + 4 + l_name)
            self.contigs[names[-1]] = l_ref
            o += 8 + l_name
        for line in text.split("\n"):
            f = line.split("\t")
            if f[0] == "@RG":
                d = dict(x.split(":", 1) for x in f[1:])
                self.rg2sm[d["ID"]] = d.get("SM", d["ID"])
                if self.rg2sm[d["ID"]] not in self.samples:
                    self.samples.append(self.rg2sm[d["ID"]])
        while o < len(data):
            bs, = struct.unpack_from("<i", data, o)
            rec = data[o + 4:o + 4 + bs]
            o += 4 + bs
            ref_id, pos, l_rn, mapq, _bin, n_cig, flag, l_seq, nref, npos, tlen = struct.unpack_from("<iiBBHHHiiii", rec, 0)
            q = 32
            qname = rec[q:q + l_rn - 1].decode()
            q += l_rn
            cig = struct.unpack_from("<%dI" % n_cig, rec, q)
            q += 4 * n_cig
            sq = rec[q:q + (l_seq + 1) // 2]
            q += (l_seq + 1) // 2
            seq = "".join("=ACMGRSVTWYHKDBN"[(sq[i >> 1] >> (0 if i & 1 else 4)) & 15] for i in range(l_seq))
            ql = rec[q:q + l_seq]
            q += l_seq
            qual = "*" if l_seq == 0 or ql[0] == 0xff else "".join(chr(c + 33) for c in ql)
            tags = []
            while q < len(rec):                                  # only RG:Z is of interest; the rest is skipped by type
                tag, typ = rec[q:q + 2].decode(), chr(rec[q + 2])
                q += 3
                if typ in "ZH":
                    e = rec.index(b"\0", q)
                    if tag == "RG":
                        tags.append("RG:Z:" + rec[q:e].decode())
                    q = e + 1
                elif typ == "B":
                    sub, n = chr(rec[q]), struct.unpack_from("<i", rec, q + 1)[0]
                    q += 5 + n * {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[sub]
                else:
                    q += {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[typ]
            cs = "".join("%d%s" % (c >> 4, CIG_OPS[c & 15]) for c in cig) or "*"
            rn = names[ref_id] if ref_id >= 0 else "*"
            rnext = "*" if nref < 0 else ("=" if nref == ref_id else names[nref])
            self.reads.append(Read([qname, str(flag), rn, str(pos + 1), str(mapq), cs, rnext, str(npos + 1), str(tlen),
                                    seq if l_seq else "*", qual] + tags))


def read_fasta(path):
    seqs, name = {}, None
    with open(path) as fh:
        for line in fh:
            line = line.strip()
            if line.startswith(">"):
                name = line[1:].split()[0]
                seqs[name] = []
            elif name:
                seqs[name].append(line)
    return {k: "".join(v) for k, v in seqs.items()}


def get_position(qpos, cigar):
    """bam2bcf.c:80-114: position within the aligned part of the read and its aligned length."""
    n_tot, iread, edist = 0, 0, qpos + 1
    for n, op in cigar:
        if op in "M=X" or op == "I":
            n_tot += n
            iread += n
        elif op == "S":
            iread += n
            if iread <= qpos:
                edist -= n
    return edist, n_tot


def pack_read(nt, baseQ, mapQ, is_rev, sclip, is_del, skip, qpos, l_qseq, cigar, want_epos=True):
    """Python twin of bcfgpu_pack_read() (include/bcfgpu.h)."""
    tail = min(qpos, l_qseq - 1 - qpos)
    tail = max(0, min(tail, 255))
    rd = (baseQ & 0xff) | ((mapQ & 0xff) << 8) | ((nt & 0xf) << 16) | (abi.RD_REV if is_rev else 0) \
        | (abi.RD_SCLIP if sclip else 0) | (abi.RD_DEL if is_del else 0) | (abi.RD_SKIP if skip else 0) | (tail << 24)
    epos = 0
    if want_epos:
        pos, ln = get_position(qpos, cigar)
        epos = min(max(int(float(pos) / (ln + 1) * 100), 0), 99)   # in range for every aligned qpos; clamp like the C packer
    return rd, epos


def walk(read, pos):
    """Pileup state of `read` at reference position `pos` (htslib bam_plp resolve_cigar semantics):
    returns (qpos, is_del, is_refskip, indel) or None when the read does not cover pos."""
    if pos < read.pos or pos >= read.end:
        return None
    x, y = read.pos, 0          # x: ref coordinate, y: query coordinate
    cig = read.cigar
    for k, (n, op) in enumerate(cig):
        if op in "M=X":
            if pos < x + n:
                qpos = y + (pos - x)
                indel = 0
                if pos == x + n - 1:            # last base of the block: look at the next operation
                    kk = k + 1
                    while kk < len(cig) and cig[kk][1] == "P":
                        kk += 1
                    if kk < len(cig):
                        if cig[kk][1] == "I":
                            indel = cig[kk][0]
                            # consecutive insertions/pads are merged by htslib
                            kk += 1
                            while kk < len(cig) and cig[kk][1] in "IP":
                                if cig[kk][1] == "I":
                                    indel += cig[kk][0]
                                kk += 1
                        elif cig[kk][1] == "D":
                            indel = -cig[kk][0]
                return qpos, 0, 0, indel
            x += n
            y += n
        elif op in "DN":
            if pos < x + n:
                return y, 1, 1 if op == "N" else 0, 0
            x += n
        elif op in "IS":
            y += n
    return None


class MplpOpts:
    """The read-level options of `bcftools mpileup` that matter for the fixtures (mpileup.c:937-950)."""

    def __init__(self, rflag_require=0, rflag_filter=BAM_FUNMAP | BAM_FSECONDARY | BAM_FQCFAIL | BAM_FDUP,
                 min_mq=0, no_orphan=True, min_baseQ=13, fmt_flag=abi.INFO_VDB | abi.INFO_RPB, cap_thres=0):
        self.cap_thres = cap_thres          # mpileup -C: sam_cap_mapq after BAQ, then the -q / orphan filters (mpileup.c:234-241)
        self.rflag_require, self.rflag_filter = rflag_require, rflag_filter
        self.min_mq, self.no_orphan = min_mq, no_orphan
        self.min_baseQ, self.fmt_flag = min_baseQ, fmt_flag


def keep_read_late(r, o):
    """The two filters of mplp_func that come after BAQ and -C (mpileup.c:240-241)."""
    if r.mapq < o.min_mq:
        return False
    if o.no_orphan and (r.flag & BAM_FPAIRED) and not (r.flag & BAM_FPROPER_PAIR):
        return False
    return True


def keep_read(r, o):
    """mplp_func, mpileup.c:183-246 (without BED and BAQ; with -C the last two filters wait for keep_read_late)."""
    if r.rname == "*" or (r.flag & BAM_FUNMAP):
        return False
    if o.rflag_require and not (o.rflag_require & r.flag):
        return False
    if o.rflag_filter and (o.rflag_filter & r.flag):
        return False
    if getattr(o, "cap_thres", 0) > 10:
        return True
    if r.mapq < o.min_mq:
        return False
    if o.no_orphan and (r.flag & BAM_FPAIRED) and not (r.flag & BAM_FPROPER_PAIR):
        return False
    return True


def build_tile(sams, ref, contig, beg, end, opts, samples=None, prep=None):
    """Pileup of region [beg,end] (0-based inclusive) over a list of Sam objects -> (HostTile pieces, positions).

    `prep(read, refseq)` may rewrite a read's qualities in place before the pileup (BAQ hook).
    Returns dict(ref16, plp_off, rd, epos, positions, samples, columns) where `columns` keeps the per-site,
    per-sample lists of (read, qpos, is_del, is_refskip, indel) for indel-side tests.
    """
    refseq = ref[contig]
    if samples is None:
        samples = []
        for s in sams:
            for sm in s.samples:
                if sm not in samples:
                    samples.append(sm)
    sm_idx = {s: i for i, s in enumerate(samples)}
    files = []
    for s in sams:
        rl = []
        for r in s.reads:
            if r.rname != contig or not keep_read(r, opts):
                continue
            sm = s.rg2sm.get(r.rg, s.samples[0] if s.samples else None)
            if sm not in sm_idx:
                continue
            if prep is not None:
                prep(r, refseq)
                if r.mapq < opts.min_mq:
                    continue
            rl.append((r, sm_idx[sm]))
        files.append(rl)
    S = len(samples)
    ref16, off, rd, epos, positions, columns = [], [0], [], [], [], []
    want_epos = bool(opts.fmt_flag & (abi.INFO_RPB | abi.INFO_VDB))
    for pos in range(beg, end + 1):
        per = [[] for _ in range(S)]
        tot = 0
        for rl in files:
            for r, si in rl:
                w = walk(r, pos)
                if w is None:
                    continue
                per[si].append((r,) + w)
                tot += 1
        if tot == 0:
            continue
        positions.append(pos)
        columns.append(per)
        rb = refseq[pos] if pos < len(refseq) else "N"
        ref16.append(nt16(rb))
        for si in range(S):
            for (r, qpos, is_del, is_refskip, indel) in per[si]:
                w, e = pack_read(nt16(r.seq[qpos]) if qpos < r.l_qseq else 15, int(r.qual[qpos]) if qpos < r.l_qseq else 0,
                                 r.mapq, bool(r.flag & BAM_FREVERSE), any(op == "S" for _, op in r.cigar),
                                 is_del, is_refskip, qpos, r.l_qseq, r.cigar, want_epos)
                rd.append(w)
                epos.append(e)
            off.append(len(rd))
    return dict(ref16=np.array(ref16, dtype=np.int8), plp_off=np.array(off, dtype=np.uint32),
                rd=np.array(rd, dtype=np.uint32), epos=np.array(epos, dtype=np.uint8),
                positions=positions, samples=samples, columns=columns)

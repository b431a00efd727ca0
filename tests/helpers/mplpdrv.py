"""Test-side driver of the mpileup stage on the reference's SAM fixtures with the *default* read preprocessing:
BAQ (sam_prob_realn, mpileup.c:234) and the mate-overlap quality tweak (bam_mplp_init_overlaps, mpileup.c:640),
then per-site glfgen/combine (SNP record) and bcf_call_gap_prep + glfgen/combine (indel record), mpileup.c:320-367.

BAQ, probaln, gap_prep and the per-base overlap tweak run in the C oracle (or through the HIP library when a context is
given); the pairing of mates and the pileup walk are restated here (htslib sam.c: overlap_push).  An *engine* (oracle
or HIP) turns tiles into results.
"""
import ctypes as C
import numpy as np

from bcftools_amd import abi, host
from . import sam as S, orc


def _realn_lib():
    L = orc.lib()
    L.orc_sam_prob_realn.restype = C.c_int
    L.orc_sam_prob_realn.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_int,
                                     C.c_int, C.c_void_p]
    L.orc_gap_prep.restype = C.c_int
    return L


NT4 = {"A": 0, "C": 1, "G": 2, "T": 3, "a": 0, "c": 1, "g": 2, "t": 3}


def apply_baq(read, refseq, flag=3):
    """sam_prob_realn(b, ref, ref_len, 3): qualities rewritten in place, ZQ kept on the read."""
    L = _realn_lib()
    if read.flag & S.BAM_FUNMAP or read.l_qseq == 0:
        return
    seq4 = np.array([NT4.get(c, 4) for c in read.seq], dtype=np.uint8)
    qual = np.ascontiguousarray(read.qual.astype(np.uint8))
    zq = np.zeros(read.l_qseq, dtype=np.uint8)
    rc = L.orc_sam_prob_realn(read.pos, read.l_qseq, seq4.ctypes.data, qual.ctypes.data, read.bamcigar.ctypes.data,
                              len(read.bamcigar), refseq.encode(), len(refseq), flag, zq.ctypes.data)
    if rc == 0:
        read.qual = qual.astype(np.int32)
        read.zq = zq
    return rc


def apply_baq_hip(reads, refseq, ctx, flag=3):
    """The same through bcfgpu_baq: one call for the whole list of reads."""
    from bcftools_amd.lib import check
    todo = [r for r in reads if not (r.flag & S.BAM_FUNMAP) and r.l_qseq > 0]
    if not todo:
        return
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    lq = i32([r.l_qseq for r in todo])
    ncig = i32([len(r.bamcigar) for r in todo])
    d = dict(r_pos=i32([r.pos for r in todo]), r_lq=lq, r_flag=i32([r.flag for r in todo]), r_ncig=ncig,
             r_cig_off=i32(np.concatenate([[0], np.cumsum(ncig)[:-1]])), r_seq_off=i32(np.concatenate([[0], np.cumsum(lq)[:-1]])),
             cig=np.ascontiguousarray(np.concatenate([r.bamcigar for r in todo]), dtype=np.uint32),
             seq16=np.ascontiguousarray(np.concatenate([[S.nt16(c) for c in r.seq] for r in todo]), dtype=np.uint8),
             qual=np.ascontiguousarray(np.concatenate([r.qual for r in todo]).astype(np.uint8)))
    d["zq"] = np.zeros(len(d["qual"]), dtype=np.uint8)
    d["r_has_zq"] = np.zeros(len(todo), dtype=np.uint8)
    rd = abi.Reads()
    rd.n_reads = len(todo)
    for k in ("r_pos", "r_lq", "r_flag", "r_ncig", "r_cig_off", "r_seq_off", "cig", "seq16", "qual", "zq", "r_has_zq"):
        setattr(rd, k, d[k].ctypes.data)
    qo, zo = np.zeros(len(d["qual"]), dtype=np.uint8), np.zeros(len(d["qual"]), dtype=np.uint8)
    ret = np.zeros(len(todo), dtype=np.int32)
    check(ctx.L.bcfgpu_baq(ctx.h, C.byref(rd), refseq.encode(), len(refseq), flag, qo.ctypes.data, zo.ctypes.data, ret.ctypes.data))
    for r, o, n, rc in zip(todo, d["r_seq_off"], lq, ret):
        if rc == 0:
            r.qual = qo[o:o + n].astype(np.int32)
            r.zq = zo[o:o + n].copy()
    return ret


def _iref2iseq(read):
    """reference position -> query index for M/=/X columns of a read."""
    m = {}
    x, y = read.pos, 0
    for n, op in read.cigar:
        if op in "M=X":
            for j in range(n):
                m[x + j] = y + j
            x += n
            y += n
        elif op in "DN":
            x += n
        elif op in "IS":
            y += n
    return m


def tweak_overlap_quality(a, b):
    """htslib sam.c tweak_overlap_quality(a,b): a arrived first."""
    ma, mb = _iref2iseq(a), _iref2iseq(b)
    for pos in sorted(set(ma) & set(mb)):
        if pos < b.pos:
            continue
        ia, ib = ma[pos], mb[pos]
        if a.seq[ia].upper() == b.seq[ib].upper():
            q = int(a.qual[ia]) + int(b.qual[ib])
            a.qual[ia] = 200 if q > 200 else q
            b.qual[ib] = 0
        else:
            if a.qual[ia] >= b.qual[ib]:
                a.qual[ia] = int(0.8 * a.qual[ia])
                b.qual[ib] = 0
            else:
                b.qual[ib] = int(0.8 * b.qual[ib])
                a.qual[ia] = 0


def overlap_pairs(reads):
    """overlap_push over the reads of one file in file order (all reads stay buffered long enough in these fixtures):
    the (first mate, second mate) pairs that htslib hands to tweak_overlap_quality."""
    pending, pairs = {}, []
    for r in reads:
        f = r.flag
        if (f & 8) or not (f & S.BAM_FPROPER_PAIR):          # BAM_FMUNMAP
            continue
        if r.rnext not in ("=", r.rname) or (abs(r.isize) >= 2 * r.l_qseq and r.mpos >= r.end):
            continue
        if r.qname not in pending:
            if r.mpos >= r.pos or ((f & S.BAM_FPAIRED) and r.mpos == -1):
                pending[r.qname] = r
        else:
            a = pending.pop(r.qname)
            # the first mate must still be in the pileup buffer: it is, unless it ended before this one starts
            if a.end > r.pos:
                pairs.append((a, r))
    return pairs


def pack_reads(reads):
    """The flat read pool of include/bcfgpu.h (bcfgpu_reads) for a list of sam.Read; returns (abi.Reads, arrays kept alive)."""
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    lq = i32([r.l_qseq for r in reads])
    ncig = i32([len(r.bamcigar) for r in reads])
    d = dict(r_pos=i32([r.pos for r in reads]), r_lq=lq, r_flag=i32([r.flag for r in reads]), r_ncig=ncig,
             r_cig_off=i32(np.concatenate([[0], np.cumsum(ncig)[:-1]])), r_seq_off=i32(np.concatenate([[0], np.cumsum(lq)[:-1]])),
             cig=np.ascontiguousarray(np.concatenate([r.bamcigar for r in reads]), dtype=np.uint32),
             seq16=np.ascontiguousarray(np.concatenate([[S.nt16(c) for c in r.seq] for r in reads]), dtype=np.uint8),
             qual=np.ascontiguousarray(np.concatenate([r.qual for r in reads]).astype(np.uint8)))
    d["zq"] = np.zeros(len(d["qual"]), dtype=np.uint8)
    d["r_has_zq"] = np.zeros(len(reads), dtype=np.uint8)
    rd = abi.Reads()
    rd.n_reads = len(reads)
    for k in ("r_pos", "r_lq", "r_flag", "r_ncig", "r_cig_off", "r_seq_off", "cig", "seq16", "qual", "zq", "r_has_zq"):
        setattr(rd, k, d[k].ctypes.data)
    return rd, d


def apply_overlaps(reads, ctx=None, python=False):
    """The mate-overlap tweak on the reads of one file: pair selection here, the per-base arithmetic in the C oracle
    (orc_overlap_tweak), through bcfgpu_overlap_tweak when a HIP context is given, or in Python (the first restatement,
    kept to pin the other two)."""
    pairs = overlap_pairs(reads)
    if python:
        for a, b in pairs:
            tweak_overlap_quality(a, b)
        return len(pairs)
    if not pairs:
        return 0
    used = [r for ab in pairs for r in ab]
    rd, d = pack_reads(used)
    pa = np.arange(0, len(used), 2, dtype=np.int32)
    pb = pa + 1
    if ctx is None:
        L = orc.lib()
        L.orc_overlap_tweak.restype = C.c_int
        L.orc_overlap_tweak.argtypes = [C.POINTER(abi.Reads), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        qo = d["qual"].copy()
        rc = L.orc_overlap_tweak(C.byref(rd), len(pairs), pa.ctypes.data, pb.ctypes.data, qo.ctypes.data)
        assert rc == 0
    else:
        from bcftools_amd.lib import check
        qo = np.zeros_like(d["qual"])
        check(ctx.L.bcfgpu_overlap_tweak(ctx.h, C.byref(rd), len(pairs), pa.ctypes.data, pb.ctypes.data, qo.ctypes.data))
    for r, o in zip(used, d["r_seq_off"]):
        r.qual = qo[o:o + r.l_qseq].astype(np.int32)
    return len(pairs)


def cap_mapq_oracle(r, refseq, thres):
    """orc_cap_mapq (oracle/capmapq.c: sam_cap_mapq of htslib) for one read object with its current qualities."""
    L = _realn_lib()
    cig = np.array([(n << 4) | "MIDNSHP=X".index(op) for n, op in r.cigar], dtype=np.uint32)
    seq16 = np.array([S.nt16(c) for c in r.seq], dtype=np.uint8)
    qual = np.ascontiguousarray(r.qual, dtype=np.uint8)
    L.orc_cap_mapq.restype = C.c_int
    L.orc_cap_mapq.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    return L.orc_cap_mapq(r.pos, len(cig), cig.ctypes.data, seq16.ctypes.data, qual.ctypes.data, refseq.encode(), len(refseq), thres)


class Prepared:
    """SAM files after mplp_func filtering + BAQ + overlap tweak, ready for pileup."""

    def __init__(self, sams, ref, contig, opts, baq=True, overlaps=True, baq_ctx=None, rg_map=None, samples=None):
        """rg_map / samples: `mpileup -G FILE` (bam_sample.c): read group -> sample name, reads of other read groups are
        dropped; `samples` fixes the output order (it may name samples that end up without reads)."""
        self.refseq = ref[contig]
        self.contig = contig
        self.opts = opts
        self.samples = list(samples) if samples else []
        if not samples:
            for s in sams:
                for sm in s.samples:
                    if sm not in self.samples:
                        self.samples.append(sm)
        self.files = []
        for s in sams:
            rl = []
            for r in s.reads:
                if r.rname != contig or not S.keep_read(r, opts):
                    continue
                if rg_map is not None and r.rg not in rg_map:
                    continue
                r.zq = None
                if baq and baq_ctx is None:
                    apply_baq(r, self.refseq)
                sm = rg_map[r.rg] if rg_map is not None else s.rg2sm.get(r.rg, s.samples[0] if s.samples else None)
                rl.append((r, self.samples.index(sm)))
            if baq and baq_ctx is not None:
                apply_baq_hip([r for r, _ in rl], self.refseq, baq_ctx)      # bcfgpu_baq, one call per file
            if getattr(opts, "cap_thres", 0) > 10:                           # mpileup -C (mpileup.c:235-241)
                kept = []
                for r, si in rl:
                    q = cap_mapq_oracle(r, self.refseq, opts.cap_thres)
                    if q < 0:
                        continue
                    r.mapq = min(r.mapq, q)
                    if S.keep_read_late(r, opts):
                        kept.append((r, si))
                rl = kept
            if overlaps:
                apply_overlaps([r for r, _ in rl], ctx=baq_ctx)      # oracle C, or bcfgpu_overlap_tweak with a HIP context
            self.files.append(rl)


def column(prep, pos):
    per = [[] for _ in prep.samples]
    for rl in prep.files:
        for r, si in rl:
            w = S.walk(r, pos)
            if w is not None:
                per[si].append((r,) + w)
    return per


def snp_tile(prep, positions):
    """Pack SNP columns (like sam.build_tile but on prepared reads); returns (HostTile, columns)."""
    o = prep.opts
    want_epos = bool(o.fmt_flag & (abi.INFO_RPB | abi.INFO_VDB))
    ref16, off, rd, epos, cols, kept = [], [0], [], [], [], []
    for pos in positions:
        per = column(prep, pos)
        if sum(len(x) for x in per) == 0:
            continue
        kept.append(pos)
        cols.append(per)
        rb = prep.refseq[pos] if pos < len(prep.refseq) else "N"
        ref16.append(S.nt16(rb))
        for lst in per:
            for (r, qpos, is_del, is_refskip, indel) in lst:
                w, e = S.pack_read(S.nt16(r.seq[qpos]) if qpos < r.l_qseq else 15,
                                   int(r.qual[qpos]) if qpos < r.l_qseq else 0, r.mapq,
                                   bool(r.flag & S.BAM_FREVERSE), any(op == "S" for _, op in r.cigar),
                                   is_del, is_refskip, qpos, r.l_qseq, r.cigar, want_epos)
                rd.append(w)
                epos.append(e)
            off.append(len(rd))
    tile = host.HostTile(len(prep.samples), np.array(ref16, dtype=np.int8), np.array(off, dtype=np.uint32),
                         np.array(rd, dtype=np.uint32), np.array(epos, dtype=np.uint8))
    return tile, cols, kept


def _flat_reads(per):
    """Flat read pool + pileup entries of one column (the layout of bcfgpu_reads / bcfgpu_indel_in)."""
    reads, ridx = [], {}
    smpl_off, p_read, p_qpos, p_indel = [0], [], [], []
    for lst in per:
        for (r, qpos, is_del, is_refskip, indel) in lst:
            if id(r) not in ridx:
                ridx[id(r)] = len(reads)
                reads.append(r)
            p_read.append(ridx[id(r)])
            p_qpos.append(qpos)
            p_indel.append(indel)
        smpl_off.append(len(p_read))
    if not p_read:
        return None
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    d = {}
    d["r_pos"] = i32([r.pos for r in reads]); d["r_lq"] = i32([r.l_qseq for r in reads]); d["r_flag"] = i32([r.flag for r in reads])
    d["r_ncig"] = i32([len(r.bamcigar) for r in reads])
    d["r_cig_off"] = i32(np.concatenate([[0], np.cumsum(d["r_ncig"])[:-1]]))
    d["cig"] = np.ascontiguousarray(np.concatenate([r.bamcigar for r in reads]), dtype=np.uint32)
    d["r_seq_off"] = i32(np.concatenate([[0], np.cumsum(d["r_lq"])[:-1]]))
    d["seq16"] = np.ascontiguousarray(np.concatenate([[S.nt16(c) for c in r.seq] for r in reads]), dtype=np.uint8)
    d["qual"] = np.ascontiguousarray(np.concatenate([r.qual for r in reads]).astype(np.uint8))
    d["r_has_zq"] = np.ascontiguousarray([1 if r.zq is not None else 0 for r in reads], dtype=np.uint8)
    d["zq"] = np.ascontiguousarray(np.concatenate([r.zq if r.zq is not None else np.zeros(r.l_qseq, dtype=np.uint8) for r in reads]), dtype=np.uint8)
    d["smpl_off"], d["p_read"], d["p_qpos"], d["p_indel"] = i32(smpl_off), i32(p_read), i32(p_qpos), i32(p_indel)
    d["n_reads"] = len(reads)
    return d


def gap_prep(prep, per, pos, openQ=40, extQ=20, tandemQ=100, min_support=1, min_frac=0.002, per_sample_flt=0, ctx=None):
    """bcf_call_gap_prep through the oracle (ctx=None) or through bcfgpu_gap_prep (ctx = engine.Context).
    Returns None (ret<0) or dict(aux, indel_types, inscns, maxins, indelreg, max_support, max_frac)."""
    d = _flat_reads(per)
    if d is None:
        return None
    aux = np.zeros(len(d["p_read"]), dtype=np.uint32)
    types = np.zeros(4, dtype=np.int32)
    CAP = 256
    inscns = np.zeros(4 * CAP, dtype=np.int8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    if ctx is None:
        L = _realn_lib()
        maxins, indelreg, msup = C.c_int(), C.c_int(), C.c_int()
        mfrac = C.c_float()
        L.orc_gap_prep.argtypes = [C.c_int] + [C.c_void_p] * 15 + [C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                                   C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        rc = L.orc_gap_prep(len(per), p(d["smpl_off"]), p(d["p_read"]), p(d["p_qpos"]), p(d["p_indel"]), p(d["r_pos"]),
                            p(d["r_lq"]), p(d["r_flag"]), p(d["r_ncig"]), p(d["r_cig_off"]), p(d["cig"]), p(d["r_seq_off"]),
                            p(d["seq16"]), p(d["qual"]), p(d["zq"]), p(d["r_has_zq"]),
                            pos, prep.refseq.encode(), openQ, extQ, tandemQ, min_support, min_frac, per_sample_flt,
                            p(aux), p(types), p(inscns), len(inscns), C.byref(maxins), C.byref(indelreg), C.byref(msup), C.byref(mfrac))
        if rc < 0:
            return None
        return dict(aux=aux, smpl_off=d["smpl_off"], indel_types=types.tolist(), inscns=inscns, maxins=maxins.value,
                    indelreg=indelreg.value, max_support=msup.value, max_frac=float(mfrac.value))
    # the HIP library: one candidate position per call here (the ABI takes batches)
    from bcftools_amd.lib import check
    rd = abi.Reads()
    rd.n_reads = d["n_reads"]
    for k in ("r_pos", "r_lq", "r_flag", "r_ncig", "r_cig_off", "r_seq_off", "cig", "seq16", "qual", "zq", "r_has_zq"):
        setattr(rd, k, d[k].ctypes.data)
    posa = np.array([pos], dtype=np.int32)
    ii = abi.IndelIn()
    ii.n_sites, ii.n_smpl = 1, len(per)
    ii.pos, ii.smpl_off, ii.p_read, ii.p_qpos, ii.p_indel = (posa.ctypes.data, d["smpl_off"].ctypes.data, d["p_read"].ctypes.data,
                                                             d["p_qpos"].ctypes.data, d["p_indel"].ctypes.data)
    refb = prep.refseq.encode()
    ii.ref = refb
    ii.openQ, ii.extQ, ii.tandemQ, ii.min_support, ii.per_sample_flt, ii.min_frac = openQ, extQ, tandemQ, min_support, per_sample_flt, min_frac
    ret = np.zeros(1, dtype=np.int32); maxins = np.zeros(1, dtype=np.int32); indelreg = np.zeros(1, dtype=np.int32)
    msup = np.zeros(1, dtype=np.int32); mfrac = np.zeros(1, dtype=np.float32)
    oo = abi.IndelOut()
    oo.ret, oo.p_aux, oo.indel_types, oo.inscns = ret.ctypes.data, aux.ctypes.data, types.ctypes.data, inscns.ctypes.data
    oo.maxins, oo.indelreg, oo.max_support, oo.max_frac = maxins.ctypes.data, indelreg.ctypes.data, msup.ctypes.data, mfrac.ctypes.data
    check(ctx.L.bcfgpu_gap_prep(ctx.h, C.byref(rd), C.byref(ii), C.byref(oo), CAP))
    if ret[0] < 0:
        return None
    return dict(aux=aux, smpl_off=d["smpl_off"], indel_types=types.tolist(), inscns=inscns, maxins=int(maxins[0]),
                indelreg=int(indelreg[0]), max_support=int(msup[0]), max_frac=float(mfrac[0]))


def indel_tile(prep, per, g):
    """The indel pass at one position: same reads, ref_base=-1, aux from gap_prep."""
    o = prep.opts
    want_epos = bool(o.fmt_flag & (abi.INFO_RPB | abi.INFO_VDB))
    off, rd, epos = [0], [], []
    for lst in per:
        for (r, qpos, is_del, is_refskip, indel) in lst:
            w, e = S.pack_read(S.nt16(r.seq[qpos]) if qpos < r.l_qseq else 15, int(r.qual[qpos]) if qpos < r.l_qseq else 0,
                               r.mapq, bool(r.flag & S.BAM_FREVERSE), any(op == "S" for _, op in r.cigar),
                               is_del, is_refskip, qpos, r.l_qseq, r.cigar, want_epos)
            rd.append(w)
            epos.append(e)
        off.append(len(rd))
    return host.HostTile(len(prep.samples), np.array([0], dtype=np.int8), np.array(off, dtype=np.uint32),
                         np.array(rd, dtype=np.uint32), np.array(epos, dtype=np.uint8), aux=g["aux"], is_indel=1)


def indel_alleles(refseq, pos, site, g):
    """REF/ALT strings of an indel record, bam2bcf.c:767-790."""
    ref = refseq
    indelreg = g["indelreg"]
    REF = ref[pos] + ref[pos + 1: pos + 1 + indelreg]
    alts = []
    for i in range(1, 4):
        ai = int(site["a"][i])
        if ai < 0:
            break
        t = g["indel_types"][ai]
        s = ref[pos]
        if t < 0:
            s += ref[pos + 1 - t: pos + 1 + indelreg]
        else:
            ins = g["inscns"][ai * g["maxins"]: ai * g["maxins"] + t]
            s += "".join("ACGTN"[int(c)] for c in ins) + ref[pos + 1: pos + 1 + indelreg]
        alts.append(s)
    return [REF] + alts

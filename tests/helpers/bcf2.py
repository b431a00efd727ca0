"""An independent reading of the BCF2.2 specification (the VCF specification, section 6), used to check what
host/vcfio.c writes: BGZF blocks -> "BCF\\2\\2" + header text -> records decoded to VCF text lines.  Written from the
specification's tables, not from vcfio.c: dictionaries are rebuilt from the header text (FILTER/INFO/FORMAT ids in
order of appearance with PASS = 0; contigs in order)."""
import re
import struct
import zlib

INT_MISSING = {1: -128, 2: -32768, 3: -2147483648}
INT_VEND = {1: -127, 2: -32767, 3: -2147483647}
FMT = {1: "<b", 2: "<h", 3: "<i"}
F_MISSING, F_VEND = 0x7F800001, 0x7F800002


def bgzf_blocks(data):
    """[(compressed size, uncompressed bytes)] of every BGZF block, checking the framing the specification demands."""
    out, p = [], 0
    while p < len(data):
        assert data[p:p + 4] == b"\x1f\x8b\x08\x04", "gzip member with FEXTRA expected"
        xlen = struct.unpack_from("<H", data, p + 10)[0]
        extra = data[p + 12:p + 12 + xlen]
        assert extra[:4] == b"BC\x02\x00"
        bsize = struct.unpack_from("<H", extra, 4)[0] + 1
        cdata = data[p + 12 + xlen:p + bsize - 8]
        crc, isize = struct.unpack_from("<II", data, p + bsize - 8)
        raw = zlib.decompress(cdata, -15)
        assert len(raw) == isize and zlib.crc32(raw) == crc and isize <= 65536
        out.append((bsize, raw))
        p += bsize
    assert out and out[-1][1] == b"" and out[-1][0] == 28, "the empty end-of-file block"
    return out


class Hdr:
    def __init__(self, text):
        self.text = text
        d, c = {"PASS": 0}, {}
        for ln in text.splitlines():
            m = re.match(r"##(FILTER|INFO|FORMAT|contig)=<ID=([^,>]+)", ln)
            if not m:
                continue
            x = re.search(r",IDX=(\d+)>$", ln)
            tab = c if m.group(1) == "contig" else d
            if m.group(2) not in tab:
                tab[m.group(2)] = int(x.group(1)) if x else len(tab)        # IDX, where given, is the index
        self.dict = {v: k for k, v in d.items()}
        self.contigs = {v: k for k, v in c.items()}
        last = [ln for ln in text.splitlines() if ln.startswith("#CHROM")][0].split("\t")
        self.samples = last[9:]

    def vcf_text(self):
        """the header as VCF text: IDX is a BCF-only attribute"""
        return "\n".join(re.sub(r",IDX=\d+>$", ">", ln) for ln in self.text.splitlines())


def _size(b, p):
    n, t = b[p] >> 4, b[p] & 15
    p += 1
    if n == 15:
        n1, t1, p = _size(b, p)
        assert n1 == 1
        n = struct.unpack_from(FMT[t1], b, p)[0]
        p += struct.calcsize(FMT[t1])
    return n, t, p


def _ints(b, p, n, t):
    w = struct.calcsize(FMT[t])
    return [struct.unpack_from(FMT[t], b, p + i * w)[0] for i in range(n)], p + n * w


def _g(bits):
    if bits == F_MISSING:
        return "."
    return "%g" % struct.unpack("<f", struct.pack("<I", bits))[0]


def _vals(b, p, n, t):
    """text of a typed vector (INFO value or one sample's FORMAT value), new offset"""
    if t == 7:
        s = b[p:p + n].split(b"\0")[0].decode()
        return s, p + n
    if t == 5:
        bits = struct.unpack_from("<%dI" % n, b, p)
        out = []
        for x in bits:
            if x == F_VEND:
                break
            out.append(_g(x))
        return ",".join(out), p + 4 * n
    v, q = _ints(b, p, n, t)
    out = []
    for x in v:
        if x == INT_VEND[t]:
            break
        out.append("." if x == INT_MISSING[t] else str(x))
    return ",".join(out), q


def records(raw, hdr, off):
    """VCF text lines of the records that follow the header"""
    lines = []
    while off < len(raw):
        l_shared, l_indiv = struct.unpack_from("<II", raw, off)
        sh = raw[off + 8:off + 8 + l_shared]
        ind = raw[off + 8 + l_shared:off + 8 + l_shared + l_indiv]
        off += 8 + l_shared + l_indiv
        chrom, pos, rlen, qual, nai, nfs = struct.unpack_from("<iiiIII", sh, 0)
        n_allele, n_info, n_fmt, n_sample = nai >> 16, nai & 0xffff, nfs >> 24, nfs & 0xffffff
        p = 24
        n, t, p = _size(sh, p)
        vid = sh[p:p + n].decode() if n else "."
        p += n
        als = []
        for _ in range(n_allele):
            n, t, p = _size(sh, p)
            assert t == 7
            als.append(sh[p:p + n].decode())
            p += n
        n, t, p = _size(sh, p)
        flt = "."
        if n:
            v, p = _ints(sh, p, n, t)
            flt = ";".join(hdr.dict[x] for x in v)
        info = []
        for _ in range(n_info):
            n, t, p = _size(sh, p)
            assert n == 1
            (k,), p = _ints(sh, p, 1, t)
            n, t, p = _size(sh, p)
            if n == 0:
                info.append(hdr.dict[k])
            else:
                s, p = _vals(sh, p, n, t)
                info.append(hdr.dict[k] + "=" + s)
        assert p == len(sh)
        cols = [hdr.contigs[chrom], str(pos + 1), vid, als[0], ",".join(als[1:]) or ".", _g(qual), flt, ";".join(info) or "."]
        if n_sample:
            keys, per = [], [[] for _ in range(n_sample)]
            p = 0
            for _ in range(n_fmt):
                n, t, p = _size(ind, p)
                (k,), p = _ints(ind, p, 1, t)
                n, t, p = _size(ind, p)
                keys.append(hdr.dict[k])
                for s in range(n_sample):
                    if hdr.dict[k] == "GT":
                        v, p = _ints(ind, p, n, t)
                        txt = ""
                        for j, x in enumerate(v):
                            if x == INT_VEND[t]:
                                break
                            txt += ("|" if x & 1 else "/") if j else ""
                            txt += str((x >> 1) - 1) if x >> 1 else "."
                        per[s].append(txt)
                    else:
                        sv, p = _vals(ind, p, n, t)
                        per[s].append(sv or ".")
            assert p == len(ind)
            cols.append(":".join(keys))
            cols += [":".join(x) for x in per]
        lines.append("\t".join(cols))
        assert rlen >= 1
    return lines


def read(path):
    """(header text, [VCF record lines]) of a BCF2 file"""
    data = open(path, "rb").read()
    raw = data if data[:5] == b"BCF\x02\x02" else b"".join(r for _, r in bgzf_blocks(data))     # -Ou: no BGZF framing
    assert raw[:5] == b"BCF\x02\x02"
    l_text = struct.unpack_from("<I", raw, 5)[0]
    text = raw[9:9 + l_text]
    assert text.endswith(b"\0")
    hdr = Hdr(text.rstrip(b"\0").decode())
    return hdr.vcf_text(), records(raw, hdr, 9 + l_text)

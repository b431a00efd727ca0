"""host/vcfio.c (VCF headers, BGZF, the VCF <-> BCF2 codec of the host drivers) without a GPU:
every golden VCF of the reference's mpileup / call tests goes VCF -> BCF2 -> VCF (and through bgzipped VCF) with
host/bcfgpu_view and comes back byte for byte; the BCF2 bytes are read by an independent decoder written from the
specification (tests/helpers/bcf2.py); known answers of the typed-value encoding from the specification's text."""
import glob
import os
import struct
import subprocess

import pytest

from tests.helpers import bcf2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VIEW = os.path.join(ROOT, "host", "bcfgpu_view")


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host"), "bcfgpu_view"])


def goldens(golden_dir):
    fs = sorted(glob.glob(os.path.join(golden_dir, "mpileup", "*.out")) + glob.glob(os.path.join(golden_dir, "call", "*.out")) +
                glob.glob(os.path.join(golden_dir, "call", "*.vcf")))
    return [f for f in fs if open(f).readline().startswith("##fileformat")]


PASS_LINE = '##FILTER=<ID=PASS,Description="All filters passed">'


def with_pass(text):
    """A header that does not declare PASS gets the line right after ##fileformat, as htslib's header parser adds it
    (every `call` golden made from such an input shows it there)."""
    lines = text.split("\n")
    if not any(ln.startswith("##FILTER=<ID=PASS") for ln in lines):
        lines.insert(1, PASS_LINE)
    return "\n".join(lines)


def test_goldens_survive_every_output_mode(golden_dir, tmp_path):
    build()
    fs = goldens(golden_dir)
    assert len(fs) > 40
    for f in fs:
        want = with_pass(open(f).read()).encode()
        for mode in "ubz":
            mid = str(tmp_path / ("x." + mode))
            subprocess.check_call([VIEW, "-O", mode, "-o", mid, f])
            head = open(mid, "rb").read(5)
            assert head == b"BCF\x02\x02" if mode == "u" else head[:2] == b"\x1f\x8b"     # -Ou: the BCF stream itself (htslib "wbu"); -Ob / -Oz: BGZF
            back = subprocess.run([VIEW, mid], check=True, stdout=subprocess.PIPE).stdout
            assert back == want, (f, mode)
        # BCF -> BCF keeps the bytes of the records (the text form is a faithful intermediate)
        a, b = str(tmp_path / "a.bcf"), str(tmp_path / "b.bcf")
        subprocess.check_call([VIEW, "-O", "u", "-o", a, f])
        subprocess.check_call([VIEW, "-O", "u", "-o", b, a])
        assert open(a, "rb").read() == open(b, "rb").read()


def test_bcf_bytes_read_by_an_independent_decoder(golden_dir, tmp_path):
    build()
    n_rec = 0
    for f in goldens(golden_dir):
        for mode in "ub":
            out = str(tmp_path / "y.bcf")
            subprocess.check_call([VIEW, "-O", mode, "-o", out, f])
            text, lines = bcf2.read(out)
            src = with_pass(open(f).read()).splitlines()
            assert text.splitlines() == [ln for ln in src if ln.startswith("#")]
            assert lines == [ln for ln in src if not ln.startswith("#")], f
            n_rec += len(lines)
    assert n_rec > 5000


def test_typed_value_known_answers(tmp_path):
    """The encodings the specification spells out: size/type byte, the smallest integer type, missing and end-of-vector
    values, PASS = dictionary entry 0, missing QUAL, GT as (allele + 1) << 1 | phased."""
    build()
    vcf = tmp_path / "k.vcf"
    vcf.write_text("##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n##contig=<ID=c1,length=1000>\n"
                   "##INFO=<ID=A,Number=1,Type=Integer,Description=\"x\">\n##INFO=<ID=B,Number=.,Type=Integer,Description=\"x\">\n"
                   "##INFO=<ID=F,Number=0,Type=Flag,Description=\"x\">\n##INFO=<ID=R,Number=1,Type=Float,Description=\"x\">\n"
                   "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"x\">\n##FORMAT=<ID=PL,Number=G,Type=Integer,Description=\"x\">\n"
                   "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts1\ts2\n"
                   "c1\t10\t.\tA\tC,<*>\t.\tPASS\tA=300;B=1,-2,.;F;R=0.5\tGT:PL\t0|1:0,300,.\t.:5\n")
    out = str(tmp_path / "k.bcf")
    subprocess.check_call([VIEW, "-O", "u", "-o", out, str(vcf)])
    raw = open(out, "rb").read()                                               # -Ou: the BCF stream itself, no BGZF framing
    assert raw[:5] == b"BCF\x02\x02"
    l_text = struct.unpack_from("<I", raw, 5)[0]
    rec = raw[9 + l_text:]
    l_shared, l_indiv = struct.unpack_from("<II", rec, 0)
    sh, ind = rec[8:8 + l_shared], rec[8 + l_shared:8 + l_shared + l_indiv]
    assert struct.unpack_from("<iii", sh, 0) == (0, 9, 1)                      # CHROM index, 0-based POS, rlen
    assert struct.unpack_from("<I", sh, 12)[0] == 0x7F800001                   # missing QUAL
    assert struct.unpack_from("<II", sh, 16) == (3 << 16 | 4, 2 << 24 | 2)     # n_allele | n_info, n_fmt | n_sample
    p = 24
    assert sh[p:p + 1] == b"\x07"                                              # missing ID: a character vector of length 0
    assert sh[p + 1:p + 9] == b"\x17A\x17C\x37<*>"                            # alleles as typed strings
    assert sh[p + 9:p + 11] == b"\x11\x00"                                    # FILTER: one int8, PASS = 0
    q = p + 11
    assert sh[q:q + 5] == b"\x11\x01\x12" + struct.pack("<h", 300)             # A (dictionary 1) = 300: int16
    assert sh[q + 5:q + 11] == b"\x11\x02\x31\x01\xfe\x80"                     # B = 1,-2,missing as int8
    assert sh[q + 11:q + 14] == b"\x11\x03\x00"                                # F: a flag carries no value
    assert sh[q + 14:q + 21] == b"\x11\x04\x15" + struct.pack("<f", 0.5)
    assert ind[:7] == b"\x11\x05\x21" + bytes([2, 5, 0, 0x81])                 # GT 0|1 -> 2,5; "." -> 0 then end-of-vector
    w = struct.pack("<6h", 0, 300, -32768, 5, -32767, -32767)                  # PL as int16: ".", then end-of-vector padding
    assert ind[7:] == b"\x11\x06\x32" + w


def test_truncated_and_malformed_bcf_is_refused_not_overrun(tmp_path, golden_dir):
    """A BCF whose record claims more bytes than it has, or whose typed vectors run past the record, makes bcfgpu_view fail
    with "truncated BCF record" instead of reading out of bounds."""
    build()
    f = goldens(golden_dir)[0]
    out = str(tmp_path / "t.bcf")
    subprocess.check_call([VIEW, "-O", "u", "-o", out, f])
    raw = bytearray(open(out, "rb").read())
    l_text = struct.unpack_from("<I", raw, 5)[0]
    at = 9 + l_text
    l_shared, l_indiv = struct.unpack_from("<II", raw, at)
    cases = []
    cut = bytes(raw[:at + 8 + l_shared // 2])                                  # the file ends inside the first record
    cases.append(cut)
    bad = bytearray(raw)
    bad[at + 8 + 24] = 0xf7                                                    # ID: a character vector with a 15+ length that is not there
    cases.append(bytes(bad))
    bad = bytearray(raw)
    struct.pack_into("<I", bad, at + 8 + 20, (1 << 24) | 0xffffff)             # 16 M samples in a record of a few hundred bytes
    cases.append(bytes(bad))
    for i, data in enumerate(cases):
        p = str(tmp_path / ("bad%d.bcf" % i))
        open(p, "wb").write(data)
        r = subprocess.run([VIEW, p], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode != 0, i


def test_records_from_integer_columns_are_the_text_line_byte_for_byte(golden_dir, tmp_path):
    """vio_write_record_int (what host/bcfgpu_sam writes its records through: the per-sample columns as integer arrays) against
    vio_write_line of the same record's text, on every mpileup golden: the BCF2 bytes and the VCF text must not differ, and the
    records must really have gone through it."""
    build()
    total = 0
    for f in goldens(golden_dir):
        if os.sep + "mpileup" + os.sep not in f:
            continue
        for mode in "uv":
            a, b = str(tmp_path / ("a." + mode)), str(tmp_path / ("b." + mode))
            subprocess.check_call([VIEW, "-O", mode, "-o", a, f])
            p = subprocess.run([VIEW, "--int-columns", "-O", mode, "-o", b, f], stderr=subprocess.PIPE, universal_newlines=True, check=True)
            assert open(a, "rb").read() == open(b, "rb").read(), (f, mode)
            if mode == "u":
                total += int(p.stderr.split()[0])
    assert total > 5000


def test_integer_columns_wider_than_any_sample_needs(golden_dir, tmp_path):
    """A caller with fixed-width planes hands over columns in which EVERY sample ends early (VIO_INT_VEND in the last places): the
    vector written is as long as the longest sample's, as the text path sizes it -- the same BCF2 bytes, not a wider vector."""
    build()
    f = [g for g in goldens(golden_dir) if g.endswith(os.path.join("mpileup", "mpileup.2.out"))][0]
    a, b = str(tmp_path / "a.u"), str(tmp_path / "b.u")
    subprocess.check_call([VIEW, "-O", "u", "-o", a, f])
    p = subprocess.run([VIEW, "--int-columns-pad", "2", "-O", "u", "-o", b, f], stderr=subprocess.PIPE, universal_newlines=True, check=True)
    assert int(p.stderr.split()[0]) > 100
    assert open(a, "rb").read() == open(b, "rb").read()

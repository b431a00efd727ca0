"""The plain-C host driver (host/bcfgpu_host.c): the C-ABI used the way a C caller would (gcc, -std=c99).

CPU part: it compiles warning-free against include/bcfgpu.h, links against the library, and fails loudly without a GPU.
GPU part: its printed call records equal what the oracle gives for the same pileup (the generator is restated here)."""
import os
import subprocess

import numpy as np
import pytest

from bcftools_amd import abi, host
from tests.helpers import orc, sam

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "host", "bcfgpu_host")
READ_LEN = 100
M64 = (1 << 64) - 1


def build_host():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "bcftools_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    assert os.access(EXE, os.X_OK)


class Rng:
    """xorshift64* of host/bcfgpu_host.c"""
    def __init__(self, seed):
        self.s = (seed * 2 + 1) & M64

    def r32(self):
        s = self.s
        s ^= s >> 12
        s ^= (s << 25) & M64
        s ^= s >> 27
        self.s = s
        return ((s * 2685821657736338717) & M64) >> 32

    def below(self, n):
        return (self.r32() * n) >> 32


def host_tile(n_sites, n_smpl, depth, seed):
    """The pileup bcfgpu_host.c generates, packed with the Python twin of bcfgpu_pack_read."""
    g = Rng(seed)
    bqv = [11, 25, 37, 40]
    ref16 = np.zeros(n_sites, dtype=np.int8)
    off = [0]
    rd, ep = [], []
    for k in range(n_sites):
        ref2 = g.below(4)
        alt2 = (ref2 + 1 + g.below(3)) & 3
        is_var = g.below(4) == 0
        ref16[k] = 1 << ref2
        for s in range(n_smpl):
            nalt = g.below(3) if is_var else 0
            n = depth + g.below(depth)
            for j in range(n):
                bq = bqv[g.below(4)]
                base = alt2 if (nalt == 2 or (nalt == 1 and (g.r32() & 1))) else ref2
                if g.below(1000) < (80 if bq < 20 else 3):
                    base = (base + 1 + g.below(3)) & 3
                mapq = 60 if g.below(10) else g.below(60)
                qpos = g.below(READ_LEN)
                rev = g.r32() & 1
                w, e = sam.pack_read(1 << base, bq, mapq, rev, 0, 0, 0, qpos, READ_LEN, [(READ_LEN, "M")], True)
                rd.append(w)
                ep.append(e)
            off.append(len(rd))
    return host.HostTile(n_smpl, ref16, np.array(off, dtype=np.uint32), np.array(rd, dtype=np.uint32),
                         np.array(ep, dtype=np.uint8))


def expected_lines(tile, n_smpl, varonly):
    cfg = abi.default_cfg(n_smpl, max_sites=tile.n_sites, max_reads=len(tile.rd),
                          call_flag=abi.CALL_VARONLY if varonly else 0)
    m = orc.mpileup(cfg, tile)
    cin = host.CallInput(n_smpl, m.site["n_alleles"], np.maximum(m.site["unseen"], 0), m.pl.astype(np.int32), m.site["qsum"])
    c = orc.mcall(cfg, cin)
    nt = "ACGTN"
    out = []
    for k in range(tile.n_sites):
        cs, ms = c.site[k], m.site[k]
        if cs["ret"] <= 0:
            continue
        alts = [("*" if i == ms["unseen"] else nt[ms["a"][i]]) for i in range(1, ms["n_alleles"]) if cs["als_map"][i] > 0]
        acs = [str(int(cs["ac"][i])) for i in range(1, cs["nals_new"])]
        g0, g1 = int(c.gt[k, 0, 0]), int(c.gt[k, 1, 0])
        out.append((k + 1, nt[ms["a"][0]], ",".join(alts) or ".", None if cs["qual_missing"] else float(cs["qual"]),
                    int(cs["an"]), ",".join(acs) or ".", int(ms["depth"]), "./." if g0 < 0 else "%d/%d" % (g0, g1)))
    return out


def test_c_host_builds_and_refuses_to_run_without_a_gpu():
    build_host()
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the refusal path is covered on CPU-only machines")
    p = subprocess.run([EXE, "4", "2", "5", "1"], capture_output=True, text=True)
    assert p.returncode != 0 and p.stdout == ""
    assert "bcfgpu_create" in p.stderr               # BCFGPU_E_NODEV, reported by the CHECK macro: no CPU fallback


def _sam_reads(path):
    """(read group, flag) of the mapped reads of a SAM file, and its @RG ID -> SM map"""
    rg2sm, reads = {}, []
    for ln in open(path):
        f = ln.rstrip("\n").split("\t")
        if ln.startswith("@RG"):
            d = dict(x.split(":", 1) for x in f[1:])
            rg2sm.setdefault(d["ID"], d["SM"])
        elif not ln.startswith("@"):
            rg = [x[5:] for x in f[11:] if x.startswith("RG:Z:")]
            reads.append((rg[0] if rg else None, int(f[1])))
    return rg2sm, reads


@pytest.mark.parametrize("opts,files,goldf,rename", [
    ("", (1, 2, 3), "mpileup.1.out", {}),
    ("-s HG00101,HG00102", (1, 2, 3), "mpileup.7.out", {}),
    ("-S ^{G}/mplp.samples", (1, 2, 3), "mpileup.8.out", {}),
    ("-S {G}/mplp.9.samples", (1, 2, 3), "mpileup.9.out", {"HG00101": "SAMPLE1", "HG00102": "SAMPLE2"}),
    ("-G {G}/mplp.10.samples", (1, 2, 3), "mpileup.10.out", None),
    ("-s ^HG99999", (3, 4), "mpileup.11.out", {}),
    ("-G {G}/mplp.11.rgs", (3, 4), "mpileup.11.out", {})])
def test_c_sam_driver_sample_plumbing_on_the_host(golden_dir, opts, files, goldf, rename):
    """The host side of host/bcfgpu_sam.c needs no device: `--list-samples` prints the output samples bam_sample.c's rules give
    (names and order = the golden's #CHROM line) and how many reads of each pass mplp_func's filters into the pileup."""
    build_host()
    G = os.path.join(golden_dir, "mpileup")
    sams = [os.path.join(G, "mpileup.%d.sam" % i) for i in files]
    out = subprocess.run([SAM_EXE, "--list-samples"] + opts.format(G=G).split() + [os.path.join(G, "mpileup.ref.fa"), "17", "1", "4200"] + sams,
                         check=True, stdout=subprocess.PIPE, text=True).stdout
    got = [ln.split("\t") for ln in out.splitlines()]
    chrom = [ln for ln in open(os.path.join(G, goldf)) if ln.startswith("#CHROM")][0].rstrip("\n").split("\t")[9:]
    assert [g[0] for g in got] == chrom
    # read counts: mapped, not secondary / QC-fail / duplicate, no orphans; grouped as the option says
    want = {}
    rgs = None
    if rename is None:                                                     # -G: read group -> sample (a row without a name keeps the header's)
        rgs = {}
        for ln in open(os.path.join(G, "mplp.10.samples")):
            w = ln.split()
            rgs[w[0]] = w[1] if len(w) > 1 else None
    for path in sams:
        rg2sm, reads = _sam_reads(path)
        for rg, flag in reads:
            if flag & (4 | 256 | 512 | 1024) or (flag & 1 and not flag & 2):
                continue
            sm = rg2sm.get(rg)
            if rgs is not None:
                if rg not in rgs:
                    continue
                sm = rgs[rg] or sm
            else:
                sm = rename.get(sm, sm)
            if sm in chrom:
                want[sm] = want.get(sm, 0) + 1
    assert {g[0]: int(g[1]) for g in got} == want
    assert all(int(g[2]) == 1 for g in got)


def test_generator_twin_packs_like_the_library():
    """The Python twin of the packer agrees with bcfgpu_pack_read on the driver's reads (host code, no GPU needed)."""
    import ctypes as C
    from bcftools_amd import lib
    L = lib.load()
    g = Rng(3)
    for _ in range(200):
        base, bq, mapq, qpos, rev = g.below(4), [11, 25, 37, 40][g.below(4)], g.below(61), g.below(READ_LEN), g.r32() & 1
        cig = np.array([READ_LEN << 4], dtype=np.uint32)
        w, e = C.c_uint32(), C.c_uint8()
        L.bcfgpu_pack_read(1 << base, bq, mapq, rev, 0, 0, 0, qpos, READ_LEN, cig.ctypes.data_as(C.c_void_p), 1, 1,
                           C.byref(w), C.byref(e))
        assert (w.value, e.value) == sam.pack_read(1 << base, bq, mapq, rev, 0, 0, 0, qpos, READ_LEN, [(READ_LEN, "M")], True)


@pytest.mark.gpu
@pytest.mark.parametrize("n_sites,n_smpl,depth,seed,varonly", [(48, 6, 20, 7, False), (64, 33, 12, 11, True)])
def test_c_host_records_match_oracle(n_sites, n_smpl, depth, seed, varonly):
    build_host()
    p = subprocess.run([EXE, str(n_sites), str(n_smpl), str(depth), str(seed)] + (["-v"] if varonly else []),
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    got = [ln.split("\t") for ln in p.stdout.strip().split("\n") if ln]
    want = expected_lines(host_tile(n_sites, n_smpl, depth, seed), n_smpl, varonly)
    assert len(got) == len(want) and len(want) > 0
    for g, w in zip(got, want):
        assert (int(g[0]), g[1], g[2]) == w[:3]
        if w[3] is None:
            assert g[3] == "."
        else:
            assert float(g[3]) == pytest.approx(w[3], rel=2e-3, abs=1e-3)      # printed with 4 significant digits
        assert (int(g[4]), g[5], int(g[6]), g[7]) == w[4:]


SAM_EXE = os.path.join(ROOT, "host", "bcfgpu_sam")
VIEW_EXE = os.path.join(ROOT, "host", "bcfgpu_view")


def normalised(text):
    """what test.pl:1579-1585 compares: the whole file but for the ##bcftools* and ##reference lines"""
    return [ln for ln in text.splitlines() if not ln.startswith("##bcftools") and not ln.startswith("##reference")]


def whole_file_checks(cmd, golden_path):
    """The driver's VCF output equals the golden file, header and records; so does its BCF output (-Ob and -Ou) once
    bcfgpu_view has turned it back into text -- the two commands test.pl runs for every mpileup / call test."""
    want = normalised(open(golden_path).read())
    out = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, text=True).stdout
    assert normalised(out) == want
    for mode in ("b", "u"):
        bcf = subprocess.run(cmd[:1] + ["-O", mode] + cmd[1:], check=True, stdout=subprocess.PIPE).stdout
        assert bcf[:2] == b"\x1f\x8b" if mode == "b" else bcf[:5] == b"BCF\x02\x02"     # -Ob: BGZF; -Ou: the stream as it is (htslib "wbu")
        back = subprocess.run([VIEW_EXE, "-"], input=bcf, check=True, stdout=subprocess.PIPE).stdout.decode()
        assert normalised(back) == want
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("region,goldf,tags,n_snp,n_indel", [
    ("100-150", "mpileup.1.out", None, 51, 0), ("100-600", "mpileup.2.out", "DP,DV", 501, 1),
    ("100-600", "mpileup.4.out", "DP,DPR,DV,DP4,INFO/DPR,SP", 501, 1),
    ("100-600", "mpileup.5.out", "DP,AD,ADF,ADR,SP,INFO/AD,INFO/ADF,INFO/ADR", 501, 1)])
def test_c_sam_driver_reproduces_reference_goldens(golden_dir, region, goldf, tags, n_snp, n_indel):
    """host/bcfgpu_sam.c: SAM files in, every stage on the device (BAQ, mate overlaps, pileup, glfgen + combine), VCF-like
    a VCF or BCF file out -- identical, header and all records (SNP and indel), to the reference's
    test/mpileup/mpileup.{1,2,4,5}.out (test.pl:640-644; ##bcftools* / ##reference lines aside, as test.pl strips them)."""
    from tests.helpers import vcf
    build_host()
    G = os.path.join(golden_dir, "mpileup")
    beg, end = region.split("-")
    out = whole_file_checks([SAM_EXE] + (["-a", tags] if tags else []) + [os.path.join(G, "mpileup.ref.fa"), "17", beg, end] +
                            [os.path.join(G, "mpileup.%d.sam" % i) for i in (1, 2, 3)], os.path.join(G, goldf))
    recs = [vcf.Rec(ln) for ln in out.splitlines() if not ln.startswith("#")]
    allg = vcf.Vcf(os.path.join(G, goldf)).recs
    key = lambda r: (r.pos, "INDEL" in r.info)
    got, gold = {key(r): r for r in recs}, {key(r): r for r in allg}
    assert [key(r) for r in recs] == [key(r) for r in allg]          # same records in the same order (SNP, then indel)
    assert sum(1 for k in gold if not k[1]) == n_snp and sum(1 for k in gold if k[1]) == n_indel
    for k, g in gold.items():
        c, p = got[k], k
        assert c.alleles == g.alleles, (p, c.alleles, g.alleles)
        assert c.info["DP"] == g.info["DP"], p
        assert c.info_floats("I16") == g.info_floats("I16"), (p, c.info["I16"], g.info["I16"])
        for a, b in zip(c.info_floats("QS"), g.info_floats("QS")):
            assert abs(a - b) <= 2e-5 * max(abs(a), abs(b)) + 1e-9, (p, c.info["QS"], g.info["QS"])
        for tag in ("VDB", "SGB", "RPB", "MQB", "MQSB", "BQB", "MQ0F", "IDV", "IMF"):
            assert (tag in c.info) == (tag in g.info), (p, tag)
            if tag in g.info:
                a, b = float(c.info[tag]), float(g.info[tag])
                assert abs(a - b) <= 2e-5 * max(abs(a), abs(b)) + 1e-9, (p, tag, a, b)
        for s in range(3):
            assert c.fmt("PL", s) == g.fmt("PL", s), (p, s, c.fmt("PL", s), g.fmt("PL", s))


@pytest.mark.gpu
@pytest.mark.parametrize("goldf,opts,region,files,n", [
    ("mpileup.3.out", "-B --ff 0x14", "1050-1060", (1,), None),                        # no BAQ, reverse-strand reads filtered
    ("mpileup.7.out", "-s HG00101,HG00102", "100-150", (1, 2, 3), 51),                  # samples by name: the first file drops out
    ("mpileup.7.out", "-S {G}/mplp.samples", "100-150", (1, 2, 3), 51),
    ("mpileup.8.out", "-s ^HG00101,HG00102", "100-150", (1, 2, 3), None),               # ... excluded
    ("mpileup.8.out", "-S ^{G}/mplp.samples", "100-150", (1, 2, 3), None),
    ("mpileup.9.out", "-S {G}/mplp.9.samples", "100-150", (1, 2, 3), None),             # renamed
    ("mpileup.10.out", "-G {G}/mplp.10.samples", "100-150", (1, 2, 3), None),           # read groups -> samples: one file, three samples
    ("mpileup.11.out", "", "1-4200", (3,), 4002),                                       # a whole contig, indel records included
    ("mpileup.11.out", "-s HG00102", "1-4200", (3, 4), 4002),                           # the second file has no wanted sample
    ("mpileup.11.out", "-s ^HG99999", "1-4200", (3, 4), 4002),
    ("mpileup.11.out", "-G {G}/mplp.11.rgs", "1-4200", (3, 4), 4002)])
def test_c_sam_driver_sample_plumbing(golden_dir, goldf, opts, region, files, n):
    """bam_sample.c in host/bcfgpu_sam.c: @RG -> sample, -s/-S (with ^ and renaming), -G read-group lists, files without a
    usable read group dropped, and the read filters -B / --ff: the whole of test/mpileup/mpileup.{3,7,8,9,10,11}.out
    (test.pl:641,647-657), as VCF and through BCF."""
    build_host()
    G = os.path.join(golden_dir, "mpileup")
    beg, end = region.split("-")
    out = whole_file_checks([SAM_EXE] + opts.format(G=G).split() + [os.path.join(G, "mpileup.ref.fa"), "17", beg, end] +
                            [os.path.join(G, "mpileup.%d.sam" % i) for i in files], os.path.join(G, goldf))
    nrec = sum(1 for ln in out.splitlines() if not ln.startswith("#"))
    assert nrec > 0 and (n is None or nrec == n)


@pytest.mark.gpu
@pytest.mark.parametrize("gpus", [2, 3, 5])
def test_c_sam_driver_region_shards_match_the_whole_contig(golden_dir, gpus):
    """`bcfgpu_sam --gpus N`: the region cut into N contiguous shards, a process per shard (shard k on device k mod the devices
    present: on a one-GPU box they share it), each reading the reads that overlap its shard; the records are emitted in shard
    order = genomic order.  The whole of test/mpileup/mpileup.11.out (a contig with SNP and indel records), byte for byte as
    the single-process run gives it, as VCF and through BCF."""
    build_host()
    G = os.path.join(golden_dir, "mpileup")
    out = whole_file_checks([SAM_EXE, "--gpus", str(gpus), "-s", "^HG99999", os.path.join(G, "mpileup.ref.fa"), "17", "1", "4200"] +
                            [os.path.join(G, "mpileup.%d.sam" % i) for i in (3, 4)], os.path.join(G, "mpileup.11.out"))
    assert sum(1 for ln in out.splitlines() if not ln.startswith("#")) == 4002


TILE_CASES = {
    # golden: (options, reference, regions, input files): mpileup's own spelling, -f REF [-r REGIONS] files (mpileup.c:952-1003)
    "mpileup.11.out": (["-s", "^HG99999"], "mpileup.ref.fa", "17:1-4200", ["mpileup.3.sam", "mpileup.4.sam"]),   # SNP and indel records over a contig
    "mpileup.2.out": (["-a", "DP,DV"], "mpileup.ref.fa", "17:100-600", ["mpileup.1.sam", "mpileup.2.sam", "mpileup.3.sam"]),
    "mpileup.6.out": (["-a", "DP,DV", "--gvcf", "0,2,5"], "mpileup.ref.fa", "17:100-600", ["mpileup.1.sam", "mpileup.2.sam", "mpileup.3.sam"]),
    "indel-AD.1.out": (["-a", "AD"], "indel-AD.1.fa", None, ["indel-AD.1.sam"]),                                 # no -r: every sequence of the file (test.pl:658)
}


def _tile_cmd(G, goldf, extra, regions="same"):
    opts, ref, reg, files = TILE_CASES[goldf]
    if regions != "same":
        reg = regions
    return [SAM_EXE] + extra + opts + ["-f", os.path.join(G, ref)] + (["-r", reg] if reg else []) + [os.path.join(G, f) for f in files]


@pytest.mark.gpu
@pytest.mark.parametrize("tile", [64, 512])
@pytest.mark.parametrize("goldf", sorted(TILE_CASES))
def test_c_sam_driver_streams_a_region_in_tiles(golden_dir, goldf, tile):
    """`bcfgpu_sam --tile N`: the region cut into tiles of N columns, the files read in step with the tiles, the depth cap's
    buffer and the open gVCF block carried from tile to tile, indel candidates next to a tile's edge realigned from the reads
    that cover them: the reference's goldens whole-file, as VCF and through BCF, whatever the tile size (mpileup_reg() walks a
    region column by column, mpileup.c:327-367; gvcf_write keeps a block over any distance, gvcf.c:88-226)."""
    build_host()
    G = os.path.join(golden_dir, "mpileup")
    whole_file_checks(_tile_cmd(G, goldf, ["--tile", str(tile)]), os.path.join(G, goldf))


@pytest.mark.gpu
@pytest.mark.parametrize("goldf,regions", [("mpileup.2.out", "17:100-300,17:301-302,17:303-600"), ("mpileup.6.out", "17:100-257,17:258-600"),
                                           ("mpileup.11.out", "17:1-1000,17:1001-4200")])
def test_c_sam_driver_runs_several_regions(golden_dir, goldf, regions):
    """`-r REG,REG,...` (mpileup.c:652-683): the regions one after the other, each with its own pass over the files; adjacent
    regions give the records of the whole stretch, a gVCF block going on across the seam as the reference's gvcf_write would
    carry it (the same sequence, the next position, the same depth range)."""
    build_host()
    G = os.path.join(golden_dir, "mpileup")
    whole_file_checks(_tile_cmd(G, goldf, ["--tile", "128"], regions), os.path.join(G, goldf))


@pytest.mark.gpu
@pytest.mark.parametrize("gpus", [2, 3])
def test_c_sam_driver_region_shards_join_gvcf_blocks(golden_dir, gpus):
    """`--gpus N --gvcf`: the block a shard ends with and the block the next one starts with are joined by the process that
    writes the shards out, by gvcf_write's rule: test/mpileup/mpileup.6.out whole-file from 2 and 3 shards."""
    build_host()
    G = os.path.join(golden_dir, "mpileup")
    whole_file_checks(_tile_cmd(G, "mpileup.6.out", ["--gpus", str(gpus)]), os.path.join(G, "mpileup.6.out"))


@pytest.mark.gpu
def test_c_sam_driver_reads_bam_and_counts_soft_clips(golden_dir):
    """BAM input (BGZF + BAM records parsed in C) and -a INFO/SCR,FMT/SCR: the whole of test/mpileup/mpileup-SCR.out (test.pl:659)."""
    build_host()
    G = os.path.join(golden_dir, "mpileup")
    out = whole_file_checks([SAM_EXE, "-a", "INFO/SCR,FMT/SCR", os.path.join(G, "mpileup-SCR.fa"), "1", "1", "150",
                             os.path.join(G, "mpileup-SCR.bam")], os.path.join(G, "mpileup-SCR.out"))
    assert sum(1 for ln in out.splitlines() if not ln.startswith("#")) == 86


CALL_EXE = os.path.join(ROOT, "host", "bcfgpu_call")


@pytest.mark.gpu
@pytest.mark.parametrize("vcff,goldf,args,n", [
    ("mpileup.vcf", "mpileup.1.out", "-v", 11), ("mpileup.vcf", "mpileup.3.out", "-v -S {G}/mpileup.3.samples", None),
    ("mpileup.vcf", "mpileup.3.out", "-v -s HG00100,HG00101,HG00102 -p 0.5 --threads 2", None),   # the same samples as a list (vcfcall.c:1050)
    ("mpileup.vcf", "mpileup.3.out", "--multiallelic-caller --variants-only --samples-file {G}/mpileup.3.samples", None),   # long option names
    ("mpileup.vcf", "mpileup.4.out", "-v -S {G}/mpileup.4.samples", None), ("mpileup.vcf", "mpileup.5.out", "-v -S {G}/mpileup.5.samples", None),
    ("mpileup.X.vcf", "mpileup.X.out", "-v -S {G}/mpileup.samples --ploidy-file {G}/mpileup.ploidy", None),      # sexes + ploidy file: haploid males on X
    ("mpileup.X.vcf", "mpileup.X.out", "-v -S {G}/mpileup.ped --ploidy-file {G}/mpileup.ploidy", None),          # the same from a PED file
    ("mpileup.X.vcf", "mpileup.X.2.out", "-v -S {G}/mpileup.2.samples --ploidy-file {G}/mpileup.ploidy", None),  # ploidy numbers in the sample list
    ("mpileup.NA19213.NA19129.vcf", "mpileup.hwe.1.out", "-v", None), ("mpileup.hwe.vcf", "mpileup.hwe.2.out", "-v", None),
    ("mpileup.NA19213.NA19129.vcf", "mpileup.hwe.1b.out", "-v -G - --group-samples-tag AD", None),               # every sample its own group
    ("mpileup.hwe.vcf", "mpileup.hwe.3.out", "-v -G - --group-samples-tag AD", None),
    ("mpileup.hwe.vcf", "mpileup.hwe.4.out", "-v -G {G}/mpileup.hwe.samples --group-samples-tag AD", None),      # groups from a file
    ("call-G.vcf", "call-G.1.out", "-v", None), ("call-G.vcf", "call-G.2.out", "-v -G - --group-samples-tag AD", None),
    ("call-G.2.vcf", "call-G.2.1.out", "-v -F AN_POP,AC_POP", None),                                             # prior from INFO tags
    ("call.af-fixation.vcf", "call.af-fixation.1.out", "", None),                                                # all records, not only variants
    ("call.af-fixation.vcf", "call.af-fixation.2.out", "-G {G}/call.af-fixation.txt", None),
    ("call.af-fixation.vcf", "call.af-fixation.3.out", "-G {G}/call.af-fixation.txt -a GP,GQ", None),
    ("mpileup.vcf", "mpileup.2.out", "-mg0", 23),                                                               # call-side gVCF blocks (test.pl:277)
])
def test_c_call_driver_reproduces_reference_golden(golden_dir, vcff, goldf, args, n):
    """host/bcfgpu_call.c: `call -m [-v] [-S samples] [--ploidy-file f] [-G groups] [-F AN,AC] [-a GP,GQ]` on the reference's
    test VCFs with mcall() on the device -- its VCF output, and its BCF output read back, equal every `call -m` golden of
    test.pl:276-308, header included (test.pl:1194-1195)."""
    build_host()
    G = os.path.join(golden_dir, "call")
    cmd = [CALL_EXE] + args.format(G=G).split() + [os.path.join(G, vcff)]
    out = whole_file_checks(cmd, os.path.join(G, goldf))
    nrec = sum(1 for ln in out.splitlines() if not ln.startswith("#"))
    assert nrec > 0 and (n is None or nrec == n)


@pytest.mark.gpu
def test_c_sam_driver_reproduces_gvcf_golden(golden_dir):
    """`bcfgpu_sam -a DP,DV --gvcf 0,2,5`: the reference-only records collapse into gVCF blocks on the device
    (bcfgpu_gvcf_blocks) -- the whole of test/mpileup/mpileup.6.out (test.pl:645), as VCF and through BCF."""
    build_host()
    G = os.path.join(golden_dir, "mpileup")
    out = whole_file_checks([SAM_EXE, "-a", "DP,DV", "--gvcf", "0,2,5", os.path.join(G, "mpileup.ref.fa"), "17", "100", "600"] +
                            [os.path.join(G, "mpileup.%d.sam" % i) for i in (1, 2, 3)], os.path.join(G, "mpileup.6.out"))
    assert sum(1 for ln in out.splitlines() if not ln.startswith("#")) == 42


@pytest.mark.gpu
def test_c_drivers_pipe_bcf(golden_dir, tmp_path):
    """`bcfgpu_sam -Ou ... | bcfgpu_call -v -`: the uncompressed-BCF pipe of `bcftools mpileup -Ou | bcftools call -mv` between the
    two drivers gives the records of the same run through a VCF file."""
    build_host()
    G = os.path.join(golden_dir, "mpileup")
    sam_cmd = [SAM_EXE, os.path.join(G, "mpileup.ref.fa"), "17", "100", "600"] + [os.path.join(G, "mpileup.%d.sam" % i) for i in (1, 2, 3)]
    vcf_path = str(tmp_path / "m.vcf")
    with open(vcf_path, "w") as f:
        subprocess.run(sam_cmd, check=True, stdout=f)
    via_file = subprocess.run([CALL_EXE, "-v", vcf_path], check=True, stdout=subprocess.PIPE, text=True).stdout
    p1 = subprocess.Popen(sam_cmd[:1] + ["-O", "u"] + sam_cmd[1:], stdout=subprocess.PIPE)
    via_pipe = subprocess.run([CALL_EXE, "-v", "-"], stdin=p1.stdout, check=True, stdout=subprocess.PIPE, text=True).stdout
    assert p1.wait() == 0
    assert normalised(via_pipe) == normalised(via_file)
    assert sum(1 for ln in via_pipe.splitlines() if not ln.startswith("#")) >= 1


MGPU_EXE = os.path.join(ROOT, "host", "bcfgpu_mgpu")


@pytest.mark.gpu
@pytest.mark.parametrize("n_sites,n_smpl,depth,seed,varonly", [(90, 6, 15, 5, True), (41, 20, 10, 9, False)])
def test_c_multi_gpu_driver_matches_single_gpu(n_sites, n_smpl, depth, seed, varonly):
    """host/bcfgpu_mgpu.c: contiguous region shards, one thread + context per rank, device-side compaction of the records
    to write, ordered gather to rank 0.  Whatever the number of ranks, the records are those of the single-context driver
    (bcfgpu_host).  On a one-GPU box the ranks share the device and gather through host memory (--share-devices: RCCL
    wants a device per rank); with two or more GPUs visible the RCCL send/recv gather runs as well."""
    import torch
    build_host()
    args = [str(n_sites), str(n_smpl), str(depth), str(seed)] + (["-v"] if varonly else [])
    want = subprocess.run([EXE] + args, check=True, stdout=subprocess.PIPE, text=True).stdout
    assert want.count("\n") > 3
    for n in (1, 2, 3, 5):
        got = subprocess.run([MGPU_EXE] + args + ["--gpus", str(n), "--share-devices"], check=True, stdout=subprocess.PIPE, text=True).stdout
        assert got == want, n
    ndev = torch.cuda.device_count()
    for n in range(2, min(ndev, 4) + 1):
        got = subprocess.run([MGPU_EXE] + args + ["--gpus", str(n), "--gather", "rccl"], check=True, stdout=subprocess.PIPE, text=True).stdout
        assert got == want, ("rccl", n)
    # one rank through the library's gather entry (no peer: the local copy only)
    got = subprocess.run([MGPU_EXE] + args + ["--gpus", "1", "--gather", "rccl"], check=True, stdout=subprocess.PIPE, text=True).stdout
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("vcff,goldf,tab,ins", [
    ("mpileup.vcf", "mpileup.cAls.out", "mpileup.tab", False), ("mpileup.2.vcf", "mpileup.cAls.2.out", "mpileup.2.tab", False),
    ("mpileup.3.vcf", "mpileup.cAls.3.out", "mpileup.3.tab", True), ("mpileup.3.vcf", "mpileup.cAls.4.out", "mpileup.4.tab", True),
    ("mpileup.3.vcf", "mpileup.cAls.5.out", "mpileup.5.tab", True), ("mpileup.4.vcf", "mpileup.cAls.6.out", "mpileup.6.tab", True),
    ("mpileup.5.vcf", "mpileup.cAls.7.out", "mpileup.7.tab", True),
    ("mpileup.cals.1.vcf", "mpileup.cals.8.out", "mpileup.cals.1.tab", False),      # an indel target paired with the SNP record
    ("mpileup.cals.2.vcf", "mpileup.cals.9.out", "mpileup.cals.2.tab", False),      # SNP and indel records at one position
])
def test_c_call_driver_constrained_alleles(golden_dir, vcff, goldf, tab, ins):
    """`bcfgpu_call -mA -C alleles -T targets [-i]` (test.pl:289-297, test_vcf_call_cAls): the records re-expressed in the
    target alleles on the host (mcall_constrain_alleles, mcall.c:1271-1421; next_line, vcfcall.c:501-605; vcmp.c), called on
    the device with -A, the lines of unmet targets inserted with -i -- the whole golden file, as VCF and through BCF."""
    build_host()
    G = os.path.join(golden_dir, "call")
    cmd = [CALL_EXE, "-m", "-A", "-C", "alleles", "-T", os.path.join(G, tab)] + (["-i"] if ins else []) + [os.path.join(G, vcff)]
    out = whole_file_checks(cmd, os.path.join(G, goldf))
    assert sum(1 for ln in out.splitlines() if not ln.startswith("#")) > 0


def _deep_sam(path, ref, sample, seed, n_reads, lo, hi, rlen=100, indels=True, err=0.01, snps=None):
    """n_reads reads of one sample over [lo, hi) of contig 17: the reference's bases with a few mismatches, some reads with a
    2-base insertion or a 3-base deletion at a common place, so that both passes have cells of several hundred usable reads.
    snps: {0-based position: fraction of the reads that carry another base there} -- at 10-30 % the PLs of a cell of 255 reads stay
    below their cap of 255 and so depend on which reads errmod_cal draws."""
    rng = np.random.default_rng(seed)
    with open(path, "w") as f:
        f.write("@HD\tVN:1.0\tSO:coordinate\n@SQ\tSN:17\tLN:%d\n@RG\tID:%s\tSM:%s\n" % (len(ref), sample, sample))
        for i, pos in enumerate(sorted(int(x) for x in rng.integers(lo, hi, n_reads))):
            seq = list(ref[pos:pos + rlen + 3])
            for k in np.flatnonzero(rng.random(len(seq)) < err):
                seq[k] = "ACGT"[int(rng.integers(0, 4))]
            for q_, fr in (snps or {}).items():
                if pos <= q_ < pos + rlen and rng.random() < fr:
                    seq[q_ - pos] = "ACGT"[("ACGT".index(ref[q_].upper()) + 1) % 4] if ref[q_].upper() in "ACGT" else "A"
            cut = (lo + hi) // 2 + 20 - pos                       # the indels sit at one reference position
            kind = rng.choice(3, p=[0.7, 0.15, 0.15]) if indels and 10 < cut < rlen - 10 else 0
            if kind == 1:
                seq, cig = seq[:cut] + ["G", "T"] + seq[cut:rlen - 2], "%dM2I%dM" % (cut, rlen - 2 - cut)
            elif kind == 2:
                seq, cig = seq[:cut] + seq[cut + 3:rlen + 3], "%dM3D%dM" % (cut, rlen - cut)
            else:
                seq, cig = seq[:rlen], "%dM" % rlen
            qual = "".join(chr(33 + int(q)) for q in rng.integers(15, 41, rlen))
            f.write("r%d\t%d\t17\t%d\t%d\t%s\t*\t0\t0\t%s\t%s\tRG:Z:%s\n" % (i, 16 * int(rng.integers(0, 2)), pos + 1, int(rng.choice([20, 40, 60])),
                                                                         cig, "".join(seq), qual, sample))


@pytest.mark.gpu
def test_c_sam_driver_draw_does_not_depend_on_the_tiles(golden_dir, tmp_path):
    """Cells of more than 255 usable reads (-d 10000, and -L 10000 so that indels are called at that depth): bcfgpu_sam plans errmod_cal's draw for every tile (bcfgpu_errmod_plan) and
    the generator goes on from tile to tile as it goes on from position to position in one mpileup process, so the records
    cannot depend on where the tiles are cut; none of the cells is left to the first-255 rule."""
    build_host()
    G = os.path.join(golden_dir, "mpileup")
    ref = "".join(ln.strip() for ln in open(os.path.join(G, "mpileup.ref.fa")) if not ln.startswith(">"))
    files = []
    for s, (name, n) in enumerate((("deepA", 900), ("deepB", 500))):
        files.append(str(tmp_path / (name + ".sam")))
        _deep_sam(files[-1], ref, name, 40 + s, n, 1000, 1150)
    outs = []
    for tile in (None, 16, 50):
        cmd = [SAM_EXE, "-d", "10000", "-L", "10000", "-a", "AD,DP"] + (["--tile", str(tile)] if tile else []) + ["-f", os.path.join(G, "mpileup.ref.fa"), "-r", "17:990-1260"] + files
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=300)
        assert p.returncode == 0, p.stderr
        assert "first 255" not in p.stderr, p.stderr
        outs.append([ln for ln in p.stdout.splitlines() if not ln.startswith("##")])
    recs = [ln.split("\t") for ln in outs[0][1:]]
    deep = [r for r in recs if max(int(x.split(":")[r[8].split(":").index("DP")]) for x in r[9:]) > 255]
    assert len(deep) > 100 and any("INDEL" in r[7] for r in deep)          # FORMAT/DP past 255 in both kinds of record
    assert outs[1] == outs[0] and outs[2] == outs[0]


@pytest.mark.gpu
def test_c_sam_driver_deep_cells_match_the_oracle_in_visit_order(golden_dir, tmp_path):
    """`bcfgpu_sam -B -I -d 10000 --tile 32` on cells of 300-900 usable reads against the oracle (oracle/errmod.c's restatement of
    errmod_cal's draw, rule 0) run the way one mpileup process runs: position by position, sample by sample, one generator.  PL,
    DP and AD of every record; the draw crosses nine tile seams on the way."""
    from tests.helpers import vcf
    build_host()
    G = os.path.join(golden_dir, "mpileup")
    ref = sam.read_fasta(os.path.join(G, "mpileup.ref.fa"))
    refs = "".join(ln.strip() for ln in open(os.path.join(G, "mpileup.ref.fa")) if not ln.startswith(">"))
    files = []
    for k, (name, n) in enumerate((("deepA", 700), ("deepB", 400))):
        files.append(str(tmp_path / (name + ".sam")))
        _deep_sam(files[-1], refs, name, 50 + k, n, 1000, 1120, indels=False, snps={q_: (0.1, 0.15, 0.2, 0.3)[(q_ // 5) % 4] for q_ in range(1003, 1215, 5)})
    out = str(tmp_path / "deep.vcf")
    cmd = [SAM_EXE, "-B", "-I", "-d", "10000", "--tile", "32", "-a", "AD,DP", "-o", out, "-f", os.path.join(G, "mpileup.ref.fa"), "-r", "17:1001-1300"] + files
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert "first 255" not in p.stderr
    got = vcf.Vcf(out)
    t = sam.build_tile([sam.Sam(f) for f in files], ref, "17", 1000, 1299, sam.MplpOpts())
    tile = host.HostTile(2, t["ref16"], t["plp_off"], t["rd"], t["epos"])
    cfg = abi.default_cfg(2, max_sites=tile.n_sites, max_reads=len(tile.rd), fmt_flag=abi.FMT_AD | abi.FMT_DP)
    assert [q + 1 for q in t["positions"]] == [r.pos for r in got.recs]
    deep = open_pl = 0
    for i, r in enumerate(got.recs):
        res = orc.mpileup(cfg, tile.select_sites([i]), deep_rule=0, reset=(i == 0))
        open_pl += int(np.sum((res.pl_of(0) > 0) & (res.pl_of(0) < 255)))
        na = int(res.site[0]["n_alleles"])
        for s_ in range(2):
            assert [int(x) for x in r.fmt("PL", s_).split(",")] == res.pl_of(0)[s_].tolist(), (r.pos, s_)
            assert [int(x) for x in r.fmt("AD", s_).split(",")] == [int(res.adf[0][a][s_]) + int(res.adr[0][a][s_]) for a in range(na)], (r.pos, s_)
            deep += int(r.fmt("DP", s_)) > 255
    assert deep > 150 and open_pl > 100


@pytest.mark.gpu
def test_c_sam_driver_illumina13_qualities(golden_dir, tmp_path):
    """`-6` / `--illumina1.3+` (mpileup.c:216-221: every quality q becomes q > 31 ? q - 31 : 0 as the read comes off the file): the
    input of mpileup.1.out with its qualities re-encoded 31 higher gives, with -6, the golden of the original."""
    build_host()
    G = os.path.join(golden_dir, "mpileup")
    shifted = []
    for i in (1, 2, 3):
        shifted.append(str(tmp_path / ("illumina13.%d.sam" % i)))
        with open(shifted[-1], "w") as out:
            for ln in open(os.path.join(G, "mpileup.%d.sam" % i)):
                f = ln.rstrip("\n").split("\t")
                if not ln.startswith("@") and len(f) > 10 and f[10] != "*":
                    assert max(ord(c) for c in f[10]) + 31 < 127
                    f[10] = "".join(chr(ord(c) + 31) for c in f[10])
                out.write("\t".join(f) + "\n")
    whole_file_checks([SAM_EXE, "-6", os.path.join(G, "mpileup.ref.fa"), "17", "100", "150"] + shifted, os.path.join(G, "mpileup.1.out"))


@pytest.mark.gpu
def test_c_sam_driver_file_list_and_regions_file(golden_dir, tmp_path):
    """`-b FILE` (the inputs listed in a file, mpileup.c:733-790, 1072) and `-R FILE` (the regions in a file, mpileup.c:1031): the
    whole of mpileup.2.out from a list of its three inputs and a one-line regions file -- also through two region shards."""
    build_host()
    G = os.path.join(golden_dir, "mpileup")
    lst, regs = str(tmp_path / "inputs.txt"), str(tmp_path / "regions.txt")
    with open(lst, "w") as f:
        f.write("\n".join(os.path.join(G, "mpileup.%d.sam" % i) + "  " for i in (1, 2, 3)) + "\n\n")
    with open(regs, "w") as f:
        f.write("# CHROM POS END\n17\t100\t600\n")
    base = [SAM_EXE, "-a", "DP,DV", "-f", os.path.join(G, "mpileup.ref.fa"), "-R", regs, "-b", lst]
    whole_file_checks(base, os.path.join(G, "mpileup.2.out"))
    whole_file_checks(base[:1] + ["--gpus", "2"] + base[1:], os.path.join(G, "mpileup.2.out"))


@pytest.mark.gpu
def test_c_sam_driver_ignore_overlaps(golden_dir):
    """`-x` / `--ignore-overlaps` (mpileup.c:1005): the mates' overlapping bases keep their qualities -- the same columns as
    mpileup.2.out, other likelihoods where mates overlap."""
    build_host()
    G = os.path.join(golden_dir, "mpileup")
    cmd = [SAM_EXE, "-a", "DP,DV", "-f", os.path.join(G, "mpileup.ref.fa"), "-r", "17:100-600"] + [os.path.join(G, "mpileup.%d.sam" % i) for i in (1, 2, 3)]
    a = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, check=True)
    b = subprocess.run(cmd[:1] + ["-x"] + cmd[1:], stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, check=True)
    ra = [ln.split("\t") for ln in a.stdout.splitlines() if not ln.startswith("#")]
    rb = [ln.split("\t") for ln in b.stdout.splitlines() if not ln.startswith("#")]
    assert [r[:2] for r in ra] == [r[:2] for r in rb] and len(ra) > 400
    assert "0 overlapping pairs" in b.stderr and "0 overlapping pairs" not in a.stderr
    assert sum(x[7:] != y[7:] for x, y in zip(ra, rb)) > 10


@pytest.mark.gpu
def test_c_call_driver_skips_masked_refs_and_variant_kinds(golden_dir, tmp_path):
    """The sites `call` passes over before calling (vcfcall.c:1095-1099): a reference allele that starts with N is skipped unless
    -M is given (CF_ACGT_ONLY is the default, vcfcall.c:937), -V snps / -V indels by htslib's bcf_is_snp.  Input: the
    reference's test/call input mpileup.c.vcf with every fifth record's REF masked."""
    from tests.helpers import vcf
    build_host()
    G = os.path.join(golden_dir, "call")
    src, masked = os.path.join(G, "mpileup.c.vcf"), str(tmp_path / "masked.vcf")
    n_rec = n_masked = 0
    with open(masked, "w") as out:
        for ln in open(src):
            if not ln.startswith("#"):
                f = ln.split("\t")
                if n_rec % 5 == 0 and len(f[3]) == 1:
                    f[3] = "N"; n_masked += 1
                ln = "\t".join(f)
                n_rec += 1
            out.write(ln)

    def run(args, path):
        p = subprocess.run([CALL_EXE] + args + [path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
        assert p.returncode == 0, p.stderr
        return [vcf.Rec(l) for l in p.stdout.splitlines() if not l.startswith("#")]
    base = run([], src)
    kept = run(["-M"], masked)                                   # every record comes out, the masked ones with REF N
    dflt = run([], masked)
    assert len(kept) == len(base) and sum(r.ref == "N" for r in kept) == n_masked > 5
    assert [(r.pos, r.ref) for r in dflt] == [(r.pos, r.ref) for r in kept if r.ref != "N"]
    snps = run(["-V", "indels"], src)
    indels = run(["-V", "snps"], src)
    assert [(r.pos, r.alleles) for r in snps] == [(r.pos, r.alleles) for r in base if vcf.is_snp(r)]
    assert [(r.pos, r.alleles) for r in indels] == [(r.pos, r.alleles) for r in base if not vcf.is_snp(r)]
    assert len(indels) >= 1 and len(snps) + len(indels) == len(base)


def test_c_sam_driver_reads_mpileups_long_option_names(golden_dir):
    """mpileup's long option names (mpileup.c:952-1003) give what their short forms give: --list-samples needs no device."""
    build_host()
    G = os.path.join(golden_dir, "mpileup")
    files = [os.path.join(G, "mpileup.%d.sam" % i) for i in (1, 2, 3, 4)]
    short = [SAM_EXE, "--list-samples", "-s", "^HG99999", "-q", "10", "-d", "100", "-A", "--ff", "0x704", "-f", os.path.join(G, "mpileup.ref.fa"), "-r", "17:1-4200"] + files
    long_ = [SAM_EXE, "--list-samples", "--samples", "^HG99999", "--min-MQ", "10", "--max-depth", "100", "--count-orphans", "--excl-flags", "0x704",
             "--fasta-ref", os.path.join(G, "mpileup.ref.fa"), "--regions", "17:1-4200"] + files
    a = subprocess.run(short, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    b = subprocess.run(long_, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    assert a.returncode == 0 and b.returncode == 0, (a.stderr, b.stderr)
    assert a.stdout == b.stdout and len(a.stdout.splitlines()) == 3


@pytest.mark.gpu
def test_c_call_driver_predefined_ploidies(golden_dir, tmp_path):
    """`call --ploidy ALIAS` (vcfcall.c:138-199, 827-855): the definitions call carries with it give what the same lines give as a
    --ploidy-file -- GRCh37 (males haploid on X outside the pseudo-autosomal regions), X, 1 -- on the reference's mpileup.X.vcf."""
    build_host()
    G = os.path.join(golden_dir, "call")
    src, smp = os.path.join(G, "mpileup.X.vcf"), os.path.join(G, "mpileup.samples")

    def run(args):
        p = subprocess.run([CALL_EXE] + args + [src], stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
        assert p.returncode == 0, p.stderr
        return [l for l in p.stdout.splitlines() if not l.startswith("##")]
    cases = {"GRCh37": "X 1 60000 M 1\nX 2699521 154931043 M 1\nY 1 59373566 M 1\nY 1 59373566 F 0\n* * * M 2\n* * * F 2\n",
             "X": "* * * M 1\n* * * F 2\n", "Y": "* * * M 1\n* * * F 0\n", "1": "* * * * 1\n"}
    outs = {}
    for alias, text in cases.items():
        f = str(tmp_path / ("ploidy." + alias))
        open(f, "w").write(text)
        outs[alias] = run(["-v", "-S", smp, "--ploidy", alias])
        assert outs[alias] == run(["-v", "-S", smp, "--ploidy-file", f]), alias
        assert len(outs[alias]) > 3
    assert outs["X"] != outs["1"] and outs["GRCh37"] == outs["X"]      # (every site of the input lies in X's first haploid stretch)
    assert run(["-v", "-S", smp, "-X"]) == outs["X"]


@pytest.mark.gpu
def test_c_call_driver_targets_and_regions(golden_dir, tmp_path):
    """`call -t / -r REGIONS`, `-T / -R FILE` (without -C alleles: the sites to look at, vcfcall.c:612-626): the records of the full run
    whose position lies in the targets, nothing else."""
    build_host()
    G = os.path.join(golden_dir, "call")
    src = os.path.join(G, "mpileup.vcf")

    def run(args):
        p = subprocess.run([CALL_EXE] + args + [src], stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
        assert p.returncode == 0, p.stderr
        return [l for l in p.stdout.splitlines() if not l.startswith("#")]
    full = run([])
    pos = [int(l.split("\t")[1]) for l in full]
    lo, hi, one = pos[len(pos) // 4], pos[len(pos) // 2], pos[-3]
    want = [l for l, q in zip(full, pos) if lo <= q <= hi or q == one]
    assert 3 < len(want) < len(full)
    chrom = full[0].split("\t")[0]
    spec = "%s:%d-%d,%s:%d" % (chrom, lo, hi, chrom, one)
    assert run(["-t", spec]) == want and run(["-r", spec]) == want
    f = str(tmp_path / "targets.tab")
    open(f, "w").write("# CHROM POS END\n%s\t%d\t%d\n%s\t%d\n" % (chrom, lo, hi, chrom, one))
    assert run(["-T", f]) == want and run(["-R", f]) == want
    assert run(["-t", chrom]) == full and run(["-t", "no_such_sequence"]) == []


@pytest.mark.gpu
def test_c_sam_driver_targets(golden_dir, tmp_path):
    """`-t REG,...` / `-T FILE` (mpileup.c:198-212, 330-335): the columns inside the targets, from the reads that overlap them.  With
    -B and -I every column depends on its own reads only, so the records are those of the untargeted run at the targets' positions
    (the depth cap aside: -d 100000); `^` gives the columns outside the list from the same reads."""
    build_host()
    G = os.path.join(golden_dir, "mpileup")
    cmd = [SAM_EXE, "-B", "-I", "-x", "-d", "100000", "-f", os.path.join(G, "mpileup.ref.fa"), "-r", "17:100-600"] + [os.path.join(G, "mpileup.%d.sam" % i) for i in (1, 2, 3)]

    def run(extra):
        p = subprocess.run(cmd[:1] + extra + cmd[1:], stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
        assert p.returncode == 0, p.stderr
        return [l for l in p.stdout.splitlines() if not l.startswith("#")]
    full = run([])
    pos = lambda l: int(l.split("\t")[1])
    inside = lambda q: 150 <= q <= 180 or q == 300 or 420 <= q <= 425
    want = [l for l in full if inside(pos(l))]
    assert len(want) == 31 + 1 + 6
    assert run(["-t", "17:150-180,17:300,17:420-425"]) == want
    f = str(tmp_path / "targets.tab")
    open(f, "w").write("17\t150\t180\n17\t300\n17\t420\t425\n")
    assert run(["-T", f]) == want
    # ^: the other columns -- of the reads that overlap the list (as the reference selects them)
    out = run(["-t", "^17:150-180,17:300,17:420-425"])
    assert out and all(not inside(pos(l)) for l in out) and {pos(l) for l in out} < {pos(l) for l in full}

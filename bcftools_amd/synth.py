"""Synthetic 30x-WGS-shaped pileup tiles: the workload of bench.py and of the
seeded parity tests.  Two generators with the same distributions:

    numpy_tile(...)  -> bcftools_amd.host.HostTile         (Philox counter RNG, CPU)
    torch_tile(...)  -> dict of torch tensors on a device  (torch's Philox generator, GPU)

Per site: ref ~ U{A,C,G,T}; variant with prob `var_rate` (alt uniform != ref, population AF ~
Beta(0.5,5) clipped to [1/2S, 0.5], genotypes HWE).  Per (site,sample): depth ~ Poisson(depth)
truncated to [0, max_depth<=200] (keeps n<=255, the regime of mpileup's default -d 250).
Per read: strand ~ Bern(.5); baseQ from an Illumina-like pmf over {2,11,25,37,40,41};
base = true allele, flipped to a uniform other base with prob 10^(-baseQ/10); mapQ = 60 w.p.
0.92 else U{0..59}; read length 150, qpos ~ U{0..149}; soft-clip flag Bern(0.03).
"""
import numpy as np

from . import abi, host

BQ_VALUES = np.array([2, 11, 25, 37, 40, 41], dtype=np.int64)
BQ_PMF = np.array([.02, .05, .08, .35, .30, .20])
READ_LEN = 150


def numpy_tile(seed, n_sites, n_smpl, depth=30.0, var_rate=0.01, max_depth=200, ref_n_rate=0.0, mapq255_rate=0.0,
               wide_qual=False):
    """`wide_qual`: unbinned base qualities (uniform 2..60) and uniform mapQ, i.e. many distinct (quality, strand)
    keys per cell -- the case errmod's sort is there for."""
    rng = np.random.Generator(np.random.Philox(key=int(seed)))
    S = n_smpl
    ref2 = rng.integers(0, 4, n_sites)
    is_var = rng.random(n_sites) < var_rate
    alt2 = (ref2 + rng.integers(1, 4, n_sites)) % 4
    af = np.clip(rng.beta(0.5, 5.0, n_sites), 1.0 / (2 * S), 0.5)
    af = np.where(is_var, af, 0.0)
    ref_is_n = rng.random(n_sites) < ref_n_rate
    ref16 = np.where(ref_is_n, 15, 1 << ref2).astype(np.int8)
    nalt = rng.binomial(2, np.repeat(af, S))                       # [cells]
    n = np.minimum(rng.poisson(depth, n_sites * S), max_depth).astype(np.int64)
    off = np.zeros(n_sites * S + 1, dtype=np.int64)
    np.cumsum(n, out=off[1:])
    R = int(off[-1])
    cell = np.repeat(np.arange(n_sites * S), n)
    site = cell // S
    strand = rng.integers(0, 2, R)
    bq = BQ_VALUES[rng.choice(len(BQ_VALUES), size=R, p=BQ_PMF)]
    if wide_qual:
        bq = rng.integers(2, 61, R)
    is_alt = rng.random(R) < nalt[cell] * 0.5
    base = np.where(is_alt, alt2[site], ref2[site])
    err = rng.random(R) < 10.0 ** (-bq / 10.0)
    base = np.where(err, (base + rng.integers(1, 4, R)) % 4, base)
    mq = np.where(rng.random(R) < 0.92, 60, rng.integers(0, 60, R))
    if wide_qual:
        mq = rng.integers(0, 61, R)
    if mapq255_rate > 0:
        mq = np.where(rng.random(R) < mapq255_rate, 255, mq)
    qpos = rng.integers(0, READ_LEN, R)
    tail = np.minimum(qpos, READ_LEN - 1 - qpos)
    epos = ((qpos + 1).astype(np.float64) / (READ_LEN + 1) * 100).astype(np.uint8)
    sclip = rng.random(R) < 0.03
    rd = (bq | (mq << 8) | ((1 << base) << 16) | (strand << 20) | (sclip.astype(np.int64) << 21) | (tail << 24)).astype(np.uint32)
    return host.HostTile(S, ref16, off.astype(np.uint32), rd, epos)


def _torch_chunk(g, n_sites, S, dev, depth, var_rate, max_depth, fixed_depth=False):
    import torch

    def rnd(n):
        return torch.rand(n, generator=g, device=dev)

    def rint(lo, hi, n):
        return torch.randint(lo, hi, (n,), generator=g, device=dev)

    ref2 = rint(0, 4, n_sites)
    is_var = rnd(n_sites) < var_rate
    alt2 = (ref2 + rint(1, 4, n_sites)) % 4
    # Beta(0.5,5)-like population AF from a power transform of a uniform (cheap, on-device)
    af = torch.clamp(rnd(n_sites) ** 4 * 0.5, 1.0 / (2 * S), 0.5)
    af = torch.where(is_var, af, torch.zeros_like(af))
    ref16 = (1 << ref2).to(torch.int8)
    afc = af.repeat_interleave(S)
    nalt = (rnd(n_sites * S) < afc).to(torch.int64) + (rnd(n_sites * S) < afc).to(torch.int64)
    n = torch.poisson(torch.full((n_sites * S,), float(depth), device=dev), generator=g).to(torch.int64).clamp_(0, max_depth)
    if fixed_depth:                                  # (an experiment's shape: every cell equally deep, no lane of a wavefront waits for a deeper cell)
        n = torch.full_like(n, int(depth))
    R = int(n.sum().item())
    cell = torch.repeat_interleave(torch.arange(n_sites * S, device=dev), n)
    site = cell // S
    strand = rint(0, 2, R)
    cdf = torch.cumsum(torch.tensor(BQ_PMF, device=dev, dtype=torch.float32), 0)
    bq = torch.tensor(BQ_VALUES, device=dev)[torch.bucketize(rnd(R), cdf[:-1], right=True)]
    is_alt = rnd(R) < nalt[cell].to(torch.float32) * 0.5
    base = torch.where(is_alt, alt2[site], ref2[site])
    err = rnd(R) < torch.pow(10.0, -bq.to(torch.float32) / 10.0)
    base = torch.where(err, (base + rint(1, 4, R)) % 4, base)
    mq = torch.where(rnd(R) < 0.92, torch.full((R,), 60, device=dev, dtype=torch.int64), rint(0, 60, R))
    qpos = rint(0, READ_LEN, R)
    tail = torch.minimum(qpos, READ_LEN - 1 - qpos)
    epos = ((qpos + 1).to(torch.float64) / (READ_LEN + 1) * 100).to(torch.uint8)
    sclip = (rnd(R) < 0.03).to(torch.int64)
    rd = (bq | (mq << 8) | ((1 << base) << 16) | (strand << 20) | (sclip << 21) | (tail << 24))
    # int32 bit pattern of the u32 record (torch has no uint32 arithmetic); the C side reads it as u32
    rd32 = torch.where(rd >= 2 ** 31, rd - 2 ** 32, rd).to(torch.int32)
    return ref16, n, rd32, epos


def torch_tile(seed, n_sites, n_smpl, device, depth=30.0, var_rate=0.01, max_depth=200, chunk_cells=1 << 21, fixed_depth=False):
    """Same shape of data generated on `device` with torch's Philox generator, in chunks of sites.
    Returns dict(ref16 i8, plp_off i32 (u32 bit pattern), rd i32 (u32 bit pattern), epos u8, n_reads, ...)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    S = n_smpl
    per = max(1, chunk_cells // S)
    refs, ns, rds, eps = [], [], [], []
    done = 0
    while done < n_sites:
        m = min(per, n_sites - done)
        r16, n, rd32, ep = _torch_chunk(g, m, S, device, depth, var_rate, max_depth, fixed_depth)
        refs.append(r16); ns.append(n); rds.append(rd32); eps.append(ep)
        done += m
    n = torch.cat(ns)
    off = torch.zeros(n_sites * S + 1, dtype=torch.int64, device=device)
    torch.cumsum(n, 0, out=off[1:])
    R = int(off[-1].item())
    assert R < 2 ** 32
    off32 = torch.where(off >= 2 ** 31, off - 2 ** 32, off).to(torch.int32)
    return dict(ref16=torch.cat(refs).contiguous(), plp_off=off32.contiguous(), rd=torch.cat(rds).contiguous(),
                epos=torch.cat(eps).contiguous(), n_reads=R, n_sites=n_sites, n_smpl=S)


def tile_from_torch(t):
    """Copy a torch_tile() dict to a HostTile (for the oracle / CPU baseline)."""
    return host.HostTile(t["n_smpl"], t["ref16"].cpu().numpy(), t["plp_off"].cpu().numpy().view(np.uint32),
                         t["rd"].cpu().numpy().view(np.uint32), t["epos"].cpu().numpy())


def indel_batch(seed, n_sites, n_smpl, depth=30.0, read_len=100, max_depth=200, lens=None, lens2=None, spacing=300):
    """Synthetic input of bcfgpu_gap_prep (BASELINE configs[2] shape: indel-candidate columns): a random reference with
    one indel locus every `spacing` (300) bp (length -3..+3, weights 1/|len|), HWE carriers, Poisson depth, reads of `read_len`
    bases placed so that the locus falls inside them, substitution errors at 10^(-Q/10).  Every pileup entry owns its
    read (the flat pool of bcfgpu_reads allows sharing; sharing does not change the work).
    `lens`: the indel lengths a locus draws its type from (default -3..+3; long ones, e.g. (-40, -25, 12, 8), put the
    realignment's band |type| + 3 of bam2bcf_indel.c:293-294 past the register-resident classes).  `lens2`: a second
    type per locus, drawn from this set and carried by a third of the carriers, so that one column has jobs of several
    band widths (types of 1-3 bp next to a long one).

    Returns dict(ref=bytes, reads=dict of the bcfgpu_reads arrays, pos, smpl_off, p_read, p_qpos, p_indel, itype)."""
    rng = np.random.Generator(np.random.Philox(key=int(seed)))
    S = n_smpl
    L = 200 + spacing * n_sites + spacing
    ref2 = rng.integers(0, 4, L)
    pos = (200 + spacing * np.arange(n_sites)).astype(np.int32)
    lens = np.array([-3, -2, -1, 1, 2, 3] if lens is None else list(lens))
    all_lens = np.concatenate([lens, [] if lens2 is None else np.array(list(lens2))]).astype(np.int64)
    max_len = int(max(3, all_lens.max()))                              # the longest insertion (inside the read); deletions only widen the span
    assert max_len + 24 < read_len and read_len - int(min(0, all_lens.min())) < spacing - 10      # (loci are `spacing` bases apart)
    w = 1.0 / np.abs(lens)
    itype = lens[rng.choice(len(lens), size=n_sites, p=w / w.sum())]
    ins2 = rng.integers(0, 4, (n_sites, max_len))
    itype2 = None
    if lens2 is not None:
        l2 = np.array(list(lens2))
        itype2 = l2[rng.integers(0, len(l2), n_sites)]
    af = np.clip(rng.beta(0.5, 5.0, n_sites), 0.05, 0.5)
    nalt = rng.binomial(2, np.repeat(af, S))
    n = np.minimum(rng.poisson(depth, n_sites * S), max_depth).astype(np.int64)
    off = np.zeros(n_sites * S + 1, dtype=np.int64)
    np.cumsum(n, out=off[1:])
    R = int(off[-1])
    cell = np.repeat(np.arange(n_sites * S), n)
    site = cell // S
    carrier = rng.random(R) < nalt[cell] * 0.5
    carrier ^= rng.random(R) < 0.005                               # alignment/sequencing indel noise
    qpos = rng.integers(8, read_len - 12 - (max_len if max_len > 3 else 0), R)
    start = pos[site].astype(np.int64) - qpos
    ilen = np.where(carrier, itype[site], 0).astype(np.int64)
    if itype2 is not None:
        ilen = np.where(carrier & (rng.random(R) < 1.0 / 3), itype2[site], ilen)
    j = np.arange(read_len)[None, :]
    after = j > qpos[:, None]
    shift = np.where(ilen[:, None] < 0, -ilen[:, None], -np.minimum(ilen[:, None], np.maximum(j - qpos[:, None], 0)))
    base = ref2[start[:, None] + j + np.where(after, shift, 0)]
    k = j - qpos[:, None] - 1
    is_ins = after & (ilen[:, None] > 0) & (k < ilen[:, None])
    base = np.where(is_ins, ins2[site[:, None], np.clip(k, 0, max_len - 1)], base)
    bq = BQ_VALUES[rng.choice(len(BQ_VALUES), size=(R, read_len), p=BQ_PMF)]
    err = rng.random((R, read_len)) < 10.0 ** (-bq / 10.0)
    base = np.where(err, (base + rng.integers(1, 4, (R, read_len))) % 4, base)
    seq16 = (1 << base).astype(np.uint8)
    # CIGAR (BAM encoding len<<4|op, M=0 I=1 D=2): 100M, or aM xD bM / aM xI bM around the locus
    a = qpos + 1
    al = np.abs(ilen)
    c3 = np.stack([a << 4, (al << 4) | np.where(ilen > 0, 1, 2), (read_len - a - np.where(ilen > 0, al, 0)) << 4], axis=1)
    ncig = np.where(ilen != 0, 3, 1).astype(np.int32)
    c3[ilen == 0, 0] = read_len << 4
    keep = np.arange(3)[None, :] < ncig[:, None]
    cig = c3[keep].astype(np.uint32)
    i32 = lambda x: np.ascontiguousarray(x, dtype=np.int32)
    reads = dict(n_reads=R, r_pos=i32(start), r_lq=i32(np.full(R, read_len)), r_flag=i32(rng.integers(0, 2, R) * 16),
                 r_ncig=ncig, r_cig_off=i32(np.concatenate([[0], np.cumsum(ncig)[:-1]])),
                 r_seq_off=i32(np.arange(R, dtype=np.int64) * read_len), cig=np.ascontiguousarray(cig),
                 seq16=np.ascontiguousarray(seq16.ravel()), qual=np.ascontiguousarray(bq.astype(np.uint8).ravel()),
                 zq=np.zeros(R * read_len, dtype=np.uint8), r_has_zq=np.zeros(R, dtype=np.uint8))
    return dict(ref=bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[ref2]), reads=reads, pos=pos, smpl_off=i32(off),
                p_read=i32(np.arange(R)), p_qpos=i32(qpos), p_indel=i32(ilen), itype=itype, n_sites=n_sites, n_smpl=S)


def indel_pool(b, mapq=60):
    """The reads of an indel_batch() as the pool bcfgpu_pileup takes: grouped by sample, ascending position inside a sample
    (a position-sorted BAM per sample).  Returns (reads dict, r_mapq, r_smpl, order): read k of the pool is read order[k]
    of the batch; the entries of a pileup column come out in pool order."""
    S, R = b["n_smpl"], b["reads"]
    n = R["n_reads"]
    cell = np.repeat(np.arange(b["n_sites"] * S), np.diff(b["smpl_off"]))      # every entry owns its read: read i = entry i
    smpl = (cell % S).astype(np.int32)
    order = np.lexsort((np.arange(n), R["r_pos"], smpl))
    L = int(R["r_lq"][0])
    assert (R["r_lq"] == L).all()
    i32 = lambda x: np.ascontiguousarray(x, dtype=np.int32)
    ncig = R["r_ncig"][order]
    coff = np.concatenate([[0], np.cumsum(ncig)[:-1]])
    idx = np.repeat(R["r_cig_off"][order] - coff, ncig) + np.arange(int(ncig.sum()))
    reads = dict(n_reads=n, r_pos=i32(R["r_pos"][order]), r_lq=i32(R["r_lq"][order]), r_flag=i32(R["r_flag"][order]),
                 r_ncig=i32(ncig), r_cig_off=i32(coff), r_seq_off=i32(np.arange(n, dtype=np.int64) * L),
                 cig=np.ascontiguousarray(R["cig"][idx]),
                 seq16=np.ascontiguousarray(R["seq16"].reshape(n, L)[order].ravel()),
                 qual=np.ascontiguousarray(R["qual"].reshape(n, L)[order].ravel()),
                 zq=np.zeros(n * L, dtype=np.uint8), r_has_zq=np.zeros(n, dtype=np.uint8))
    return reads, np.full(n, mapq, dtype=np.uint8), np.ascontiguousarray(smpl[order]), order


def indel_tile_from_batch(b, aux, ret, mapq=60):
    """The indel pass of mpileup (mpileup.c:354-365) for the columns of an indel_batch() where bcf_call_gap_prep
    returned 0: the same pileup entries as a tile with ref_base = -1 and aux = p->aux."""
    S = b["n_smpl"]
    R = b["reads"]
    live = np.nonzero(np.asarray(ret) == 0)[0]
    so = b["smpl_off"].astype(np.int64)
    cells = (live[:, None] * S + np.arange(S)[None, :]).ravel()
    beg, end = so[cells], so[cells + 1]
    n = end - beg
    off = np.zeros(len(cells) + 1, dtype=np.int64)
    np.cumsum(n, out=off[1:])
    e = np.repeat(beg - off[:-1], n) + np.arange(int(off[-1]))            # pileup entries, column-major
    r, qpos = b["p_read"][e].astype(np.int64), b["p_qpos"][e].astype(np.int64)
    lq = R["r_lq"][r].astype(np.int64)
    at = R["r_seq_off"][r].astype(np.int64) + qpos
    tail = np.minimum(np.minimum(qpos, lq - 1 - qpos), 255)
    rd = (R["qual"][at].astype(np.int64) | (mapq << 8) | (R["seq16"][at].astype(np.int64) << 16)
          | ((R["r_flag"][r].astype(np.int64) >> 4 & 1) << 20) | (tail << 24)).astype(np.uint32)
    # no soft clips in these reads: the aligned length is the read length (bam2bcf.c:80-114)
    epos = ((qpos + 1).astype(np.float64) / (lq + 1) * 100).astype(np.uint8)
    return host.HostTile(S, np.zeros(len(live), dtype=np.int8), off.astype(np.uint32), rd, epos,
                         aux=np.asarray(aux, dtype=np.uint32)[e], is_indel=1), live


def wgs_reads(seed, n_sites, n_smpl, depth=30.0, indel_read_rate=0.005, true_indel_rate=0.01, long_indel_frac=0.05):
    """BASELINE configs[3] as READS (bench.py --mode wgs, tests/test_gpu_fullsize.py): `n_smpl` samples x `depth` of 100-base reads
    over `n_sites` columns of one contig (position-sorted inside a sample), substitution errors at 0.3 %, four binned base
    qualities, `indel_read_rate` of the reads with a noise indel and true indel sites at `true_indel_rate` of the columns (allele
    frequency 0.1, HWE genotypes) -- indels of 1..3 bases by 1/len or, `long_indel_frac` of them, 8 / 12 / 25 / 40 bases by 1/len.
    The bases and qualities are drawn on the device (torch) and brought to the host as the arrays of bcfgpu_reads.
    Returns dict(reads, mapq, smpl, refseq, read_len, beg, end, n_reads, n_true)."""
    import torch
    S = n_smpl
    L, beg = 100, 300
    end = beg + n_sites
    rng = np.random.default_rng(seed)
    per = int((n_sites + L) * depth / L)
    n = per * S
    pos = np.sort(rng.integers(beg - L + 1, end, size=(S, per)), axis=1).astype(np.int32).ravel()
    smpl = np.repeat(np.arange(S, dtype=np.int32), per)
    ref_codes = rng.integers(0, 4, end + 3 * L).astype(np.uint8)
    refseq = "".join("ACGT"[i] for i in ref_codes)
    # ---- indels: noise on a fraction of the reads (anywhere 10 bases off the ends; length 1..3 with weights 1/len, or -- one in
    # twenty -- 8..40 bases, the lengths whose realignment band |type| + 3 is past the register-resident classes), and true indel
    # sites (allele frequency 0.1, genotypes HWE, the same mix of lengths) carried by the reads that span them ----
    lens = np.array([1, 2, 3]); w = 1.0 / lens
    long_lens = np.array([8, 12, 25, 40])

    def draw_lens(k):
        v = lens[rng.choice(3, k, p=w / w.sum())]
        lg = rng.random(k) < long_indel_frac
        v[lg] = long_lens[rng.choice(len(long_lens), int(lg.sum()), p=(1.0 / long_lens) / (1.0 / long_lens).sum())]      # weights 1/len as well
        return v * rng.choice([-1, 1], k)
    ilen = np.zeros(n, np.int64)                                   # > 0 insertion, < 0 deletion
    ioff = np.zeros(n, np.int64)                                   # query bases before the indel
    noisy = rng.random(n) < indel_read_rate
    ilen[noisy] = draw_lens(int(noisy.sum()))
    ioff[noisy] = rng.integers(10, L - 10 - np.maximum(ilen[noisy], 0))
    n_true = int(n_sites * true_indel_rate + 0.5)
    if n_true:
        tsite = np.sort(rng.choice(np.arange(beg + 20, end - 20), n_true, replace=False)).astype(np.int64)
        tlen = draw_lens(n_true)
        geno = rng.binomial(2, 0.1, (S, n_true))
        k = np.searchsorted(tsite, pos.astype(np.int64) + 10)
        kk = np.minimum(k, n_true - 1)
        span = (k < n_true) & (tsite[kk] < pos.astype(np.int64) + L - 10 - np.maximum(tlen[kk], 0))
        carry = span & (rng.random(n) < geno[smpl, kk] * 0.5)
        ilen[carry] = tlen[kk[carry]]
        ioff[carry] = tsite[kk[carry]] - pos[carry] + 1             # the indel follows reference position tsite
    al = np.abs(ilen)
    ins = np.where(ilen > 0, al, 0)
    c3 = np.stack([ioff << 4, (al << 4) | np.where(ilen > 0, 1, 2), (L - ioff - ins) << 4], axis=1)
    ncig = np.where(ilen != 0, 3, 1).astype(np.int32)
    c3[ilen == 0, 0] = L << 4
    cig = c3[np.arange(3)[None, :] < ncig[:, None]].astype(np.uint32)
    cig_off = np.concatenate([[0], np.cumsum(ncig)[:-1]]).astype(np.int32)
    # the bases and qualities (n x L bytes each: half a gigabyte at the default size) are drawn on the device, in chunks of reads
    dev = torch.device("cuda", 0)
    tg = torch.Generator(device=dev)
    tg.manual_seed(int(seed))
    ref_t = torch.from_numpy(ref_codes.astype(np.int64)).to(dev)
    seq_t = torch.empty(n * L, dtype=torch.uint8, device=dev)
    qual_t = torch.empty(n * L, dtype=torch.uint8, device=dev)
    q_vals = torch.tensor([11, 25, 37, 40], dtype=torch.uint8, device=dev)
    q_cdf = torch.tensor([0.07, 0.15, 0.50], device=dev)
    j = torch.arange(L, dtype=torch.int64, device=dev)[None, :]
    CH = 1 << 19
    for r0 in range(0, n, CH):
        r1 = min(n, r0 + CH)
        il = torch.from_numpy(ilen[r0:r1]).to(dev)[:, None]
        io = torch.from_numpy(ioff[r0:r1]).to(dev)[:, None]
        p0 = torch.from_numpy(pos[r0:r1].astype(np.int64)).to(dev).clamp_(min=0)[:, None]
        after = j >= io
        shift = torch.where(il < 0, -il, -torch.minimum(il, (j - io).clamp(min=0)))
        b = ref_t[p0 + j + torch.where(after, shift, torch.zeros_like(shift))]
        is_ins = after & (il > 0) & (j - io < il)
        b = torch.where(is_ins, torch.randint(0, 4, b.shape, generator=tg, device=dev), b)
        err = torch.rand(b.shape, generator=tg, device=dev) < 0.003
        b = torch.where(err, (b + torch.randint(1, 4, b.shape, generator=tg, device=dev)) & 3, b)
        seq_t[r0 * L:r1 * L] = (1 << b).to(torch.uint8).reshape(-1)
        qual_t[r0 * L:r1 * L] = q_vals[torch.bucketize(torch.rand((r1 - r0) * L, generator=tg, device=dev), q_cdf)]
    seq, qual = seq_t.cpu().numpy(), qual_t.cpu().numpy()
    del seq_t, qual_t, ref_t
    torch.cuda.empty_cache()
    mapq = np.where(rng.random(n) < 0.92, 60, rng.integers(0, 60, n)).astype(np.uint8)
    arrs = dict(r_pos=pos, r_lq=np.full(n, L, np.int32), r_flag=(rng.integers(0, 2, n) * 16).astype(np.int32), r_ncig=ncig, r_cig_off=cig_off,
                r_seq_off=(np.arange(n, dtype=np.int64) * L).astype(np.int32), cig=cig, seq16=seq, qual=qual, zq=np.zeros(1, np.uint8),
                r_has_zq=np.zeros(n, np.uint8))
    return dict(reads=arrs, mapq=mapq, smpl=smpl, refseq=refseq, read_len=L, beg=beg, end=end, n_reads=n, n_true=n_true,
                ilen=ilen, ioff=ioff)          # (per read: the indel it carries, > 0 insertion, and the query bases in front of it)

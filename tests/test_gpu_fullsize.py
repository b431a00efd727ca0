"""Parity at BASELINE.json's full per-GPU shape (1000 samples x 30x, a 4096-site tile = 1.2e8 reads), where the oracle
cannot run the whole input in seconds: size-independent properties plus an oracle spot check.

 * determinism: two runs give identical bytes (the kernels use integer atomics only);
 * region-shard invariance: the tile run as two half tiles (what two GPUs would do) gives the same records -- workgroup
   boundaries fall on different cells, so this also exercises the staging/boundary logic at scale;
 * conservation laws of the record: DP = sum of DP4, AD totals = plane sums, AN/AC = genotype counts;
 * a random sample of sites re-run through the oracle: integers exact, QUAL 1e-4."""
import numpy as np
import pytest

from bcftools_amd import abi, synth, host
from tests.helpers import orc
from tests.test_gpu_parity import assert_mplp_equal, assert_call_equal

pytestmark = pytest.mark.gpu

N_SITES, N_SMPL = 4096, 1000
FMT = abi.INFO_VDB | abi.INFO_RPB | abi.FMT_AD


@pytest.fixture(scope="module")
def big(gpu_ctx_factory):
    import torch
    t = synth.torch_tile(20260105, N_SITES, N_SMPL, torch.device("cuda", 0), depth=30.0, var_rate=0.02)
    tile = synth.tile_from_torch(t)
    del t
    torch.cuda.empty_cache()
    cfg = abi.default_cfg(N_SMPL, max_sites=N_SITES, max_reads=len(tile.rd), fmt_flag=FMT)
    ctx = gpu_ctx_factory(cfg)
    m, c = ctx.pipeline(tile)
    return tile, cfg, ctx, m, c


def _same(a, b):
    for k in ["pl", "dp4", "adf", "adr"]:
        np.testing.assert_array_equal(getattr(a[0], k), getattr(b[0], k), err_msg=k)
    assert a[0].site.tobytes() == b[0].site.tobytes()
    assert a[1].site.tobytes() == b[1].site.tobytes()
    np.testing.assert_array_equal(a[1].gt, b[1].gt)
    np.testing.assert_array_equal(a[1].pl, b[1].pl)


def test_two_runs_are_identical(big):
    tile, cfg, ctx, m, c = big
    m2, c2 = ctx.pipeline(tile)
    _same((m, c), (m2, c2))


def test_region_shards_give_the_same_records(big):
    tile, cfg, ctx, m, c = big
    cut = 1777                                            # not a multiple of anything: shifts every workgroup boundary
    parts = [ctx.pipeline(tile.select_sites(np.arange(a, b))) for a, b in ((0, cut), (cut, N_SITES))]
    for k in ["pl", "dp4", "adf", "adr"]:
        np.testing.assert_array_equal(np.concatenate([getattr(p[0], k) for p in parts]), getattr(m, k), err_msg=k)
    assert np.concatenate([p[0].site for p in parts]).tobytes() == m.site.tobytes()
    assert np.concatenate([p[1].site for p in parts]).tobytes() == c.site.tobytes()
    np.testing.assert_array_equal(np.concatenate([p[1].gt for p in parts]), c.gt)


def test_record_conservation_laws(big):
    tile, cfg, ctx, m, c = big
    n = np.diff(tile.plp_off.astype(np.int64)).reshape(N_SITES, N_SMPL)
    assert n.sum() == len(tile.rd)
    # bam2bcf.c:650-659,718-727: anno[0..3] are the sums of the DP4 planes and add up to the depth
    dp4 = m.dp4.astype(np.int64).sum(axis=2)
    np.testing.assert_array_equal(dp4, m.site["anno"][:, :4].astype(np.int64))
    np.testing.assert_array_equal(dp4.sum(axis=1), m.site["depth"].astype(np.int64))
    np.testing.assert_array_equal(m.site["ori_depth"].astype(np.int64), n.sum(axis=1))     # no skipped/deleted reads here
    assert (m.site["depth"] <= m.site["ori_depth"]).all()
    # bam2bcf.c:676-697: site AD totals are the plane sums; every counted read shows one of the listed alleles or is in none
    np.testing.assert_array_equal(m.adf.astype(np.int64).sum(axis=2), m.site["adf_tot"])
    np.testing.assert_array_equal(m.adr.astype(np.int64).sum(axis=2), m.site["adr_tot"])
    # mcall.c:745-886: AN/AC are the genotype counts of the called samples
    live = c.site["ret"] > 0
    g = c.gt[live].astype(np.int64)
    called = g >= 0
    np.testing.assert_array_equal(called.sum(axis=(1, 2)), c.site["an"][live])
    for a in range(5):
        np.testing.assert_array_equal(((g == a) & called).sum(axis=(1, 2)), c.site["ac"][live][:, a], err_msg="AC[%d]" % a)
    # PL of a called genotype is the row minimum 0 wherever the sample has data
    assert ((m.pl.min(axis=1) == 0) | (n == 0)).all()


def test_sampled_sites_match_oracle(big):
    tile, cfg, ctx, m, c = big
    rng = np.random.default_rng(5)
    var = np.nonzero(c.site["als_new"] != 1)[0]
    pick = np.unique(np.concatenate([rng.choice(N_SITES, 8, replace=False), rng.choice(var, min(8, len(var)), replace=False)]))
    sub = tile.select_sites(pick)
    scfg = abi.default_cfg(N_SMPL, max_sites=sub.n_sites, max_reads=len(sub.rd), fmt_flag=FMT)
    mw = orc.mpileup(scfg, sub)
    cin = host.CallInput(N_SMPL, mw.site["n_alleles"], np.maximum(mw.site["unseen"], 0), mw.pl.astype(np.int32), mw.site["qsum"],
                         i16=mw.site["anno"].astype(np.float32))
    cw = orc.mcall(scfg, cin)

    class Rows:                                            # the picked rows of the full-size results
        pass
    mg, cg = Rows(), Rows()
    mg.site = m.site[pick]
    for k in ["pl", "dp4", "adf", "adr", "qs", "scr", "sp"]:
        setattr(mg, k, getattr(m, k)[pick])
    cg.site, cg.gt, cg.pl = c.site[pick], c.gt[pick], c.pl[pick]
    mw.qs[:] = mg.qs                                       # FMT/QS and SCR planes are not requested in this run
    mw.scr[:] = mg.scr
    assert_mplp_equal(mg, mw)
    assert_call_equal(cg, cw, N_SMPL)


def test_groups_and_ploidy_at_full_size(big, gpu_ctx_factory):
    """BASELINE configs[4] shape on the same tile: call -G on four sample groups (frequencies from FORMAT/AD) with a ploidy array.
    Region-shard invariance of the call records, and a sample of sites -- variant ones among them -- against the oracle."""
    tile, cfg0, ctx0, m0, c0 = big
    n_grp = 4
    cfg = abi.default_cfg(N_SMPL, max_sites=N_SITES, max_reads=len(tile.rd), fmt_flag=FMT, n_grp=n_grp)
    ctx = gpu_ctx_factory(cfg)
    rng = np.random.default_rng(11)
    ploidy = rng.choice([0, 1, 2, 2, 2], size=N_SMPL).astype(np.uint8)
    grp = (np.arange(N_SMPL) * n_grp // N_SMPL).astype(np.int32)
    m, c = ctx.pipeline(tile, ploidy=ploidy, grp=grp)
    np.testing.assert_array_equal(m.pl, m0.pl)                       # the mpileup stage does not know about groups
    cut = 2311
    parts = [ctx.pipeline(tile.select_sites(np.arange(a, b)), ploidy=ploidy, grp=grp) for a, b in ((0, cut), (cut, N_SITES))]
    assert np.concatenate([p[1].site for p in parts]).tobytes() == c.site.tobytes()
    np.testing.assert_array_equal(np.concatenate([p[1].gt for p in parts]), c.gt)
    var = np.nonzero(c.site["als_new"] != 1)[0]
    assert len(var) > 0
    pick = np.unique(np.concatenate([rng.choice(N_SITES, 6, replace=False), rng.choice(var, min(10, len(var)), replace=False)]))
    sub = tile.select_sites(pick)
    scfg = abi.default_cfg(N_SMPL, max_sites=sub.n_sites, max_reads=len(sub.rd), fmt_flag=FMT, n_grp=n_grp)
    mw = orc.mpileup(scfg, sub)
    na = mw.site["n_alleles"]
    src = mw.adf.astype(np.int32) + mw.adr.astype(np.int32)
    ad = np.where(np.arange(5)[None, :, None] < na[:, None, None], src, abi.INT32_VECTOR_END).astype(np.int32)
    cin = host.CallInput(N_SMPL, na, np.maximum(mw.site["unseen"], 0), mw.pl.astype(np.int32), mw.site["qsum"],
                         ad=ad, ploidy=ploidy, grp=grp, i16=mw.site["anno"].astype(np.float32))
    cw = orc.mcall(scfg, cin)

    class Rows:
        pass
    cg = Rows()
    cg.site, cg.gt, cg.pl = c.site[pick], c.gt[pick], c.pl[pick]
    assert_call_equal(cg, cw, N_SMPL)


def test_pileup_at_scale_counts_and_packed_form(gpu_ctx_factory):
    """bcfgpu_pileup over 2048 columns x 1000 samples x 30x (6e7 entries): every cell's entry count equals the number of the
    sample's reads whose reference span covers the column (a difference array on the host), and the packed form of the same
    pool (4-bit bases, 2-bit palette qualities, 12-byte read records, the samples' offsets) gives the same tile byte for byte."""
    import ctypes as C
    from bcftools_amd.lib import check
    S, n_sites, L, beg = N_SMPL, 2048, 100, 300
    end = beg + n_sites
    rng = np.random.default_rng(20260106)
    per = int((n_sites + L) * 30.0 / L)
    n = per * S
    pos = np.sort(rng.integers(beg - L + 1, end, size=(S, per)), axis=1).astype(np.int32).ravel()
    smpl = np.repeat(np.arange(S, dtype=np.int32), per)
    kind = rng.choice(4, n, p=[0.94, 0.02, 0.02, 0.02])
    table = {0: [L << 4], 1: [48 << 4, 2 << 4 | 2, 50 << 4], 2: [40 << 4, 3 << 4 | 1, 57 << 4], 3: [8 << 4 | 4, 92 << 4]}
    span = np.array([100, 100, 97, 92], np.int32)[kind]              # reference bases a read of each kind covers
    ncig = np.array([1, 3, 3, 2], np.int32)[kind]
    cig_off = np.concatenate([[0], np.cumsum(ncig)[:-1]]).astype(np.int32)
    cig = np.zeros(int(ncig.sum()), np.uint32)
    for k, ops in table.items():
        idx = np.nonzero(kind == k)[0]
        for j, op in enumerate(ops):
            cig[cig_off[idx] + j] = op
    refseq = "".join("ACGT"[i] for i in rng.integers(0, 4, end + 2 * L))
    seq = (1 << rng.integers(0, 4, n * L)).astype(np.uint8)
    palette = np.array([2, 12, 23, 37], np.uint8)
    qidx = rng.integers(0, 4, n * L).astype(np.uint8)
    qual = palette[qidx]
    lq = np.full(n, L, np.int32)
    flag = (rng.integers(0, 2, n) * 16).astype(np.int32)
    seq_off = (np.arange(n, dtype=np.int64) * L).astype(np.int32)
    mapq = rng.integers(0, 61, n).astype(np.uint8)
    rd = abi.Reads()
    rd.n_reads = n
    keep = dict(r_pos=pos, r_lq=lq, r_flag=flag, r_ncig=ncig, r_cig_off=cig_off, r_seq_off=seq_off, cig=cig, seq16=seq, qual=qual)
    for k, v in keep.items():
        setattr(rd, k, v.ctypes.data)
    ctx = gpu_ctx_factory(abi.default_cfg(S, max_sites=1, max_reads=64))
    t = abi.Tile()
    col_n = np.zeros(n_sites, np.int32)
    check(ctx.L.bcfgpu_pileup(ctx.h, C.byref(rd), mapq.ctypes.data, smpl.ctypes.data, beg, end, refseq.encode(), len(refseq), C.byref(t), col_n.ctypes.data, None))
    total = int(t.n_reads)

    def fetch():
        off = np.zeros(n_sites * S + 1, np.uint32)
        w, e = np.zeros(total, np.uint32), np.zeros(total, np.uint8)
        for dst, src in ((off, t.plp_off), (w, t.rd), (e, t.epos)):
            check(ctx.L.bcfgpu_memcpy_d2h(ctx.h, dst.ctypes.data, src, dst.nbytes))
        ctx.sync()
        return off, w, e
    off, w, e = fetch()
    # coverage by a difference array: +1 at the read's first column, -1 past its last
    d = np.zeros((S, n_sites + 1), np.int32)
    a = np.clip(pos - beg, 0, n_sites)
    b = np.clip(pos + span - beg, 0, n_sites)
    np.add.at(d, (smpl, a), 1)
    np.add.at(d, (smpl, b), -1)
    cov = np.cumsum(d[:, :n_sites], axis=1).T                         # [site][sample]
    np.testing.assert_array_equal(np.diff(off.astype(np.int64)).reshape(n_sites, S), cov)
    np.testing.assert_array_equal(col_n, cov.sum(axis=1))
    assert total == int(cov.sum()) and total > 5e7
    # the packed form of the same pool
    pk = abi.Packed()
    seq4, qual2 = abi.pack_nibbles(seq), abi.pack_crumbs(qidx)
    rec = abi.read12(pos, lq, ncig, flag, mapq)
    soff = (np.arange(S + 1, dtype=np.int64) * per).astype(np.int32)
    pk.seq4, pk.qual4, pk.qual_bits, pk.recs, pk.smpl_off = seq4.ctypes.data, qual2.ctypes.data, 2, rec.ctypes.data, soff.ctypes.data
    pk.n_bases, pk.n_cig = n * L, len(cig)
    for j, q in enumerate(palette):
        pk.palette[j] = int(q)
    rd2 = abi.Reads()
    rd2.n_reads, rd2.cig = n, cig.ctypes.data
    check(ctx.L.bcfgpu_pileup_packed(ctx.h, C.byref(rd2), C.byref(pk), None, None, beg, end, refseq.encode(), len(refseq), C.byref(t), None, None))
    assert int(t.n_reads) == total
    off2, w2, e2 = fetch()
    assert off2.tobytes() == off.tobytes() and w2.tobytes() == w.tobytes() and e2.tobytes() == e.tobytes()


def test_indel_path_at_tile_scale(gpu_ctx_factory):
    """bcf_call_gap_prep at the cohort's shape (1000 samples x 30x of reads, 2048 columns, 0.5 % indel-noise reads, true indel
    columns, a twentieth of the indels 8-40 bases), where the oracle takes a second a column: size-independent properties plus
    an oracle spot check.
     * the candidates the pooled support filter turns away (bam2bcf_indel.c:150-154) are exactly the columns with ret = -1 and
       nothing else of them comes out; the ones it passes follow from the pileup's two per-column counts;
     * partition invariance: the candidate list run as one call, and as two calls over its halves, gives the same per-column
       results and the same indel tile (entries, p->aux) -- what two tiles, or two GPUs, would do;
     * determinism: a second run gives identical bytes;
     * three realigned columns and two rejected ones re-run through the oracle (types, p->aux of every entry)."""
    import ctypes as C
    from bcftools_amd import engine
    from bcftools_amd.lib import check
    from tests.helpers import indeldrv
    S, n_sites = 1000, 2048
    W = synth.wgs_reads(20260106, n_sites, S, 30.0)
    arrs, mapq, smpl, n, L, beg, end = W["reads"], W["mapq"], W["smpl"], W["n_reads"], W["read_len"], W["beg"], W["end"]
    ref_b = W["refseq"].encode()
    rd = abi.Reads()
    rd.n_reads = n
    for k, v in arrs.items():
        setattr(rd, k, v.ctypes.data)
    with engine.Context(abi.default_cfg(S, max_sites=1, max_reads=64)) as c0:
        t = abi.Tile()
        check(c0.L.bcfgpu_pileup(c0.h, C.byref(rd), mapq.ctypes.data, smpl.ctypes.data, beg, end, ref_b, len(ref_b), C.byref(t), None, None))
        entries = int(t.n_reads)
    ctx = gpu_ctx_factory(abi.default_cfg(S, max_sites=n_sites, max_reads=entries + 64))
    Lb = ctx.L
    tile = abi.Tile()
    col_n, col_indel = np.zeros(n_sites, np.int32), np.zeros(n_sites, np.uint8)
    check(Lb.bcfgpu_pool_upload(ctx.h, C.byref(rd), None, mapq.ctypes.data))
    check(Lb.bcfgpu_pool_pileup(ctx.h, smpl.ctypes.data, None, beg, end, ref_b, len(ref_b), C.byref(tile), col_n.ctypes.data, col_indel.ctypes.data))
    cand = np.ascontiguousarray(np.nonzero((col_indel != 0) & (col_n < 250 * S))[0], dtype=np.int32)
    assert len(cand) > n_sites // 2                                 # with 1000 samples most columns have some read with an indel
    par = abi.IndelIn()
    par.ref = ref_b
    for k, v in indeldrv.DEFAULTS.items():
        setattr(par, k, v)
    CAP = indeldrv.CAP

    def run(cols):
        nc = len(cols)
        out = dict(ret=np.zeros(nc, np.int32), indel_types=np.zeros((nc, 4), np.int32), inscns=np.zeros((nc, 4 * CAP), np.int8),
                   maxins=np.zeros(nc, np.int32), indelreg=np.zeros(nc, np.int32), max_support=np.zeros(nc, np.int32), max_frac=np.zeros(nc, np.float32))
        oo = abi.IndelOut()
        oo.ret, oo.indel_types, oo.inscns = out["ret"].ctypes.data, out["indel_types"].ctypes.data, out["inscns"].ctypes.data
        oo.maxins, oo.indelreg, oo.max_support, oo.max_frac = (out["maxins"].ctypes.data, out["indelreg"].ctypes.data,
                                                               out["max_support"].ctypes.data, out["max_frac"].ctypes.data)
        it = abi.Tile()
        cc = np.ascontiguousarray(cols, dtype=np.int32)
        check(Lb.bcfgpu_gap_prep_tile(ctx.h, nc, cc.ctypes.data, None, C.byref(par), C.byref(oo), CAP, C.byref(it)))
        off = np.zeros(it.n_sites * S + 1, np.uint32)
        aux, rdw = np.zeros(it.n_reads, np.uint32), np.zeros(it.n_reads, np.uint32)
        if it.n_sites:
            check(Lb.bcfgpu_memcpy_d2h(ctx.h, off.ctypes.data, it.plp_off, off.nbytes))
            check(Lb.bcfgpu_memcpy_d2h(ctx.h, aux.ctypes.data, it.aux, aux.nbytes))
            check(Lb.bcfgpu_memcpy_d2h(ctx.h, rdw.ctypes.data, it.rd, rdw.nbytes))
        st = abi.GapStats()
        check(Lb.bcfgpu_gap_prep_stats(ctx.h, C.byref(st)))
        return out, off, aux, rdw, st
    whole, off, aux, rdw, st = run(cand)
    live = whole["ret"] == 0
    assert 5 <= live.sum() <= n_sites // 20 and st.n_wide > 0       # the true indel columns (about 1 %), some of them long
    assert (whole["indel_types"][~live] == 10000).all() and (whole["maxins"][~live] == 0).all()
    assert len(off) == live.sum() * S + 1 and off[-1] == len(aux) == int(col_n[cand[live]].sum())
    # determinism
    again = run(cand)
    for k in whole:
        np.testing.assert_array_equal(again[0][k], whole[k], err_msg=k)
    np.testing.assert_array_equal(again[2], aux)
    # the candidate list in two calls
    cut = len(cand) // 2 + 7
    a, b = run(cand[:cut]), run(cand[cut:])
    for k in whole:
        np.testing.assert_array_equal(np.concatenate([a[0][k], b[0][k]]), whole[k], err_msg=k)
    np.testing.assert_array_equal(np.concatenate([a[2], b[2]]), aux, err_msg="p->aux of the indel tile")
    np.testing.assert_array_equal(np.concatenate([a[3], b[3]]), rdw, err_msg="read records of the indel tile")
    # oracle spot check: realigned columns (a long type among them if there is one), rejected ones, and -- the case a call without
    # any long TYPE must still get right -- a column of 1-3 base types one of whose reads carries a long indel elsewhere in the
    # window: that read's realignments have the band |l_ref - l_query| (probaln.c), past the register classes
    lt = whole["indel_types"]
    has_long = ((np.abs(lt) >= 8) & (lt != 10000)).any(axis=1)
    longc = [int(i) for i in np.nonzero(live & has_long)[0][:1]]
    rpos, rlen, roff = arrs["r_pos"].astype(np.int64), W["ilen"], W["ioff"]
    far = []
    for ci in np.nonzero(live & ~has_long)[0]:
        x = int(cand[ci]) + beg
        at = rpos + roff                                            # the reference position in front of the read's indel
        if ((np.abs(rlen) >= 8) & (rpos <= x) & (x < rpos + L) & (np.abs(at - x) > 3) & (np.abs(at - x) < 45)).any():
            far.append(int(ci))
    assert far, "no short-type column with a long indel in a read's window: another seed"
    short_only = np.nonzero(~(live & has_long))[0]                  # a call whose columns have no long type at all
    sh = run(cand[short_only])
    for k in whole:
        np.testing.assert_array_equal(sh[0][k], whole[k][short_only], err_msg="short-type columns alone: " + k)
    np.testing.assert_array_equal(sh[2], np.concatenate([aux[off[j * S]:off[(j + 1) * S]] for j in np.nonzero((~has_long)[live])[0]]),
                                  err_msg="short-type columns alone: p->aux")
    assert sh[4].n_wide > 0                                         # wide-band jobs without a wide type
    pick = sorted(set(longc + far[:2] + [int(i) for i in np.nonzero(live)[0][:1]] + [int(i) for i in np.nonzero(~live)[0][:2]]))
    cols = np.ascontiguousarray(cand[pick])
    tot_e = int(col_n[cols].sum())
    so = np.zeros(len(pick) * S + 1, np.int32)
    pr, pq, pi = np.zeros(tot_e, np.int32), np.zeros(tot_e, np.int32), np.zeros(tot_e, np.int32)
    check(Lb.bcfgpu_pileup_entries(ctx.h, len(pick), cols.ctypes.data, so.ctypes.data, pr.ctypes.data, pq.ctypes.data, pi.ctypes.data, tot_e))
    bb = dict(n_sites=len(pick), n_smpl=S, ref=ref_b, pos=(cols + beg).astype(np.int32), smpl_off=so, p_read=pr, p_qpos=pq, p_indel=pi,
              reads=dict(arrs, n_reads=n, zq=np.zeros(n * L, np.uint8), r_has_zq=np.zeros(n, np.uint8)))
    tile_col = np.cumsum(live) - 1                                  # a live candidate's site in the indel tile
    for k, ci in enumerate(pick):
        want = indeldrv.gap_prep_oracle_site(bb, k)
        assert (want is None) == (not live[ci]), ci
        if want is None:
            continue
        np.testing.assert_array_equal(whole["indel_types"][ci], want["indel_types"])
        assert (whole["maxins"][ci], whole["indelreg"][ci], whole["max_support"][ci]) == (want["maxins"], want["indelreg"], want["max_support"])
        j = int(tile_col[ci])
        np.testing.assert_array_equal(aux[off[j * S]:off[(j + 1) * S]], want["aux"], err_msg="p->aux of candidate %d" % ci)

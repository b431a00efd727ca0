"""GPU parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on the same
seeded inputs.  Integer/byte outputs must be bit-exact; float32 site statistics whose last bits
depend on libm (SGB, MWU, VDB) get a 2e-6 relative tolerance; QUAL 1e-4 (BASELINE.json)."""
import numpy as np
import pytest

from bcftools_amd import abi, synth, host
from tests.helpers import orc

pytestmark = pytest.mark.gpu

EXACT_SITE = ["a", "n_alleles", "unseen", "ori_ref", "shift", "ret", "depth", "ori_depth", "mq0",
              "adf_tot", "adr_tot", "scr_tot", "anno", "qsum"]
FLOAT_SITE = ["vdb", "mwu_pos", "mwu_mq", "mwu_bq", "mwu_mqs", "seg_bias"]


def assert_mplp_equal(got, want, float_rtol=2e-6):
    for k in EXACT_SITE:
        np.testing.assert_array_equal(got.site[k], want.site[k], err_msg="site." + k)
    for k in FLOAT_SITE:
        g, w = got.site[k].astype(np.float64), want.site[k].astype(np.float64)
        assert np.array_equal(np.isinf(g), np.isinf(w)), k
        m = ~np.isinf(w)
        np.testing.assert_allclose(g[m], w[m], rtol=float_rtol, atol=1e-30, err_msg="site." + k)
    for k in ["pl", "dp4", "adf", "adr", "qs", "scr", "sp"]:
        np.testing.assert_array_equal(getattr(got, k), getattr(want, k), err_msg=k)


def assert_call_equal(got, want, n_smpl, qual_tol=1e-4):
    for k in ["ret", "nals_new", "als_new", "als_map", "ac", "an", "qual_missing", "pl_dropped", "has_i16", "dp4", "mq", "pv4_tested"]:
        np.testing.assert_array_equal(got.site[k], want.site[k], err_msg="call." + k)
    np.testing.assert_allclose(got.site["pv4"], want.site["pv4"], rtol=2e-6, atol=1e-30, err_msg="call.pv4")
    np.testing.assert_allclose(got.site["qual"], want.site["qual"], rtol=qual_tol, atol=qual_tol)
    live = want.site["ret"] > 0
    np.testing.assert_array_equal(got.gt[live], want.gt[live], err_msg="gt")
    keep = live & (want.site["pl_dropped"] == 0)
    for i in np.nonzero(keep)[0]:
        nn = int(want.site["nals_new"][i])
        ng = nn * (nn + 1) // 2
        g, w = got.pl[i, :ng].copy(), want.pl[i, :ng].copy()
        # values behind a vector_end are never written out (haploid samples): not part of the contract
        after = np.cumsum(w == abi.INT32_VECTOR_END, axis=0) - (w == abi.INT32_VECTOR_END) > 0
        g[after] = 0
        w[after] = 0
        np.testing.assert_array_equal(g, w, err_msg="pl site %d" % i)


@pytest.mark.parametrize("n_sites,n_smpl,depth,var_rate,seed", [
    (64, 100, 30.0, 0.05, 20260102),      # config[1]-shaped: 100 samples x 30x
    (16, 1000, 30.0, 0.10, 20260104),     # config[3]-shaped: 1000 samples x 30x
    (200, 3, 12.0, 0.20, 7),              # few samples: global-atomic histogram path
    (300, 1, 40.0, 0.20, 8),              # single-sample calling
    (50, 37, 5.0, 0.30, 9),               # shallow, many empty cells
    (8, 300, 150.0, 0.20, 10),            # deep cells (up to 200 reads)
])
def test_mpileup_matches_oracle(gpu_ctx_factory, n_sites, n_smpl, depth, var_rate, seed):
    fmt = abi.INFO_VDB | abi.INFO_RPB | abi.FMT_AD | abi.FMT_QS | abi.FMT_SCR | abi.INFO_SCR | abi.FMT_SP
    tile = synth.numpy_tile(seed, n_sites, n_smpl, depth=depth, var_rate=var_rate, ref_n_rate=0.02, mapq255_rate=0.01)
    cfg = abi.default_cfg(n_smpl, max_sites=n_sites, max_reads=len(tile.rd), fmt_flag=fmt)
    want = orc.mpileup(cfg, tile)
    ctx = gpu_ctx_factory(cfg)
    got = ctx.mpileup(tile)
    assert_mplp_equal(got, want)
    if var_rate >= 0.1 and depth >= 12:
        assert (want.sp > 0).any()          # FMT/SP: some heterozygous cells went through the Fisher exact test


@pytest.mark.parametrize("n_sites,n_smpl,depth,var_rate,seed", [
    (40, 64, 30.0, 0.3, 31),              # ~25 distinct (quality, strand) keys per cell: slot refills in the errmod walk
    (12, 100, 120.0, 0.5, 32),            # deep cells, up to ~100 distinct keys, het cells with many non-reference reads
    (60, 7, 60.0, 0.5, 33),
])
def test_mpileup_unbinned_qualities(gpu_ctx_factory, n_sites, n_smpl, depth, var_rate, seed):
    fmt = abi.INFO_VDB | abi.INFO_RPB | abi.FMT_AD | abi.FMT_QS | abi.FMT_SCR | abi.INFO_SCR
    tile = synth.numpy_tile(seed, n_sites, n_smpl, depth=depth, var_rate=var_rate, ref_n_rate=0.05, wide_qual=True)
    cfg = abi.default_cfg(n_smpl, max_sites=n_sites, max_reads=len(tile.rd), fmt_flag=fmt)
    want = orc.mpileup(cfg, tile)
    ctx = gpu_ctx_factory(cfg)
    got = ctx.mpileup(tile)
    assert_mplp_equal(got, want)


@pytest.mark.parametrize("n_sites,n_smpl,seed,flags,tags", [
    (64, 100, 11, 0, abi.CALL_FMT_PV4),
    (64, 100, 12, abi.CALL_VARONLY, 0),
    (32, 1000, 13, 0, abi.CALL_FMT_GQ | abi.CALL_FMT_GP),
    (100, 5, 14, abi.CALL_KEEPALT, abi.CALL_FMT_GQ),
])
def test_pipeline_matches_oracle(gpu_ctx_factory, n_sites, n_smpl, seed, flags, tags):
    tile = synth.numpy_tile(seed, n_sites, n_smpl, depth=25.0, var_rate=0.3)
    cfg = abi.default_cfg(n_smpl, max_sites=n_sites, max_reads=len(tile.rd), call_flag=flags, output_tags=tags)
    rng = np.random.default_rng(seed)
    ploidy = rng.choice([1, 2, 2, 2], size=n_smpl).astype(np.uint8)
    mwant = orc.mpileup(cfg, tile)
    cin = host.CallInput(n_smpl, mwant.site["n_alleles"], np.maximum(mwant.site["unseen"], 0),
                         mwant.pl.astype(np.int32), mwant.site["qsum"], ploidy=ploidy, i16=mwant.site["anno"].astype(np.float32))
    cwant = orc.mcall(cfg, cin)
    ctx = gpu_ctx_factory(cfg)
    mgot, cgot = ctx.pipeline(tile, ploidy=ploidy)
    assert_mplp_equal(mgot, mwant)
    assert_call_equal(cgot, cwant, n_smpl)
    if tags & abi.CALL_FMT_GQ:
        live = (cwant.site["ret"] > 0) & (cwant.site["als_new"] != 1)
        np.testing.assert_array_equal(cgot.gq[live], cwant.gq[live])


@pytest.mark.parametrize("n_sites,n_smpl,depth,seed,flags,tags", [
    (64, 1000, 30.0, 21, 0, abi.CALL_FMT_PV4),          # config[3]-shaped cohort; PV4 from the mpileup stage's I16
    (64, 999, 8.0, 22, 0, abi.CALL_FMT_GQ),             # sample count not a multiple of 4 (ragged plane tail)
    (128, 37, 0.7, 23, 0, 0),                           # most samples carry no reads at a site
    (128, 130, 3.0, 24, abi.CALL_VARONLY, 0),
    (200, 1, 20.0, 25, 0, abi.CALL_FMT_GQ | abi.CALL_FMT_GP),
    (100, 17, 12.0, 26, abi.CALL_KEEPALT, abi.CALL_FMT_PV4),
    (96, 300, 0.5, 27, 0, abi.CALL_FMT_GQ),             # many samples, few reads: sites of two or three alleles in the 15-subset instantiation
])
def test_pipeline_diploid_matches_oracle(gpu_ctx_factory, n_sites, n_smpl, depth, seed, flags, tags):
    """All-diploid, single-group calling from the mpileup stage's u8 PL planes: the allele-subset scan runs on
    the f64 matrix cores (mcall.hip, FAST instantiations)."""
    tile = synth.numpy_tile(seed, n_sites, n_smpl, depth=depth, var_rate=0.3)
    cfg = abi.default_cfg(n_smpl, max_sites=n_sites, max_reads=len(tile.rd), call_flag=flags, output_tags=tags)
    mwant = orc.mpileup(cfg, tile)
    cin = host.CallInput(n_smpl, mwant.site["n_alleles"], np.maximum(mwant.site["unseen"], 0),
                         mwant.pl.astype(np.int32), mwant.site["qsum"], i16=mwant.site["anno"].astype(np.float32))
    cwant = orc.mcall(cfg, cin)
    ctx = gpu_ctx_factory(cfg)
    mgot, cgot = ctx.pipeline(tile)
    assert_mplp_equal(mgot, mwant)
    assert_call_equal(cgot, cwant, n_smpl)
    if tags & abi.CALL_FMT_GQ:
        live = (cwant.site["ret"] > 0) & (cwant.site["als_new"] != 1)
        np.testing.assert_array_equal(cgot.gq[live], cwant.gq[live])


@pytest.mark.parametrize("n_grp", [1, 3])
@pytest.mark.parametrize("with_ploidy", [False, True])
@pytest.mark.parametrize("n_sites,n_smpl,depth,seed", [(96, 40, 0.7, 41), (64, 260, 1.5, 42), (48, 1000, 2.0, 43), (96, 42, 0.7, 44), (16, 4100, 0.3, 45)])
def test_pipeline_ref_only_sites(gpu_ctx_factory, n_sites, n_smpl, depth, seed, with_ploidy, n_grp):
    """Sites that stay REF-only: GT is 0/0 (0 for a haploid sample) where a sample has data and ./. (.) where it has none or
    its ploidy is 0 (mcall_set_ref_genotypes, mcall.c:529-541).  Sample counts that are multiples of four take the kernel's
    four-samples-per-lane path -- from the subset scan's notes with one group, from the PL planes with several -- and 42 the
    general loop; 260 and 1000 need more than one round of 256 samples; past 4096 samples the scan keeps no notes and the path reads
    the PL planes, four samples a load."""
    tile = synth.numpy_tile(seed, n_sites, n_smpl, depth=depth, var_rate=0.0)
    kw = dict(fmt_flag=abi.INFO_VDB | abi.INFO_RPB | abi.FMT_AD, n_grp=n_grp) if n_grp > 1 else {}
    cfg = abi.default_cfg(n_smpl, max_sites=n_sites, max_reads=len(tile.rd), **kw)
    rng = np.random.default_rng(seed)
    ploidy = rng.choice([0, 1, 2, 2], size=n_smpl).astype(np.uint8) if with_ploidy else None
    grp = rng.integers(0, n_grp, n_smpl).astype(np.int32) if n_grp > 1 else None
    mwant = orc.mpileup(cfg, tile)
    na = mwant.site["n_alleles"]
    ad = None
    if n_grp > 1:
        src = mwant.adf.astype(np.int32) + mwant.adr.astype(np.int32)
        ad = np.where(np.arange(5)[None, :, None] < na[:, None, None], src, abi.INT32_VECTOR_END).astype(np.int32)
    cin = host.CallInput(n_smpl, na, np.maximum(mwant.site["unseen"], 0), mwant.pl.astype(np.int32), mwant.site["qsum"],
                         ad=ad, ploidy=ploidy, grp=grp, i16=mwant.site["anno"].astype(np.float32))
    cwant = orc.mcall(cfg, cin)
    ref_only = (cwant.site["ret"] > 0) & (cwant.site["als_new"] == 1)
    assert ref_only.sum() >= 8
    g = cwant.gt[ref_only]
    assert (g == -1).any() and (g == 0).any() and (not with_ploidy or (g == -2).any())
    mgot, cgot = gpu_ctx_factory(cfg).pipeline(tile, ploidy=ploidy, grp=grp)
    assert_mplp_equal(mgot, mwant)
    assert_call_equal(cgot, cwant, n_smpl)
    np.testing.assert_array_equal(cgot.gt[ref_only], g)
    np.testing.assert_array_equal(cgot.site["ac"][ref_only], cwant.site["ac"][ref_only])


@pytest.mark.parametrize("theta", [0.0, 1e-2, 1e-4, 0.9])
@pytest.mark.parametrize("n_smpl,with_ploidy", [(100, False), (40, True)])
def test_pipeline_nondefault_prior(gpu_ctx_factory, theta, n_smpl, with_ploidy):
    """call -P / --prior (mcall.c:397-416, the reference-site QUAL of :1639-1644): no prior at all (-P 0), a larger and a
    smaller mutation rate than the default 1.1e-3, and one large enough for the 0.99 cap (BASELINE configs[4] names --prior)."""
    n_sites, seed = 96, 71
    tile = synth.numpy_tile(seed, n_sites, n_smpl, depth=15.0, var_rate=0.3)
    cfg = abi.default_cfg(n_smpl, max_sites=n_sites, max_reads=len(tile.rd), call_theta=theta, output_tags=abi.CALL_FMT_GQ)
    ploidy = np.random.default_rng(seed).choice([1, 2, 2], size=n_smpl).astype(np.uint8) if with_ploidy else None
    mwant = orc.mpileup(cfg, tile)
    cin = host.CallInput(n_smpl, mwant.site["n_alleles"], np.maximum(mwant.site["unseen"], 0),
                         mwant.pl.astype(np.int32), mwant.site["qsum"], ploidy=ploidy, i16=mwant.site["anno"].astype(np.float32))
    cwant = orc.mcall(cfg, cin)
    mgot, cgot = gpu_ctx_factory(cfg).pipeline(tile, ploidy=ploidy)
    assert_mplp_equal(mgot, mwant)
    assert_call_equal(cgot, cwant, n_smpl)
    # the prior moves calls: the test would be vacuous if every theta gave the default's records
    if theta != 1e-4:
        cfg0 = abi.default_cfg(n_smpl, max_sites=n_sites, max_reads=len(tile.rd), output_tags=abi.CALL_FMT_GQ)
        c0 = orc.mcall(cfg0, cin)
        assert not np.allclose(c0.site["qual"], cwant.site["qual"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("n_sites,n_smpl,n_grp,seed,use_qs,runs", [
    (48, 120, 4, 61, False, False), (32, 1000, 4, 62, False, False), (40, 33, 33, 63, False, False), (48, 90, 3, 64, True, False),
    # groups as runs of consecutive samples (the usual -G file): their sums run side by side from the fractions scratch;
    # 1001 samples: a ragged last four; 12 groups: the most the 64 lanes take; 13: the general path again; an empty group
    (32, 1000, 4, 65, False, True), (40, 1001, 5, 66, False, True), (24, 97, 12, 67, True, True), (24, 97, 13, 68, False, True),
    (30, 50, 6, 69, False, "gap")])
def test_pipeline_with_sample_groups_matches_oracle(gpu_ctx_factory, n_sites, n_smpl, n_grp, seed, use_qs, runs):
    """call -G through the fused pipeline (BASELINE configs[4] shape): group allele frequencies from the mpileup stage's
    ADF+ADR planes (= FORMAT/AD, bam2bcf.c:892-896) or its QS planes, ploidy array, one group per sample at the extreme."""
    tile = synth.numpy_tile(seed, n_sites, n_smpl, depth=20.0, var_rate=0.4)
    fmt = abi.INFO_VDB | abi.INFO_RPB | (abi.FMT_QS if use_qs else abi.FMT_AD)
    cfg = abi.default_cfg(n_smpl, max_sites=n_sites, max_reads=len(tile.rd), fmt_flag=fmt, n_grp=n_grp)
    cfg.grp_tag_is_qs = 1 if use_qs else 0
    rng = np.random.default_rng(seed)
    ploidy = rng.choice([1, 2, 2], size=n_smpl).astype(np.uint8)
    grp = (np.arange(n_smpl) * n_grp // n_smpl).astype(np.int32)
    if runs == "gap":
        grp[grp == 2] = 3                                      # a group without samples between the runs
    elif n_grp < n_smpl and not runs:
        rng.shuffle(grp)                                       # groups need not be contiguous
    mwant = orc.mpileup(cfg, tile)
    src = mwant.qs.astype(np.int32) if use_qs else mwant.adf.astype(np.int32) + mwant.adr.astype(np.int32)
    na = mwant.site["n_alleles"]
    ad = np.where(np.arange(5)[None, :, None] < na[:, None, None], src, abi.INT32_VECTOR_END).astype(np.int32)
    cin = host.CallInput(n_smpl, na, np.maximum(mwant.site["unseen"], 0), mwant.pl.astype(np.int32), mwant.site["qsum"],
                         ad=ad, ploidy=ploidy, grp=grp, i16=mwant.site["anno"].astype(np.float32))
    cwant = orc.mcall(cfg, cin)
    mgot, cgot = gpu_ctx_factory(cfg).pipeline(tile, ploidy=ploidy, grp=grp)
    assert_mplp_equal(mgot, mwant)
    assert_call_equal(cgot, cwant, n_smpl)


@pytest.mark.parametrize("n_sites,n_smpl,n_grp,depth,ploidies,seed", [
    (400, 40, 8, 14.0, None, 71),        # five samples a group: some groups lack an allele that others (and later alleles) show
    (48, 90, 3, 12.0, None, 72),         # groups without a ploidy array, variant sites
    (48, 203, 5, 10.0, [0, 1, 2, 2], 73),    # ploidy 0 among variant sites, a sample count not divisible by four, shuffled groups
    (16, 1003, 1, 8.0, None, 74),        # one group, the last word of three samples
    (12, 5000, 1, 6.0, [1, 2, 2], 75),   # past the 4096 samples the scan keeps data-presence notes for
    (12, 4104, 4, 6.0, [0, 1, 2], 76),
    (60, 300, 1, 12.0, None, 77),        # N in the reference at a third of the sites: five alleles, the reference allele without a frequency
    (60, 260, 3, 12.0, [1, 2, 2], 78)])
def test_subset_scan_corners(gpu_ctx_factory, n_sites, n_smpl, n_grp, depth, ploidies, seed):
    """The lane-per-sample subset scan of mcall_find_best_alleles (csrc/mcall.hip, sparse_scan): the alleles with a frequency are
    visited first by permuting the PL planes, so a group whose frequency is zero for an allele BETWEEN two it has must pick the
    reference's subsets (mcall.c:617-698 skips qsum == 0) in the reference's order; samples of ploidy 0 enter the single-allele
    rows only; ragged sample counts; more samples than the scan notes data presence for."""
    tile = synth.numpy_tile(seed, n_sites, n_smpl, depth=depth, var_rate=0.6, ref_n_rate=0.35 if seed >= 77 else 0.0)
    fmt = abi.INFO_VDB | abi.INFO_RPB | (abi.FMT_AD if n_grp > 1 else 0)
    kw = dict(n_grp=n_grp) if n_grp > 1 else {}
    cfg = abi.default_cfg(n_smpl, max_sites=n_sites, max_reads=len(tile.rd), fmt_flag=fmt, **kw)
    rng = np.random.default_rng(seed)
    ploidy = rng.choice(ploidies, size=n_smpl).astype(np.uint8) if ploidies else None
    grp = None
    if n_grp > 1:
        grp = (np.arange(n_smpl) * n_grp // n_smpl).astype(np.int32)
        if seed % 2: rng.shuffle(grp)
    mwant = orc.mpileup(cfg, tile)
    na = mwant.site["n_alleles"]
    ad = None
    if n_grp > 1:
        src = mwant.adf.astype(np.int32) + mwant.adr.astype(np.int32)
        ad = np.where(np.arange(5)[None, :, None] < na[:, None, None], src, abi.INT32_VECTOR_END).astype(np.int32)
        if seed == 71:
            # the case the test is for: some (site, group) has no read of allele j but reads of an allele after it
            gsum = np.stack([np.where(ad >= 0, ad, 0)[:, :, grp == g].sum(axis=2) for g in range(n_grp)], axis=1)   # [site][group][allele]
            hole = (gsum[:, :, 1:-1] == 0) & (np.cumsum(gsum[:, :, ::-1], axis=2)[:, :, ::-1][:, :, 2:] > 0)
            assert hole.sum() >= 5
    cin = host.CallInput(n_smpl, na, np.maximum(mwant.site["unseen"], 0), mwant.pl.astype(np.int32), mwant.site["qsum"],
                         ad=ad, ploidy=ploidy, grp=grp, i16=mwant.site["anno"].astype(np.float32))
    cwant = orc.mcall(cfg, cin)
    mgot, cgot = gpu_ctx_factory(cfg).pipeline(tile, ploidy=ploidy, grp=grp)
    assert_mplp_equal(mgot, mwant)
    assert_call_equal(cgot, cwant, n_smpl)
    assert (cwant.site["nals_new"] > 1).any()
    if seed >= 77:
        assert (na == 5).any()


def test_empty_and_zero_depth(gpu_ctx_factory):
    n_smpl = 4
    cfg = abi.default_cfg(n_smpl, max_sites=8, max_reads=64)
    ctx = gpu_ctx_factory(cfg)
    # a tile whose cells are all empty: bcf_call_glfgen returns -1 for every sample (bam2bcf.c:162)
    tile = host.HostTile(n_smpl, np.array([1, 2, 4], dtype=np.int8), np.zeros(3 * n_smpl + 1, dtype=np.uint32),
                         np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=np.uint8))
    want = orc.mpileup(cfg, tile)
    got = ctx.mpileup(tile)
    assert_mplp_equal(got, want)
    # zero sites: a no-op
    empty = host.HostTile(n_smpl, np.zeros(0, dtype=np.int8), np.zeros(1, dtype=np.uint32),
                          np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=np.uint8))
    res = ctx.mpileup(empty)
    assert res.site.shape == (0,)


@pytest.mark.parametrize("n_smpl,depths,seed", [(2, [420, 0], 31), (40, None, 32), (3, [345, 255, 900], 33), (5, [256, 5000, 257, 1300, 2048], 34)])
def test_cells_over_255_count_every_read(gpu_ctx_factory, n_smpl, depths, seed):
    """A cell with more than 255 usable reads (256 ... 5000 here) is not an error and loses nothing: DP4, AD/ADF/ADR, QS, SCR, SP,
    the site's I16 sums, depth and the bias-test histograms are over ALL of its reads, as bcf_call_glfgen has them
    (bam2bcf.c:203-252); only errmod_cal's input is cut to 255 reads (bam2bcf.c:256).  The oracle is fed the uncut pileup."""
    if depths is None:
        tile = synth.numpy_tile(seed, 6, n_smpl, depth=240.0, var_rate=0.5, max_depth=400)
    else:
        rng = np.random.default_rng(seed)
        R = int(sum(depths))
        rd = (rng.choice([11, 25, 37, 40], R) | (rng.choice([0, 20, 60, 60, 60], R) << 8) | ((1 << rng.integers(0, 4, R)) << 16)
              | (rng.integers(0, 2, R) << 20) | (rng.integers(0, 2, R) << 21) | (rng.integers(0, 40, R) << 24)).astype(np.uint32)
        tile = host.HostTile(n_smpl, np.array([1], dtype=np.int8), np.r_[0, np.cumsum(depths)].astype(np.uint32), rd,
                             rng.integers(0, 100, R).astype(np.uint8))
    flags = abi.INFO_VDB | abi.INFO_RPB | abi.FMT_AD | abi.FMT_QS | abi.FMT_SCR | abi.INFO_SCR | abi.FMT_SP | abi.FMT_DP4
    cfg = abi.default_cfg(n_smpl, max_sites=len(tile.ref16), max_reads=len(tile.rd), fmt_flag=flags)
    ctx = gpu_ctx_factory(cfg)
    got = ctx.mpileup(tile)
    want, cr = orc.mpileup(cfg, tile, want_callret=True)
    assert (cr["n"] > 255).any() and int(want.dp4.astype(np.int64).sum(axis=1).max()) > 255
    assert_mplp_equal(got, want)
    n = abi.C.c_uint32()
    assert ctx.L.bcfgpu_truncated_cells(ctx.h, abi.C.byref(n)) == 0 and n.value == int((cr["n"] > 255).sum())
    assert ctx.L.bcfgpu_truncated_cells(ctx.h, abi.C.byref(n)) == 0 and n.value == 0      # reading resets the counter
    # every integer field is what the reference's own rule (the random draw of 255) gives too: only the likelihoods differ
    ref = orc.mpileup(cfg, tile, deep_rule=0)
    for k in ["dp4", "adf", "adr", "qs", "scr", "sp"]:
        np.testing.assert_array_equal(getattr(got, k), getattr(ref, k), err_msg=k)
    for k in ["depth", "ori_depth", "mq0", "adf_tot", "adr_tot", "scr_tot", "anno"]:
        np.testing.assert_array_equal(got.site[k], ref.site[k], err_msg="site." + k)


def test_count_planes_past_65535_are_refused(gpu_ctx_factory):
    """The count planes are 16 bits wide: a cell with more reads of one base and strand than that is reported (BCFGPU_E_DEPTH),
    never wrapped."""
    n = 70000
    rd = np.full(n, 40 | (60 << 8) | (1 << 16) | (20 << 24), np.uint32)
    tile = host.HostTile(1, np.array([1], dtype=np.int8), np.array([0, n], np.uint32), rd, np.zeros(n, np.uint8))
    ctx = gpu_ctx_factory(abi.default_cfg(1, max_sites=1, max_reads=n))
    with pytest.raises(Exception) as e:
        ctx.mpileup(tile)
    assert "65535" in str(e.value) or "-4" in str(e.value)


def _lone_deep_cells(n_smpl, depths, seed, usable_frac=0.015, n_sites=1):
    """A tile of `n_sites` x n_smpl cells with the given entries per cell (low base qualities: thousands of pileup entries,
    few usable reads)."""
    rng = np.random.default_rng(seed)
    R = int(np.sum(depths))
    bq = np.where(rng.random(R) < usable_frac, rng.choice([25, 37, 40], R), 5)
    rd = (bq | (rng.choice([0, 20, 60, 60, 60], R) << 8) | ((1 << rng.integers(0, 4, R)) << 16) | (rng.integers(0, 2, R) << 20)
          | (rng.integers(0, 40, R) << 24)).astype(np.uint32)
    return host.HostTile(n_smpl, rng.choice([1, 2, 4, 8], n_sites).astype(np.int8), np.r_[0, np.cumsum(depths)].astype(np.uint32), rd,
                         rng.integers(0, 100, R).astype(np.uint8))


def test_cell_at_the_edge_of_the_staging_window(gpu_ctx_factory):
    """Cells that fill a workgroup's LDS key window to the last key and beyond, at every alignment of their first read (the
    window starts at a multiple of four reads): inside the window they are worked on in the tile launch, past it in the launch
    for listed cells -- the same results either way, the oracle's; nothing is refused (the reference never fails here)."""
    n_smpl = 300
    cfg = abi.default_cfg(n_smpl, max_sites=1, max_reads=13000)
    ctx = gpu_ctx_factory(cfg)
    # a shallow tile of 300 samples gets the six-workgroups-per-CU window: 6016 keys (csrc/api.hip), a lone cell stops
    # fitting at 6014 entries; the sweep crosses that edge at every alignment
    for big in range(5990, 6040, 3):
        for lead in (0, 1, 2, 3):
            depths = np.zeros(n_smpl, np.int64)
            depths[0] = lead
            depths[1] = big
            tile = _lone_deep_cells(n_smpl, depths, big * 4 + lead)
            assert_mplp_equal(ctx.mpileup(tile), orc.mpileup(cfg, tile))


@pytest.mark.parametrize("n_smpl,n_sites,where,size,usable,seed", [
    (40, 3, [5], 3000, 0.015, 51),                       # a shallow tile with one cell of a few thousand entries
    (40, 3, [0, 60, 119], 7000, 0.015, 52),              # the first cell of the tile, one in the middle, the last one
    (8, 2, [3, 4, 5], 20000, 0.01, 53),                  # neighbours, each far past the largest window (16384 keys)
    (3, 1, [1], 40000, 0.02, 54),                        # > 255 usable reads inside a listed cell: every read counts, 255 feed the likelihoods
    (300, 1, [7, 250], 9000, 0.05, 55),
])
def test_cells_deeper_than_the_key_window(gpu_ctx_factory, n_smpl, n_sites, where, size, usable, seed):
    """Amplicon-like pile-ups: a few cells with thousands of entries in an otherwise ordinary tile.  They are listed by the tile
    launch and worked on by a workgroup each; the rest of the tile proceeds.  Equal to the oracle on the same (uncut) pileup."""
    rng = np.random.default_rng(seed)
    depths = rng.poisson(12, n_sites * n_smpl).astype(np.int64)
    for k, c in enumerate(where):
        depths[c] = size + 37 * k
    tile = _lone_deep_cells(n_smpl, depths, seed, usable_frac=usable, n_sites=n_sites)
    # the ordinary cells carry ordinary reads
    off = tile.plp_off.astype(np.int64)
    cell = np.repeat(np.arange(n_sites * n_smpl), np.diff(off))
    plain = ~np.isin(cell, where)
    tile.rd[plain] = (tile.rd[plain] & ~np.uint32(0xff)) | rng.choice([11, 25, 37, 40], int(plain.sum())).astype(np.uint32)
    cfg = abi.default_cfg(n_smpl, max_sites=n_sites, max_reads=len(tile.rd), fmt_flag=abi.INFO_VDB | abi.INFO_RPB | abi.FMT_AD)
    ctx = gpu_ctx_factory(cfg)
    got = ctx.mpileup(tile)
    assert_mplp_equal(got, orc.mpileup(cfg, tile))
    got2 = ctx.mpileup(tile)                              # the list and its counters start over with every launch
    assert_mplp_equal(got2, got)


@pytest.mark.parametrize("n_sites,n_smpl,seed", [(40, 100, 21), (64, 7, 22)])
def test_indel_pass_matches_oracle(gpu_ctx_factory, n_sites, n_smpl, seed):
    """glfgen/combine with ref_base=-1 and p->aux set (mpileup.c:357-360): type<<16|seqQ<<8|indelQ as left by
    bcf_call_gap_prep; low indelQ reads fall back to REF (bam2bcf.c:183); sites without ALT support return -1."""
    base = synth.numpy_tile(seed, n_sites, n_smpl, depth=20.0, var_rate=0.0)
    rng = np.random.default_rng(seed)
    R = len(base.rd)
    cell = np.repeat(np.arange(n_sites * n_smpl), np.diff(base.plp_off.astype(np.int64)))
    site = cell // n_smpl
    has_indel = rng.random(n_sites) < 0.7
    carrier = rng.random(n_sites * n_smpl) < 0.3
    typ = np.where(has_indel[site] & carrier[cell] & (rng.random(R) < 0.5), rng.integers(1, 5, R), 0)
    indelQ = rng.integers(0, 80, R)
    seqQ = rng.integers(10, 200, R)
    aux = (typ.astype(np.uint32) << 16) | (seqQ.astype(np.uint32) << 8) | indelQ.astype(np.uint32)
    rd = base.rd | np.where(rng.random(R) < 0.05, abi.RD_DEL, 0).astype(np.uint32)
    tile = host.HostTile(n_smpl, base.ref16, base.plp_off, rd, base.epos, aux=aux, is_indel=1)
    fmt = abi.INFO_VDB | abi.INFO_RPB | abi.FMT_AD
    cfg = abi.default_cfg(n_smpl, max_sites=n_sites, max_reads=R, fmt_flag=fmt)
    want = orc.mpileup(cfg, tile)
    got = gpu_ctx_factory(cfg).mpileup(tile)
    assert (want.site["ret"] == -1).any() and (want.site["ret"] == 0).any()
    dead = want.site["ret"] == -1
    np.testing.assert_array_equal(got.site["ret"], want.site["ret"])
    for k in ("a", "n_alleles", "qsum"):
        np.testing.assert_array_equal(got.site[k], want.site[k], err_msg=k)
    live = ~dead
    for k in EXACT_SITE:
        np.testing.assert_array_equal(got.site[k][live], want.site[k][live], err_msg="site." + k)
    for k in ["pl", "dp4", "adf", "adr"]:
        np.testing.assert_array_equal(getattr(got, k)[live], getattr(want, k)[live], err_msg=k)
    for k in FLOAT_SITE:
        g, w = got.site[k][live].astype(np.float64), want.site[k][live].astype(np.float64)
        assert np.array_equal(np.isinf(g), np.isinf(w)), k
        m = ~np.isinf(w)
        np.testing.assert_allclose(g[m], w[m], rtol=2e-6, atol=1e-30, err_msg=k)


@pytest.mark.parametrize("n_sites,n_smpl,seed,flags,with_ploidy", [(40, 100, 31, 0, False), (64, 7, 32, abi.CALL_VARONLY, False),
                                                                   (48, 260, 33, 0, True)])
def test_indel_records_through_the_fused_caller(gpu_ctx_factory, n_sites, n_smpl, seed, flags, with_ploidy):
    """`mpileup | call -m` for indel records (mpileup.c:357-364 -> vcfcall.c:1137): the indel pass's tile through
    bcfgpu_pipeline -- PLs of the indel types stay in HBM between the two stages -- against the oracle's mpileup followed by
    its mcall on the records mpileup would have written (ret == 0).  An indel record carries no <*> allele."""
    base = synth.numpy_tile(seed, n_sites, n_smpl, depth=20.0, var_rate=0.0)
    rng = np.random.default_rng(seed)
    R = len(base.rd)
    cell = np.repeat(np.arange(n_sites * n_smpl), np.diff(base.plp_off.astype(np.int64)))
    site = cell // n_smpl
    has_indel = rng.random(n_sites) < 0.7
    carrier = rng.random(n_sites * n_smpl) < 0.3
    typ = np.where(has_indel[site] & carrier[cell] & (rng.random(R) < 0.5), rng.integers(1, 4, R), 0)
    aux = (typ.astype(np.uint32) << 16) | (rng.integers(10, 200, R).astype(np.uint32) << 8) | rng.integers(0, 80, R).astype(np.uint32)
    tile = host.HostTile(n_smpl, base.ref16, base.plp_off, base.rd, base.epos, aux=aux, is_indel=1)
    cfg = abi.default_cfg(n_smpl, max_sites=n_sites, max_reads=R, call_flag=flags, output_tags=abi.CALL_FMT_GQ)
    ploidy = rng.choice([1, 2, 2, 2], size=n_smpl).astype(np.uint8) if with_ploidy else None
    mwant = orc.mpileup(cfg, tile)
    live = mwant.site["ret"] == 0
    assert live.any() and (~live).any()
    mgot, cgot = gpu_ctx_factory(cfg).pipeline(tile, ploidy=ploidy)
    np.testing.assert_array_equal(mgot.site["ret"], mwant.site["ret"])
    np.testing.assert_array_equal(mgot.pl[live], mwant.pl[live])
    assert (mwant.site["unseen"][live] < 0).all()
    idx = np.nonzero(live)[0]
    cin = host.CallInput(n_smpl, mwant.site["n_alleles"][idx], np.zeros(len(idx), np.int32), mwant.pl[idx].astype(np.int32),
                         mwant.site["qsum"][idx], ploidy=ploidy, i16=mwant.site["anno"][idx].astype(np.float32))
    cwant = orc.mcall(cfg, cin)
    assert (cgot.site["ret"][~live] == 0).all()               # no record from mpileup: nothing called, no error
    for k in ("ret", "nals_new", "als_new", "an"):
        np.testing.assert_array_equal(cgot.site[k][idx], cwant.site[k], err_msg=k)
    called = cwant.site["ret"] > 0
    assert called.any()
    np.testing.assert_allclose(cgot.site["qual"][idx][called], cwant.site["qual"][called], rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(cgot.site["ac"][idx][called], cwant.site["ac"][called])
    np.testing.assert_array_equal(cgot.gt[idx][called], cwant.gt[called])
    var = called & (cwant.site["als_new"] != 1)
    np.testing.assert_array_equal(cgot.gq[idx][var], cwant.gq[var])


def test_library_and_torch_share_one_hip_runtime():
    """PyTorch ships its own libamdhip64: the binding imports torch before dlopen()ing libbcfgpu.so, so a process can
    create a context first and touch torch's device afterwards (the order the unit tests use when run one file at a time)."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("from bcftools_amd import abi, engine\n"
            "ctx = engine.Context(abi.default_cfg(4, max_sites=8, max_reads=64))\n"
            "import torch\n"
            "x = torch.ones(8, device='cuda').sum().item()\n"
            "ctx.close()\n"
            "print('ok', x)\n")
    p = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True)
    assert p.returncode == 0 and "ok 8.0" in p.stdout, p.stderr[-2000:]

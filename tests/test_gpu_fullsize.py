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

"""bcfgpu_gap_prep on a batch of synthetic indel-candidate columns (many sites in one call, the way a tile is run)
against the oracle's bcf_call_gap_prep one site at a time: p->aux of every pileup entry, the candidate types, the
insertion consensus, indelreg, max_support, max_frac -- all exact."""
import numpy as np
import pytest

from bcftools_amd import abi, synth
from tests.helpers import indeldrv, orc
from tests.test_gpu_parity import EXACT_SITE, FLOAT_SITE

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_sites,n_smpl,depth,seed,kw", [
    (24, 20, 15.0, 41, {}),
    (16, 3, 40.0, 42, dict(min_support=2)),
    (12, 50, 8.0, 43, dict(per_sample_flt=1, min_frac=0.05)),
    (6, 1, 60.0, 44, dict(openQ=30, extQ=10)),
    (4, 500, 30.0, 45, {}),                               # BASELINE configs[2]: 500 samples x 30x
])
def test_batched_gap_prep_matches_oracle(gpu_ctx_factory, n_sites, n_smpl, depth, seed, kw):
    b = synth.indel_batch(seed, n_sites, n_smpl, depth=depth)
    ctx = gpu_ctx_factory(abi.default_cfg(n_smpl, max_sites=n_sites, max_reads=64))
    got, st = indeldrv.gap_prep_gpu(ctx, b, **kw)
    live = 0
    for k in range(n_sites):
        want = indeldrv.gap_prep_oracle_site(b, k, **kw)
        indeldrv.assert_site_equal(got, k, want)
        live += want is not None
    assert live > 0
    assert st.n_jobs > 0 and st.n_passes >= st.n_jobs and st.dp_cells > 0 and st.kernel_ms > 0


LONG = (-40, -25, -12, -8, 8, 12, 25, 40)      # bands of 11..43: past PROBALN_BW_MAX, the row of the pair-HMM in LDS
SHORT = (-3, -2, -1, 1, 2, 3)                  # bands of 4..6: the register-resident classes


@pytest.mark.parametrize("n_sites,n_smpl,depth,seed,lens,lens2,kw", [
    (16, 12, 15.0, 101, LONG, SHORT, {}),                  # a long and a short type in the same column
    (10, 40, 20.0, 102, LONG, None, {}),                   # one long type per column
    (8, 6, 30.0, 103, (-40, 40, -25, 25), SHORT, dict(min_support=2)),
    (12, 25, 12.0, 104, LONG, LONG, dict(per_sample_flt=1, min_frac=0.05)),   # two long types in a column
    (6, 100, 30.0, 105, (-9, -8, 8, 9, 7, -7), SHORT, {}),  # the boundary: |type| = 7 is the widest register class
    (8, 16, 15.0, 106, (-90, -80, -74, -71, -70, 45, 60), SHORT, {}),   # bands past 73: sixteen jobs a wavefront (LDS holds no 64 such rows)
    (5, 300, 30.0, 107, (10, -10, 15, -15, 47, -47), (8, -8, 2), {}),   # many jobs per band class: whole wavefronts, several launches' worth
    (3, 1000, 30.0, 108, (-25, 12, 40), SHORT, {}),                     # BASELINE configs[3] shape: 1000 samples x 30x
    (4, 8, 12.0, 109, (-320, -305), (-2, 1, 3), dict(spacing=900)),     # bands past 300: the rolling rows in global scratch (probaln_wide_kernel)
])
def test_long_indels_through_the_wide_band_kernel(gpu_ctx_factory, n_sites, n_smpl, depth, seed, lens, lens2, kw):
    """Indels of 8 bp and more: bam2bcf_indel.c:293-294 gives their realignment the band |type| + 3, wider than the widest
    register-resident class, so probaln_glocal (:346, :352) runs in probaln_lds_kernel<64> (bands up to 73) or <16> (up to 300) -- for every read of the column, against that type.  The same column's short types stay in the register classes.  Every output of bcf_call_gap_prep
    (p->aux, types, inscns, indelreg, max_support, max_frac) against the oracle, through both entry points, on the product
    build (no environment switch)."""
    kw = dict(kw)
    spacing = kw.pop("spacing", 300)
    b = synth.indel_batch(seed, n_sites, n_smpl, depth=depth, lens=lens, lens2=lens2, spacing=spacing)
    assert (np.abs(b["itype"]) >= 7).any()
    ctx = gpu_ctx_factory(abi.default_cfg(n_smpl, max_sites=n_sites, max_reads=len(b["p_read"]) + 64))
    got, st = indeldrv.gap_prep_gpu(ctx, b, **kw)
    live = long_seen = 0
    for k in range(n_sites):
        want = indeldrv.gap_prep_oracle_site(b, k, **kw)
        indeldrv.assert_site_equal(got, k, want)
        if want is not None:
            live += 1
            t = want["indel_types"]
            long_seen += bool(((np.abs(t) >= 8) & (t != 10000)).any())
    assert live > 0 and long_seen > 0
    assert 0 < st.n_wide < st.n_jobs, (st.n_wide, st.n_jobs)      # wide-band jobs and register-class jobs in one call
    assert (st.n_scratch > 0) == (max(abs(x) for x in lens) > 297)      # bands past 300 only: rolling rows in global scratch
    assert st.n_passes >= st.n_jobs // 2 and st.dp_cells > 0
    # the device-pool form: the same core fed from bcfgpu_pileup's pool
    pool = indeldrv.DevicePool(ctx, b)
    got_t, st_t, tile = pool.gap_prep_tile(want_aux=True, **kw)
    indeldrv.assert_tile_matches_host_batch(ctx, b, pool, got_t, tile, got)
    assert st_t.n_wide == st.n_wide


@pytest.mark.parametrize("n_sites,n_smpl,depth,seed,bkw", [(24, 40, 15.0, 51, {}), (10, 200, 25.0, 52, {}),
                                                           (12, 30, 20.0, 53, dict(lens=LONG, lens2=SHORT))])
def test_indel_records_match_oracle(gpu_ctx_factory, n_sites, n_smpl, depth, seed, bkw):
    """The whole indel record path on synthetic columns: bcfgpu_gap_prep -> p->aux -> the indel pass of
    glfgen/combine (ref_base = -1), against the oracle running the same two steps."""
    b = synth.indel_batch(seed, n_sites, n_smpl, depth=depth, **bkw)
    fmt = abi.INFO_VDB | abi.INFO_RPB | abi.FMT_AD
    gctx = gpu_ctx_factory(abi.default_cfg(n_smpl, max_sites=n_sites, max_reads=64))
    got, _ = indeldrv.gap_prep_gpu(gctx, b)
    aux_w, ret_w = np.zeros_like(got["aux"]), np.full(n_sites, -1, np.int32)
    for k in range(n_sites):
        w = indeldrv.gap_prep_oracle_site(b, k)
        if w is not None:
            aux_w[w["e0"]:w["e1"]] = w["aux"]
            ret_w[k] = 0
    np.testing.assert_array_equal(got["ret"], ret_w)
    tile_g, live = synth.indel_tile_from_batch(b, got["aux"], got["ret"])
    tile_w, _ = synth.indel_tile_from_batch(b, aux_w, ret_w)
    np.testing.assert_array_equal(tile_g.aux, tile_w.aux)
    assert len(live) > 0
    cfg = abi.default_cfg(n_smpl, max_sites=len(live), max_reads=len(tile_g.rd), fmt_flag=fmt)
    want = orc.mpileup(cfg, tile_w)
    res = gpu_ctx_factory(cfg).mpileup(tile_g)
    np.testing.assert_array_equal(res.site["ret"], want.site["ret"])
    ok = want.site["ret"] == 0
    assert ok.any()
    for k in EXACT_SITE:
        np.testing.assert_array_equal(res.site[k][ok], want.site[k][ok], err_msg="site." + k)
    for k in ["pl", "dp4", "adf", "adr"]:
        np.testing.assert_array_equal(getattr(res, k)[ok], getattr(want, k)[ok], err_msg=k)
    for k in FLOAT_SITE:
        g, w = res.site[k][ok].astype(np.float64), want.site[k][ok].astype(np.float64)
        assert np.array_equal(np.isinf(g), np.isinf(w)), k
        m = ~np.isinf(w)
        np.testing.assert_allclose(g[m], w[m], rtol=2e-6, atol=1e-30, err_msg="site." + k)


def test_two_callers_with_their_own_contexts(gpu_ctx_factory):
    """Contexts are independent: two caller threads, one context each, run bcfgpu_gap_prep at the same time (how a host
    program overlaps the typing of one batch with the scoring of another) and get what a single caller gets."""
    import threading
    n_smpl = 30
    batches = [synth.indel_batch(70 + j, 16, n_smpl, depth=12.0) for j in range(6)]
    ctxs = [gpu_ctx_factory(abi.default_cfg(n_smpl, max_sites=16, max_reads=64)) for _ in range(2)]
    want = [indeldrv.gap_prep_gpu(ctxs[0], b)[0] for b in batches]
    got = [None] * len(batches)
    errs = []

    def run(k):
        try:
            for _ in range(3):
                for j in range(k, len(batches), 2):
                    got[j] = indeldrv.gap_prep_gpu(ctxs[k], batches[j])[0]
        except Exception as e:
            errs.append(e)
    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for g, w in zip(got, want):
        for key in w:
            np.testing.assert_array_equal(g[key], w[key], err_msg=key)


@pytest.mark.parametrize("seed,n_sites,n_smpl,depth", [(91, 12, 20, 15.0), (92, 5, 64, 30.0)])
def test_gap_prep_tile_on_a_device_pool_matches_the_host_batch(gpu_ctx_factory, seed, n_sites, n_smpl, depth):
    """bcfgpu_gap_prep_tile (reads and entries never leave HBM after bcfgpu_pileup) against bcfgpu_gap_prep on the same
    batch with host pointers (itself checked against the oracle above).  The pool is position-sorted per sample, so the
    entries of a column come in another order than the batch lists them: p->aux is compared read by read."""
    b = synth.indel_batch(seed, n_sites, n_smpl, depth=depth)
    ctx = gpu_ctx_factory(abi.default_cfg(n_smpl, max_sites=n_sites, max_reads=len(b["p_read"]) + 64))
    want, _ = indeldrv.gap_prep_gpu(ctx, b)
    pool = indeldrv.DevicePool(ctx, b)
    assert np.array_equal(pool.col_n[pool.cols], np.diff(b["smpl_off"][::n_smpl]))
    for rep in range(2):                      # the second call runs on workspaces the first one left behind
        got, st, tile = pool.gap_prep_tile(want_aux=True)
        indeldrv.assert_tile_matches_host_batch(ctx, b, pool, got, tile, want)
        assert (want["ret"] == 0).any()
        assert st.n_jobs > 0 and st.n_passes >= st.n_jobs // 2

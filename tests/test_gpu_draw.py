"""errmod_cal's random draw for cells of more than 255 usable reads, replayed on the device (bcfgpu_errmod_plan, csrc/draw.hip):
hts_drand48 is one generator per process and mpileup_reg() visits, position by position, the samples of the SNP pass and then
the samples of the indel pass where bcf_call_gap_prep returned >= 0 (mpileup.c:343-360).  The oracle (oracle/errmod.c, rule 0)
is run in exactly that order, one site at a time, its generator going on from call to call; the device ranks the deep cells of
both passes, jumps the generator to each cell's place and must give the same PLs, bit for bit.  PARITY UNPINNED at the root
(both sides restate htslib's published hts_drand48 / ks_shuffle; tests/test_oracle_errmod_deep.py checks the oracle's
against libc)."""
import numpy as np
import pytest

from bcftools_amd import abi, host
from tests.helpers import orc
from tests.test_gpu_parity import assert_mplp_equal

pytestmark = pytest.mark.gpu


def _tile(rng, n_sites, n_smpl, depths, is_indel=False):
    R = int(np.sum(depths))
    rd = (rng.choice([11, 25, 37, 40], R) | (rng.choice([0, 20, 60, 60, 60], R) << 8) | ((1 << rng.integers(0, 4, R)) << 16)
          | (rng.integers(0, 2, R) << 20) | (rng.integers(0, 40, R) << 24)).astype(np.uint32)
    aux = None
    if is_indel:
        aux = (rng.choice([0, 0, 0, 1, 2], R).astype(np.uint32) << 16) | (rng.integers(20, 60, R).astype(np.uint32) << 8) | rng.integers(5, 60, R).astype(np.uint32)
    return host.HostTile(n_smpl, rng.choice([1, 2, 4, 8], n_sites).astype(np.int8), np.r_[0, np.cumsum(depths)].astype(np.uint32), rd,
                         rng.integers(0, 100, R).astype(np.uint8), aux=aux, is_indel=1 if is_indel else 0)


def _oracle_in_visit_order(cfg, snp, indel, cols, ret):
    """The reference's loop: per position the SNP pass, then the indel pass of that position if it runs; one generator."""
    want_s, want_i = host.MplpResult(snp.n_sites, snp.n_smpl), (host.MplpResult(indel.n_sites, indel.n_smpl) if indel is not None else None)
    col_of = {int(c): i for i, c in enumerate(cols)} if indel is not None else {}
    first = True
    for k in range(snp.n_sites):
        r = orc.mpileup(cfg, snp.select_sites([k]), deep_rule=0, reset=first)
        first = False
        for name in ("site", "pl", "dp4", "adf", "adr", "qs", "scr", "sp"):
            getattr(want_s, name)[k] = getattr(r, name)[0]
        if k in col_of and ret[col_of[k]] == 0:
            i = col_of[k]
            r = orc.mpileup(cfg, indel.select_sites([i]), deep_rule=0, reset=False)
            for name in ("site", "pl", "dp4", "adf", "adr", "qs", "scr", "sp"):
                getattr(want_i, name)[i] = getattr(r, name)[0]
    return want_s, want_i, int(orc.lib().orc_rand48_state())


@pytest.mark.parametrize("seed", [3, 4])
def test_draw_of_both_passes_in_visit_order(gpu_ctx_factory, seed):
    rng = np.random.default_rng(seed)
    n_smpl, n_sites = 3, 6
    ds = rng.poisson(40, n_sites * n_smpl)
    for c in (1, 5, 6, 11, 16):                                  # deep cells scattered over the SNP pass
        ds[c] = rng.integers(300, 1500)
    snp = _tile(rng, n_sites, n_smpl, ds)
    cols = np.array([1, 3, 4], np.int32)
    ret = np.array([0, -1, 0], np.int32)                         # the indel pass runs at columns 1 and 4 only
    di = rng.poisson(40, len(cols) * n_smpl)
    for c in (0, 2, 4, 7):                                       # deep cells in the indel pass too, one of them (4) at the column without a pass
        di[c] = rng.integers(300, 900)
    indel = _tile(rng, len(cols), n_smpl, di, is_indel=True)
    flags = abi.INFO_VDB | abi.INFO_RPB | abi.FMT_AD | abi.FMT_QS
    cfg = abi.default_cfg(n_smpl, max_sites=n_sites, max_reads=max(len(snp.rd), len(indel.rd)), fmt_flag=flags)
    ctx = gpu_ctx_factory(cfg)
    want_s, want_i, state = _oracle_in_visit_order(cfg, snp, indel, cols, ret)
    got_s, got_i = ctx.mpileup_planned(snp, indel, cols, ret)
    assert_mplp_equal(got_s, want_s)
    live = ret == 0
    for k in ("pl", "dp4", "adf", "adr", "qs"):
        np.testing.assert_array_equal(getattr(got_i, k)[live], getattr(want_i, k)[live], err_msg="indel pass " + k)
    np.testing.assert_array_equal(got_i.site["shift"][live], want_i.site["shift"][live])
    assert int(ctx.L.bcfgpu_errmod_state(ctx.h)) == state       # the generator stands where the reference's would
    n = abi.C.c_uint32()
    assert ctx.L.bcfgpu_truncated_cells(ctx.h, abi.C.byref(n)) == 0
    assert n.value == 0                                          # (the deep cell of the column whose indel pass does not run spends no draw and is not written out)
    # the next tile goes on with the same generator: the same tile again draws other reads
    got2, _ = ctx.mpileup_planned(snp)
    assert not np.array_equal(got2.pl, got_s.pl)
    first = orc.mpileup(cfg, snp, deep_rule=0)                    # a fresh process on the SNP pass alone ...
    ctx.L.bcfgpu_errmod_seed(ctx.h, 0x1234ABCD330E)
    got3, _ = ctx.mpileup_planned(snp)                           # ... is what a re-seeded context gives
    assert_mplp_equal(got3, first)


def test_columns_outside_the_targets_spend_no_draw(gpu_ctx_factory):
    """mpileup -t / -T: mpileup_reg() passes a position outside the targets over before bcf_call_glfgen (mpileup.c:330-335), so a
    cell of more than 255 reads there takes nothing from hts_drand48 -- the deep cells of the visited columns draw what a reference run
    over the visited columns alone would (bcfgpu_errmod_plan_visit)."""
    rng = np.random.default_rng(21)
    n_smpl, n_sites = 2, 6
    ds = rng.poisson(40, n_sites * n_smpl)
    for c in (0, 3, 4, 7, 10):
        ds[c] = rng.integers(300, 1200)
    snp = _tile(rng, n_sites, n_smpl, ds)
    visit = np.array([1, 0, 1, 1, 0, 1], np.uint8)                # columns 1 and 4 lie outside the targets (cells 3 and 8, 9: deep cell 3)
    cfg = abi.default_cfg(n_smpl, max_sites=n_sites, max_reads=len(snp.rd))
    ctx = gpu_ctx_factory(cfg)
    kept = np.nonzero(visit)[0]
    want = orc.mpileup(cfg, snp.select_sites(list(kept)), deep_rule=0)             # the reference's run: the visited columns, one generator
    state = int(orc.lib().orc_rand48_state())
    got, _ = ctx.mpileup_planned(snp, visit=visit)
    for k in ("pl", "dp4", "adf", "adr", "qs"):
        np.testing.assert_array_equal(getattr(got, k)[kept], getattr(want, k), err_msg=k)
    assert int(ctx.L.bcfgpu_errmod_state(ctx.h)) == state
    n = abi.C.c_uint32()
    assert ctx.L.bcfgpu_truncated_cells(ctx.h, abi.C.byref(n)) == 0 and n.value == 0


def test_without_a_plan_the_first_255_are_taken(gpu_ctx_factory):
    rng = np.random.default_rng(9)
    snp = _tile(rng, 2, 2, [400, 30, 20, 700])
    cfg = abi.default_cfg(2, max_sites=2, max_reads=len(snp.rd))
    ctx = gpu_ctx_factory(cfg)
    assert_mplp_equal(ctx.mpileup(snp), orc.mpileup(cfg, snp, deep_rule=1))
    n = abi.C.c_uint32()
    assert ctx.L.bcfgpu_truncated_cells(ctx.h, abi.C.byref(n)) == 0 and n.value == 2


def test_a_plan_without_deep_cells_spends_nothing_and_bad_arguments_are_refused(gpu_ctx_factory):
    rng = np.random.default_rng(11)
    snp = _tile(rng, 3, 2, [40, 255, 30, 200, 10, 0])             # 255 reads is not over-deep
    cfg = abi.default_cfg(2, max_sites=3, max_reads=len(snp.rd))
    ctx = gpu_ctx_factory(cfg)
    s0 = int(ctx.L.bcfgpu_errmod_state(ctx.h))
    got, _ = ctx.mpileup_planned(snp)
    assert int(ctx.L.bcfgpu_errmod_state(ctx.h)) == s0 == 0x1234ABCD330E
    assert_mplp_equal(got, orc.mpileup(cfg, snp, deep_rule=0))
    assert ctx.L.bcfgpu_errmod_plan(ctx.h, None, None, None, None) == abi.E_ARG
    assert ctx.L.bcfgpu_errmod_plan(None, None, None, None, None) == abi.E_ARG
    # an indel tile without the columns it belongs to cannot be ranked
    ind = _tile(rng, 1, 2, [300, 20], is_indel=True)
    ds, sb = ctx.upload_tile(snp)
    di, ib = ctx.upload_tile(ind)
    try:
        assert ctx.L.bcfgpu_errmod_plan(ctx.h, abi.C.byref(ds), abi.C.byref(di), None, None) == abi.E_ARG
        assert int(ctx.L.bcfgpu_errmod_state(ctx.h)) == s0
    finally:
        ctx.release(sb + ib)

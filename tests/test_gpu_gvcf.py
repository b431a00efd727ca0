"""bcfgpu_gvcf_blocks (gvcf.hip) against the oracle's sequential gvcf_write (oracle/gvcf.c), bit for bit, on seeded
record streams: depth ranges that change, records that cannot join, position gaps, sequence changes, records followed by
an indel record of the same position; and the reference's golden mpileup.6.out through the device."""
import numpy as np
import pytest

from bcftools_amd import abi, engine, host
from bcftools_amd.lib import BcfGpuError
from tests.helpers import orc
from tests.test_oracle_golden_gvcf import run_gvcf_case

pytestmark = pytest.mark.gpu


def synth(n, S, seed, p_var=0.1, p_gap=0.05, p_brk=0.05, n_rid=2, depth=6):
    rng = np.random.default_rng(seed)
    res = host.MplpResult(n, S)
    var = rng.random(n) < p_var
    res.site["n_alleles"] = np.where(var, rng.integers(3, 5, n), 2)
    res.site["unseen"] = res.site["n_alleles"] - 1
    lone = rng.random(n) < 0.02                     # n_alleles == 1 (no <*>): never joins
    res.site["n_alleles"][lone] = 1
    res.site["unseen"][lone] = -1
    # depth drifts along the sites so that the smallest per-sample DP crosses the range limits
    base = np.clip(depth + np.cumsum(rng.integers(-1, 2, n)), 0, 60)
    tot = np.clip(base[:, None] + rng.integers(-2, 3, (n, S)), 0, 250)
    f = rng.integers(0, 256, (n, 4, S))
    w = f / np.maximum(f.sum(axis=1, keepdims=True), 1)
    dp4 = np.floor(w * tot[:, None, :]).astype(np.uint8)
    res.dp4[:] = dp4
    res.pl[:] = rng.integers(0, 256, res.pl.shape, dtype=np.uint8)
    res.pl[:, 1, :] = rng.integers(0, 4, (n, S), dtype=np.uint8) * 3       # many ties on PL[1]: PL[2] decides
    step = np.where(rng.random(n) < p_gap, rng.integers(2, 50, n), 1)
    pos = np.cumsum(step).astype(np.int32)
    rid = np.sort(rng.integers(0, n_rid, n)).astype(np.int32)
    brk = ((rng.random(n) < p_brk) | ((rng.random(n) < p_brk / 2) << 1)).astype(np.uint8)
    return res, pos, rid, brk


def same(a, b):
    assert a.n_blocks == b.n_blocks
    assert np.array_equal(a.blk, b.blk) and np.array_equal(a.min_dp, b.min_dp)
    for k in ("first_site", "last_site", "start_pos", "end1", "min_dp", "range"):
        assert np.array_equal(a.block[k], b.block[k]), k
    assert np.array_equal(a.dp, b.dp) and np.array_equal(a.pl, b.pl)


@pytest.mark.parametrize("n,S,seed,ranges,kw", [
    (1, 1, 1, [0, 2, 5], {}), (2, 3, 2, [0, 2, 5], {}), (257, 3, 3, [0, 2, 5], {}), (1000, 64, 4, [0, 2, 5], {}),
    (3000, 65, 5, [1, 3, 5, 10, 20], {}),                                   # first limit > 0: range 0 sites stay as they are
    (2048, 1000, 6, [0, 2, 5], dict(depth=4)), (5000, 130, 7, [0], dict(p_var=0.0, p_gap=0.0, p_brk=0.0, n_rid=1)),   # one long block
    (4096, 17, 8, [0, 5, 10, 15, 20, 25, 30, 35, 40, 45, 50, 55, 60, 65, 70, 75], dict(depth=30)),
])
def test_gvcf_blocks_match_oracle(n, S, seed, ranges, kw):
    res, pos, rid, brk = synth(n, S, seed, **kw)
    want = orc.gvcf_blocks(res, pos, ranges, rid=rid, brk=brk)
    with engine.Context(abi.default_cfg(S)) as ctx:
        got = ctx.gvcf_blocks(res, pos, ranges, rid=rid, brk=brk)
        same(got, want)
        # without the optional arrays
        same(ctx.gvcf_blocks(res, pos, ranges), orc.gvcf_blocks(res, pos, ranges))
    assert want.n_blocks > 0 or n < 3
    if n >= 1000 and len(ranges) > 1 and S <= 130:
        assert len(set(want.block["range"].tolist())) > 1 and (want.blk < 0).any()


def test_gvcf_blocks_argument_errors():
    res, pos, rid, brk = synth(8, 3, 1)
    with engine.Context(abi.default_cfg(3)) as ctx:
        with pytest.raises(BcfGpuError):
            ctx.gvcf_blocks(res, pos, list(range(17)))                     # more than 16 limits
        assert ctx.gvcf_blocks(host.MplpResult(0, 3), np.zeros(0, np.int32), [0, 2, 5]).n_blocks == 0


def test_hip_reproduces_gvcf_golden(golden_dir):
    """mpileup.6.out with every stage on the device: BAQ, overlaps, glfgen/combine, gap_prep, and the block merging."""
    ctxs = {}

    def hip_engine(cfg, tile):
        if "m" not in ctxs:
            ctxs["m"] = engine.Context(abi.default_cfg(tile.n_smpl, max_sites=8192, max_reads=1 << 20, fmt_flag=cfg.fmt_flag))
        return ctxs["m"].mpileup(tile)

    def aux_ctx():
        if "g" not in ctxs:
            ctxs["g"] = engine.Context(abi.default_cfg(1))
        return ctxs["g"]

    def hip_gvcf(cfg, res, pos, dp_range, brk):
        return ctxs["m"].gvcf_blocks(res, pos, dp_range, brk=brk)

    try:
        run_gvcf_case(golden_dir, hip_engine, hip_gvcf, gap_ctx=aux_ctx, baq_ctx=aux_ctx)
    finally:
        for c in ctxs.values():
            c.close()


def test_call_side_gvcf_blocks_match_golden_and_oracle(golden_dir, gpu_ctx_factory):
    """`bcftools call -mg0` (test/mpileup.2.out): mcall() on the device, then the `call -g` form of bcfgpu_gvcf_blocks --
    every block line of the golden; and against the oracle's sequential gvcf_write on random records with several ranges,
    sequences, records at one position and records that are blocks already (INFO/END)."""
    from tests.test_oracle_golden_gvcf import run_call_gvcf_case, orc_call_blocks
    from tests.helpers import orc
    import ctypes as C
    ctxs = {}

    def dev_mcall(cfg, cin):
        key = (cfg.n_smpl, cin.n_sites)
        c = ctxs.get(key)
        if c is None:
            cfg.max_sites = cin.n_sites
            c = ctxs[key] = gpu_ctx_factory(cfg)
        return c.mcall(cin)

    def dev_blocks(n, S, pos, is_ref, dp, ranges):
        c = gpu_ctx_factory(abi.default_cfg(S, max_sites=n, max_reads=64))
        return c.gvcf_call_blocks(pos, is_ref, dp, ranges)
    run_call_gvcf_case(golden_dir, dev_mcall, dev_blocks)
    # random records
    rng = np.random.default_rng(77)
    for S, n in ((3, 400), (70, 900), (1, 50)):
        ctx = gpu_ctx_factory(abi.default_cfg(S, max_sites=n, max_reads=64))
        step = rng.choice([0, 1, 1, 1, 1, 3], size=n)                       # 0: a second record at the position (SNP + indel)
        pos = np.cumsum(step).astype(np.int32)
        rid = np.sort(rng.integers(0, 3, n)).astype(np.int32)
        is_ref = (rng.random(n) < 0.85).astype(np.uint8)
        dp = (rng.integers(0, 9, n)[:, None] + rng.integers(0, 3, (n, S))).astype(np.int32)     # a record's samples are about equally deep
        dp[rng.random((n, S)) < 0.01] = np.iinfo(np.int32).min              # FORMAT/DP missing
        got = ctx.gvcf_call_blocks(pos, is_ref, dp, [1, 3, 6], rid=rid)
        p = orc._p
        rngs = np.array([1, 3, 6], np.int32)
        from bcftools_amd.host import GVCF_BLOCK_DTYPE
        blk, mdp = np.zeros(n, np.int32), np.zeros(n, np.int32)
        block = np.zeros(n, GVCF_BLOCK_DTYPE)
        dpo, plo, pl = np.zeros((n, S), np.int32), np.zeros((n, 3, S), np.int32), np.zeros((n, 3, S), np.int32)
        nb = orc.lib().orc_gvcf_blocks(n, S, p(pos), p(rid), None, p(is_ref), p(dp), p(pl), p(rngs), 3, p(blk), p(mdp), p(block), p(dpo), p(plo))
        assert got[0] == nb and nb > 0
        np.testing.assert_array_equal(got[1], blk)
        np.testing.assert_array_equal(got[2], mdp)
        for f in ("first_site", "last_site", "start_pos", "end1", "min_dp", "range"):
            np.testing.assert_array_equal(got[3][f][:nb], block[f][:nb], err_msg=f)
        np.testing.assert_array_equal(got[4][:nb], dpo[:nb])

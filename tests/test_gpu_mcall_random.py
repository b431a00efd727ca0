"""Differential test of the stand-alone caller (bcfgpu_mcall: int32 PLs as parsed from a VCF/BCF) against the oracle on
randomised records: 1-5 alleles with or without the unseen allele, missing PL entries and all-missing samples,
haploid / absent samples (ploidy 1 / 0, vector_end in PL), -G sample groups on FORMAT/AD, -F AN,AC priors, -v, -A,
GQ/GP.  Integers exact, QUAL within 1e-4."""
import numpy as np
import pytest

from bcftools_amd import abi, host
from tests.helpers import orc
from tests.test_gpu_parity import assert_call_equal

pytestmark = pytest.mark.gpu

MISSING, VEND = abi.INT32_MISSING, abi.INT32_VECTOR_END


def random_records(seed, n_sites, n_smpl, use_ploidy, n_grp, use_prior):
    rng = np.random.default_rng(seed)
    nals = rng.integers(1, 6, n_sites).astype(np.int32)
    unseen = np.where(rng.random(n_sites) < 0.7, nals - 1, 0).astype(np.int32)
    unseen[nals == 1] = 0
    ploidy = rng.choice([0, 1, 2, 2, 2, 2], size=n_smpl).astype(np.uint8) if use_ploidy else None
    pl = np.full((n_sites, 15, n_smpl), VEND, dtype=np.int32)
    qs = np.zeros((n_sites, 5), dtype=np.float32)
    ad = np.full((n_sites, 5, n_smpl), VEND, dtype=np.int32) if n_grp > 1 else None
    for k in range(n_sites):
        na = int(nals[k])
        ng = na * (na + 1) // 2
        af = rng.dirichlet(np.r_[8.0, np.full(na - 1, 0.6)]) if na > 1 else np.array([1.0])
        if unseen[k] > 0:
            # no read shows the unseen allele <*>: with QS = 0 it is never part of a candidate subset.  (If it were
            # selected, the reference indexes GPs/gts beyond nals_new -- mcall.c:1571-1577 drops it from nals_new but
            # not from als_map -- which is undefined behaviour, not something to test parity on.)
            af[unseen[k]] = 0.0
            af /= af.sum()
        q = np.zeros(5)
        for s in range(n_smpl):
            pd = 2 if ploidy is None else int(ploidy[s])
            depth = rng.poisson(8)
            if depth == 0 and rng.random() < 0.5:
                pl[k, :ng, s] = MISSING if rng.random() < 0.5 else 0
            else:
                g = sorted(rng.choice(na, size=2, p=af))
                cnt = np.bincount(rng.choice(g, size=max(depth, 1)), minlength=na)
                # phred likelihoods of every genotype from the allele counts, error 1 %
                v = np.zeros(ng)
                z = 0
                for b in range(na):
                    for a in range(b + 1):
                        pa = np.array([0.99 if i in (a, b) and a == b else 0.495 if i in (a, b) else 0.005 for i in range(na)])
                        v[z] = -10 * (cnt * np.log10(pa)).sum()
                        z += 1
                v = np.minimum(np.round(v - v.min()), 255).astype(np.int32)
                pl[k, :ng, s] = v
                if pd == 2 and rng.random() < 0.1:           # some entries missing (filled from the unseen allele, mcall.c:495-527;
                                                             # diploid only: a haploid vector with holes is not valid input to set_pdg)
                    pl[k, rng.integers(0, ng), s] = MISSING
                q[:na] += cnt * 30
                if ad is not None:
                    ad[k, :na, s] = cnt
            if pd == 1 and ng > na:                          # haploid: na values then vector_end
                diag = [(a + 1) * (a + 2) // 2 - 1 for a in range(na)]
                vals = pl[k, diag, s].copy()
                pl[k, :, s] = VEND
                pl[k, :na, s] = vals
            elif pd == 0:
                pl[k, :, s] = VEND
                pl[k, 0, s] = MISSING
        if q.sum() > 0:
            qs[k] = (q / q.sum()).astype(np.float32)
    grp = rng.integers(0, n_grp, n_smpl).astype(np.int32) if n_grp > 1 else None
    prior_an = prior_ac = None
    if use_prior:
        prior_an = np.full(n_sites, 200, dtype=np.int32)
        prior_ac = np.full((n_sites, 4), VEND, dtype=np.int32)
        for k in range(n_sites):
            na = int(nals[k])
            if na > 1:
                prior_ac[k, :na - 1] = rng.multinomial(40, np.full(na - 1, 1.0 / (na - 1)))
            if rng.random() < 0.2:
                prior_an[k] = MISSING
    # INFO/I16 with arbitrary (consistent) moments: counts, then sum and sum of squares of three quantities for ref and alt
    i16 = np.zeros((n_sites, 16), dtype=np.float32)
    for k in range(n_sites):
        cnt = rng.integers(0, 40, 4) * (rng.random(4) < 0.8)
        i16[k, :4] = cnt
        for t in range(3):
            for side, n in ((0, int(cnt[0] + cnt[1])), (1, int(cnt[2] + cnt[3]))):
                v = rng.integers(0, 60, n) if n else np.zeros(0)
                i16[k, 4 + 4 * t + 2 * side] = v.sum()
                i16[k, 5 + 4 * t + 2 * side] = (v * v).sum()
    return host.CallInput(n_smpl, nals, unseen, pl, qs, ad=ad, ploidy=ploidy, grp=grp, prior_an=prior_an, prior_ac=prior_ac,
                          i16=i16 if seed % 2 else None)


@pytest.mark.parametrize("seed,n_sites,n_smpl,use_ploidy,n_grp,use_prior,flags,tags", [
    (1, 60, 40, False, 1, False, 0, abi.CALL_FMT_PV4),
    (2, 60, 70, True, 1, False, 0, abi.CALL_FMT_GQ | abi.CALL_FMT_GP),
    (3, 40, 33, True, 3, False, abi.CALL_VARONLY, abi.CALL_FMT_PV4),
    (4, 40, 20, False, 1, True, abi.CALL_KEEPALT, abi.CALL_FMT_GQ),
    (5, 30, 130, True, 4, True, 0, abi.CALL_FMT_PV4),
    (6, 80, 1, False, 1, False, 0, abi.CALL_FMT_GP),
])
def test_mcall_matches_oracle_on_random_records(gpu_ctx_factory, seed, n_sites, n_smpl, use_ploidy, n_grp, use_prior, flags, tags):
    cin = random_records(seed, n_sites, n_smpl, use_ploidy, n_grp, use_prior)
    cfg = abi.default_cfg(n_smpl, max_sites=n_sites, call_flag=flags, output_tags=tags, n_grp=n_grp)
    want = orc.mcall(cfg, cin)
    got = gpu_ctx_factory(cfg).mcall(cin)
    assert_call_equal(got, want, n_smpl)
    live = (want.site["ret"] > 0) & (want.site["als_new"] != 1)
    if tags & abi.CALL_FMT_GQ:
        np.testing.assert_array_equal(got.gq[live], want.gq[live])
    if tags & abi.CALL_FMT_GP:
        for i in np.nonzero(live)[0]:
            nn = int(want.site["nals_new"][i])
            ng = nn * (nn + 1) // 2
            g, w = got.gp[i, :ng], want.gp[i, :ng]
            assert np.array_equal(np.isnan(g), np.isnan(w))
            np.testing.assert_allclose(g[~np.isnan(w)], w[~np.isnan(w)], rtol=1e-5, atol=1e-7)


def test_pv4_of_reference_golden_on_device(golden_dir, gpu_ctx_factory):
    """INFO/PV4, DP4 of the reference's golden test/mpileup.c.1.out (test.pl:298) from the INFO/I16 of its input, through
    bcfgpu_mcall with the I16 vectors attached to dummy one-sample records (the calling itself is not what is checked)."""
    import os
    from tests.helpers import vcf
    g = os.path.join(golden_dir, "call")
    src = {(r.pos, "INDEL" in r.info): r for r in vcf.Vcf(os.path.join(g, "mpileup.c.vcf")).recs}
    out = [r for r in vcf.Vcf(os.path.join(g, "mpileup.c.1.out")).recs]
    n = len(out)
    i16 = np.array([src[(r.pos, "INDEL" in r.info)].info_floats("I16") for r in out], dtype=np.float32)
    pl = np.zeros((n, 3, 1), dtype=np.int32)
    pl[:, 1, 0], pl[:, 2, 0] = 30, 60
    cin = host.CallInput(1, np.full(n, 2, np.int32), np.zeros(n, np.int32), pl, np.tile(np.array([[0.9, 0.1, 0, 0, 0]], np.float32), (n, 1)),
                         i16=i16)
    cfg = abi.default_cfg(1, max_sites=n, output_tags=abi.CALL_FMT_PV4)
    got = gpu_ctx_factory(cfg).mcall(cin)
    n_pv4 = 0
    for k, r in enumerate(out):
        st = got.site[k]
        assert int(st["has_i16"]) == 1 and [int(x) for x in st["dp4"]] == r.info_ints("DP4")
        if "PV4" in r.info:
            assert int(st["pv4_tested"]) == 1
            np.testing.assert_allclose(st["pv4"], r.info_floats("PV4"), rtol=5e-6, atol=1e-12)
            n_pv4 += 1
        else:
            assert int(st["pv4_tested"]) == 0
    assert n_pv4 == 11


def test_records_outside_the_supported_range_are_refused(gpu_ctx_factory):
    """mcall() itself takes up to 32 alleles (mcall.c:1539); the planes of this ABI hold B2B_MAX_ALLELES = 5.  A record with
    more, or a sample group id outside [0, n_grp), must not index past the kernel's tables: the record gets ret = -2, the
    others are called as usual, and the call reports BCFGPU_E_RANGE."""
    from bcftools_amd.lib import BcfGpuError
    n_sites, n_smpl = 12, 30
    cin = random_records(5, n_sites, n_smpl, False, 1, False)
    nals0 = cin.nals.copy()
    cin.nals = cin.nals.copy()
    cin.nals[3] = 6
    cin.nals[7] = 0
    ctx = gpu_ctx_factory(abi.default_cfg(n_smpl, max_sites=n_sites, max_reads=64))
    with pytest.raises(BcfGpuError) as e:
        ctx.mcall(cin)
    assert e.value.code == abi.E_RANGE
    # the context stays usable and the valid records of a later call are untouched by the refusal
    cin.nals = nals0
    res = ctx.mcall(cin)
    want = orc.mcall(abi.default_cfg(n_smpl), cin)
    assert_call_equal(res, want, n_smpl)
    # a group id out of range
    cing = random_records(6, n_sites, n_smpl, False, 3, False)
    cing.grp = cing.grp.copy()
    cing.grp[5] = 7
    cfg = abi.default_cfg(n_smpl, max_sites=n_sites, max_reads=64, n_grp=3)
    with pytest.raises(BcfGpuError) as e:
        gpu_ctx_factory(cfg).mcall(cing)
    assert e.value.code == abi.E_RANGE

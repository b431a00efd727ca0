"""The reference's own fixtures through the HIP path (C-ABI): the BAQ-free mpileup golden and every `call -m` golden."""
import os
import pytest

from bcftools_amd import abi, engine
from tests.test_oracle_golden_call import run_case, N_CASES
from tests.test_oracle_golden_cals import run_cals_case, CASES as CALS_CASES
from tests.test_oracle_golden_mpileup import build, check_against_golden

pytestmark = pytest.mark.gpu


def test_hip_reproduces_mpileup3_golden(golden_dir, gpu_ctx_factory):
    t, tile, gold = build(golden_dir)
    cfg = abi.default_cfg(1, max_sites=tile.n_sites, max_reads=len(tile.rd))
    res = gpu_ctx_factory(cfg).mpileup(tile)
    check_against_golden(t, res, gold)


@pytest.mark.parametrize("idx", range(N_CASES))
def test_hip_reproduces_call_golden(golden_dir, idx):
    def hip_engine(cfg, cin):
        with engine.Context(cfg) as ctx:
            return ctx.mcall(cin)
    run_case(os.path.join(golden_dir, "call"), idx, hip_engine)


@pytest.mark.parametrize("idx", range(len(CALS_CASES)))
def test_hip_reproduces_constrained_alleles_golden(golden_dir, idx):
    """`call -mA -C alleles -T tab [-i]` (test.pl:289-297): mcall with BCFGPU_CALL_KEEPALT on the device."""
    def hip_engine(cfg, cin):
        with engine.Context(cfg) as ctx:
            return ctx.mcall(cin)
    run_cals_case(os.path.join(golden_dir, "call"), idx, hip_engine)


from tests.test_oracle_golden_baq import CASES as BAQ_CASES, run_case as run_baq_case


@pytest.mark.parametrize("idx", range(len(BAQ_CASES)))
def test_hip_reproduces_default_mpileup_golden(golden_dir, idx):
    """SNP and indel records of the BAQ-on goldens with every device stage in the loop: BAQ (bcfgpu_baq), the SNP and
    indel passes (glfgen_kernel/combine_kernel) and the indel realignment (bcfgpu_gap_prep)."""
    ctxs = {}

    def hip_engine(cfg, tile):
        key = tile.n_smpl
        if key not in ctxs:
            c = abi.default_cfg(tile.n_smpl, max_sites=8192, max_reads=1 << 20, fmt_flag=cfg.fmt_flag)
            ctxs[key] = engine.Context(c)
        return ctxs[key].mpileup(tile)
    def gap_ctx():
        # bcf_call_gap_prep through bcfgpu_gap_prep (host typing + probaln_kernel on the device)
        if "gap" not in ctxs:
            ctxs["gap"] = engine.Context(abi.default_cfg(1))
        return ctxs["gap"]
    try:
        run_baq_case(golden_dir, BAQ_CASES[idx], hip_engine, gap_ctx=gap_ctx, baq_ctx=gap_ctx)       # BAQ through bcfgpu_baq too
    finally:
        for c in ctxs.values():
            c.close()

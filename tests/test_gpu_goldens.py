"""The reference's own fixtures through the HIP path (C-ABI): the BAQ-free mpileup golden and every `call -m` golden."""
import os
import pytest

from bcftools_amd import abi, engine
from tests.test_oracle_golden_call import run_case, N_CASES
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

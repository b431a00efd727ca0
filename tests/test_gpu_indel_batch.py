"""bcfgpu_gap_prep on a batch of synthetic indel-candidate columns (many sites in one call, the way a tile is run)
against the oracle's bcf_call_gap_prep one site at a time: p->aux of every pileup entry, the candidate types, the
insertion consensus, indelreg, max_support, max_frac -- all exact."""
import numpy as np
import pytest

from bcftools_amd import abi, synth
from tests.helpers import indeldrv

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_sites,n_smpl,depth,seed,kw", [
    (24, 20, 15.0, 41, {}),
    (16, 3, 40.0, 42, dict(min_support=2)),
    (12, 50, 8.0, 43, dict(per_sample_flt=1, min_frac=0.05)),
    (6, 1, 60.0, 44, dict(openQ=30, extQ=10)),
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

"""Determinism probe of bcfgpu_gap_prep: the same batch twice on one context, on a fresh one, and against the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bcftools_amd import abi, synth, engine
from tests.helpers import indeldrv

n_smpl = 30
batches = [synth.indel_batch(70 + j, 16, n_smpl, depth=12.0) for j in range(6)]
c0 = engine.Context(abi.default_cfg(n_smpl, max_sites=16, max_reads=64))
c1 = engine.Context(abi.default_cfg(n_smpl, max_sites=16, max_reads=64))
want = [indeldrv.gap_prep_gpu(c0, b)[0] for b in batches]
bad = 0
for rep in range(3):
    for j, b in enumerate(batches):
        for name, c in (("c0", c0), ("c1", c1)):
            g = indeldrv.gap_prep_gpu(c, b)[0]
            d = np.nonzero(g["aux"] != want[j]["aux"])[0]
            if len(d):
                bad += 1
                print("rep", rep, "batch", j, name, "aux differs at", len(d), "entries, first", d[:5], g["aux"][d[:5]], want[j]["aux"][d[:5]])
# oracle on batch 1
for j in (0, 1):
    b = batches[j]
    for k in range(b["n_sites"]):
        try:
            indeldrv.assert_site_equal(want[j], k, indeldrv.gap_prep_oracle_site(b, k))
        except AssertionError as e:
            print("oracle mismatch batch", j, "site", k, str(e)[:300])
            bad += 1
print("bad =", bad)

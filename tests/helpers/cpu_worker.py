"""One worker of bench.py's all-cores CPU baseline: the oracle (mpileup + call -m) on a region shard of synthetic
pileup of the benchmark's shape, on one core.  Prints `sites seconds` for the timed part.

    python -m tests.helpers.cpu_worker <seed> <n_sites> <n_smpl> <depth> <go_file> [reps]

The tile is generated first (not timed); the worker then waits for <go_file> to appear so that all workers of a run
compute at the same time, which is what "all cores, region-sharded" means."""
import os
import sys
import time

import numpy as np


def main():
    seed, n_sites, n_smpl, depth, go = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), sys.argv[5]
    reps = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    from bcftools_amd import abi, synth, host
    from tests.helpers import orc
    tile = synth.numpy_tile(seed, n_sites, n_smpl, depth=depth, var_rate=0.01)
    cfg = abi.default_cfg(n_smpl, max_sites=n_sites, max_reads=len(tile.rd))
    orc.mpileup(cfg, tile.select_sites(np.arange(min(2, n_sites))))         # load the library, build errmod tables
    open(go + ".ready.%d" % os.getpid(), "w").close()
    while not os.path.exists(go):
        time.sleep(0.01)
    t0 = time.perf_counter()
    for _ in range(reps):
        m = orc.mpileup(cfg, tile)
        cin = host.CallInput(n_smpl, m.site["n_alleles"], np.maximum(m.site["unseen"], 0), m.pl.astype(np.int32), m.site["qsum"])
        orc.mcall(cfg, cin)
    print(n_sites * reps, time.perf_counter() - t0, flush=True)


if __name__ == "__main__":
    main()

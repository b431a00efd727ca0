"""Runs a fixed set of seeded inputs through every stage of the library the process loads (BCFGPU_SO picks the build) and prints one
SHA-256 per stage: tests/test_gpu_plain_build.py compares the product build with the build without the per-file LLVM options."""
import hashlib
import sys

import numpy as np


def main():
    from bcftools_amd import abi, synth, engine
    from tests.helpers import indeldrv
    out = {}

    def h(*arrs):
        m = hashlib.sha256()
        for a in arrs:
            m.update(np.ascontiguousarray(a).tobytes())
        return m.hexdigest()
    # SNP path + call -m (glfgen, combine, mcall), with sample groups and a ploidy array
    n_sites, S = 96, 130
    tile = synth.numpy_tile(777, n_sites, S, depth=25.0, var_rate=0.3)
    rng = np.random.default_rng(5)
    ploidy = rng.choice([1, 2, 2, 2], size=S).astype(np.uint8)
    grp = (np.arange(S) * 3 // S).astype(np.int32)
    cfg = abi.default_cfg(S, max_sites=n_sites, max_reads=len(tile.rd), n_grp=3, output_tags=abi.CALL_FMT_GQ,
                          fmt_flag=abi.INFO_VDB | abi.INFO_RPB | abi.FMT_AD | abi.FMT_SP)
    with engine.Context(cfg) as ctx:
        m, c = ctx.pipeline(tile, ploidy=ploidy, grp=grp)
        out["snp"] = h(m.pl, m.dp4, m.adf, m.adr, m.site, c.gt, c.gq, c.site)
    # indel stage: register classes, the wide-band kernels
    b = synth.indel_batch(778, 10, 40, depth=15.0, lens=(-40, -12, -8, 8, 25, 3, -2), lens2=(-3, 1, 2))
    with engine.Context(abi.default_cfg(40, max_sites=10, max_reads=len(b["p_read"]) + 64)) as ctx:
        got, _ = indeldrv.gap_prep_gpu(ctx, b)
        out["indel"] = h(*[got[k] for k in sorted(got)])
    # BAQ (bcfgpu_baq on the batch's reads)
    import ctypes as C
    from bcftools_amd.lib import check
    with engine.Context(abi.default_cfg(1, max_sites=1, max_reads=64)) as ctx:
        R = b["reads"]
        rd = abi.Reads()
        rd.n_reads = R["n_reads"]
        for k in indeldrv.READ_KEYS:
            setattr(rd, k, R[k].ctypes.data)
        nb = len(R["qual"])
        qo, zo, ret = np.zeros(nb, np.uint8), np.zeros(nb, np.uint8), np.zeros(R["n_reads"], np.int32)
        check(ctx.L.bcfgpu_baq(ctx.h, C.byref(rd), b["ref"], len(b["ref"]), 3, qo.ctypes.data, zo.ctypes.data, ret.ctypes.data))
        out["baq"] = h(qo, zo, ret)
    for k in sorted(out):
        print(k, out[k])


if __name__ == "__main__":
    sys.exit(main())

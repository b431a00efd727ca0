"""bcfgpu_overlap_tweak (the mate-overlap quality tweak on the device) against the C oracle: the overlapping mates of the
reference's SAM fixtures, random pairs with every CIGAR operation, and the argument checks."""
import ctypes as C
import os

import numpy as np
import pytest

from bcftools_amd import abi
from bcftools_amd.lib import BcfGpuError
from tests.helpers import sam, mplpdrv as M, ovlfuzz

pytestmark = pytest.mark.gpu


def _run(pairs, ctx):
    reads = [r for ab in pairs for r in ab]
    q0 = [r.qual.copy() for r in reads]
    rd, d = M.pack_reads(reads)
    pa = np.arange(0, len(reads), 2, dtype=np.int32)
    pb = pa + 1
    from tests.helpers import orc
    L = orc.lib()
    L.orc_overlap_tweak.restype = C.c_int
    L.orc_overlap_tweak.argtypes = [C.POINTER(abi.Reads), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    want = d["qual"].copy()
    assert L.orc_overlap_tweak(C.byref(rd), len(pairs), pa.ctypes.data, pb.ctypes.data, want.ctypes.data) == 0
    got = np.zeros_like(want)
    from bcftools_amd.lib import check
    check(ctx.L.bcfgpu_overlap_tweak(ctx.h, C.byref(rd), len(pairs), pa.ctypes.data, pb.ctypes.data, got.ctypes.data))
    np.testing.assert_array_equal(d["qual"], np.concatenate(q0).astype(np.uint8))     # the input pool is left alone
    return want, got, d["qual"]


@pytest.mark.parametrize("samf", ["mpileup.1.sam", "mpileup.2.sam", "mpileup.4.sam"])
def test_overlap_matches_oracle_on_reference_mates(golden_dir, gpu_ctx_factory, samf):
    s = sam.Sam(os.path.join(golden_dir, "mpileup", samf))
    reads = [r for r in s.reads if sam.keep_read(r, sam.MplpOpts())]
    pairs = M.overlap_pairs(reads)
    ctx = gpu_ctx_factory(abi.default_cfg(1, max_sites=1, max_reads=64))
    want, got, q0 = _run(pairs, ctx)
    np.testing.assert_array_equal(got, want)
    assert (want != q0).any()


def test_overlap_matches_oracle_on_random_pairs(gpu_ctx_factory):
    ctx = gpu_ctx_factory(abi.default_cfg(1, max_sites=1, max_reads=64))
    for seed, n in ((11, 1), (12, 63), (13, 2000)):
        want, got, q0 = _run(ovlfuzz.pairs(seed, n), ctx)
        np.testing.assert_array_equal(got, want)
    assert (want != q0).any() and int(want.max()) == 200


def test_overlap_argument_checks(gpu_ctx_factory):
    ctx = gpu_ctx_factory(abi.default_cfg(1, max_sites=1, max_reads=64))
    pairs = ovlfuzz.pairs(3, 4)
    reads = [r for ab in pairs for r in ab]
    rd, d = M.pack_reads(reads)
    out = np.zeros_like(d["qual"])
    from bcftools_amd.lib import check
    # no pairs: the pool comes back unchanged
    check(ctx.L.bcfgpu_overlap_tweak(ctx.h, C.byref(rd), 0, None, None, out.ctypes.data))
    np.testing.assert_array_equal(out, d["qual"])
    # a read in two pairs, an index out of range
    for pa, pb in (([0, 0], [1, 2]), ([0, 2], [1, 99])):
        a, b = np.array(pa, np.int32), np.array(pb, np.int32)
        with pytest.raises(BcfGpuError):
            check(ctx.L.bcfgpu_overlap_tweak(ctx.h, C.byref(rd), 2, a.ctypes.data, b.ctypes.data, out.ctypes.data))

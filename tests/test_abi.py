"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol include/bcfgpu.h declares,
its struct sizes match the Python mirror, the host-side packer agrees with the test twin, and -- without a GPU --
the product path refuses to run instead of falling back to the CPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from bcftools_amd import abi, lib
from tests.helpers import sam

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = lib.load()
    hdr = open(os.path.join(ROOT, "include", "bcfgpu.h")).read()
    declared = set(re.findall(r"^(?:int|void|size_t|uint64_t|const char|bcfgpu_[a-z_]+)\s+\*?(bcfgpu_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(L, name), "libbcfgpu.so does not export %s" % name
    assert declared == set(abi.PROTOTYPES), declared ^ set(abi.PROTOTYPES)


def test_struct_sizes_match():
    L = lib.load()
    sizes = (C.c_int32 * 8)()
    L.bcfgpu_abi_sizes(sizes)
    want = [C.sizeof(x) for x in (abi.Cfg, abi.Tile, abi.Site, abi.MplpOut, abi.CallIn, abi.CallSite, abi.CallOut, abi.Timing)]
    assert list(sizes) == want


def test_pack_read_matches_python_twin():
    L = lib.load()
    rng = np.random.default_rng(5)
    cigars = [[(100, "M")], [(5, "S"), (90, "M"), (5, "S")], [(30, "M"), (2, "I"), (68, "M")],
              [(10, "H"), (50, "M"), (3, "D"), (47, "M"), (3, "S")], [(20, "S"), (80, "M")]]
    for cig in cigars:
        lq = sum(n for n, op in cig if op in "MIS=X")
        bam = np.array([(n << 4) | "MIDNSHP=X".index(op) for n, op in cig], dtype=np.uint32)
        for _ in range(50):
            qpos = int(rng.integers(0, lq))
            nt, bq, mq = int(rng.integers(0, 16)), int(rng.integers(0, 94)), int(rng.integers(0, 256))
            fl = [int(x) for x in rng.integers(0, 2, 4)]
            rd, ep = C.c_uint32(), C.c_uint8()
            L.bcfgpu_pack_read(nt, bq, mq, fl[0], fl[1], fl[2], fl[3], qpos, lq, bam.ctypes.data_as(C.c_void_p),
                               len(bam), 1, C.byref(rd), C.byref(ep))
            w, e = sam.pack_read(nt, bq, mq, fl[0], fl[1], fl[2], fl[3], qpos, lq, cig, True)
            assert (rd.value, ep.value) == (w, e)


def test_no_cpu_fallback_without_device():
    L = lib.load()
    if L.bcfgpu_device_count() > 0:
        pytest.skip("a HIP device is present")
    cfg = abi.default_cfg(4, max_sites=4, max_reads=64)
    h = C.c_void_p()
    rc = L.bcfgpu_create(C.byref(cfg), C.byref(h))
    assert rc == abi.E_NODEV and not h.value
    assert b"no HIP device" in L.bcfgpu_last_error()

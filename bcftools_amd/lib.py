"""Loader of the in-tree HIP library libbcfgpu.so (the C-ABI of include/bcfgpu.h).

The product path has no CPU fallback: if the library is missing or no HIP device is
present, the calls fail loudly.
"""
import ctypes as C
import os
import subprocess

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# BCFGPU_SO: another build of the same library (tools/file_variants.sh, tools/so_variants.sh time experiment builds without
# touching the in-tree product library)
SO_PATH = os.environ.get("BCFGPU_SO") or os.path.join(_HERE, "libbcfgpu.so")
_LIB = None


class BcfGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("bcfgpu error %d: %s" % (code, msg))
        self.code = code


def build(force=False):
    """Compile the HIP sources for gfx950 with hipcc (bcftools_amd/csrc/Makefile)."""
    cmd = ["make", "-s", "-C", os.path.join(_HERE, "csrc")]
    if force:
        subprocess.check_call(cmd + ["clean"])
    subprocess.check_call(cmd + ["-j%d" % max(1, min(8, os.cpu_count() or 1))])      # (a file per job: mcall.hip alone is a minute)
    return SO_PATH


def load():
    """dlopen libbcfgpu.so and attach prototypes for every symbol the header declares."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise ImportError("%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback for the hot path)" % SO_PATH)
        # One HIP runtime per process: PyTorch ships its own libamdhip64; if torch is imported after this library has
        # pulled in the system runtime, torch's device initialisation fails (hipErrorNoDevice).  Importing torch first
        # makes both use the copy torch loads.  Without torch (a plain C caller, smoke) nothing changes.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(SO_PATH)
        for name, (res, args) in abi.PROTOTYPES.items():
            fn = getattr(L, name)          # AttributeError here = the library does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        sizes = (C.c_int32 * 8)()
        L.bcfgpu_abi_sizes(sizes)
        want = [C.sizeof(x) for x in (abi.Cfg, abi.Tile, abi.Site, abi.MplpOut, abi.CallIn, abi.CallSite,
                                      abi.CallOut, abi.Timing)]
        if list(sizes) != want:
            raise ImportError("bcftools_amd.abi is out of sync with include/bcfgpu.h: %s vs %s" % (list(sizes), want))
        _LIB = L
    return _LIB


def check(rc):
    if rc != 0:
        raise BcfGpuError(rc, load().bcfgpu_last_error().decode())

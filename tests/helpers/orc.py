"""ctypes binding of the CPU oracle (oracle/liboracle.so) for tests, smoke() and bench's cpu_baseline leg.

The oracle takes the same SoA structs as the device library, with host (numpy) pointers.
"""
import ctypes as C
import os
import subprocess
import numpy as np

from bcftools_amd import abi

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_ODIR = os.path.join(_ROOT, "oracle")
_LIB = None


class CallRet(C.Structure):
    _fields_ = [("p", C.c_float * 25), ("anno", C.c_double * 16),
                ("QS", C.c_int32 * 4), ("ADF", C.c_int32 * 4), ("ADR", C.c_int32 * 4),
                ("SCR", C.c_int32), ("n", C.c_int32), ("ori_depth", C.c_uint32), ("mq0", C.c_uint32)]


CALLRET_DTYPE = np.dtype([("p", "<f4", 25), ("anno", "<f8", 16), ("QS", "<i4", 4), ("ADF", "<i4", 4),
                          ("ADR", "<i4", 4), ("SCR", "<i4"), ("n", "<i4"), ("ori_depth", "<u4"), ("mq0", "<u4")],
                         align=True)

from bcftools_amd.host import (SITE_DTYPE, CALLSITE_DTYPE, HostTile, MplpResult, CallInput, CallResult, _p)

assert CALLRET_DTYPE.itemsize == C.sizeof(CallRet)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_ODIR, "liboracle.so")
        srcs = [os.path.join(_ODIR, f) for f in os.listdir(_ODIR) if f.endswith((".c", ".h"))]
        srcs.append(os.path.join(_ROOT, "include", "bcfgpu.h"))
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["make", "-s", "-C", _ODIR])
        L = C.CDLL(so)
        L.orc_mpileup.restype = C.c_int
        L.orc_mpileup.argtypes = [C.POINTER(abi.Cfg), C.POINTER(abi.Tile), C.POINTER(abi.MplpOut), C.c_void_p]
        L.orc_mcall.restype = C.c_int
        L.orc_mcall.argtypes = [C.POINTER(abi.Cfg), C.POINTER(abi.CallIn), C.POINTER(abi.CallOut)]
        L.orc_errmod_init.restype = C.c_void_p
        L.orc_errmod_init.argtypes = [C.c_double]
        L.orc_errmod_destroy.argtypes = [C.c_void_p]
        L.orc_errmod_cal.restype = C.c_int
        L.orc_errmod_cal.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        for f in ("orc_errmod_fk", "orc_errmod_beta", "orc_errmod_lhet"):
            getattr(L, f).restype = C.POINTER(C.c_double)
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_kf_erfc.restype = C.c_double
        L.orc_kf_erfc.argtypes = [C.c_double]
        L.orc_calc_vdb.restype = C.c_double
        L.orc_calc_vdb.argtypes = [C.c_void_p, C.c_int]
        L.orc_calc_mwu_bias.restype = C.c_double
        L.orc_calc_mwu_bias.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_format_sp.restype = C.c_int
        L.orc_format_sp.argtypes = [C.c_int] * 4
        L.orc_gvcf_blocks.restype = C.c_int
        L.orc_gvcf_blocks.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_int] + [C.c_void_p] * 5
        L.orc_errmod_deep_rule.argtypes = [C.c_int]
        L.orc_srand48_reset.argtypes = []
        L.orc_drand48.restype = C.c_double
        L.orc_rand48_state.restype = C.c_uint64
        _LIB = L
    return _LIB


def gvcf_blocks(res, pos, dp_range, rid=None, brk=None):
    """gvcf_write over the records of a host MplpResult, sequentially as the reference does it (oracle/gvcf.c)."""
    from bcftools_amd.host import GVCF_BLOCK_DTYPE, GvcfResult
    n, S = res.n_sites, res.n_smpl
    pos = np.ascontiguousarray(pos, dtype=np.int32)
    rid = None if rid is None else np.ascontiguousarray(rid, dtype=np.int32)
    brk = None if brk is None else np.ascontiguousarray(brk, dtype=np.uint8)
    is_ref = np.ascontiguousarray((res.site["n_alleles"] == 2) & (res.site["unseen"] == 1), dtype=np.uint8)   # mpileup.c:309-315
    dp = np.ascontiguousarray(res.dp4.sum(axis=1, dtype=np.int32))                                       # bam2bcf.c:853-858
    pl = np.ascontiguousarray(res.pl[:, :3, :].astype(np.int32))
    rng = np.ascontiguousarray(dp_range, dtype=np.int32)
    blk, min_dp = np.zeros(n, np.int32), np.zeros(n, np.int32)
    block = np.zeros(max(n, 1), GVCF_BLOCK_DTYPE)
    dpo, plo = np.zeros((max(n, 1), S), np.int32), np.zeros((max(n, 1), 3, S), np.int32)
    nb = lib().orc_gvcf_blocks(n, S, _p(pos), _p(rid), _p(brk), _p(is_ref), _p(dp), _p(pl), _p(rng), len(rng),
                               _p(blk), _p(min_dp), _p(block), _p(dpo), _p(plo))
    return GvcfResult(nb, blk, min_dp, block, dpo, plo)


def mpileup(cfg, tile, want_callret=False, deep_rule=1, reset=True):
    """Run the oracle's mpileup stage on a HostTile.  deep_rule: what errmod_cal does to a cell of more than 255 usable reads --
    1 (default here): the first 255 in pileup order, the rule of the device library when it is not told where the reference's
    generator stands; 0: the reference's own ks_shuffle draw from a freshly started hts_drand48 (oracle/errmod.c).  Every count,
    QS, I16 sum and histogram is over all reads under either rule."""
    lib().orc_errmod_deep_rule(int(deep_rule))
    if reset:                                # (reset=False: the generator goes on from where the last call left it, as the process-wide one does)
        lib().orc_srand48_reset()
    res = MplpResult(tile.n_sites, tile.n_smpl)
    t, o = tile.as_struct(), res.as_struct()
    cr = np.zeros(tile.n_sites * tile.n_smpl, dtype=CALLRET_DTYPE) if want_callret else None
    rc = lib().orc_mpileup(C.byref(cfg), C.byref(t), C.byref(o), _p(cr))
    if rc != 0:
        raise RuntimeError("orc_mpileup failed: %d" % rc)
    return (res, cr) if want_callret else res


def mcall(cfg, cin):
    res = CallResult(cin.n_sites, cin.n_smpl, cin.n_gt_max)
    i, o = cin.as_struct(), res.as_struct()
    rc = lib().orc_mcall(C.byref(cfg), C.byref(i), C.byref(o))
    if rc != 0:
        raise RuntimeError("orc_mcall failed: %d" % rc)
    return res

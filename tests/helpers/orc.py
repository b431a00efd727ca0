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

SITE_DTYPE = np.dtype([("a", "<i4", 5), ("n_alleles", "<i4"), ("unseen", "<i4"), ("ori_ref", "<i4"),
                       ("shift", "<i4"), ("ret", "<i4"), ("depth", "<u4"), ("ori_depth", "<u4"), ("mq0", "<u4"),
                       ("qsum", "<f4", 5), ("vdb", "<f4"), ("mwu_pos", "<f4"), ("mwu_mq", "<f4"),
                       ("mwu_bq", "<f4"), ("mwu_mqs", "<f4"), ("seg_bias", "<f4"),
                       ("adf_tot", "<i4", 5), ("adr_tot", "<i4", 5), ("scr_tot", "<i4"), ("pad", "<i4"),
                       ("anno", "<f8", 16)], align=True)

CALLSITE_DTYPE = np.dtype([("ret", "<i4"), ("nals_new", "<i4"), ("als_new", "<i4"), ("als_map", "<i4", 5),
                           ("ac", "<i4", 5), ("an", "<i4"), ("qual_missing", "<i4"), ("qual", "<f4"),
                           ("pl_dropped", "<i4")], align=True)

assert SITE_DTYPE.itemsize == C.sizeof(abi.Site), (SITE_DTYPE.itemsize, C.sizeof(abi.Site))
assert CALLSITE_DTYPE.itemsize == C.sizeof(abi.CallSite)
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
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class HostTile:
    """A pileup tile held in numpy arrays (see bcfgpu_tile)."""

    def __init__(self, n_smpl, ref16, plp_off, rd, epos, aux=None, is_indel=0):
        self.n_smpl = int(n_smpl)
        self.ref16 = np.ascontiguousarray(ref16, dtype=np.int8)
        self.plp_off = np.ascontiguousarray(plp_off, dtype=np.uint32)
        self.rd = np.ascontiguousarray(rd, dtype=np.uint32)
        self.epos = np.ascontiguousarray(epos, dtype=np.uint8)
        self.aux = None if aux is None else np.ascontiguousarray(aux, dtype=np.uint32)
        self.is_indel = int(is_indel)
        self.n_sites = len(self.ref16)
        assert len(self.plp_off) == self.n_sites * self.n_smpl + 1
        assert len(self.rd) == len(self.epos) == int(self.plp_off[-1])

    def as_struct(self):
        t = abi.Tile()
        t.n_sites, t.is_indel, t.n_reads = self.n_sites, self.is_indel, len(self.rd)
        t.ref16, t.plp_off, t.rd, t.epos, t.aux = _p(self.ref16), _p(self.plp_off), _p(self.rd), _p(self.epos), _p(self.aux)
        return t


class MplpResult:
    """Host copy of bcfgpu_mplp_out (same plane layout as the device)."""

    def __init__(self, n_sites, n_smpl):
        self.n_sites, self.n_smpl = n_sites, n_smpl
        self.site = np.zeros(n_sites, dtype=SITE_DTYPE)
        self.pl = np.zeros((n_sites, abi.MAX_PL, n_smpl), dtype=np.uint8)
        self.dp4 = np.zeros((n_sites, 4, n_smpl), dtype=np.uint8)
        self.adf = np.zeros((n_sites, 5, n_smpl), dtype=np.uint8)
        self.adr = np.zeros((n_sites, 5, n_smpl), dtype=np.uint8)
        self.qs = np.zeros((n_sites, 5, n_smpl), dtype=np.uint16)
        self.scr = np.zeros((n_sites, n_smpl), dtype=np.uint8)

    def as_struct(self):
        o = abi.MplpOut()
        o.site, o.pl, o.dp4, o.adf, o.adr, o.qs, o.scr = (_p(self.site), _p(self.pl), _p(self.dp4), _p(self.adf),
                                                           _p(self.adr), _p(self.qs), _p(self.scr))
        return o

    def pl_of(self, isite):
        """PL in the reference's layout: int32 [n_smpl][x] (bam2bcf.c:636-648)."""
        na = int(self.site["n_alleles"][isite])
        x = na * (na + 1) // 2
        return self.pl[isite, :x, :].T.astype(np.int32)


def mpileup(cfg, tile, want_callret=False):
    """Run the oracle's mpileup stage on a HostTile."""
    res = MplpResult(tile.n_sites, tile.n_smpl)
    t, o = tile.as_struct(), res.as_struct()
    cr = np.zeros(tile.n_sites * tile.n_smpl, dtype=CALLRET_DTYPE) if want_callret else None
    rc = lib().orc_mpileup(C.byref(cfg), C.byref(t), C.byref(o), _p(cr))
    if rc != 0:
        raise RuntimeError("orc_mpileup failed: %d" % rc)
    return (res, cr) if want_callret else res


class CallInput:
    """Host copy of bcfgpu_call_in."""

    def __init__(self, n_smpl, nals, unseen, pl, qs, ad=None, ploidy=None, grp=None, prior_an=None, prior_ac=None):
        self.n_smpl = n_smpl
        self.nals = np.ascontiguousarray(nals, dtype=np.int32)
        self.unseen = np.ascontiguousarray(unseen, dtype=np.int32)
        self.pl = np.ascontiguousarray(pl, dtype=np.int32)        # [site][n_gt_max][n_smpl]
        self.qs = np.ascontiguousarray(qs, dtype=np.float32)      # [site][5]
        self.ad = None if ad is None else np.ascontiguousarray(ad, dtype=np.int32)
        self.ploidy = None if ploidy is None else np.ascontiguousarray(ploidy, dtype=np.uint8)
        self.grp = None if grp is None else np.ascontiguousarray(grp, dtype=np.int32)
        self.prior_an = None if prior_an is None else np.ascontiguousarray(prior_an, dtype=np.int32)
        self.prior_ac = None if prior_ac is None else np.ascontiguousarray(prior_ac, dtype=np.int32)
        self.n_sites = len(self.nals)
        self.n_gt_max = self.pl.shape[1]
        self.n_al_max = 0 if self.ad is None else self.ad.shape[1]

    def as_struct(self):
        s = abi.CallIn()
        s.n_sites, s.n_gt_max, s.n_al_max = self.n_sites, self.n_gt_max, self.n_al_max
        s.nals, s.unseen, s.pl, s.qs, s.ad = _p(self.nals), _p(self.unseen), _p(self.pl), _p(self.qs), _p(self.ad)
        s.ploidy, s.grp, s.prior_an, s.prior_ac = _p(self.ploidy), _p(self.grp), _p(self.prior_an), _p(self.prior_ac)
        return s


class CallResult:
    def __init__(self, n_sites, n_smpl, n_gt_max):
        self.site = np.zeros(n_sites, dtype=CALLSITE_DTYPE)
        self.gt = np.zeros((n_sites, 2, n_smpl), dtype=np.int8)
        self.pl = np.zeros((n_sites, n_gt_max, n_smpl), dtype=np.int32)
        self.gq = np.zeros((n_sites, n_smpl), dtype=np.int32)
        self.gp = np.zeros((n_sites, n_gt_max, n_smpl), dtype=np.float32)

    def as_struct(self):
        o = abi.CallOut()
        o.site, o.gt, o.pl, o.gq, o.gp = _p(self.site), _p(self.gt), _p(self.pl), _p(self.gq), _p(self.gp)
        return o


def mcall(cfg, cin):
    res = CallResult(cin.n_sites, cin.n_smpl, cin.n_gt_max)
    i, o = cin.as_struct(), res.as_struct()
    rc = lib().orc_mcall(C.byref(cfg), C.byref(i), C.byref(o))
    if rc != 0:
        raise RuntimeError("orc_mcall failed: %d" % rc)
    return res

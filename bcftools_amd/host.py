"""Host-side (numpy) containers mirroring the SoA structs of include/bcfgpu.h.

HostTile      <-> bcfgpu_tile      (one batch of pileup columns: site x sample x read CSR)
MplpResult    <-> bcfgpu_mplp_out  (per-site bcf_call_t scalars + per-sample planes)
CallInput     <-> bcfgpu_call_in   (what mcall() reads from a record)
CallResult    <-> bcfgpu_call_out
"""
import ctypes as C
import numpy as np

from . import abi

SITE_DTYPE = np.dtype([("a", "<i4", 5), ("n_alleles", "<i4"), ("unseen", "<i4"), ("ori_ref", "<i4"),
                       ("shift", "<i4"), ("ret", "<i4"), ("depth", "<u4"), ("ori_depth", "<u4"), ("mq0", "<u4"),
                       ("qsum", "<f4", 5), ("vdb", "<f4"), ("mwu_pos", "<f4"), ("mwu_mq", "<f4"),
                       ("mwu_bq", "<f4"), ("mwu_mqs", "<f4"), ("seg_bias", "<f4"),
                       ("adf_tot", "<i4", 5), ("adr_tot", "<i4", 5), ("scr_tot", "<i4"), ("pad", "<i4"),
                       ("anno", "<f8", 16)], align=True)

CALLSITE_DTYPE = np.dtype([("ret", "<i4"), ("nals_new", "<i4"), ("als_new", "<i4"), ("als_map", "<i4", 5),
                           ("ac", "<i4", 5), ("an", "<i4"), ("qual_missing", "<i4"), ("qual", "<f4"),
                           ("pl_dropped", "<i4"), ("has_i16", "<i4"), ("dp4", "<i4", 4), ("mq", "<i4"),
                           ("pv4_tested", "<i4"), ("pv4", "<f4", 4)], align=True)

GVCF_BLOCK_DTYPE = np.dtype([(k, "<i4") for k in ("first_site", "last_site", "start_pos", "end1", "min_dp", "range")])
assert GVCF_BLOCK_DTYPE.itemsize == C.sizeof(abi.GvcfBlock)

assert SITE_DTYPE.itemsize == C.sizeof(abi.Site), (SITE_DTYPE.itemsize, C.sizeof(abi.Site))
assert CALLSITE_DTYPE.itemsize == C.sizeof(abi.CallSite)



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

    def select_sites(self, sites):
        """A new tile holding the given sites (any order), e.g. a region shard or a spot-check sample."""
        S = self.n_smpl
        sites = np.asarray(sites, dtype=np.int64)
        cells = (sites[:, None] * S + np.arange(S)[None, :]).ravel()
        beg, end = self.plp_off[cells].astype(np.int64), self.plp_off[cells + 1].astype(np.int64)
        n = end - beg
        off = np.zeros(len(cells) + 1, dtype=np.int64)
        np.cumsum(n, out=off[1:])
        idx = np.repeat(beg - off[:-1], n) + np.arange(int(off[-1]))
        return HostTile(S, self.ref16[sites], off.astype(np.uint32), self.rd[idx], self.epos[idx],
                        None if self.aux is None else self.aux[idx], self.is_indel)

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
        self.dp4 = np.zeros((n_sites, 4, n_smpl), dtype=np.uint16)
        self.adf = np.zeros((n_sites, 5, n_smpl), dtype=np.uint16)
        self.adr = np.zeros((n_sites, 5, n_smpl), dtype=np.uint16)
        self.qs = np.zeros((n_sites, 5, n_smpl), dtype=np.int32)
        self.scr = np.zeros((n_sites, n_smpl), dtype=np.uint16)
        self.sp = np.zeros((n_sites, n_smpl), dtype=np.uint8)

    def as_struct(self):
        o = abi.MplpOut()
        o.site, o.pl, o.dp4, o.adf, o.adr, o.qs, o.scr, o.sp = (_p(self.site), _p(self.pl), _p(self.dp4), _p(self.adf),
                                                                 _p(self.adr), _p(self.qs), _p(self.scr), _p(self.sp))
        return o

    def pl_of(self, isite):
        """PL in the reference's layout: int32 [n_smpl][x] (bam2bcf.c:636-648)."""
        na = int(self.site["n_alleles"][isite])
        x = na * (na + 1) // 2
        return self.pl[isite, :x, :].T.astype(np.int32)



class GvcfResult:
    """Host copy of bcfgpu_gvcf_out, cut to the blocks found (PL widened to int32)."""

    def __init__(self, n_blocks, blk, min_dp, block, dp, pl):
        self.n_blocks, self.blk, self.min_dp = n_blocks, blk, min_dp
        self.block, self.dp, self.pl = block[:n_blocks], dp[:n_blocks], pl[:n_blocks].astype(np.int32)


class CallInput:
    """Host copy of bcfgpu_call_in."""

    def __init__(self, n_smpl, nals, unseen, pl, qs, ad=None, ploidy=None, grp=None, prior_an=None, prior_ac=None, i16=None):
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
        self.i16 = None if i16 is None else np.ascontiguousarray(i16, dtype=np.float32)     # [site][16]
        self.n_sites = len(self.nals)
        self.n_gt_max = self.pl.shape[1]
        self.n_al_max = 0 if self.ad is None else self.ad.shape[1]

    def as_struct(self):
        s = abi.CallIn()
        s.n_sites, s.n_gt_max, s.n_al_max = self.n_sites, self.n_gt_max, self.n_al_max
        s.nals, s.unseen, s.pl, s.qs, s.ad = _p(self.nals), _p(self.unseen), _p(self.pl), _p(self.qs), _p(self.ad)
        s.ploidy, s.grp, s.prior_an, s.prior_ac = _p(self.ploidy), _p(self.grp), _p(self.prior_an), _p(self.prior_ac)
        s.i16 = _p(self.i16)
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



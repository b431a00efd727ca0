"""Host-side mirror of the reference's operator interface for the hot path, over the C-ABI.

    Context(cfg)                      ~ bcf_call_init + mcall_init   (bam2bcf.h:135, call.h:135)
    Context.mpileup(tile)             ~ bcf_call_glfgen x n_smpl + bcf_call_combine per site (mpileup.c:343-347)
    Context.mcall(call_input)         ~ mcall() per record (vcfcall.c:1137)
    Context.pipeline(tile, ...)       ~ `mpileup -Ou | call -m` with the PL/QS/I16 hand-off kept in HBM
    Context.close()                   ~ bcf_call_destroy + mcall_destroy

Inputs/outputs are the numpy containers of bcftools_amd.host; device memory is managed
through bcfgpu_malloc/memcpy so no other GPU runtime is required.  Every method raises
BcfGpuError on failure -- nothing here computes on the CPU.
"""
import ctypes as C
import numpy as np

from . import abi, host
from .lib import load, check, BcfGpuError  # noqa: F401


class DevBuf:
    """A device allocation owned by a Context."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        check(ctx.L.bcfgpu_malloc(ctx.h, max(self.nbytes, 16), C.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(self.ctx.L.bcfgpu_memcpy_h2d(self.ctx.h, self.ptr, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        return self

    def download(self, arr):
        assert arr.flags["C_CONTIGUOUS"] and arr.nbytes <= self.nbytes
        check(self.ctx.L.bcfgpu_memcpy_d2h(self.ctx.h, arr.ctypes.data_as(C.c_void_p), self.ptr, arr.nbytes))
        return arr

    def free(self):
        if self.ptr:
            self.ctx.L.bcfgpu_free(self.ctx.h, self.ptr)
            self.ptr = None


class Context:
    def __init__(self, cfg):
        self.L = load()
        self.cfg = cfg
        h = C.c_void_p()
        check(self.L.bcfgpu_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self._bufs = []

    # -- memory ------------------------------------------------------------------------
    def buf(self, nbytes):
        b = DevBuf(self, nbytes)
        self._bufs.append(b)
        return b

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return self.buf(arr.nbytes).upload(arr)

    def release(self, bufs):
        for b in bufs:
            b.free()
            if b in self._bufs:
                self._bufs.remove(b)

    def sync(self):
        check(self.L.bcfgpu_sync(self.h))

    def close(self):
        if self.h:
            for b in self._bufs:
                b.free()
            self._bufs = []
            self.L.bcfgpu_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- tiles -------------------------------------------------------------------------
    def upload_tile(self, t):
        """HostTile -> (abi.Tile with device pointers, [DevBuf])"""
        bufs = [self.to_device(t.ref16), self.to_device(t.plp_off), self.to_device(t.rd), self.to_device(t.epos)]
        d = abi.Tile()
        d.n_sites, d.is_indel, d.n_reads = t.n_sites, t.is_indel, len(t.rd)
        d.ref16, d.plp_off, d.rd, d.epos = bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr
        if t.aux is not None:
            bufs.append(self.to_device(t.aux))
            d.aux = bufs[-1].ptr
        return d, bufs

    def flagged_planes(self):
        """The planes of bcfgpu_mplp_out the context's fmt_flag / grouping asks for: the others may be NULL (mpileup.c:612-636
        allocates ADF/ADR/SCR only under their flags) and are then neither computed nor written."""
        f, names = self.cfg.fmt_flag, ["site", "pl", "dp4"]
        grp_ad = self.cfg.n_grp > 1 and not self.cfg.grp_tag_is_qs
        if grp_ad or f & (abi.FMT_AD | abi.FMT_ADF | abi.FMT_ADR | abi.FMT_DPR | abi.INFO_AD | abi.INFO_ADF | abi.INFO_ADR | abi.INFO_DPR):
            names += ["adf", "adr"]
        if (self.cfg.n_grp > 1 and self.cfg.grp_tag_is_qs) or f & abi.FMT_QS:
            names.append("qs")
        if f & (abi.FMT_SCR | abi.INFO_SCR):
            names.append("scr")
        if f & abi.FMT_SP:
            names.append("sp")
        return names

    def alloc_mplp_out(self, n_sites, names=None):
        S = self.cfg.n_smpl
        res = host.MplpResult(n_sites, S)
        if names is None:
            names = ["site", "pl", "dp4", "adf", "adr", "qs", "scr", "sp"]
        bufs = {k: self.buf(getattr(res, k).nbytes) for k in names}
        o = abi.MplpOut()
        for k in names:
            setattr(o, k, bufs[k].ptr)
        return o, bufs, res

    def alloc_call_out(self, n_sites, n_gt_max):
        S = self.cfg.n_smpl
        res = host.CallResult(n_sites, S, n_gt_max)
        names = ["site", "gt", "pl", "gq", "gp"]
        bufs = {k: self.buf(getattr(res, k).nbytes) for k in names}
        o = abi.CallOut()
        for k in names:
            setattr(o, k, bufs[k].ptr)
        return o, bufs, res

    @staticmethod
    def _download(bufs, res):
        for k, b in bufs.items():
            b.download(getattr(res, k))
        return res

    # -- the hot path --------------------------------------------------------------------
    def mpileup(self, tile):
        """Run glfgen+combine on a HostTile, return a host MplpResult."""
        assert tile.n_smpl == self.cfg.n_smpl
        dt, tb = self.upload_tile(tile)
        o, ob, res = self.alloc_mplp_out(tile.n_sites)
        # planes the kernels skip for unused alleles must read as zero, like the oracle's
        for b in ob.values():
            check(self.L.bcfgpu_memset(self.h, b.ptr, 0, b.nbytes))
        try:
            check(self.L.bcfgpu_mpileup(self.h, C.byref(dt), C.byref(o)))
            self.sync()
            self._download(ob, res)
        finally:
            self.release(tb + list(ob.values()))
        return res

    def mpileup_planned(self, snp, indel=None, cols=None, ret=None, visit=None):
        """bcfgpu_errmod_plan over the two passes of a tile, then bcfgpu_mpileup of each: the over-deep cells' likelihoods come from
        errmod_cal's own draw (hts_drand48, the context's generator), in mpileup_reg()'s visit order.  snp / indel: HostTiles (indel
        may be None); cols: the SNP-tile column of every indel site; ret: bcf_call_gap_prep's return per indel site (None: all 0).
        Returns (MplpResult of the SNP pass, MplpResult of the indel pass or None)."""
        ds, sb = self.upload_tile(snp)
        di, ib = (self.upload_tile(indel) if indel is not None else (None, []))
        outs = []
        try:
            c = None if cols is None else np.ascontiguousarray(cols, dtype=np.int32)
            r = None if ret is None else np.ascontiguousarray(ret, dtype=np.int32)
            if visit is None:
                check(self.L.bcfgpu_errmod_plan(self.h, C.byref(ds), C.byref(di) if di is not None else None,
                                                None if c is None else c.ctypes.data, None if r is None else r.ctypes.data))
            else:                                            # visit: 0 = mpileup_reg() passes the SNP-tile column over (outside the targets)
                v = np.ascontiguousarray(visit, dtype=np.uint8)
                check(self.L.bcfgpu_errmod_plan_visit(self.h, C.byref(ds), v.ctypes.data, C.byref(di) if di is not None else None,
                                                      None if c is None else c.ctypes.data, None if r is None else r.ctypes.data))
            for dt, t in ((ds, snp), (di, indel)):
                if dt is None:
                    outs.append(None)
                    continue
                o, ob, res = self.alloc_mplp_out(t.n_sites)
                for b in ob.values():
                    check(self.L.bcfgpu_memset(self.h, b.ptr, 0, b.nbytes))
                check(self.L.bcfgpu_mpileup(self.h, C.byref(dt), C.byref(o)))
                self.sync()
                self._download(ob, res)
                self.release(list(ob.values()))
                outs.append(res)
        finally:
            self.release(sb + ib)
        return outs[0], outs[1]

    def gvcf_blocks(self, res, pos, dp_range, rid=None, brk=None):
        """gvcf_write over the records of a host MplpResult (bcfgpu_gvcf_blocks); returns a host GvcfResult."""
        n, S = res.n_sites, self.cfg.n_smpl
        rng = np.ascontiguousarray(dp_range, dtype=np.int32)
        bufs = [self.to_device(np.ascontiguousarray(pos, dtype=np.int32)), self.to_device(res.site), self.to_device(res.pl),
                self.to_device(res.dp4)]
        gi = abi.GvcfIn()
        gi.n_sites, gi.n_range, gi.dp_range = n, len(rng), rng.ctypes.data_as(C.c_void_p)
        gi.pos, gi.site, gi.pl, gi.dp4 = (b.ptr for b in bufs)
        if rid is not None:
            bufs.append(self.to_device(np.ascontiguousarray(rid, dtype=np.int32)))
            gi.rid = bufs[-1].ptr
        if brk is not None:
            bufs.append(self.to_device(np.ascontiguousarray(brk, dtype=np.uint8)))
            gi.brk = bufs[-1].ptr
        blk, min_dp = np.zeros(n, np.int32), np.zeros(n, np.int32)
        block = np.zeros(n, host.GVCF_BLOCK_DTYPE)
        dp, pl = np.zeros((n, S), np.int32), np.zeros((n, 3, S), np.uint8)
        outs = [(a, self.buf(a.nbytes)) for a in (blk, min_dp, block, dp, pl)]
        go = abi.GvcfOut()
        go.blk, go.min_dp, go.block, go.dp, go.pl = (b.ptr for _, b in outs)
        nb = C.c_int32(0)
        try:
            check(self.L.bcfgpu_gvcf_blocks(self.h, C.byref(gi), C.byref(go), C.byref(nb)))
            self.sync()
            for a, b in outs:
                b.download(a)
        finally:
            self.release(bufs + [b for _, b in outs])
        return host.GvcfResult(nb.value, blk, min_dp, block, dp, pl)

    def gvcf_call_blocks(self, pos, ref_only, dp, dp_range, rid=None, end=None):
        """The `call -g` form of bcfgpu_gvcf_blocks: records given by position, "may join" flag and FORMAT/DP [n][n_smpl].
        Returns (n_blocks, blk, min_dp, block table, block DP)."""
        dp = np.ascontiguousarray(dp, dtype=np.int32)
        n, S = dp.shape
        assert S == self.cfg.n_smpl
        rng = np.ascontiguousarray(dp_range, dtype=np.int32)
        bufs = [self.to_device(np.ascontiguousarray(pos, dtype=np.int32)), self.to_device(np.ascontiguousarray(ref_only, dtype=np.uint8)),
                self.to_device(dp)]
        gi = abi.GvcfIn()
        gi.n_sites, gi.n_range, gi.dp_range = n, len(rng), rng.ctypes.data_as(C.c_void_p)
        gi.pos, gi.ref_only, gi.dp = (b.ptr for b in bufs)
        if rid is not None:
            bufs.append(self.to_device(np.ascontiguousarray(rid, dtype=np.int32)))
            gi.rid = bufs[-1].ptr
        if end is not None:
            bufs.append(self.to_device(np.ascontiguousarray(end, dtype=np.int32)))
            gi.end = bufs[-1].ptr
        blk, min_dp = np.zeros(n, np.int32), np.zeros(n, np.int32)
        block = np.zeros(max(n, 1), host.GVCF_BLOCK_DTYPE)
        bdp = np.zeros((max(n, 1), S), np.int32)
        outs = [(a, self.buf(a.nbytes)) for a in (blk, min_dp, block, bdp)]
        go = abi.GvcfOut()
        go.blk, go.min_dp, go.block, go.dp = (b.ptr for _, b in outs)
        nb = C.c_int32(0)
        try:
            check(self.L.bcfgpu_gvcf_blocks(self.h, C.byref(gi), C.byref(go), C.byref(nb)))
            self.sync()
            for a, b in outs:
                b.download(a)
        finally:
            self.release(bufs + [b for _, b in outs])
        return nb.value, blk, min_dp, block, bdp

    def mcall(self, cin):
        """Run the caller on a host CallInput, return a host CallResult."""
        assert cin.n_smpl == self.cfg.n_smpl
        keep = []
        d = abi.CallIn()
        d.n_sites, d.n_gt_max, d.n_al_max = cin.n_sites, cin.n_gt_max, cin.n_al_max
        for k in ("nals", "unseen", "pl", "qs", "ad", "ploidy", "grp", "prior_an", "prior_ac", "i16"):
            a = getattr(cin, k)
            if a is not None:
                b = self.to_device(a)
                keep.append(b)
                setattr(d, k, b.ptr)
        o, ob, res = self.alloc_call_out(cin.n_sites, cin.n_gt_max)
        for b in ob.values():
            check(self.L.bcfgpu_memset(self.h, b.ptr, 0, b.nbytes))
        try:
            check(self.L.bcfgpu_mcall(self.h, C.byref(d), C.byref(o)))
            self.sync()
            self._download(ob, res)
        finally:
            self.release(keep + list(ob.values()))
        return res

    def pipeline(self, tile, ploidy=None, grp=None):
        """mpileup stage + call stage on a SNP HostTile with the hand-off kept on the device.
        Returns (MplpResult, CallResult)."""
        dt, tb = self.upload_tile(tile)
        mo, mb, mres = self.alloc_mplp_out(tile.n_sites)
        co, cb, cres = self.alloc_call_out(tile.n_sites, abi.MAX_PL)
        keep = []
        pp = gp = None
        if ploidy is not None:
            keep.append(self.to_device(np.ascontiguousarray(ploidy, dtype=np.uint8)))
            pp = keep[-1].ptr
        if grp is not None:
            keep.append(self.to_device(np.ascontiguousarray(grp, dtype=np.int32)))
            gp = keep[-1].ptr
        for b in list(mb.values()) + list(cb.values()):
            check(self.L.bcfgpu_memset(self.h, b.ptr, 0, b.nbytes))
        try:
            check(self.L.bcfgpu_pipeline(self.h, C.byref(dt), pp, gp, C.byref(mo), C.byref(co)))
            self.sync()
            self._download(mb, mres)
            self._download(cb, cres)
        finally:
            self.release(tb + keep + list(mb.values()) + list(cb.values()))
        return mres, cres

    def timing(self, on=True):
        check(self.L.bcfgpu_timing_enable(self.h, 1 if on else 0))

    def last_timing(self):
        t = abi.Timing()
        check(self.L.bcfgpu_timing_get(self.h, C.byref(t)))
        return dict(glfgen_ms=t.glfgen_ms, combine_ms=t.combine_ms, mcall_ms=t.mcall_ms, total_ms=t.total_ms)

"""bcf_call_gap_prep on a synthetic batch (synth.indel_batch): through bcfgpu_gap_prep (one call for the whole batch)
and through the oracle (orc_gap_prep, one site at a time)."""
import ctypes as C

import numpy as np

from bcftools_amd import abi
from tests.helpers import mplpdrv

CAP = 64
READ_KEYS = ("r_pos", "r_lq", "r_flag", "r_ncig", "r_cig_off", "r_seq_off", "cig", "seq16", "qual", "zq", "r_has_zq")
DEFAULTS = dict(openQ=40, extQ=20, tandemQ=100, min_support=1, min_frac=0.002, per_sample_flt=0)   # mpileup.c:937-950


def gap_prep_gpu(ctx, b, **kw):
    """Returns (dict of output arrays, abi.GapStats)."""
    from bcftools_amd.lib import check
    o = dict(DEFAULTS, **kw)
    ns, E = b["n_sites"], len(b["p_read"])
    rd = abi.Reads()
    rd.n_reads = b["reads"]["n_reads"]
    for k in READ_KEYS:
        setattr(rd, k, b["reads"][k].ctypes.data)
    ii = abi.IndelIn()
    ii.n_sites, ii.n_smpl = ns, b["n_smpl"]
    ii.pos, ii.smpl_off, ii.p_read, ii.p_qpos, ii.p_indel = (b["pos"].ctypes.data, b["smpl_off"].ctypes.data, b["p_read"].ctypes.data,
                                                             b["p_qpos"].ctypes.data, b["p_indel"].ctypes.data)
    ii.ref = b["ref"]
    ii.openQ, ii.extQ, ii.tandemQ, ii.min_support, ii.per_sample_flt, ii.min_frac = (o["openQ"], o["extQ"], o["tandemQ"],
                                                                                      o["min_support"], o["per_sample_flt"], o["min_frac"])
    out = dict(ret=np.zeros(ns, np.int32), aux=np.zeros(E, np.uint32), indel_types=np.zeros((ns, 4), np.int32),
               inscns=np.zeros((ns, 4 * CAP), np.int8), maxins=np.zeros(ns, np.int32), indelreg=np.zeros(ns, np.int32),
               max_support=np.zeros(ns, np.int32), max_frac=np.zeros(ns, np.float32))
    oo = abi.IndelOut()
    oo.ret, oo.p_aux, oo.indel_types, oo.inscns = (out["ret"].ctypes.data, out["aux"].ctypes.data, out["indel_types"].ctypes.data,
                                                  out["inscns"].ctypes.data)
    oo.maxins, oo.indelreg, oo.max_support, oo.max_frac = (out["maxins"].ctypes.data, out["indelreg"].ctypes.data,
                                                           out["max_support"].ctypes.data, out["max_frac"].ctypes.data)
    check(ctx.L.bcfgpu_gap_prep(ctx.h, C.byref(rd), C.byref(ii), C.byref(oo), CAP))
    st = abi.GapStats()
    check(ctx.L.bcfgpu_gap_prep_stats(ctx.h, C.byref(st)))
    return out, st


class DevicePool:
    """A synthetic batch's reads as a pool in HBM: bcfgpu_pileup over the batch's whole reference (done once), after which
    bcfgpu_gap_prep_tile runs on the candidate columns with nothing crossing PCIe but the per-column results."""

    def __init__(self, ctx, b):
        from bcftools_amd import synth
        from bcftools_amd.lib import check
        self.ctx, self.b = ctx, b
        self.reads, mapq, smpl, self.order = synth.indel_pool(b)
        self.rd = abi.Reads()
        self.rd.n_reads = self.reads["n_reads"]
        for k in READ_KEYS:
            setattr(self.rd, k, self.reads[k].ctypes.data)
        self.beg, self.end = 0, len(b["ref"])
        self.tile = abi.Tile()
        self.col_n = np.zeros(self.end - self.beg, np.int32)
        self.col_indel = np.zeros(self.end - self.beg, np.uint8)
        check(ctx.L.bcfgpu_pileup(ctx.h, C.byref(self.rd), mapq.ctypes.data, smpl.ctypes.data, self.beg, self.end, b["ref"], len(b["ref"]),
                                  C.byref(self.tile), self.col_n.ctypes.data, self.col_indel.ctypes.data))
        self.cols = np.ascontiguousarray(b["pos"] - self.beg, dtype=np.int32)

    def gap_prep_tile(self, want_aux=False, **kw):
        """Returns (dict of output arrays, abi.GapStats, abi.Tile): bcfgpu_gap_prep_tile over the batch's columns."""
        from bcftools_amd.lib import check
        o = dict(DEFAULTS, **kw)
        ctx, b = self.ctx, self.b
        ns = b["n_sites"]
        par = abi.IndelIn()
        par.ref = b["ref"]
        for k, v in o.items():
            setattr(par, k, v)
        E = int(self.col_n[self.cols].sum())
        out = dict(ret=np.zeros(ns, np.int32), aux=np.zeros(E if want_aux else 0, np.uint32), indel_types=np.zeros((ns, 4), np.int32),
                   inscns=np.zeros((ns, 4 * CAP), np.int8), maxins=np.zeros(ns, np.int32), indelreg=np.zeros(ns, np.int32),
                   max_support=np.zeros(ns, np.int32), max_frac=np.zeros(ns, np.float32))
        oo = abi.IndelOut()
        oo.ret, oo.indel_types, oo.inscns = out["ret"].ctypes.data, out["indel_types"].ctypes.data, out["inscns"].ctypes.data
        oo.p_aux = out["aux"].ctypes.data if want_aux else None
        oo.maxins, oo.indelreg, oo.max_support, oo.max_frac = (out["maxins"].ctypes.data, out["indelreg"].ctypes.data,
                                                               out["max_support"].ctypes.data, out["max_frac"].ctypes.data)
        t = abi.Tile()
        check(ctx.L.bcfgpu_gap_prep_tile(ctx.h, ns, self.cols.ctypes.data, None, C.byref(par), C.byref(oo), CAP, C.byref(t)))
        st = abi.GapStats()
        check(ctx.L.bcfgpu_gap_prep_stats(ctx.h, C.byref(st)))
        return out, st, t


def gap_prep_oracle_site(b, k, **kw):
    """orc_gap_prep for site k of the batch; returns None (ret<0) or dict like one row of gap_prep_gpu's output."""
    o = dict(DEFAULTS, **kw)
    L = mplpdrv._realn_lib()
    S = b["n_smpl"]
    r = b["reads"]
    soff = np.ascontiguousarray(b["smpl_off"][k * S:(k + 1) * S + 1])
    e0, e1 = int(soff[0]), int(soff[-1])
    soff = soff - e0
    sl = lambda a: np.ascontiguousarray(a[e0:e1])
    p_read, p_qpos, p_indel = sl(b["p_read"]), sl(b["p_qpos"]), sl(b["p_indel"])
    aux = np.zeros(e1 - e0, np.uint32)
    types = np.zeros(4, np.int32)
    inscns = np.zeros(4 * CAP, np.int8)
    maxins, indelreg, msup, mfrac = C.c_int(), C.c_int(), C.c_int(), C.c_float()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L.orc_gap_prep.restype = C.c_int
    L.orc_gap_prep.argtypes = [C.c_int] + [C.c_void_p] * 15 + [C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                               C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = L.orc_gap_prep(S, p(soff), p(p_read), p(p_qpos), p(p_indel), p(r["r_pos"]), p(r["r_lq"]), p(r["r_flag"]), p(r["r_ncig"]),
                        p(r["r_cig_off"]), p(r["cig"]), p(r["r_seq_off"]), p(r["seq16"]), p(r["qual"]), p(r["zq"]), p(r["r_has_zq"]),
                        int(b["pos"][k]), b["ref"], o["openQ"], o["extQ"], o["tandemQ"], o["min_support"], o["min_frac"],
                        o["per_sample_flt"], p(aux), p(types), p(inscns), len(inscns), C.byref(maxins), C.byref(indelreg),
                        C.byref(msup), C.byref(mfrac))
    if rc < 0:
        return None
    return dict(aux=aux, e0=e0, e1=e1, indel_types=types, inscns=inscns, maxins=maxins.value, indelreg=indelreg.value,
                max_support=msup.value, max_frac=float(mfrac.value))


def assert_site_equal(got, k, want):
    """Row k of the batched HIP result against the oracle's result for that site."""
    if want is None:
        assert got["ret"][k] < 0
        return
    assert got["ret"][k] == 0
    np.testing.assert_array_equal(got["aux"][want["e0"]:want["e1"]], want["aux"], err_msg="p->aux site %d" % k)
    np.testing.assert_array_equal(got["indel_types"][k], want["indel_types"])
    assert (got["maxins"][k], got["indelreg"][k], got["max_support"][k]) == (want["maxins"], want["indelreg"], want["max_support"])
    assert got["max_frac"][k] == np.float32(want["max_frac"])
    n = 4 * want["maxins"]
    np.testing.assert_array_equal(got["inscns"][k][:n], want["inscns"][:n])


def assert_tile_matches_host_batch(ctx, b, pool, got_t, tile, got):
    """bcfgpu_gap_prep_tile's outputs (got_t, tile: the pool form) against bcfgpu_gap_prep's on the same batch (got: host pointers,
    entries in the batch's order).  The tile holds the columns with ret == 0 and only those, in order (mpileup.c:354-360); its
    entries come in pool order (position-sorted per sample), so p->aux is compared read by read."""
    from bcftools_amd.lib import check
    S, ns = b["n_smpl"], b["n_sites"]
    np.testing.assert_array_equal(got_t["ret"], got["ret"])
    ok = got["ret"] == 0          # (what bca holds after a call that returned -1 is read by no one: mpileup.c:354-364)
    for key in ("indel_types", "inscns", "maxins", "indelreg", "max_support", "max_frac"):
        np.testing.assert_array_equal(got_t[key][ok], got[key][ok], err_msg=key)
    assert (got_t["indel_types"][~ok] == 10000).all()
    live = np.nonzero(ok)[0]
    assert tile.n_sites == len(live) and tile.is_indel == 1
    full = np.diff(b["smpl_off"].astype(np.int64))                       # entries per (column, sample) cell of the batch
    want_cnt = full.reshape(ns, S)[live].ravel()
    assert tile.n_reads == int(want_cnt.sum())
    if len(live) == 0:
        return
    off = np.zeros(len(live) * S + 1, np.uint32)
    check(ctx.L.bcfgpu_memcpy_d2h(ctx.h, off.ctypes.data, tile.plp_off, off.nbytes))
    np.testing.assert_array_equal(np.diff(off.astype(np.int64)), want_cnt, err_msg="tile.plp_off")
    cell = np.repeat(np.arange(ns * S), full)
    dev2batch = pool.order[np.argsort(cell[pool.order], kind="stable")]  # the batch's entries, column-major, each cell in pool order
    keep = np.repeat(np.repeat(ok, S), full)
    if got_t["aux"].size:
        np.testing.assert_array_equal(got_t["aux"][:tile.n_reads], got["aux"][dev2batch][keep], err_msg="p->aux")
    aux_d = np.zeros(tile.n_reads, np.uint32)
    check(ctx.L.bcfgpu_memcpy_d2h(ctx.h, aux_d.ctypes.data, tile.aux, aux_d.nbytes))
    np.testing.assert_array_equal(aux_d, got["aux"][dev2batch][keep], err_msg="tile.aux")

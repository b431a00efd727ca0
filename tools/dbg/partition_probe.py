"""Debug: which p->aux words of the indel tile depend on how the candidate list is cut, and which cut agrees with the oracle."""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, ".")
from bcftools_amd import abi, synth, engine
from bcftools_amd.lib import check
from tests.helpers import indeldrv

S, n_sites = 1000, 2048
W = synth.wgs_reads(20260106, n_sites, S, 30.0)
arrs, mapq, smpl, n, L, beg, end = W["reads"], W["mapq"], W["smpl"], W["n_reads"], W["read_len"], W["beg"], W["end"]
ref_b = W["refseq"].encode()
rd = abi.Reads(); rd.n_reads = n
for k, v in arrs.items():
    setattr(rd, k, v.ctypes.data)
c0 = engine.Context(abi.default_cfg(S, max_sites=1, max_reads=64))
t = abi.Tile()
check(c0.L.bcfgpu_pileup(c0.h, C.byref(rd), mapq.ctypes.data, smpl.ctypes.data, beg, end, ref_b, len(ref_b), C.byref(t), None, None))
entries = int(t.n_reads); c0.close()
ctx = engine.Context(abi.default_cfg(S, max_sites=n_sites, max_reads=entries + 64))
Lb = ctx.L
tile = abi.Tile()
col_n, col_indel = np.zeros(n_sites, np.int32), np.zeros(n_sites, np.uint8)
check(Lb.bcfgpu_pool_upload(ctx.h, C.byref(rd), None, mapq.ctypes.data))
check(Lb.bcfgpu_pool_pileup(ctx.h, smpl.ctypes.data, None, beg, end, ref_b, len(ref_b), C.byref(tile), col_n.ctypes.data, col_indel.ctypes.data))
cand = np.ascontiguousarray(np.nonzero((col_indel != 0) & (col_n < 250 * S))[0], dtype=np.int32)
par = abi.IndelIn(); par.ref = ref_b
for k, v in indeldrv.DEFAULTS.items():
    setattr(par, k, v)
CAP = indeldrv.CAP

def run(cols):
    nc = len(cols)
    out = dict(ret=np.zeros(nc, np.int32), indel_types=np.zeros((nc, 4), np.int32), inscns=np.zeros((nc, 4 * CAP), np.int8))
    oo = abi.IndelOut()
    oo.ret, oo.indel_types, oo.inscns = out["ret"].ctypes.data, out["indel_types"].ctypes.data, out["inscns"].ctypes.data
    it = abi.Tile()
    cc = np.ascontiguousarray(cols, dtype=np.int32)
    check(Lb.bcfgpu_gap_prep_tile(ctx.h, nc, cc.ctypes.data, None, C.byref(par), C.byref(oo), CAP, C.byref(it)))
    off = np.zeros(it.n_sites * S + 1, np.uint32); aux = np.zeros(it.n_reads, np.uint32)
    check(Lb.bcfgpu_memcpy_d2h(ctx.h, off.ctypes.data, it.plp_off, off.nbytes)); check(Lb.bcfgpu_memcpy_d2h(ctx.h, aux.ctypes.data, it.aux, aux.nbytes))
    st = abi.GapStats(); check(Lb.bcfgpu_gap_prep_stats(ctx.h, C.byref(st)))
    return out, off, aux, st
whole, off, aux, st = run(cand)
print("band jobs", list(st.band_jobs))
cut = len(cand) // 2 + 7
a, b = run(cand[:cut]), run(cand[cut:])
aux2 = np.concatenate([a[2], b[2]])
bad = np.nonzero(aux2 != aux)[0]
live = np.nonzero(whole["ret"] == 0)[0]
colstart = off[::S]
site_of = np.searchsorted(colstart, bad, side="right") - 1
print("mismatches", len(bad), "in tile sites", sorted(set(site_of.tolist())))
for j in sorted(set(site_of.tolist())):
    ci = int(live[j]); col = int(cand[ci])
    print("site", j, "cand", ci, "col", col, "types", whole["indel_types"][ci], "first half" if ci < cut else "second half")
    cols = np.array([col], np.int32); tot_e = int(col_n[col])
    so = np.zeros(S + 1, np.int32); pr, pq, pi = (np.zeros(tot_e, np.int32) for _ in range(3))
    check(Lb.bcfgpu_pileup_entries(ctx.h, 1, cols.ctypes.data, so.ctypes.data, pr.ctypes.data, pq.ctypes.data, pi.ctypes.data, tot_e))
    bb = dict(n_sites=1, n_smpl=S, ref=ref_b, pos=(cols + beg).astype(np.int32), smpl_off=so, p_read=pr, p_qpos=pq, p_indel=pi,
              reads=dict(arrs, n_reads=n, zq=np.zeros(n * L, np.uint8), r_has_zq=np.zeros(n, np.uint8)))
    want = indeldrv.gap_prep_oracle_site(bb, 0)
    w = want["aux"]; g1 = aux[off[j * S]:off[(j + 1) * S]]; g2 = aux2[off[j * S]:off[(j + 1) * S]]
    print("   whole vs oracle mismatches:", int((g1 != w).sum()), " split vs oracle mismatches:", int((g2 != w).sum()))
    for e in np.nonzero((g1 != w) | (g2 != w))[0][:6]:
        print("     entry", int(e), "indel", int(pi[e]), "oracle %x whole %x split %x" % (int(w[e]), int(g1[e]), int(g2[e])))

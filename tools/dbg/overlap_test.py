"""Experiment: the SNP step with tiles alternating between two contexts (two streams): does combine / mcall of tile i overlap
glfgen of tile i+1?  python tools/dbg/overlap_test.py [n_ctx] [steps]"""
import ctypes as C, sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from bcftools_amd import abi, engine, synth
from bcftools_amd.lib import check

nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
S, T = 1000, 32768
dev = torch.device("cuda", 0)
tile = synth.torch_tile(1, T, S, dev, depth=30.0, var_rate=0.01)
torch.cuda.synchronize()
R = tile["n_reads"]
cfg = abi.default_cfg(S, max_sites=T, max_reads=R, device=0, n_grp=1, fmt_flag=abi.INFO_VDB | abi.INFO_RPB)
ctxs = [engine.Context(cfg) for _ in range(nctx)]
dt = abi.Tile()
dt.n_sites, dt.is_indel, dt.n_reads = T, 0, R
dt.ref16, dt.plp_off, dt.rd, dt.epos = (tile["ref16"].data_ptr(), tile["plp_off"].data_ptr(), tile["rd"].data_ptr(), tile["epos"].data_ptr())
outs = []
rec_cap = max(64 << 20, (T * (512 + S * (2 + 4 * abi.MAX_PL))) // 4)
for c in ctxs:
    mo, mbufs, _ = c.alloc_mplp_out(T)
    co = abi.CallOut()
    csite = torch.zeros(T * C.sizeof(abi.CallSite), dtype=torch.uint8, device=dev)
    cgt = torch.zeros(T * 2 * S, dtype=torch.int8, device=dev)
    cpl = torch.zeros(T * abi.MAX_PL * S, dtype=torch.int32, device=dev)
    co.site, co.gt, co.pl, co.gq, co.gp = csite.data_ptr(), cgt.data_ptr(), cpl.data_ptr(), None, None
    rec = torch.empty(rec_cap, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(4, dtype=torch.int64, device=dev)
    outs.append((mo, mbufs, co, (csite, cgt, cpl), rec, cnt))


def step(i):
    c = ctxs[i % nctx]
    mo, _, co, _, rec, cnt = outs[i % nctx]
    check(c.L.bcfgpu_pipeline(c.h, C.byref(dt), None, None, C.byref(mo), C.byref(co)))
    check(c.L.bcfgpu_compact_calls_async(c.h, T, 0, mo.site, C.byref(co), abi.MAX_PL, 2, rec.data_ptr(), rec_cap, cnt.data_ptr()))


for i in range(4):
    step(i)
for c in ctxs:
    c.sync()
t0 = time.perf_counter()
for i in range(steps):
    step(i)
for c in ctxs:
    c.sync()
t = (time.perf_counter() - t0) / steps
print("contexts %d: %.3f ms per step = %.0f sites/s" % (nctx, t * 1e3, T / t))

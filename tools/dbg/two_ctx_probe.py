"""usage (GPU box): python tools/dbg/two_ctx_probe.py [sites]  -- the SNP step (bcfgpu_pipeline + record compaction) on the bench's tile with one
context, and with the steps dealt to two contexts (streams, workspaces and outputs of their own): do combine / call -m of one tile run beside
glfgen of the next?  Prints columns/s for both."""
import ctypes as C, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bcftools_amd import abi, synth, engine
from bcftools_amd.lib import check

T = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
S, steps, warm = 1000, 10, 2
dev = torch.device("cuda", 0)
tile = synth.torch_tile(20260104, T, S, dev, depth=30.0, var_rate=0.01)
torch.cuda.synchronize()
R = tile["n_reads"]
dt = abi.Tile()
dt.n_sites, dt.is_indel, dt.n_reads = T, 0, R
dt.ref16, dt.plp_off, dt.rd, dt.epos = (tile["ref16"].data_ptr(), tile["plp_off"].data_ptr(), tile["rd"].data_ptr(), tile["epos"].data_ptr())
rec_cap = max(64 << 20, (T * (512 + S * (2 + 4 * abi.MAX_PL))) // 4)


class Lane:
    def __init__(self):
        cfg = abi.default_cfg(S, max_sites=T, max_reads=R, device=0, fmt_flag=abi.INFO_VDB | abi.INFO_RPB)
        self.ctx = engine.Context(cfg)
        self.mo, self.mbufs, _ = self.ctx.alloc_mplp_out(T, self.ctx.flagged_planes())
        self.co = abi.CallOut()
        self.csite = torch.zeros(T * C.sizeof(abi.CallSite), dtype=torch.uint8, device=dev)
        self.cgt = torch.zeros(T * 2 * S, dtype=torch.int8, device=dev)
        self.cpl = torch.zeros(T * abi.MAX_PL * S, dtype=torch.int32, device=dev)
        self.co.site, self.co.gt, self.co.pl, self.co.gq, self.co.gp = self.csite.data_ptr(), self.cgt.data_ptr(), self.cpl.data_ptr(), None, None
        self.rec = torch.empty(rec_cap, dtype=torch.uint8, device=dev)
        self.cnt = torch.zeros(4, dtype=torch.int64, device=dev)

    def step(self):
        L = self.ctx.L
        check(L.bcfgpu_pipeline(self.ctx.h, C.byref(dt), None, None, C.byref(self.mo), C.byref(self.co)))
        check(L.bcfgpu_compact_calls_async(self.ctx.h, T, 0, self.mo.site, C.byref(self.co), abi.MAX_PL, 2, self.rec.data_ptr(), rec_cap, self.cnt.data_ptr()))


def run(lanes):
    def fence():
        for ln in lanes:
            check(ln.ctx.L.bcfgpu_sync(ln.ctx.h))
        torch.cuda.synchronize()
    for i in range(warm * len(lanes)):
        lanes[i % len(lanes)].step()
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        lanes[i % len(lanes)].step()
    fence()
    dtm = time.perf_counter() - t0
    return T * steps / dtm, dtm / steps * 1e3


a = Lane()
v1, ms1 = run([a])
b = Lane()
v2, ms2 = run([a, b])
v1b, ms1b = run([a])
print("one context   %.4g columns/s  %.3f ms a step" % (v1, ms1))
print("two contexts  %.4g columns/s  %.3f ms a step" % (v2, ms2))
print("one context   %.4g columns/s  %.3f ms a step (again)" % (v1b, ms1b))
nb, nr = C.c_uint64(), C.c_uint32()
check(a.ctx.L.bcfgpu_compact_counts(a.ctx.h, a.cnt.data_ptr(), C.byref(nb), C.byref(nr)))
n1 = (nb.value, nr.value)
check(b.ctx.L.bcfgpu_compact_counts(b.ctx.h, b.cnt.data_ptr(), C.byref(nb), C.byref(nr)))
print("records of the last steps: context a", n1, "context b", (nb.value, nr.value))

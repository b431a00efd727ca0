"""Experiment: the hot path of consecutive tiles on two contexts / two streams, so that combine + call of tile k run beside the
likelihood kernel of tile k+1.  python tools/overlap_probe.py [--contexts 2] [--steps 12] -> one line per setting."""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bcftools_amd import abi, engine, synth
from bcftools_amd.lib import check


def run(n_ctx, steps, warmup, T, S, tile, dev):
    R = tile["n_reads"]
    cfg = abi.default_cfg(S, max_sites=T, max_reads=R, device=0, fmt_flag=abi.INFO_VDB | abi.INFO_RPB)
    ctxs, outs, streams = [], [], []
    for k in range(n_ctx):
        ctx = engine.Context(cfg)
        st = torch.cuda.Stream(device=dev)
        check(ctx.L.bcfgpu_set_stream(ctx.h, C.c_void_p(st.cuda_stream)))
        mo, mbufs, _ = ctx.alloc_mplp_out(T, ctx.flagged_planes())
        co = abi.CallOut()
        csite = torch.zeros(T * C.sizeof(abi.CallSite), dtype=torch.uint8, device=dev)
        cgt = torch.zeros(T * 2 * S, dtype=torch.int8, device=dev)
        cpl = torch.zeros(T * abi.MAX_PL * S, dtype=torch.int32, device=dev)
        co.site, co.gt, co.pl, co.gq, co.gp = csite.data_ptr(), cgt.data_ptr(), cpl.data_ptr(), None, None
        rec = torch.empty(max(64 << 20, (T * (512 + S * (2 + 4 * abi.MAX_PL))) // 4), dtype=torch.uint8, device=dev)
        cnt = torch.zeros(4, dtype=torch.int64, device=dev)
        ctxs.append(ctx); streams.append(st); outs.append((mo, mbufs, co, csite, cgt, cpl, rec, cnt))
    dt = abi.Tile()
    dt.n_sites, dt.is_indel, dt.n_reads = T, 0, R
    dt.ref16, dt.plp_off, dt.rd, dt.epos = (tile["ref16"].data_ptr(), tile["plp_off"].data_ptr(), tile["rd"].data_ptr(), tile["epos"].data_ptr())
    L = ctxs[0].L

    def step(i):
        k = i % n_ctx
        mo, _, co, _, _, _, rec, cnt = outs[k]
        check(L.bcfgpu_pipeline(ctxs[k].h, C.byref(dt), None, None, C.byref(mo), C.byref(co)))
        check(L.bcfgpu_compact_calls_async(ctxs[k].h, T, 0, mo.site, C.byref(co), abi.MAX_PL, 2, rec.data_ptr(), rec.numel(), cnt.data_ptr()))

    def fence():
        for c in ctxs:
            check(L.bcfgpu_sync(c.h))
        torch.cuda.synchronize()

    for i in range(warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    fence()
    t1 = time.perf_counter()
    sums = [int(o[3].to(torch.int64).sum().item()) for o in outs]
    print("contexts %d: %.3f ms per tile, %.3e sites/s   (checksums of the call records %s)" % (n_ctx, (t1 - t0) / steps * 1e3, T * steps / (t1 - t0), sums), flush=True)
    del ctxs, outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--sites", type=int, default=32768)
    ap.add_argument("--samples", type=int, default=1000)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    tile = synth.torch_tile(1234, a.sites, a.samples, dev, depth=30, var_rate=0.01)
    torch.cuda.synchronize()
    for n in (1, 2, 3):
        run(n, a.steps, a.warmup, a.sites, a.samples, tile, dev)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

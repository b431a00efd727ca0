"""The library on a caller's stream (bcfgpu_set_stream), the way bench.py drives it for N > 1: the pipeline is enqueued
on a torch stream, torch work on the same stream (here a copy of the call records, in bench.py the RCCL gather) is ordered
after it by the stream alone, and the results equal those of the library's own stream."""
import ctypes as C

import numpy as np
import pytest

from bcftools_amd import abi, synth, engine
from bcftools_amd.lib import check

pytestmark = pytest.mark.gpu


def test_pipeline_on_a_torch_stream():
    import torch
    n_sites, n_smpl = 256, 200
    dev = torch.device("cuda", 0)
    tile = synth.torch_tile(77, n_sites, n_smpl, dev, depth=20.0, var_rate=0.2)
    torch.cuda.synchronize()
    R = tile["n_reads"]
    cfg = abi.default_cfg(n_smpl, max_sites=n_sites, max_reads=R)
    dt = abi.Tile()
    dt.n_sites, dt.is_indel, dt.n_reads = n_sites, 0, R
    dt.ref16, dt.plp_off, dt.rd, dt.epos = (tile["ref16"].data_ptr(), tile["plp_off"].data_ptr(),
                                            tile["rd"].data_ptr(), tile["epos"].data_ptr())

    def run(ctx, stream):
        L = ctx.L
        mo, mbufs, _ = ctx.alloc_mplp_out(n_sites)
        co = abi.CallOut()
        csite = torch.zeros(n_sites * C.sizeof(abi.CallSite), dtype=torch.uint8, device=dev)
        cgt = torch.zeros(n_sites * 2 * n_smpl, dtype=torch.int8, device=dev)
        cpl = torch.zeros(n_sites * abi.MAX_PL * n_smpl, dtype=torch.int32, device=dev)
        co.site, co.gt, co.pl, co.gq, co.gp = csite.data_ptr(), cgt.data_ptr(), cpl.data_ptr(), None, None
        copies = []
        if stream is not None:
            check(L.bcfgpu_set_stream(ctx.h, C.c_void_p(stream.cuda_stream)))
            with torch.cuda.stream(stream):
                for _ in range(3):                     # consecutive steps, each followed by stream-ordered torch work
                    check(L.bcfgpu_pipeline(ctx.h, C.byref(dt), None, None, C.byref(mo), C.byref(co)))
                    copies.append((csite.clone(), cgt.clone()))
            stream.synchronize()
        else:
            check(L.bcfgpu_pipeline(ctx.h, C.byref(dt), None, None, C.byref(mo), C.byref(co)))
            ctx.sync()
            copies.append((csite.clone(), cgt.clone()))
        torch.cuda.synchronize()
        ctx.release(list(mbufs.values()))
        return [(a.cpu().numpy(), b.cpu().numpy()) for a, b in copies]

    with engine.Context(cfg) as ctx:
        want = run(ctx, None)[0]
    with engine.Context(cfg) as ctx:
        got = run(ctx, torch.cuda.Stream(device=dev))
    assert len(got) == 3
    for site_bytes, gt in got:
        np.testing.assert_array_equal(site_bytes, want[0])
        np.testing.assert_array_equal(gt, want[1])
    assert want[1].any()

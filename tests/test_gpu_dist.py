"""The N > 1 path with the DEVICE engine on every rank (tests/test_dist_gloo.py covers the same exchange on CPU with the
oracle as the per-rank engine): two processes, one context each, run the fused pipeline on their region shard, compact the
records `call -mv` would write on the device (bcfgpu_compact_calls) and gather them to rank 0 in rank order
(shard.gather_packed).  With two or more GPUs visible the ranks sit on different devices and the exchange is RCCL
(backend "nccl"); on a one-GPU box both ranks share the device and the exchange runs over gloo -- the same code path but
for the transport."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bcftools_amd import abi, synth, shard

pytestmark = pytest.mark.gpu
N_SITES, N_SMPL, SEED = 96, 40, 4242


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _records(ctx, tile, site0):
    """bytes of the compacted variant records of `tile` (a HostTile) run through the fused pipeline on ctx"""
    from bcftools_amd.lib import check
    dt, tb = ctx.upload_tile(tile)
    mo, mb, _ = ctx.alloc_mplp_out(tile.n_sites)
    co, cb, _ = ctx.alloc_call_out(tile.n_sites, abi.MAX_PL)
    cap = 8 << 20
    buf = ctx.to_device(np.zeros(cap, np.uint8))
    nb, nr = C.c_uint64(), C.c_uint32()
    try:
        check(ctx.L.bcfgpu_pipeline(ctx.h, C.byref(dt), None, None, C.byref(mo), C.byref(co)))
        check(ctx.L.bcfgpu_compact_calls(ctx.h, tile.n_sites, site0, mo.site, C.byref(co), abi.MAX_PL, 2, buf.ptr, cap, C.byref(nb), C.byref(nr)))
        ctx.sync()
        out = np.zeros(int(nb.value), np.uint8)
        if nb.value:
            check(ctx.L.bcfgpu_memcpy_d2h(ctx.h, out.ctypes.data, buf.ptr, int(nb.value)))
    finally:
        ctx.release(list(tb) + list(mb.values()) + list(cb.values()) + [buf])
    return out, int(nr.value)


def _sub(tile, beg, end):
    from bcftools_amd import host
    S = tile.n_smpl
    o = tile.plp_off.astype(np.int64)
    return host.HostTile(S, tile.ref16[beg:end], (o[beg * S: end * S + 1] - o[beg * S]).astype(np.uint32),
                         tile.rd[o[beg * S]: o[end * S]], tile.epos[o[beg * S]: o[end * S]])


def _worker(rank, world, port, backend, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from bcftools_amd import engine
    ndev = torch.cuda.device_count()
    dev = rank % ndev
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    tile = synth.numpy_tile(SEED, N_SITES, N_SMPL, depth=18.0, var_rate=0.3)
    beg, end = shard.shard_range(N_SITES, rank, world)
    sub = _sub(tile, beg, end)
    with engine.Context(abi.default_cfg(N_SMPL, max_sites=max(1, end - beg), max_reads=len(sub.rd) + 64, device=dev)) as ctx:
        rec, _ = _records(ctx, sub, beg)
    where = torch.device("cuda", dev) if backend == "nccl" else torch.device("cpu")
    local = torch.from_numpy(np.concatenate([rec, np.zeros(16, np.uint8)])).to(where)
    outb = torch.zeros(8 << 20, dtype=torch.uint8, device=where) if rank == 0 else None
    sizes = shard.gather_packed(local, len(rec), outb, dst=0)
    if rank == 0:
        np.save(out_path, outb[:sum(sizes)].cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_the_device_engine_match_one_context(tmp_path, gpu_ctx_factory):
    out = str(tmp_path / "gathered.npy")
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    print("[test_gpu_dist] transport of the ordered gather: %s (%d device(s) visible)" % ("RCCL (backend nccl)" if backend == "nccl" else "gloo, ranks share the device", torch.cuda.device_count()))
    for attempt in range(2):
        try:
            mp.spawn(_worker, args=(2, _free_port(), backend, out), nprocs=2, join=True)
            break
        except Exception:
            if attempt:
                raise
    tile = synth.numpy_tile(SEED, N_SITES, N_SMPL, depth=18.0, var_rate=0.3)
    ctx = gpu_ctx_factory(abi.default_cfg(N_SMPL, max_sites=N_SITES, max_reads=len(tile.rd) + 64))
    want, n_rec = _records(ctx, tile, 0)
    got = np.load(out)
    assert n_rec > 5 and got.tobytes() == want.tobytes()
    # the records are self-describing: walk them
    o, sites = 0, []
    hdr = np.dtype([("site", "<i4"), ("n_gt", "<i4"), ("bytes", "<u4"), ("pad", "<i4")])
    while o < len(got):
        h = got[o:o + 16].view(hdr)[0]
        sites.append(int(h["site"]))
        o += int(h["bytes"])
    assert o == len(got) and sites == sorted(sites) and len(sites) == n_rec


def test_bench_rehearsal_two_ranks_match_one_process():
    """bench.py's N > 1 path as the round-end driver starts it -- `python -m torch.distributed.run --nproc-per-node 2 bench.py
    --gpus 2` -- rehearsed on one GPU (BCFGPU_BENCH_REHEARSE=1: both ranks share the device, the gather goes over gloo): the
    JSON line's gathered bytes per step must be what the two ranks' shards compact to, i.e. twice what one process compacts from
    the same per-rank tile (every rank draws its tile from the same seed + rank, so rank 0's share is the single run's).  The
    launcher is a fresh child of pytest: nothing in it has touched the GPU before torch.distributed.run starts the ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "2", "--warmup", "1", "--sites", "2048", "--samples", "200", "--cpu-seconds", "0", "--cpu-all-cores", "0", "--extras", "0"]
    env = dict(os.environ, BCFGPU_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")

    def last_json(out):
        for ln in reversed(out.strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        raise AssertionError("no JSON line:\n" + out[-2000:])
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    j1 = last_json(one.stdout)
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2"] + common,
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, (two.stderr or two.stdout)[-3000:]
    j2 = last_json(two.stdout)
    assert j2["n_gpus"] == 2 and j2["scaling"] == "weak" and "rehearsal" in j2
    assert j2["config"]["sites_per_step_per_gpu"] == j1["config"]["sites_per_step_per_gpu"] == 2048
    b1 = j1["config"]["record_bytes_per_step_per_gpu"]
    assert b1 > 0 and j2["config"]["record_bytes_per_step_per_gpu"] == b1          # rank 0's shard is the single process's tile
    g = j2["config"]["gathered_bytes_last_step"]
    assert g >= b1 and g > 0
    # every rank's records arrive: the other rank's shard (another seed) compacts to about as much as rank 0's
    assert 1.5 * b1 < g < 2.5 * b1, (g, b1)
    assert j2["value"] > 0 and j2["ms_per_step"] > 0

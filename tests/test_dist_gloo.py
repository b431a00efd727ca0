"""N>1 path on CPU: two gloo ranks each process their contiguous shard of a tile (with the oracle standing in for the
device, since this box has no GPU) and the ordered gather must reproduce the single-process result byte for byte."""
import os
import socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bcftools_amd import abi, synth, host, shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_shard(tile, cfg, beg, end):
    from tests.helpers import orc
    S = tile.n_smpl
    sub = host.HostTile(S, tile.ref16[beg:end], tile.plp_off[beg * S: end * S + 1] - tile.plp_off[beg * S],
                        tile.rd[tile.plp_off[beg * S]: tile.plp_off[end * S]],
                        tile.epos[tile.plp_off[beg * S]: tile.plp_off[end * S]])
    m = orc.mpileup(cfg, sub)
    cin = host.CallInput(S, m.site["n_alleles"], np.maximum(m.site["unseen"], 0), m.pl.astype(np.int32), m.site["qsum"])
    c = orc.mcall(cfg, cin)
    return c.site.tobytes()


def _worker(rank, world, port, n_sites, n_smpl, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tile = synth.numpy_tile(99, n_sites, n_smpl, depth=15.0, var_rate=0.3)
    cfg = abi.default_cfg(n_smpl)
    beg, end = shard.shard_range(n_sites, rank, world)
    rec = _run_shard(tile, cfg, beg, end)
    got = shard.gather_records(torch.frombuffer(bytearray(rec), dtype=torch.uint8), dst=0)
    if rank == 0:
        np.save(out_path, got.numpy())
    # the fixed-size exchange bench.py uses (equal shards, preallocated receive buffers): twice, as consecutive steps do
    fixed = torch.full((64,), rank + 1, dtype=torch.uint8)
    bufs = shard.gather_buffers(fixed, dst=0)
    for it in range(2):
        fixed.fill_(10 * it + rank + 1)
        shard.gather_fixed(fixed, bufs, dst=0)
        if rank == 0:
            assert [int(b[0]) for b in bufs] == [10 * it + r + 1 for r in range(world)]
            assert all(bool((b == b[0]).all()) for b in bufs)
    # the packed exchange of the record buffers (uneven sizes, one rank with nothing to send)
    for sizes in ([5, 11], [0, 7], [9, 0]):
        local = torch.full((16,), 40 + rank, dtype=torch.uint8)
        outb = torch.zeros(64, dtype=torch.uint8) if rank == 0 else None
        got_sizes = shard.gather_packed(local, sizes[rank], outb, dst=0)
        if rank == 0:
            assert got_sizes == sizes
            assert outb[:sum(sizes)].tolist() == [40] * sizes[0] + [41] * sizes[1]
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_cover_in_order():
    for n in (0, 1, 7, 64, 1001):
        for w in (1, 2, 3, 8):
            r = [shard.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(e - b for b, e in r) - min(e - b for b, e in r) <= 1


def test_two_rank_gather_matches_single_process(tmp_path):
    n_sites, n_smpl = 37, 9
    out = str(tmp_path / "gathered.npy")
    for attempt in range(2):                       # a probed-free port can be taken before the ranks bind it: one retry
        try:
            mp.spawn(_worker, args=(2, _free_port(), n_sites, n_smpl, out), nprocs=2, join=True)
            break
        except Exception:
            if attempt:
                raise
    tile = synth.numpy_tile(99, n_sites, n_smpl, depth=15.0, var_rate=0.3)
    want = _run_shard(tile, abi.default_cfg(n_smpl), 0, n_sites)
    got = np.load(out).tobytes()
    assert got == want

"""Multi-GPU partitioning of the hot path: sites are independent, so every rank owns a
contiguous range of sites (a genomic region shard) and no collective touches the data path.  The single
exchange step is the ordered gather of the fixed-stride per-site records to rank 0 -- shards are contiguous
and ranks ascending, so concatenating in rank order restores site (= VCF) order."""
import torch
import torch.distributed as dist


def shard_range(n_sites, rank, world):
    """[begin, end) of the sites owned by `rank`: contiguous, sizes differ by at most one."""
    base, rem = divmod(n_sites, world)
    beg = rank * base + min(rank, rem)
    return beg, beg + base + (1 if rank < rem else 0)


def gather_records(local, dst=0):
    """Gather byte tensors of per-site records (possibly different lengths) to `dst`, in rank order.
    Returns the concatenated tensor on `dst`, None elsewhere.  Works on any backend (gloo on CPU, RCCL on GPUs)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    pad = torch.zeros(mx, dtype=local.dtype, device=local.device)
    pad[: local.numel()] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)])


def gather_buffers(local, dst=0):
    """Receive buffers for gather_fixed(): one tensor like `local` per rank on `dst`, None elsewhere (allocate once)."""
    if dist.get_rank() != dst:
        return None
    return [torch.empty_like(local) for _ in range(dist.get_world_size())]


def gather_fixed(local, bufs, dst=0):
    """The same exchange when every rank contributes the same number of bytes (equal shards, as in the benchmark):
    one gather into preallocated buffers, no size exchange and no host synchronisation.  bufs from gather_buffers();
    on `dst` bufs[r] then holds rank r's records (rank order = site order)."""
    dist.gather(local, bufs, dst=dst)
    return bufs


def gather_packed(local, n_bytes, out, dst=0):
    """The ordered gather of packed record buffers (bcfgpu_compact_calls): rank r contributes local[:n_bytes]; `dst` ends up
    with rank 0's, rank 1's, ... bytes back to back in `out` and gets their sizes back.  The byte counts are exchanged first
    (a tiny all_gather), then one grouped send/recv: `dst` posts a receive per peer, every peer one send -- ncclGroupStart /
    ncclRecv x (N-1) / ncclSend / ncclGroupEnd on RCCL, the exchange bcfgpu_gather_bytes makes for the C driver."""
    world, rank = dist.get_world_size(), dist.get_rank()
    # n_bytes: a host integer, or a one-element int64 tensor on the device (the counter bcfgpu_compact_calls_async leaves there:
    # then the only host wait is the .item() below, on whatever stream is current -- bench.py makes that a side stream that
    # waits for the compaction alone, so the next step's kernels run meanwhile)
    n = n_bytes.reshape(1).to(torch.int64) if torch.is_tensor(n_bytes) else torch.tensor([int(n_bytes)], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(x.item()) for x in sizes]
    ops = []
    if rank == dst:
        assert out.numel() >= sum(sizes)
        off = 0
        for r in range(world):
            if r == dst:
                out[off:off + sizes[r]] = local[:sizes[r]]
            elif sizes[r]:
                ops.append(dist.P2POp(dist.irecv, out[off:off + sizes[r]], r))
            off += sizes[r]
    elif sizes[rank]:
        ops.append(dist.P2POp(dist.isend, local[:sizes[rank]], dst))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return sizes if rank == dst else None

"""The one collective of the multi-GPU path: gathering decoded shards into a single contiguous buffer.

Decoding itself never communicates (libgiddy_amd.sharding).  Only when the caller asks for ONE
contiguous output does rank `dst` receive every other rank's shard — over RCCL/xGMI when the
process group's backend is "nccl" (which IS RCCL on ROCm), one point-to-point transfer per peer:
each peer's shard crosses exactly one xGMI link once, so the gather is bound by the root's inbound
links, and a ring (per-link bound, shards forwarded hop by hop) would be strictly worse.
Works unchanged with the "gloo" backend on CPU tensors (that is how the tests exercise it).
"""
import torch
import torch.distributed as dist


def shard_extents(counts):
    """element offsets of each rank's shard in the contiguous buffer"""
    offsets, running = [], 0
    for c in counts:
        offsets.append(running)
        running += int(c)
    return offsets, running


def gather_decoded(local, counts, dst=0, group=None, out=None):
    """Gathers 1-D tensors `local` (rank r holds counts[r] elements) into one contiguous tensor on `dst`.

    Returns the contiguous tensor on `dst` (allocated there unless `out` is given), None elsewhere."""
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    assert len(counts) == world and local.numel() == int(counts[rank])
    offsets, total = shard_extents(counts)
    if rank == dst:
        if out is None:
            out = torch.empty(total, dtype=local.dtype, device=local.device)
        out[offsets[rank]: offsets[rank] + local.numel()].copy_(local)
        ops = [dist.P2POp(dist.irecv, out[offsets[r]: offsets[r] + int(counts[r])], r, group)
               for r in range(world) if r != dst and int(counts[r]) > 0]
    else:
        ops = [dist.P2POp(dist.isend, local.contiguous(), dst, group)] if local.numel() > 0 else []
    if ops:
        for request in dist.batch_isend_irecv(ops):
            request.wait()
    return out if rank == dst else None

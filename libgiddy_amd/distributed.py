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


# ---- move compressed, expand at the destination -------------------------------------------------------------------------------
#
# The library's premise (reference README.md:15-24) applied to xGMI: when ONE contiguous decoded column is wanted on rank `dst`,
# the peers do not decode and ship 8 bytes per element — they ship their COMPRESSED shards (2.75 bytes per element for C5) and
# `dst` decodes them into place.  Each peer's shard still crosses one link once; the root's inbound links carry 2.9 x less, and
# the decoding (the root now decodes the whole column, 1 / world of it its own) overlaps the transfers: every peer cuts its
# shard into `pieces` sub-shards with the scheme's own sharding function, piece k of ALL peers is one batch of point-to-point
# receives (all links busy at once), and piece k is decoded while piece k + 1 arrives.

def _describe(fields):
    """what a receiver must know to take a shard's fields: (dtype, numel) per tensor, everything else as it is"""
    return {k: (("tensor", str(v.dtype).replace("torch.", ""), int(v.numel())) if torch.is_tensor(v) else ("value", v)) for k, v in fields.items()}


def _empty_like_description(description, device):
    return {k: (torch.empty(d[2], dtype=getattr(torch, d[1]), device=device) if d[0] == "tensor" else d[1]) for k, d in description.items()}


def gather_compressed(shard_fields, shard_function, decode, out, counts, dst=0, group=None, pieces=4, device=None):
    """Rank r holds `shard_fields` — its shard of a column, as libgiddy_amd.sharding cut it, tensors where the arrays are — decoding to
    counts[r] elements.  On `dst`, `out` (sum(counts) elements) receives the whole decoded column: `dst` decodes its own shard and every
    peer's, whose compressed arrays arrive in `pieces` sub-shards (cut by `shard_function(fields, k, pieces)`).
    `decode(fields, out_view)` enqueues the decoding of a (sub-)shard into its range of `out` (on the current stream, for device tensors).
    Returns (bytes received by dst, bytes it would have received decoded) on `dst`, None elsewhere."""
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    offsets, total = shard_extents(counts)
    if device is None:
        device = out.device if out is not None else next(v.device for v in shard_fields.values() if torch.is_tensor(v))
    mine = [shard_function(shard_fields, k, pieces) for k in range(pieces)]      # Shard(fields, start, stop) relative to my shard
    described = [None] * world
    dist.all_gather_object(described, [(_describe(s.fields), s.start, s.stop) for s in mine], group=group)
    if rank != dst:
        for s in mine:
            ops = [dist.P2POp(dist.isend, v.contiguous(), dst, group) for v in s.fields.values() if torch.is_tensor(v) and v.numel() > 0]
            if ops:
                for request in dist.batch_isend_irecv(ops):
                    request.wait()
        return None
    received = 0
    def post(k):   # piece k of every peer: one batch — all links at once
        arrived, ops = {}, []
        for r in range(world):
            if r == dst:
                continue
            description, start, stop = described[r][k]
            arrived[r] = (_empty_like_description(description, device), start, stop)
            ops += [dist.P2POp(dist.irecv, v, r, group) for v in arrived[r][0].values() if torch.is_tensor(v) and v.numel() > 0]
        return arrived, (dist.batch_isend_irecv(ops) if ops else [])
    in_flight = post(0)
    for s in mine:    # my own shard meanwhile
        if s.stop > s.start:
            decode(s.fields, out[offsets[dst] + s.start: offsets[dst] + s.stop])
    for k in range(pieces):
        arrived, requests = in_flight
        in_flight = post(k + 1) if k + 1 < pieces else None     # the next piece is on its way before this one is decoded
        for request in requests:
            request.wait()
        for r, (fields, start, stop) in arrived.items():
            received += sum(v.numel() * v.element_size() for v in fields.values() if torch.is_tensor(v))
            if stop > start:
                decode(fields, out[offsets[r] + start: offsets[r] + stop])
    return received, (total - int(counts[dst])) * out.element_size()


def gather_compressed_on_node(key, root_device, root_streams, shards):
    """giddy_enqueue_gather_compressed_on_node (include/giddy.h): ONE process driving every device of a node — the compressed twin of
    giddy_enqueue_gather_on_node, and the one-process counterpart of gather_compressed above.  `shards`: a list of dicts
        device        where the shard's compressed arrays lie
        stream        a stream of that device (hipStream_t as an integer) behind whose work the arrays are complete
        arguments     the wrapper's launch arguments for the shard, name -> libgiddy_amd.registered typed value ("decompressed" points INTO the
                      contiguous buffer on the root)
        nbytes        name -> bytes of each array that travels (the others stay where they point)
        resolution    name -> size_t values for launch-configuration resolution
        copy_even_if_local   copy into the root's scratch even when device == root_device (tests on one GPU)
        decode_in_place      no copy: the root's decoder reads the shard's arrays over the link where they lie (peer access)
    Everything is enqueued; nothing is waited for.  root_streams[0] sees the whole column once it has run."""
    import ctypes
    from . import _native
    from .registered import _marshal, _check
    c_shards = (_native.giddy_compressed_shard * max(len(shards), 1))()
    keep = []
    for slot, shard in zip(c_shards, shards):
        arguments, kept = _marshal(shard["arguments"])
        resolution, kept_too = _marshal(shard["resolution"])
        nbytes = (ctypes.c_uint64 * max(len(shard["arguments"]), 1))(*[int(shard["nbytes"].get(name, 0)) for name in shard["arguments"]])
        keep += [arguments, kept, resolution, kept_too, nbytes]
        slot.device, slot.flags = int(shard["device"]), (1 if shard.get("copy_even_if_local") else 0) | (2 if shard.get("decode_in_place") else 0)
        slot.hip_stream = ctypes.c_void_p(int(shard["stream"]) if shard["stream"] else None)
        slot.arguments, slot.bytes, slot.num_arguments = arguments, nbytes, len(shard["arguments"])
        slot.resolution_arguments, slot.num_resolution_arguments = resolution, len(shard["resolution"])
    handles = (ctypes.c_void_p * len(root_streams))(*[int(s) if s else None for s in root_streams])
    _check(_native.lib.giddy_enqueue_gather_compressed_on_node(key.encode(), int(root_device), handles, len(root_streams), c_shards, len(shards)))

"""Transfer/decode overlap: decompress a column whose COMPRESSED form lives in (pinned) host memory.

The point of lightweight compression on a discrete GPU is to move the compressed bytes across the
host link and expand them on the device (reference README.md:15-24).  This module
does that for every scheme that shards (libgiddy_amd.sharding): the column is cut into `num_chunks`
ranges of whole anchored segments — a chunk is exactly a shard, with anchors rebased on the host —
and pipelined on `depth` HIP streams:   H2D copy of chunk i+1   ||   decode of chunk i.
The kernels are the ones the device-resident path uses (libgiddy_amd.schemes); torch supplies pinned
buffers, streams, events and async copies only.

Host fields: the dictionary a libgiddy_amd.formats writer returns, with the LARGE arrays replaced by
pinned torch tensors (`pin(fields)` does that, by name) so that their slices copy asynchronously; small arrays
(anchors, dictionaries, model records) stay numpy and are rebased per chunk.
"""
import numpy as np
import torch

from . import schemes, sharding

# the arrays a chunk takes a plain SLICE of: the bulk arrays of the formats and the per-segment arrays that need no rebasing (Delta's
# anchor values, frame-of-reference's model records — n / L of them, tens of MB for short segments) — pinned, their slices copy
# asynchronously.  position_anchors are rebased per chunk on the host and stay numpy; so do the small side tables.
BULK_ARRAYS = ("compressed_input", "compressed_data", "packed_element_sizes", "run_data", "run_lengths", "run_start_positions",
               "anchor_values", "segment_model_coefficients", "dictionary_entries")


def pin(fields):
    """the bulk arrays of a column (BULK_ARRAYS) -> pinned torch tensors holding the same bits, so that their slices copy
    asynchronously (a copy from pageable memory blocks the host thread, and with it the enqueueing of the other streams' work);
    everything else unchanged.  (position_anchors are NOT pinned whatever their size: a chunk's are rebased on the host.)"""
    out = {}
    for name, value in fields.items():
        if name in BULK_ARRAYS and isinstance(value, np.ndarray):
            out[name] = torch.from_numpy(np.ascontiguousarray(value)).pin_memory()
        else:
            out[name] = value
    return out


def _itemsize(x):
    return x.element_size() if isinstance(x, torch.Tensor) else np.asarray(x).dtype.itemsize


def _launch_dictionary(out, f, stream, kw):
    schemes.dictionary(out, f["compressed_input"], f["dictionary_entries"], f["length"], kw["num_dictionary_entries"],
                       uncompressed_size=kw["out_size"], dictionary_index_size=kw["sizes"]["compressed_input"], stream=stream)


def _launch_frame_of_reference(out, f, stream, kw):
    schemes.frame_of_reference(out, f["compressed_input"], f["segment_model_coefficients"], f["length"], f["segment_length"],
                               uncompressed_size=kw["out_size"], compressed_size=kw["sizes"]["compressed_input"],
                               model=kw.get("model", "constant"), stream=stream)


def _launch_delta(out, f, stream, kw):
    schemes.delta(out, f["compressed_input"], f["anchor_values"], f["length"], f["segment_length"],
                  uncompressed_size=kw["out_size"], compressed_size=kw["sizes"]["compressed_input"], stream=stream)


def _launch_dzb_fixed(out, f, stream, kw):
    schemes.discard_zero_bytes_fixed(out, f["compressed_input"], f["length"], uncompressed_size=kw["out_size"],
                                     compressed_size=kw["sizes"]["compressed_input"], stream=stream)


def _launch_dzb_variable(out, f, stream, kw):
    schemes.discard_zero_bytes_variable(out, f["compressed_data"], f["packed_element_sizes"], f["position_anchors"],
                                        f["segment_length"], f["length"], f["min_represented_element_size"], f["bits_per_element_size"],
                                        uncompressed_size=kw["out_size"], index_size=kw["sizes"]["position_anchors"],
                                        container_size=kw["sizes"]["packed_element_sizes"], stream=stream)


def _launch_rle(out, f, stream, kw):
    schemes.run_length_encoding(out, f["run_data"], f["run_lengths"], f["position_anchors"], f["intra_run_anchor_offsets"],
                                f["segment_length"], f["num_segments"], f["num_element_runs"], f["length"],
                                uncompressed_size=kw["out_size"], run_length_size=kw["sizes"]["run_lengths"],
                                index_size=kw["sizes"]["position_anchors"], stream=stream)


def _launch_rpe(out, f, stream, kw):
    schemes.run_position_encoding(out, f["run_data"], f["run_start_positions"], f["position_anchors"], f["segment_length"],
                                  f["num_anchors"], f["num_element_runs"], f["length"], uncompressed_size=kw["out_size"],
                                  position_size=kw["sizes"]["run_start_positions"], relative=kw.get("relative", True),
                                  index_size=kw["sizes"]["position_anchors"], stream=stream)


_SCHEMES = {   # name -> (how a chunk is cut out of the column, how it is launched)
    "dictionary": (sharding.elementwise, _launch_dictionary),
    "frame_of_reference": (sharding.frame_of_reference, _launch_frame_of_reference),
    "delta": (sharding.delta, _launch_delta),
    "discard_zero_bytes_fixed": (sharding.elementwise, _launch_dzb_fixed),
    "discard_zero_bytes_variable": (sharding.discard_zero_bytes_variable, _launch_dzb_variable),
    "run_length_encoding": (sharding.run_length_encoding, _launch_rle),
    "run_position_encoding": (lambda f, k, n: sharding.run_position_encoding(f, k, n, relative=True), _launch_rpe),
}


def _as_bytes(value):
    """host array (pinned torch tensor or numpy) -> flat uint8 torch tensor over the same bits"""
    if isinstance(value, torch.Tensor):
        return value.contiguous().view(torch.uint8).reshape(-1)
    return torch.from_numpy(np.ascontiguousarray(value).reshape(-1).view(np.uint8))


def decode_from_host(scheme, host_fields, out_dev, *, num_chunks=16, depth=3, **instantiation):
    """Decompress `host_fields` (see the module docstring) into the device tensor `out_dev`, `num_chunks` chunks of
    whole segments pipelined over `depth` streams.  Returns the streams used.  The side streams first wait for whatever
    the caller's current stream has queued (an earlier kernel or fill on `out_dev`); the CALLER synchronises the returned
    streams before reading `out_dev` and before reusing or freeing the host fields (the copies are asynchronous).
    `instantiation`: what the field dtypes do not say (model=..., num_dictionary_entries=..., relative=...)."""
    shard, launch = _SCHEMES[scheme]
    arrays = [k for k, v in host_fields.items() if isinstance(v, (torch.Tensor, np.ndarray))]
    kw = dict(instantiation, out_size=out_dev.element_size(), sizes={k: _itemsize(host_fields[k]) for k in arrays})
    if scheme == "dictionary":
        kw.setdefault("num_dictionary_entries", len(host_fields["dictionary_entries"]))
    flat_out = out_dev.view(-1)
    streams = [torch.cuda.Stream(device=out_dev.device) for _ in range(depth)]
    for stream in streams:
        stream.wait_stream(torch.cuda.current_stream(out_dev.device))   # work already queued on out_dev comes first
    staging = [dict() for _ in range(depth)]   # per slot: field name -> device byte buffer, grown on demand
    decoded = [None] * depth
    for chunk in range(num_chunks):
        piece = shard(host_fields, chunk, num_chunks)
        if piece.stop <= piece.start:
            continue
        slot = chunk % depth
        stream = streams[slot]
        with torch.cuda.stream(stream):
            if decoded[slot] is not None:
                stream.wait_event(decoded[slot])   # the previous decode out of this slot's staging buffers
            fields = dict(piece.fields)
            for name in arrays:
                source = _as_bytes(piece.fields[name])
                room = staging[slot].get(name)
                if room is None or room.numel() < source.numel():
                    room = torch.empty(max(source.numel(), 16) * 5 // 4, dtype=torch.uint8, device=out_dev.device)
                    staging[slot][name] = room
                room[: source.numel()].copy_(source, non_blocking=True)
                fields[name] = room
            launch(flat_out[piece.start:piece.stop], fields, stream.cuda_stream, kw)
            decoded[slot] = torch.cuda.Event()
            decoded[slot].record(stream)
    return streams


def dictionary_from_host(keys_host, entries_dev, out_dev, *, uncompressed_size=4, dictionary_index_size=1,
                         chunk_elements=1 << 26, depth=3):
    """The Dictionary special case with the dictionary already on the device: keys_host is a pinned CPU tensor of
    keys.  Returns the streams used; they wait for the caller's current stream first (the dictionary upload, earlier work
    on `out_dev`), and the caller synchronises them before reading `out_dev` or reusing `keys_host`."""
    n = keys_host.numel()
    num_entries = entries_dev.numel()
    streams = [torch.cuda.Stream(device=out_dev.device) for _ in range(depth)]
    for stream in streams:
        stream.wait_stream(torch.cuda.current_stream(out_dev.device))
    staging = [torch.empty(min(chunk_elements, n), dtype=keys_host.dtype, device=out_dev.device) for _ in range(depth)]
    ready = [None] * depth
    for i, lo in enumerate(range(0, n, chunk_elements)):
        hi = min(n, lo + chunk_elements)
        slot = i % depth
        s = streams[slot]
        with torch.cuda.stream(s):
            if ready[slot] is not None:
                s.wait_event(ready[slot])           # the previous decode out of this staging buffer
            staging[slot][: hi - lo].copy_(keys_host[lo:hi], non_blocking=True)
            schemes.dictionary(out_dev[lo:hi], staging[slot], entries_dev, hi - lo, num_entries,
                               uncompressed_size=uncompressed_size, dictionary_index_size=dictionary_index_size,
                               stream=s.cuda_stream)
            ready[slot] = torch.cuda.Event()
            ready[slot].record(s)
    return streams


# ---- the same pipeline in C++ behind the C ABI (giddy_decode_from_host) ---------------------------------------------------------
#
# The loop above spends ~60 us of Python per chunk (slicing, the anchors' rebasing, four or five copy_ calls, a ctypes launch): with
# 256 chunks that, not the link, bounds it.  libgiddy_amd/csrc/wrappers/sharding.hip does the whole loop natively — the cuts of
# include/giddy/sharding.hpp, hipMemcpyAsync of the chunk's slices into the stream's scratch block, the anchors rebased on the device,
# the wrapper's launch — and this is its Python face: the column's arguments are built exactly as for an ordinary launch (the
# scheme functions of libgiddy_amd.schemes, with HOST addresses), and handed over once.

_NATIVE_SCHEMES = {"dictionary": "elementwise", "discard_zero_bytes_fixed": "elementwise", "frame_of_reference": "frame_of_reference",
                   "delta": "delta", "discard_zero_bytes_variable": "discard_zero_bytes_variable", "run_length_encoding": "run_length_encoding"}


def decode_directly_from_pinned_host(scheme, host_fields, out_dev, *, stream=None, **instantiation):
    """ONE launch, no copies: the decoder reads the compressed column out of PINNED host memory itself, over the host link, while it writes
    the decoded column to the device.  Pinned memory (hipHostMalloc — what torch's pin_memory() gives) is mapped into the device's address
    space at the same address, so the kernels take the host pointers as they are; every byte of the compressed column crosses the link once,
    as with the chunked pipelines, but there is no chunk loop on the host at all (the chunked pipelines pay 3-5 us of host time per enqueued
    copy or launch: 256 chunks of a run-length column cost more host time than the link needs for its 50 MB).  Decoders that read their
    input more than once — the pre-passes for long segments — read it over the link as often: for those shapes prefer the chunked pipeline.
    EVERY array of `host_fields` must be pinned (pin_all()); returns nothing — the launch is on `stream` (default: torch's current)."""
    _, launch = _SCHEMES[scheme]
    for name, value in host_fields.items():
        if isinstance(value, np.ndarray) or (isinstance(value, torch.Tensor) and not value.is_cuda and not value.is_pinned()):
            raise ValueError("%s is not in pinned memory: decode_directly_from_pinned_host needs every array pinned (pipeline.pin_all)" % name)
    arrays = [k for k, v in host_fields.items() if isinstance(v, torch.Tensor)]
    kw = dict(instantiation, out_size=out_dev.element_size(), sizes={k: _itemsize(host_fields[k]) for k in arrays})
    if scheme == "dictionary":
        kw.setdefault("num_dictionary_entries", len(host_fields["dictionary_entries"]))
    launch(out_dev.view(-1), host_fields, stream if stream is not None else torch.cuda.current_stream(out_dev.device).cuda_stream, kw)


def pin_all(fields):
    """every array of a column -> a pinned torch tensor over the same bits (the small ones too: anchors, side tables)"""
    return {name: (torch.from_numpy(np.ascontiguousarray(value)).pin_memory() if isinstance(value, np.ndarray) else
                   (value if (value.is_cuda or value.is_pinned()) else value.pin_memory()) if isinstance(value, torch.Tensor) else value)
            for name, value in fields.items()}


def _captured_launch(launch, out, fields, kw):
    """(factory key, launch arguments) of schemes' launch of `fields` — nothing is enqueued"""
    captured = {}
    original = schemes._run

    def capture(key, resolve_arguments, launch_arguments, stream, device, limits, launch_config):
        captured.update(key=key, launch=launch_arguments)
        return None

    schemes._run = capture
    try:
        launch(out, fields, 0, kw)
    finally:
        schemes._run = original
    return captured["key"], captured["launch"]


def decode_from_host_native(scheme, host_fields, out_dev, *, num_chunks=16, depth=3, **instantiation):
    """decode_from_host, the chunk loop in C++ (include/giddy.h: giddy_decode_from_host).  Same contract: returns the streams used, which
    first wait for the caller's current stream; the caller synchronises them — and KEEPS `host_fields` ALIVE until then: these copies are
    plain hipMemcpyAsync calls that torch's pinned-memory cache knows nothing about (a pinned tensor that dies early may be unmapped or
    handed to someone else while the DMA still reads it)."""
    import ctypes
    from . import _native as native, registered as r
    _, launch = _SCHEMES[scheme]
    arrays = [k for k, v in host_fields.items() if isinstance(v, (torch.Tensor, np.ndarray))]
    kw = dict(instantiation, out_size=out_dev.element_size(), sizes={k: _itemsize(host_fields[k]) for k in arrays})
    if scheme == "dictionary":
        kw.setdefault("num_dictionary_entries", len(host_fields["dictionary_entries"]))
    key, arguments = _captured_launch(launch, out_dev.view(-1), host_fields, kw)
    array, keep = r._marshal(arguments)
    shape = native.giddy_column_shape()
    shape.length = int(host_fields["length"])
    shape.segment_length = int(host_fields.get("segment_length", 0))
    shape.output_itemsize = out_dev.element_size()
    if scheme == "discard_zero_bytes_variable":
        shape.bits_per_element_size = int(host_fields["bits_per_element_size"])
        shape.size_container_size = _itemsize(host_fields["packed_element_sizes"])
        shape.num_units = len(host_fields["compressed_data"])
    if scheme in ("run_length_encoding", "run_position_encoding"):
        shape.num_units = int(host_fields["num_element_runs"])
    name = _NATIVE_SCHEMES.get(scheme) or ("run_position_encoding_relative" if instantiation.get("relative", True) else "run_position_encoding_absolute")
    if "position_anchors" in host_fields:
        anchors = host_fields["position_anchors"]
        contiguous_anchors = anchors if isinstance(anchors, torch.Tensor) else np.ascontiguousarray(anchors)   # (alive until the C call has returned)
        shape.position_anchors = contiguous_anchors.data_ptr() if isinstance(anchors, torch.Tensor) else contiguous_anchors.ctypes.data
        shape.anchor_itemsize = _itemsize(anchors)
        shape.num_anchors = len(anchors)
    streams = [torch.cuda.Stream(device=out_dev.device) for _ in range(depth)]
    for stream in streams:
        stream.wait_stream(torch.cuda.current_stream(out_dev.device))
    handles = (ctypes.c_void_p * depth)(*[s.cuda_stream for s in streams])
    status = native.lib.giddy_decode_from_host(key.encode(), native.SCHEMES[name], ctypes.byref(shape), array, len(arguments), num_chunks, handles, depth)
    if status != 0:
        raise RuntimeError(native.lib.giddy_last_error().decode())
    return streams

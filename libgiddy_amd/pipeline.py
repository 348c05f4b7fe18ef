"""Transfer/decode overlap: decompress a column whose COMPRESSED form lives in (pinned) host memory.

The point of lightweight compression on a discrete GPU is to move the compressed bytes across the
host link and expand them on the device (reference README.md:15-24).  This module does that for any
elementwise-shardable scheme by cutting the column into chunks of whole segments and pipelining, on
`depth` HIP streams:   H2D copy of chunk i+1   ||   decode of chunk i.
The kernels are the same ones the device-resident path uses (libgiddy_amd.schemes); torch supplies
pinned buffers, streams and async copies only.
"""
import torch

from . import schemes


def dictionary_from_host(keys_host, entries_dev, out_dev, *, uncompressed_size=4, dictionary_index_size=1,
                         chunk_elements=1 << 26, depth=3):
    """keys_host: pinned CPU tensor of keys; entries_dev / out_dev: device tensors.  Returns the streams used
    (the caller synchronises).  Each chunk is copied into one of `depth` staging buffers and decoded
    straight into its slice of out_dev."""
    n = keys_host.numel()
    num_entries = entries_dev.numel()
    streams = [torch.cuda.Stream() for _ in range(depth)]
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

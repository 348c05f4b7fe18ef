"""PCIe-inclusive decode rates: the compressed column starts in pinned host memory and is decoded through
libgiddy_amd.pipeline (H2D copy of chunk i+1 overlapped with the decode of chunk i).
    python tools/pcie_pipeline.py [log2n=30] [scheme ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from libgiddy_amd import pipeline
import workloads as w

log2n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30
which = [a for a in sys.argv[1:] if not a.isdigit()] or ["dictionary", "frame_of_reference", "delta", "run_length_encoding", "discard_zero_bytes_variable"]
n = 1 << log2n
dev = torch.device("cuda", 0)


def host(t, dtype):
    """device tensor -> pinned host tensor holding the same bits under an unsigned dtype"""
    return torch.from_numpy(t.cpu().numpy().view(dtype)).pin_memory()


def small(t, dtype):
    return t.cpu().numpy().view(dtype).copy()


def fields_of(name):
    if name == "dictionary":
        x = w.Dictionary(n, dev)
        return x, {"compressed_input": host(x.keys, np.uint8), "dictionary_entries": small(x.entries, np.uint32), "length": n}, {}
    if name == "frame_of_reference":
        x = w.FrameOfReference(n, dev)
        return x, {"compressed_input": host(x.comp, np.int16), "segment_model_coefficients": small(x.coefs, np.int32), "length": n, "segment_length": x.L}, {"model": "constant"}
    if name == "delta":
        x = w.DeltaPatched(n, dev)
        x.run = lambda stream: w.schemes.delta(x.output, x.deltas, x.anchors, x.n, x.L, stream=stream)   # the decode alone
        return x, {"compressed_input": host(x.deltas, np.uint16), "anchor_values": small(x.anchors, np.uint64), "length": n, "segment_length": x.L}, {}
    if name == "run_length_encoding":
        x = w.RunLength(n, dev)
        return x, {"run_data": host(x.run_data, np.uint32), "run_lengths": host(x.run_lengths, np.uint16), "position_anchors": small(x.anchors, np.uint32),
                   "intra_run_anchor_offsets": small(x.intra, np.uint16), "segment_length": x.L, "num_segments": x.num_segments,
                   "num_element_runs": x.num_runs, "length": n}, {}
    if name == "discard_zero_bytes_variable":
        x = w.DiscardZeroBytesVariable(min(n, 1 << 28), dev)
        return x, {"compressed_data": host(x.data, np.uint8), "packed_element_sizes": host(x.packed, np.uint32), "position_anchors": small(x.anchors, np.uint64),
                   "segment_length": x.L, "length": x.n, "min_represented_element_size": x.min_size, "bits_per_element_size": x.bits}, {}
    raise SystemExit("unknown scheme " + name)


for name in which:
    x, fields, extra = fields_of(name)
    x.run(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    truth = x.output.clone()
    compressed = sum(v.numel() * v.element_size() for v in fields.values() if isinstance(v, torch.Tensor))
    probe = next(v for v in fields.values() if isinstance(v, torch.Tensor))
    landing = torch.empty_like(probe, device=dev)
    torch.cuda.synchronize(); t = time.perf_counter(); landing.copy_(probe, non_blocking=True); torch.cuda.synchronize()
    h2d = probe.numel() * probe.element_size() / (time.perf_counter() - t) / 1e9
    del landing
    fields = pipeline.pin(fields)
    for loop, decode in (("python", pipeline.decode_from_host), ("native", pipeline.decode_from_host_native)):   # the chunk loop in Python / in C++ behind the C ABI
        for chunks, depth in ((16, 2), (64, 3), (256, 3), (1024, 3)):
            best = 1e9
            for rep in range(3):
                x.output.fill_(0)
                torch.cuda.synchronize(); t = time.perf_counter()
                for s in decode(name, fields, x.output, num_chunks=chunks, depth=depth, **extra):
                    s.synchronize()
                best = min(best, time.perf_counter() - t)
            ok = bool((x.output == truth).all())
            out_bytes = x.output.numel() * x.output.element_size()
            print("%-28s %-6s 2^%d elements, %4d chunks, %d streams: %7.1f ms -> %7.1f GB/s decompressed, %5.1f GB/s compressed over PCIe (plain H2D copy: %.1f GB/s), ratio %.2f  ok=%s"
                  % (name, loop, int(np.log2(x.output.numel())), chunks, depth, best * 1e3, out_bytes / best / 1e9, compressed / best / 1e9, h2d, out_bytes / compressed, ok), flush=True)
    # ONE launch reading the pinned column over the link (libgiddy_amd.pipeline.decode_directly_from_pinned_host)
    everything = pipeline.pin_all(fields)
    best = 1e9
    for rep in range(4):
        x.output.fill_(0)
        torch.cuda.synchronize(); t = time.perf_counter()
        pipeline.decode_directly_from_pinned_host(name, everything, x.output, **extra)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    ok = bool((x.output == truth).all())
    out_bytes = x.output.numel() * x.output.element_size()
    print("%-28s %-6s 2^%d elements, ONE launch reading pinned host memory:  %7.1f ms -> %7.1f GB/s decompressed, %5.1f GB/s compressed over PCIe (plain H2D copy: %.1f GB/s), ratio %.2f  ok=%s"
          % (name, "direct", int(np.log2(x.output.numel())), best * 1e3, out_bytes / best / 1e9, compressed / best / 1e9, h2d, out_bytes / compressed, ok), flush=True)
    del x, fields, truth, everything
    torch.cuda.empty_cache()

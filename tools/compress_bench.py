"""Device-side compressors at full size: compress a device-resident column, decode it back with the library's decoders, compare, time.
    python tools/compress_bench.py [log2n=28]        (development aid; parity at small sizes is tests/test_gpu_compressors.py)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from libgiddy_amd import compressors, schemes
import workloads as w

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << log2n
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = float("inf")
    for _ in range(reps):
        t = time.perf_counter(); out = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return out, best


def report(name, column_bytes, compressed_bytes, seconds, ok):
    # algorithmic bytes: the column read once + what is written; against the 8 TB/s HBM peak, on the wall clock of the whole Python call
    algorithmic = column_bytes + compressed_bytes
    print("%-44s %8.2f ms  %7.1f GB/s of column  ratio %5.2f  algorithmic %5.2f GB -> %5.1f %% of the HBM peak  round trip %s" % (
        name, seconds * 1e3, column_bytes / seconds / 1e9, column_bytes / max(compressed_bytes, 1), algorithmic / 1e9, algorithmic / seconds / 8e12 * 100, "ok" if ok else "MISMATCH"), flush=True)


def nbytes(fields):
    return sum(v.numel() * v.element_size() for v in fields.values() if isinstance(v, torch.Tensor))

# Delta: the C3 column (uint64, steps below 2^16)
x = w.DeltaPatched(n, dev)
x.run(st); torch.cuda.synchronize()
column = x.output.clone(); column[x.patch_idx.long()] = column[x.patch_idx.long() - 1]   # (undo the patches: keep the steps small)
del x; torch.cuda.empty_cache()
column = torch.cumsum(torch.randint(0, 65536, (n,), dtype=torch.int64, device=dev), 0) + (1 << 40)
f, t = timed(lambda: compressors.delta(column, 4096, np.uint16))
out = torch.empty_like(column)
schemes.delta(out, f["compressed_input"], f["anchor_values"], n, 4096, stream=st)
report("delta<4,8,2>, segments of 4096", 8 * n, nbytes(f), t, bool((out == column).all()))
del f, out, column; torch.cuda.empty_cache()

# run-length / run-position: the C4 column
x = w.RunLength(n, dev)
x.run(st); torch.cuda.synchronize()
column = x.output.clone()
truth_runs = x.num_runs
del x; torch.cuda.empty_cache()
def by_events(fn, reps=5):
    """the launches alone (HIP events around them; the Python call's allocations and its read-back of the count are on the host's clock above)"""
    times = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record(); fn(); b.record(); torch.cuda.synchronize()
        times.append(a.elapsed_time(b))
    return sorted(times)[len(times) // 2]

CHAIN = {"patience": 1 << 23}   # (naming the chain's arguments asks for the one-launch kernels of round 3-4)
for name, call in (("run lengths", compressors.run_length_encoding), ("run positions", compressors.run_position_encoding)):
    for staged in (None, 256, 64):
        print("#   %-13s listed (no chain), %3s runs staged per wavefront: %6.3f ms by events (incl. the count's read-back)" % (
            name, staged or compressors.STAGED_PER_WAVE, by_events(lambda: call(column, 4096, np.uint16, capacity=n // 8, staged_per_wave=staged))), flush=True)
    print("#   %-13s ONE launch, chained tiles (rounds 3-4):            %6.3f ms by events" % (name, by_events(lambda: call(column, 4096, np.uint16, capacity=n // 8, chain=CHAIN))), flush=True)
f, t = timed(lambda: compressors.run_length_encoding(column, 4096, np.uint16, capacity=n // 8))
out = torch.empty_like(column)
schemes.run_length_encoding(out, f["run_data"], f["run_lengths"], f["position_anchors"], f["intra_run_anchor_offsets"], 4096, f["num_segments"], f["num_element_runs"], n, stream=st)
report("run_length_encoding<4,4,2>, mean run 32", 4 * n, nbytes(f), t, bool((out == column).all()) and f["num_element_runs"] == truth_runs)
f, t = timed(lambda: compressors.run_position_encoding(column, 4096, np.uint16, True, capacity=n // 8))
out.zero_()
schemes.run_position_encoding(out, f["run_data"], f["run_start_positions"], f["position_anchors"], 4096, f["num_anchors"], f["num_element_runs"], n, stream=st)
report("run_position_encoding<4,4,2,true>", 4 * n, nbytes(f), t, bool((out == column).all()))
del f, out, column; torch.cuda.empty_cache()

# variable-width discard-zero-bytes: the C5 column
m = min(n, 1 << 28)
x = w.DiscardZeroBytesVariable(m, dev)
column = x.values.clone()
reference = {"compressed_data": x.data, "packed_element_sizes": x.packed, "position_anchors": x.anchors}
f, t = timed(lambda: compressors.discard_zero_bytes_variable(column, 1024, 1, 2, np.uint32, index_size=8))
same = all(bool((f[k].view(torch.uint8) == reference[k].view(torch.uint8)).all()) for k in reference)   # the workload's own (torch) writer, byte for byte
report("discard_zero_bytes_variable<8,8,4>, 1..4 bytes", 8 * m, nbytes(f), t, same)
del f, x, column; torch.cuda.empty_cache()

# dictionary: 256 distinct 32-bit values
entries = (torch.randperm(1 << 24, device=dev)[:256] * 101 + 7).to(torch.int32)   # (below 2^31: signed and unsigned orders agree)
column = entries[torch.randint(0, 256, (n,), device=dev)]
f, t = timed(lambda: compressors.dictionary(column, np.uint8, entries=torch.sort(entries)[0]))
out = torch.empty_like(column)
schemes.dictionary(out, f["compressed_input"], f["dictionary_entries"], n, 256, stream=st)
report("dictionary<4,4,1> encoding, 256 entries", 4 * n, nbytes(f), t, bool((out == column).all()))
f, t = timed(lambda: compressors.dictionary(column, np.uint8))
out.zero_()
schemes.dictionary(out, f["compressed_input"], f["dictionary_entries"], n, 256, stream=st)
report("dictionary<4,4,1>: construction + encoding", 4 * n, nbytes(f), t, bool((out == column).all()) and f["num_dictionary_entries"] == 256)

"""What the number of workgroups is worth to a SHARD of the run decoders' columns: a 2^27-output shard of C4 is 32 768 segments of 4096 outputs; the resolved
grid is what is resident at once (256 CUs x 6 workgroups x 4 wavefronts = 6144 wavefronts at grid stride), i.e. 5.33 trips per wavefront — a sixth trip that a
third of the wavefronts make.  Here: the same shard decoded with grids of other sizes (the launch configuration is the caller's to give), the eight shards of one 2^30
column in turn.

    python tools/grid_quantisation.py [scheme=run_length_encoding|run_position_encoding_relative|run_position_encoding_absolute] [log2n=30] [shards=8]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import workloads as w
from libgiddy_amd import registered as r

scheme = sys.argv[1] if len(sys.argv) > 1 else "run_length_encoding"
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
device = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
n = 1 << log2n
whole = w.RunLength(n, device) if scheme == "run_length_encoding" else w.RunPosition(n, device, relative=scheme.endswith("relative"))
shards = [w.Sharded(whole, k, world) for k in range(world)]
resolved = shards[0].config
print("# %s, 2^%d outputs cut %d ways: %d segments per shard; resolved: %r" % (scheme, log2n, world, shards[0].n // whole.L, resolved))


def rate(config_of):
    def run():
        for s in shards: s.whole.launch_fields(s.output, s.fields, stream, config_of(s))
    for _ in range(3): run()
    torch.cuda.synchronize()
    times = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); run(); b.record(); torch.cuda.synchronize()
        times.append(a.elapsed_time(b) / (2 * len(shards)))
    return sorted(times)[len(times) // 2]


def column_ms():
    for _ in range(3): whole.run(stream)
    torch.cuda.synchronize()
    times = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); whole.run(stream); whole.run(stream); b.record(); torch.cuda.synchronize()
        times.append(a.elapsed_time(b) / 2)
    return sorted(times)[len(times) // 2]


whole_ms = column_ms()
print("# the column in one launch: %.4f ms = %.1f %% of the HBM peak; a shard at that rate: %.4f ms" % (whole_ms, whole.algorithmic_bytes / whole_ms / 8e7, whole_ms / world))
base = rate(lambda s: s.config)
print("resolved grid %6d workgroups: %.4f ms per shard  -> shard rate / column rate %.3f" % (resolved.grid_dimensions[0], base, whole_ms / world / base))
segments = shards[0].n // whole.L
waves_per_block = resolved.block_dimensions[0] // 64
grids = sorted(set([1024, 1152, 1280, 1366, 1408, 1536, 1640, 1792, 2048, 2304, 2560, 2731, 3072, 4096, 8192]
                   + [(segments + waves_per_block * t - 1) // (waves_per_block * t) for t in (2, 3, 4, 5, 6, 7, 8)]))
for grid in grids:
    def config_of(s, grid=grid):
        return r.launch_configuration_t((grid, 1, 1), s.config.block_dimensions, s.config.dynamic_shared_memory_size)
    ms = rate(config_of)
    print("grid %6d workgroups (%.2f trips per wavefront): %.4f ms per shard  -> shard rate / column rate %.3f" % (
        grid, segments / (grid * waves_per_block), ms, whole_ms / world / ms), flush=True)

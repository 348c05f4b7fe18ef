import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import sweep_all as sw
from libgiddy_amd import schemes, formats
n = 1 << 30; p = 1073742
rng = np.random.default_rng(4)
idx = np.sort(rng.choice(n, p, replace=False)).astype(np.uint32)
val = rng.integers(0, 1 << 62, size=p, dtype=np.uint64)
t = time.time(); c = formats.compressed_indices(idx, val, 1024, 4, 4); print("format writer: %.1f s, %d runs, pool %d B" % (time.time() - t, c["num_scatter_position_runs"], len(c["scatter_position_offset_bytes"])))
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()
target = torch.zeros(n, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
d_idx, d_val = dev(idx), dev(val)
cfg = schemes.scatter(target, d_val, d_idx, p, stream=st)
print("naive scatter: %.1f us  %s" % (1e3 * sw.timeit(lambda: schemes.scatter(target, d_val, d_idx, p, stream=st, launch_config=cfg)), cfg))
expect = target.clone()
target.zero_()
args = [dev(c[k]) for k in ("data_to_scatter", "scatter_position_run_lengths", "scatter_position_run_individual_offset_sizes", "scatter_position_run_baseline_values",
                             "scatter_position_run_offsets_start_pos", "scatter_position_offset_bytes", "scatter_position_anchors")]
kw = dict(output_index_size=4, element_size=8, input_index_size=4, stream=st)
cfg = schemes.compressed_indices_scatter(target, *args, 1024, c["num_scatter_position_runs"], p, **kw)
print("compressed-indices scatter: %.1f us  %s" % (1e3 * sw.timeit(lambda: schemes.compressed_indices_scatter(target, *args, 1024, c["num_scatter_position_runs"], p, launch_config=cfg, **kw)), cfg))
print("same result:", bool((target == expect).all()), " index bytes: naive %d, compressed %d" % (4 * p, sum(a.numel() for a in args[1:])))

"""GPU box: random shapes against the round-4 paths — Delta at any segment length (short segments sharing rows, ragged, odd, long) at shifted output bases with and without
patches of any density; run lengths and run positions with the one-kernel builds taken wherever they can be (GIDDY_RUN_FUSED_FROM=1 in the environment) or not;
DZB-variable at any segment length — each compared with a torch / numpy restatement.     [GIDDY_RUN_FUSED_FROM=1] python tools/stress_round4.py [seconds=120] [seed]"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import torch
import cases
from libgiddy_amd import schemes, formats

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rnd = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
T = {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}
def to_dev(a): return torch.from_numpy(np.ascontiguousarray(a).view({1: np.uint8, 2: np.int16, 4: np.int32, 8: np.int64}[a.dtype.itemsize])).to(dev)
def pick_length():
    return int(rnd.choice([rnd.integers(1, 20), rnd.integers(20, 600), rnd.integers(600, 9000), rnd.integers(9000, 70000), 2 ** int(rnd.integers(1, 17))]))

start, rounds, failures = time.time(), 0, 0
while time.time() - start < budget:
    rounds += 1
    kind = rnd.choice(["delta", "delta_patched", "rle", "rpe", "dzbv"])
    n = int(rnd.choice([rnd.integers(1, 3000), rnd.integers(3000, 300000), rnd.integers(300000, 3000000)]))
    L = pick_length()
    try:
        if kind in ("delta", "delta_patched"):
            usize, csize = [(8, 2), (8, 1), (8, 4), (4, 1), (4, 2), (2, 1)][int(rnd.integers(0, 6))]
            ud, cd = np.dtype("u%d" % usize), np.dtype("u%d" % csize)
            f, truth = cases.delta_case(n, L, ud, cd, seed=int(rnd.integers(0, 1 << 30)))
            shift = int(rnd.choice([0, 0, 16, 48, 496, 1008])) // usize
            out = torch.zeros(n + shift + 8, dtype=T[usize], device=dev)
            if kind == "delta":
                schemes.delta(out.data_ptr() + shift * usize, to_dev(f["compressed_input"]), to_dev(f["anchor_values"]), n, L, uncompressed_size=usize, compressed_size=csize, stream=st)
                expect = truth
            else:
                density = float(rnd.choice([0.0005, 0.01, 0.05, 0.3, 0.9]))
                positions = np.unique(rnd.integers(0, n, max(1, int(n * density)))).astype(np.uint32)
                values = rnd.integers(0, np.iinfo(ud).max, len(positions), dtype=ud)
                expect = truth.copy(); expect[positions.astype(np.int64)] = values
                schemes.patched_delta(out.data_ptr() + shift * usize, to_dev(f["compressed_input"]), to_dev(f["anchor_values"]), to_dev(values), to_dev(positions), n, L, len(positions),
                                      uncompressed_size=usize, compressed_size=csize, stream=st)
            got = out.cpu().numpy().view(ud)[shift:shift + n]
            what = (kind, usize, csize, n, L, shift)
        elif kind in ("rle", "rpe"):
            usize = int(rnd.choice([1, 2, 4, 8]))
            ud = np.dtype("u%d" % usize)
            col = cases.runs_column(n, float(rnd.choice([1.5, 4, 20, 40, 300])), ud, seed=int(rnd.integers(0, 1 << 30)))
            out = torch.zeros(n, dtype=T[usize], device=dev)
            if kind == "rle":
                f = formats.run_length_encoding(col, L, np.uint16 if rnd.random() < 0.7 else np.uint8)
                lsize = f["run_lengths"].dtype.itemsize
                schemes.run_length_encoding(out, to_dev(f["run_data"]), to_dev(f["run_lengths"]), to_dev(f["position_anchors"]), to_dev(f["intra_run_anchor_offsets"]), L,
                                            f["num_segments"], f["num_element_runs"], n, uncompressed_size=usize, run_length_size=lsize, stream=st)
                what = (kind, usize, lsize, n, L)
            else:
                relative = bool(rnd.random() < 0.5) and L <= 65536
                f = formats.run_position_encoding(col, L, np.uint16 if relative else np.uint32, relative)
                schemes.run_position_encoding(out, to_dev(f["run_data"]), to_dev(f["run_start_positions"]), to_dev(f["position_anchors"]), L, f["num_anchors"], f["num_element_runs"], n,
                                              uncompressed_size=usize, position_size=2 if relative else 4, relative=relative, stream=st)
                what = (kind, usize, relative, n, L)
            expect, got = col, out.cpu().numpy().view(ud)
        else:
            usize = int(rnd.choice([2, 4, 8])); bits = int(rnd.choice([1, 2, 3, 4])); min_size = int(rnd.integers(0, 2))
            if min_size + (1 << bits) - 1 > 8: bits = 2
            ud = np.dtype("u%d" % usize)
            col, sizes = cases.dzbv_column(n, ud, min_size, bits, seed=int(rnd.integers(0, 1 << 30)))
            f = formats.discard_zero_bytes_variable(col, L, min_size, bits, "u4", 4, element_sizes=sizes)
            out = torch.zeros(n, dtype=T[usize], device=dev)
            schemes.discard_zero_bytes_variable(out, to_dev(f["compressed_data"]), to_dev(f["packed_element_sizes"]), to_dev(f["position_anchors"]), L, n, min_size, bits,
                                                uncompressed_size=usize, container_size=4, index_size=4, stream=st)
            expect, got = col, out.cpu().numpy().view(ud)
            what = (kind, usize, bits, min_size, n, L)
        bad = np.flatnonzero(got != expect)
        if len(bad):
            failures += 1
            print("MISMATCH", what, "first at", bad[:5], flush=True)
    except Exception as e:
        failures += 1
        print("ERROR", kind, n, L, str(e).split("\n")[0][:200], flush=True)
print("%d random shapes, %d failures" % (rounds, failures), flush=True)
sys.exit(1 if failures else 0)

"""GPU box: random shapes against the five fused decompress-and-patch kernels — family, instantiation, length, patch density (none ... every element), duplicates,
clusters, index size, output displaced or not — each compared with the PLAIN decoder's output (the same library, another kernel) patched on the host by numpy
(repeated positions: the last value wins, as the oracle's sequential scatter).            python tools/stress_patched.py [seconds=120] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import cases
from libgiddy_amd import schemes

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rnd = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
T = {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}


def to_dev(a):
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view({1: np.uint8, 2: np.int16, 4: np.int32, 8: np.int64}[a.dtype.itemsize])).to(dev)


def to_host(t, dtype, n):
    return t.cpu().numpy().view(dtype)[:n]


def patches(n, dtype, index_size):
    kind = rnd.choice(["none", "sparse", "some", "many", "dense", "all", "duplicates", "clustered", "edges"])
    if kind == "none": positions = np.zeros(0, np.int64)
    elif kind == "all": positions = np.arange(n)
    elif kind == "duplicates": positions = np.sort(rnd.integers(0, n, size=max(2, int(n * rnd.choice([0.01, 0.1, 0.6])))))
    elif kind == "clustered":
        starts = rnd.integers(0, max(1, n - 500), size=int(rnd.integers(1, 6)))
        positions = np.unique(np.concatenate([np.arange(s, min(n, s + int(rnd.integers(1, 500)))) for s in starts]))
    elif kind == "edges":
        marks = np.concatenate([np.arange(0, n, 64), np.arange(0, n, 4096) - 1, np.arange(0, n, 1024) + 1, np.arange(0, n, 256) + 255, [0, n - 1]])
        positions = np.unique(marks[(marks >= 0) & (marks < n)])
    else:
        density = {"sparse": 0.001, "some": float(rnd.choice([0.004, 0.008, 0.012])), "many": float(rnd.choice([0.02, 0.05, 0.1])), "dense": float(rnd.choice([0.25, 0.5, 0.9]))}[kind]
        positions = np.flatnonzero(rnd.random(n) < density)
    info = np.iinfo(dtype)
    values = rnd.integers(info.min, info.max, size=len(positions), dtype=dtype, endpoint=True)
    return kind, positions.astype("u%d" % index_size), values


def patched_on_the_host(decoded, positions, values):
    out = decoded.copy()
    if len(positions):
        out[positions.astype(np.int64)] = values   # (numpy: for a repeated index the last value is assigned)
        # ... which numpy does not promise for every build: make sure of it where positions repeat
        repeated = np.flatnonzero(positions[1:] == positions[:-1])
        if len(repeated):
            last = np.ones(len(positions), bool); last[repeated] = False
            out[positions[last].astype(np.int64)] = values[last]
    return out


start, rounds, failures, by_family = time.time(), 0, 0, {}
while time.time() - start < budget:
    rounds += 1
    family = str(rnd.choice(["delta", "dictionary", "discard_zero_bytes_fixed", "frame_of_reference", "model"]))
    n = int(rnd.choice([rnd.integers(1, 3000), rnd.integers(3000, 300000), rnd.integers(300000, 3000000)]))
    index_size = int(rnd.choice([4, 8]))
    shift_bytes = int(rnd.choice([0, 0, 0, 16, 512, 4096]))
    try:
        if family == "delta":
            usize, csize = [(2, 1), (4, 1), (4, 2), (8, 1), (8, 2), (8, 4)][int(rnd.integers(0, 6))]
            ud = np.dtype("u%d" % usize)
            L = int(rnd.choice([rnd.integers(1, 40), rnd.integers(40, 5000), 2 ** int(rnd.integers(6, 17)), n]))
            f, _ = cases.delta_case(n, L, ud, np.dtype("u%d" % csize), seed=int(rnd.integers(0, 1 << 30)))
            d = [to_dev(f["compressed_input"]), to_dev(f["anchor_values"])]
            plain = lambda out: schemes.delta(out, d[0], d[1], n, L, uncompressed_size=usize, compressed_size=csize, index_size=index_size, stream=st)
            fused = lambda out, pv, pp, k: schemes.patched_delta(out, d[0], d[1], pv, pp, n, L, k, uncompressed_size=usize, compressed_size=csize, index_size=index_size, stream=st)
        elif family == "dictionary":
            usize, dsize = [(1, 1), (2, 1), (2, 2), (4, 1), (4, 2), (8, 1), (8, 2)][int(rnd.integers(0, 7))]
            ud = np.dtype("u%d" % usize)
            entries = int(rnd.choice([1, 17, 256])) if dsize == 1 else int(rnd.choice([300, 4096, 40000]))
            f = cases.dictionary_case(n, entries, ud, np.dtype("u%d" % dsize), seed=int(rnd.integers(0, 1 << 30)))
            d = [to_dev(f["compressed_input"]), to_dev(f["dictionary_entries"])]
            plain = lambda out: schemes.dictionary(out, d[0], d[1], n, entries, uncompressed_size=usize, dictionary_index_size=dsize, index_size=index_size, stream=st)
            fused = lambda out, pv, pp, k: schemes.patched_dictionary(out, d[0], d[1], pv, pp, n, entries, k, uncompressed_size=usize, dictionary_index_size=dsize, index_size=index_size, stream=st)
        elif family == "discard_zero_bytes_fixed":
            usize, csize = [(2, 1), (4, 1), (4, 2), (8, 1), (8, 2), (8, 4)][int(rnd.integers(0, 6))]
            ud, cd = np.dtype("u%d" % usize), np.dtype("u%d" % csize)
            narrow = rnd.integers(0, np.iinfo(cd).max, size=n, endpoint=True, dtype=cd)
            d = [to_dev(narrow)]
            plain = lambda out: schemes.discard_zero_bytes_fixed(out, d[0], n, uncompressed_size=usize, compressed_size=csize, index_size=index_size, stream=st)
            fused = lambda out, pv, pp, k: schemes.patched_discard_zero_bytes_fixed(out, d[0], pv, pp, n, k, uncompressed_size=usize, compressed_size=csize, index_size=index_size, stream=st)
        elif family == "frame_of_reference":
            isz, usize, csize, model, misz = [(4, 2, 1, "linear", 4), (4, 4, 1, "linear", 4), (4, 4, 2, "linear", 4), (8, 4, 2, "linear", 4), (8, 4, 2, "linear", 8),
                                              (4, 2, 1, "constant", 4), (4, 4, 1, "constant", 4), (4, 4, 2, "constant", 4)][int(rnd.integers(0, 8))]
            index_size = isz
            ud = np.dtype("i%d" % usize)
            L = max(4, int(rnd.choice([rnd.integers(4, 200), rnd.integers(200, 9000), 2 ** int(rnd.integers(2, 15))])) // 4 * 4)
            f = cases.for_case(n, L, ud, np.dtype("i%d" % csize), model, seed=int(rnd.integers(0, 1 << 30)))
            d = [to_dev(f["compressed_input"]), to_dev(f["segment_model_coefficients"])]
            plain = lambda out: schemes.frame_of_reference(out, d[0], d[1], n, L, uncompressed_size=usize, compressed_size=csize, model=model, index_size=isz, model_index_size=misz, stream=st)
            fused = lambda out, pv, pp, k: schemes.patched_frame_of_reference(out, d[0], d[1], pv, pp, n, L, k, uncompressed_size=usize, compressed_size=csize, model=model, index_size=isz,
                                                                              model_index_size=misz, stream=st)
        else:
            isz, usize, model = [(4, 2, "linear"), (4, 4, "linear"), (8, 4, "linear"), (4, 2, "constant"), (4, 4, "constant")][int(rnd.integers(0, 5))]
            index_size = isz
            ud = np.dtype("i%d" % usize)
            coefficients = np.array([int(rnd.integers(-30000, 30000)), int(rnd.integers(-50, 50))] if model == "linear" else [int(rnd.integers(-30000, 30000))], dtype=ud)
            plain = lambda out: schemes.model(out, coefficients, n, uncompressed_size=usize, model=model, index_size=isz, stream=st)
            fused = lambda out, pv, pp, k: schemes.patched_model(out, coefficients, pv, pp, n, k, uncompressed_size=usize, model=model, index_size=isz, stream=st)
        shift = shift_bytes // usize
        kind, positions, values = patches(n, ud, index_size)
        base = torch.zeros(n + shift + 64, dtype=T[usize], device=dev)
        plain(base.data_ptr() + shift * usize)
        decoded = to_host(base[shift:], ud, n).copy()
        out = torch.full((n + shift + 64,), 0x5A if usize == 1 else 0x5A5A, dtype=T[usize], device=dev)
        before_and_after = (to_host(out, ud, n + shift + 64)[:shift].copy(), to_host(out, ud, n + shift + 64)[shift + n:].copy())
        fused(out.data_ptr() + shift * usize, to_dev(values) if len(values) else torch.zeros(1, dtype=T[usize], device=dev),
              to_dev(positions) if len(positions) else torch.zeros(1, dtype=T[index_size], device=dev), len(positions))
        torch.cuda.synchronize()
        whole = to_host(out, ud, n + shift + 64)
        got, expect = whole[shift:shift + n], patched_on_the_host(decoded, positions, values)
        if os.environ.get("STRESS_SELFTEST") and rounds == 3: expect = expect.copy(); expect[len(expect) // 2] ^= 1   # (the harness itself: one planted difference must be reported)
        ok = np.array_equal(got, expect) and np.array_equal(whole[:shift], before_and_after[0]) and np.array_equal(whole[shift + n:], before_and_after[1])
        by_family[family] = by_family.get(family, 0) + 1
        if not ok:
            failures += 1
            bad = np.flatnonzero(got != expect)
            print("MISMATCH %s u%d index %d n=%d shift=%d patches=%s (%d): first wrong %s (got %s, expected %s); guard bands intact: %s" % (
                family, usize, index_size, n, shift, kind, len(positions), bad[:4], got[bad[:4]], expect[bad[:4]],
                np.array_equal(whole[:shift], before_and_after[0]) and np.array_equal(whole[shift + n:], before_and_after[1])), flush=True)
    except Exception as problem:
        failures += 1
        print("ERROR %s n=%d: %s" % (family, n, str(problem).splitlines()[0][:200]), flush=True)
print("%d random fused-patch cases (%s), %d failures" % (rounds, ", ".join("%s %d" % kv for kv in sorted(by_family.items())), failures))

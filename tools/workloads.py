"""Synthetic, seeded, device-resident workloads for bench.py — the BASELINE.json configs.

Inputs are generated ON THE GPU with torch (torch is the allocator / RNG here, nothing more), each
workload exposes:
    run(stream)            enqueue one pass of the hot path (one "step") through libgiddy_amd.schemes
    output                 the decompressed device buffer
    algorithmic_bytes      compulsory HBM bytes of one step (reads + writes)
    decompressed_bytes     bytes of decompressed output per step
    kernel                 name of the dominant kernel (for the rocprof summary)
    verify(oracle)         bit-exact checks: a bounded sample against the CPU oracle + full-size identities
    cpu_sample(...)        host copies of a bounded sample for the CPU baseline
"""
import numpy as np
import torch

from libgiddy_amd import schemes, sharding


SEED_OFFSET = 0   # bench.py sets this to 1000 * rank so that every GPU decodes different data



def output_buffer(n, dtype, device):
    """an output of n elements on a 2 MiB boundary — where hipMalloc puts an allocation of this size.  torch's caching allocator may hand out any
    512-byte offset inside a block it has cached, and the rate of some decoders depends on it (the run decoders lose 6 points
    when the 64 KiB a workgroup writes lie across two 64 KiB pages — tools/alignment_probe.py)"""
    size = torch.empty(0, dtype=dtype).element_size()
    raw = torch.empty(n + (2 << 20) // size, dtype=dtype, device=device)
    skip = (-raw.data_ptr() % (2 << 20)) // size
    return raw[skip:skip + n]


def _gen(seed, device):
    g = torch.Generator(device=device)
    g.manual_seed(seed + SEED_OFFSET)
    return g


def _np(t, dtype=None):
    a = t.detach().cpu().numpy()
    return a.view(dtype) if dtype is not None else a


def checksum(t, first_index=0):
    """position-weighted 64-bit checksum of a decoded (device) range: sum of element * (2 * global index + 1), wrapping —
    the sum of the shards' checksums is the column's whatever the cut"""
    total = 0
    step = 1 << 27
    flat = t.reshape(-1)
    for lo in range(0, flat.numel(), step):
        hi = min(flat.numel(), lo + step)
        weights = 2 * torch.arange(first_index + lo, first_index + hi, device=t.device, dtype=torch.int64) + 1
        total = (total + int((flat[lo:hi].to(torch.int64) * weights).sum())) & ((1 << 64) - 1)
    return total


class Sharded:
    """Rank `rank` of `world`'s shard of a workload's column: cut from the device-resident arrays by libgiddy_amd.sharding (views
    + rebased anchors, no host round trip), decoded into its range of the workload's output buffer."""

    def __init__(self, work, rank, world):
        self.whole, self.rank, self.world = work, rank, world
        shard = work.shard_function(work.column_fields(), rank, world)
        self.start, self.stop, self.fields = shard.start, shard.stop, shard.fields
        self.n = self.stop - self.start
        self.output = work.output[self.start:self.stop]
        self.name, self.kernel = work.name, work.kernel
        self.decompressed_bytes = self.output.numel() * self.output.element_size()
        self.algorithmic_bytes = self.decompressed_bytes + sum(v.numel() * v.element_size() for v in self.fields.values() if isinstance(v, torch.Tensor))
        self.config = work.launch_fields(self.output, self.fields, torch.cuda.current_stream().cuda_stream, None) if self.n else None

    def run(self, stream):
        if self.n:
            self.whole.launch_fields(self.output, self.fields, stream, self.config)


class Dictionary:
    """C2: 8-bit keys -> 32-bit values, 256-entry dictionary, uniform keys (seed 2)."""
    name = "dictionary<4,4,1>"
    kernel = "dictionary::decompress"
    dtype = "u32"

    def __init__(self, n, device, seed=2, zipf=False):
        g = _gen(seed, device)
        self.n = n
        self.entries = torch.randperm(1 << 24, generator=g, device=device)[:256].to(torch.int32) * 251 + 7
        if zipf:
            u = torch.rand(n, generator=g, device=device)
            self.keys = torch.clamp((u ** (-1.0 / 0.2)).to(torch.int64) - 1, 0, 255).to(torch.uint8)
        else:
            self.keys = torch.randint(0, 256, (n,), dtype=torch.uint8, generator=g, device=device)
        self.output = output_buffer(n, torch.int32, device)
        self.algorithmic_bytes = 5 * n + 1024
        self.decompressed_bytes = 4 * n
        self.config = schemes.dictionary(self.output, self.keys, self.entries, n, 256, stream=torch.cuda.current_stream().cuda_stream)

    def run(self, stream):
        schemes.dictionary(self.output, self.keys, self.entries, self.n, 256, stream=stream, launch_config=self.config)

    # the column in the format writers' vocabulary; how a shard of it is cut and launched (tools/size_probe.py, bench.py --scaling strong)
    def column_fields(self):
        return {"compressed_input": self.keys, "dictionary_entries": self.entries, "length": self.n}

    shard_function = staticmethod(sharding.elementwise)

    def launch_fields(self, out, f, stream, config):
        return schemes.dictionary(out, f["compressed_input"], f["dictionary_entries"], f["length"], 256, stream=stream, launch_config=config)

    def verify(self, oracle, sample=1 << 22):
        torch.cuda.synchronize()
        n = self.n
        # (1) a bounded sample, bit-exact against the CPU oracle: head and tail
        entries = _np(self.entries, np.uint32)
        for lo, hi in ((0, min(sample, n)), (max(0, n - sample), n)):
            expect = oracle.dictionary(_np(self.keys[lo:hi]), entries)
            assert (expect == _np(self.output[lo:hi], np.uint32)).all(), "dictionary: sample mismatch vs oracle"
        # (2) full size: sum of outputs == sum_k count_k * entry_k (size-independent identity), and
        #     an independent torch gather, bit for bit
        counts = torch.bincount(self.keys.to(torch.int64), minlength=256)
        assert int((counts * self.entries.to(torch.int64)).sum()) == int(self.output.sum(dtype=torch.int64))
        step = 1 << 28
        for lo in range(0, n, step):
            hi = min(n, lo + step)
            assert bool((self.entries[self.keys[lo:hi].to(torch.int64)] == self.output[lo:hi]).all())

    def cpu_run(self, oracle, sample, threads):
        keys = _np(self.keys[:sample]).copy()
        entries = _np(self.entries, np.uint32).copy()
        out = np.empty(sample, dtype=np.uint32)
        return (lambda: oracle.dictionary(keys, entries, threads=threads, out=out)), 4 * sample


class FrameOfReference:
    """FOR int32 <- int16, constant model, segment 4096 (the north-star FOR target; C1 is its 1-segment 2^20 plumbing case)."""
    name = "frame_of_reference<4,int,short,constant>"
    kernel = "frame_of_reference::decompress"

    def __init__(self, n, device, seed=1, segment_length=4096):
        g = _gen(seed, device)
        self.n, self.L = n, segment_length
        nseg = (n + segment_length - 1) // segment_length
        self.comp = torch.randint(-32768, 32768, (n,), dtype=torch.int16, generator=g, device=device)
        self.coefs = torch.randint(-1000000, 1000000, (nseg, 1), dtype=torch.int32, generator=g, device=device)
        self.output = output_buffer(n, torch.int32, device)
        self.algorithmic_bytes = 6 * n + 4 * nseg
        self.decompressed_bytes = 4 * n
        self.config = schemes.frame_of_reference(self.output, self.comp, self.coefs, n, segment_length,
                                                 stream=torch.cuda.current_stream().cuda_stream)

    def run(self, stream):
        schemes.frame_of_reference(self.output, self.comp, self.coefs, self.n, self.L, stream=stream, launch_config=self.config)

    def column_fields(self):
        return {"compressed_input": self.comp, "segment_model_coefficients": self.coefs, "length": self.n, "segment_length": self.L}

    shard_function = staticmethod(sharding.frame_of_reference)

    def launch_fields(self, out, f, stream, config):
        return schemes.frame_of_reference(out, f["compressed_input"], f["segment_model_coefficients"], f["length"], f["segment_length"], stream=stream, launch_config=config)

    def verify(self, oracle, sample=1 << 22):
        torch.cuda.synchronize()
        n, L = self.n, self.L
        sample = (sample // L) * L
        for lo, hi in ((0, min(sample, n)), (max(0, (n - sample) // L * L), n)):
            expect = oracle.frame_of_reference(_np(self.comp[lo:hi]), _np(self.coefs[lo // L:(hi + L - 1) // L]), hi - lo, L, np.int32)
            assert (expect == _np(self.output[lo:hi])).all(), "FOR: sample mismatch vs oracle"
        # full size: out - in == per-segment constant
        step = 1 << 28
        for lo in range(0, n, step):
            hi = min(n, lo + step)
            ref = self.coefs[lo // L:(hi + L - 1) // L].view(-1).repeat_interleave(L)[: hi - lo]
            assert bool(((self.output[lo:hi] - self.comp[lo:hi].to(torch.int32)) == ref).all())

    def cpu_run(self, oracle, sample, threads):
        sample = (sample // self.L) * self.L
        comp = _np(self.comp[:sample]).copy()
        coefs = _np(self.coefs[: sample // self.L]).copy()
        out = np.empty(sample, dtype=np.int32)
        return (lambda: oracle.frame_of_reference(comp, coefs, sample, self.L, np.int32, threads=threads, out=out)), 4 * sample


class FrameOfReferenceFused(FrameOfReference):
    """FOR int32 <- int16 plus 0.1 % ascending patches merged inside the kernel"""
    name = "frame_of_reference::patched<4,int,short,constant>"
    kernel = "frame_of_reference::patched::decompress"

    def __init__(self, n, device, seed=1, segment_length=4096, patch_fraction=0.001):
        super().__init__(n, device, seed=seed, segment_length=segment_length)
        self.p = max(1, int(round(n * patch_fraction)))
        self.patch_idx = torch.sort(torch.randperm(n, generator=_gen(seed + 1, device), device=device)[: self.p])[0].to(torch.int32)
        self.patch_val = torch.randint(-(1 << 31), 1 << 31, (self.p,), dtype=torch.int32, generator=_gen(seed + 2, device), device=device)
        self.algorithmic_bytes += 8 * self.p
        self.config = schemes.patched_frame_of_reference(self.output, self.comp, self.coefs, self.patch_val, self.patch_idx,
                                                         n, segment_length, self.p, stream=torch.cuda.current_stream().cuda_stream)

    def run(self, stream):
        schemes.patched_frame_of_reference(self.output, self.comp, self.coefs, self.patch_val, self.patch_idx, self.n, self.L, self.p,
                                           stream=stream, launch_config=self.config)

    def verify(self, oracle, sample=1 << 22):
        torch.cuda.synchronize()
        n, L, idx = self.n, self.L, self.patch_idx.to(torch.int64)
        sample = (sample // L) * L
        pidx, pval = _np(self.patch_idx, np.uint32).astype(np.int64), _np(self.patch_val)
        for lo, hi in ((0, min(sample, n)), (max(0, (n - sample) // L * L), n)):
            expect = oracle.frame_of_reference(_np(self.comp[lo:hi]), _np(self.coefs[lo // L:(hi + L - 1) // L]), hi - lo, L, np.int32)
            inside = (pidx >= lo) & (pidx < hi)
            expect[pidx[inside] - lo] = pval[inside]
            assert (expect == _np(self.output[lo:hi])).all(), "FOR+patches: sample mismatch vs oracle"
        step = 1 << 28
        for lo in range(0, n, step):   # full size: out - in == per-segment constant except at patched positions, which hold the patch
            hi = min(n, lo + step)
            ref = self.coefs[lo // L:(hi + L - 1) // L].view(-1).repeat_interleave(L)[: hi - lo]
            ok = (self.output[lo:hi] - self.comp[lo:hi].to(torch.int32)) == ref
            inside = idx[(idx >= lo) & (idx < hi)]
            ok[inside - lo] = True
            assert bool(ok.all())
        assert bool((self.output[idx] == self.patch_val).all())


class DeltaPatched:
    """C3: Delta, uint64 <- uint16 deltas, segment 4096, anchors = exact prefix at segment starts (base 2^40),
    then 0.1 % naive patches (scatter<4,8,4>) on the same stream."""
    name = "delta<4,8,2>+scatter<4,8,4>"
    kernel = "delta::decompress"

    def __init__(self, n, device, seed=3, segment_length=4096, patch_fraction=0.001):
        g = _gen(seed, device)
        self.n, self.L = n, segment_length
        self.deltas = torch.randint(0, 65536, (n,), dtype=torch.int32, generator=g, device=device).to(torch.int16)  # bit pattern of uint16
        self.csize = 2
        nseg = (n + segment_length - 1) // segment_length
        seg_sums = (self.deltas.view(-1)[: nseg * segment_length if n % segment_length == 0 else n].to(torch.int32) & 0xFFFF)
        if n % segment_length == 0:
            per_segment = seg_sums.view(nseg, segment_length).sum(dim=1, dtype=torch.int64)
        else:
            per_segment = torch.zeros(nseg, dtype=torch.int64, device=device)
            per_segment.index_add_(0, torch.arange(n, device=device) // segment_length, seg_sums.to(torch.int64))
        self.anchors = ((1 << 40) + torch.cumsum(per_segment, 0) - per_segment).contiguous()
        self.p = max(1, int(round(n * patch_fraction)))
        self.patch_idx = torch.sort(torch.randperm(n, generator=_gen(seed + 1, device), device=device)[: self.p])[0].to(torch.int32)
        self.patch_val = torch.randint(0, 1 << 62, (self.p,), dtype=torch.int64, generator=_gen(seed + 2, device), device=device)
        self.output = output_buffer(n, torch.int64, device)
        self.algorithmic_bytes = 10 * n + 8 * nseg             # the dominant kernel (Delta) alone
        self.step_algorithmic_bytes = self.algorithmic_bytes + self.p * 20   # + patches: index, value, scattered store
        self.decompressed_bytes = 8 * n
        st = torch.cuda.current_stream().cuda_stream
        self.config = schemes.delta(self.output, self.deltas, self.anchors, n, segment_length, stream=st)
        self.patch_config = schemes.scatter(self.output, self.patch_val, self.patch_idx, self.p, stream=st)

    def run(self, stream, around_dominant_kernel=None):
        """one step = the Delta decode, then the naive patches on the same stream; `around_dominant_kernel`
        (begin, end) callables bracket the Delta launch alone, for the per-kernel roofline"""
        if around_dominant_kernel:
            around_dominant_kernel[0]()
        schemes.delta(self.output, self.deltas, self.anchors, self.n, self.L, stream=stream, launch_config=self.config)
        if around_dominant_kernel:
            around_dominant_kernel[1]()
        schemes.scatter(self.output, self.patch_val, self.patch_idx, self.p, stream=stream, launch_config=self.patch_config)

    # (the Delta decode alone: what tools/size_probe.py cuts into pieces)
    def column_fields(self):
        return {"compressed_input": self.deltas, "anchor_values": self.anchors, "length": self.n, "segment_length": self.L}

    shard_function = staticmethod(sharding.delta)

    def launch_fields(self, out, f, stream, config):
        return schemes.delta(out, f["compressed_input"], f["anchor_values"], f["length"], f["segment_length"], stream=stream, launch_config=config)

    def verify(self, oracle, sample=1 << 22):
        torch.cuda.synchronize()
        n, L = self.n, self.L
        sample = (sample // L) * L
        pidx = _np(self.patch_idx, np.uint32).astype(np.int64)
        pval = _np(self.patch_val, np.uint64)
        for lo, hi in ((0, min(sample, n)), (max(0, (n - sample) // L * L), n)):
            expect = oracle.delta(_np(self.deltas[lo:hi], np.uint16), _np(self.anchors[lo // L:(hi + L - 1) // L], np.uint64), L)
            inside = (pidx >= lo) & (pidx < hi)
            expect[pidx[inside] - lo] = pval[inside]
            assert (expect == _np(self.output[lo:hi], np.uint64)).all(), "delta+patch: sample mismatch vs oracle"
        # full size: un-patched positions satisfy out[p] - out[p-1] == delta[p] inside a segment; patched hold the patch
        assert bool((self.output[self.patch_idx.to(torch.int64)] == self.patch_val).all())
        step = 1 << 27
        is_patch = torch.zeros(n, dtype=torch.bool, device=self.output.device)
        is_patch[self.patch_idx.to(torch.int64)] = True
        for lo in range(0, n, step):
            hi = min(n, lo + step)
            a = max(lo, 1)
            d = self.output[a:hi] - self.output[a - 1:hi - 1]
            want = self.deltas[a:hi].to(torch.int64) & 0xFFFF
            pos = torch.arange(a, hi, device=d.device)
            skip = is_patch[a:hi] | is_patch[a - 1:hi - 1] | (pos % L == 0)
            assert bool(((d == want) | skip).all())
        heads = torch.arange(0, n, L, device=self.output.device)
        ok = (self.output[heads] == self.anchors + (self.deltas[heads].to(torch.int64) & 0xFFFF)) | is_patch[heads]
        assert bool(ok.all())

    def cpu_run(self, oracle, sample, threads):
        sample = (sample // self.L) * self.L
        d = _np(self.deltas[:sample], np.uint16).copy()
        a = _np(self.anchors[: sample // self.L], np.uint64).copy()
        out = np.empty(sample, dtype=np.uint64)
        return (lambda: oracle.delta(d, a, self.L, threads=threads, out=out)), 8 * sample


class DeltaFused(DeltaPatched):
    """C3 with the patches merged inside the Delta kernel (the SURVEY.md §8f.1 extension): one launch, no scattered
    read-modify-write.  Same inputs and same expected output as DeltaPatched."""
    name = "delta::patched<4,8,2>"
    kernel = "delta::patched::decompress"

    def __init__(self, n, device, **kw):
        super().__init__(n, device, **kw)
        self.algorithmic_bytes = 10 * n + 8 * ((n + self.L - 1) // self.L) + self.p * 12   # + patch position and value, read once
        self.step_algorithmic_bytes = self.algorithmic_bytes
        self.config = schemes.patched_delta(self.output, self.deltas, self.anchors, self.patch_val, self.patch_idx, n, self.L, self.p,
                                            stream=torch.cuda.current_stream().cuda_stream)

    def run(self, stream):
        schemes.patched_delta(self.output, self.deltas, self.anchors, self.patch_val, self.patch_idx, self.n, self.L, self.p,
                              stream=stream, launch_config=self.config)


def _compressed_positions_on_device(patch_idx, period, device):
    """formats.compressed_indices of the (ascending, distinct) patch positions, as device tensors"""
    from libgiddy_amd import formats
    c = formats.compressed_indices(_np(patch_idx, np.uint32), np.zeros(0, np.uint8), period)
    return {k: (torch.from_numpy(v.view({1: np.uint8, 4: np.int32}[v.dtype.itemsize])).to(device) if isinstance(v, np.ndarray) and v.size else v)
            for k, v in c.items() if k != "data_to_scatter"}


class DeltaCompressedPatches(DeltaPatched):
    """C3 with the patch positions in the compressed-indices form (runs of baseline + 8/16/32-bit offsets, anchored
    every 1024 patches): Delta, then the compressed-indices scatter on the same stream — the reference's pipeline."""
    name = "delta<4,8,2>+scatter::compressed_indices<4,8,4>"

    def __init__(self, n, device, anchoring_period=1024, **kw):
        super().__init__(n, device, **kw)
        self.positions = _compressed_positions_on_device(self.patch_idx, anchoring_period, device)
        c = self.positions
        self.position_bytes = sum(v.numel() * v.element_size() for v in c.values() if torch.is_tensor(v))
        self.step_algorithmic_bytes = self.algorithmic_bytes + self.p * 16 + self.position_bytes
        self.patch_config = self._scatter(torch.cuda.current_stream().cuda_stream, None)

    def _scatter(self, stream, config):
        c = self.positions
        return schemes.compressed_indices_scatter(
            self.output, self.patch_val, c["scatter_position_run_lengths"], c["scatter_position_run_individual_offset_sizes"],
            c["scatter_position_run_baseline_values"], c["scatter_position_run_offsets_start_pos"], c["scatter_position_offset_bytes"],
            c["scatter_position_anchors"], c["anchoring_period"], c["num_scatter_position_runs"], self.p,
            output_index_size=4, element_size=8, input_index_size=4, stream=stream, launch_config=config)

    def run(self, stream, around_dominant_kernel=None):
        if around_dominant_kernel:
            around_dominant_kernel[0]()
        schemes.delta(self.output, self.deltas, self.anchors, self.n, self.L, stream=stream, launch_config=self.config)
        if around_dominant_kernel:
            around_dominant_kernel[1]()
        self._scatter(stream, self.patch_config)


class DeltaFusedCompressedPatches(DeltaCompressedPatches):
    """... and with those compressed positions handed to the fused Delta wrapper: a small kernel expands them into a
    scratch array (bytes proportional to the patches), then the one pass over the column merges them"""
    name = "delta::patched<4,8,2> (compressed positions)"
    kernel = "delta::patched::decompress"

    def __init__(self, n, device, **kw):
        super().__init__(n, device, **kw)
        self.positions["patch_positions_scratch"] = torch.empty(self.p, dtype=torch.int32, device=device)
        self.algorithmic_bytes = 10 * n + 8 * ((n + self.L - 1) // self.L) + self.p * 12
        self.step_algorithmic_bytes = self.algorithmic_bytes + self.position_bytes + 4 * self.p
        self.config = schemes.patched_delta(self.output, self.deltas, self.anchors, self.patch_val, self.positions, n, self.L, self.p,
                                            stream=torch.cuda.current_stream().cuda_stream)

    def run(self, stream):
        schemes.patched_delta(self.output, self.deltas, self.anchors, self.patch_val, self.positions, self.n, self.L, self.p,
                              stream=stream, launch_config=self.config)


class DictionaryFused(Dictionary):
    """C2 plus 0.1 % ascending patches merged inside the Dictionary kernel"""
    name = "dictionary::patched<4,4,1>"
    kernel = "dictionary::patched::decompress"

    def __init__(self, n, device, seed=2, patch_fraction=0.001):
        super().__init__(n, device, seed=seed)
        self.p = max(1, int(round(n * patch_fraction)))
        self.patch_idx = torch.sort(torch.randperm(n, generator=_gen(seed + 1, device), device=device)[: self.p])[0].to(torch.int32)
        self.patch_val = torch.randint(0, 1 << 31, (self.p,), dtype=torch.int32, generator=_gen(seed + 2, device), device=device)
        self.algorithmic_bytes += 8 * self.p
        self.config = schemes.patched_dictionary(self.output, self.keys, self.entries, self.patch_val, self.patch_idx, n, 256, self.p,
                                                 stream=torch.cuda.current_stream().cuda_stream)

    def run(self, stream):
        schemes.patched_dictionary(self.output, self.keys, self.entries, self.patch_val, self.patch_idx, self.n, 256, self.p,
                                   stream=stream, launch_config=self.config)

    def verify(self, oracle, sample=1 << 22):
        torch.cuda.synchronize()
        n, idx = self.n, self.patch_idx.to(torch.int64)
        pidx, pval = _np(self.patch_idx, np.uint32).astype(np.int64), _np(self.patch_val, np.uint32)
        entries = _np(self.entries, np.uint32)
        for lo, hi in ((0, min(sample, n)), (max(0, n - sample), n)):
            expect = oracle.dictionary(_np(self.keys[lo:hi]), entries)
            inside = (pidx >= lo) & (pidx < hi)
            expect[pidx[inside] - lo] = pval[inside]
            assert (expect == _np(self.output[lo:hi], np.uint32)).all(), "dictionary+patches: sample mismatch vs oracle"
        step = 1 << 28
        for lo in range(0, n, step):   # full size, against an independent torch gather + index_put
            hi = min(n, lo + step)
            expect = self.entries[self.keys[lo:hi].to(torch.int64)]
            inside = idx[(idx >= lo) & (idx < hi)]
            expect[inside - lo] = self.output[inside]          # patched positions: checked separately below
            assert bool((expect == self.output[lo:hi]).all())
        assert bool((self.output[idx] == self.patch_val).all())


def _geometric_runs(n, mean_run, device, seed, max_run=65535):
    g = _gen(seed, device)
    est = int(n / mean_run * 1.15) + 1024
    u = torch.rand(est, generator=g, device=device, dtype=torch.float64)
    lengths = torch.clamp(torch.floor(torch.log(u) / np.log(1.0 - 1.0 / mean_run)).to(torch.int64) + 1, 1, max_run)
    ends = torch.cumsum(lengths, 0)
    assert int(ends[-1]) >= n, "not enough runs generated"
    count = int(torch.searchsorted(ends, torch.tensor([n], device=device), right=False)[0]) + 1
    lengths = lengths[:count].clone()
    lengths[-1] -= int(ends[count - 1]) - n
    starts = torch.cumsum(lengths, 0) - lengths
    values = torch.randint(0, 1 << 31, (count,), dtype=torch.int64, generator=g, device=device)
    values = (values * 2 + (torch.arange(count, device=device) & 1)).to(torch.int32)  # adjacent runs distinct
    return lengths, starts, values


class RunLength:
    """C4 (RLE): uint32 values, uint16 lengths i.i.d. geometric mean 32, segment 4096."""
    name = "run_length_encoding<4,4,2>"
    kernel = "run_length_encoding::decompress"

    def __init__(self, n, device, seed=6, segment_length=4096, mean_run=32):
        self.n, self.L = n, segment_length
        lengths, starts, values = _geometric_runs(n, mean_run, device, seed)
        self.num_runs = len(lengths)
        seg_starts = torch.arange(0, n, segment_length, device=device)
        self.num_segments = len(seg_starts)
        anchor = torch.searchsorted(starts, seg_starts, right=True) - 1
        self.starts = starts
        self.run_data = values
        self.run_lengths = lengths.to(torch.int16)
        self.anchors = anchor.to(torch.int32)
        self.intra = (seg_starts - starts[anchor]).to(torch.int16)
        self.output = output_buffer(n, torch.int32, device)
        self.algorithmic_bytes = 4 * n + 6 * self.num_runs + 6 * self.num_segments
        self.decompressed_bytes = 4 * n
        self.config = self._launch(torch.cuda.current_stream().cuda_stream, None)

    def _launch(self, stream, config):
        return schemes.run_length_encoding(self.output, self.run_data, self.run_lengths, self.anchors, self.intra, self.L,
                                           self.num_segments, self.num_runs, self.n, stream=stream, launch_config=config)

    def run(self, stream):
        self._launch(stream, self.config)

    def column_fields(self):
        return {"run_data": self.run_data, "run_lengths": self.run_lengths, "position_anchors": self.anchors, "intra_run_anchor_offsets": self.intra,
                "segment_length": self.L, "num_segments": self.num_segments, "num_element_runs": self.num_runs, "length": self.n}

    shard_function = staticmethod(sharding.run_length_encoding)

    def launch_fields(self, out, f, stream, config):
        return schemes.run_length_encoding(out, f["run_data"], f["run_lengths"], f["position_anchors"], f["intra_run_anchor_offsets"], f["segment_length"],
                                           f["num_segments"], f["num_element_runs"], f["length"], stream=stream, launch_config=config)

    def _expected_slice(self, lo, hi):
        pos = torch.arange(lo, hi, device=self.output.device)
        return self.run_data[torch.searchsorted(self.starts, pos, right=True) - 1]

    def verify(self, oracle, sample=1 << 22):
        torch.cuda.synchronize()
        n, L = self.n, self.L
        sample = min((sample // L) * L, n)
        # oracle on the head of the column (anchors + the runs they reach)
        nseg = (sample + L - 1) // L
        last_run = int(self.anchors[min(nseg, self.num_segments - 1)]) + 1 if nseg < self.num_segments else self.num_runs
        expect = oracle.run_length_encoding(_np(self.run_data[:last_run], np.uint32), _np(self.run_lengths[:last_run], np.uint16),
                                            _np(self.anchors[:nseg], np.uint32), _np(self.intra[:nseg], np.uint16), L, sample)
        assert (expect == _np(self.output[:sample], np.uint32)).all(), "RLE: sample mismatch vs oracle"
        step = 1 << 27
        for lo in range(0, n, step):
            hi = min(n, lo + step)
            assert bool((self._expected_slice(lo, hi) == self.output[lo:hi]).all())

    def cpu_run(self, oracle, sample, threads):
        L = self.L
        sample = min((sample // L) * L, self.n)
        nseg = sample // L
        last_run = int(self.anchors[nseg]) + 1 if nseg < self.num_segments else self.num_runs
        rd = _np(self.run_data[:last_run], np.uint32).copy(); rl = _np(self.run_lengths[:last_run], np.uint16).copy()
        an = _np(self.anchors[:nseg], np.uint32).copy(); it = _np(self.intra[:nseg], np.uint16).copy()
        out = np.empty(sample, dtype=np.uint32)
        return (lambda: oracle.run_length_encoding(rd, rl, an, it, L, sample, threads=threads, out=out)), 4 * sample


class RunPosition(RunLength):
    """C4 (RPE): same runs as RLE, start positions relative to the segment (uint16) or absolute (uint32)."""
    kernel = "run_position_encoding::decompress"

    def __init__(self, n, device, seed=6, segment_length=4096, mean_run=32, relative=True):
        self.relative = relative
        self.name = "run_position_encoding<4,4,%d,%s>" % (2 if relative else 4, "true" if relative else "false")
        self.n, self.L = n, segment_length
        lengths, starts, values = _geometric_runs(n, mean_run, device, seed)
        self.num_runs = len(lengths)
        seg_starts = torch.arange(0, n, segment_length, device=device)
        self.num_segments = len(seg_starts)
        self.starts = starts
        self.run_data = values
        self.positions = (starts % segment_length).to(torch.int16) if relative else starts.to(torch.int32)
        self.anchors = (torch.searchsorted(starts, seg_starts, right=True) - 1).to(torch.int32)
        self.output = output_buffer(n, torch.int32, device)
        psize = 2 if relative else 4
        self.algorithmic_bytes = 4 * n + (4 + psize) * self.num_runs + 4 * self.num_segments
        self.decompressed_bytes = 4 * n
        self.config = self._launch(torch.cuda.current_stream().cuda_stream, None)

    def _launch(self, stream, config):
        return schemes.run_position_encoding(self.output, self.run_data, self.positions, self.anchors, self.L, self.num_segments,
                                             self.num_runs, self.n, position_size=2 if self.relative else 4,
                                             relative=self.relative, stream=stream, launch_config=config)

    def column_fields(self):
        return {"run_data": self.run_data, "run_start_positions": self.positions, "position_anchors": self.anchors, "segment_length": self.L,
                "num_anchors": self.num_segments, "num_element_runs": self.num_runs, "length": self.n}

    def shard_function(self, fields, rank, world):
        return sharding.run_position_encoding(fields, rank, world, relative=self.relative)

    def launch_fields(self, out, f, stream, config):
        return schemes.run_position_encoding(out, f["run_data"], f["run_start_positions"], f["position_anchors"], f["segment_length"], f["num_anchors"],
                                             f["num_element_runs"], f["length"], position_size=2 if self.relative else 4, relative=self.relative,
                                             stream=stream, launch_config=config)

    def verify(self, oracle, sample=1 << 22):
        torch.cuda.synchronize()
        n, L = self.n, self.L
        sample = min((sample // L) * L, n)
        nseg = (sample + L - 1) // L
        last_run = int(self.anchors[nseg]) + 1 if nseg < self.num_segments else self.num_runs
        pd = np.uint16 if self.relative else np.uint32
        anchors = _np(self.anchors[:nseg + (1 if nseg < self.num_segments else 0)], np.uint32)
        expect = oracle.run_position_encoding(_np(self.run_data[:last_run], np.uint32), _np(self.positions[:last_run], pd),
                                              anchors, L, min(sample + (L if nseg < self.num_segments else 0), n), self.relative)[:sample]
        assert (expect == _np(self.output[:sample], np.uint32)).all(), "RPE: sample mismatch vs oracle"
        step = 1 << 27
        for lo in range(0, n, step):
            hi = min(n, lo + step)
            assert bool((self._expected_slice(lo, hi) == self.output[lo:hi]).all())

    def cpu_run(self, oracle, sample, threads):
        L = self.L
        sample = min((sample // L) * L, self.n)
        nseg = sample // L
        last_run = int(self.anchors[nseg]) + 1 if nseg < self.num_segments else self.num_runs
        pd = np.uint16 if self.relative else np.uint32
        rd = _np(self.run_data[:last_run], np.uint32).copy(); ps = _np(self.positions[:last_run], pd).copy()
        an = _np(self.anchors[:nseg + (1 if nseg < self.num_segments else 0)], np.uint32).copy()
        total = min(sample + (L if nseg < self.num_segments else 0), self.n)
        out = np.empty(total, dtype=np.uint32)
        return (lambda: oracle.run_position_encoding(rd, ps, an, L, total, self.relative, threads=threads, out=out)), 4 * total


class DiscardZeroBytesVariable:
    """C5 (per-GPU shard): uint64 out, sizes 1..4 bytes uniform (2-bit fields, min 1), 32-bit containers, segment 1024, 64-bit anchors."""
    name = "discard_zero_bytes::variable_width<8,8,4>"
    kernel = "variable_width::decompress"

    def __init__(self, n, device, seed=8, segment_length=1024, bits=2, min_size=1):
        self.n, self.L, self.bits, self.min_size = n, segment_length, bits, min_size
        self.name = "discard_zero_bytes::variable_width<8,8,4> (%d-bit sizes, %d..%d bytes)" % (bits, min_size, min(8, min_size + (1 << bits) - 1))
        per = 32 // bits
        largest = min(8, min_size + (1 << bits) - 1)
        ncont = (n + per - 1) // per
        block = (1 << 26) // (per * segment_length) * (per * segment_length) or per * segment_length   # whole containers and segments
        # generated block by block (a 2^31-element column's temporaries would not fit otherwise): block b draws from generator (seed, b)
        self.values = torch.empty(n, dtype=torch.int64, device=device)
        self.packed = torch.empty(ncont, dtype=torch.int32, device=device)
        self.anchors = torch.empty((n + segment_length - 1) // segment_length, dtype=torch.int64, device=device)
        pieces, total = [], 0
        for index, lo in enumerate(range(0, n, block)):
            hi = min(n, lo + block)
            g = _gen(seed + 7919 * index, device)
            sizes = torch.randint(min_size, min_size + (1 << bits), (hi - lo,), dtype=torch.int64, generator=g, device=device).clamp_(max=8)
            offsets = torch.cumsum(sizes, 0) - sizes
            bytes_here = int(offsets[-1] + sizes[-1])
            values = torch.randint(0, 1 << 62, (hi - lo,), dtype=torch.int64, generator=g, device=device)
            mask = torch.where(sizes >= 8, torch.full_like(sizes, -1), (torch.ones_like(sizes) << (8 * sizes)) - 1)
            values = ((values & mask) | (torch.ones_like(sizes) << (8 * (sizes - 1)).clamp_(min=0))) & mask   # exactly `size` significant bytes
            self.values[lo:hi] = values
            # packed size fields, LSB first, `per` fields per 32-bit container
            containers = (hi - lo + per - 1) // per
            fields = torch.zeros(containers * per, dtype=torch.int64, device=device)
            fields[: hi - lo] = sizes - min_size
            shifts = torch.arange(per, device=device, dtype=torch.int64) * bits
            self.packed[lo // per: lo // per + containers] = (fields.view(containers, per) << shifts).sum(dim=1).to(torch.int32)
            # compressed bytes: byte k of element i at offsets[i] + k for k < sizes[i]
            data = torch.empty(bytes_here, dtype=torch.uint8, device=device)
            for k in range(largest):
                sel = sizes > k
                data[(offsets[sel] + k)] = ((values[sel] >> (8 * k)) & 0xFF).to(torch.uint8)
            pieces.append(data)
            self.anchors[lo // segment_length: (hi + segment_length - 1) // segment_length] = offsets[::segment_length] + total
            total += bytes_here
            del sizes, offsets, values, mask, fields, data
        self.data = torch.cat(pieces) if len(pieces) > 1 else pieces[0]
        del pieces
        self.output = output_buffer(n, torch.int64, device)
        self.total_bytes = total
        self.algorithmic_bytes = total + 4 * ncont + 8 * n + 8 * len(self.anchors)
        self.decompressed_bytes = 8 * n
        self.config = self._launch(torch.cuda.current_stream().cuda_stream, None)

    def _launch(self, stream, config):
        return schemes.discard_zero_bytes_variable(self.output, self.data, self.packed, self.anchors, self.L, self.n, self.min_size,
                                                   self.bits, stream=stream, launch_config=config)

    def run(self, stream):
        self._launch(stream, self.config)

    # the column in the format writers' vocabulary (libgiddy_amd.formats), device-resident; how a shard of it is cut and launched
    def column_fields(self):
        return {"compressed_data": self.data, "packed_element_sizes": self.packed, "position_anchors": self.anchors, "segment_length": self.L,
                "length": self.n, "min_represented_element_size": self.min_size, "bits_per_element_size": self.bits}

    shard_function = staticmethod(sharding.discard_zero_bytes_variable)

    def launch_fields(self, out, f, stream, config):
        return schemes.discard_zero_bytes_variable(out, f["compressed_data"], f["packed_element_sizes"], f["position_anchors"], f["segment_length"],
                                                   f["length"], self.min_size, self.bits, stream=stream, launch_config=config)

    def verify(self, oracle, sample=1 << 21):
        torch.cuda.synchronize()
        n, L = self.n, self.L
        sample = min((sample // L) * L, n)
        nbytes = int(self.anchors[sample // L]) if sample < n else self.total_bytes
        expect = oracle.dzb_variable(_np(self.data[:nbytes]), _np(self.packed[: -(-sample // (32 // self.bits)) + 1], np.uint32),
                                     _np(self.anchors[: (sample + L - 1) // L], np.uint64), L, sample, self.min_size, self.bits, np.uint64)
        assert (expect == _np(self.output[:sample], np.uint64)).all(), "DZB-V: sample mismatch vs oracle"
        assert bool((self.output == self.values).all())

    def cpu_run(self, oracle, sample, threads):
        L = self.L
        sample = min((sample // L) * L, self.n)
        nbytes = int(self.anchors[sample // L]) if sample < self.n else self.total_bytes
        data = _np(self.data[:nbytes]).copy(); packed = _np(self.packed[: -(-sample // (32 // self.bits)) + 1], np.uint32).copy()
        an = _np(self.anchors[: sample // L], np.uint64).copy()
        out = np.empty(sample, dtype=np.uint64)
        return (lambda: oracle.dzb_variable(data, packed, an, L, sample, self.min_size, self.bits, np.uint64, threads=threads, out=out)), 8 * sample


class DiscardZeroBytesFixed:
    """DZB-fixed uint64 <- uint16 (coverage sweep, not a BASELINE config)."""
    name = "discard_zero_bytes::fixed_width<4,8,2>"
    kernel = "fixed_width::decompress"

    def __init__(self, n, device, seed=9):
        self.n = n
        self.comp = torch.randint(-32768, 32768, (n,), dtype=torch.int16, generator=_gen(seed, device), device=device)
        self.output = output_buffer(n, torch.int64, device)
        self.algorithmic_bytes = 10 * n
        self.decompressed_bytes = 8 * n
        self.config = schemes.discard_zero_bytes_fixed(self.output, self.comp, n, uncompressed_size=8, compressed_size=2,
                                                       stream=torch.cuda.current_stream().cuda_stream)

    def run(self, stream):
        schemes.discard_zero_bytes_fixed(self.output, self.comp, self.n, uncompressed_size=8, compressed_size=2, stream=stream, launch_config=self.config)

    def verify(self, oracle, sample=1 << 22):
        torch.cuda.synchronize()
        s = min(sample, self.n)
        assert (oracle.dzb_fixed(_np(self.comp[:s], np.uint16), np.uint64) == _np(self.output[:s], np.uint64)).all()
        assert bool((self.output == (self.comp.to(torch.int64) & 0xFFFF)).all())

    def cpu_run(self, oracle, sample, threads):
        c = _np(self.comp[:sample], np.uint16).copy()
        return (lambda: oracle.dzb_fixed(c, np.uint64, threads=threads)), 8 * sample


class Model:
    """Model int32, linear (coverage sweep): a pure store stream."""
    name = "model<4,int,linear>"
    kernel = "model::decompress"

    def __init__(self, n, device, seed=10):
        self.n = n
        self.coefs = np.array([-12345, 37], dtype=np.int32)
        self.output = output_buffer(n, torch.int32, device)
        self.algorithmic_bytes = 4 * n
        self.decompressed_bytes = 4 * n
        self.config = schemes.model(self.output, self.coefs, n, stream=torch.cuda.current_stream().cuda_stream)

    def run(self, stream):
        schemes.model(self.output, self.coefs, self.n, stream=stream, launch_config=self.config)

    def verify(self, oracle, sample=1 << 22):
        torch.cuda.synchronize()
        s = min(sample, self.n)
        assert (oracle.model(self.coefs, s, np.int32, "linear") == _np(self.output[:s])).all()
        expect = (torch.arange(self.n, device=self.output.device, dtype=torch.int64) * 37 - 12345).to(torch.int32)
        assert bool((self.output == expect).all())

    def cpu_run(self, oracle, sample, threads):
        return (lambda: oracle.model(self.coefs, sample, np.int32, "linear", threads=threads)), 4 * sample


def _ascending_patches(n, fraction, value_dtype, high, device, seed):
    p = max(1, int(round(n * fraction)))
    idx = torch.sort(torch.randperm(n, generator=_gen(seed, device), device=device)[:p])[0]
    val = torch.randint(0, high, (p,), dtype=value_dtype, generator=_gen(seed + 1, device), device=device)
    return p, idx, val


class ModelFused(Model):
    """Model int32 plus 0.1 % ascending patches merged inside the kernel — how a modelled column is stored"""
    name = "model::patched<4,int,linear>"
    kernel = "model::patched::decompress"

    def __init__(self, n, device, seed=10, patch_fraction=0.001):
        super().__init__(n, device, seed=seed)
        self.p, self.patch_idx64, self.patch_val = _ascending_patches(n, patch_fraction, torch.int32, 1 << 31, device, seed + 5)
        self.patch_idx = self.patch_idx64.to(torch.int32)
        self.algorithmic_bytes += 8 * self.p
        self.config = schemes.patched_model(self.output, self.coefs, self.patch_val, self.patch_idx, n, self.p,
                                            stream=torch.cuda.current_stream().cuda_stream)

    def run(self, stream):
        schemes.patched_model(self.output, self.coefs, self.patch_val, self.patch_idx, self.n, self.p, stream=stream, launch_config=self.config)

    def verify(self, oracle, sample=1 << 22):
        torch.cuda.synchronize()
        s = min(sample, self.n)
        expect = oracle.model(self.coefs, s, np.int32, "linear")
        inside = _np(self.patch_idx64[self.patch_idx64 < s], np.int64)
        expect[inside] = _np(self.patch_val[: len(inside)])
        assert (expect == _np(self.output[:s])).all(), "fused model: sample mismatch vs oracle + scatter"
        full = (torch.arange(self.n, device=self.output.device, dtype=torch.int64) * 37 - 12345).to(torch.int32)
        full[self.patch_idx64] = self.patch_val
        assert bool((self.output == full).all())


class DiscardZeroBytesFixedFused(DiscardZeroBytesFixed):
    """DZB-fixed uint64 <- uint16 plus 0.1 % ascending patches (the elements that do not fit two bytes) merged inside the kernel"""
    name = "discard_zero_bytes::fixed_width::patched<4,8,2>"
    kernel = "fixed_width::patched::decompress"

    def __init__(self, n, device, seed=9, patch_fraction=0.001):
        super().__init__(n, device, seed=seed)
        self.p, self.patch_idx64, self.patch_val = _ascending_patches(n, patch_fraction, torch.int64, 1 << 62, device, seed + 5)
        self.patch_idx = self.patch_idx64.to(torch.int32)
        self.algorithmic_bytes += 12 * self.p
        self.config = schemes.patched_discard_zero_bytes_fixed(self.output, self.comp, self.patch_val, self.patch_idx, n, self.p,
                                                               stream=torch.cuda.current_stream().cuda_stream)

    def run(self, stream):
        schemes.patched_discard_zero_bytes_fixed(self.output, self.comp, self.patch_val, self.patch_idx, self.n, self.p,
                                                 stream=stream, launch_config=self.config)

    def verify(self, oracle, sample=1 << 22):
        torch.cuda.synchronize()
        s = min(sample, self.n)
        expect = oracle.dzb_fixed(_np(self.comp[:s], np.uint16), np.uint64)
        inside = _np(self.patch_idx64[self.patch_idx64 < s], np.int64)
        expect[inside] = _np(self.patch_val[: len(inside)], np.uint64)
        assert (expect == _np(self.output[:s], np.uint64)).all(), "fused DZB-fixed: sample mismatch vs oracle + scatter"
        full = self.comp.to(torch.int64) & 0xFFFF
        full[self.patch_idx64] = self.patch_val
        assert bool((self.output == full).all())


class IncidenceBitmaps:
    """Incidence bitmaps, 16 distinct uint32 values (coverage sweep)."""
    name = "incidence_bitmaps<4,4,2,true>"
    kernel = "incidence_bitmaps::decompress"

    def __init__(self, n, device, seed=11, num_values=16):
        g = _gen(seed, device)
        self.n, self.nb = n, num_values
        words = n // 32
        self.words = words
        self.keys = torch.randint(0, num_values, (n,), dtype=torch.int64, generator=g, device=device)
        self.entries = (torch.randperm(1 << 20, generator=g, device=device)[:num_values] * 1009 + 3).to(torch.int32)
        aligned = (words + 1) // 2 * 2
        bitmaps = torch.zeros((num_values, aligned), dtype=torch.int32, device=device)
        shifts = torch.arange(32, device=device, dtype=torch.int64)
        for b in range(num_values):
            bits = (self.keys == b).view(words, 32).to(torch.int64)
            word = (bits << shifts).sum(dim=1)
            bitmaps[b, :words] = torch.where(word >= (1 << 31), word - (1 << 32), word).to(torch.int32)
        self.bitmaps = bitmaps.view(-1)
        self.output = output_buffer(n, torch.int32, device)
        self.algorithmic_bytes = 4 * n + num_values * 4 * words
        self.decompressed_bytes = 4 * n
        self.config = schemes.incidence_bitmaps(self.output, self.bitmaps, self.entries, words, num_values,
                                                stream=torch.cuda.current_stream().cuda_stream)

    def run(self, stream):
        schemes.incidence_bitmaps(self.output, self.bitmaps, self.entries, self.words, self.nb, stream=stream, launch_config=self.config)

    def verify(self, oracle, sample=1 << 20):
        torch.cuda.synchronize()
        assert bool((self.output == self.entries[self.keys]).all())

    def cpu_run(self, oracle, sample, threads):
        words = min(sample // 32, self.words)
        aligned = (self.words + 1) // 2 * 2
        bm = _np(self.bitmaps, np.uint32).reshape(self.nb, aligned)[:, :words]
        al = (words + 1) // 2 * 2
        laid = np.zeros((self.nb, al), dtype=np.uint32); laid[:, :words] = bm
        ent = _np(self.entries, np.uint32).copy()
        flat = laid.reshape(-1)
        return (lambda: oracle.incidence_bitmaps(flat, ent, words, self.nb, 2, np.uint32, threads=threads)), 4 * 32 * words

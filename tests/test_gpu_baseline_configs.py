"""-m gpu: the BASELINE.json configurations AT THEIR FULL SIZES, through the C ABI, bit-exact.

Inputs are generated on the device (tools/workloads.py — the same seeded workloads bench.py times).  Every case is
checked three ways:
  * `verify`: the oracle on a head (and tail) sample + full-size torch identities that do not depend on the size
    (an independent gather / searchsorted / prefix-difference statement of the whole column);
  * oracle WINDOWS at the head, the middle and the tail: the window is cut out of the device-resident column with
    libgiddy_amd.sharding (a window is shard k of 1024: views + rebased anchors), copied to the host, decoded by the
    oracle and compared with the same range of the GPU output — so the device-side cut is exercised at full size too;
  * for IndexSize = 8: run positions and outputs beyond 2^32.
C2 Dictionary, C3 Delta<4,8,2> + 0.1 % patches (two launches and fused), C4 RLE<4,4,2> / RPE<4,4,2,true> /
RPE<4,4,4,false> at 2^30 outputs with mean run 32, C5 DZB-variable<8,8,4> at 2^28 (2-bit and 3-bit size fields)."""
import os
import sys

import numpy as np
import pytest

from libgiddy_amd import sharding
from oracle import oracle

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

WINDOWS = 1024            # a window = shard k of this many
WHICH = (0, 511, 1023)    # head, middle, tail


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _host(value, dtype=None):
    if isinstance(value, torch.Tensor):
        a = value.detach().cpu().numpy()
        return np.ascontiguousarray(a.view(dtype) if dtype is not None else a)
    return value


def _windows(work, decode, out_dtype):
    fields = work.column_fields()
    for k in WHICH:
        shard = work.shard_function(fields, k, WINDOWS)
        assert shard.stop > shard.start
        expect = decode(shard.fields)
        got = _host(work.output[shard.start:shard.stop], out_dtype)
        bad = np.flatnonzero(expect != got)
        assert len(bad) == 0, (work.name, "window", k, shard.start + bad[:4], got[bad[:4]], expect[bad[:4]])


def test_c2_dictionary_full_size():
    import workloads as w
    for zipf in (False, True):
        work = w.Dictionary(1 << 30, torch.device("cuda"), zipf=zipf)
        work.output.fill_(-1)
        work.run(_stream())
        work.verify(oracle)
        entries = _host(work.entries, np.uint32)
        n = work.n
        for lo in (n // 2 - 12345, n // 2 + (1 << 20)):   # middle windows (verify does head and tail)
            assert (oracle.dictionary(_host(work.keys[lo:lo + (1 << 20)]), entries) == _host(work.output[lo:lo + (1 << 20)], np.uint32)).all()
        del work
        torch.cuda.empty_cache()


@pytest.mark.parametrize("fused", [False, True])
def test_c3_delta_and_patches_full_size(fused):
    """Delta<4,8,2>, 2^30 elements, segments of 4096, 1 073 742 patches — scatter<4,8,4> as a second launch, and fused"""
    import workloads as w
    work = (w.DeltaFused if fused else w.DeltaPatched)(1 << 30, torch.device("cuda"))
    assert work.p == 1073742
    work.output.fill_(-1)
    work.run(_stream())
    work.verify(oracle)
    fields = {"compressed_input": work.deltas, "anchor_values": work.anchors, "segment_length": work.L, "length": work.n}
    patches = {"indices": work.patch_idx, "data": work.patch_val}
    for k in WHICH:
        shard = sharding.delta(fields, k, WINDOWS)
        expect = oracle.delta(_host(shard.fields["compressed_input"], np.uint16), _host(shard.fields["anchor_values"], np.uint64), work.L)
        inside = sharding.patches(patches, shard.start, shard.stop)
        assert inside["data_length"] > 0
        expect[_host(inside["indices"], np.uint32).astype(np.int64)] = _host(inside["data"], np.uint64)
        assert (expect == _host(work.output[shard.start:shard.stop], np.uint64)).all(), (work.name, k)


@pytest.mark.parametrize("scheme", ["rle", "rpe_relative", "rpe_absolute"])
def test_c4_runs_full_size(scheme):
    """RLE<4,4,2>, RPE<4,4,2,true>, RPE<4,4,4,false>: 2^30 outputs, i.i.d. geometric runs of mean 32, segments of 4096"""
    import workloads as w
    device = torch.device("cuda")
    if scheme == "rle":
        work = w.RunLength(1 << 30, device)
        decode = lambda f: oracle.run_length_encoding(_host(f["run_data"], np.uint32), _host(f["run_lengths"], np.uint16), _host(f["position_anchors"], np.uint32),
                                                      _host(f["intra_run_anchor_offsets"], np.uint16), f["segment_length"], f["length"])
    else:
        relative = scheme == "rpe_relative"
        work = w.RunPosition(1 << 30, device, relative=relative)
        decode = lambda f: oracle.run_position_encoding(_host(f["run_data"], np.uint32), _host(f["run_start_positions"], np.uint16 if relative else np.uint32),
                                                        _host(f["position_anchors"], np.uint32), f["segment_length"], f["length"], relative)
    assert 30e6 < work.num_runs < 37e6   # mean run 32
    work.output.fill_(-1)
    work.run(_stream())
    work.verify(oracle)
    _windows(work, decode, np.uint32)


@pytest.mark.parametrize("bits", [2, 3])
def test_c5_dzb_variable_full_size(bits):
    """DZB-variable<8,8,4>: 2^28 elements (one GPU's share of C5), sizes 1..4 (2-bit fields) and 1..8 (3-bit), segments of 1024, 64-bit anchors"""
    import workloads as w
    work = w.DiscardZeroBytesVariable(1 << 28, torch.device("cuda"), bits=bits)
    work.output.fill_(-1)
    work.run(_stream())
    work.verify(oracle)
    decode = lambda f: oracle.dzb_variable(_host(f["compressed_data"]), _host(f["packed_element_sizes"], np.uint32), _host(f["position_anchors"], np.uint64),
                                           f["segment_length"], f["length"], work.min_size, work.bits, np.uint64)
    _windows(work, decode, np.uint64)


def test_run_positions_and_outputs_beyond_32_bits():
    """RLE<8,4,2,8> and RPE<8,4,8,false,8> over 2^32 + 12 288 outputs (16 GiB): anchors, positions and output indices cross 2^32"""
    from libgiddy_amd import schemes
    import workloads as w
    n, L = (1 << 32) + 3 * 4096, 4096
    device = torch.device("cuda")
    try:
        lengths, starts, values = w._geometric_runs(n, 64, device, seed=31)
        out = torch.empty(n, dtype=torch.int32, device=device)
    except torch.OutOfMemoryError:
        pytest.skip("not enough device memory")
    num_runs = len(lengths)
    seg_starts = torch.arange(0, n, L, device=device)
    num_segments = len(seg_starts)
    anchors = torch.searchsorted(starts, seg_starts, right=True) - 1      # int64: RunIndexSize 8
    intra = (seg_starts - starts[anchors]).to(torch.int16)

    def check():
        torch.cuda.synchronize()
        step = 1 << 27
        for lo in list(range(0, 1 << 28, step)) + list(range((1 << 32) - step, n, step)):   # the first 2^28 and everything from 2^32 - 2^27 on
            hi = min(n, lo + step)
            pos = torch.arange(lo, hi, device=device)
            assert bool((values[torch.searchsorted(starts, pos, right=True) - 1] == out[lo:hi]).all()), lo
        tail = num_segments - 4   # oracle on the last four segments (positions beyond 2^32)
        r0 = int(anchors[tail])
        expect = oracle.run_length_encoding(_host(values[r0:], np.uint32), _host(lengths[r0:].to(torch.int16), np.uint16),
                                            _host(anchors[tail:] - r0).astype(np.uint64), _host(intra[tail:], np.uint16), L, n - tail * L)
        assert (expect == _host(out[tail * L:], np.uint32)).all()

    out.fill_(-1)
    schemes.run_length_encoding(out, values, lengths.to(torch.int16), anchors, intra, L, num_segments, num_runs, n, index_size=8, run_index_size=8, stream=_stream())
    check()
    out.fill_(-1)
    schemes.run_position_encoding(out, values, starts, anchors, L, num_segments, num_runs, n, position_size=8, relative=False, index_size=8, run_index_size=8,
                                  stream=_stream())
    check()


@pytest.mark.parametrize("scheme", ["delta", "run_length_encoding", "run_position_encoding", "discard_zero_bytes_variable", "dictionary"])
def test_compress_then_decompress_at_full_size(scheme):
    """the size-independent property the device-side compressors make available: decompress(compress(column)) == column, at the BASELINE
    sizes (2^30 elements; DZB-variable 2^28), compressed ON THE DEVICE into the reference's formats and decoded by the kernels under test"""
    import workloads as w
    from libgiddy_amd import compressors, schemes
    device, st = torch.device("cuda"), _stream()
    n = 1 << (28 if scheme == "discard_zero_bytes_variable" else 30)
    gen = torch.Generator(device=device); gen.manual_seed(77)
    if scheme == "delta":
        column = torch.cumsum(torch.randint(0, 65536, (n,), dtype=torch.int64, generator=gen, device=device), 0) + (1 << 40)
        f = compressors.delta(column, 4096, np.uint16)
        out = torch.empty_like(column)
        schemes.delta(out, f["compressed_input"], f["anchor_values"], n, 4096, stream=st)
    elif scheme in ("run_length_encoding", "run_position_encoding"):
        lengths, starts, values = w._geometric_runs(n, 32, device, seed=41)
        column = values[torch.searchsorted(starts, torch.arange(n, device=device), right=True) - 1]
        del lengths, starts, values
        out = torch.empty_like(column)
        if scheme == "run_length_encoding":
            f = compressors.run_length_encoding(column, 4096, np.uint16, capacity=n // 8)
            schemes.run_length_encoding(out, f["run_data"], f["run_lengths"], f["position_anchors"], f["intra_run_anchor_offsets"], 4096, f["num_segments"], f["num_element_runs"], n, stream=st)
        else:
            f = compressors.run_position_encoding(column, 4096, np.uint16, True, capacity=n // 8)
            schemes.run_position_encoding(out, f["run_data"], f["run_start_positions"], f["position_anchors"], 4096, f["num_anchors"], f["num_element_runs"], n, stream=st)
        assert 30e6 < f["num_element_runs"] < 37e6
    elif scheme == "discard_zero_bytes_variable":
        sizes = torch.randint(1, 5, (n,), dtype=torch.int64, generator=gen, device=device)
        column = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, generator=gen, device=device) & ((torch.ones_like(sizes) << (8 * sizes)) - 1)
        del sizes
        f = compressors.discard_zero_bytes_variable(column, 1024, 1, 2, np.uint32, index_size=8)
        out = torch.empty_like(column)
        schemes.discard_zero_bytes_variable(out, f["compressed_data"], f["packed_element_sizes"], f["position_anchors"], 1024, n, 1, 2, stream=st)
    else:
        entries = torch.sort((torch.randperm(1 << 24, generator=gen, device=device)[:256] * 101 + 7).to(torch.int32))[0]
        column = entries[torch.randint(0, 256, (n,), generator=gen, device=device)]
        f = compressors.dictionary(column, np.uint8, entries=entries)
        out = torch.empty_like(column)
        schemes.dictionary(out, f["compressed_input"], f["dictionary_entries"], n, 256, stream=st)
    torch.cuda.synchronize()
    assert bool((out == column).all()), scheme

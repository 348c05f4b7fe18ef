"""-m gpu: sizes and offsets beyond 32 bits (the IndexSize = 8 instantiations), checked through
size-independent properties plus oracle samples."""
import numpy as np
import pytest

import cases
from libgiddy_amd import formats
from oracle import oracle

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def test_dictionary_more_than_2_to_32_elements():
    """dictionary<8,4,1> over 2^32 + 4099 elements (16 GiB of output): sampled windows vs the oracle
    (head, across the 2^32 boundary, ragged tail) and the histogram identity over the whole column."""
    import gpu_util as g
    from libgiddy_amd import schemes
    n = (1 << 32) + 4099
    try:
        gen = torch.Generator(device="cuda"); gen.manual_seed(5)
        keys = torch.randint(0, 256, (n,), dtype=torch.uint8, generator=gen, device="cuda")
        entries = (torch.randperm(1 << 24, generator=gen, device="cuda")[:256] * 131 + 17).to(torch.int32)
        out = torch.empty(n, dtype=torch.int32, device="cuda")
    except torch.OutOfMemoryError:
        pytest.skip("not enough device memory")
    schemes.dictionary(out, keys, entries, n, 256, index_size=8, stream=g.stream())
    torch.cuda.synchronize()
    e = entries.cpu().numpy().view(np.uint32)
    for lo, hi in ((0, 1 << 20), ((1 << 32) - (1 << 19), (1 << 32) + 4099), (n - 5000, n)):
        assert (oracle.dictionary(keys[lo:hi].cpu().numpy(), e) == out[lo:hi].cpu().numpy().view(np.uint32)).all()
    counts = torch.zeros(256, dtype=torch.int64, device="cuda")
    total = 0
    step = 1 << 30
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        counts += torch.bincount(keys[lo:hi].to(torch.int64), minlength=256)
        total += int(out[lo:hi].sum(dtype=torch.int64))
    assert int((counts * entries.to(torch.int64)).sum()) == total


def test_patched_dictionary_more_than_2_to_32_elements():
    """the fused decompress-and-patch kernel with 64-bit patch positions on both sides of 2^32"""
    from libgiddy_amd import schemes
    n = (1 << 32) + 4099
    try:
        gen = torch.Generator(device="cuda"); gen.manual_seed(8)
        keys = torch.randint(0, 256, (n,), dtype=torch.uint8, generator=gen, device="cuda")
        entries = (torch.randperm(1 << 24, generator=gen, device="cuda")[:256] * 131 + 17).to(torch.int32)
        out = torch.empty(n, dtype=torch.int32, device="cuda")
        num_patches = 1 << 20
        positions = torch.sort(torch.randint(0, n, (num_patches,), dtype=torch.int64, generator=gen, device="cuda"))[0]
        positions[-3:] = torch.tensor([n - 3, n - 2, n - 1], device="cuda")            # the very end of the column
        positions[:2] = torch.tensor([0, 1], device="cuda")
        positions = torch.sort(positions)[0]
        values = torch.randint(-(1 << 31), 1 << 31, (num_patches,), dtype=torch.int32, generator=gen, device="cuda")
    except torch.OutOfMemoryError:
        pytest.skip("not enough device memory")
    schemes.patched_dictionary(out, keys, entries, values, positions, n, 256, num_patches, uncompressed_size=4, dictionary_index_size=1,
                               index_size=8, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    # patched positions hold the LAST patch for that position; everything else is the plain decode
    last = torch.ones(num_patches, dtype=torch.bool, device="cuda")
    last[:-1] = positions[1:] != positions[:-1]
    assert bool((out[positions[last]] == values[last]).all())
    assert int((positions >= (1 << 32)).sum()) > 0
    step = 1 << 30
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        expect = entries[keys[lo:hi].to(torch.int64)]
        inside = positions[(positions >= lo) & (positions < hi)]
        expect[inside - lo] = out[inside]
        assert bool((expect == out[lo:hi]).all())


def test_dzb_variable_byte_offsets_beyond_32_bits():
    """discard_zero_bytes::variable_width<8,8,4>: position anchors above 2^33.  The compressed bytes live in
    an ordinary buffer; the kernel is handed (buffer - 2^33) as compressed_data and anchors + 2^33, so every
    address it forms needs the full 64-bit anchor (no memory near the fake base is ever touched)."""
    import gpu_util as g
    from libgiddy_amd import schemes
    n, seglen, bias = 200000, 1024, 1 << 33
    col, sizes = cases.dzbv_column(n, np.uint64, 1, 2, seed=21)
    f = formats.discard_zero_bytes_variable(col, seglen, 1, 2, np.uint32, 8, element_sizes=sizes)
    data = g.dev(f["compressed_data"])
    anchors = g.dev(f["position_anchors"] + np.uint64(bias))
    out = g.dev_empty(n, np.uint64)
    schemes.discard_zero_bytes_variable(out, data.data_ptr() - bias, g.dev(f["packed_element_sizes"]), anchors, seglen, n, 1, 2,
                                        index_size=8, stream=g.stream())
    assert (g.host(out, np.uint64, n) == col).all()


def test_delta_and_rle_index_size_8_large_positions():
    """IndexSize = 8 instantiations of Delta and RLE/RPE on a 2^27-element column, full compare on device."""
    import gpu_util as g
    from libgiddy_amd import schemes
    n, L = 1 << 27, 4096
    gen = torch.Generator(device="cuda"); gen.manual_seed(9)
    d = torch.randint(0, 256, (n,), dtype=torch.uint8, generator=gen, device="cuda")
    pref = torch.cumsum(d.to(torch.int64), 0)
    anchors = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), pref[L - 1:-1:L]]).contiguous()
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    schemes.delta(out, d, anchors, n, L, uncompressed_size=8, compressed_size=1, index_size=8, stream=g.stream())
    torch.cuda.synchronize()
    assert bool((out == pref).all())


@pytest.mark.parametrize("usize,csize,index_size", [(8, 2, 4), (4, 1, 4), (8, 4, 8), (2, 1, 4)])
def test_delta_with_very_long_segments_in_parts(usize, csize, index_size):
    """the Delta wrapper's extension for long segments ("part_anchors_scratch"): segments of 2^17 elements or more are decoded in
    parts — the parts' delta sums, the value before each part, then the ordinary decoder with the parts as its segments.  One giant
    segment, several long ones, lengths that are not a multiple of the part, a ragged last segment, and a length no part divides
    (which takes the ordinary path)"""
    import gpu_util as g
    from libgiddy_amd import schemes
    ud, cd = np.dtype("u%d" % usize), np.dtype("u%d" % csize)
    for n, L in [(1 << 22, 1 << 22), ((1 << 22) + 12345, 1 << 22), (5 * (3 << 17) - 7, 3 << 17), (1 << 21, (1 << 17) + 4096), (600000, 200000), (1 << 20, 1 << 17)]:
        f, truth = cases.delta_case(n, L, ud, cd, seed=n % 1000 + usize)
        scratch = torch.zeros(n // 4096 + 1024, dtype={2: torch.int16, 4: torch.int32, 8: torch.int64}[usize], device="cuda")
        out = g.dev_empty(n, ud)
        schemes.delta(out, g.dev(f["compressed_input"]), g.dev(f["anchor_values"]), n, L, uncompressed_size=usize, compressed_size=csize, index_size=index_size,
                      part_anchors_scratch=scratch, stream=g.stream())
        got = g.host(out, ud, n)
        bad = np.flatnonzero(got != truth)
        assert len(bad) == 0, (n, L, bad[:5], got[bad[:5]], truth[bad[:5]])

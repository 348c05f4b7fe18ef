"""-m gpu: the device-side compressors (libgiddy_amd.compressors -> C ABI -> HIP kernels under
include/giddy/kernels/compression/) against the numpy format writers (libgiddy_amd.formats): every output array equal BYTE FOR
BYTE, every size equal; then the round trip — what the library compressed, the library's decoders turn back into the column."""
import os

import numpy as np
import pytest

import cases
from libgiddy_amd import formats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

SIGNED = {1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def dev(a):
    """numpy array -> device tensor holding the same bits (unsigned dtypes as the signed ones of their width)"""
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(SIGNED[a.dtype.itemsize]) if a.dtype.kind == "u" else a).cuda()


def same(got, want, name):
    assert set(want) - {"data_to_scatter"} <= set(got), (name, set(want) - set(got))
    for key, expected in want.items():
        if isinstance(expected, np.ndarray):
            actual = got[key].cpu().numpy().reshape(-1)
            assert actual.dtype.itemsize == expected.dtype.itemsize, (name, key, actual.dtype, expected.dtype)
            actual = actual.view(expected.dtype)
            assert actual.shape == expected.reshape(-1).shape, (name, key, actual.shape, expected.shape)
            bad = np.flatnonzero(actual != expected.reshape(-1))
            assert len(bad) == 0, (name, key, bad[:5], actual[bad[:5]], expected.reshape(-1)[bad[:5]])
        else:
            assert int(got[key]) == int(expected), (name, key, got[key], expected)


SHAPES = [(1, 1), (1, 64), (100, 7), (4096, 4096), (4097, 1024), (70001, 1000), (300007, 4096), (1 << 20, 1 << 18)]


@pytest.mark.parametrize("usize,csize", [(2, 1), (4, 1), (4, 2), (8, 1), (8, 2), (8, 4)])
@pytest.mark.parametrize("index_size", [4, 8])
def test_delta(usize, csize, index_size):
    from libgiddy_amd import compressors, schemes
    ud = np.dtype("u%d" % usize)
    for n, L in SHAPES:
        rng = cases.rng(n + usize + csize)
        steps = rng.integers(0, 1 << (8 * csize), size=n, dtype=np.uint64)
        column = (np.cumsum(steps) + np.uint64(rng.integers(0, 1 << 20))).astype(ud)   # (wraps for narrow columns: deltas are modular)
        want = formats.delta(column, L, np.dtype("u%d" % csize))
        got = compressors.delta(dev(column), L, np.dtype("u%d" % csize), index_size=index_size)
        same(got, want, ("delta", n, L))
        out = torch.empty(n, dtype=got["anchor_values"].dtype, device="cuda")
        schemes.delta(out, got["compressed_input"], got["anchor_values"], n, L, uncompressed_size=usize, compressed_size=csize, index_size=index_size,
                      stream=torch.cuda.current_stream().cuda_stream)
        assert (out.cpu().numpy().view(ud) == column).all()
    with pytest.raises(compressors.Unrepresentable):
        compressors.delta(dev(np.array([0, 1 << (8 * csize)], dtype=ud)), 2, np.dtype("u%d" % csize), index_size=index_size)


@pytest.mark.parametrize("usize,csize", [(2, 1), (4, 2), (8, 1), (8, 4)])
def test_discard_zero_bytes_fixed(usize, csize):
    from libgiddy_amd import compressors
    for n in (1, 1000, 300007):
        column = cases.rng(n).integers(0, 1 << (8 * csize), size=n, dtype=np.uint64).astype("u%d" % usize)
        same(compressors.discard_zero_bytes_fixed(dev(column), np.dtype("u%d" % csize)), formats.discard_zero_bytes_fixed(column, np.dtype("u%d" % csize)), ("dzbf", n))
    with pytest.raises(compressors.Unrepresentable):
        compressors.discard_zero_bytes_fixed(dev(np.array([1 << (8 * csize)], dtype="u%d" % usize)), np.dtype("u%d" % csize))


@pytest.mark.parametrize("index_size,usize,container", [(4, 2, 4), (4, 4, 4), (4, 8, 4), (8, 2, 4), (8, 4, 4), (8, 8, 4), (4, 2, 8), (4, 4, 8), (4, 8, 8)])
@pytest.mark.parametrize("min_size,bits", [(1, 2), (1, 3), (0, 2), (2, 1), (0, 4)])
def test_discard_zero_bytes_variable(index_size, usize, container, min_size, bits):
    from libgiddy_amd import compressors, schemes
    if min_size > usize:
        pytest.skip("min size exceeds element width")
    ud = np.dtype("u%d" % usize)
    for n, L in [(1, 32), (1000, 64), (4096, 1024), (333333, 1024), (50, 7), (70000, 1000)]:
        column, _ = cases.dzbv_column(n, ud, min_size, bits, seed=n + bits + usize)
        want = formats.discard_zero_bytes_variable(column, L, min_size, bits, "u%d" % container, index_size)
        got = compressors.discard_zero_bytes_variable(dev(column), L, min_size, bits, np.dtype("u%d" % container), index_size)
        same(got, want, ("dzbv", n, L))
        out = torch.empty(n, dtype=torch.from_numpy(np.zeros(1, SIGNED[usize])).dtype, device="cuda")
        schemes.discard_zero_bytes_variable(out, got["compressed_data"], got["packed_element_sizes"], got["position_anchors"], L, n, min_size, bits,
                                            uncompressed_size=usize, container_size=container, index_size=index_size, stream=torch.cuda.current_stream().cuda_stream)
        assert (out.cpu().numpy().view(ud) == column).all()
    if min_size + (1 << bits) - 1 < usize:   # a value needing more bytes than a field can state
        with pytest.raises(compressors.Unrepresentable):
            compressors.discard_zero_bytes_variable(dev(np.array([np.iinfo(ud).max], dtype=ud)), 4, min_size, bits, np.dtype("u%d" % container), index_size)


@pytest.mark.parametrize("index_size,uncompressed,compressed", [(4, np.int16, np.int8), (4, np.int32, np.int8), (4, np.int32, np.int16), (8, np.int32, np.int16)])
def test_frame_of_reference_constant_model(index_size, uncompressed, compressed):
    from libgiddy_amd import compressors
    for n, L in SHAPES:
        rng = cases.rng(n + L)
        info = np.iinfo(compressed)
        base = rng.integers(-20000, 20000, size=-(-n // L)).repeat(L)[:n]
        column = (base + rng.integers(info.min // 2, info.max // 2, size=n)).astype(uncompressed)
        want = formats.frame_of_reference(column, L, compressed, "constant")
        got = compressors.frame_of_reference(dev(column), L, compressed, "constant", index_size=index_size)
        same(got, want, ("for", n, L))
    with pytest.raises(compressors.Unrepresentable):
        compressors.frame_of_reference(dev(np.array([np.iinfo(uncompressed).min // 2, np.iinfo(uncompressed).max // 2], dtype=uncompressed)), 2, compressed)


@pytest.mark.parametrize("index_size,uncompressed,compressed", [(4, np.int16, np.int8), (4, np.int32, np.int8), (4, np.int32, np.int16), (8, np.int32, np.int16)])
def test_frame_of_reference_linear_model(index_size, uncompressed, compressed):
    """a fitted line per segment: the exact least-squares slope (128-bit integer arithmetic on the device, Python integers in the writer), rounded half to even"""
    from libgiddy_amd import compressors, schemes
    for n, L in [(1, 4), (2, 4), (100, 8), (4096, 4096), (4097, 1024), (70001, 1000), (300007, 4096), (1 << 19, 1 << 18)]:
        rng = cases.rng(n + L + 1)
        info = np.iinfo(compressed)
        segments = -(-n // L)
        slopes = rng.integers(-3, 4, size=segments).repeat(L)[:n] if uncompressed == np.int16 else rng.integers(-40, 41, size=segments).repeat(L)[:n]
        base = rng.integers(-2000, 2000, size=segments).repeat(L)[:n]
        if uncompressed == np.int16 and L > 4096:
            continue   # (a line over 2^18 positions leaves int16)
        noise = rng.integers(info.min // 4, info.max // 4, size=n)
        column = (base + slopes * (np.arange(n) % L) + noise).astype(uncompressed)
        want = formats.frame_of_reference(column, L, compressed, "linear")
        got = compressors.frame_of_reference(dev(column), L, compressed, "linear", index_size=index_size)
        same(got, want, ("for linear", n, L))
        out = torch.empty(n, dtype=torch.from_numpy(np.zeros(1, uncompressed)).dtype, device="cuda")
        schemes.frame_of_reference(out, got["compressed_input"], got["segment_model_coefficients"], n, L, uncompressed_size=np.dtype(uncompressed).itemsize,
                                   compressed_size=np.dtype(compressed).itemsize, model="linear", index_size=index_size, model_index_size=4,
                                   stream=torch.cuda.current_stream().cuda_stream)
        assert (out.cpu().numpy() == column).all(), ("for linear round trip", n, L)


@pytest.mark.parametrize("usize,ksize", [(1, 1), (2, 1), (2, 2), (4, 1), (4, 2), (4, 4), (8, 1), (8, 2), (8, 4)])
def test_dictionary(usize, ksize):
    from libgiddy_amd import compressors
    ud = np.dtype("u%d" % usize)
    for n, distinct in [(1, 1), (1000, 7), (300007, min(256, 1 << (8 * min(usize, ksize)))), (1 << 20, min(40000, 1 << (8 * min(usize, ksize))))]:
        rng = cases.rng(n + distinct)
        pool = np.unique(rng.integers(0, np.iinfo(ud).max, size=distinct, dtype=np.uint64, endpoint=True).astype(ud))
        column = pool[rng.integers(0, len(pool), size=n)]
        want = formats.dictionary(column, np.dtype("u%d" % ksize))
        if ksize <= 2:   # the dictionary found and sorted on the device (hash set + rank sort: up to 65 536 entries)
            same(compressors.dictionary(dev(column), np.dtype("u%d" % ksize)), want, ("dictionary", n, distinct))
        same(compressors.dictionary(dev(column), np.dtype("u%d" % ksize), entries=dev(want["dictionary_entries"])), want, ("dictionary, given", n))
    if usize > 1:
        with pytest.raises(compressors.Unrepresentable):
            compressors.dictionary(dev(np.array([5, 6, 7], dtype=ud)), np.dtype("u%d" % ksize), entries=dev(np.array([5, 7], dtype=ud)))


RUN_COLUMNS = [(1, 1, 1), (100, 7, 3), (4096, 4096, 32), (70001, 1000, 2), (300007, 4096, 32), (300007, 1024, 1000), (1 << 20, 4096, 5), (200000, 512, 70000)]


@pytest.mark.parametrize("index_size,usize,lsize", [(4, 1, 1), (4, 1, 2), (4, 2, 2), (4, 4, 1), (4, 4, 2), (4, 4, 4), (4, 8, 2), (8, 4, 2), (8, 1, 8), (8, 8, 8)])
def test_run_length_encoding(index_size, usize, lsize):
    """... including runs longer than a length field holds, which the format cuts into pieces (formats._runs_of)"""
    from libgiddy_amd import compressors, schemes
    ud = np.dtype("u%d" % usize)
    for n, L, mean_run in RUN_COLUMNS:
        column = cases.runs_column(n, mean_run, ud, seed=n + mean_run)
        want = formats.run_length_encoding(column, L, np.dtype("u%d" % lsize), index_size)
        got = compressors.run_length_encoding(dev(column), L, np.dtype("u%d" % lsize), index_size)
        same(got, want, ("rle", n, L, mean_run))
        out = torch.empty(n, dtype=got["run_data"].dtype, device="cuda")
        schemes.run_length_encoding(out, got["run_data"], got["run_lengths"], got["position_anchors"], got["intra_run_anchor_offsets"], L, got["num_segments"],
                                    got["num_element_runs"], n, uncompressed_size=usize, run_length_size=lsize, index_size=index_size,
                                    stream=torch.cuda.current_stream().cuda_stream)
        assert (out.cpu().numpy().view(ud) == column).all(), ("rle round trip", n, L, mean_run)
    # runs far longer than any narrow length field holds, segments beginning inside their pieces
    column = np.repeat(np.array([5, 6, 5, 9, 200], dtype=ud), [70000, 1, 140001, 3, 66000])
    for L in (64, 4096, 100000):
        same(compressors.run_length_encoding(dev(column), L, np.dtype("u%d" % lsize), index_size), formats.run_length_encoding(column, L, np.dtype("u%d" % lsize), index_size),
             ("rle, long runs", L))


@pytest.mark.parametrize("index_size,usize,psize,relative", [(4, 1, 2, True), (4, 4, 2, True), (4, 8, 2, True), (8, 4, 2, True), (4, 2, 4, False), (4, 4, 4, False), (8, 4, 8, False)])
def test_run_position_encoding(index_size, usize, psize, relative):
    from libgiddy_amd import compressors, schemes
    ud = np.dtype("u%d" % usize)
    for n, L, mean_run in RUN_COLUMNS:
        if relative and L > (1 << (8 * psize)):
            continue
        column = cases.runs_column(n, mean_run, ud, seed=n + mean_run + 1)
        want = formats.run_position_encoding(column, L, np.dtype("u%d" % psize), relative, index_size)
        got = compressors.run_position_encoding(dev(column), L, np.dtype("u%d" % psize), relative, index_size)
        same(got, want, ("rpe", n, L, mean_run))
        out = torch.empty(n, dtype=got["run_data"].dtype, device="cuda")
        schemes.run_position_encoding(out, got["run_data"], got["run_start_positions"], got["position_anchors"], L, got["num_anchors"], got["num_element_runs"], n,
                                      uncompressed_size=usize, position_size=psize, relative=relative, index_size=index_size, stream=torch.cuda.current_stream().cuda_stream)
        assert (out.cpu().numpy().view(ud) == column).all(), ("rpe round trip", n, L, mean_run)


@pytest.mark.parametrize("usize", [1, 2, 4, 8])
def test_run_compressors_by_three_launches_and_by_the_chain_agree(usize):
    """The run compressors take three launches and no chain when they are given "run_scratch" (kernels/compression/listed_runs.hpp) and the one chained launch
    otherwise: both against the format writers — with a scratch that holds 16 runs per 4096 elements (nearly every wavefront of the dense columns is listed a
    second time), 64, and the most there is (512); columns of one run, of runs longer than a 2-byte length holds, with segments that begin inside later pieces."""
    from libgiddy_amd import compressors
    ud = np.dtype("u%d" % usize)
    columns = [(cases.runs_column(n, mean_run, ud, seed=5 * n + mean_run), L) for n, L, mean_run in RUN_COLUMNS + [(50000, 333, 2), (12289, 4096, 1)]]
    columns += [(np.repeat(np.array([5, 6, 5, 9, 200], dtype=ud), [70000, 1, 140001, 3, 66000]), L) for L in (64, 4096, 100000)]
    columns += [(np.full(300000, 7, dtype=ud), 1000), (np.arange(9000, dtype=np.uint64).astype(ud), 4096)]
    for column, L in columns:
        want_lengths = formats.run_length_encoding(column, L, np.uint16, 4)
        want_positions = formats.run_position_encoding(column, L, np.uint32, False, 4)
        want_relative = formats.run_position_encoding(column, L, np.uint16, True, 4) if L <= 65536 else None
        for staged, chain in ((16, None), (64, None), (512, None), (None, {"patience": 1 << 23})):
            what = ("three launches, %d staged" % staged) if chain is None else "chained"
            same(compressors.run_length_encoding(dev(column), L, np.uint16, 4, chain=chain, staged_per_wave=staged), want_lengths, ("rle", what, len(column), L))
            same(compressors.run_position_encoding(dev(column), L, np.uint32, False, 4, chain=chain, staged_per_wave=staged), want_positions, ("rpe absolute", what, len(column), L))
            if want_relative is not None:
                same(compressors.run_position_encoding(dev(column), L, np.uint16, True, 4, chain=chain, staged_per_wave=staged), want_relative, ("rpe relative", what, len(column), L))
        # ... and the column handed over one element past a 16-byte boundary (every wavefront reads element by element)
        displaced = dev(np.concatenate((column[:1], column)))[1:]
        same(compressors.run_length_encoding(displaced, L, np.uint16, 4), want_lengths, ("rle, displaced", len(column), L))
        same(compressors.run_position_encoding(displaced, L, np.uint32, False, 4, staged_per_wave=16), want_positions, ("rpe absolute, displaced", len(column), L))


def test_run_compressors_over_random_shapes():
    """the run compressors without a chain over 240 random shapes — lengths of 1 ... 300 000 elements, segment lengths of 1 ... 100 000, mean runs of 1 ... the whole column, every
    element size, length fields of 2 / 4 / 8 bytes, positions relative and absolute, 16 / 48 / 512 runs staged per wavefront — against the format writers"""
    from libgiddy_amd import compressors
    rng = cases.rng(20251003)
    segment_lengths = (1, 2, 3, 64, 100, 1000, 4096, 5000, 65536, 100000)
    for trial in range(240):
        n = int(rng.integers(1, 300001)) if trial % 4 else int(rng.integers(1, 9000))
        L = int(segment_lengths[rng.integers(0, len(segment_lengths))])
        if L < 64 and n > 40000:
            n = n % 40000 + 1   # (an anchor per element or two: keep the host writer quick)
        mean_run = int((1, 2, 5, 40, 3000, n)[rng.integers(0, 6)])
        usize = int((1, 2, 4, 8)[rng.integers(0, 4)])
        staged = int((16, 48, 512)[rng.integers(0, 3)])
        ud = np.dtype("u%d" % usize)
        column = cases.runs_column(n, mean_run, ud, seed=1000 + trial, max_run=1 << 40)
        what = (trial, n, L, mean_run, usize, staged)
        lsize = int((2, 4, 8)[rng.integers(0, 3)])
        index_size = 8 if lsize == 8 else 4
        from libgiddy_amd import registered as r
        if "compression::run_length_encoding::kernel_t<%du, %du, %du, %du>" % (index_size, usize, lsize, index_size) in r.kernel_t.registered_keys():
            same(compressors.run_length_encoding(dev(column), L, np.dtype("u%d" % lsize), index_size, staged_per_wave=staged),
                 formats.run_length_encoding(column, L, np.dtype("u%d" % lsize), index_size), ("rle", lsize) + what)
        relative = bool(rng.integers(0, 2))
        psize = 2 if relative and L <= 65536 else (4 if rng.integers(0, 2) else 8)
        relative = relative and L <= 65536
        index_size = 8 if psize == 8 else 4
        key = "compression::run_position_encoding::kernel_t<%du, %du, %du, %s, %du>" % (index_size, usize, psize, "true" if relative else "false", index_size)
        if key not in r.kernel_t.registered_keys():
            continue
        same(compressors.run_position_encoding(dev(column), L, np.dtype("u%d" % psize), relative, index_size, staged_per_wave=staged),
             formats.run_position_encoding(column, L, np.dtype("u%d" % psize), relative, index_size), ("rpe", relative, psize) + what)


def test_a_run_scratch_that_is_too_small_is_refused():
    from libgiddy_amd import compressors, registered as r, schemes
    n = 1 << 20
    column = dev(cases.runs_column(n, 32, np.dtype("u4"), seed=3))
    k = schemes.kernel("compression::run_position_encoding::kernel_t<4u, 4u, 2u, true, 4u>")
    out = {name: torch.empty(n, dtype=dtype, device="cuda") for name, dtype in (("run_data", torch.int32), ("run_start_positions", torch.int16), ("position_anchors", torch.int32))}
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    scratch = torch.empty(4 * 256 + 100, dtype=torch.int64, device="cuda")   # the per-wavefront tables fit; not even 16 runs per wavefront do
    with pytest.raises(Exception, match="run_scratch"):
        k.enqueue_launch(torch.cuda.current_stream().cuda_stream, k.resolve_launch_configuration(r.CURRENT_DEVICE, {"length": r.size_t(n), "segment_length": r.size_t(4096)}),
                         {"run_data": r.pointer(out["run_data"], "u32"), "run_start_positions": r.pointer(out["run_start_positions"], "u16"),
                          "position_anchors": r.pointer(out["position_anchors"], "u32"), "input": r.const_pointer(column, "u32"), "length": r.uint_t(4)(n),
                          "segment_length": r.uint_t(4)(4096), "num_element_runs": r.pointer(count, "u64"), "capacity": r.size_t(n),
                          "run_scratch": r.pointer(scratch, "u64"), "run_scratch_entries": r.size_t(scratch.numel())})
    torch.cuda.synchronize()


@pytest.mark.parametrize("index_size,esize", [(4, 1), (4, 4), (4, 8), (8, 2), (8, 8)])
def test_patches(index_size, esize):
    from libgiddy_amd import compressors
    ud = np.dtype("u%d" % esize)
    for n, differing in [(1, 0), (1, 1), (5000, 50), (300007, 301), (1 << 20, 1 << 19)]:
        rng = cases.rng(n + differing)
        truth = rng.integers(0, np.iinfo(ud).max, size=n, dtype=np.uint64, endpoint=True).astype(ud)
        decoded = truth.copy()
        where = rng.choice(n, size=differing, replace=False)
        decoded[where] = decoded[where] + ud.type(1)
        same(compressors.patches(dev(decoded), dev(truth), index_size), formats.patches(decoded, truth, index_size), ("patches", n, differing))


@pytest.mark.parametrize("usize", [1, 2, 4, 8])
def test_dictionary_construction(usize):
    """the column's distinct values, sorted, found on the device: numpy.unique"""
    from libgiddy_amd import compressors
    ud = np.dtype("u%d" % usize)
    for n, distinct in [(1, 1), (10, 3), (100000, 256), (1 << 20, 1000), (1 << 21, 65536), (300000, 40000)]:
        distinct = min(distinct, 1 << (8 * usize))
        rng = cases.rng(n + distinct + usize)
        pool = np.unique(rng.integers(0, np.iinfo(ud).max, size=distinct, dtype=np.uint64, endpoint=True).astype(ud))
        column = pool[rng.integers(0, len(pool), size=n)]
        got = compressors.dictionary_entries(dev(column)).cpu().numpy().view(ud)
        assert np.array_equal(got, np.unique(column)), (usize, n, distinct, len(got))
    if usize >= 4:
        with pytest.raises(compressors.Unrepresentable):
            compressors.dictionary_entries(dev(np.arange(70000, dtype=ud)))
        with pytest.raises(compressors.Unrepresentable):
            compressors.dictionary_entries(dev(np.arange(300, dtype=ud)), limit=256)
        # more than 65 536 entries (4- and 8-byte keys; round 5): a hash set sized by the limit, the entries sorted by a bitonic network — limits that are and are not
        # powers of two, fewer entries than a workgroup sorts in LDS, the largest value among the entries (what the network pads with), every element distinct
        for n, distinct, limit in [(1 << 21, 300000, 1 << 19), (1 << 20, 70000, 100000), (500000, 500000, 500000), (1 << 20, 1000, 1 << 17), (3000000, 2000000, 3000000)]:
            rng = cases.rng(n + distinct)
            pool = np.unique(rng.integers(0, np.iinfo(ud).max, size=distinct, dtype=np.uint64, endpoint=True).astype(ud))
            pool[-1] = np.iinfo(ud).max
            column = pool[rng.integers(0, len(pool), size=n)] if distinct < n else rng.permutation(pool)[:n]
            got = compressors.dictionary_entries(dev(column), limit=limit).cpu().numpy().view(ud)
            assert np.array_equal(got, np.unique(column)), (usize, n, distinct, limit, len(got))
        with pytest.raises(compressors.Unrepresentable):
            compressors.dictionary_entries(dev(np.arange(200000, dtype=ud)), limit=100000)


@pytest.mark.parametrize("usize,ksize", [(4, 4), (8, 4), (8, 8)])
def test_dictionary_with_wide_keys_built_and_encoded_on_the_device(usize, ksize):
    """compressors.dictionary with 4-byte keys and no dictionary given: construction (more than 65 536 entries), encoding, and the decode back"""
    from libgiddy_amd import compressors, schemes
    ud = np.dtype("u%d" % usize)
    rng = cases.rng(usize)
    pool = np.unique(rng.integers(0, np.iinfo(ud).max, size=150000, dtype=np.uint64, endpoint=True).astype(ud))
    column = pool[rng.integers(0, len(pool), size=1 << 21)]
    want = formats.dictionary(column, "u%d" % ksize)
    got = compressors.dictionary(dev(column), "u%d" % ksize)
    same(got, want, ("dictionary with wide keys", usize, ksize))
    out = torch.empty(len(column), dtype=got["dictionary_entries"].dtype, device="cuda")
    schemes.dictionary(out, got["compressed_input"], got["dictionary_entries"], len(column), got["num_dictionary_entries"], uncompressed_size=usize, dictionary_index_size=ksize,
                       stream=torch.cuda.current_stream().cuda_stream)
    assert (out.cpu().numpy().view(ud) == column).all()


@pytest.mark.parametrize("usize,use_dictionary", [(1, True), (2, True), (4, True), (8, True), (1, False)])
def test_incidence_bitmaps(usize, use_dictionary):
    from libgiddy_amd import compressors, schemes
    ud = np.dtype("u%d" % usize)
    for words, distinct in [(1, 1), (3, 2), (64, 16), (1001, 11), (32768, 256)]:
        rng = cases.rng(words + distinct)
        pool = np.unique(rng.integers(0, np.iinfo(ud).max, size=distinct, dtype=np.uint64, endpoint=True).astype(ud)) if use_dictionary else np.arange(distinct, dtype=ud)
        column = pool[rng.integers(0, len(pool), size=32 * words)]
        if not use_dictionary:
            column[-1] = distinct - 1   # (the writer sizes the bitmaps by the largest value)
        want = formats.incidence_bitmaps(column, 2, use_dictionary)
        got = compressors.incidence_bitmaps(dev(column), 2, use_dictionary)
        same(got, want, ("incidence bitmaps", words, distinct))
        out = torch.empty(32 * words, dtype=got["dictionary_entries"].dtype, device="cuda")
        schemes.incidence_bitmaps(out, got["incidence_bitmaps"], got["dictionary_entries"], words, got["num_bitmaps"], uncompressed_size=usize, use_dictionary=use_dictionary,
                                  stream=torch.cuda.current_stream().cuda_stream)
        assert (out.cpu().numpy().view(ud) == column).all(), ("incidence bitmaps round trip", words, distinct)


def test_a_column_goes_through_model_plus_patches_without_leaving_the_device():
    """what the pieces are for: a nearly linear column -> Model decode of its fit -> patches found on the device -> fused Model + patches == the column"""
    from libgiddy_amd import compressors, schemes
    n = 1 << 20
    rng = cases.rng(5)
    column = (np.arange(n, dtype=np.int64) * 37 - 12345).astype(np.int32)
    odd = rng.choice(n, size=1000, replace=False)
    column[odd] += rng.integers(1, 100, size=1000).astype(np.int32)
    truth = dev(column)
    coefficients = np.array([-12345, 37], dtype=np.int32)
    stream = torch.cuda.current_stream().cuda_stream
    modelled = torch.empty(n, dtype=torch.int32, device="cuda")
    schemes.model(modelled, coefficients, n, uncompressed_size=4, model="linear", stream=stream)
    p = compressors.patches(modelled, truth)
    assert p["data_length"] == 1000 and (p["indices"].cpu().numpy() == np.sort(odd)).all()
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    schemes.patched_model(out, coefficients, p["data"], p["indices"], n, p["data_length"], uncompressed_size=4, model="linear", stream=stream)
    assert bool((out == truth).all())


@pytest.mark.parametrize("output_index_size,input_index_size", [(4, 4), (8, 8), (8, 4), (4, 8), (4, 2), (4, 1)])
def test_compressed_patch_positions(output_index_size, input_index_size):
    """the compressed-indices arrays written on the device == formats.compressed_indices, and a fused decoder takes them as they are"""
    from libgiddy_amd import compressors, schemes
    limit = (1 << (8 * input_index_size)) - 1 if input_index_size < 4 else 1 << 40
    for n, period, span in [(1, 16, 100), (300, 64, 100000), (5000, 256, 1 << 24), (20000, 1024, 1 << 31), (100000, 1000, 1 << 20), (3000, 7, 1 << 33)]:
        if n > limit or (span > (1 << 32) and output_index_size == 4):
            continue
        rng = cases.rng(n + period)
        positions = np.unique(rng.integers(0, span, size=n * 2, dtype=np.uint64))[:n]
        n_here = len(positions)
        want = formats.compressed_indices(positions, np.zeros(0, np.uint8), period, output_index_size, input_index_size)
        want.pop("data_to_scatter")
        got = compressors.compressed_indices(dev(positions.astype("u%d" % output_index_size)), period, output_index_size, input_index_size)
        same(got, want, ("compressed indices", n_here, period, span))
    # ... straight into the fused Delta decoder
    if (output_index_size, input_index_size) == (4, 4):
        n, L = 1 << 18, 4096
        f, truth = cases.delta_case(n, L, np.uint64, np.uint16, seed=9)
        where = np.sort(cases.rng(10).choice(n, size=777, replace=False)).astype(np.uint32)
        values = cases.rng(11).integers(0, 1 << 62, size=777, dtype=np.uint64)
        compressed = compressors.compressed_indices(dev(where), 128)
        compressed["patch_positions_scratch"] = torch.empty(777, dtype=torch.int32, device="cuda")
        out = torch.empty(n, dtype=torch.int64, device="cuda")
        schemes.patched_delta(out, dev(f["compressed_input"]), dev(f["anchor_values"]), dev(values), compressed, n, L, 777, stream=torch.cuda.current_stream().cuda_stream)
        expect = truth.copy(); expect[where] = values
        assert (out.cpu().numpy().view(np.uint64) == expect).all()


def test_capacity_is_told_to_the_kernels():
    """a `capacity` below the true number of runs / patches: nothing is stored beyond it (the arrays sit inside one allocation, with canaries
    behind them), the true number comes back, and a second call with that much room gives the format writers' arrays"""
    from libgiddy_amd import compressors
    n, L = 300007, 4096
    column = cases.runs_column(n, 8, np.uint32, seed=77)
    long_runs = np.repeat(np.array([5, 6, 5, 9, 200], dtype=np.uint32), [70000, 1, 140001, 3, 66000])   # pieces far outnumber the natural runs (1-byte lengths)
    for col, lsize in ((column, 2), (long_runs, 1)):
        want = formats.run_length_encoding(col, L, np.dtype("u%d" % lsize), 4)
        with pytest.raises(compressors.CapacityExceeded) as caught:
            compressors.run_length_encoding(dev(col), L, np.dtype("u%d" % lsize), 4, capacity=want["num_element_runs"] // 3)
        assert caught.value.needed == want["num_element_runs"]
        same(compressors.run_length_encoding(dev(col), L, np.dtype("u%d" % lsize), 4, capacity=caught.value.needed), want, "rle at exact capacity")
    want = formats.run_position_encoding(column, L, np.uint16, True)
    with pytest.raises(compressors.CapacityExceeded) as caught:
        compressors.run_position_encoding(dev(column), L, np.uint16, True, capacity=100)
    assert caught.value.needed == want["num_element_runs"]
    same(compressors.run_position_encoding(dev(column), L, np.uint16, True, capacity=caught.value.needed), want, "rpe at exact capacity")
    truth = cases.rng(5).integers(0, 1 << 30, size=n, dtype=np.uint64).astype(np.uint32)
    decoded = truth.copy(); where = np.sort(cases.rng(6).choice(n, size=5000, replace=False)); decoded[where] ^= 1
    with pytest.raises(compressors.CapacityExceeded) as caught:
        compressors.patches(dev(decoded), dev(truth), capacity=1000)
    assert caught.value.needed == 5000
    got = compressors.patches(dev(decoded), dev(truth), capacity=5000)
    assert (got["indices"].cpu().numpy().view(np.uint32) == where).all() and (got["data"].cpu().numpy().view(np.uint32) == truth[where]).all()
    # canaries: the kernels given arrays that END where the capacity ends, inside a larger buffer
    from libgiddy_amd import registered as r, schemes
    room = 1000
    arena = torch.full((4 * room,), -1, dtype=torch.int32, device="cuda")
    positions, values = arena[:room], arena[2 * room: 3 * room]
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    scratch = torch.empty(2 * (-(-n // 4096) + 2) + 2, dtype=torch.int64, device="cuda")
    k = schemes.kernel("compression::patches::kernel_t<4u, 4u>")
    k.enqueue_launch(torch.cuda.current_stream().cuda_stream, k.resolve_launch_configuration(r.CURRENT_DEVICE, {"length": r.size_t(n)}),
                     {"patch_positions": r.pointer(positions, "u32"), "patch_values": r.pointer(values, "u32"), "decoded": r.const_pointer(dev(decoded), "u32"),
                      "truth": r.const_pointer(dev(truth), "u32"), "length": r.uint_t(4)(n), "tile_totals": r.pointer(scratch, "u64"),
                      "num_patches": r.pointer(count, "u64"), "capacity": r.size_t(room)})
    torch.cuda.synchronize()
    assert int(count[0]) == 5000 and bool((arena[room: 2 * room] == -1).all()) and bool((arena[3 * room:] == -1).all())
    assert (positions.cpu().numpy().view(np.uint32) == where[:room]).all()


@pytest.mark.parametrize("scheme", ["run_length_encoding", "run_position_encoding"])
def test_a_broken_chain_of_tiles_is_reported_not_waited_for(scheme):
    """The one-launch compressors' tiles wait for the tiles before them (kernels/compression/chained.hpp).  Every wait is bounded: here one tile
    plays dead ("chain_fault_tile"), the tiles behind it run out of a SHORT patience, give up one after the other, and the count the kernel leaves
    is 2^64 - 1 — ChainBroken in Python.  Run in a child process under a watchdog: a regression that waits for ever must fail the test, not hang
    the GPU box (as tests/test_gpu_dictionary_barrier_guard.py does for the dictionary's barrier)."""
    import subprocess
    import sys
    program = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np, torch
from libgiddy_amd import compressors
n = 1 << 22
column = torch.from_numpy(np.repeat(np.arange(n // 8, dtype=np.uint32), 8).view(np.int32)).cuda()
compress = getattr(compressors, %r)
ok = compress(column, 4096)                                     # the chain as it is: fine
assert ok["num_element_runs"] == n // 8
for fault_tile in (0, 3, 40):
    try:
        compress(column, 4096, chain={"patience": 4000, "fault_tile": fault_tile})
    except compressors.ChainBroken:
        continue
    raise SystemExit("tile %%d played dead and nobody noticed" %% fault_tile)
again = compress(column, 4096)                                  # and the next launch is none the worse
assert again["num_element_runs"] == n // 8 and bool((again["run_data"] == ok["run_data"]).all())
print("chain OK")
''' % (ROOT, scheme)
    result = subprocess.run([sys.executable, "-c", program], capture_output=True, text=True, timeout=300)
    assert result.returncode == 0 and "chain OK" in result.stdout, result.stdout[-2000:] + result.stderr[-2000:]

"""CPU tests: the C oracle against (a) the reference's only known-answer test, (b) the independent
numpy statement, (c) round trips through the format writers — including the edge shapes the
reference documents (ragged tails, single-run segments, runs ending on segment boundaries,
size-container slack bits, 1-element columns)."""
import numpy as np
import pytest

import cases
from libgiddy_amd import formats
from oracle import naive, oracle


def test_kat_for_1():
    # examples/src/modes_of_use/common.cuh:31-49 (inputs), :91-102 (expected values)
    length, seglen = 2 << 12, 2 << 10
    nseg = length // seglen
    comp = np.tile(np.arange(seglen, dtype=np.int16), nseg)
    coefs = (np.arange(nseg, dtype=np.int32) * 100).reshape(nseg, 1)
    out = oracle.frame_of_reference(comp, coefs, length, seglen, np.int32, "constant")
    for s in range(nseg):
        assert (out[s * seglen:(s + 1) * seglen] == 100 * s + np.arange(seglen)).all()


@pytest.mark.parametrize("out_dtype,comp_dtype", [(np.int16, np.int8), (np.int32, np.int8), (np.int32, np.int16)])
@pytest.mark.parametrize("model", ["constant", "linear"])
@pytest.mark.parametrize("n,seglen", [(1, 4), (1000, 64), (4096, 4096), (5000, 1024), (777, 12)])
def test_for_oracle_vs_naive(out_dtype, comp_dtype, model, n, seglen):
    f = cases.for_case(n, seglen, out_dtype, comp_dtype, model, seed=n)
    a = oracle.frame_of_reference(f["compressed_input"], f["segment_model_coefficients"], n, seglen, out_dtype, model)
    b = naive.frame_of_reference(f["compressed_input"], f["segment_model_coefficients"], n, seglen, out_dtype, model)
    assert (a == b).all()
    a2 = oracle.frame_of_reference(f["compressed_input"], f["segment_model_coefficients"], n, seglen, out_dtype, model, threads=4)
    assert (a == a2).all()


def test_for_roundtrip():
    r = cases.rng(11)
    col = (r.integers(-20000, 20000, size=3000) + np.repeat(r.integers(-10**6, 10**6, size=3), 1000)).astype(np.int32)
    for model in ("constant", "linear"):
        f = formats.frame_of_reference(col, 1000, np.int16, model)
        out = oracle.frame_of_reference(f["compressed_input"], f["segment_model_coefficients"], 3000, 1000, np.int32, model)
        assert (out == col).all()


@pytest.mark.parametrize("out_dtype,model", [(np.int16, "linear"), (np.int32, "linear"), (np.int16, "constant"), (np.int32, "constant")])
def test_model_oracle_vs_naive(out_dtype, model):
    coefs = np.array([12345, -7] if model == "linear" else [-321], dtype=out_dtype)
    a = oracle.model(coefs, 70000, out_dtype, model)
    b = naive.model(coefs, 70000, out_dtype, model)
    assert (a == b).all()


@pytest.mark.parametrize("value_dtype,key_dtype,entries", [
    (np.uint8, np.uint8, 200), (np.uint16, np.uint8, 256), (np.uint16, np.uint16, 5000),
    (np.uint32, np.uint8, 256), (np.uint32, np.uint16, 65536), (np.uint32, np.uint32, 100000),
    (np.uint64, np.uint8, 7), (np.uint64, np.uint16, 1000), (np.uint64, np.uint32, 70000), (np.uint64, np.uint64, 10)])
def test_dictionary_oracle_vs_naive(value_dtype, key_dtype, entries):
    for n in (0, 1, 3, 1023, 20000):
        f = cases.dictionary_case(n, entries, value_dtype, key_dtype, seed=n + 1)
        a = oracle.dictionary(f["compressed_input"], f["dictionary_entries"])
        assert (a == naive.dictionary(f["compressed_input"], f["dictionary_entries"])).all()


def test_dictionary_roundtrip():
    col = cases.rng(5).integers(0, 100, size=5000).astype(np.uint32) * 1000003
    f = formats.dictionary(col)
    assert (oracle.dictionary(f["compressed_input"], f["dictionary_entries"]) == col).all()


@pytest.mark.parametrize("out_dtype,comp_dtype", [(np.uint16, np.uint8), (np.uint32, np.uint8), (np.uint32, np.uint16),
                                                  (np.uint64, np.uint8), (np.uint64, np.uint16), (np.uint64, np.uint32)])
@pytest.mark.parametrize("n,seglen", [(1, 1), (63, 7), (4096, 4096), (10000, 4096), (1000, 1), (5000, 128), (100, 1000)])
def test_delta_oracle_vs_naive(out_dtype, comp_dtype, n, seglen):
    f, truth = cases.delta_case(n, seglen, out_dtype, comp_dtype, seed=n + seglen)
    a = oracle.delta(f["compressed_input"], f["anchor_values"], seglen)
    assert (a == truth).all()
    assert (a == naive.delta(f["compressed_input"], f["anchor_values"], seglen)).all()
    assert (a == oracle.delta(f["compressed_input"], f["anchor_values"], seglen, threads=3)).all()


def test_delta_roundtrip_and_patches():
    r = cases.rng(9)
    col = np.cumsum(r.integers(0, 300, size=9000)).astype(np.uint64) + np.uint64(1 << 35)
    truth = col.copy()
    outliers = np.sort(r.choice(9000, 9, replace=False))
    # patched column: a few wild values the delta stream cannot express
    truth[outliers] = r.integers(0, 1 << 62, size=9, dtype=np.uint64)
    f = formats.delta(col, 512, np.uint16)
    decoded = oracle.delta(f["compressed_input"], f["anchor_values"], 512)
    assert (decoded == col).all()
    p = formats.patches(decoded, truth)
    patched = oracle.scatter(decoded.copy(), p["data"], p["indices"])
    assert (patched == truth).all()
    assert (naive.scatter(decoded, p["data"], p["indices"]) == truth).all()


@pytest.mark.parametrize("out_dtype,comp_dtype", [(np.uint16, np.uint8), (np.uint32, np.uint8), (np.uint32, np.uint16),
                                                  (np.uint64, np.uint8), (np.uint64, np.uint16), (np.uint64, np.uint32)])
def test_dzb_fixed(out_dtype, comp_dtype):
    c = cases.rng(3).integers(0, np.iinfo(comp_dtype).max, size=1237, endpoint=True).astype(comp_dtype)
    a = oracle.dzb_fixed(c, out_dtype)
    assert (a == naive.dzb_fixed(c, out_dtype)).all() and a.dtype == np.dtype(out_dtype)


@pytest.mark.parametrize("out_dtype", [np.uint16, np.uint32, np.uint64])
@pytest.mark.parametrize("min_size,bits,container", [(1, 2, np.uint32), (1, 3, np.uint32), (0, 2, np.uint32),
                                                     (1, 3, np.uint64), (2, 1, np.uint32), (1, 2, np.uint64)])
@pytest.mark.parametrize("n,seglen", [(1, 32), (1000, 64), (1024, 1024), (3333, 1024), (50, 7)])
def test_dzb_variable(out_dtype, min_size, bits, container, n, seglen):
    usize = np.dtype(out_dtype).itemsize
    if min_size > usize:
        pytest.skip("min size exceeds element width")
    col, sizes = cases.dzbv_column(n, out_dtype, min_size, bits, seed=n + bits)
    for index_size in (4, 8):
        f = formats.discard_zero_bytes_variable(col, seglen, min_size, bits, container, index_size, element_sizes=sizes)
        a = oracle.dzb_variable(f["compressed_data"], f["packed_element_sizes"], f["position_anchors"], seglen, n, min_size, bits, out_dtype)
        assert (a == col).all()
        b = naive.dzb_variable(f["compressed_data"], f["packed_element_sizes"], f["position_anchors"], seglen, n, min_size, bits, out_dtype)
        assert (a == b).all()
    # writer computing the sizes itself
    f = formats.discard_zero_bytes_variable(col, seglen, max(min_size, 1), 3, container, 8)
    a = oracle.dzb_variable(f["compressed_data"], f["packed_element_sizes"], f["position_anchors"], seglen, n, max(min_size, 1), 3, out_dtype, threads=2)
    assert (a == col).all()


@pytest.mark.parametrize("out_dtype", [np.uint8, np.uint16, np.uint32, np.uint64])
@pytest.mark.parametrize("n_words,num_values", [(1, 1), (5, 3), (64, 17), (1001, 256)])
def test_incidence_bitmaps(out_dtype, n_words, num_values):
    r = cases.rng(n_words)
    if num_values > np.iinfo(out_dtype).max + 1:
        pytest.skip("not enough distinct values")
    pool = r.choice(min(int(np.iinfo(out_dtype).max) + 1, 1 << 40), size=num_values, replace=False).astype(out_dtype)
    col = pool[r.integers(0, num_values, size=32 * n_words)]
    f = formats.incidence_bitmaps(col, 2)
    a, uncovered = oracle.incidence_bitmaps(f["incidence_bitmaps"], f["dictionary_entries"], f["bitmap_length"], f["num_bitmaps"], 2, out_dtype)
    assert uncovered == 0 and (a == col).all()
    b = naive.incidence_bitmaps(f["incidence_bitmaps"], f["dictionary_entries"], f["bitmap_length"], f["num_bitmaps"], 2, out_dtype)
    assert (a == b).all()
    if out_dtype == np.uint8:
        small = r.integers(0, 9, size=32 * n_words).astype(np.uint8)
        f = formats.incidence_bitmaps(small, 2, use_dictionary=False)
        a, _ = oracle.incidence_bitmaps(f["incidence_bitmaps"], f["dictionary_entries"], f["bitmap_length"], f["num_bitmaps"], 2, np.uint8, use_dictionary=False)
        assert (a == small).all()


def _run_columns():
    yield "geom32", cases.runs_column(20000, 32, np.uint32, seed=6)
    yield "geom2", cases.runs_column(5000, 2, np.uint32, seed=7)
    yield "single", np.full(5000, 77, dtype=np.uint32)
    yield "one", np.array([5], dtype=np.uint32)
    yield "alldistinct", np.arange(3000, dtype=np.uint32)
    # runs ending exactly on segment boundaries (segment 1024)
    yield "boundary", np.repeat(np.arange(10, dtype=np.uint32) * 3 + 1, [1024, 512, 512, 2048, 1, 1023, 1024, 100, 924, 1000])


@pytest.mark.parametrize("value_dtype", [np.uint8, np.uint16, np.uint32, np.uint64])
@pytest.mark.parametrize("len_dtype", [np.uint8, np.uint16, np.uint32])
def test_rle(value_dtype, len_dtype):
    for name, col in _run_columns():
        col = col.astype(np.uint64).astype(value_dtype)  # wraps
        n = len(col)
        for seglen in (1024, 4096, 100):
            f = formats.run_length_encoding(col, seglen, len_dtype)
            a = oracle.run_length_encoding(f["run_data"], f["run_lengths"], f["position_anchors"], f["intra_run_anchor_offsets"], seglen, n)
            assert (a == col).all(), (name, seglen)
            assert (naive.run_length_encoding(f["run_data"], f["run_lengths"], n) == col).all()


@pytest.mark.parametrize("value_dtype", [np.uint8, np.uint32, np.uint64])
@pytest.mark.parametrize("relative,pos_dtype", [(True, np.uint16), (False, np.uint32), (False, np.uint64)])
def test_rpe(value_dtype, relative, pos_dtype):
    for name, col in _run_columns():
        col = col.astype(np.uint64).astype(value_dtype)  # wraps
        n = len(col)
        for seglen in (1024, 4096, 100, 65536):
            f = formats.run_position_encoding(col, seglen, pos_dtype, relative)
            a = oracle.run_position_encoding(f["run_data"], f["run_start_positions"], f["position_anchors"], seglen, n, relative)
            assert (a == col).all(), (name, seglen)
            b = naive.run_position_encoding(f["run_data"], f["run_start_positions"], f["position_anchors"], seglen, n, relative)
            assert (b == col).all(), (name, seglen)


@pytest.mark.parametrize("esize,osize,isize", [(1, 4, 4), (4, 4, 2), (8, 4, 4), (4, 8, 4), (8, 8, 8), (2, 4, 1)])
def test_scatter_and_compressed_indices(esize, osize, isize):
    r = cases.rng(esize * 100 + osize)
    n = 200000
    npatch = 200 if isize == 1 else 3000
    target = r.integers(0, 255, size=n).astype(formats.uint_t(esize))
    p = cases.patch_case(n, npatch, formats.uint_t(esize), osize, seed=esize + osize)
    expect = naive.scatter(target, p["data"], p["indices"])
    assert (oracle.scatter(target.copy(), p["data"], p["indices"]) == expect).all()
    period = 64 if isize == 1 else 1024
    for maxrun in (None, 5):
        c = formats.compressed_indices(p["indices"], p["data"], period, osize, isize, max_run_length=maxrun)
        got = oracle.compressed_indices_scatter(
            target.copy(), c["data_to_scatter"], c["scatter_position_run_lengths"],
            c["scatter_position_run_individual_offset_sizes"], c["scatter_position_run_baseline_values"],
            c["scatter_position_run_offsets_start_pos"], c["scatter_position_offset_bytes"],
            c["scatter_position_anchors"], period)
        assert (got == expect).all()
        got2 = naive.compressed_indices_scatter(
            target, c["data_to_scatter"], c["scatter_position_run_lengths"],
            c["scatter_position_run_individual_offset_sizes"], c["scatter_position_run_baseline_values"],
            c["scatter_position_run_offsets_start_pos"], c["scatter_position_offset_bytes"],
            c["scatter_position_anchors"], period, osize)
        assert (got2 == expect).all()
        # ... and the positions alone (what the fused decoders merge from) are the original ascending indices
        positions = naive.compressed_indices_positions(
            len(p["indices"]), c["scatter_position_run_lengths"], c["scatter_position_run_individual_offset_sizes"],
            c["scatter_position_run_baseline_values"], c["scatter_position_run_offsets_start_pos"],
            c["scatter_position_offset_bytes"], c["scatter_position_anchors"], period, osize)
        assert (positions == p["indices"]).all()


def test_config_c1_frame_of_reference_single_segment_1m():
    """BASELINE.json configs[0]: FOR, int32 <- int16, ONE segment of 2^20 elements, reference value 1 000 000,
    offsets uniform over the int16 range (seed 1) — the CPU plumbing case: C oracle (1 and 8 threads) == numpy."""
    n = 1 << 20
    r = cases.rng(1)
    comp = r.integers(-32768, 32767, size=n, endpoint=True).astype(np.int16)
    coefs = np.array([[1000000]], dtype=np.int32)
    a = oracle.frame_of_reference(comp, coefs, n, n, np.int32, "constant")
    assert (a == (1000000 + comp.astype(np.int64)).astype(np.int32)).all()
    assert (a == oracle.frame_of_reference(comp, coefs, n, n, np.int32, "constant", threads=8)).all()
    assert (a[:5000] == naive.frame_of_reference(comp[:5000], coefs, 5000, n, np.int32, "constant")).all()


def test_scan_oracle_against_plain_loops():
    """oracle/scan.py (numpy) == element-by-element Python loops with explicit wrap-around, for every operator used"""
    from oracle import scan as s
    r = np.random.default_rng(0)
    cases = [("plus", np.uint16, np.uint16, "identity"), ("plus", np.uint16, np.uint64, "identity"), ("maximum", np.int32, np.int32, "identity"),
             ("minimum", np.int64, np.int64, "identity"), ("bit_or", np.uint32, np.uint32, "identity"), ("plus", np.int32, np.int64, "absolute")]
    for op, in_dtype, out_dtype, pre in cases:
        info = np.iinfo(in_dtype)
        data = r.integers(info.min, info.max, size=300, dtype=in_dtype, endpoint=True)
        bits = 8 * np.dtype(out_dtype).itemsize
        apply = {"plus": lambda a, b: a + b, "maximum": max, "minimum": min, "bit_or": lambda a, b: a | b}[op]

        def wrap(v):
            v &= (1 << bits) - 1
            return v - (1 << bits) if np.dtype(out_dtype).kind == "i" and v >> (bits - 1) else v
        acc, inclusive, exclusive = int(s.neutral_value(op, out_dtype)), [], []
        for x in data.tolist():
            x = abs(x) if pre == "absolute" else x
            exclusive.append(acc)
            acc = wrap(apply(acc, x))
            inclusive.append(acc)
        assert s.scan(data, op, out_dtype, True, pre).tolist() == inclusive
        assert s.scan(data, op, out_dtype, False, pre).tolist() == exclusive
        segs = s.reduce_segments(data, 64, op, out_dtype, pre).tolist()
        for i, lo in enumerate(range(0, 300, 64)):
            acc = int(s.neutral_value(op, out_dtype))
            for x in data[lo:lo + 64].tolist():
                acc = wrap(apply(acc, abs(x) if pre == "absolute" else x))
            assert segs[i] == acc
    assert s.scan(np.zeros(0, np.uint32), "plus", np.uint32).shape == (0,)

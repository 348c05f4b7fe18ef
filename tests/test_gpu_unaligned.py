"""-m gpu: base pointers that are element-aligned but NOT 16-byte aligned (a column handed over at an odd
element offset): every decoder must fall back to its element-wise paths and still be bit-exact, without
touching the element before the output or anything past its end (guard bands)."""
import numpy as np
import pytest

import cases
from libgiddy_amd import formats

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _odd(g, array):
    """device copy of `array` placed one element past a 16-byte boundary; returns (keepalive, address)"""
    array = np.ascontiguousarray(array)
    item = array.dtype.itemsize
    buf = g.dev(np.concatenate([np.zeros(1, array.dtype), array.reshape(-1)]))
    return buf, buf.data_ptr() + item


def _odd_out(g, n, dtype):
    item = np.dtype(dtype).itemsize
    buf = g.dev_empty(n + 1, dtype)
    return buf, buf.data_ptr() + item


def _read(g, buf, dtype, n):
    full = g.host(buf, dtype, n + 1)
    assert full[0] == np.frombuffer(bytes([0xCD] * np.dtype(dtype).itemsize), dtype=dtype)[0], "wrote before the output"
    return full[1:]


def test_unaligned_bases():
    import gpu_util as g
    from libgiddy_amd import schemes
    st = g.stream()
    n = 70001

    f, truth = cases.delta_case(n, 4096, np.uint64, np.uint16, seed=1)
    keep_in, in_ptr = _odd(g, f["compressed_input"])
    out, out_ptr = _odd_out(g, n, np.uint64)
    schemes.delta(out_ptr, in_ptr, g.dev(f["anchor_values"]), n, 4096, stream=st)
    assert (_read(g, out, np.uint64, n) == truth).all()

    f = cases.dictionary_case(n, 256, np.uint32, np.uint8, seed=2)
    keep_in, in_ptr = _odd(g, f["compressed_input"])
    out, out_ptr = _odd_out(g, n, np.uint32)
    schemes.dictionary(out_ptr, in_ptr, g.dev(f["dictionary_entries"]), n, 256, stream=st)
    assert (_read(g, out, np.uint32, n) == f["dictionary_entries"][f["compressed_input"].astype(np.int64)]).all()

    col = cases.runs_column(n, 32, np.uint32, seed=3)
    f = formats.run_length_encoding(col, 4096, np.uint16)
    out, out_ptr = _odd_out(g, n, np.uint32)
    keep_l, len_ptr = _odd(g, f["run_lengths"])
    keep_d, data_ptr = _odd(g, f["run_data"])
    schemes.run_length_encoding(out_ptr, data_ptr, len_ptr, g.dev(f["position_anchors"]), g.dev(f["intra_run_anchor_offsets"]), 4096,
                                f["num_segments"], f["num_element_runs"], n, stream=st)
    assert (_read(g, out, np.uint32, n) == col).all()
    f = formats.run_position_encoding(col, 4096, np.uint16, True)
    out, out_ptr = _odd_out(g, n, np.uint32)
    keep_p, pos_ptr = _odd(g, f["run_start_positions"])
    schemes.run_position_encoding(out_ptr, g.dev(f["run_data"]), pos_ptr, g.dev(f["position_anchors"]), 4096, f["num_anchors"],
                                  f["num_element_runs"], n, stream=st)
    assert (_read(g, out, np.uint32, n) == col).all()

    colv, sizes = cases.dzbv_column(n, np.uint64, 1, 2, seed=4)
    f = formats.discard_zero_bytes_variable(colv, 1024, 1, 2, np.uint32, 8, element_sizes=sizes)
    out, out_ptr = _odd_out(g, n, np.uint64)
    keep_b, bytes_ptr = _odd(g, f["compressed_data"])          # compressed bytes at an odd address
    schemes.discard_zero_bytes_variable(out_ptr, bytes_ptr, g.dev(f["packed_element_sizes"]), g.dev(f["position_anchors"]), 1024, n, 1, 2, stream=st)
    assert (_read(g, out, np.uint64, n) == colv).all()

    pool = cases.rng(5).choice(1 << 31, size=9, replace=False).astype(np.uint32)
    colb = pool[cases.rng(6).integers(0, 9, size=32 * 2001)]
    f = formats.incidence_bitmaps(colb, 2)
    out, out_ptr = _odd_out(g, len(colb), np.uint32)
    schemes.incidence_bitmaps(out_ptr, g.dev(f["incidence_bitmaps"]), g.dev(f["dictionary_entries"]), f["bitmap_length"], f["num_bitmaps"], stream=st)
    assert (_read(g, out, np.uint32, len(colb)) == colb).all()

    c = cases.rng(7).integers(0, 65535, size=n, endpoint=True).astype(np.uint16)
    keep_in, in_ptr = _odd(g, c)
    out, out_ptr = _odd_out(g, n, np.uint64)
    schemes.discard_zero_bytes_fixed(out_ptr, in_ptr, n, uncompressed_size=8, compressed_size=2, stream=st)
    assert (_read(g, out, np.uint64, n) == c).all()

    out, out_ptr = _odd_out(g, n, np.int32)
    schemes.model(out_ptr, np.array([7, -3], np.int32), n, stream=st)
    assert (_read(g, out, np.int32, n) == (7 - 3 * np.arange(n)).astype(np.int32)).all()


@pytest.mark.parametrize("usize,ksize", [(1, 1), (2, 1), (2, 2), (4, 1), (4, 2), (4, 4), (8, 1), (8, 2), (8, 4)])
def test_dictionary_at_any_alignment(usize, ksize):
    """Dictionary's build for columns at an odd element offset (kernels/decompression/dictionary.hpp, decompress_at_any_alignment): output and keys at every
    combination of a few element offsets from a 16-byte boundary, lengths from nothing to several tiles — bit-exact, nothing written before or past the output."""
    import gpu_util as g
    from libgiddy_amd import schemes
    st = g.stream()
    ud, kd = np.dtype("u%d" % usize), np.dtype("u%d" % ksize)
    guard = np.frombuffer(bytes([0xCD] * usize), dtype=ud)[0]
    entries_count = 256 if ksize == 1 else 3000
    for n in (0, 1, 3, 17, 1000, 70001, 300007):
        f = cases.dictionary_case(max(n, 1), entries_count, ud, kd, seed=n + usize)
        keys = f["compressed_input"][:n]
        expect = f["dictionary_entries"][keys.astype(np.int64)]
        entries = g.dev(f["dictionary_entries"])
        for out_offset, key_offset in ((1, 1), (1, 0), (0, 1), (3, 5), (16 // usize - 1 if usize < 16 else 0, 2), (5, 3), (16 // usize, 0), (0, 0)):
            keys_buffer = g.dev(np.concatenate([np.zeros(key_offset, kd), keys]))
            out_buffer = g.dev_empty(out_offset + n + 8, ud)
            schemes.dictionary(out_buffer.data_ptr() + out_offset * usize, keys_buffer.data_ptr() + key_offset * ksize, entries, n, entries_count,
                               uncompressed_size=usize, dictionary_index_size=ksize, stream=st)
            full = g.host(out_buffer, ud, out_offset + n + 8)
            assert (full[:out_offset] == guard).all() and (full[out_offset + n:] == guard).all(), (n, out_offset, key_offset, "wrote outside the output")
            bad = np.flatnonzero(full[out_offset:out_offset + n] != expect)
            assert len(bad) == 0, (n, out_offset, key_offset, bad[:5])


_OFFSETS = ((1, 1), (1, 0), (0, 1), (3, 5), (5, 3), (0, 0))


@pytest.mark.parametrize("index_size,usize,csize,model,model_index_size", [
    (4, 2, 1, "linear", 4), (4, 4, 1, "linear", 4), (4, 4, 2, "linear", 4), (8, 4, 2, "linear", 8), (4, 2, 1, "constant", 4), (4, 4, 1, "constant", 4), (4, 4, 2, "constant", 4)])
def test_frame_of_reference_at_any_alignment(index_size, usize, csize, model, model_index_size):
    """Frame of Reference's build for columns at an odd element offset (decompress_at_any_alignment): its 16-byte groups start wherever in a segment the output's
    first 128-byte boundary falls — segments that are and are not whole groups, shorter than a group, longer than the column"""
    import gpu_util as g
    from oracle import oracle
    from libgiddy_amd import schemes
    st = g.stream()
    ud, cd = np.dtype("i%d" % usize), np.dtype("i%d" % csize)
    guard = np.frombuffer(bytes([0xCD] * usize), dtype=ud)[0]
    for n, seglen in ((0, 4), (1, 4), (19, 4), (4099, 64), (70001, 12), (70001, 20), (300007, 4096), (300007, 1000), (333, 1000)):
        f = cases.for_case(max(n, 1), seglen, ud, cd, model, seed=n + usize)
        comp = f["compressed_input"][:n]
        expect = oracle.frame_of_reference(comp, f["segment_model_coefficients"], n, seglen, ud, model, model_index_size=model_index_size)
        coefficients = g.dev(f["segment_model_coefficients"])
        for out_offset, in_offset in _OFFSETS + ((16 // usize, 0),):
            in_buffer = g.dev(np.concatenate([np.zeros(in_offset, cd), comp]))
            out_buffer = g.dev_empty(out_offset + n + 8, ud)
            schemes.frame_of_reference(out_buffer.data_ptr() + out_offset * usize, in_buffer.data_ptr() + in_offset * csize, coefficients, n, seglen,
                                       uncompressed_size=usize, compressed_size=csize, model=model, index_size=index_size, model_index_size=model_index_size, stream=st)
            full = g.host(out_buffer, ud, out_offset + n + 8)
            assert (full[:out_offset] == guard).all() and (full[out_offset + n:] == guard).all(), (n, seglen, out_offset, in_offset, "wrote outside the output")
            bad = np.flatnonzero(full[out_offset:out_offset + n] != expect)
            assert len(bad) == 0, (n, seglen, out_offset, in_offset, bad[:5])


@pytest.mark.parametrize("usize,csize", [(2, 1), (4, 1), (4, 2), (8, 1), (8, 2), (8, 4)])
def test_discard_zero_bytes_fixed_at_any_alignment(usize, csize):
    import gpu_util as g
    from libgiddy_amd import schemes
    st = g.stream()
    ud, cd = np.dtype("u%d" % usize), np.dtype("u%d" % csize)
    guard = np.frombuffer(bytes([0xCD] * usize), dtype=ud)[0]
    for n in (0, 1, 17, 1000, 70001, 300007):
        comp = cases.rng(n + usize).integers(0, np.iinfo(cd).max, size=n, endpoint=True, dtype=np.uint64).astype(cd)
        for out_offset, in_offset in _OFFSETS + ((16 // usize, 0),):
            in_buffer = g.dev(np.concatenate([np.zeros(in_offset, cd), comp]))
            out_buffer = g.dev_empty(out_offset + n + 8, ud)
            schemes.discard_zero_bytes_fixed(out_buffer.data_ptr() + out_offset * usize, in_buffer.data_ptr() + in_offset * csize, n,
                                             uncompressed_size=usize, compressed_size=csize, stream=st)
            full = g.host(out_buffer, ud, out_offset + n + 8)
            assert (full[:out_offset] == guard).all() and (full[out_offset + n:] == guard).all(), (n, out_offset, in_offset, "wrote outside the output")
            assert (full[out_offset:out_offset + n] == comp.astype(ud)).all(), (n, out_offset, in_offset)


@pytest.mark.parametrize("usize,csize", [(2, 1), (4, 1), (4, 2), (8, 1), (8, 2), (8, 4)])
def test_delta_at_any_alignment(usize, csize):
    """Delta's build for columns at an odd element offset (kernels/decompression/delta.hpp, decompress_at_any_alignment): windows of the column whose rows follow the
    output's boundaries — segment lengths that are and are not multiples of a lane's elements or of a row, shorter than a row, as long as the 32 rows a window
    takes, and longer (the windows' bases then come from a pre-pass: decompress_windows_at_any_alignment); outputs and deltas at several element offsets"""
    import gpu_util as g
    from libgiddy_amd import schemes
    st = g.stream()
    ud, cd = np.dtype("u%d" % usize), np.dtype("u%d" % csize)
    guard = np.frombuffer(bytes([0xCD] * usize), dtype=ud)[0]
    row = 64 * 16 // usize
    for n, seglen in ((1, 4), (19, 4), (5000, 7), (70001, 48), (70001, row), (300007, 1000), (300007, 4096), (300007, 32 * row), (150001, 32 * row + 8), (70001, 100000),
                      # (round 6: segments longer than the 32-row window — windows whose bases come from the pre-pass: whole and ragged multiples of the window, lengths that are
                      # not multiples of a lane's elements, one boundary early / late in a window, a column shorter than its one segment)
                      (300007, 64 * row), (300007, 32 * row + 1), (1000003, 5 * 32 * row + 3), (700001, 33 * row), (1 << 20, 1 << 18), ((1 << 20) + 5, 3 * 32 * row - 1), (100003, 1 << 20)):
        f, truth = cases.delta_case(n, seglen, ud, cd, seed=n + usize + csize, base=(1 << (8 * usize - 2)))
        anchors = g.dev(f["anchor_values"])
        for out_offset, in_offset in _OFFSETS + ((16 // usize, 1),):
            in_buffer = g.dev(np.concatenate([np.zeros(in_offset, cd), f["compressed_input"]]))
            out_buffer = g.dev_empty(out_offset + n + 8, ud)
            schemes.delta(out_buffer.data_ptr() + out_offset * usize, in_buffer.data_ptr() + in_offset * csize, anchors, n, seglen,
                          uncompressed_size=usize, compressed_size=csize, stream=st)
            full = g.host(out_buffer, ud, out_offset + n + 8)
            assert (full[:out_offset] == guard).all() and (full[out_offset + n:] == guard).all(), (n, seglen, out_offset, in_offset, "wrote outside the output")
            bad = np.flatnonzero(full[out_offset:out_offset + n] != truth)
            assert len(bad) == 0, (n, seglen, out_offset, in_offset, bad[:5])


@pytest.mark.parametrize("usize,csize", [(2, 1), (4, 1), (4, 2), (8, 1), (8, 2), (8, 4)])
def test_delta_long_segments_at_random_offsets(usize, csize):
    """decompress_windows_at_any_alignment (segments longer than the 32-row window, bases from the pre-pass) over random lengths, segment lengths and element
    offsets: the boundary early, late and absent in a window, windows that begin segments, a ragged last window, columns of a single window"""
    import gpu_util as g
    from libgiddy_amd import schemes
    st = g.stream()
    ud, cd = np.dtype("u%d" % usize), np.dtype("u%d" % csize)
    guard = np.frombuffer(bytes([0xCD] * usize), dtype=ud)[0]
    window = 32 * 64 * 16 // usize
    r = cases.rng(1234 + usize * 10 + csize)
    for trial in range(24):
        seglen = int(r.integers(window + 1, 6 * window))
        if trial % 3 == 0:
            seglen = int(r.integers(1, 5)) * window + int(r.integers(0, 2)) * (16 // usize)   # whole windows, or a lane's group more
        n = int(r.integers(1, 12 * window))
        out_offset, in_offset = int(r.integers(0, 3 * 16 // usize)), int(r.integers(0, 17))
        if out_offset % (16 // usize) == 0 and in_offset % (16 // usize) == 0:
            out_offset += 1   # (aligned addresses take the ordinary paths)
        f, truth = cases.delta_case(n, seglen, ud, cd, seed=trial + usize, base=(1 << (8 * usize - 2)))
        anchors = g.dev(f["anchor_values"])
        in_buffer = g.dev(np.concatenate([np.zeros(in_offset, cd), f["compressed_input"]]))
        out_buffer = g.dev_empty(out_offset + n + 8, ud)
        schemes.delta(out_buffer.data_ptr() + out_offset * usize, in_buffer.data_ptr() + in_offset * csize, anchors, n, seglen,
                      uncompressed_size=usize, compressed_size=csize, stream=st)
        full = g.host(out_buffer, ud, out_offset + n + 8)
        assert (full[:out_offset] == guard).all() and (full[out_offset + n:] == guard).all(), (n, seglen, out_offset, in_offset, "wrote outside the output")
        bad = np.flatnonzero(full[out_offset:out_offset + n] != truth)
        assert len(bad) == 0, (n, seglen, out_offset, in_offset, bad[:5])


@pytest.mark.parametrize("usize", [1, 2, 4, 8])
def test_run_decoders_store_their_rows_at_any_output_offset(usize):
    """run lengths and run positions (absolute, relative) into outputs 1, 3 and 5 elements past a 16-byte boundary: a tile's rows are stored 16 bytes per lane at
    whatever address (kernels/decompression/run_expansion.hpp, expand) — every element size, segments that are and are not whole tiles, short runs and long"""
    import gpu_util as g
    from libgiddy_amd import schemes
    st = g.stream()
    ud = np.dtype("u%d" % usize)
    guard = np.frombuffer(bytes([0xCD] * usize), dtype=ud)[0]
    n = 150001
    for mean_run in (3, 40):
        col = cases.runs_column(n, mean_run, ud, seed=usize + mean_run, max_run=255)
        for seglen in (4096, 1000, 100):
            rle = formats.run_length_encoding(col, seglen, np.uint8 if usize < 8 else np.uint16)
            rpe_absolute = formats.run_position_encoding(col, seglen, "u4", False, 4)
            rpe_relative = formats.run_position_encoding(col, seglen, "u2", True, 4)
            for out_offset in (1, 3, 5):
                def check(decode):
                    out_buffer = g.dev_empty(out_offset + n + 8, ud)
                    decode(out_buffer.data_ptr() + out_offset * usize)
                    full = g.host(out_buffer, ud, out_offset + n + 8)
                    assert (full[:out_offset] == guard).all() and (full[out_offset + n:] == guard).all(), (seglen, out_offset, "wrote outside the output")
                    bad = np.flatnonzero(full[out_offset:out_offset + n] != col)
                    assert len(bad) == 0, (mean_run, seglen, out_offset, bad[:5])
                check(lambda out: schemes.run_length_encoding(out, g.dev(rle["run_data"]), g.dev(rle["run_lengths"]), g.dev(rle["position_anchors"]),
                                                              g.dev(rle["intra_run_anchor_offsets"]), seglen, rle["num_segments"], rle["num_element_runs"], n,
                                                              uncompressed_size=usize, run_length_size=rle["run_lengths"].dtype.itemsize, stream=st))
                for f, psize, relative in ((rpe_absolute, 4, False), (rpe_relative, 2, True)):
                    check(lambda out: schemes.run_position_encoding(out, g.dev(f["run_data"]), g.dev(f["run_start_positions"]), g.dev(f["position_anchors"]), seglen,
                                                                    f["num_anchors"], f["num_element_runs"], n, uncompressed_size=usize, position_size=psize,
                                                                    relative=relative, stream=st))

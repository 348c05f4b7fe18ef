"""Batches (include/giddy/batch.hpp): many small columns decoded by ONE launch — every piece bit-exact against the CPU oracle.

The pieces of a batch are independent columns of mixed lengths (0 and 1 among them), mixed segment lengths and, for the
run decoders, mixed run densities; each is checked against the oracle's decode of ITS arrays, its output's guard bands
included (gpu_util.dev_empty).  Every registration of the seven wrappers that launch batches is covered, through the C ABI
(giddy_batch_create / giddy_batch_add / giddy_enqueue_batch, and giddy_enqueue_batch_of_device_tables for tables that lie
on the device already).  Run with `-m gpu` on an MI355X."""
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


def _imports():
    from libgiddy_amd import schemes, registered
    import gpu_util
    return schemes, registered, gpu_util


LENGTHS = (0, 1, 5, 64, 1000, 4096, 4097, 50000, 262144 + 7, 3, 12288)


def _check(pieces, expected, outputs, dtypes, g, stream):
    """launch the batch twice (the second launch re-uses the uploaded tables) and compare every piece"""
    for _ in range(2):
        for out in outputs:
            out.fill_(0xCD)
        pieces.enqueue(stream)
        for i, (out, want, dtype) in enumerate(zip(outputs, expected, dtypes)):
            got = g.host(out, dtype, len(want))
            bad = np.flatnonzero(got != want)
            assert len(bad) == 0, ("piece", i, "of", len(outputs), "length", len(want), bad[:5], got[bad[:5]], want[bad[:5]])


DICTIONARY_TUPLES = [(i, u, d) for i in (4, 8) for (u, d) in ((1, 1), (2, 1), (2, 2), (4, 1), (4, 2), (4, 4), (8, 1), (8, 2), (8, 4), (8, 8))]


@pytest.mark.parametrize("index_size,usize,dsize", DICTIONARY_TUPLES)
def test_dictionary_batch(index_size, usize, dsize):
    schemes, r, g = _imports()
    ud, kd = np.dtype("u%d" % usize), np.dtype("u%d" % dsize)
    outputs, expected, keep = [], [], []
    with schemes.into_batch() as pieces:
        for i, n in enumerate(LENGTHS):
            entries = (7, 256, 300, 5000)[i % 4]
            entries = min(entries, 1 << (8 * dsize))
            f = cases.dictionary_case(n, entries, ud, kd, seed=100 + i)
            expected.append(oracle.dictionary(f["compressed_input"], f["dictionary_entries"]))
            out, keys, table = g.dev_empty(n, ud), g.dev(f["compressed_input"]), g.dev(f["dictionary_entries"])
            schemes.dictionary(out, keys, table, n, entries, uncompressed_size=usize, dictionary_index_size=dsize, index_size=index_size)
            outputs.append(out); keep += [keys, table]
    assert pieces.num_pieces == len(LENGTHS)
    _check(pieces, expected, outputs, [ud] * len(outputs), g, g.stream())


FOR_TUPLES = [(4, 2, 1, "linear", 4), (4, 4, 1, "linear", 4), (4, 4, 2, "linear", 4), (8, 4, 2, "linear", 4),
              (8, 4, 2, "linear", 8), (4, 2, 1, "constant", 4), (4, 4, 1, "constant", 4), (4, 4, 2, "constant", 4)]


@pytest.mark.parametrize("index_size,usize,csize,model,model_index_size", FOR_TUPLES)
def test_frame_of_reference_batch(index_size, usize, csize, model, model_index_size):
    schemes, r, g = _imports()
    ud, cd = np.dtype("i%d" % usize), np.dtype("i%d" % csize)
    outputs, expected, keep = [], [], []
    with schemes.into_batch() as pieces:
        for i, n in enumerate(LENGTHS):
            seglen = (4096, 64, 1 << 20, 12, 1000)[i % 5]
            f = cases.for_case(n, seglen, ud, cd, model, seed=200 + i)
            expected.append(oracle.frame_of_reference(f["compressed_input"], f["segment_model_coefficients"], n, seglen, ud, model, model_index_size=model_index_size))
            out, comp, coefs = g.dev_empty(n, ud), g.dev(f["compressed_input"]), g.dev(f["segment_model_coefficients"])
            schemes.frame_of_reference(out, comp, coefs, n, seglen, uncompressed_size=usize, compressed_size=csize, model=model,
                                       index_size=index_size, model_index_size=model_index_size)
            outputs.append(out); keep += [comp, coefs]
    _check(pieces, expected, outputs, [ud] * len(outputs), g, g.stream())


DELTA_TUPLES = [(i, u, c) for i in (4, 8) for (u, c) in ((2, 1), (4, 1), (4, 2), (8, 1), (8, 2), (8, 4))]


@pytest.mark.parametrize("index_size,usize,csize", DELTA_TUPLES)
def test_delta_batch(index_size, usize, csize):
    """segment lengths of every path of delta::decompress: whole rows, shorter than a row, ragged, odd, one segment for the piece"""
    schemes, r, g = _imports()
    ud, cd = np.dtype("u%d" % usize), np.dtype("u%d" % csize)
    outputs, expected, keep = [], [], []
    with schemes.into_batch() as pieces:
        for i, n in enumerate(LENGTHS + (100000, 70001)):
            seglen = (4096, 16, 1000, 3000, 1 << 20, 1001, 128, 8192)[i % 8]
            f, column = cases.delta_case(n, seglen, ud, cd, seed=300 + i)
            if n:
                assert (oracle.delta(f["compressed_input"], f["anchor_values"], seglen) == column).all()
            expected.append(column)
            out, comp, anchors = g.dev_empty(n, ud), g.dev(f["compressed_input"]), g.dev(f["anchor_values"])
            schemes.delta(out, comp, anchors, n, seglen, uncompressed_size=usize, compressed_size=csize, index_size=index_size)
            outputs.append(out); keep += [comp, anchors]
    _check(pieces, expected, outputs, [ud] * len(outputs), g, g.stream())


DZBF_TUPLES = [(i, u, c) for i in (4, 8) for (u, c) in ((2, 1), (4, 1), (4, 2), (8, 1), (8, 2), (8, 4))]


@pytest.mark.parametrize("index_size,usize,csize", DZBF_TUPLES)
def test_discard_zero_bytes_fixed_batch(index_size, usize, csize):
    schemes, r, g = _imports()
    ud, cd = np.dtype("u%d" % usize), np.dtype("u%d" % csize)
    outputs, expected, keep = [], [], []
    with schemes.into_batch() as pieces:
        for i, n in enumerate(LENGTHS):
            narrow = cases.rng(400 + i).integers(0, np.iinfo(cd).max, size=n, endpoint=True, dtype=np.uint64).astype(cd)
            expected.append(oracle.dzb_fixed(narrow, ud))
            out, comp = g.dev_empty(n, ud), g.dev(narrow)
            schemes.discard_zero_bytes_fixed(out, comp, n, uncompressed_size=usize, compressed_size=csize, index_size=index_size)
            outputs.append(out); keep.append(comp)
    _check(pieces, expected, outputs, [ud] * len(outputs), g, g.stream())


DZBV_TUPLES = [(4, u, 4) for u in (2, 4, 8)] + [(8, u, 4) for u in (2, 4, 8)] + [(4, u, 8) for u in (2, 4, 8)]


@pytest.mark.parametrize("index_size,usize,container", DZBV_TUPLES)
@pytest.mark.parametrize("min_size,bits", [(1, 2), (0, 3), (1, 1)])
def test_discard_zero_bytes_variable_batch(index_size, usize, container, min_size, bits):
    schemes, r, g = _imports()
    if min_size > usize:
        pytest.skip("min size exceeds element width")
    ud = np.dtype("u%d" % usize)
    outputs, expected, keep = [], [], []
    with schemes.into_batch() as pieces:
        for i, n in enumerate(LENGTHS):
            if n == 0:
                continue   # (the format writer has no empty column)
            seglen = (1024, 64, 1000, 4096, 16, 1 << 20)[i % 6]
            col, sizes = cases.dzbv_column(n, ud, min_size, bits, seed=500 + i)
            f = formats.discard_zero_bytes_variable(col, seglen, min_size, bits, "u%d" % container, index_size, element_sizes=sizes)
            expected.append(col)
            out = g.dev_empty(n, ud)
            arrays = [g.dev(f["compressed_data"]), g.dev(f["packed_element_sizes"]), g.dev(f["position_anchors"])]
            schemes.discard_zero_bytes_variable(out, arrays[0], arrays[1], arrays[2], seglen, n, min_size, bits,
                                                uncompressed_size=usize, container_size=container, index_size=index_size)
            outputs.append(out); keep += arrays
    _check(pieces, expected, outputs, [ud] * len(outputs), g, g.stream())


def _run_pieces(ud, dense):
    """columns of runs for a batch: by `dense`, mean runs of 2-6 (the batch's mean run picks the dense build), of 30-1000 (sparse) or both (the two-path build)"""
    means = {True: (2, 3, 6, 4), False: (32, 200, 1000, 60), None: (2, 500, 3, 40)}[dense]
    for i, n in enumerate(LENGTHS + (200000, 65536)):
        if n == 0:
            continue   # (the format writers have no empty column)
        yield i, cases.runs_column(n, means[i % 4], np.uint32, seed=600 + i).astype(np.uint64).astype(ud)


RLE_TUPLES = [(4, u, l, 4) for l in (1, 2, 4) for u in (1, 2, 4, 8)] + [(8, 4, 1, 8), (8, 4, 2, 8), (8, 4, 4, 8), (8, 1, 8, 8), (8, 4, 8, 8), (8, 8, 8, 8)]


@pytest.mark.parametrize("index_size,usize,lsize,run_index_size", RLE_TUPLES)
@pytest.mark.parametrize("dense", [True, False, None], ids=["short_runs", "long_runs", "mixed_runs"])
def test_run_length_encoding_batch(index_size, usize, lsize, run_index_size, dense):
    schemes, r, g = _imports()
    ud = np.dtype("u%d" % usize)
    outputs, expected, keep = [], [], []
    with schemes.into_batch() as pieces:
        for i, col in _run_pieces(ud, dense):
            n = len(col)
            seglen = (4096, 100, 1024, 20000, 65536, 7, 3000)[i % 7]
            f = formats.run_length_encoding(col, seglen, "u%d" % lsize, run_index_size)
            assert (oracle.run_length_encoding(f["run_data"], f["run_lengths"], f["position_anchors"], f["intra_run_anchor_offsets"], seglen, n) == col).all()
            expected.append(col)
            out = g.dev_empty(n, ud)
            arrays = [g.dev(f[k]) for k in ("run_data", "run_lengths", "position_anchors", "intra_run_anchor_offsets")]
            schemes.run_length_encoding(out, *arrays, seglen, f["num_segments"], f["num_element_runs"], n,
                                        uncompressed_size=usize, run_length_size=lsize, index_size=index_size, run_index_size=run_index_size)
            outputs.append(out); keep += arrays
    _check(pieces, expected, outputs, [ud] * len(outputs), g, g.stream())


RPE_TUPLES = [(4, 1, 2, True), (4, 2, 2, True), (4, 4, 2, True), (4, 8, 2, True), (8, 4, 2, True),
              (4, 1, 4, False), (4, 2, 4, False), (4, 4, 4, False), (4, 8, 4, False), (8, 4, 8, False)]


@pytest.mark.parametrize("index_size,usize,psize,relative", RPE_TUPLES)
@pytest.mark.parametrize("dense", [True, False, None], ids=["short_runs", "long_runs", "mixed_runs"])
def test_run_position_encoding_batch(index_size, usize, psize, relative, dense):
    schemes, r, g = _imports()
    ud = np.dtype("u%d" % usize)
    outputs, expected, keep = [], [], []
    with schemes.into_batch() as pieces:
        for i, col in _run_pieces(ud, dense):
            n = len(col)
            seglen = (4096, 100, 1024, 20000, 65536, 7, 3000)[i % 7]
            f = formats.run_position_encoding(col, seglen, "u%d" % psize, relative, index_size)
            assert (oracle.run_position_encoding(f["run_data"], f["run_start_positions"], f["position_anchors"], seglen, n, relative) == col).all()
            expected.append(col)
            out = g.dev_empty(n, ud)
            arrays = [g.dev(f[k]) for k in ("run_data", "run_start_positions", "position_anchors")]
            schemes.run_position_encoding(out, *arrays, seglen, f["num_anchors"], f["num_element_runs"], n, uncompressed_size=usize,
                                          position_size=psize, relative=relative, index_size=index_size)
            outputs.append(out); keep += arrays
    _check(pieces, expected, outputs, [ud] * len(outputs), g, g.stream())


def test_many_pieces_and_pieces_at_odd_offsets():
    """2000 dictionary columns of 1 ... 3000 elements decoded into consecutive slices of ONE buffer (most begin inside a 128-byte line:
    the batch takes the any-alignment build), then the same with every piece on a line"""
    schemes, r, g = _imports()
    rr = cases.rng(7)
    lengths = rr.integers(1, 3000, size=2000)
    entries = rr.integers(0, 1 << 32, size=256, dtype=np.uint64).astype(np.uint32)
    keys = rr.integers(0, 256, size=int(lengths.sum()) + 64 * len(lengths)).astype(np.uint8)
    table, all_keys = g.dev(entries), g.dev(keys)
    for aligned in (False, True):
        starts = np.concatenate(([0], np.cumsum((lengths + 31) // 32 * 32 if aligned else lengths)[:-1]))
        out = g.dev_empty(int(starts[-1] + lengths[-1]), np.uint32)
        with schemes.into_batch() as pieces:
            for start, n in zip(starts, lengths):
                schemes.dictionary(out.data_ptr() + 4 * int(start), all_keys.data_ptr() + int(start), table, int(n), 256)
        assert pieces.num_pieces == 2000
        pieces.enqueue(g.stream())
        got = g.host(out, np.uint32, int(starts[-1] + lengths[-1]))
        for start, n in zip(starts, lengths):
            assert (got[start:start + n] == entries[keys[start:start + n]]).all(), (aligned, start, n)


def test_tables_written_elsewhere_and_errors():
    """giddy_enqueue_batch_of_device_tables on the tables of an uploaded batch (where a caller's own kernel may as well have written them);
    a wrapper that does not launch batches says so; a batch holds one kernel instantiation"""
    schemes, r, g = _imports()
    f = cases.dictionary_case(70000, 256, np.uint32, np.uint8, seed=9)
    keys, table = g.dev(f["compressed_input"]), g.dev(f["dictionary_entries"])
    outs = [g.dev_empty(70000, np.uint32) for _ in range(3)]
    with schemes.into_batch() as pieces:
        for out in outs:
            schemes.dictionary(out, keys, table, 70000, 256)
    pieces.batch.upload(g.stream())
    tables = pieces.batch.tables()
    assert tables.num_pieces == 3 and tables.num_blocks == pieces.num_blocks and tables.block_size > 0
    schemes.kernel(pieces.key).enqueue_launch_of_batch(g.stream(), tables)
    want = oracle.dictionary(f["compressed_input"], f["dictionary_entries"])
    for out in outs:
        assert (g.host(out, np.uint32, 70000) == want).all()
    scatter = schemes.kernel(schemes.key_scatter(4, 8, 4))
    assert not scatter.launches_batches() and schemes.kernel(pieces.key).launches_batches()
    with pytest.raises(r.LogicError):
        scatter.batch()
    with pytest.raises(r.InvalidArgument):
        with schemes.into_batch():
            schemes.dictionary(outs[0], keys, table, 70000, 256)
            schemes.dictionary(outs[0], keys, table, 70000, 256, uncompressed_size=8)
    with pytest.raises(r.OutOfRange):   # a piece with an argument missing: as enqueue_launch
        batch = schemes.kernel(pieces.key).batch()
        batch.add({"decompressed": r.pointer(outs[0], "u32")}, {"num_dictionary_entries": r.size_t(256), "data_length": r.size_t(70000)})

"""GPU parity (bit-exact) of the remaining HIP kernels against the CPU oracle, through the C ABI:
DZB fixed / variable, Model, Incidence bitmaps, RLE, RPE, compressed-indices scatter — every
registered instantiation at least once, plus the edge shapes of tests/test_oracle.py."""
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


@pytest.mark.parametrize("usize,csize", [(2, 1), (4, 1), (4, 2), (8, 1), (8, 2), (8, 4)])
@pytest.mark.parametrize("index_size", [4, 8])
def test_dzb_fixed(usize, csize, index_size):
    schemes, r, g = _imports()
    for n in (0, 1, 1237, 1 << 20):
        c = cases.rng(n).integers(0, np.iinfo("u%d" % csize).max, size=n, endpoint=True).astype("u%d" % csize)
        out = g.dev_empty(n, "u%d" % usize)
        schemes.discard_zero_bytes_fixed(out, g.dev(c), n, uncompressed_size=usize, compressed_size=csize,
                                         index_size=index_size, stream=g.stream())
        assert (g.host(out, "u%d" % usize, n) == oracle.dzb_fixed(c, "u%d" % usize)).all()


@pytest.mark.parametrize("index_size,usize,model", [(4, 2, "linear"), (4, 4, "linear"), (8, 4, "linear"), (4, 2, "constant"), (4, 4, "constant")])
def test_model(index_size, usize, model):
    schemes, r, g = _imports()
    dt = np.dtype("i%d" % usize)
    coefs = np.array([12345, -7] if model == "linear" else [-321], dtype=dt)
    for n in (1, 70001, 1 << 20):
        out = g.dev_empty(n, dt)
        schemes.model(out, coefs, n, uncompressed_size=usize, model=model, index_size=index_size, stream=g.stream())
        assert (g.host(out, dt, n) == oracle.model(coefs, n, dt, model, model_index_size=index_size)).all()


DZBV_TUPLES = [(4, 2, 4), (4, 4, 4), (4, 8, 4), (8, 2, 4), (8, 4, 4), (8, 8, 4), (4, 2, 8), (4, 4, 8), (4, 8, 8)]


@pytest.mark.parametrize("index_size,usize,container", DZBV_TUPLES)
@pytest.mark.parametrize("min_size,bits", [(1, 2), (1, 3), (0, 2), (2, 1), (0, 4)])
def test_dzb_variable(index_size, usize, container, min_size, bits):
    schemes, r, g = _imports()
    if min_size > usize:
        pytest.skip("min size exceeds element width")
    ud = np.dtype("u%d" % usize)
    # (long segments that are whole 1024s: finer anchors from the size fields first — incl. a segment of more than 4096 x 1024 elements,
    # a column that ends inside a 1024, one segment for the whole column)
    shapes = [(1, 32), (1000, 64), (1024, 1024), (333333, 1024), (50, 7), (100000, 4096), (70000, 1000), (300000, 32768), (200000 + 77, 65536), (150000, 1 << 20)]
    # (segments shorter than a row of 64 lanes' elements are taken several at a time, through every k-th anchor: lengths that do and do not divide
    # the super-row, odd ones, a column that ends inside a coalesced segment — the reference's cuh:366-369 says the anchors may be ignored)
    shapes += [(250000, 64), (250000 + 37, 16), (99999, 48), (123456, 100), (65536 + 8, 128), (77777, 3), (200000, 255), (200000, 96)]
    # (segments of a row or more that are not whole 1024s: every k-th anchor up to the least common multiple with 1024, then finer anchors — and one odd segment for the column)
    shapes += [(400000, 1000), (400001, 1023), (250000, 2050), (300000, 10000), (123457, 300000), (70000, 520)]
    if (usize, bits, min_size) == (8, 2, 1):
        shapes += [((5 << 20) + 3000, 5 << 20), (3 << 20, 1 << 20)]
    for n, seglen in shapes:
        col, sizes = cases.dzbv_column(n, ud, min_size, bits, seed=n + bits + usize)
        f = formats.discard_zero_bytes_variable(col, seglen, min_size, bits, "u%d" % container, index_size, element_sizes=sizes)
        out = g.dev_empty(n, ud)
        # exact-size device buffer for the compressed bytes: nothing may be read past its last 16-byte chunk
        schemes.discard_zero_bytes_variable(out, g.dev(f["compressed_data"]), g.dev(f["packed_element_sizes"]),
                                            g.dev(f["position_anchors"]), seglen, n, min_size, bits,
                                            uncompressed_size=usize, container_size=container, index_size=index_size,
                                            stream=g.stream())
        got = g.host(out, ud, n)
        bad = np.flatnonzero(got != col)
        assert len(bad) == 0, (n, seglen, bad[:5], got[bad[:5]], col[bad[:5]])


@pytest.mark.parametrize("usize,use_dictionary", [(1, True), (2, True), (4, True), (8, True), (1, False)])
def test_incidence_bitmaps(usize, use_dictionary):
    schemes, r, g = _imports()
    ud = np.dtype("u%d" % usize)
    for n_words, num_values in [(1, 1), (5, 3), (64, 17), (1001, 200), (50000, 33)]:
        rr = cases.rng(n_words + usize)
        if use_dictionary:
            num_values = min(num_values, int(np.iinfo(ud).max) + 1)
            pool = rr.choice(min(int(np.iinfo(ud).max) + 1, 1 << 40), size=num_values, replace=False).astype(ud)
            col = pool[rr.integers(0, num_values, size=32 * n_words)]
        else:
            col = rr.integers(0, min(num_values, 200), size=32 * n_words).astype(ud)
        f = formats.incidence_bitmaps(col, 2, use_dictionary)
        out = g.dev_empty(32 * n_words, ud)
        schemes.incidence_bitmaps(out, g.dev(f["incidence_bitmaps"]), g.dev(f["dictionary_entries"]), f["bitmap_length"],
                                  f["num_bitmaps"], uncompressed_size=usize, use_dictionary=use_dictionary, stream=g.stream())
        assert (g.host(out, ud, 32 * n_words) == col).all(), (n_words, num_values)


@pytest.mark.parametrize("usize,use_dictionary", [(1, True), (2, True), (4, True), (8, True), (1, False)])
def test_incidence_bitmaps_that_overlap(usize, use_dictionary):
    """Bitmaps that claim the same elements: the HIGHEST bitmap index wins, as in the reference's loop over the bitmaps in ascending
    order (src/kernels/decompression/incidence_bitmaps.cuh:47-60: a later bitmap's set bit overwrites an earlier one's).  The format
    writer only produces disjoint bitmaps, so these are made by hand: the writer's bitmaps, then bits of LOWER-index bitmaps set on
    top of elements that belong to higher ones (must not change the result), and bits of HIGHER-index bitmaps set on elements of lower
    ones (must) — two and three bitmaps per element, single bits at container and tile boundaries, and whole containers."""
    schemes, r, g = _imports()
    from oracle import oracle
    ud = np.dtype("u%d" % usize)
    for n_words, num_values in [(1, 2), (5, 3), (64, 17), (1001, 60), (20000, 33)]:
        rr = cases.rng(1000 + n_words + usize)
        if use_dictionary:
            num_values = min(num_values, int(np.iinfo(ud).max) + 1)
            pool = rr.choice(min(int(np.iinfo(ud).max) + 1, 1 << 40), size=num_values, replace=False).astype(ud)
            col = pool[rr.integers(0, num_values, size=32 * n_words)]
        else:
            col = rr.integers(0, min(num_values, 200), size=32 * n_words).astype(ud)
        f = formats.incidence_bitmaps(col, 2, use_dictionary)
        num_bitmaps, stride = f["num_bitmaps"], len(f["incidence_bitmaps"]) // max(f["num_bitmaps"], 1)
        if num_bitmaps < 2:
            continue
        bitmaps = f["incidence_bitmaps"].copy().reshape(num_bitmaps, stride)
        n = 32 * n_words
        # elements to be claimed twice or three times: random ones, plus the first / last bit of containers and of 64-container tiles
        picks = set(int(e) for e in rr.integers(0, n, size=min(n, 400)))
        for w in {0, n_words - 1, min(63, n_words - 1), min(64, n_words - 1), n_words // 2}:
            picks.update({32 * w, 32 * w + 31})
        for e in sorted(picks):
            word, bit = e // 32, np.uint32(1 << (e % 32))
            claimants = rr.choice(num_bitmaps, size=min(num_bitmaps, int(rr.integers(1, 3))), replace=False)
            for b in claimants:
                bitmaps[b, word] |= bit
        # ... and one whole container claimed by every bitmap
        bitmaps[:, n_words // 3] = np.uint32(0xFFFFFFFF)
        flat = np.ascontiguousarray(bitmaps.reshape(-1))
        expect, uncovered = oracle.incidence_bitmaps(flat, f["dictionary_entries"], f["bitmap_length"], num_bitmaps, 2, ud, use_dictionary)
        assert uncovered == 0
        assert (expect != col).any(), "the overlaps changed nothing: the case tests nothing"
        out = g.dev_empty(n, ud)
        schemes.incidence_bitmaps(out, g.dev(flat), g.dev(f["dictionary_entries"]), f["bitmap_length"], num_bitmaps,
                                  uncompressed_size=usize, use_dictionary=use_dictionary, stream=g.stream())
        got = g.host(out, ud, n)
        bad = np.flatnonzero(got != expect)
        assert len(bad) == 0, (n_words, num_bitmaps, bad[:5], got[bad[:5]], expect[bad[:5]])


def _run_columns():
    yield "geom32", cases.runs_column(300000, 32, np.uint32, seed=6)
    yield "geom2", cases.runs_column(50000, 2, np.uint32, seed=7)
    yield "geom1000", cases.runs_column(400000, 1000, np.uint32, seed=8)
    yield "single", np.full(50000, 77, dtype=np.uint32)
    yield "one", np.array([5], dtype=np.uint32)
    yield "alldistinct", np.arange(30000, dtype=np.uint32)
    yield "boundary", np.repeat(np.arange(10, dtype=np.uint32) * 3 + 1, [1024, 512, 512, 2048, 1, 1023, 1024, 100, 924, 1000])
    # one-byte columns expand 16 outputs per lane by byte permutation when at most 8 runs cover them: around that limit
    yield "pairs", np.repeat(np.arange(20000, dtype=np.uint32), 2)
    yield "eight_per_16", np.repeat(np.arange(24000, dtype=np.uint32), np.tile([1, 1, 1, 1, 1, 1, 1, 9], 3000))
    yield "nine_per_16", np.repeat(np.arange(27000, dtype=np.uint32), np.tile([1, 1, 1, 1, 1, 1, 1, 1, 8], 3000))
    yield "short_mix", np.repeat(np.arange(30000, dtype=np.uint32), cases.rng(21).integers(1, 5, size=30000))
    yield "tileboundary", np.repeat(np.arange(6, dtype=np.uint32) + 9, [4096, 4096, 8192, 1, 4095, 12288])
    # stretches of very short runs (dense tiles: cut where a batch of 256 run descriptors ends, values prefetched for the next tile) between
    # stretches of long ones, so that segments of several tiles change from one kind of tile to the other and back
    r = cases.rng(33)
    lengths = np.concatenate([np.concatenate((r.integers(1, 4, size=int(r.integers(300, 3000))), r.integers(500, 5000, size=int(r.integers(2, 30))))) for _ in range(12)])
    yield "density_changes", np.repeat(np.arange(len(lengths), dtype=np.uint32) * 7 + 1, lengths)
    # a batch of 256 runs that ends exactly on the tile's end (the carry is found by one more batch), and one that ends on a whole piece of a dense tile
    yield "sixteens", np.repeat(np.arange(5000, dtype=np.uint32), 16)
    yield "eights", np.repeat(np.arange(9000, dtype=np.uint32), 8)


RLE_TUPLES = [(4, u, l, 4) for l in (1, 2, 4) for u in (1, 2, 4, 8)] + [(8, 4, 1, 8), (8, 4, 2, 8), (8, 4, 4, 8), (8, 1, 8, 8), (8, 4, 8, 8), (8, 8, 8, 8)]


@pytest.mark.parametrize("index_size,usize,lsize,run_index_size", RLE_TUPLES)
def test_rle(index_size, usize, lsize, run_index_size):
    schemes, r, g = _imports()
    ud = np.dtype("u%d" % usize)
    for name, col in _run_columns():
        col = col.astype(np.uint64).astype(ud)
        n = len(col)
        # (the short ones: several segments per wavefront trip; 16384 and up: finer anchors first — kernels/decompression/run_anchors.hpp)
        for seglen in (1024, 4096, 100, 20000, 65536, 64, 128, 512, 16384, 1 << 20, 1000, 3000, 50000, 7):
            f = formats.run_length_encoding(col, seglen, "u%d" % lsize, run_index_size)
            expect = oracle.run_length_encoding(f["run_data"], f["run_lengths"], f["position_anchors"], f["intra_run_anchor_offsets"], seglen, n)
            assert (expect == col).all()
            out = g.dev_empty(n, ud)
            schemes.run_length_encoding(out, g.dev(f["run_data"]), g.dev(f["run_lengths"]), g.dev(f["position_anchors"]),
                                        g.dev(f["intra_run_anchor_offsets"]), seglen, f["num_segments"], f["num_element_runs"], n,
                                        uncompressed_size=usize, run_length_size=lsize, index_size=index_size,
                                        run_index_size=run_index_size, stream=g.stream())
            got = g.host(out, ud, n)
            bad = np.flatnonzero(got != col)
            assert len(bad) == 0, (name, seglen, bad[:5], got[bad[:5]], col[bad[:5]])


@pytest.mark.parametrize("switches", [{"GIDDY_RUN_FINE_ONE_LAUNCH": "2", "GIDDY_RUN_FINE_FORCE": "1"}, {"GIDDY_RUN_FINE_ONE_LAUNCH": "0", "GIDDY_RUN_FINE_FORCE": "1"},
                                      {"GIDDY_RUN_FINE_ONE_LAUNCH": "2", "GIDDY_RUN_FINE_FORCE": "1", "GIDDY_RUN_FUSED_PREPASS": "0"},
                                      {"GIDDY_RUN_FUSED_FROM": "1", "GIDDY_RUN_GROUP_ALL": "1"}],
                         ids=["pre_pass_and_decoder_in_one_kernel", "three_launch_pre_pass", "one_launch_pre_pass", "groups_of_segments_in_one_kernel"])
def test_rle_with_each_pre_pass_forced(switches):
    """The wrapper picks the pre-pass that finds finer anchors for long segments by the column's shape (one launch: a workgroup per segment;
    three launches: sums, scan, search).  Here every run-length registration decodes every column of test_rle with EACH of them forced for
    every shape that admits finer anchors (the switches are read once per process: the cases run in a child interpreter)."""
    import os
    import subprocess
    import sys
    environment = dict(os.environ, **switches)
    result = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", "test_rle and not forced", "-p", "no:cacheprovider"],
                            capture_output=True, text=True, env=environment, timeout=1500)
    assert result.returncode == 0, result.stdout[-3000:] + result.stderr[-2000:]
    assert "18 passed" in result.stdout, result.stdout[-500:]


RPE_TUPLES = [(4, 1, 2, True), (4, 2, 2, True), (4, 4, 2, True), (4, 8, 2, True), (8, 4, 2, True),
              (4, 1, 4, False), (4, 2, 4, False), (4, 4, 4, False), (4, 8, 4, False), (8, 4, 8, False)]


@pytest.mark.parametrize("index_size,usize,psize,relative", RPE_TUPLES)
def test_rpe(index_size, usize, psize, relative):
    schemes, r, g = _imports()
    ud = np.dtype("u%d" % usize)
    for name, col in _run_columns():
        col = col.astype(np.uint64).astype(ud)
        n = len(col)
        # (the short ones: several segments per wavefront trip; 16384 and up: finer anchors first — kernels/decompression/run_anchors.hpp)
        for seglen in (1024, 4096, 100, 20000, 65536, 64, 128, 512, 16384, 1 << 20, 1000, 3000, 50000, 7):
            if relative and seglen > 1 << (8 * psize): continue   # (relative positions must fit their type)
            f = formats.run_position_encoding(col, seglen, "u%d" % psize, relative, index_size)
            expect = oracle.run_position_encoding(f["run_data"], f["run_start_positions"], f["position_anchors"], seglen, n, relative)
            assert (expect == col).all()
            out = g.dev_empty(n, ud)
            schemes.run_position_encoding(out, g.dev(f["run_data"]), g.dev(f["run_start_positions"]), g.dev(f["position_anchors"]),
                                          seglen, f["num_anchors"], f["num_element_runs"], n, uncompressed_size=usize,
                                          position_size=psize, relative=relative, index_size=index_size, stream=g.stream())
            got = g.host(out, ud, n)
            bad = np.flatnonzero(got != col)
            assert len(bad) == 0, (name, seglen, bad[:5], got[bad[:5]], col[bad[:5]])


def test_rpe_with_the_one_kernel_builds_forced():
    """every run-position registration decodes every column of test_rpe with the long segments' one-kernel build — and groups of segments through it (relative
    positions: from segments of 256 outputs on) — taken wherever the shape admits it, however few workgroups that makes (the switches are read once per process: a child interpreter)"""
    import os
    import subprocess
    import sys
    environment = dict(os.environ, GIDDY_RUN_FUSED_FROM="1", GIDDY_RUN_GROUP_ALL="1")
    result = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", "test_rpe and not forced", "-p", "no:cacheprovider"],
                            capture_output=True, text=True, env=environment, timeout=1500)
    assert result.returncode == 0, result.stdout[-3000:] + result.stderr[-2000:]
    assert "10 passed" in result.stdout, result.stdout[-500:]


@pytest.mark.parametrize("osize", [4, 8])
@pytest.mark.parametrize("esize", [1, 2, 4, 8])
@pytest.mark.parametrize("isize", [1, 2, 4, 8])
def test_compressed_indices_scatter(osize, esize, isize):
    schemes, r, g = _imports()
    ed = np.dtype("u%d" % esize)
    n = 300000
    rr = cases.rng(esize * 100 + osize + isize)
    target = rr.integers(0, 255, size=n).astype(ed)
    npatch = {1: 200, 2: 5000}.get(isize, 20000)
    period = {1: 64, 2: 1024}.get(isize, 1024)
    p = cases.patch_case(n, npatch, ed, osize, seed=esize + osize)
    expect = oracle.scatter(target.copy(), p["data"], p["indices"])
    for maxrun in (None, 5, 1):
        c = formats.compressed_indices(p["indices"], p["data"], period, osize, isize, max_run_length=maxrun)
        t = g.dev(target)
        schemes.compressed_indices_scatter(
            t, g.dev(c["data_to_scatter"]), g.dev(c["scatter_position_run_lengths"]),
            g.dev(c["scatter_position_run_individual_offset_sizes"]), g.dev(c["scatter_position_run_baseline_values"]),
            g.dev(c["scatter_position_run_offsets_start_pos"]), g.dev(c["scatter_position_offset_bytes"]),
            g.dev(c["scatter_position_anchors"]), period, c["num_scatter_position_runs"], npatch,
            output_index_size=osize, element_size=esize, input_index_size=isize, stream=g.stream())
        got = g.host(t, ed, n)
        assert (got == expect).all(), (maxrun, np.flatnonzero(got != expect)[:5])

"""-m gpu: Delta with segments that are not whole rows, at output bases on every kind of 16-byte boundary.

The decoders cut such segments into rows that follow the OUTPUT's 1 KiB boundaries (a short first row, then whole cache
lines: kernels/decompression/delta.hpp, wave_per_ragged_segment_in_aligned_rows / ..._with_staged_input / block_per_segment),
so where a segment's rows fall depends on the address the column is decoded to — every placement has to give the bytes the
oracle gives (reference: src/kernels/decompression/delta.cuh:42-91, any segment length)."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


TUPLES = [(8, 2), (8, 1), (8, 4), (4, 1), (4, 2), (2, 1)]
# a row is 1 KiB of output: segments under a row, of a few rows (wave paths, staged or not), of 33..127 rows (the workgroup path)
SEGMENTS = [8, 16, 24, 40, 64, 120, 128, 136, 248, 504, 1000, 3000, 3504, 5000, 10008, 12344]   # (under a row: several segments share a row)


@pytest.mark.parametrize("usize,csize", TUPLES)
@pytest.mark.parametrize("seglen", SEGMENTS)
def test_ragged_segments_at_shifted_outputs(usize, csize, seglen):
    import gpu_util as g
    from libgiddy_amd import schemes
    ud, cd = np.dtype("u%d" % usize), np.dtype("u%d" % csize)
    st = g.stream()
    for n in (seglen * 37 + 24, 300007 // 8 * 8, 7):
        f, truth = cases.delta_case(n, seglen, ud, cd, seed=n + seglen)
        d_in, d_anchor = g.dev(f["compressed_input"]), g.dev(f["anchor_values"])
        for shift_bytes in (0, 16, 496, 1008):
            lead = shift_bytes // usize
            out = g.dev_empty(n + lead + 64, ud)
            schemes.delta(out.data_ptr() + shift_bytes, d_in, d_anchor, n, seglen, uncompressed_size=usize, compressed_size=csize, stream=st)
            got = g.host(out, ud, n + lead + 64)
            guard = np.frombuffer(bytes([0xCD] * usize), dtype=ud)[0]
            assert (got[:lead] == guard).all() and (got[lead + n:] == guard).all(), "wrote outside the output"
            bad = np.flatnonzero(got[lead:lead + n] != truth)
            assert len(bad) == 0, (n, seglen, shift_bytes, bad[:5])


@pytest.mark.parametrize("usize,csize", [(8, 2), (4, 1), (8, 1)])
@pytest.mark.parametrize("seglen", [8, 40, 64, 120, 1000, 3000, 5000, 10008])
def test_ragged_segments_with_patches_at_shifted_outputs(usize, csize, seglen):
    import gpu_util as g
    from libgiddy_amd import schemes
    ud, cd = np.dtype("u%d" % usize), np.dtype("u%d" % csize)
    st = g.stream()
    n = seglen * 29 + 8
    f, truth = cases.delta_case(n, seglen, ud, cd, seed=n)
    rnd = cases.rng(seglen + usize)
    positions = np.unique(np.concatenate([rnd.integers(0, n, 900), np.arange(0, n, seglen), np.arange(seglen - 1, n, seglen), [n - 1]])).astype(np.uint32)
    values = rnd.integers(0, np.iinfo(ud).max, len(positions), dtype=ud)
    expect = truth.copy()
    expect[positions.astype(np.int64)] = values
    d_in, d_anchor, d_values, d_positions = g.dev(f["compressed_input"]), g.dev(f["anchor_values"]), g.dev(values), g.dev(positions)
    for shift_bytes in (0, 48, 1008):
        lead = shift_bytes // usize
        out = g.dev_empty(n + lead + 64, ud)
        schemes.patched_delta(out.data_ptr() + shift_bytes, d_in, d_anchor, d_values, d_positions, n, seglen, len(positions),
                              uncompressed_size=usize, compressed_size=csize, stream=st)
        got = g.host(out, ud, n + lead + 64)
        bad = np.flatnonzero(got[lead:lead + n] != expect)
        assert len(bad) == 0, (n, seglen, shift_bytes, bad[:5])


@pytest.mark.parametrize("usize,csize", [(8, 1), (8, 2), (8, 4)])   # (2-byte deltas stay with the workgroup path: the same test, the other way)
def test_many_long_ragged_segments_go_by_wavefront(usize, csize):
    """thousands of ragged segments of more than 32 rows of 64-bit elements: a wavefront each, the input staged in pieces
    (detail::takes_wave_path; fewer segments: a workgroup each, covered above)"""
    import gpu_util as g
    from libgiddy_amd import schemes
    ud, cd = np.dtype("u%d" % usize), np.dtype("u%d" % csize)
    st = g.stream()
    for seglen in (4232, 9000):
        n = seglen * 8200 - 6
        f, truth = cases.delta_case(n, seglen, ud, cd, seed=seglen)
        d_in, d_anchor = g.dev(f["compressed_input"]), g.dev(f["anchor_values"])
        for shift_bytes in (0, 496):
            lead = shift_bytes // usize
            out = g.dev_empty(n + lead + 64, ud)
            schemes.delta(out.data_ptr() + shift_bytes, d_in, d_anchor, n, seglen, uncompressed_size=usize, compressed_size=csize, stream=st)
            got = g.host(out, ud, n + lead + 64)
            guard = np.frombuffer(bytes([0xCD] * usize), dtype=ud)[0]
            assert (got[:lead] == guard).all() and (got[lead + n:] == guard).all(), "wrote outside the output"
            bad = np.flatnonzero(got[lead:lead + n] != truth)
            assert len(bad) == 0, (n, seglen, shift_bytes, bad[:5])
            del out, got


# segment lengths that are NOT multiples of a lane's 16 bytes of output: elements of one lane in different segments (wave_windows_of_any_segments) —
# 1 (every element its own anchor), shorter than a lane's group, around a row, up to 32 rows; beyond that the element-wise path
ODD_SEGMENTS = [1, 2, 3, 5, 6, 10, 63, 100, 255, 257, 1001, 2002, 4094, 4097, 8190, 8191, 16383]


@pytest.mark.parametrize("usize,csize", TUPLES)
@pytest.mark.parametrize("seglen", ODD_SEGMENTS)
def test_segment_lengths_that_split_a_lanes_group(usize, csize, seglen):
    import gpu_util as g
    from libgiddy_amd import schemes
    ud, cd = np.dtype("u%d" % usize), np.dtype("u%d" % csize)
    st = g.stream()
    for n in (max(seglen * 21 + 5, 70001), 333337, 9):
        f, truth = cases.delta_case(n, seglen, ud, cd, seed=n + seglen)
        d_in, d_anchor = g.dev(f["compressed_input"]), g.dev(f["anchor_values"])
        for shift_bytes in (0, 496):
            lead = shift_bytes // usize
            out = g.dev_empty(n + lead + 64, ud)
            schemes.delta(out.data_ptr() + shift_bytes, d_in, d_anchor, n, seglen, uncompressed_size=usize, compressed_size=csize, stream=st)
            got = g.host(out, ud, n + lead + 64)
            guard = np.frombuffer(bytes([0xCD] * usize), dtype=ud)[0]
            assert (got[:lead] == guard).all() and (got[lead + n:] == guard).all(), "wrote outside the output"
            bad = np.flatnonzero(got[lead:lead + n] != truth)
            assert len(bad) == 0, (n, seglen, shift_bytes, bad[:5])


@pytest.mark.parametrize("usize,csize", [(4, 1), (8, 2), (2, 1)])
@pytest.mark.parametrize("seglen", [3, 10, 1001, 4094])
def test_odd_segment_lengths_with_patches(usize, csize, seglen):
    import gpu_util as g
    from libgiddy_amd import schemes
    ud, cd = np.dtype("u%d" % usize), np.dtype("u%d" % csize)
    n = 200003
    f, truth = cases.delta_case(n, seglen, ud, cd, seed=n + seglen)
    rnd = cases.rng(seglen + usize)
    positions = np.unique(np.concatenate([rnd.integers(0, n, 1500), np.arange(0, n, max(seglen, 97))[:2000], [n - 1]])).astype(np.uint32)
    values = rnd.integers(0, np.iinfo(ud).max, len(positions), dtype=ud)
    expect = truth.copy()
    expect[positions.astype(np.int64)] = values
    out = g.dev_empty(n, ud)
    schemes.patched_delta(out, g.dev(f["compressed_input"]), g.dev(f["anchor_values"]), g.dev(values), g.dev(positions), n, seglen, len(positions),
                          uncompressed_size=usize, compressed_size=csize, stream=g.stream())
    got = g.host(out, ud, n)
    bad = np.flatnonzero(got != expect)
    assert len(bad) == 0, (n, seglen, bad[:5])


@pytest.mark.parametrize("block,grid", [(64, 1), (64, 5), (256, 3), (512, 2), (1024, 1), (1024, 9)])
def test_new_paths_under_any_launch_configuration(block, grid):
    """the kernels take any grid and any workgroup size (modes B and C hand them their own): short segments sharing rows, ragged ones, lengths that split a
    lane's group — whose LDS tables exist for 4 wavefronts: larger workgroups take the element-wise path —, many long ragged segments, with and without patches"""
    import gpu_util as g
    from libgiddy_amd import schemes, registered as r
    st = g.stream()
    config = r.launch_configuration_t((grid, 1, 1), (block, 1, 1), 0)
    for usize, csize in ((4, 1), (8, 2), (2, 1)):
        ud, cd = np.dtype("u%d" % usize), np.dtype("u%d" % csize)
        for seglen in (16, 48, 100, 1000, 1001, 3000, 4094):
            n = 150000 + seglen
            f, truth = cases.delta_case(n, seglen, ud, cd, seed=seglen + usize)
            out = g.dev_empty(n, ud)
            schemes.delta(out, g.dev(f["compressed_input"]), g.dev(f["anchor_values"]), n, seglen, uncompressed_size=usize, compressed_size=csize, stream=st, launch_config=config)
            got = g.host(out, ud, n)
            bad = np.flatnonzero(got != truth)
            assert len(bad) == 0, (usize, csize, seglen, bad[:5])
            positions = np.unique(cases.rng(seglen).integers(0, n, 3000)).astype(np.uint32)
            values = cases.rng(seglen + 1).integers(0, np.iinfo(ud).max, len(positions), dtype=ud)
            expect = truth.copy(); expect[positions.astype(np.int64)] = values
            out = g.dev_empty(n, ud)
            patched_config = r.launch_configuration_t((grid, 1, 1), (min(block, 256), 1, 1), 0)
            schemes.patched_delta(out, g.dev(f["compressed_input"]), g.dev(f["anchor_values"]), g.dev(values), g.dev(positions), n, seglen, len(positions),
                                  uncompressed_size=usize, compressed_size=csize, stream=st, launch_config=patched_config)
            got = g.host(out, ud, n)
            bad = np.flatnonzero(got != expect)
            assert len(bad) == 0, ("patched", usize, csize, seglen, bad[:5])

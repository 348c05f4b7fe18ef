"""-m gpu: every kernel decodes correctly under one-time launch limits (tiny grids / single-wavefront
workgroups force the grid-stride and wave-stride paths) — "any launch configuration is correct"."""
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


LIMITS = [dict(block_size=64), dict(grid_size=3), dict(block_size=64, grid_size=1), dict(block_size=128, grid_size=7)]


@pytest.mark.parametrize("limits", LIMITS)
def test_all_schemes_under_limits(limits):
    import gpu_util as g
    from libgiddy_amd import schemes, registered as r
    lim = r.launch_configuration_limits_t(**limits)
    st = g.stream()
    n = 200000

    def check(cfg):
        if "block_size" in limits:
            assert cfg.block_dimensions[0] <= limits["block_size"]
        if "grid_size" in limits:
            assert cfg.grid_dimensions[0] <= limits["grid_size"]

    # DZB fixed
    c = cases.rng(1).integers(0, 65535, size=n, endpoint=True).astype(np.uint16)
    out = g.dev_empty(n, np.uint64)
    check(schemes.discard_zero_bytes_fixed(out, g.dev(c), n, uncompressed_size=8, compressed_size=2, stream=st, limits=lim))
    assert (g.host(out, np.uint64, n) == c).all()
    # Model
    out = g.dev_empty(n, np.int32)
    check(schemes.model(out, np.array([5, 3], np.int32), n, stream=st, limits=lim))
    assert (g.host(out, np.int32, n) == (5 + 3 * np.arange(n)).astype(np.int32)).all()
    # Dictionary
    f = cases.dictionary_case(n, 256, np.uint32, np.uint8, seed=2)
    out = g.dev_empty(n, np.uint32)
    check(schemes.dictionary(out, g.dev(f["compressed_input"]), g.dev(f["dictionary_entries"]), n, 256, stream=st, limits=lim))
    assert (g.host(out, np.uint32, n) == f["dictionary_entries"][f["compressed_input"].astype(np.int64)]).all()
    # DZB variable
    col, sizes = cases.dzbv_column(n, np.uint64, 1, 2, seed=3)
    f = formats.discard_zero_bytes_variable(col, 1024, 1, 2, np.uint32, 8, element_sizes=sizes)
    out = g.dev_empty(n, np.uint64)
    check(schemes.discard_zero_bytes_variable(out, g.dev(f["compressed_data"]), g.dev(f["packed_element_sizes"]), g.dev(f["position_anchors"]),
                                              1024, n, 1, 2, stream=st, limits=lim))
    assert (g.host(out, np.uint64, n) == col).all()
    # RLE / RPE
    col = cases.runs_column(n, 32, np.uint32, seed=4)
    f = formats.run_length_encoding(col, 4096, np.uint16)
    out = g.dev_empty(n, np.uint32)
    check(schemes.run_length_encoding(out, g.dev(f["run_data"]), g.dev(f["run_lengths"]), g.dev(f["position_anchors"]),
                                      g.dev(f["intra_run_anchor_offsets"]), 4096, f["num_segments"], f["num_element_runs"], n, stream=st, limits=lim))
    assert (g.host(out, np.uint32, n) == col).all()
    f = formats.run_position_encoding(col, 4096, np.uint16, True)
    out = g.dev_empty(n, np.uint32)
    check(schemes.run_position_encoding(out, g.dev(f["run_data"]), g.dev(f["run_start_positions"]), g.dev(f["position_anchors"]), 4096,
                                        f["num_anchors"], f["num_element_runs"], n, stream=st, limits=lim))
    assert (g.host(out, np.uint32, n) == col).all()
    # Incidence bitmaps
    pool = cases.rng(5).choice(1 << 31, size=9, replace=False).astype(np.uint32)
    col = pool[cases.rng(6).integers(0, 9, size=32 * 5000)]
    f = formats.incidence_bitmaps(col, 2)
    out = g.dev_empty(len(col), np.uint32)
    check(schemes.incidence_bitmaps(out, g.dev(f["incidence_bitmaps"]), g.dev(f["dictionary_entries"]), f["bitmap_length"], f["num_bitmaps"], stream=st, limits=lim))
    assert (g.host(out, np.uint32, len(col)) == col).all()
    # scatter + compressed-indices scatter
    target = cases.rng(7).integers(0, 255, size=n).astype(np.uint32)
    p = cases.patch_case(n, 5000, np.uint32, 4, seed=8)
    expect = target.copy(); expect[p["indices"].astype(np.int64)] = p["data"]
    t = g.dev(target)
    check(schemes.scatter(t, g.dev(p["data"]), g.dev(p["indices"]), 5000, element_size=4, stream=st, limits=lim))
    assert (g.host(t, np.uint32, n) == expect).all()
    c = formats.compressed_indices(p["indices"], p["data"], 1024, 4, 4)
    t = g.dev(target)
    check(schemes.compressed_indices_scatter(
        t, g.dev(c["data_to_scatter"]), g.dev(c["scatter_position_run_lengths"]), g.dev(c["scatter_position_run_individual_offset_sizes"]),
        g.dev(c["scatter_position_run_baseline_values"]), g.dev(c["scatter_position_run_offsets_start_pos"]),
        g.dev(c["scatter_position_offset_bytes"]), g.dev(c["scatter_position_anchors"]), 1024, c["num_scatter_position_runs"], 5000,
        stream=st, limits=lim))
    assert (g.host(t, np.uint32, n) == expect).all()


@pytest.mark.parametrize("limits", LIMITS)
def test_fused_wrappers_two_launches_under_limits(limits):
    """Where a fused decompress-and-patch wrapper launches the plain decoder + scatter_ascending instead of its fused kernel (2-byte outputs at 1 % of patches:
    kernel_wrappers/decompression/patched/decode_then_scatter.hpp) a caller's block-size limit carries over to both launches, and the result is the fused one's."""
    import gpu_util as g
    from libgiddy_amd import schemes, registered as r
    from oracle import oracle
    lim = r.launch_configuration_limits_t(**limits)
    st = g.stream()
    n = 300001
    rng = cases.rng(77)
    positions = np.sort(rng.choice(n, size=n // 100, replace=False)).astype(np.uint32)
    # Delta <4,2,1>
    f, truth = cases.delta_case(n, 4096, np.dtype("u2"), np.dtype("u1"), seed=5)
    values = rng.integers(0, 65535, size=len(positions), endpoint=True).astype(np.uint16)
    expect = oracle.scatter(truth.copy(), values, positions, threads=1)
    out = g.dev_empty(n, np.uint16)
    cfg = schemes.patched_delta(out, g.dev(f["compressed_input"]), g.dev(f["anchor_values"]), g.dev(values), g.dev(positions), n, 4096, len(positions),
                                uncompressed_size=2, compressed_size=1, index_size=4, stream=st, limits=lim)
    if "block_size" in limits:
        assert cfg.block_dimensions[0] <= limits["block_size"]
    assert (g.host(out, np.uint16, n) == expect).all()
    # Model <4,short,linear> with 0.1 % of patches (two launches below 0.4 %)
    few = np.ascontiguousarray(positions[::10])
    signed_values = rng.integers(-30000, 30000, size=len(few)).astype(np.int16)
    coefficients = np.array([-12345, 37], dtype=np.int16)
    decoded = oracle.model(coefficients, n, np.dtype("i2"), "linear", model_index_size=4)
    expect = oracle.scatter(decoded.copy(), signed_values, few, threads=1)
    out = g.dev_empty(n, np.int16)
    schemes.patched_model(out, coefficients, g.dev(signed_values), g.dev(few), n, len(few), uncompressed_size=2, model="linear", index_size=4, stream=st, limits=lim)
    assert (g.host(out, np.int16, n) == expect).all()

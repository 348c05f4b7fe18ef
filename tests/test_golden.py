"""Committed golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from the
independent numpy statement) decoded by (a) the C oracle — runs everywhere — and (b) the HIP kernels
through the C ABI — `-m gpu`."""
import glob
import os

import numpy as np
import pytest

from oracle import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def test_fixtures_present():
    assert len(NAMES) >= 20 and "kat_for_1" in NAMES


def _decode_oracle(name, f):
    i = lambda k: int(f[k])
    if name.startswith(("kat_for", "for_")):
        model = "linear" if f["segment_model_coefficients"].shape[1] == 2 else "constant"
        return oracle.frame_of_reference(f["compressed_input"], f["segment_model_coefficients"], i("length"), i("segment_length"), f["expected"].dtype, model)
    if name.startswith("model_"):
        return oracle.model(f["model_coefficients"], i("length"), f["expected"].dtype, "linear")
    if name.startswith("dictionary_"):
        return oracle.dictionary(f["compressed_input"], f["dictionary_entries"])
    if name.startswith("delta_"):
        return oracle.delta(f["compressed_input"], f["anchor_values"], i("segment_length"))
    if name.startswith("dzb_fixed"):
        return oracle.dzb_fixed(f["compressed_input"], f["expected"].dtype)
    if name.startswith("dzb_variable"):
        return oracle.dzb_variable(f["compressed_data"], f["packed_element_sizes"], f["position_anchors"], i("segment_length"), i("length"),
                                   i("min_represented_element_size"), i("bits_per_element_size"), f["expected"].dtype)
    if name.startswith("incidence_bitmaps"):
        return oracle.incidence_bitmaps(f["incidence_bitmaps"], f["dictionary_entries"], i("bitmap_length"), i("num_bitmaps"), 2, f["expected"].dtype)[0]
    if name.startswith("rle_"):
        return oracle.run_length_encoding(f["run_data"], f["run_lengths"], f["position_anchors"], f["intra_run_anchor_offsets"], i("segment_length"), i("length"))
    if name.startswith("rpe_"):
        return oracle.run_position_encoding(f["run_data"], f["run_start_positions"], f["position_anchors"], i("segment_length"), i("length"), name.startswith("rpe_rel"))
    raise KeyError(name)


@pytest.mark.parametrize("name", [n for n in NAMES if not n.startswith("compressed_indices")])
def test_oracle_matches_golden(name):
    f = load(name)
    assert (_decode_oracle(name, f) == f["expected"]).all()
    if "expected_patched" in f:
        dec = _decode_oracle(name, f)
        assert (oracle.scatter(dec, f["patch_data"], f["patch_indices"]) == f["expected_patched"]).all()


def _cis_target(f):
    import cases
    return cases.rng(int(f["target_seed"])).integers(0, 1 << 30, size=int(f["target_length"])).astype(np.uint32)


def test_oracle_matches_golden_compressed_indices():
    f = load("compressed_indices_scatter_u32")
    got = oracle.compressed_indices_scatter(
        _cis_target(f), f["data_to_scatter"], f["scatter_position_run_lengths"], f["scatter_position_run_individual_offset_sizes"],
        f["scatter_position_run_baseline_values"], f["scatter_position_run_offsets_start_pos"], f["scatter_position_offset_bytes"],
        f["scatter_position_anchors"], int(f["anchoring_period"]))
    assert (got[f["patch_indices"].astype(np.int64)] == f["expected_at_patches"]).all()


def _decode_hip(name, f):
    import gpu_util as g
    from libgiddy_amd import schemes
    i = lambda k: int(f[k])
    d = lambda k: g.dev(f[k])
    ex = f["expected"]
    out = g.dev_empty(len(ex), ex.dtype)
    us = ex.dtype.itemsize
    st = g.stream()
    if name.startswith(("kat_for", "for_")):
        model = "linear" if f["segment_model_coefficients"].shape[1] == 2 else "constant"
        schemes.frame_of_reference(out, d("compressed_input"), d("segment_model_coefficients"), i("length"), i("segment_length"),
                                   uncompressed_size=us, compressed_size=f["compressed_input"].dtype.itemsize, model=model, stream=st)
    elif name.startswith("model_"):
        schemes.model(out, f["model_coefficients"], i("length"), uncompressed_size=us, model="linear", stream=st)
    elif name.startswith("dictionary_"):
        schemes.dictionary(out, d("compressed_input"), d("dictionary_entries"), i("length"), i("num_dictionary_entries"),
                           uncompressed_size=us, dictionary_index_size=f["compressed_input"].dtype.itemsize, stream=st)
    elif name.startswith("delta_"):
        schemes.delta(out, d("compressed_input"), d("anchor_values"), i("length"), i("segment_length"), uncompressed_size=us,
                      compressed_size=f["compressed_input"].dtype.itemsize, stream=st)
        if "expected_patched" in f:
            first = g.host(out, ex.dtype, len(ex))
            schemes.scatter(out, d("patch_data"), d("patch_indices"), len(f["patch_data"]), element_size=us, stream=st)
            assert (g.host(out, ex.dtype, len(ex)) == f["expected_patched"]).all()
            return first
    elif name.startswith("dzb_fixed"):
        schemes.discard_zero_bytes_fixed(out, d("compressed_input"), i("length"), uncompressed_size=us,
                                         compressed_size=f["compressed_input"].dtype.itemsize, stream=st)
    elif name.startswith("dzb_variable"):
        schemes.discard_zero_bytes_variable(out, d("compressed_data"), d("packed_element_sizes"), d("position_anchors"), i("segment_length"),
                                            i("length"), i("min_represented_element_size"), i("bits_per_element_size"), uncompressed_size=us,
                                            container_size=f["packed_element_sizes"].dtype.itemsize,
                                            index_size=f["position_anchors"].dtype.itemsize, stream=st)
    elif name.startswith("incidence_bitmaps"):
        schemes.incidence_bitmaps(out, d("incidence_bitmaps"), d("dictionary_entries"), i("bitmap_length"), i("num_bitmaps"), uncompressed_size=us, stream=st)
    elif name.startswith("rle_"):
        schemes.run_length_encoding(out, d("run_data"), d("run_lengths"), d("position_anchors"), d("intra_run_anchor_offsets"), i("segment_length"),
                                    i("num_segments"), i("num_element_runs"), i("length"), uncompressed_size=us,
                                    run_length_size=f["run_lengths"].dtype.itemsize, stream=st)
    elif name.startswith("rpe_"):
        schemes.run_position_encoding(out, d("run_data"), d("run_start_positions"), d("position_anchors"), i("segment_length"), i("num_anchors"),
                                      i("num_element_runs"), i("length"), uncompressed_size=us, position_size=f["run_start_positions"].dtype.itemsize,
                                      relative=name.startswith("rpe_rel"), stream=st)
    else:
        raise KeyError(name)
    return g.host(out, ex.dtype, len(ex))


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in NAMES if not n.startswith("compressed_indices")])
def test_hip_matches_golden(name):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    f = load(name)
    assert (_decode_hip(name, f) == f["expected"]).all()


@pytest.mark.gpu
def test_hip_matches_golden_compressed_indices():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import gpu_util as g
    from libgiddy_amd import schemes
    f = load("compressed_indices_scatter_u32")
    t = g.dev(_cis_target(f))
    schemes.compressed_indices_scatter(
        t, g.dev(f["data_to_scatter"]), g.dev(f["scatter_position_run_lengths"]), g.dev(f["scatter_position_run_individual_offset_sizes"]),
        g.dev(f["scatter_position_run_baseline_values"]), g.dev(f["scatter_position_run_offsets_start_pos"]),
        g.dev(f["scatter_position_offset_bytes"]), g.dev(f["scatter_position_anchors"]), int(f["anchoring_period"]),
        int(f["num_scatter_position_runs"]), int(f["input_data_length"]), stream=g.stream())
    got = g.host(t, np.uint32, int(f["target_length"]))
    assert (got[f["patch_indices"].astype(np.int64)] == f["expected_at_patches"]).all()

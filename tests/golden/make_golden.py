"""Generates the committed golden fixtures (inputs + expected outputs, a few KiB each).

Expected outputs come from oracle/naive.py — the independent numpy statement — NOT from the C
oracle and NOT from the HIP kernels, so both of those are checked against these files.
KAT-FOR-1 is the reference's own known-answer test restated as data
(examples/src/modes_of_use/common.cuh:31-49 inputs, :91-102 expected values).

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import cases  # noqa: E402
from libgiddy_amd import formats  # noqa: E402
from oracle import naive  # noqa: E402


def save(name, **arrays):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    print(name, sum(np.asarray(a).nbytes for a in arrays.values()), "bytes raw")


def main():
    # ---- KAT-FOR-1 -----------------------------------------------------------------------
    length, seglen = 2 << 12, 2 << 10
    nseg = length // seglen
    comp = np.tile(np.arange(seglen, dtype=np.int16), nseg)
    coefs = (np.arange(nseg, dtype=np.int32) * 100).reshape(nseg, 1)
    expected = (np.repeat(np.arange(nseg) * 100, seglen) + np.tile(np.arange(seglen), nseg)).astype(np.int32)
    save("kat_for_1", compressed_input=comp, segment_model_coefficients=coefs, length=length,
         segment_length=seglen, expected=expected)

    # ---- FOR, linear model, ragged tail, int16<-int8 --------------------------------------
    f = cases.for_case(3000, 1024, np.int16, np.int8, "linear", seed=21)
    save("for_linear_i16_i8", **f, expected=naive.frame_of_reference(
        f["compressed_input"], f["segment_model_coefficients"], 3000, 1024, np.int16, "linear"))
    f = cases.for_case(2500, 500, np.int32, np.int16, "linear", seed=22)
    save("for_linear_i32_i16", **f, expected=naive.frame_of_reference(
        f["compressed_input"], f["segment_model_coefficients"], 2500, 500, np.int32, "linear"))

    # ---- Model ----------------------------------------------------------------------------
    c = np.array([-1000, 37], dtype=np.int32)
    save("model_linear_i32", model_coefficients=c, length=5000, expected=naive.model(c, 5000, np.int32, "linear"))
    c = np.array([30000, 123], dtype=np.int16)
    save("model_linear_i16_wrap", model_coefficients=c, length=3000, expected=naive.model(c, 3000, np.int16, "linear"))

    # ---- Dictionary -----------------------------------------------------------------------
    f = cases.dictionary_case(4099, 256, np.uint32, np.uint8, seed=23)
    save("dictionary_u32_u8", **f, expected=naive.dictionary(f["compressed_input"], f["dictionary_entries"]))
    f = cases.dictionary_case(1027, 1000, np.uint8, np.uint8, seed=24)
    f["dictionary_entries"] = f["dictionary_entries"][:256]; f["num_dictionary_entries"] = 256
    save("dictionary_u8_u8", **f, expected=naive.dictionary(f["compressed_input"], f["dictionary_entries"]))
    f = cases.dictionary_case(2050, 3000, np.uint64, np.uint16, seed=25)
    save("dictionary_u64_u16", **f, expected=naive.dictionary(f["compressed_input"], f["dictionary_entries"]))

    # ---- Delta (+ patches) ------------------------------------------------------------------
    f, _ = cases.delta_case(5000, 1024, np.uint64, np.uint16, seed=26)
    dec = naive.delta(f["compressed_input"], f["anchor_values"], 1024)
    p = cases.patch_case(5000, 7, np.uint64, 4, seed=27)
    save("delta_u64_u16_patched", **f, expected=dec, patch_indices=p["indices"], patch_data=p["data"],
         expected_patched=naive.scatter(dec, p["data"], p["indices"]))
    f, _ = cases.delta_case(777, 100, np.uint32, np.uint8, seed=28, base=(1 << 32) - 5000)  # wraps
    save("delta_u32_u8_wrap", **f, expected=naive.delta(f["compressed_input"], f["anchor_values"], 100))
    f, _ = cases.delta_case(300, 3, np.uint64, np.uint32, seed=29)  # tiny segments
    save("delta_u64_u32_tinyseg", **f, expected=naive.delta(f["compressed_input"], f["anchor_values"], 3))

    # ---- DZB fixed / variable -----------------------------------------------------------------
    c = cases.rng(30).integers(0, 65535, size=1500, endpoint=True).astype(np.uint16)
    save("dzb_fixed_u64_u16", compressed_input=c, length=1500, expected=naive.dzb_fixed(c, np.uint64))
    col, sizes = cases.dzbv_column(3001, np.uint64, 1, 2, seed=31)
    f = formats.discard_zero_bytes_variable(col, 1024, 1, 2, np.uint32, 8, element_sizes=sizes)
    save("dzb_variable_u64_2bit", **f, expected=naive.dzb_variable(
        f["compressed_data"], f["packed_element_sizes"], f["position_anchors"], 1024, 3001, 1, 2, np.uint64))
    col, sizes = cases.dzbv_column(2000, np.uint64, 1, 3, seed=32)  # 10 sizes per container, 2 slack bits
    f = formats.discard_zero_bytes_variable(col, 256, 1, 3, np.uint32, 4, element_sizes=sizes)
    save("dzb_variable_u64_3bit_slack", **f, expected=naive.dzb_variable(
        f["compressed_data"], f["packed_element_sizes"], f["position_anchors"], 256, 2000, 1, 3, np.uint64))
    col, sizes = cases.dzbv_column(1000, np.uint32, 0, 2, seed=33)  # zero-byte elements allowed
    f = formats.discard_zero_bytes_variable(col, 128, 0, 2, np.uint64, 4, element_sizes=sizes)
    save("dzb_variable_u32_min0_c64", **f, expected=naive.dzb_variable(
        f["compressed_data"], f["packed_element_sizes"], f["position_anchors"], 128, 1000, 0, 2, np.uint32))

    # ---- Incidence bitmaps ----------------------------------------------------------------------
    r = cases.rng(34)
    pool = r.choice(1 << 31, size=11, replace=False).astype(np.uint32)
    col = pool[r.integers(0, 11, size=32 * 37)]
    f = formats.incidence_bitmaps(col, 2)
    save("incidence_bitmaps_u32", **f, expected=naive.incidence_bitmaps(
        f["incidence_bitmaps"], f["dictionary_entries"], f["bitmap_length"], f["num_bitmaps"], 2, np.uint32))

    # ---- RLE / RPE: boundary-heavy column ----------------------------------------------------------
    col = np.repeat(np.arange(10, dtype=np.uint32) * 3 + 1, [1024, 512, 512, 2048, 1, 1023, 1024, 100, 924, 1000])
    f = formats.run_length_encoding(col, 1024, np.uint16)
    save("rle_u32_u16_boundaries", **f, expected=naive.run_length_encoding(f["run_data"], f["run_lengths"], len(col)))
    f = formats.run_position_encoding(col, 1024, np.uint16, True)
    save("rpe_rel_u32_u16_boundaries", **f, expected=naive.run_position_encoding(
        f["run_data"], f["run_start_positions"], f["position_anchors"], 1024, len(col), True))
    col = cases.runs_column(6000, 5, np.uint32, seed=35)
    f = formats.run_length_encoding(col, 512, np.uint8)
    save("rle_u32_u8_short_runs", **f, expected=naive.run_length_encoding(f["run_data"], f["run_lengths"], len(col)))
    f = formats.run_position_encoding(col, 512, np.uint32, False)
    save("rpe_abs_u32_u32_short_runs", **f, expected=naive.run_position_encoding(
        f["run_data"], f["run_start_positions"], f["position_anchors"], 512, len(col), False))

    # ---- compressed-indices scatter --------------------------------------------------------------------
    r = cases.rng(36)
    target = r.integers(0, 1 << 30, size=300000).astype(np.uint32)
    p = cases.patch_case(300000, 700, np.uint32, 4, seed=37)
    c = formats.compressed_indices(p["indices"], p["data"], 256, 4, 4)
    exp_idx = p["indices"]
    save("compressed_indices_scatter_u32", target_length=300000, target_seed=36, **c, patch_indices=exp_idx,
         expected_at_patches=naive.compressed_indices_scatter(
             target, c["data_to_scatter"], c["scatter_position_run_lengths"],
             c["scatter_position_run_individual_offset_sizes"], c["scatter_position_run_baseline_values"],
             c["scatter_position_run_offsets_start_pos"], c["scatter_position_offset_bytes"],
             c["scatter_position_anchors"], 256, 4)[exp_idx.astype(np.int64)])


if __name__ == "__main__":
    main()

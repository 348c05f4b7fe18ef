"""The C++ cuts (include/giddy/sharding.hpp, through the C ABI's giddy_shard) against libgiddy_amd/sharding.py: the same slices and the
same biases for every scheme at world sizes 1, 2, 3 and 8 — no GPU needed, the anchors are host arrays here."""
import ctypes

import numpy as np
import pytest

import cases
from libgiddy_amd import _native as native, formats, sharding


def _shape(**kw):
    shape = native.giddy_column_shape()
    anchors = kw.pop("anchors", None)
    for k, v in kw.items():
        setattr(shape, k, v)
    if anchors is not None:
        anchors = np.ascontiguousarray(anchors)
        shape.position_anchors = anchors.ctypes.data
        shape.anchor_itemsize = anchors.dtype.itemsize
        shape.num_anchors = len(anchors)
        shape._keep = anchors
    return shape


def _cut(scheme, shape, rank, world):
    out = native.giddy_shard()
    status = native.lib.giddy_shard(native.SCHEMES[scheme], ctypes.byref(shape), rank, world, ctypes.byref(out))
    assert status == 0, native.lib.giddy_last_error()
    return out


WORLDS = [1, 2, 3, 8]


@pytest.mark.parametrize("world", WORLDS)
def test_cuts_equal_the_python_ones(world):
    n = 300000 + 77
    # elementwise
    f = cases.dictionary_case(n, 256, np.uint32, np.uint8, seed=1)
    for rank in range(world):
        ref = sharding.elementwise(dict(f, length=n), rank, world)
        got = _cut("elementwise", _shape(length=n, output_itemsize=4), rank, world)
        assert (got.first_element, got.end_element) == (ref.start, ref.stop)
    # delta and frame of reference, several segment lengths (page-sized granules and not)
    for L in (4096, 1000, 512, 70000):
        fd, _ = cases.delta_case(n, L, np.uint64, np.uint16, seed=2)
        for rank in range(world):
            ref = sharding.delta(fd, rank, world)
            got = _cut("delta", _shape(length=n, segment_length=L, output_itemsize=8), rank, world)
            assert (got.first_element, got.end_element) == (ref.start, ref.stop)
            assert got.end_segment - got.first_segment == len(ref.fields["anchor_values"])
            ref = sharding.frame_of_reference(dict(fd, segment_model_coefficients=fd["anchor_values"]), rank, world, output_itemsize=4)
            got = _cut("frame_of_reference", _shape(length=n, segment_length=L, output_itemsize=4), rank, world)
            assert (got.first_element, got.end_element) == (ref.start, ref.stop)
    # variable-width discard-zero-bytes: 2- and 3-bit fields, 4- and 8-byte containers
    for bits, container in ((2, 4), (3, 4), (2, 8)):
        col, sizes = cases.dzbv_column(n, np.uint64, 1, bits, seed=3)
        fv = formats.discard_zero_bytes_variable(col, 1024, 1, bits, "u%d" % container, 8, element_sizes=sizes)
        shape = _shape(length=n, segment_length=1024, output_itemsize=8, bits_per_element_size=bits, size_container_size=container,
                       num_units=len(fv["compressed_data"]), anchors=fv["position_anchors"])
        for rank in range(world):
            ref = sharding.discard_zero_bytes_variable(fv, rank, world)
            got = _cut("discard_zero_bytes_variable", shape, rank, world)
            assert (got.first_element, got.end_element) == (ref.start, ref.stop)
            assert got.end_unit - got.first_unit == len(ref.fields["compressed_data"])
            assert got.end_size_container - got.first_size_container == len(ref.fields["packed_element_sizes"])
            mine = fv["position_anchors"][got.first_segment:got.end_segment] - fv["position_anchors"].dtype.type(got.anchor_bias)
            assert (mine == ref.fields["position_anchors"]).all()
            assert (fv["compressed_data"][got.first_unit:got.end_unit] == ref.fields["compressed_data"]).all()
    # the run encodings
    col = cases.runs_column(n, 16, np.uint32, seed=4)
    for L in (4096, 1000):
        fr = formats.run_length_encoding(col, L, np.uint16)
        shape = _shape(length=n, segment_length=L, output_itemsize=4, num_units=fr["num_element_runs"], anchors=fr["position_anchors"])
        for rank in range(world):
            ref = sharding.run_length_encoding(fr, rank, world)
            got = _cut("run_length_encoding", shape, rank, world)
            assert (got.first_element, got.end_element) == (ref.start, ref.stop)
            assert got.end_unit - got.first_unit == ref.fields["num_element_runs"] and got.end_segment - got.first_segment == ref.fields["num_segments"]
            if ref.fields["num_segments"]:
                assert (fr["position_anchors"][got.first_segment:got.end_segment] - fr["position_anchors"].dtype.type(got.anchor_bias) == ref.fields["position_anchors"]).all()
                assert (fr["run_lengths"][got.first_unit:got.end_unit] == ref.fields["run_lengths"]).all()
        for relative, pd, name in ((True, np.uint16, "run_position_encoding_relative"), (False, np.uint32, "run_position_encoding_absolute")):
            fp = formats.run_position_encoding(col, L, pd, relative)
            shape = _shape(length=n, segment_length=L, output_itemsize=4, num_units=fp["num_element_runs"], anchors=fp["position_anchors"])
            for rank in range(world):
                ref = sharding.run_position_encoding(fp, rank, world, relative=relative)
                got = _cut(name, shape, rank, world)
                assert (got.first_element, got.end_element) == (ref.start, ref.stop) and got.end_unit - got.first_unit == ref.fields["num_element_runs"]
                if ref.fields["num_anchors"]:
                    positions = fp["run_start_positions"][got.first_unit:got.end_unit] - fp["run_start_positions"].dtype.type(got.position_bias)
                    # (the first run of a shard may begin before it: its wrapped position is never read — compare from the second on)
                    assert (positions[1:] == ref.fields["run_start_positions"][1:]).all()


def test_bad_requests_are_refused():
    out = native.giddy_shard()
    shape = _shape(length=100, segment_length=0, output_itemsize=4)
    assert native.lib.giddy_shard(native.SCHEMES["delta"], ctypes.byref(shape), 0, 2, ctypes.byref(out)) != 0
    assert b"segment length" in native.lib.giddy_last_error()
    shape = _shape(length=100, segment_length=10, output_itemsize=4)
    assert native.lib.giddy_shard(native.SCHEMES["delta"], ctypes.byref(shape), 3, 2, ctypes.byref(out)) != 0
    assert native.lib.giddy_shard(99, ctypes.byref(shape), 0, 2, ctypes.byref(out)) != 0
    assert native.lib.giddy_shard(native.SCHEMES["run_length_encoding"], ctypes.byref(shape), 0, 2, ctypes.byref(out)) != 0   # no anchors given

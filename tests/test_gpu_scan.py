"""GPU parity of the un-anchored scan kernels (reduce_segments, scan_using_segment_reductions,
scan_single_segment — kernels the reference has without wrappers, so they are driven through a small C++ program,
tests/cpp/scan_tool.hip, built against the public headers) against the numpy oracle (oracle/scan.py), bit for bit:
operators, input/result type pairs and pre-transformations, inclusive and exclusive, ragged and tiny inputs, segment
lengths that are not whole rows, imposed grids smaller and larger than the number of segments, displaced pointers."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import scan as oracle_scan

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

TOOL = entry.SCAN_TOOL


@pytest.fixture(scope="module", autouse=True)
def _tool():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if not os.path.exists(TOOL):
        entry.build_scan_tool()


INSTANTIATIONS = {   # name -> (operator, input dtype, result dtype, inclusive, pre-transformation)
    "plus_u32_u32_inclusive": ("plus", np.uint32, np.uint32, True, "identity"),
    "plus_u32_u32_exclusive": ("plus", np.uint32, np.uint32, False, "identity"),
    "plus_u64_u16_inclusive": ("plus", np.uint16, np.uint64, True, "identity"),
    "plus_u64_u32_inclusive_idx8": ("plus", np.uint32, np.uint64, True, "identity"),
    "plus_u16_u16_inclusive": ("plus", np.uint16, np.uint16, True, "identity"),
    "maximum_i32_i32_inclusive": ("maximum", np.int32, np.int32, True, "identity"),
    "maximum_i32_i32_exclusive": ("maximum", np.int32, np.int32, False, "identity"),
    "minimum_i64_i64_inclusive": ("minimum", np.int64, np.int64, True, "identity"),
    "bit_or_u32_u32_inclusive": ("bit_or", np.uint32, np.uint32, True, "identity"),
    "plus_i64_absolute_i32": ("plus", np.int32, np.int64, True, "absolute"),
    "plus_f64_f64_inclusive": ("plus", np.float64, np.float64, True, "identity"),
}


def _data(name, n, seed):
    op, in_dtype, result_dtype, inclusive, pre = INSTANTIATIONS[name]
    r = np.random.default_rng(seed)
    if np.dtype(in_dtype).kind == "f":
        return r.integers(-1000, 1000, size=n).astype(in_dtype)    # sums of integers below 2^53: exact in any order
    if op == "bit_or":
        return (np.uint32(1) << r.integers(0, 32, size=n).astype(np.uint32)) * (r.random(n) < 0.01)
    if op in ("maximum", "minimum"):
        walk = np.cumsum(r.integers(-1000, 1001, size=n))             # extremes keep changing along the array
        return (walk if op == "maximum" else -walk).astype(in_dtype)
    info = np.iinfo(in_dtype)
    return r.integers(info.min, info.max, size=n, dtype=in_dtype, endpoint=True)


def _run(tmp_path, name, mode, n, segment_length, grid=0, offset=0, seed=0):
    op, in_dtype, result_dtype, inclusive, pre = INSTANTIATIONS[name]
    data = _data(name, n, seed)
    in_path, out_path = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    data.astype(in_dtype).tofile(in_path)
    done = subprocess.run([TOOL, name, mode, str(n), str(segment_length), str(grid), str(offset), in_path, out_path],
                          capture_output=True, text=True, timeout=300)
    assert done.returncode == 0 and done.stdout.startswith("OK"), done.stdout + done.stderr
    out = np.fromfile(out_path, dtype=result_dtype)
    assert len(out) == n + 64
    assert (out[n:].view(np.uint8) == 0xCD).all(), "wrote past the last result"
    expect = oracle_scan.scan(data, op, result_dtype, inclusive, pre)
    bad = np.flatnonzero(out[:n] != expect)
    assert len(bad) == 0, (name, mode, n, segment_length, grid, offset, bad[:5], out[bad[:5]], expect[bad[:5]])
    if mode == "two_phase":
        segments = np.fromfile(out_path + ".segments", dtype=result_dtype)
        assert (segments == oracle_scan.reduce_segments(data, segment_length, op, result_dtype, pre)).all()
    return float(done.stdout.split()[1])


@pytest.mark.parametrize("name", sorted(INSTANTIATIONS))
def test_two_phase_scan(tmp_path, name):
    for n, segment_length in [(1, 1), (1000, 1 << 20), (100003, 4096), (1 << 20, 1 << 16), (3000001, 50000), (70000, 7)]:
        _run(tmp_path, name, "two_phase", n, segment_length, seed=n)


@pytest.mark.parametrize("name", ["plus_u32_u32_inclusive", "plus_u64_u16_inclusive", "maximum_i32_i32_exclusive"])
def test_two_phase_scan_any_grid_and_displaced_pointers(tmp_path, name):
    n, segment_length = 1234567, 8192      # 151 segments
    for grid in (1, 3, 151, 1000):          # fewer workgroups than segments, exactly as many, more
        _run(tmp_path, name, "two_phase", n, segment_length, grid=grid, seed=grid)
    for offset in (1, 3):                   # neither input nor output 16-byte aligned: guarded element access
        _run(tmp_path, name, "two_phase", 200001, 4096, offset=offset, seed=offset)


@pytest.mark.parametrize("name", sorted(INSTANTIATIONS))
def test_single_segment_scan(tmp_path, name):
    for n in (1, 63, 64, 4097, 500000):
        _run(tmp_path, name, "single", n, 1, seed=n)
    _run(tmp_path, name, "single", 10000, 1, grid=4, seed=1)   # surplus workgroups stay idle
    _run(tmp_path, name, "single", 10001, 1, offset=1, seed=2)


@pytest.mark.parametrize("name,element,index", [("gather_4_4_2", np.uint32, np.uint16), ("gather_4_1_1", np.uint8, np.uint8),
                                                ("gather_8_8_4", np.uint64, np.uint32), ("gather_4_2_4", np.uint16, np.uint32)])
def test_stand_alone_gather(tmp_path, name, element, index):
    """reordered_data[p] = data[indices[p]] (gather.cuh:151-164) == numpy fancy indexing, for tables staged in LDS and not"""
    limit = 1 << (8 * np.dtype(index).itemsize)
    for num_indices, table_length, grid, offset in [(1, 1, 0, 0), (100003, min(limit, 1000), 0, 0), (1 << 20, min(limit, 60000), 0, 0),
                                                    (300000, min(limit, 3000000), 0, 0), (50001, min(limit, 256), 3, 1)]:
        r = np.random.default_rng(num_indices)
        indices = r.integers(0, table_length, size=num_indices).astype(index)
        table = ((np.arange(table_length, dtype=np.uint64) + 1) * np.uint64(2654435761)).astype(element)
        in_path, out_path = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
        indices.tofile(in_path)
        done = subprocess.run([TOOL, name, "gather", str(num_indices), str(table_length), str(grid), str(offset), in_path, out_path],
                              capture_output=True, text=True, timeout=300)
        assert done.returncode == 0 and done.stdout.startswith("OK"), done.stdout + done.stderr
        out = np.fromfile(out_path, dtype=element)
        assert (out[num_indices:].view(np.uint8) == 0xCD).all(), "wrote past the last element"
        assert (out[:num_indices] == table[indices.astype(np.int64)]).all(), (name, num_indices, table_length)


@pytest.mark.parametrize("name,out_dtype,model", [("intervals_4_int_linear", np.int32, "linear"), ("intervals_4_short_linear", np.int16, "linear"),
                                                  ("intervals_8_int_constant", np.int32, "constant")])
def test_model_decompress_with_intervals(tmp_path, name, out_dtype, model):
    """model::decompress_with_intervals (model.cuh:55-98) is Frame of Reference with all offsets zero: checked against the C oracle's FOR"""
    from oracle import oracle
    for n, period in [(1, 4), (4099, 64), (1000003, 4096), (70000, 12), (333, 1000), (1 << 20, 1 << 20)]:
        for grid, offset in ((0, 0), (3, 1)):
            r = np.random.default_rng(n + period)
            num_intervals = -(-n // period)
            records = r.integers(-1000000, 1000000, size=(num_intervals, 2 if model == "linear" else 1)).astype(np.int32)
            if model == "linear":
                records[:, 1] = r.integers(-300, 300, size=num_intervals)
            in_path, out_path = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
            records.tofile(in_path)
            done = subprocess.run([TOOL, name, "intervals", str(n), str(period), str(grid), str(offset), in_path, out_path],
                                  capture_output=True, text=True, timeout=300)
            assert done.returncode == 0 and done.stdout.startswith("OK"), done.stdout + done.stderr
            out = np.fromfile(out_path, dtype=out_dtype)
            assert (out[n:].view(np.uint8) == 0xCD).all(), "wrote past the last element"
            zero_offsets = np.zeros(n, dtype=np.int8)
            expect = oracle.frame_of_reference(zero_offsets, records, n, period, out_dtype, model)
            assert (out[:n] == expect).all(), (name, n, period, grid, offset)


def test_scan_throughput_is_reported(tmp_path, capsys):
    """not a pass/fail bar: prints the two-phase rate at 2^28 elements so that the log of the GPU run holds it"""
    n, L = 1 << 28, 1 << 17
    ms = _run(tmp_path, "plus_u32_u32_inclusive", "two_phase", n, L, seed=5)
    with capsys.disabled():
        print("\n[scan] reduce_segments + scan_using_segment_reductions, plus<u32>, 2^28 elements, segments of 2^17: %.3f ms = %.0f GB/s "
              "algorithmic (12 B/element)" % (ms, 12 * n / ms / 1e6))

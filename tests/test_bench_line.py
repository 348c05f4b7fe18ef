"""The final stdout line of bench.py is what the driver parses: compact (round 5's 22 kB line came back `parsed: null`), strict JSON, and
carrying every key of the contract + `roofline` + `cpu_baseline`."""
import io
import json
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline")
ROOFLINE = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "algorithmic_bytes_per_launch", "mean_launch_ms")
CPU_BASELINE = ("value", "unit", "cores", "kind", "sample", "single_thread_value", "cpu")


def full_record():
    """round 5's whole record as committed (the one the driver could not recover), with every block the coverage run adds"""
    with open(os.path.join(ROOT, "profiles", "r05_bench_default.json")) as f:
        record = json.loads(f.read().strip().splitlines()[-1])
    entry = next(iter(record["schemes"].values()))
    for k in range(11):   # --schemes coverage: eleven more entries
        record["schemes"]["extra_workload_with_a_long_name_%d" % k] = entry
    return record


def test_final_line_is_compact_and_complete(tmp_path):
    record = full_record()
    out, err = io.StringIO(), io.StringIO()
    with redirect_stdout(out), redirect_stderr(err):
        bench.emit(record, str(tmp_path / "bench_details.json"))
    lines = out.getvalue().splitlines()
    assert len(lines) == 1, "exactly one stdout line: the compact one"
    final = lines[-1]
    assert len(final) < 4096 and len(final) < bench.FINAL_LINE_LIMIT
    parsed = json.loads(final, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))   # no NaN / Infinity
    for key in CONTRACT:
        assert key in parsed, key
    for key in ROOFLINE:
        assert key in parsed["roofline"], key
    for key in CPU_BASELINE:
        assert key in parsed["cpu_baseline"], key
    assert parsed["config"]["workload"] and "model" not in parsed["config"]
    assert parsed["roofline"]["bound"] == "hbm" and parsed["roofline"]["peak"] == 8000.0
    assert abs(parsed["roofline"]["frac"] - parsed["roofline"]["achieved"] / parsed["roofline"]["peak"]) < 1e-3
    assert set(parsed["schemes_frac"]) == set(record["schemes"])
    assert all(isinstance(v, float) for v in parsed["schemes_frac"].values())
    # the whole record is beside it, unabridged
    with open(tmp_path / "bench_details.json") as f:
        assert json.load(f) == record
    assert json.loads(err.getvalue()) == record


def test_nan_is_refused(tmp_path):
    import pytest
    record = full_record()
    record["value"] = float("nan")
    with pytest.raises(ValueError):
        with redirect_stdout(io.StringIO()), redirect_stderr(io.StringIO()):
            bench.emit(record, str(tmp_path / "d.json"))


def test_driver_flags_are_the_defaults_the_contract_names():
    sys_argv, sys.argv = sys.argv, ["bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5"]
    try:
        args = bench.parse()
    finally:
        sys.argv = sys_argv
    assert (args.gpus, args.steps, args.warmup) == (1, 20, 5)

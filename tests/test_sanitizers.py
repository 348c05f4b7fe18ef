"""CPU-only sanitizer runs (GPU AddressSanitizer is not available on this pool):

  * the oracle's C restatement built with the address and undefined-behaviour sanitizers (oracle/Makefile: libgiddy_oracle_asan.so) replays
    tests/test_oracle.py and tests/test_golden.py — every edge shape the oracle is pinned on — in a child process;
  * the C ABI and the launch-configuration resolver (libgiddy_amd/csrc/c_abi.cpp, resolve_launch_configuration.cpp) built
    host-side with the same sanitizers and linked with the library's kernel-wrapper objects replay tests/test_boundary.py.
A report from either sanitizer fails the test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORTS = ("ERROR: AddressSanitizer", "runtime error:", "ERROR: LeakSanitizer")


def _run_under(preload, extra_env, tests):
    env = dict(os.environ, LD_PRELOAD=preload, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1",
               PYTHONPATH=ROOT, **extra_env)
    result = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + tests,
                            cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    output = result.stdout + result.stderr
    assert not any(marker in output for marker in REPORTS), output[-6000:]
    assert result.returncode == 0, output[-6000:]
    assert " passed" in output
    return output


def test_oracle_under_address_and_undefined_behaviour_sanitizers():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "libgiddy_oracle_asan.so"])
    runtime = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(runtime):
        pytest.skip("gcc's libasan is not installed")
    _run_under(runtime, {"GIDDY_ORACLE_LIBRARY_PATH": os.path.join(ROOT, "oracle", "libgiddy_oracle_asan.so")},
               ["tests/test_oracle.py", "tests/test_golden.py"])


def test_c_abi_and_resolver_under_address_and_undefined_behaviour_sanitizers():
    resource_dir = subprocess.check_output(["/opt/rocm/lib/llvm/bin/clang", "-print-resource-dir"], text=True).strip()
    runtime = os.path.join(resource_dir, "lib", "linux", "libclang_rt.asan-x86_64.so")
    if not os.path.exists(runtime):
        pytest.skip("clang's ASan runtime is not installed")
    # the recipe (sanitizer flags included) is the `sanitized` target of libgiddy_amd/csrc/Makefile
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "libgiddy_amd", "csrc"), "-j", str(min(8, os.cpu_count() or 1)), "-s", "sanitized"])
    library = os.path.join(ROOT, "build", "sanitized", "libgiddy_amd_sanitized.so")
    _run_under(runtime, {"GIDDY_LIBRARY_PATH": library}, ["tests/test_boundary.py"])

"""CPU test: the C++ header API compiled (hipcc, host + gfx950 device code) against the reference's include
paths and exercised on the host alone — typed/type-erased launch-configuration resolution of every wrapper
family, the declarative resolver and its exceptions, the factory (tests/cpp/api_surface.hip)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_api_surface():
    out_dir = os.path.join(ROOT, "build", "tests")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "api_surface")
    src = os.path.join(ROOT, "tests", "cpp", "api_surface.hip")
    lib = os.path.join(ROOT, "libgiddy_amd")
    subprocess.check_call(["hipcc", "-std=c++17", "-O1", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "include", "reference_paths"), src, "-o", exe,
                           "-L" + lib, "-lgiddy_amd", "-Wl,-rpath," + lib])
    result = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert result.returncode == 0, result.stdout + result.stderr
    assert "api_surface OK" in result.stdout


REFERENCE_CALLERS = "/root/reference/examples/src/modes_of_use"
MODES = ["factory_provided_type_erased_wrapper", "instantiation_of_wrapper", "direct_use_of_kernel"]


@pytest.mark.parametrize("mode", MODES)
def test_reference_callers_compile_unchanged(mode):
    """The reference's own three callers (examples/src/modes_of_use/*.cu: factory-produced type-erased wrapper with
    `__restrict__` pointers in the argument map, :8-45; instantiated wrapper launched with five of the kernel's six
    arguments, instantiation_of_wrapper.cu:31-38; direct kernel launch) compile from where they lie in the reference
    tree — no copy, no edit — against include/reference_paths and link with libgiddy_amd.so.  (They run on the GPU in tests/test_gpu_examples.py.)"""
    source = os.path.join(REFERENCE_CALLERS, mode + ".cu")
    if not os.path.exists(source):
        pytest.skip("the reference tree is not present on this machine")
    out_dir = os.path.join(ROOT, "build", "tests")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(ROOT, "libgiddy_amd")
    result = subprocess.run(["hipcc", "-std=c++17", "-O1", "--offload-arch=gfx950", "-Wall", "-I" + os.path.join(ROOT, "include"),
                             "-I" + os.path.join(ROOT, "include", "reference_paths"), "-x", "hip", source,
                             "-o", os.path.join(out_dir, "reference_" + mode), "-L" + lib, "-lgiddy_amd", "-Wl,-rpath," + lib],
                            capture_output=True, text=True)
    assert result.returncode == 0, result.stderr
    assert "error" not in result.stderr

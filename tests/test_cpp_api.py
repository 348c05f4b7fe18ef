"""CPU test: the C++ header API compiled (hipcc, host + gfx950 device code) against the reference's include
paths and exercised on the host alone — typed/type-erased launch-configuration resolution of every wrapper
family, the declarative resolver and its exceptions, the factory (tests/cpp/api_surface.hip)."""
import os
import subprocess

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

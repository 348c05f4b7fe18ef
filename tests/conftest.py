import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Built artefacts are kept out of git history; in a fresh checkout build them once (hipcc cross-compiles
    # gfx950 without a GPU, ~15 s).  There is no non-native fallback to test instead.
    if not os.path.exists(os.path.join(ROOT, "libgiddy_amd", "libgiddy_amd.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "libgiddy_amd", "csrc"), "-j", str(min(8, os.cpu_count() or 1)), "-s"])
    if not os.path.exists(os.path.join(ROOT, "oracle", "libgiddy_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])

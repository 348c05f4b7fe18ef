"""-m gpu: the three C++ "modes of use" programs (examples/) — factory-produced type-erased wrapper,
instantiated wrapper, direct kernel launch — each decodes the reference's known-answer test through the
C++ API and checks every element itself."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROGRAMS = ["factory_provided_type_erased_wrapper", "instantiation_of_wrapper", "direct_use_of_kernel", "fused_decode_and_patch", "sharded_decode_on_all_devices", "batch_of_small_columns"]


@pytest.mark.parametrize("program", PROGRAMS)
def test_mode_of_use(program):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = os.path.join(ROOT, "examples", "bin", program)
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
    result = subprocess.run([path], capture_output=True, text=True, timeout=120)
    assert result.returncode == 0, result.stdout + result.stderr
    assert "OK" in result.stdout


REFERENCE_MODES = ["factory_provided_type_erased_wrapper", "instantiation_of_wrapper", "direct_use_of_kernel"]


@pytest.mark.parametrize("mode", REFERENCE_MODES)
def test_reference_caller_runs(mode):
    """examples/bin/reference_<mode>: the reference's OWN caller for this mode (its source compiled unchanged from the
    reference tree by examples/Makefile where that tree exists; the binary travels with the snapshot) decoding its
    known-answer column through this library and checking it in its own main() (examples/src/modes_of_use/common.cuh:91-102)."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = os.path.join(ROOT, "examples", "bin", "reference_" + mode)
    if not os.path.exists(path):
        if not os.path.exists("/root/reference/examples/src/modes_of_use"):
            pytest.skip("neither the prebuilt program nor the reference tree is here")
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
    result = subprocess.run([path], capture_output=True, text=True, timeout=120)
    assert result.returncode == 0, result.stdout + result.stderr
    assert "Unexpected value" not in result.stderr


def test_polynomial_model_functor_through_the_kernel_templates():
    """tests/cpp/polynomial_model.hip: model::decompress and frame_of_reference::decompress instantiated with the reference's generic
    parametric_model::polynomial<4, int, 3> (model_functors.hpp:45-74; unregistered there and here), against the functor on the host"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as entry
    if not os.path.exists(entry.POLYNOMIAL_MODEL):
        entry.build_cpp_test_tools()
    result = subprocess.run([entry.POLYNOMIAL_MODEL], capture_output=True, text=True, timeout=120)
    assert result.returncode == 0 and "polynomial_model OK" in result.stdout, result.stdout + result.stderr

"""-m gpu: every shape of workgroup and column through the dictionary decoders' overlapped staging.

include/giddy/kernels/decompression/dictionary.hpp stages a small dictionary in LDS with a __syncthreads() inside a hook that
stream_tiles calls behind the first tile's (first chunk's) key loads.  Until round 6 the hook was called per tile behind a flag, so the
barrier sat at three call sites; since then the tile and chunk loops are software-pipelined and the hook is called from ONE place outside
them (tools/scan_isa.py, scan 3, asserts that on the built code objects).  This test stays as the functional side of it: every Dictionary
instantiation, plain and fused with patches, is launched with workgroups of 64 ... 1024 threads, grids from one workgroup to the natural
one, and column lengths that leave some wavefronts of a workgroup without a tile, some with a ragged one and some with several — in a
CHILD process under a watchdog, which the parent kills (and fails) if it does not finish in time."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import itertools, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import numpy as np, torch
from libgiddy_amd import schemes, registered as r
keys = [k for k in r.kernel_t.registered_keys() if k.startswith("decompression::dictionary::")]
assert len(keys) >= 20, keys
stream = torch.cuda.current_stream().cuda_stream
signed = {1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}
launches = 0
for key in keys:
    patched = "patched" in key
    index_size, usize, ksize = (int(x.strip(" u")) for x in key[key.index("<") + 1: key.index(">")].split(","))
    for entries in (1, 7, 256, 1000):
        if entries > (1 << (8 * ksize)): continue
        for n in (1, 63, 64, 4095, 4096, 4097, 70001, 1 << 20):
            rng = np.random.default_rng(n + entries + usize)
            dictionary = rng.integers(0, 1 << min(8 * usize - 1, 62), size=entries, dtype=np.int64)
            column_keys = rng.integers(0, entries, size=n, dtype=np.int64)
            d = torch.from_numpy(dictionary).to(signed[usize]).cuda(); k = torch.from_numpy(column_keys).to(signed[ksize]).cuda()
            expect = d[torch.from_numpy(column_keys).cuda()]
            patch_positions = torch.arange(0, n, max(n // 7, 1), dtype=signed[index_size]).cuda()
            patch_values = torch.full((patch_positions.numel(),), 5, dtype=signed[usize]).cuda()
            if patched: expect = expect.clone(); expect[patch_positions.long()] = 5
            for block, grid in itertools.product((None, 64, 128, 256, 512, 1024), (None, 1, 3, 1000)):
                limits = r.launch_configuration_limits_t(block_size=block, grid_size=grid)
                out = torch.full((n,), -1, dtype=signed[usize], device="cuda")
                try:
                    if patched:
                        schemes.patched_dictionary(out, k, d, patch_values, patch_positions, n, entries, patch_positions.numel(), uncompressed_size=usize,
                                                   dictionary_index_size=ksize, index_size=index_size, stream=stream, limits=limits)
                    else:
                        schemes.dictionary(out, k, d, n, entries, uncompressed_size=usize, dictionary_index_size=ksize, index_size=index_size, stream=stream, limits=limits)
                except RuntimeError as e:   # (a limit the resolver refuses — e.g. a grid too small for a non-serializing configuration — is not this test's business)
                    if "invalid_argument" in str(e) or "logic_error" in str(e): continue
                    raise
                torch.cuda.synchronize()
                assert bool((out == expect).all()), (key, entries, n, block, grid)
                launches += 1
print("GUARD OK", launches, "launches of", len(keys), "instantiations")
'''


def test_every_dictionary_instantiation_at_every_block_size_finishes():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    code = CHILD % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}
    child = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    try:
        output, _ = child.communicate(timeout=600)
    except subprocess.TimeoutExpired:
        os.killpg(child.pid, 9)   # exactly the process group this test started
        child.communicate()
        pytest.fail("a Dictionary launch did not finish within the watchdog's 600 s: the staging barrier (dictionary.hpp) may hang with this compiler")
    assert child.returncode == 0 and "GUARD OK" in output, output[-3000:]

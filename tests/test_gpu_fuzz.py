"""-m gpu: seeded random shapes for every scheme against the CPU oracle — segment lengths that do not
divide anything, ragged tails, tiny and long segments, all element widths.  Output buffers carry guard
bands (gpu_util), so out-of-bounds writes fail too."""
import numpy as np
import pytest

import cases
from libgiddy_amd import formats
from oracle import oracle

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _shapes(seed, count, max_n=60000):
    r = cases.rng(seed)
    for _ in range(count):
        n = int(r.integers(1, max_n))
        kind = r.integers(0, 4)
        if kind == 0:
            seglen = int(r.integers(1, 40))
        elif kind == 1:
            seglen = int(r.choice([64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768]))
        elif kind == 2:
            seglen = int(r.integers(40, 20000))
        else:
            seglen = n + int(r.integers(0, 5))
        yield n, max(seglen, 1)


@pytest.mark.parametrize("usize,csize", [(2, 1), (4, 1), (4, 2), (8, 1), (8, 2), (8, 4)])
def test_delta_fuzz(usize, csize):
    import gpu_util as g
    from libgiddy_amd import schemes
    for n, seglen in _shapes(usize * 10 + csize, 40):
        f, truth = cases.delta_case(n, seglen, "u%d" % usize, "u%d" % csize, seed=n)
        out = g.dev_empty(n, "u%d" % usize)
        schemes.delta(out, g.dev(f["compressed_input"]), g.dev(f["anchor_values"]), n, seglen, uncompressed_size=usize,
                      compressed_size=csize, stream=g.stream())
        got = g.host(out, "u%d" % usize, n)
        assert (got == truth).all(), (n, seglen, np.flatnonzero(got != truth)[:4])


@pytest.mark.parametrize("usize,csize,model", [(2, 1, "linear"), (4, 1, "constant"), (4, 2, "linear"), (4, 2, "constant")])
def test_for_fuzz(usize, csize, model):
    import gpu_util as g
    from libgiddy_amd import schemes
    for n, seglen in _shapes(usize + csize, 30):
        seglen = max(4, seglen // 4 * 4)   # the launch parameters require segment_length % 4 == 0 (frame_of_reference.cuh:102)
        f = cases.for_case(n, seglen, "i%d" % usize, "i%d" % csize, model, seed=n)
        expect = oracle.frame_of_reference(f["compressed_input"], f["segment_model_coefficients"], n, seglen, "i%d" % usize, model)
        out = g.dev_empty(n, "i%d" % usize)
        schemes.frame_of_reference(out, g.dev(f["compressed_input"]), g.dev(f["segment_model_coefficients"]), n, seglen,
                                   uncompressed_size=usize, compressed_size=csize, model=model, stream=g.stream())
        assert (g.host(out, "i%d" % usize, n) == expect).all(), (n, seglen)


@pytest.mark.parametrize("usize,lsize", [(1, 1), (2, 2), (4, 1), (4, 2), (4, 4), (8, 2)])
def test_rle_rpe_fuzz(usize, lsize):
    import gpu_util as g
    from libgiddy_amd import schemes
    r = cases.rng(usize * 7 + lsize)
    for n, seglen in _shapes(usize * 5 + lsize, 25):
        mean_run = float(r.choice([1.5, 4, 32, 300, 5000]))
        col = cases.runs_column(n, mean_run, np.uint32, seed=n).astype(np.uint64).astype("u%d" % usize)
        f = formats.run_length_encoding(col, seglen, "u%d" % lsize)
        out = g.dev_empty(n, "u%d" % usize)
        schemes.run_length_encoding(out, g.dev(f["run_data"]), g.dev(f["run_lengths"]), g.dev(f["position_anchors"]),
                                    g.dev(f["intra_run_anchor_offsets"]), seglen, f["num_segments"], f["num_element_runs"], n,
                                    uncompressed_size=usize, run_length_size=lsize, stream=g.stream())
        got = g.host(out, "u%d" % usize, n)
        assert (got == col).all(), ("rle", n, seglen, mean_run, np.flatnonzero(got != col)[:4])
        for relative in (True, False):
            if relative and seglen > 65536:
                continue
            f = formats.run_position_encoding(col, seglen, np.uint16 if relative else np.uint32, relative)
            out = g.dev_empty(n, "u%d" % usize)
            schemes.run_position_encoding(out, g.dev(f["run_data"]), g.dev(f["run_start_positions"]), g.dev(f["position_anchors"]), seglen,
                                          f["num_anchors"], f["num_element_runs"], n, uncompressed_size=usize,
                                          position_size=2 if relative else 4, relative=relative, stream=g.stream())
            got = g.host(out, "u%d" % usize, n)
            assert (got == col).all(), ("rpe", relative, n, seglen, mean_run, np.flatnonzero(got != col)[:4])


@pytest.mark.parametrize("usize,container,index_size", [(2, 4, 4), (4, 4, 8), (8, 4, 8), (8, 8, 4), (4, 8, 4)])
def test_dzb_variable_fuzz(usize, container, index_size):
    import gpu_util as g
    from libgiddy_amd import schemes
    r = cases.rng(usize + container)
    for n, seglen in _shapes(usize * 3 + container, 30):
        bits = int(r.integers(1, 5))
        min_size = int(r.integers(0, 2))
        if min_size + (1 << bits) - 1 < 1:
            continue
        col, sizes = cases.dzbv_column(n, "u%d" % usize, min_size, bits, seed=n)
        f = formats.discard_zero_bytes_variable(col, seglen, min_size, bits, "u%d" % container, index_size, element_sizes=sizes)
        out = g.dev_empty(n, "u%d" % usize)
        schemes.discard_zero_bytes_variable(out, g.dev(f["compressed_data"]), g.dev(f["packed_element_sizes"]), g.dev(f["position_anchors"]),
                                            seglen, n, min_size, bits, uncompressed_size=usize, container_size=container,
                                            index_size=index_size, stream=g.stream())
        got = g.host(out, "u%d" % usize, n)
        assert (got == col).all(), (n, seglen, bits, min_size, np.flatnonzero(got != col)[:4])


@pytest.mark.parametrize("usize,dsize", [(1, 1), (2, 2), (4, 1), (8, 2), (8, 8)])
def test_dictionary_and_dzb_fixed_fuzz(usize, dsize):
    import gpu_util as g
    from libgiddy_amd import schemes
    r = cases.rng(usize * 3 + dsize)
    for _ in range(20):
        n = int(r.integers(1, 100000))
        entries = int(r.integers(1, min(1 << (8 * dsize), 5000) + 1))
        f = cases.dictionary_case(n, entries, "u%d" % usize, "u%d" % dsize, seed=n)
        out = g.dev_empty(n, "u%d" % usize)
        schemes.dictionary(out, g.dev(f["compressed_input"]), g.dev(f["dictionary_entries"]), n, entries, uncompressed_size=usize,
                           dictionary_index_size=dsize, stream=g.stream())
        assert (g.host(out, "u%d" % usize, n) == oracle.dictionary(f["compressed_input"], f["dictionary_entries"])).all(), (n, entries)
        if usize > dsize:
            c = r.integers(0, np.iinfo("u%d" % dsize).max, size=n, endpoint=True).astype("u%d" % dsize)
            out = g.dev_empty(n, "u%d" % usize)
            schemes.discard_zero_bytes_fixed(out, g.dev(c), n, uncompressed_size=usize, compressed_size=dsize, stream=g.stream())
            assert (g.host(out, "u%d" % usize, n) == c).all()

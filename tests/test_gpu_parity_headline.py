"""GPU parity (bit-exact) of the HIP kernels against the CPU oracle, through the C ABI:
Frame-of-Reference, Dictionary, Delta and the naive scatter.  Run with `-m gpu` on an MI355X."""
import numpy as np
import pytest

import cases
from oracle import oracle

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _imports():
    from libgiddy_amd import schemes, registered
    import gpu_util
    return schemes, registered, gpu_util


def test_kat_for_1_on_gpu():
    """the reference's own known-answer test (examples/src/modes_of_use/common.cuh:31-49,91-102)"""
    schemes, r, g = _imports()
    length, seglen = 2 << 12, 2 << 10
    nseg = length // seglen
    comp = np.tile(np.arange(seglen, dtype=np.int16), nseg)
    coefs = (np.arange(nseg, dtype=np.int32) * 100).reshape(nseg, 1)
    out = g.dev_empty(length, np.int32)
    schemes.frame_of_reference(out, g.dev(comp), g.dev(coefs), length, seglen, stream=g.stream())
    got = g.host(out, np.int32, length)
    for s in range(nseg):
        assert (got[s * seglen:(s + 1) * seglen] == 100 * s + np.arange(seglen)).all()


FOR_TUPLES = [  # (index_size, uncompressed, compressed, model, model_index_size) — frame_of_reference.cu:132-142
    (4, 2, 1, "linear", 4), (4, 4, 1, "linear", 4), (4, 4, 2, "linear", 4), (8, 4, 2, "linear", 4),
    (8, 4, 2, "linear", 8), (4, 2, 1, "constant", 4), (4, 4, 1, "constant", 4), (4, 4, 2, "constant", 4)]


@pytest.mark.parametrize("index_size,usize,csize,model,model_index_size", FOR_TUPLES)
@pytest.mark.parametrize("n,seglen", [(1, 4), (4099, 64), (1 << 20, 1 << 20), (1000003, 4096), (70000, 12), (333, 1000)])
def test_frame_of_reference(index_size, usize, csize, model, model_index_size, n, seglen):
    schemes, r, g = _imports()
    ud, cd = np.dtype("i%d" % usize), np.dtype("i%d" % csize)
    f = cases.for_case(n, seglen, ud, cd, model, seed=n + usize)
    expect = oracle.frame_of_reference(f["compressed_input"], f["segment_model_coefficients"], n, seglen, ud, model,
                                       model_index_size=model_index_size)
    out = g.dev_empty(n, ud)
    schemes.frame_of_reference(out, g.dev(f["compressed_input"]), g.dev(f["segment_model_coefficients"]), n, seglen,
                               uncompressed_size=usize, compressed_size=csize, model=model, index_size=index_size,
                               model_index_size=model_index_size, stream=g.stream())
    assert (g.host(out, ud, n) == expect).all()


def test_frame_of_reference_unaligned_and_limits():
    schemes, r, g = _imports()
    n, seglen = 50001, 256
    f = cases.for_case(n + 3, seglen, np.int32, np.int16, "linear", seed=5)
    comp = f["compressed_input"]
    # misaligned input pointer (2 bytes off) -> scalar path
    buf = g.dev(comp)
    expect = oracle.frame_of_reference(comp[1:], f["segment_model_coefficients"], n, seglen, np.int32, "linear")
    out = g.dev_empty(n, np.int32)
    schemes.frame_of_reference(out, buf.data_ptr() + 2, g.dev(f["segment_model_coefficients"]), n, seglen,
                               model="linear", stream=g.stream())
    assert (g.host(out, np.int32, n) == expect).all()
    # one-time limits: tiny grid and block still decode everything
    out = g.dev_empty(n, np.int32)
    cfg = schemes.frame_of_reference(out, buf.data_ptr() + 2, g.dev(f["segment_model_coefficients"]), n, seglen,
                                     model="linear", stream=g.stream(),
                                     limits=r.launch_configuration_limits_t(block_size=64, grid_size=3))
    assert cfg.block_dimensions[0] == 64 and cfg.grid_dimensions[0] <= 3
    assert (g.host(out, np.int32, n) == expect).all()


DICT_TUPLES = [(1, 1), (2, 1), (2, 2), (4, 1), (4, 2), (4, 4), (8, 1), (8, 2), (8, 4), (8, 8)]  # dictionary.cu:113-139


@pytest.mark.parametrize("usize,dsize", DICT_TUPLES)
@pytest.mark.parametrize("index_size", [4, 8])
def test_dictionary(usize, dsize, index_size):
    schemes, r, g = _imports()
    ud, kd = np.dtype("u%d" % usize), np.dtype("u%d" % dsize)
    for n, entries in [(0, 4), (1, 1), (1027, 256 if dsize == 1 else 3000), (1 << 20, 256 if dsize == 1 else 40000)]:
        entries = min(entries, 1 << (8 * dsize))
        f = cases.dictionary_case(n, entries, ud, kd, seed=n + usize)
        expect = oracle.dictionary(f["compressed_input"], f["dictionary_entries"])
        for cache in (None, False):
            out = g.dev_empty(n, ud)
            schemes.dictionary(out, g.dev(f["compressed_input"]), g.dev(f["dictionary_entries"]), n, entries,
                               uncompressed_size=usize, dictionary_index_size=dsize, index_size=index_size,
                               cache_dictionary_in_shared_mem=cache, stream=g.stream())
            assert (g.host(out, ud, n) == expect).all(), (n, entries, cache)


def test_dictionary_too_large_for_lds_falls_back():
    schemes, r, g = _imports()
    n, entries = 300000, 1 << 16
    f = cases.dictionary_case(n, entries, np.uint64, np.uint16, seed=3)   # 512 KiB dictionary
    out = g.dev_empty(n, np.uint64)
    cfg = schemes.dictionary(out, g.dev(f["compressed_input"]), g.dev(f["dictionary_entries"]), n, entries,
                             uncompressed_size=8, dictionary_index_size=2, stream=g.stream())
    assert cfg.dynamic_shared_memory_size == 0
    assert (g.host(out, np.uint64, n) == oracle.dictionary(f["compressed_input"], f["dictionary_entries"])).all()


DELTA_TUPLES = [(2, 1), (4, 1), (4, 2), (8, 1), (8, 2), (8, 4)]  # delta.cu:94-106


@pytest.mark.parametrize("usize,csize", DELTA_TUPLES)
@pytest.mark.parametrize("index_size", [4, 8])
@pytest.mark.parametrize("n,seglen", [
    (1, 1), (63, 7), (4096, 4096), (1000003, 4096), (300000, 512), (100000, 1), (50000, 100), (100, 1000),
    (1 << 20, 1 << 18), (3000000, 1 << 20), (70000, 65536 + 512), (12345, 1024),
    (1000003, 1000), (300000, 3000), (500000, 5000), (200000, 136), (99999, 6),   # not whole rows: ragged wave path / block path
    # long segments: tiles chained by look-back (whole tiles per segment, or one segment), 4 / 8 / 16 wavefronts per tile by chain length
    (1 << 21, 16384), (1 << 21, 32768), ((1 << 21) + 12345, 65536), (5 << 20, 1 << 20), ((1 << 22) + 4096 * 3 + 8, 1 << 22),
    (3000001, 3000001), (1 << 24, 1 << 24), ((1 << 24) + 999, 1 << 25)])
def test_delta(usize, csize, index_size, n, seglen):
    schemes, r, g = _imports()
    ud, cd = np.dtype("u%d" % usize), np.dtype("u%d" % csize)
    f, truth = cases.delta_case(n, seglen, ud, cd, seed=n + seglen + usize)
    expect = oracle.delta(f["compressed_input"], f["anchor_values"], seglen)
    assert (expect == truth).all()
    out = g.dev_empty(n, ud)
    schemes.delta(out, g.dev(f["compressed_input"]), g.dev(f["anchor_values"]), n, seglen,
                  uncompressed_size=usize, compressed_size=csize, index_size=index_size, stream=g.stream())
    got = g.host(out, ud, n)
    bad = np.flatnonzero(got != expect)
    assert len(bad) == 0, (bad[:5], got[bad[:5]], expect[bad[:5]])


def test_delta_with_limits():
    schemes, r, g = _imports()
    n, seglen = 500000, 4096
    f, truth = cases.delta_case(n, seglen, np.uint64, np.uint16, seed=77)
    for limits in (r.launch_configuration_limits_t(block_size=64), r.launch_configuration_limits_t(grid_size=5),
                   r.launch_configuration_limits_t(block_size=128, grid_size=1)):
        out = g.dev_empty(n, np.uint64)
        schemes.delta(out, g.dev(f["compressed_input"]), g.dev(f["anchor_values"]), n, seglen, stream=g.stream(), limits=limits)
        assert (g.host(out, np.uint64, n) == truth).all()


@pytest.mark.parametrize("osize,esize,isize", [(4, 1, 4), (4, 2, 2), (4, 4, 4), (4, 8, 4), (8, 8, 8), (8, 4, 1), (4, 8, 2), (8, 1, 4)])
def test_scatter(osize, esize, isize):
    schemes, r, g = _imports()
    ed = np.dtype("u%d" % esize)
    n = 1 << 20
    for npatch in (0, 1, 5, 1001, 200 if isize == 1 else 50000):
        npatch = min(npatch, (1 << (8 * isize)) - 1)
        target = cases.rng(npatch).integers(0, 255, size=n).astype(ed)
        p = cases.patch_case(n, npatch, ed, osize, seed=npatch + 1) if npatch else {"indices": np.zeros(0, "u%d" % osize), "data": np.zeros(0, ed)}
        expect = oracle.scatter(target.copy(), p["data"], p["indices"]) if npatch else target
        t = g.dev(target)
        schemes.scatter(t, g.dev(p["data"]), g.dev(p["indices"]), npatch, output_index_size=osize, element_size=esize,
                        input_index_size=isize, stream=g.stream())
        assert (g.host(t, ed, n) == expect).all()


def test_delta_then_patch_pipeline_full_fixture():
    """C3 in miniature: delta decode + naive patches on one stream, vs the golden fixture."""
    import os
    schemes, r, g = _imports()
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "delta_u64_u16_patched.npz"))
    n, seglen = int(fx["length"]), int(fx["segment_length"])
    out = g.dev_empty(n, np.uint64)
    schemes.delta(out, g.dev(fx["compressed_input"]), g.dev(fx["anchor_values"]), n, seglen, stream=g.stream())
    schemes.scatter(out, g.dev(fx["patch_data"]), g.dev(fx["patch_indices"]), len(fx["patch_data"]), stream=g.stream())
    assert (g.host(out, np.uint64, n) == fx["expected_patched"]).all()


def test_launches_are_graph_capturable():
    """enqueue_launch only enqueues: a Delta + patch pipeline captured into a HIP graph and replayed
    gives the same bits (no hidden synchronisation or allocation inside the library)."""
    schemes, r, g = _imports()
    n, L = 1 << 20, 4096
    f, truth = cases.delta_case(n, L, np.uint64, np.uint16, seed=123)
    p = cases.patch_case(n, 1000, np.uint64, 4, seed=124)
    expect = truth.copy(); expect[p["indices"].astype(np.int64)] = p["data"]
    d_in, d_anchor, d_idx, d_val = g.dev(f["compressed_input"]), g.dev(f["anchor_values"]), g.dev(p["indices"]), g.dev(p["data"])
    out = g.dev_empty(n, np.uint64)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        cfg_d = schemes.delta(out, d_in, d_anchor, n, L, stream=side.cuda_stream)          # resolve outside the capture
        cfg_s = schemes.scatter(out, d_val, d_idx, 1000, stream=side.cuda_stream)
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        schemes.delta(out, d_in, d_anchor, n, L, stream=torch.cuda.current_stream().cuda_stream, launch_config=cfg_d)
        schemes.scatter(out, d_val, d_idx, 1000, stream=torch.cuda.current_stream().cuda_stream, launch_config=cfg_s)
    out.fill_(0xCD)
    graph.replay()
    assert (g.host(out, np.uint64, n) == expect).all()


def test_launches_with_scratch_are_graph_capturable():
    """The wrappers that take scratch of their own — chained Delta tiles, finer anchors for long run-encoded and discard-zero-bytes segments — inside a
    HIP graph: allocation and release become nodes of the graph (stream_scratch.hpp), two replays give the same bits."""
    schemes, r, g = _imports()
    from libgiddy_amd import formats
    n = 1 << 20
    f, truth = cases.delta_case(n, 1 << 18, np.uint64, np.uint16, seed=321)
    col = cases.runs_column(n, 40, np.uint32, seed=322)
    rle = formats.run_length_encoding(col, 1 << 16, np.uint16)
    rpe = formats.run_position_encoding(col, 1 << 16, np.uint16, True)
    d = {k: g.dev(v) for k, v in (("din", f["compressed_input"]), ("danchor", f["anchor_values"]), ("rd", rle["run_data"]), ("rl", rle["run_lengths"]),
                                  ("ra", rle["position_anchors"]), ("ri", rle["intra_run_anchor_offsets"]), ("pd", rpe["run_data"]), ("pp", rpe["run_start_positions"]),
                                  ("pa", rpe["position_anchors"]))}
    out_delta, out_rle, out_rpe = g.dev_empty(n, np.uint64), g.dev_empty(n, np.uint32), g.dev_empty(n, np.uint32)
    side = torch.cuda.Stream()

    def launches(stream, configs=None):
        c = configs or [None] * 3
        return [schemes.delta(out_delta, d["din"], d["danchor"], n, 1 << 18, stream=stream, launch_config=c[0]),
                schemes.run_length_encoding(out_rle, d["rd"], d["rl"], d["ra"], d["ri"], 1 << 16, rle["num_segments"], rle["num_element_runs"], n, stream=stream, launch_config=c[1]),
                schemes.run_position_encoding(out_rpe, d["pd"], d["pp"], d["pa"], 1 << 16, rpe["num_anchors"], rpe["num_element_runs"], n, stream=stream, launch_config=c[2])]
    with torch.cuda.stream(side):
        configs = launches(side.cuda_stream)   # (resolve outside the capture)
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        launches(torch.cuda.current_stream().cuda_stream, configs)
    for _ in range(2):
        for out in (out_delta, out_rle, out_rpe): out.fill_(0xCD)
        graph.replay()
        assert (g.host(out_delta, np.uint64, n) == truth).all()
        assert (g.host(out_rle, np.uint32, n) == col).all()
        assert (g.host(out_rpe, np.uint32, n) == col).all()


def test_dictionary_128_kib_staged_in_lds():
    """a full 16-bit-key dictionary of 16-bit values (128 KiB) is staged in LDS by one 1024-thread workgroup per CU"""
    schemes, r, g = _imports()
    n, entries = 3000001, 1 << 16
    f = cases.dictionary_case(n, entries, np.uint16, np.uint16, seed=31)
    out = g.dev_empty(n, np.uint16)
    cfg = schemes.dictionary(out, g.dev(f["compressed_input"]), g.dev(f["dictionary_entries"]), n, entries,
                             uncompressed_size=2, dictionary_index_size=2, stream=g.stream())
    assert cfg.dynamic_shared_memory_size == 2 * entries and cfg.block_dimensions[0] == 1024
    assert (g.host(out, np.uint16, n) == oracle.dictionary(f["compressed_input"], f["dictionary_entries"])).all()

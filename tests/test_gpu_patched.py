"""GPU parity of the fused decompress-and-patch kernels (the extension of SURVEY.md §8f.1): the result must be
bit-for-bit what the oracle's decoder followed by the oracle's naive scatter leaves — for every registered
instantiation, patch densities from none to every element, patches on row / tile / segment boundaries, duplicate
positions (the last one wins), ragged tails, unaligned bases and one-time launch limits."""
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


def _patches(n, how, value_dtype, index_size, seed):
    """ascending positions (with the odd duplicate) and values"""
    r = cases.rng(seed)
    if how == "none" or n == 0:
        positions = np.zeros(0, np.int64)
    elif how == "sparse":     # ~0.1 %, the C3 density
        positions = np.sort(r.choice(n, size=max(1, n // 1000), replace=False))
    elif how == "dense":      # every third element on average: more than 64 patches per row-group
        positions = np.flatnonzero(r.random(n) < 0.35)
    elif how == "all":
        positions = np.arange(n)
    elif how == "edges":      # first / last elements and everything around multiples of 64, 256, 1024, 4096
        marks = np.concatenate([np.arange(0, n, 64), np.arange(0, n, 4096) - 1, np.arange(0, n, 1024) + 1, [0, n - 1]])
        positions = np.unique(marks[(marks >= 0) & (marks < n)])
    elif how == "duplicates":
        positions = np.sort(r.integers(0, n, size=max(2, n // 50)))     # many repeats
    elif how == "repeats_at_one_percent":   # 1 % of patches drawn with replacement from a smaller pool: many equal positions, at a density where the
        # 2-byte wrappers launch the plain decoder + scatter_ascending instead of their fused kernel (kernel_wrappers/decompression/patched/decode_then_scatter.hpp)
        pool = np.sort(r.choice(n, size=max(1, n // 150), replace=False))
        positions = np.sort(r.choice(pool, size=max(2, n // 100), replace=True))
    elif how == "clustered":  # a few hundred consecutive patches, then nothing for a long while
        starts = r.integers(0, max(1, n - 300), size=max(1, n // 100000 + 1))
        positions = np.unique(np.concatenate([np.arange(s, min(n, s + 300)) for s in starts]))
    else:
        raise ValueError(how)
    values = r.integers(0, np.iinfo(value_dtype).max, size=len(positions), dtype=value_dtype, endpoint=True)
    return positions.astype("u%d" % index_size), values


def _expected(decoded, positions, values):
    out = decoded.copy()
    if len(positions):
        # sequential semantics of the scatter for equal positions: the last one wins
        out = oracle.scatter(out, values, positions, threads=1)
    return out


PATCH_SHAPES = ["none", "sparse", "dense", "all", "edges", "duplicates", "clustered", "repeats_at_one_percent"]
DELTA_TUPLES = [(2, 1), (4, 1), (4, 2), (8, 1), (8, 2), (8, 4)]


@pytest.mark.parametrize("usize,csize", DELTA_TUPLES)
@pytest.mark.parametrize("index_size", [4, 8])
@pytest.mark.parametrize("n,seglen", [(1, 1), (4096, 4096), (1000003, 4096), (300000, 256), (100000, 100), (1 << 20, 1 << 18), (70000, 65536 + 512),
                                      (1000003, 1000), (500000, 5000)])
def test_patched_delta(usize, csize, index_size, n, seglen):
    schemes, r, g = _imports()
    ud, cd = np.dtype("u%d" % usize), np.dtype("u%d" % csize)
    f, truth = cases.delta_case(n, seglen, ud, cd, seed=n + seglen + usize)
    d_in, d_anchor = g.dev(f["compressed_input"]), g.dev(f["anchor_values"])
    for how in PATCH_SHAPES:
        positions, values = _patches(n, how, ud, index_size, seed=len(how) + n)
        expect = _expected(truth, positions, values)
        out = g.dev_empty(n, ud)
        schemes.patched_delta(out, d_in, d_anchor, g.dev(values), g.dev(positions), n, seglen, len(positions),
                              uncompressed_size=usize, compressed_size=csize, index_size=index_size, stream=g.stream())
        got = g.host(out, ud, n)
        bad = np.flatnonzero(got != expect)
        assert len(bad) == 0, (how, bad[:5], got[bad[:5]], expect[bad[:5]])


FOR_TUPLES = [(4, 2, 1, "linear", 4), (4, 4, 1, "linear", 4), (4, 4, 2, "linear", 4), (8, 4, 2, "linear", 4),
              (8, 4, 2, "linear", 8), (4, 2, 1, "constant", 4), (4, 4, 1, "constant", 4), (4, 4, 2, "constant", 4)]


@pytest.mark.parametrize("index_size,usize,csize,model,model_index_size", FOR_TUPLES)
@pytest.mark.parametrize("n,seglen", [(1, 4), (4099, 64), (1000003, 4096), (70000, 12)])
def test_patched_frame_of_reference(index_size, usize, csize, model, model_index_size, n, seglen):
    schemes, r, g = _imports()
    ud, cd = np.dtype("i%d" % usize), np.dtype("i%d" % csize)
    f = cases.for_case(n, seglen, ud, cd, model, seed=n + usize)
    decoded = oracle.frame_of_reference(f["compressed_input"], f["segment_model_coefficients"], n, seglen, ud, model,
                                        model_index_size=model_index_size)
    d_in, d_coef = g.dev(f["compressed_input"]), g.dev(f["segment_model_coefficients"])
    for how in PATCH_SHAPES:
        positions, values = _patches(n, how, ud, index_size, seed=len(how) + n)
        expect = _expected(decoded, positions, values)
        out = g.dev_empty(n, ud)
        schemes.patched_frame_of_reference(out, d_in, d_coef, g.dev(values), g.dev(positions), n, seglen, len(positions),
                                           uncompressed_size=usize, compressed_size=csize, model=model, index_size=index_size,
                                           model_index_size=model_index_size, stream=g.stream())
        got = g.host(out, ud, n)
        bad = np.flatnonzero(got != expect)
        assert len(bad) == 0, (how, bad[:5], got[bad[:5]], expect[bad[:5]])


MODEL_TUPLES = [(4, 2, "linear"), (4, 4, "linear"), (8, 4, "linear"), (4, 2, "constant"), (4, 4, "constant")]


@pytest.mark.parametrize("index_size,usize,model", MODEL_TUPLES)
@pytest.mark.parametrize("n", [1, 255, 4099, 1000003])
def test_patched_model(index_size, usize, model, n):
    schemes, r, g = _imports()
    ud = np.dtype("i%d" % usize)
    coefficients = np.array([-12345, 37] if model == "linear" else [-4321], dtype=ud)
    decoded = oracle.model(coefficients, n, ud, model, model_index_size=index_size)
    for how in PATCH_SHAPES:
        positions, values = _patches(n, how, ud, index_size, seed=len(how) + n)
        expect = _expected(decoded, positions, values)
        out = g.dev_empty(n, ud)
        schemes.patched_model(out, coefficients, g.dev(values), g.dev(positions), n, len(positions),
                              uncompressed_size=usize, model=model, index_size=index_size, stream=g.stream())
        got = g.host(out, ud, n)
        bad = np.flatnonzero(got != expect)
        assert len(bad) == 0, (how, bad[:5], got[bad[:5]], expect[bad[:5]])
    # ... and with the positions compressed
    positions, values = _distinct(*_patches(n, "dense", ud, index_size, seed=n))
    if len(positions):
        compressed = _compressed(positions, values, 100, index_size, g, 5)
        out = g.dev_empty(n, ud)
        schemes.patched_model(out, coefficients, g.dev(values), compressed, n, len(positions),
                              uncompressed_size=usize, model=model, index_size=index_size, stream=g.stream())
        assert np.array_equal(g.host(out, ud, n), _expected(decoded, positions, values))


DZB_FIXED_TUPLES = [(2, 1), (4, 1), (4, 2), (8, 1), (8, 2), (8, 4)]


@pytest.mark.parametrize("usize,csize", DZB_FIXED_TUPLES)
@pytest.mark.parametrize("index_size", [4, 8])
def test_patched_discard_zero_bytes_fixed(usize, csize, index_size):
    schemes, r, g = _imports()
    ud, cd = np.dtype("u%d" % usize), np.dtype("u%d" % csize)
    for n in (1, 255, 4099, 70001, 1000003):
        narrow = cases.rng(n + usize).integers(0, np.iinfo(cd).max, size=n, endpoint=True).astype(cd)
        decoded = narrow.astype(ud)
        d_in = g.dev(narrow)
        for how in PATCH_SHAPES:
            positions, values = _patches(n, how, ud, index_size, seed=len(how) + n)
            expect = _expected(decoded, positions, values)
            out = g.dev_empty(n, ud)
            schemes.patched_discard_zero_bytes_fixed(out, d_in, g.dev(values), g.dev(positions), n, len(positions),
                                                     uncompressed_size=usize, compressed_size=csize, index_size=index_size, stream=g.stream())
            got = g.host(out, ud, n)
            bad = np.flatnonzero(got != expect)
            assert len(bad) == 0, (n, how, bad[:5], got[bad[:5]], expect[bad[:5]])
        # ... with the positions compressed, and from a displaced (unaligned) input
        positions, values = _distinct(*_patches(n, "sparse", ud, index_size, seed=n))
        if len(positions):
            compressed = _compressed(positions, values, 100, index_size, g, 5)
            out = g.dev_empty(n, ud)
            schemes.patched_discard_zero_bytes_fixed(out, d_in, g.dev(values), compressed, n, len(positions),
                                                     uncompressed_size=usize, compressed_size=csize, index_size=index_size, stream=g.stream())
            assert np.array_equal(g.host(out, ud, n), _expected(decoded, positions, values))
            padded = g.dev(np.concatenate([np.zeros(1, cd), narrow]))
            out = g.dev_empty(n, ud)
            schemes.patched_discard_zero_bytes_fixed(out, padded.data_ptr() + csize, g.dev(values), g.dev(positions), n, len(positions),
                                                     uncompressed_size=usize, compressed_size=csize, index_size=index_size, stream=g.stream())
            assert np.array_equal(g.host(out, ud, n), _expected(decoded, positions, values))


DICT_TUPLES = [(1, 1), (2, 1), (2, 2), (4, 1), (4, 2), (4, 4), (8, 1), (8, 2), (8, 4), (8, 8)]


@pytest.mark.parametrize("usize,dsize", DICT_TUPLES)
@pytest.mark.parametrize("index_size", [4, 8])
def test_patched_dictionary(usize, dsize, index_size):
    schemes, r, g = _imports()
    ud, kd = np.dtype("u%d" % usize), np.dtype("u%d" % dsize)
    for n, entries in [(1, 1), (1027, 256 if dsize == 1 else 3000), (1 << 20, 256 if dsize == 1 else 40000)]:
        entries = min(entries, 1 << (8 * dsize))
        f = cases.dictionary_case(n, entries, ud, kd, seed=n + usize)
        decoded = oracle.dictionary(f["compressed_input"], f["dictionary_entries"])
        d_in, d_dict = g.dev(f["compressed_input"]), g.dev(f["dictionary_entries"])
        for how in PATCH_SHAPES:
            positions, values = _patches(n, how, ud, index_size, seed=len(how) + n)
            expect = _expected(decoded, positions, values)
            for cache in (None, False):
                out = g.dev_empty(n, ud)
                schemes.patched_dictionary(out, d_in, d_dict, g.dev(values), g.dev(positions), n, entries, len(positions),
                                           uncompressed_size=usize, dictionary_index_size=dsize, index_size=index_size,
                                           cache_dictionary_in_shared_mem=cache, stream=g.stream())
                got = g.host(out, ud, n)
                bad = np.flatnonzero(got != expect)
                assert len(bad) == 0, (n, how, cache, bad[:5], got[bad[:5]], expect[bad[:5]])


@pytest.mark.parametrize("usize", [1, 2, 4, 8])
@pytest.mark.parametrize("how", ["dense", "duplicates", "all", "clustered"])
def test_many_patches_in_workgroups_larger_than_the_row_areas(usize, how):
    """The ManyPatches builds pass a row's patches to the lanes through a wave-private row area in LDS, provisioned for the four wavefronts of the
    workgroups the wrappers launch; a caller of the kernels may launch workgroups of up to 1024 threads: wavefronts 4 ... 15 find their patches
    by the search over the window instead.  Both forms in one launch, against the oracle."""
    schemes, r, g = _imports()
    ud = np.dtype("u%d" % usize)
    n = 700001
    narrow = cases.rng(n + usize).integers(0, 255, size=n, endpoint=True).astype(np.uint8)
    positions, values = _patches(n, how, ud, 4, seed=usize + len(how))
    d_values, d_positions = g.dev(values), g.dev(positions)

    def both(launch, decoded):
        expect = _expected(decoded, positions, values)
        out = g.dev_empty(n, ud)
        resolved = launch(out, None)
        assert np.array_equal(g.host(out, ud, n), expect)
        for threads, blocks in ((1024, 40), (512, 97), (320, 64)):
            out = g.dev_empty(n, ud)
            launch(out, r.launch_configuration_t((blocks, 1, 1), (threads, 1, 1), resolved.dynamic_shared_memory_size))
            got = g.host(out, ud, n)
            bad = np.flatnonzero(got != expect)
            assert len(bad) == 0, (threads, blocks, bad[:5], got[bad[:5]], expect[bad[:5]])

    if usize > 1:
        d_in = g.dev(narrow)
        both(lambda out, config: schemes.patched_discard_zero_bytes_fixed(out, d_in, d_values, d_positions, n, len(positions), uncompressed_size=usize, compressed_size=1,
                                                                           index_size=4, stream=g.stream(), launch_config=config), narrow.astype(ud))
    f = cases.dictionary_case(n, 256, ud, np.dtype("u1"), seed=n + usize)
    d_keys, d_dict = g.dev(f["compressed_input"]), g.dev(f["dictionary_entries"])
    both(lambda out, config: schemes.patched_dictionary(out, d_keys, d_dict, d_values, d_positions, n, 256, len(positions), uncompressed_size=usize, dictionary_index_size=1,
                                                         index_size=4, stream=g.stream(), launch_config=config), oracle.dictionary(f["compressed_input"], f["dictionary_entries"]))
    if usize in (2, 4):
        sd = np.dtype("i%d" % usize)
        coefficients = np.array([-12345, 37], dtype=sd)
        decoded = oracle.model(coefficients, n, sd, "linear", model_index_size=4)
        signed_values = values.view(sd)
        d_signed = g.dev(signed_values)

        def model_both():
            expect = _expected(decoded, positions, signed_values)
            out = g.dev_empty(n, sd)
            resolved = schemes.patched_model(out, coefficients, d_signed, d_positions, n, len(positions), uncompressed_size=usize, model="linear", index_size=4, stream=g.stream())
            assert np.array_equal(g.host(out, sd, n), expect)
            out = g.dev_empty(n, sd)
            schemes.patched_model(out, coefficients, d_signed, d_positions, n, len(positions), uncompressed_size=usize, model="linear", index_size=4, stream=g.stream(),
                                  launch_config=r.launch_configuration_t((40, 1, 1), (1024, 1, 1), resolved.dynamic_shared_memory_size))
            assert np.array_equal(g.host(out, sd, n), expect)
        model_both()


def test_patched_kernels_equal_the_two_launch_pipeline_on_the_golden_fixture():
    """C3 in miniature, fused: same golden fixture as the two-launch pipeline test"""
    import os
    schemes, r, g = _imports()
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "delta_u64_u16_patched.npz"))
    n, seglen = int(fx["length"]), int(fx["segment_length"])
    order = np.argsort(fx["patch_indices"], kind="stable")
    out = g.dev_empty(n, np.uint64)
    schemes.patched_delta(out, g.dev(fx["compressed_input"]), g.dev(fx["anchor_values"]), g.dev(fx["patch_data"][order]),
                          g.dev(fx["patch_indices"][order]), n, seglen, len(order), stream=g.stream())
    assert (g.host(out, np.uint64, n) == fx["expected_patched"]).all()


def test_patched_unaligned_bases_and_limits():
    """scalar fall-backs (misaligned pointers) and tiny one-time launch limits patch correctly too"""
    schemes, r, g = _imports()
    n, seglen = 200001, 1024
    f, truth = cases.delta_case(n + 8, seglen, np.uint32, np.uint16, seed=9)
    positions, values = _patches(n, "dense", np.uint32, 4, seed=10)
    d_pos, d_val = g.dev(positions), g.dev(values)
    # Delta: input 2 bytes off its natural vector alignment -> the general path
    comp = f["compressed_input"]
    expect = _expected(oracle.delta(comp[1:n + 1], f["anchor_values"], seglen), positions, values)
    out = g.dev_empty(n, np.uint32)
    d_comp = g.dev(comp)   # (kept alive: a raw address does not hold the buffer)
    schemes.patched_delta(out, d_comp.data_ptr() + 2, g.dev(f["anchor_values"]), d_val, d_pos, n, seglen, len(positions),
                          uncompressed_size=4, compressed_size=2, stream=g.stream())
    assert (g.host(out, np.uint32, n) == expect).all()
    for limits in (r.launch_configuration_limits_t(block_size=64), r.launch_configuration_limits_t(grid_size=3),
                   r.launch_configuration_limits_t(block_size=128, grid_size=1)):
        expect = _expected(truth[:n], positions, values)
        out = g.dev_empty(n, np.uint32)
        schemes.patched_delta(out, g.dev(comp[:n]), g.dev(f["anchor_values"]), d_val, d_pos, n, seglen, len(positions),
                              uncompressed_size=4, compressed_size=2, stream=g.stream(), limits=limits)
        assert (g.host(out, np.uint32, n) == expect).all()
    # Dictionary: keys one byte off -> everything through the scalar tail with per-element searches
    d = cases.dictionary_case(n + 4, 256, np.uint32, np.uint8, seed=11)
    keys = d["compressed_input"]
    expect = _expected(oracle.dictionary(keys[1:n + 1], d["dictionary_entries"]), positions, values)
    out = g.dev_empty(n, np.uint32)
    d_keys = g.dev(keys)
    schemes.patched_dictionary(out, d_keys.data_ptr() + 1, g.dev(d["dictionary_entries"]), d_val, d_pos, n, 256, len(positions),
                               stream=g.stream())
    assert (g.host(out, np.uint32, n) == expect).all()
    for limits in (r.launch_configuration_limits_t(block_size=64, grid_size=2),):
        expect = _expected(oracle.dictionary(keys[:n], d["dictionary_entries"]), positions, values)
        out = g.dev_empty(n, np.uint32)
        schemes.patched_dictionary(out, g.dev(keys[:n]), g.dev(d["dictionary_entries"]), d_val, d_pos, n, 256, len(positions),
                                   stream=g.stream(), limits=limits)
        assert (g.host(out, np.uint32, n) == expect).all()


@pytest.mark.parametrize("usize,csize", DELTA_TUPLES)
def test_patched_delta_fuzz(usize, csize):
    """seeded random lengths, segment lengths of every kind (tiny, not whole rows, longer than the column) and patch densities"""
    schemes, r, g = _imports()
    rnd = cases.rng(usize * 10 + csize)
    ud, cd = np.dtype("u%d" % usize), np.dtype("u%d" % csize)
    for _ in range(40):
        n = int(rnd.integers(1, 60000))
        seglen = int(rnd.choice([int(rnd.integers(1, 40)), int(rnd.choice([64, 128, 1000, 1024, 3000, 4096, 8192, 16384])),
                                 int(rnd.integers(40, 20000)), n + int(rnd.integers(0, 5))]))
        how = str(rnd.choice(PATCH_SHAPES))
        f, truth = cases.delta_case(n, seglen, ud, cd, seed=n)
        positions, values = _patches(n, how, ud, 4, seed=n + 1)
        expect = _expected(truth, positions, values)
        out = g.dev_empty(n, ud)
        schemes.patched_delta(out, g.dev(f["compressed_input"]), g.dev(f["anchor_values"]), g.dev(values), g.dev(positions), n, seglen,
                              len(positions), uncompressed_size=usize, compressed_size=csize, stream=g.stream())
        got = g.host(out, ud, n)
        bad = np.flatnonzero(got != expect)
        assert len(bad) == 0, (n, seglen, how, bad[:5], got[bad[:5]], expect[bad[:5]])


# ---- patch positions in the compressed-indices form ------------------------------------------------

def _compressed(positions, values, period, index_size, g, max_run_length=None):
    """formats.compressed_indices on the device, plus the scratch the fused decoders expand the positions into"""
    from libgiddy_amd import formats
    c = formats.compressed_indices(positions, values, period, output_index_size=index_size, input_index_size=index_size,
                                   max_run_length=max_run_length)
    on_device = {k: (g.dev(v) if isinstance(v, np.ndarray) else v) for k, v in c.items()}
    on_device["patch_positions_scratch"] = g.dev_empty(len(positions), np.dtype("u%d" % index_size))
    return on_device


def _distinct(positions, values):
    keep = np.concatenate([positions[1:] != positions[:-1], [True]]) if len(positions) else np.zeros(0, bool)
    return positions[keep], values[keep]


@pytest.mark.parametrize("output_index_size,input_index_size", [(4, 4), (8, 8), (4, 2), (8, 4), (4, 1), (4, 8)])
@pytest.mark.parametrize("n_patches,period,max_run", [(1, 1, None), (200, 64, None), (100000, 1024, None), (100000, 100, 3), (65000, 65000, 1), (5000, 7, None)])
def test_compressed_positions_expand_to_the_plain_ones(output_index_size, input_index_size, n_patches, period, max_run):
    schemes, r, g = _imports()
    from libgiddy_amd import formats
    if n_patches > np.iinfo("u%d" % input_index_size).max:
        n_patches, period = 200, min(period, 100)
    rng = cases.rng(n_patches + period)
    span = min(1 << 40 if output_index_size == 8 else 1 << 31, n_patches * 3000)
    positions = np.unique(rng.integers(0, span, size=n_patches, dtype=np.uint64))
    c = formats.compressed_indices(positions, np.zeros(len(positions), np.uint8), period, output_index_size=output_index_size,
                                   input_index_size=input_index_size, max_run_length=max_run)
    if c["num_scatter_position_runs"] > np.iinfo("u%d" % input_index_size).max:
        pytest.skip("more runs than the index type counts")
    on_device = {k: (g.dev(v) if isinstance(v, np.ndarray) else v) for k, v in c.items()}
    out = g.dev_empty(len(positions), np.dtype("u%d" % output_index_size))
    schemes.compressed_indices_positions(out, on_device, len(positions), output_index_size=output_index_size,
                                         input_index_size=input_index_size, stream=g.stream())
    got = g.host(out, np.dtype("u%d" % output_index_size), len(positions))
    assert np.array_equal(got, positions.astype(got.dtype))
    from oracle import naive
    assert np.array_equal(got, naive.compressed_indices_positions(
        len(positions), c["scatter_position_run_lengths"], c["scatter_position_run_individual_offset_sizes"],
        c["scatter_position_run_baseline_values"], c["scatter_position_run_offsets_start_pos"],
        c["scatter_position_offset_bytes"], c["scatter_position_anchors"], period, output_index_size))


@pytest.mark.parametrize("index_size", [4, 8])
@pytest.mark.parametrize("how", ["sparse", "dense", "edges", "clustered", "all"])
@pytest.mark.parametrize("period,max_run", [(1024, None), (100, 5)])
def test_fused_decoders_take_compressed_patch_positions(index_size, how, period, max_run):
    schemes, r, g = _imports()
    n, seglen = 300007, 1024
    # Delta
    ud, cd = np.dtype("u4"), np.dtype("u1")
    f, truth = cases.delta_case(n, seglen, ud, cd, seed=11)
    positions, values = _distinct(*_patches(n, how, ud, index_size, seed=5))
    expect = _expected(truth, positions, values)
    compressed = _compressed(positions, values, period, index_size, g, max_run)
    out = g.dev_empty(n, ud)
    schemes.patched_delta(out, g.dev(f["compressed_input"]), g.dev(f["anchor_values"]), g.dev(values), compressed, n, seglen, len(positions),
                          uncompressed_size=4, compressed_size=1, index_size=index_size, stream=g.stream())
    assert np.array_equal(g.host(out, ud, n), expect)
    assert np.array_equal(g.host(compressed["patch_positions_scratch"], np.dtype("u%d" % index_size), len(positions)), positions)
    # Frame of Reference
    sd, scd = np.dtype("i4"), np.dtype("i2")
    f = cases.for_case(n, seglen, sd, scd, "linear", seed=12)
    decoded = oracle.frame_of_reference(f["compressed_input"], f["segment_model_coefficients"], n, seglen, sd, "linear", model_index_size=4)
    positions, values = _distinct(*_patches(n, how, sd, index_size, seed=6))
    expect = _expected(decoded, positions, values)
    compressed = _compressed(positions, values, period, index_size, g, max_run)
    out = g.dev_empty(n, sd)
    schemes.patched_frame_of_reference(out, g.dev(f["compressed_input"]), g.dev(f["segment_model_coefficients"]), g.dev(values), compressed,
                                       n, seglen, len(positions), uncompressed_size=4, compressed_size=2, model="linear",
                                       index_size=index_size, model_index_size=4, stream=g.stream())
    assert np.array_equal(g.host(out, sd, n), expect)
    # Dictionary
    kd = np.dtype("u1")
    f = cases.dictionary_case(n, 256, ud, kd, seed=13)
    decoded = oracle.dictionary(f["compressed_input"], f["dictionary_entries"])
    positions, values = _distinct(*_patches(n, how, ud, index_size, seed=7))
    expect = _expected(decoded, positions, values)
    compressed = _compressed(positions, values, period, index_size, g, max_run)
    out = g.dev_empty(n, ud)
    schemes.patched_dictionary(out, g.dev(f["compressed_input"]), g.dev(f["dictionary_entries"]), g.dev(values), compressed,
                               n, 256, len(positions), uncompressed_size=4, dictionary_index_size=1, index_size=index_size, stream=g.stream())
    assert np.array_equal(g.host(out, ud, n), expect)


def test_fused_decoder_with_no_compressed_patches_launches_no_expansion():
    schemes, r, g = _imports()
    n, seglen = 5000, 1000
    ud, cd = np.dtype("u4"), np.dtype("u2")
    f, truth = cases.delta_case(n, seglen, ud, cd, seed=3)
    empty = np.zeros(1, np.uint32)
    compressed = {name: g.dev(empty.astype(np.uint8 if "bytes" in name or "sizes" in name else np.uint32))
                  for name, _ in schemes._COMPRESSED_POSITION_ARRAYS}
    compressed.update(anchoring_period=64, num_scatter_position_runs=0, patch_positions_scratch=g.dev(empty))
    out = g.dev_empty(n, ud)
    schemes.patched_delta(out, g.dev(f["compressed_input"]), g.dev(f["anchor_values"]), g.dev(empty), compressed, n, seglen, 0,
                          uncompressed_size=4, compressed_size=2, index_size=4, stream=g.stream())
    assert np.array_equal(g.host(out, ud, n), truth)


def test_expanded_positions_of_runs_the_scatter_skips_are_all_ones():
    """a run whose offset width is none of 8/16/32/64 bits is skipped by the scatter (compressed_indices_scatter.cuh:93-118);
    its patches' positions come out as all-ones, exactly as the oracle's restatement has them"""
    schemes, r, g = _imports()
    from libgiddy_amd import formats
    from oracle import naive
    rng = cases.rng(77)
    positions = np.unique(rng.integers(0, 1 << 24, size=5000, dtype=np.uint64))
    c = formats.compressed_indices(positions, np.zeros(len(positions), np.uint8), 256, max_run_length=7)
    widths = c["scatter_position_run_individual_offset_sizes"].copy()
    widths[::5] = 24
    c["scatter_position_run_individual_offset_sizes"] = widths
    on_device = {k: (g.dev(v) if isinstance(v, np.ndarray) else v) for k, v in c.items()}
    out = g.dev_empty(len(positions), np.dtype("u4"))
    schemes.compressed_indices_positions(out, on_device, len(positions), stream=g.stream())
    got = g.host(out, np.dtype("u4"), len(positions))
    expect = naive.compressed_indices_positions(
        len(positions), c["scatter_position_run_lengths"], widths, c["scatter_position_run_baseline_values"],
        c["scatter_position_run_offsets_start_pos"], c["scatter_position_offset_bytes"], c["scatter_position_anchors"], 256, 4)
    assert np.array_equal(got, expect)
    assert (got == 0xFFFFFFFF).sum() > 0 and (got != 0xFFFFFFFF).sum() > 0
    # ... and a caller who passes "unsupported_patch_count" (a zeroed device counter) learns how many patches that is
    import torch
    counter = torch.zeros(1, dtype=torch.int32, device="cuda")
    schemes.compressed_indices_positions(out, dict(on_device, unsupported_patch_count=counter), len(positions), stream=g.stream())
    torch.cuda.synchronize()
    assert int(counter[0]) == int((got == 0xFFFFFFFF).sum())
    counter.zero_()
    clean = {k: (g.dev(v) if isinstance(v, np.ndarray) else v) for k, v in formats.compressed_indices(positions, np.zeros(len(positions), np.uint8), 256).items()}
    schemes.compressed_indices_positions(out, dict(clean, unsupported_patch_count=counter), len(positions), stream=g.stream())
    torch.cuda.synchronize()
    assert int(counter[0]) == 0 and np.array_equal(g.host(out, np.dtype("u4"), len(positions)), positions.astype(np.uint32))


def test_fused_decoder_with_compressed_positions_is_graph_capturable():
    """the expansion of the positions and the decoder are two plain launches on the caller's stream: captured into a HIP
    graph and replayed they give the same bits (no hidden synchronisation or allocation)"""
    schemes, r, g = _imports()
    n, L = 1 << 20, 4096
    ud = np.dtype("u8")
    f, truth = cases.delta_case(n, L, ud, np.uint16, seed=321)
    positions, values = _distinct(*_patches(n, "sparse", ud, 4, seed=9))
    expect = _expected(truth, positions, values)
    compressed = _compressed(positions, values, 256, 4, g)
    d_in, d_anchor, d_val = g.dev(f["compressed_input"]), g.dev(f["anchor_values"]), g.dev(values)
    out = g.dev_empty(n, ud)
    side = torch.cuda.Stream()
    kw = dict(uncompressed_size=8, compressed_size=2, index_size=4)
    with torch.cuda.stream(side):
        cfg = schemes.patched_delta(out, d_in, d_anchor, d_val, compressed, n, L, len(positions), stream=side.cuda_stream, **kw)   # resolve outside the capture
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        schemes.patched_delta(out, d_in, d_anchor, d_val, compressed, n, L, len(positions),
                              stream=torch.cuda.current_stream().cuda_stream, launch_config=cfg, **kw)
    out.fill_(0xCD)
    compressed["patch_positions_scratch"].fill_(0)
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(g.host(out, ud, n), expect)


def test_the_two_launch_rule_is_what_the_wrappers_document():
    """decode_then_scatter_is_faster (patched/decode_then_scatter.hpp), restated: the parity tests above cover both ways only if their shapes fall on both
    sides of it — 'sparse' (0.1 %), 'repeats_at_one_percent', 'duplicates' (2 %), 'dense' (35 %) and 'all' do for every 2-byte family."""
    def two_launches(family, usize, density, staged=0, many=False):
        if family == "dictionary" and usize == 8 and many and staged >= 16384:
            return True
        if family == "dictionary" and usize == 8 and not many and staged >= 8192 and density >= 0.008:
            return True
        if usize != 2:
            return False
        return {"delta": 0.0015 <= density < 0.015 or density >= 0.15, "frame_of_reference": 0.004 <= density < 0.015 or density >= 0.35,
                "model": density < 0.004 or density >= 0.35, "dictionary": density >= 0.35, "discard_zero_bytes_fixed": False}[family]
    for family in ("delta", "frame_of_reference", "model", "dictionary"):
        ways = {two_launches(family, 2, d) for d in (0.001, 0.01, 0.02, 0.36, 1.0)}
        assert ways == {True, False}, family
    assert two_launches("dictionary", 8, 0.02, staged=3000 * 8, many=True) and not two_launches("dictionary", 8, 0.02, staged=256 * 8, many=True)

"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol include/giddy.h
declares, the factory holds the reference's 163 registrations under the reference's key strings, the
launch-configuration resolver behaves (and fails) like the reference's, and argument maps are
type-checked.  No kernel is launched here."""
import ctypes
import os
import re

import pytest

from libgiddy_amd import _native, registered as r, schemes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = r.NOMINAL_MI355X


def test_every_declared_symbol_is_exported():
    header = open(os.path.join(ROOT, "include", "giddy.h")).read()
    declared = set(re.findall(r"\b(giddy_[a-z_]+)\s*\(", header))
    assert declared == set(_native.EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(_native.LIBRARY_PATH)
    for symbol in declared:
        assert hasattr(lib, symbol), symbol
    assert _native.lib.giddy_version().decode().startswith("libgiddy_amd")


def test_factory_has_the_163_reference_registrations():
    all_keys = r.kernel_t.registered_keys()
    assert len(set(all_keys)) == len(all_keys)
    # the fused decompress-and-patch wrappers are this library's extension, under keys of their own
    extensions = [k for k in all_keys if "::patched::" in k]
    assert sorted(extensions) == sorted(schemes.key_patched(k) for k in all_keys
                                        if k.startswith(("decompression::delta::kernel_t", "decompression::dictionary::kernel_t",
                                                         "decompression::frame_of_reference::kernel_t", "decompression::model::kernel_t",
                                                         "decompression::discard_zero_bytes::fixed_width::kernel_t")))
    assert len(extensions) == 12 + 20 + 8 + 5 + 12
    # ... and so is the stand-alone expansion of compressed patch positions the fused wrappers use
    expansions = [k for k in all_keys if "::positions::" in k]
    assert sorted(expansions) == sorted(schemes.key_compressed_indices_positions(o, i) for o in (4, 8) for i in (1, 2, 4, 8))
    # ... and the device-side compressors (the reference has none), under "compression::..."
    compressors = [k for k in all_keys if k.startswith("compression::")]
    assert len(compressors) == 112
    keys = [k for k in all_keys if "::patched::" not in k and "::positions::" not in k and not k.startswith("compression::")]
    assert len(keys) == 163
    counts = {}
    for k in keys:
        counts[k.split("::kernel_t")[0]] = counts.get(k.split("::kernel_t")[0], 0) + 1
    # SURVEY.md Appendix B
    assert counts == {"decompression::delta": 12, "decompression::dictionary": 20,
                      "decompression::discard_zero_bytes::fixed_width": 12, "decompression::discard_zero_bytes::variable_width": 9,
                      "decompression::frame_of_reference": 8, "decompression::incidence_bitmaps": 5, "decompression::model": 5,
                      "decompression::run_length_encoding": 18, "decompression::run_position_encoding": 10,
                      "scatter": 32, "scatter::compressed_indices": 32}
    # the key the reference's own example uses (factory_provided_type_erased_wrapper.cu:14-15)
    assert r.kernel_t.canProduce("decompression::frame_of_reference::kernel_t<4u, int, short, "
                                 "cuda::functors::unary::parametric_model::constant<4u, int> >")
    # every key builder in schemes.py produces registered keys
    for key in [schemes.key_dictionary(4, 4, 1), schemes.key_delta(8, 8, 4), schemes.key_discard_zero_bytes_fixed(4, 8, 2),
                schemes.key_discard_zero_bytes_variable(8, 8, 4), schemes.key_incidence_bitmaps(4, 1, 2, False),
                schemes.key_model(8, 4, "linear"), schemes.key_model(4, 2, "constant"), schemes.key_run_length_encoding(8, 1, 8),
                schemes.key_run_position_encoding(4, 8, 2, True), schemes.key_run_position_encoding(8, 4, 8, False),
                schemes.key_scatter(8, 1, 2), schemes.key_compressed_indices_scatter(4, 8, 1),
                schemes.key_frame_of_reference(8, 4, 2, "linear", 8)]:
        assert r.kernel_t.canProduce(key), key


def test_expansion_of_compressed_positions_resolves_without_a_device_query():
    k = schemes.kernel(schemes.key_compressed_indices_positions(4, 4))
    c = k.resolve_launch_configuration(N, {"input_data_length": r.size_t(1 << 20), "anchoring_period": r.size_t(1024)})
    assert c.block_dimensions == (256, 1, 1) and c.grid_dimensions[0] == 256    # a wavefront per block of patches
    with pytest.raises(r.InvalidArgument):
        k.resolve_launch_configuration(N, {"input_data_length": r.size_t(0), "anchoring_period": r.size_t(1024)})
    with pytest.raises(r.InvalidArgument):
        k.resolve_launch_configuration(N, {"input_data_length": r.size_t(10), "anchoring_period": r.size_t(0)})


def test_the_build_entry_point_accepts_the_library_as_built():
    """what __graft_entry__.build() asserts after compiling (kept in step with the registrations)"""
    import __graft_entry__
    __graft_entry__.check_library()


def test_unknown_key_is_invalid_argument():
    with pytest.raises(r.InvalidArgument, match="No subclass of the base type cuda::registered::kernel_t is registered with key"):
        r.kernel_t.produceSubclass("decompression::nope::kernel_t<4u>")


def test_resolver_shapes():
    k = schemes.kernel(schemes.key_dictionary(4, 4, 1))
    c = k.resolve_launch_configuration(N, {"num_dictionary_entries": r.size_t(256), "data_length": r.size_t(1 << 30)})
    assert c.block_dimensions == (256, 1, 1) and c.grid_dimensions[0] == (1 << 30) // (256 * 16) and c.dynamic_shared_memory_size == 26624   # (the 1 KiB dictionary, padded so that 6 workgroups share a CU's 160 KiB: the occupancy throttle)
    # a dictionary too large for LDS falls back to look-ups through the caches
    c = k.resolve_launch_configuration(N, {"num_dictionary_entries": r.size_t(1 << 20), "data_length": r.size_t(1 << 20)})
    assert c.dynamic_shared_memory_size == 0
    # explicit opt-out
    c = k.resolve_launch_configuration(N, {"num_dictionary_entries": r.size_t(256), "data_length": r.size_t(1 << 20),
                                           "cache_input_data_in_shared_mem": r.boolean(False)})
    assert c.dynamic_shared_memory_size == 0
    # small inputs: no more threads than the input can occupy, rounded to whole wavefronts
    c = k.resolve_launch_configuration(N, {"num_dictionary_entries": r.size_t(4), "data_length": r.size_t(100)})
    assert c.block_dimensions[0] == 64 and c.grid_dimensions[0] == 1
    # empty input: an empty grid, which enqueue treats as a no-op
    c = k.resolve_launch_configuration(N, {"num_dictionary_entries": r.size_t(4), "data_length": r.size_t(0)})
    assert c.grid_dimensions[0] == 0

    d = schemes.kernel(schemes.key_delta(4, 8, 2))
    c = d.resolve_launch_configuration(N, {"length": r.size_t(1 << 30), "segment_length": r.size_t(4096)})
    assert c.block_dimensions[0] == 256 and c.grid_dimensions[0] == (1 << 30) // 4096 // 4          # one wavefront per segment
    c = d.resolve_launch_configuration(N, {"length": r.size_t(1 << 30), "segment_length": r.size_t(1 << 20)})
    assert c.block_dimensions[0] == 1024 and c.grid_dimensions[0] == 1024                            # one workgroup per long segment
    c = d.resolve_launch_configuration(N, {"length": r.size_t(1 << 30), "segment_length": r.size_t(128)})
    assert c.block_dimensions[0] == 256 and c.grid_dimensions[0] == (1 << 30) // (32 * 128) // 4     # 32 one-row segments per wavefront
    c = d.resolve_launch_configuration(N, {"length": r.size_t(1 << 30), "segment_length": r.size_t(8192)})
    assert c.block_dimensions[0] == 512 and c.grid_dimensions[0] == (1 << 30) // 8192                # 64 rows: 8 wavefronts sweep a segment


def test_one_time_limits():
    k = schemes.kernel(schemes.key_frame_of_reference(4, 4, 2))
    args = {"length": r.size_t(1 << 24), "segment_length": r.size_t(4096)}
    c = k.resolve_launch_configuration(N, args, r.launch_configuration_limits_t(block_size=128))
    assert c.block_dimensions[0] == 128
    c = k.resolve_launch_configuration(N, args, r.launch_configuration_limits_t(grid_size=7))
    assert c.grid_dimensions[0] == 7
    with pytest.raises(r.InvalidArgument, match="do not allow any threads in a block"):
        k.resolve_launch_configuration(N, args, r.launch_configuration_limits_t(block_size=32))   # less than a wavefront


def test_reference_error_behaviour():
    k = schemes.kernel(schemes.key_frame_of_reference(4, 4, 2))
    with pytest.raises(r.InvalidArgument, match="segment length must be a positive number"):
        k.resolve_launch_configuration(N, {"length": r.size_t(100), "segment_length": r.size_t(0)})
    with pytest.raises(r.InvalidArgument, match="aligned with the GPU's native word size"):
        k.resolve_launch_configuration(N, {"length": r.size_t(100), "segment_length": r.size_t(6)})
    # the reference cannot launch a single segment longer than a block (frame_of_reference.cuh:111-113); here it resolves
    c = k.resolve_launch_configuration(N, {"length": r.size_t(1 << 20), "segment_length": r.size_t(1 << 20)})
    assert c.grid_dimensions[0] == (1 << 20) // (256 * 8)   # 256 threads, two 16-byte stores (of 4 values) each per tile
    cis = schemes.kernel(schemes.key_compressed_indices_scatter(4, 4, 4))
    with pytest.raises(r.InvalidArgument, match="Zero-length scatters not currently supported"):
        cis.resolve_launch_configuration(N, {"input_data_length": r.size_t(0), "anchoring_period": r.size_t(1024)})
    # missing key -> out_of_range, wrong C++ type -> bad_any_cast (kernel_wrapper.cuh:47-49)
    with pytest.raises(r.OutOfRange):
        k.resolve_launch_configuration(N, {"length": r.size_t(100)})
    with pytest.raises(r.BadAnyCast):
        k.resolve_launch_configuration(N, {"length": r.uint32(100), "segment_length": r.size_t(4)})
    # a serialization factor on a kernel that does not serialize per request is accepted only for Dictionary
    d = schemes.kernel(schemes.key_dictionary(4, 4, 1))
    d.resolve_launch_configuration(N, {"num_dictionary_entries": r.size_t(4), "data_length": r.size_t(100),
                                       "serialization_factor": r.serialization_factor_t(16)})


def test_untyped_arguments_are_rejected_before_reaching_c():
    k = schemes.kernel(schemes.key_delta(4, 8, 2))
    with pytest.raises(TypeError, match="typed constructors"):
        k.resolve_launch_configuration(N, {"length": 100, "segment_length": 4096})


def test_occupancy_throttle_is_an_lds_reservation_a_caller_can_refuse():
    """resident_blocks_per_multiprocessor_cap: Model reserves dynamic LDS it never touches, so that only a few
    workgroups share a CU; launch_configuration_limits_t::dynamic_shared_memory takes it back; the headline Dictionary has none"""
    k = schemes.kernel(schemes.key_model(4, 4, "linear"))
    arguments = {"length": r.size_t(1 << 24)}
    throttled = k.resolve_launch_configuration(r.NOMINAL_MI355X, arguments, None)
    assert throttled.dynamic_shared_memory_size >= 32 * 1024
    refused = k.resolve_launch_configuration(r.NOMINAL_MI355X, arguments, r.launch_configuration_limits_t(dynamic_shared_memory=0))
    assert refused.dynamic_shared_memory_size == 0
    assert refused.grid_dimensions == throttled.grid_dimensions and refused.block_dimensions == throttled.block_dimensions
    d = schemes.kernel(schemes.key_dictionary(4, 4, 1))
    config = d.resolve_launch_configuration(r.NOMINAL_MI355X, {"num_dictionary_entries": r.size_t(256), "data_length": r.size_t(1 << 24)}, None)
    assert config.dynamic_shared_memory_size == 26624   # (1 KiB of dictionary + the occupancy throttle's padding: 6 workgroups per CU)


def test_every_build_of_a_run_decoder_accepts_the_configuration_resolved_for_the_base_kernel():
    """get_device_function() of the run-length / run-position wrappers names `decompress` — the kernel of the reference's signature — while
    their typed launch may run `decompress_sparse_runs`, `decompress_dense_runs` or `decompress_relative_with_finer_anchors` with the
    configuration the caller resolved.  Read from the code objects' metadata, without a GPU: every such build accepts at
    least the workgroup sizes the base kernel accepts, and its static LDS leaves the base kernel's dynamic LDS room — so a configuration
    that is valid for the base kernel launches whichever build is picked.  (At run time the wrappers go further: attributes() merges all builds'
    limits — tests/test_gpu_modes.py.)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import code_object_metadata
    kernels = code_object_metadata.kernels_of(os.path.join(ROOT, "libgiddy_amd", "libgiddy_amd.so"))
    assert len(kernels) > 500, len(kernels)
    families = 0
    for name, base in kernels.items():
        m = re.match(r"void cuda::kernels::decompression::(run_length_encoding|run_position_encoding)::decompress<([^>]*)>\(", name)
        if not m:
            continue
        scheme, arguments = m.groups()
        builds = ["decompress_sparse_runs<%s>" % arguments, "decompress_dense_runs<%s>" % arguments]
        parts = [a.strip() for a in arguments.split(",")]
        if scheme == "run_position_encoding" and parts[3] == "true":
            builds.append("decompress_relative_with_finer_anchors<%s>" % ", ".join(parts[:3] + parts[4:]))
        for build in builds:
            prefix = "void cuda::kernels::decompression::%s::%s(" % (scheme, build)
            found = [record for other, record in kernels.items() if other.startswith(prefix)]
            assert len(found) == 1, (prefix, len(found))
            alternate = found[0]
            assert alternate["max_flat_workgroup_size"] >= base["max_flat_workgroup_size"], (name, build)
            assert alternate["group_segment_fixed_size"] <= 65536 - 24576, (build, alternate["group_segment_fixed_size"])   # (room for the resolver's largest LDS reservation)
            assert alternate["private_segment_fixed_size"] == 0, (build, "scratch memory: the build spills vector registers")
        families += 1
    assert families == 28, families   # 18 run-length + 10 run-position registrations


def test_round_4_entry_points_fail_politely_without_a_device():
    """giddy_prepare / giddy_enqueue_prepared / the scratch entry points: null handles and bad policies are status codes with a message, and the
    pool is empty in a process that never launched anything (no device is touched here)."""
    lib = _native.lib
    prepared = ctypes.c_void_p()
    assert lib.giddy_prepare(None, None, 0, ctypes.byref(prepared)) == -1 and b"null kernel handle" in lib.giddy_last_error()
    assert lib.giddy_enqueue_prepared(None, None, None) == -1
    assert lib.giddy_prepared_update(None, None, 0) == -1
    lib.giddy_release_prepared(None)
    assert lib.giddy_set_capture_scratch_policy(7) == -1
    assert lib.giddy_set_capture_scratch_policy(0) == 0
    assert lib.giddy_scratch_held_bytes() == 0 and lib.giddy_trim() == 0
    kernel = r.kernel_t.produceSubclass(schemes.key_run_length_encoding(4, 4, 2, 4))
    assert lib.giddy_device_function_count(kernel._handle) == 4          # the kernel of the reference's signature + the builds for short / long runs + long segments in one kernel
    assert len(set(kernel.device_functions())) == 4 and kernel.device_functions()[0] == kernel.get_device_function()
    assert lib.giddy_device_function_count(r.kernel_t.produceSubclass(schemes.key_dictionary(4, 4, 1))._handle) == 2   # ... + the build for columns at any element offset
    assert lib.giddy_device_function_count(r.kernel_t.produceSubclass(schemes.key_frame_of_reference(4, 4, 2, "constant", 4, 4))._handle) == 2
    assert lib.giddy_device_function_count(r.kernel_t.produceSubclass(schemes.key_discard_zero_bytes_fixed(4, 8, 2))._handle) == 2
    relative = r.kernel_t.produceSubclass("decompression::run_position_encoding::kernel_t<4u, 4u, 2u, true, 4u>")
    assert lib.giddy_device_function_count(relative._handle) == 5        # ... + the one over finer anchors + long segments in one kernel
    # a prepared launch on a real wrapper, map and all — nothing is launched
    arguments = {"length": r.size_t(1 << 20), "segment_length": r.size_t(4096)}
    handle = kernel.prepare(arguments)
    handle.update({"length": r.size_t(1 << 21)})
    del handle


def test_batches_are_built_on_the_host_alone():
    """giddy_batch_* (include/giddy/batch.hpp) with the nominal device: pieces are added and counted without touching a GPU — a piece brings the workgroups
    its own resolved configuration has, an empty piece brings none —, the wrappers that launch batches say so, the others refuse, a piece's argument map is
    checked like a launch's, a batch for the nominal device cannot be uploaded, and null handles are status codes."""
    lib = _native.lib
    lengths = (0, 1, 4097, 1 << 20, 300000)
    for key, resolve, launch in (
            (schemes.key_dictionary(4, 4, 1), lambda n: {"num_dictionary_entries": r.size_t(256), "data_length": r.size_t(n)},
             lambda n: {"decompressed": r.pointer(0x10000000, "u32"), "compressed_input": r.const_pointer(0x20000000, "u8"), "dictionary_entries": r.const_pointer(0x30000000, "u32"),
                        "length": r.uint32(n), "num_dictionary_entries": r.uint32(256)}),
            (schemes.key_run_length_encoding(4, 4, 2, 4), lambda n: {"length": r.size_t(n), "segment_length": r.size_t(4096)},
             lambda n: {"decompressed": r.pointer(0x10000000, "u32"), "run_data": r.const_pointer(0x20000000, "u32"), "run_lengths": r.const_pointer(0x30000000, "u16"),
                        "position_anchors": r.const_pointer(0x40000000, "u32"), "intra_run_anchor_offsets": r.const_pointer(0x50000000, "u16"),
                        "segment_length": r.uint32(4096), "num_segments": r.uint32(-(-n // 4096)), "num_element_runs": r.uint32(max(1, n // 32)), "length": r.uint32(n)})):
        kernel = r.kernel_t.produceSubclass(key)
        assert kernel.launches_batches()
        batch = kernel.batch(r.NOMINAL_MI355X)
        expected_blocks = 0
        for n in lengths:
            batch.add(launch(n), resolve(n))
            expected_blocks += kernel.resolve_launch_configuration(r.NOMINAL_MI355X, resolve(n)).grid_dimensions[0]
        assert batch.num_pieces == len(lengths) and batch.num_blocks == expected_blocks > 0
        with pytest.raises(r.LogicError):
            batch.upload(0)
        with pytest.raises(r.LogicError):
            batch.tables()
        with pytest.raises(r.OutOfRange):      # a required argument is missing: as enqueue_launch
            batch.add({"decompressed": r.pointer(0x10000000, "u32")}, resolve(100))
        with pytest.raises(r.BadAnyCast):      # ... or holds another type
            wrong = launch(100); wrong["length"] = r.uint64(100)
            batch.add(wrong, resolve(100))
        assert batch.num_pieces == len(lengths)   # (the failed pieces were not added)
        batch.clear()
        assert batch.num_pieces == 0 and batch.num_blocks == 0
    # every registration of the seven families launches batches; nothing else does
    families = ("decompression::dictionary::kernel_t", "decompression::frame_of_reference::kernel_t", "decompression::delta::kernel_t",
                "decompression::discard_zero_bytes::fixed_width::kernel_t", "decompression::discard_zero_bytes::variable_width::kernel_t",
                "decompression::run_length_encoding::kernel_t", "decompression::run_position_encoding::kernel_t")
    batching = [k for k in r.kernel_t.registered_keys() if r.kernel_t.produceSubclass(k).launches_batches()]
    assert len(batching) == 20 + 8 + 12 + 12 + 9 + 18 + 10 and all(k.startswith(families) for k in batching)
    scatter = r.kernel_t.produceSubclass(schemes.key_scatter(4, 8, 4))
    with pytest.raises(r.LogicError):
        scatter.batch(r.NOMINAL_MI355X)
    handle = ctypes.c_void_p()
    assert lib.giddy_batch_create(None, -1, ctypes.byref(handle)) == -1
    assert lib.giddy_batch_add(None, None, 0, None, 0) == -1 and lib.giddy_enqueue_batch(None, None) == -1 and lib.giddy_batch_upload(None, None) == -1
    assert lib.giddy_batch_num_pieces(None) == 0 and lib.giddy_can_launch_batches(None) == 0
    assert lib.giddy_enqueue_batch_of_device_tables(scatter._handle, None, None) == -1
    lib.giddy_batch_release(None)


def test_no_uniform_select_on_a_stale_scc_in_the_built_library():
    """Round 5 met a hipcc 7.2 miscompile: `high = new_high < new_low ? new_low : new_high` on wave-uniform 64-bit values became a VOPC compare followed by an s_cselect whose
    SCC had last been written by the s_addc of an address computation (run positions' wave-wide search returned after its first round).  tools/scan_isa.py looks for that
    signature in every kernel of the built library; it also lists scalar loads issued behind the wait for an earlier one (informative only)."""
    import subprocess
    import sys
    library = os.path.join(ROOT, "libgiddy_amd", "libgiddy_amd.so")
    if not os.path.exists(library):
        pytest.skip("the library is not built")
    result = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scan_isa.py"), library], capture_output=True, text=True, timeout=900)
    assert result.returncode == 0, result.stdout[-3000:] + result.stderr[-2000:]
    assert "a VOPC compare in between: 0 kernels" in result.stdout
    # (scan 3: the dictionary decoders' staging barrier — one textual site outside the tile / chunk loops since round 6)
    assert "more than three, or one inside a loop: 0 kernels" in result.stdout

"""One convenience function per decompression scheme: builds the factory key for the requested
instantiation, resolves the launch configuration and enqueues the kernel — i.e. exactly the three
calls of the reference's type-erased mode of use, through libgiddy_amd.registered (C ABI ->
C++ wrappers -> HIP kernels).  Nothing here computes anything on the host.

Buffers are device buffers (torch tensors, or raw device addresses); element types are stated by
the caller as byte sizes / dtype codes because they select the kernel instantiation.
Argument names and types: SURVEY.md §8b (reference file:line cited there and in INTEGRATION.md).
"""
import numpy as np

from . import registered as r

_SIGNED_NAMES = {1: "signed char", 2: "short", 4: "int", 8: "long"}
_kernels = {}


def kernel(key):
    """factory-produced wrapper, cached per key (wrappers are stateless)"""
    if key not in _kernels:
        _kernels[key] = r.kernel_t.produceSubclass(key)
    return _kernels[key]


class into_batch:
    """`with schemes.into_batch() as pieces:` — the scheme functions called inside ADD their column to one batch (libgiddy_amd.registered.batch_t,
    include/giddy/batch.hpp) instead of launching it: same arguments, nothing is enqueued (`stream`, `launch_config` are ignored; every
    call must name the same kernel instantiation).  Afterwards `pieces.enqueue(stream)` decodes them all with ONE launch — as often as wanted."""
    _current = None

    def __init__(self, device=r.CURRENT_DEVICE):
        self.device, self.key, self.batch = device, None, None

    def __enter__(self):
        assert into_batch._current is None, "batches do not nest"
        into_batch._current = self
        return self

    def __exit__(self, *exc):
        into_batch._current = None
        return False

    def _add(self, key, resolve_arguments, launch_arguments):
        if self.batch is None:
            self.key, self.batch = key, kernel(key).batch(self.device)
        if key != self.key:
            raise r.InvalidArgument("one batch, one kernel instantiation: %s after %s" % (key, self.key))
        self.batch.add(launch_arguments, resolve_arguments)

    @property
    def num_pieces(self):
        return self.batch.num_pieces if self.batch is not None else 0

    @property
    def num_blocks(self):
        return self.batch.num_blocks if self.batch is not None else 0

    def enqueue(self, stream):
        if self.batch is not None:
            self.batch.enqueue(stream)


def _run(key, resolve_arguments, launch_arguments, stream, device, limits, launch_config):
    if into_batch._current is not None:
        into_batch._current._add(key, resolve_arguments, launch_arguments)
        return None
    k = kernel(key)
    if launch_config is None:
        launch_config = k.resolve_launch_configuration(device, resolve_arguments, limits)
    k.enqueue_launch(stream, launch_config, launch_arguments)
    return launch_config


def _model_name(model, model_index_size, coefficient_size):
    return "cuda::functors::unary::parametric_model::%s<%du, %s>" % (model, model_index_size, _SIGNED_NAMES[coefficient_size])


# ---- factory keys (libstdc++-demangled type names, "cuda::kernels::" stripped) --------------------

def key_frame_of_reference(index_size, uncompressed_size, compressed_size, model="constant", model_index_size=4, coefficient_size=4):
    return "decompression::frame_of_reference::kernel_t<%du, %s, %s, %s >" % (
        index_size, _SIGNED_NAMES[uncompressed_size], _SIGNED_NAMES[compressed_size],
        _model_name(model, model_index_size, coefficient_size))


def key_model(index_size, uncompressed_size, model="linear", model_index_size=None, coefficient_size=None):
    return "decompression::model::kernel_t<%du, %s, %s >" % (
        index_size, _SIGNED_NAMES[uncompressed_size],
        _model_name(model, model_index_size or index_size, coefficient_size or uncompressed_size))


def key_dictionary(index_size, uncompressed_size, dictionary_index_size):
    return "decompression::dictionary::kernel_t<%du, %du, %du>" % (index_size, uncompressed_size, dictionary_index_size)


def key_delta(index_size, uncompressed_size, compressed_size):
    return "decompression::delta::kernel_t<%du, %du, %du>" % (index_size, uncompressed_size, compressed_size)


def key_patched(key):
    """the fused decompress-and-patch variant of a Delta / Dictionary / Frame-of-Reference / Model / DZB-fixed key (an extension)"""
    assert "::kernel_t<" in key and "::patched::" not in key
    return key.replace("::kernel_t<", "::patched::kernel_t<", 1)


def key_discard_zero_bytes_fixed(index_size, uncompressed_size, compressed_size):
    return ("decompression::discard_zero_bytes::fixed_width::kernel_t<%du, %du, %du, (util::endianness_t)1, (unsigned short)32>"
            % (index_size, uncompressed_size, compressed_size))


def key_discard_zero_bytes_variable(index_size, uncompressed_size, container_size=4):
    return "decompression::discard_zero_bytes::variable_width::kernel_t<%du, %du, %du>" % (index_size, uncompressed_size, container_size)


def key_incidence_bitmaps(index_size, uncompressed_size, alignment_in_ints=2, use_dictionary=True):
    return "decompression::incidence_bitmaps::kernel_t<%du, %du, %du, %s>" % (
        index_size, uncompressed_size, alignment_in_ints, "true" if use_dictionary else "false")


def key_run_length_encoding(index_size, uncompressed_size, run_length_size, run_index_size=None):
    return "decompression::run_length_encoding::kernel_t<%du, %du, %du, %du>" % (
        index_size, uncompressed_size, run_length_size, run_index_size or index_size)


def key_run_position_encoding(index_size, uncompressed_size, position_size, relative, run_index_size=None):
    return "decompression::run_position_encoding::kernel_t<%du, %du, %du, %s, %du>" % (
        index_size, uncompressed_size, position_size, "true" if relative else "false", run_index_size or index_size)


def key_scatter(output_index_size, element_size, input_index_size):
    return "scatter::kernel_t<%du, %du, %du, (unsigned short)16>" % (output_index_size, element_size, input_index_size)


def key_compressed_indices_scatter(output_index_size, element_size, input_index_size, run_length_size=None):
    return "scatter::compressed_indices::kernel_t<%du, %du, %du, %du>" % (
        output_index_size, element_size, input_index_size, run_length_size or input_index_size)


# ---- launches -------------------------------------------------------------------------------------

def frame_of_reference(decompressed, compressed_input, segment_model_coefficients, length, segment_length, *,
                       uncompressed_size=4, compressed_size=2, model="constant", index_size=4, model_index_size=4,
                       stream=0, device=r.CURRENT_DEVICE, limits=None, launch_config=None):
    dim = 1 if model == "constant" else 2
    return _run(
        key_frame_of_reference(index_size, uncompressed_size, compressed_size, model, model_index_size, 4),
        {"length": r.size_t(length), "segment_length": r.size_t(segment_length)},
        {"decompressed": r.pointer(decompressed, "i%d" % (8 * uncompressed_size)),
         "compressed_input": r.const_pointer(compressed_input, "i%d" % (8 * compressed_size)),
         "segment_model_coefficients": r.const_coefficients_pointer(segment_model_coefficients, "i32", dim),
         "length": r.uint_t(index_size)(length), "segment_length": r.uint_t(index_size)(segment_length)},
        stream, device, limits, launch_config)


def model(decompressed, model_coefficients, length, *, uncompressed_size=4, model="linear", index_size=4,
          stream=0, device=r.CURRENT_DEVICE, limits=None, launch_config=None):
    return _run(
        key_model(index_size, uncompressed_size, model),
        {"length": r.size_t(length)},
        {"decompressed": r.pointer(decompressed, "i%d" % (8 * uncompressed_size)),
         "model_coefficients": r.coefficients(model_coefficients, "i%d" % (8 * uncompressed_size)),
         "length": r.uint_t(index_size)(length)},
        stream, device, limits, launch_config)


def dictionary(decompressed, compressed_input, dictionary_entries, length, num_dictionary_entries, *,
               uncompressed_size=4, dictionary_index_size=1, index_size=4, cache_dictionary_in_shared_mem=None,
               stream=0, device=r.CURRENT_DEVICE, limits=None, launch_config=None):
    resolve = {"num_dictionary_entries": r.size_t(num_dictionary_entries), "data_length": r.size_t(length)}
    launch = {"decompressed": r.pointer(decompressed, "u%d" % (8 * uncompressed_size)),
              "compressed_input": r.const_pointer(compressed_input, "u%d" % (8 * dictionary_index_size)),
              "dictionary_entries": r.const_pointer(dictionary_entries, "u%d" % (8 * uncompressed_size)),
              "length": r.size_type_by_index_size(index_size)(length),
              "num_dictionary_entries": r.size_type_by_index_size(dictionary_index_size)(num_dictionary_entries)}
    if cache_dictionary_in_shared_mem is not None:
        # the reference spells the flag differently at resolve and launch time (dictionary.cu:59 vs :78)
        resolve["cache_input_data_in_shared_mem"] = r.boolean(cache_dictionary_in_shared_mem)
        launch["cache_dictionary_in_shared_mem"] = r.boolean(cache_dictionary_in_shared_mem)
    return _run(key_dictionary(index_size, uncompressed_size, dictionary_index_size), resolve, launch,
                stream, device, limits, launch_config)


def delta(decompressed, compressed_input, anchor_values, length, segment_length, *,
          uncompressed_size=8, compressed_size=2, index_size=4, part_anchors_scratch=None,
          stream=0, device=r.CURRENT_DEVICE, limits=None, launch_config=None):
    """`part_anchors_scratch` (an extension; a device buffer of ceil(length / 4096) + 1024 uncompressed-type entries): segments of 2^17 elements
    or more are decoded in parts — a column that is ONE segment uses the whole GPU instead of one workgroup"""
    launch = {"decompressed": r.pointer(decompressed, "u%d" % (8 * uncompressed_size)),
              "compressed_input": r.const_pointer(compressed_input, "u%d" % (8 * compressed_size)),
              "anchor_values": r.const_pointer(anchor_values, "u%d" % (8 * uncompressed_size)),
              "length": r.uint_t(index_size)(length), "segment_length": r.uint_t(index_size)(segment_length)}
    if part_anchors_scratch is not None:
        launch["part_anchors_scratch"] = r.pointer(part_anchors_scratch, "u%d" % (8 * uncompressed_size))
    return _run(
        key_delta(index_size, uncompressed_size, compressed_size),
        {"length": r.size_t(length), "segment_length": r.size_t(segment_length)}, launch,
        stream, device, limits, launch_config)


# ---- fused decompress-and-patch launches (extension; patch positions ascending) ---------------------

_COMPRESSED_POSITION_ARRAYS = (   # name, element width (None: the column's index size)
    ("scatter_position_run_lengths", None), ("scatter_position_run_individual_offset_sizes", 1),
    ("scatter_position_run_baseline_values", None), ("scatter_position_run_offsets_start_pos", None),
    ("scatter_position_offset_bytes", 1), ("scatter_position_anchors", None))


def _compressed_position_arguments(compressed, output_index_size, input_index_size, run_length_size):
    widths = {"scatter_position_run_lengths": run_length_size, "scatter_position_run_baseline_values": output_index_size}
    arguments = {name: r.const_pointer(compressed[name], "u%d" % (8 * (width or widths.get(name, input_index_size))))
                 for name, width in _COMPRESSED_POSITION_ARRAYS}
    arguments["anchoring_period"] = r.uint_t(input_index_size)(compressed["anchoring_period"])
    arguments["num_scatter_position_runs"] = r.uint_t(input_index_size)(compressed["num_scatter_position_runs"])
    return arguments


def _patch_arguments(patch_values, patch_positions, num_patches, value_code, index_size):
    """`patch_positions`: a plain ascending array, or a dict of the compressed-indices arrays (formats.compressed_indices,
    built with output_index_size = input_index_size = the column's index size) plus "patch_positions_scratch", a
    buffer of num_patches indices the positions are expanded into on the way (patched/patch_positions.hpp)"""
    arguments = {"patch_values": r.const_pointer(patch_values, value_code), "num_patches": r.uint_t(index_size)(num_patches)}
    if isinstance(patch_positions, dict):
        arguments.update(_compressed_position_arguments(patch_positions, index_size, index_size, index_size))
        arguments["patch_positions_scratch"] = r.pointer(patch_positions["patch_positions_scratch"], "u%d" % (8 * index_size))
        if "unsupported_patch_count" in patch_positions:   # optional: a zeroed device counter of patches no plain position can stand for
            arguments["unsupported_patch_count"] = r.pointer(patch_positions["unsupported_patch_count"], "u32")
    else:
        arguments["patch_positions"] = r.const_pointer(patch_positions, "u%d" % (8 * index_size))
    return arguments


def patched_delta(decompressed, compressed_input, anchor_values, patch_values, patch_positions, length, segment_length, num_patches, *,
                  uncompressed_size=8, compressed_size=2, index_size=4, stream=0, device=r.CURRENT_DEVICE, limits=None, launch_config=None):
    """delta() followed by scatter(decompressed, patch_values, patch_positions, num_patches), in one kernel"""
    launch = {"decompressed": r.pointer(decompressed, "u%d" % (8 * uncompressed_size)),
              "compressed_input": r.const_pointer(compressed_input, "u%d" % (8 * compressed_size)),
              "anchor_values": r.const_pointer(anchor_values, "u%d" % (8 * uncompressed_size)),
              "length": r.uint_t(index_size)(length), "segment_length": r.uint_t(index_size)(segment_length)}
    launch.update(_patch_arguments(patch_values, patch_positions, num_patches, "u%d" % (8 * uncompressed_size), index_size))
    return _run(key_patched(key_delta(index_size, uncompressed_size, compressed_size)),
                {"length": r.size_t(length), "segment_length": r.size_t(segment_length), "num_patches": r.size_t(num_patches)}, launch,
                stream, device, limits, launch_config)


def patched_model(decompressed, model_coefficients, patch_values, patch_positions, length, num_patches, *,
                  uncompressed_size=4, model="linear", index_size=4, stream=0, device=r.CURRENT_DEVICE, limits=None, launch_config=None):
    """model() followed by scatter(decompressed, patch_values, patch_positions, num_patches), in one kernel"""
    launch = {"decompressed": r.pointer(decompressed, "i%d" % (8 * uncompressed_size)),
              "model_coefficients": r.coefficients(model_coefficients, "i%d" % (8 * uncompressed_size)),
              "length": r.uint_t(index_size)(length)}
    launch.update(_patch_arguments(patch_values, patch_positions, num_patches, "i%d" % (8 * uncompressed_size), index_size))
    resolve = {"length": r.size_t(length), "num_patches": r.size_t(num_patches)}   # (the count is optional when resolving: it picks the residency)
    return _run(key_patched(key_model(index_size, uncompressed_size, model)), resolve, launch,
                stream, device, limits, launch_config)


def patched_discard_zero_bytes_fixed(decompressed, compressed_input, patch_values, patch_positions, length, num_patches, *,
                                     uncompressed_size=8, compressed_size=2, index_size=4, stream=0, device=r.CURRENT_DEVICE, limits=None, launch_config=None):
    """discard_zero_bytes_fixed() followed by scatter(decompressed, patch_values, patch_positions, num_patches), in one kernel"""
    launch = {"decompressed": r.pointer(decompressed, "u%d" % (8 * uncompressed_size)),
              "compressed_input": r.const_pointer(compressed_input, "u%d" % (8 * compressed_size)),
              "length": r.uint_t(index_size)(length)}
    launch.update(_patch_arguments(patch_values, patch_positions, num_patches, "u%d" % (8 * uncompressed_size), index_size))
    return _run(key_patched(key_discard_zero_bytes_fixed(index_size, uncompressed_size, compressed_size)), {"length": r.size_t(length), "num_patches": r.size_t(num_patches)}, launch,
                stream, device, limits, launch_config)


def patched_frame_of_reference(decompressed, compressed_input, segment_model_coefficients, patch_values, patch_positions,
                               length, segment_length, num_patches, *,
                               uncompressed_size=4, compressed_size=2, model="constant", index_size=4, model_index_size=4,
                               stream=0, device=r.CURRENT_DEVICE, limits=None, launch_config=None):
    """frame_of_reference() followed by scatter(), in one kernel"""
    dim = 1 if model == "constant" else 2
    launch = {"decompressed": r.pointer(decompressed, "i%d" % (8 * uncompressed_size)),
              "compressed_input": r.const_pointer(compressed_input, "i%d" % (8 * compressed_size)),
              "segment_model_coefficients": r.const_coefficients_pointer(segment_model_coefficients, "i32", dim),
              "length": r.uint_t(index_size)(length), "segment_length": r.uint_t(index_size)(segment_length)}
    launch.update(_patch_arguments(patch_values, patch_positions, num_patches, "i%d" % (8 * uncompressed_size), index_size))
    return _run(key_patched(key_frame_of_reference(index_size, uncompressed_size, compressed_size, model, model_index_size, 4)),
                {"length": r.size_t(length), "segment_length": r.size_t(segment_length)}, launch,
                stream, device, limits, launch_config)


def patched_dictionary(decompressed, compressed_input, dictionary_entries, patch_values, patch_positions,
                       length, num_dictionary_entries, num_patches, *,
                       uncompressed_size=4, dictionary_index_size=1, index_size=4, cache_dictionary_in_shared_mem=None,
                       stream=0, device=r.CURRENT_DEVICE, limits=None, launch_config=None):
    """dictionary() followed by scatter(), in one kernel"""
    resolve = {"num_dictionary_entries": r.size_t(num_dictionary_entries), "data_length": r.size_t(length), "num_patches": r.size_t(num_patches)}
    launch = {"decompressed": r.pointer(decompressed, "u%d" % (8 * uncompressed_size)),
              "compressed_input": r.const_pointer(compressed_input, "u%d" % (8 * dictionary_index_size)),
              "dictionary_entries": r.const_pointer(dictionary_entries, "u%d" % (8 * uncompressed_size)),
              "length": r.size_type_by_index_size(index_size)(length),
              "num_dictionary_entries": r.size_type_by_index_size(dictionary_index_size)(num_dictionary_entries)}
    launch.update(_patch_arguments(patch_values, patch_positions, num_patches, "u%d" % (8 * uncompressed_size), index_size))
    if cache_dictionary_in_shared_mem is not None:
        resolve["cache_input_data_in_shared_mem"] = r.boolean(cache_dictionary_in_shared_mem)
        launch["cache_dictionary_in_shared_mem"] = r.boolean(cache_dictionary_in_shared_mem)
    return _run(key_patched(key_dictionary(index_size, uncompressed_size, dictionary_index_size)), resolve, launch,
                stream, device, limits, launch_config)


def discard_zero_bytes_fixed(decompressed, compressed_input, length, *, uncompressed_size=8, compressed_size=4,
                             index_size=4, stream=0, device=r.CURRENT_DEVICE, limits=None, launch_config=None):
    return _run(
        key_discard_zero_bytes_fixed(index_size, uncompressed_size, compressed_size),
        {"length": r.size_t(length)},
        {"decompressed": r.pointer(decompressed, "u%d" % (8 * uncompressed_size)),
         "compressed_input": r.const_pointer(compressed_input, "u%d" % (8 * compressed_size)),
         "length": r.uint_t(index_size)(length)},
        stream, device, limits, launch_config)


def discard_zero_bytes_variable(decompressed, compressed_data, packed_element_sizes, position_anchors,
                                segment_length, length, min_represented_element_size, bits_per_element_size, *,
                                uncompressed_size=8, container_size=4, index_size=8,
                                stream=0, device=r.CURRENT_DEVICE, limits=None, launch_config=None):
    return _run(
        key_discard_zero_bytes_variable(index_size, uncompressed_size, container_size),
        {"length": r.size_t(length), "segment_length": r.size_t(segment_length)},
        {"decompressed": r.pointer(decompressed, "u%d" % (8 * uncompressed_size)),
         "compressed_data": r.const_pointer(compressed_data, "u8"),
         "packed_element_sizes": r.const_pointer(packed_element_sizes, "u%d" % (8 * container_size)),
         "position_anchors": r.const_pointer(position_anchors, "u%d" % (8 * index_size)),
         "segment_length": r.uint_t(index_size)(segment_length), "length": r.uint_t(index_size)(length),
         "min_represented_element_size": r.uint32(min_represented_element_size),
         "bits_per_element_size": r.uint32(bits_per_element_size)},
        stream, device, limits, launch_config)


def incidence_bitmaps(decompressed, bitmaps, dictionary_entries, bitmap_length, num_bitmaps, *,
                      uncompressed_size=4, alignment_in_ints=2, use_dictionary=True, index_size=4,
                      stream=0, device=r.CURRENT_DEVICE, limits=None, launch_config=None):
    return _run(
        key_incidence_bitmaps(index_size, uncompressed_size, alignment_in_ints, use_dictionary),
        {"length": r.size_t(32 * bitmap_length)},
        {"decompressed": r.pointer(decompressed, "u%d" % (8 * uncompressed_size)),
         "incidence_bitmaps": r.const_pointer(bitmaps, "u32"),
         "dictionary_entries": r.const_pointer(dictionary_entries, "u%d" % (8 * uncompressed_size)),
         "bitmap_length": r.size_type_by_index_size(index_size)(bitmap_length),
         "num_bitmaps": r.uint32(num_bitmaps)},
        stream, device, limits, launch_config)


def run_length_encoding(decompressed, run_data, run_lengths, position_anchors, intra_run_anchor_offsets,
                        segment_length, num_segments, num_element_runs, length, *,
                        uncompressed_size=4, run_length_size=2, index_size=4, run_index_size=None,
                        stream=0, device=r.CURRENT_DEVICE, limits=None, launch_config=None):
    run_index_size = run_index_size or index_size
    return _run(
        key_run_length_encoding(index_size, uncompressed_size, run_length_size, run_index_size),
        {"length": r.size_t(length), "segment_length": r.size_t(segment_length)},
        {"decompressed": r.pointer(decompressed, "u%d" % (8 * uncompressed_size)),
         "run_data": r.const_pointer(run_data, "u%d" % (8 * uncompressed_size)),
         "run_lengths": r.const_pointer(run_lengths, "u%d" % (8 * run_length_size)),
         "position_anchors": r.const_pointer(position_anchors, "u%d" % (8 * run_index_size)),
         "intra_run_anchor_offsets": r.const_pointer(intra_run_anchor_offsets, "u%d" % (8 * run_length_size)),
         "segment_length": r.size_type_by_index_size(index_size)(segment_length),
         "num_segments": r.size_type_by_index_size(index_size)(num_segments),
         "num_element_runs": r.size_type_by_index_size(run_index_size)(num_element_runs),
         "length": r.size_type_by_index_size(index_size)(length)},
        stream, device, limits, launch_config)


def run_position_encoding(decompressed, run_data, run_start_positions, position_anchors, segment_length,
                          num_anchors, num_element_runs, length, *, uncompressed_size=4, position_size=2,
                          relative=True, index_size=4, run_index_size=None,
                          stream=0, device=r.CURRENT_DEVICE, limits=None, launch_config=None):
    run_index_size = run_index_size or index_size
    return _run(
        key_run_position_encoding(index_size, uncompressed_size, position_size, relative, run_index_size),
        {"length": r.size_t(length), "segment_length": r.size_t(segment_length)},
        {"decompressed": r.pointer(decompressed, "u%d" % (8 * uncompressed_size)),
         "run_data": r.const_pointer(run_data, "u%d" % (8 * uncompressed_size)),
         "run_start_positions": r.const_pointer(run_start_positions, "u%d" % (8 * position_size)),
         "position_anchors": r.const_pointer(position_anchors, "u%d" % (8 * run_index_size)),
         "segment_length": r.size_type_by_index_size(position_size)(segment_length),
         "num_anchors": r.size_type_by_index_size(index_size)(num_anchors),
         "num_element_runs": r.size_type_by_index_size(run_index_size)(num_element_runs),
         "length": r.size_type_by_index_size(index_size)(length)},
        stream, device, limits, launch_config)


def scatter(target, data, indices, data_length, *, output_index_size=4, element_size=8, input_index_size=4,
            stream=0, device=r.CURRENT_DEVICE, limits=None, launch_config=None):
    return _run(
        key_scatter(output_index_size, element_size, input_index_size),
        {"data_length": r.size_t(data_length)},
        {"target": r.pointer(target, "u%d" % (8 * element_size)),
         "data": r.const_pointer(data, "u%d" % (8 * element_size)),
         "indices": r.const_pointer(indices, "u%d" % (8 * output_index_size)),
         "data_length": r.size_type_by_index_size(input_index_size)(data_length)},
        stream, device, limits, launch_config)


def compressed_indices_scatter(target, data_to_scatter, scatter_position_run_lengths,
                               scatter_position_run_individual_offset_sizes, scatter_position_run_baseline_values,
                               scatter_position_run_offsets_start_pos, scatter_position_offset_bytes,
                               scatter_position_anchors, anchoring_period, num_scatter_position_runs,
                               input_data_length, *, output_index_size=4, element_size=4, input_index_size=4,
                               run_length_size=None, stream=0, device=r.CURRENT_DEVICE, limits=None, launch_config=None):
    run_length_size = run_length_size or input_index_size
    return _run(
        key_compressed_indices_scatter(output_index_size, element_size, input_index_size, run_length_size),
        {"input_data_length": r.size_t(input_data_length), "anchoring_period": r.size_t(anchoring_period)},
        {"target": r.pointer(target, "u%d" % (8 * element_size)),
         "data_to_scatter": r.const_pointer(data_to_scatter, "u%d" % (8 * element_size)),
         "scatter_position_run_lengths": r.const_pointer(scatter_position_run_lengths, "u%d" % (8 * run_length_size)),
         "scatter_position_run_individual_offset_sizes": r.const_pointer(scatter_position_run_individual_offset_sizes, "u8"),
         "scatter_position_run_baseline_values": r.const_pointer(scatter_position_run_baseline_values, "u%d" % (8 * output_index_size)),
         "scatter_position_run_offsets_start_pos": r.const_pointer(scatter_position_run_offsets_start_pos, "u%d" % (8 * input_index_size)),
         "scatter_position_offset_bytes": r.const_pointer(scatter_position_offset_bytes, "u8"),
         "scatter_position_anchors": r.const_pointer(scatter_position_anchors, "u%d" % (8 * input_index_size)),
         "anchoring_period": r.uint_t(input_index_size)(anchoring_period),
         "num_scatter_position_runs": r.uint_t(input_index_size)(num_scatter_position_runs),
         "input_data_length": r.size_type_by_index_size(input_index_size)(input_data_length)},
        stream, device, limits, launch_config)


def key_compressed_indices_positions(output_index_size, input_index_size, run_length_size=None):
    return "scatter::compressed_indices::positions::kernel_t<%du, %du, %du>" % (
        output_index_size, input_index_size, run_length_size or input_index_size)


def compressed_indices_positions(positions, compressed, input_data_length, *, output_index_size=4, input_index_size=4,
                                 run_length_size=None, stream=0, device=r.CURRENT_DEVICE, limits=None, launch_config=None):
    """positions[k] = where patch k goes: the position half of compressed_indices_scatter(), as a plain ascending array
    (`compressed`: the arrays of formats.compressed_indices, on the device)"""
    run_length_size = run_length_size or input_index_size
    launch = {"positions": r.pointer(positions, "u%d" % (8 * output_index_size)),
              "input_data_length": r.size_type_by_index_size(input_index_size)(input_data_length)}
    launch.update(_compressed_position_arguments(compressed, output_index_size, input_index_size, run_length_size))
    if "unsupported_patch_count" in compressed:
        launch["unsupported_patch_count"] = r.pointer(compressed["unsupported_patch_count"], "u32")
    return _run(key_compressed_indices_positions(output_index_size, input_index_size, run_length_size),
                {"input_data_length": r.size_t(input_data_length), "anchoring_period": r.size_t(compressed["anchoring_period"])},
                launch, stream, device, limits, launch_config)


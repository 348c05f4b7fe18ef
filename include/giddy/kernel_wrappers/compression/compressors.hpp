// Kernel wrappers for the device-side COMPRESSORS (an extension — the reference has decompressors only): registered in the
// same factory under keys "compression::<scheme>::kernel_t<...>", driven through the same three calls (and the same C ABI) as
// the decompressors.  enqueue_launch() enqueues every phase of the compressor on the caller's stream; the launch configuration
// that resolve_launch_configuration() returns is that of the first phase, and is otherwise not used (each phase sizes its own
// grid from the column's length).  Common argument-map keys:
//   "input"                    const uint_t<UncompressedSize>*   the column (device)
//   "length"                   uint_t<IndexSize>
//   "tile_totals"              uint64_t*   scratch, scratch_entries(length) entries (variable-rate schemes)
//   "unrepresentable_count"    unsigned*   optional, caller-zeroed: elements the requested widths cannot hold
// Outputs are named as the decompressors' inputs are (libgiddy_amd/formats.py).  Sizes the caller cannot know in advance
// (bytes, runs, patches) are left in a device scalar it names ("total_bytes" / "num_element_runs" / "num_patches": uint64_t*).
#pragma once
#include <cstdlib>
#include "giddy/kernel_wrappers/common.hpp"
#include "giddy/kernels/compression/discard_zero_bytes_variable.hpp"
#include "giddy/kernels/compression/compressed_indices.hpp"
#include "giddy/kernels/compression/dictionary_construction.hpp"
#include "giddy/kernels/compression/elementwise.hpp"
#include "giddy/kernels/compression/incidence_bitmaps.hpp"
#include "giddy/kernels/compression/patches.hpp"
#include "giddy/kernels/compression/runs.hpp"
#include "giddy/kernels/compression/listed_runs.hpp"

namespace cuda {
namespace kernels {
namespace compression {

/// uint64 entries of "tile_totals" scratch enough for the run, patch and elementwise compressors of a column of `length` elements,
/// and for the variable-width discard-zero-bytes compressor with at least 8 size fields per container
inline size_t scratch_entries(size_t length) { return 2 * (util::div_rounding_up(length, (size_t) tile_elements) + 2) + 2; }
/// ... and for variable-width discard-zero-bytes in general: its tiles are threads_per_block containers, of `fields_per_container`
/// elements each (8 * ElementSizesContainerSize / bits_per_element_size: fewer than 8 from 5-bit fields in 4-byte containers on)
inline size_t discard_zero_bytes_variable_scratch_entries(size_t length, size_t fields_per_container)
{
	return std::max(scratch_entries(length), util::div_rounding_up(length, (size_t) threads_per_block * std::max<size_t>(fields_per_container, 1)) + 3);
}

/// "run_scratch" (uint64_t*, with "run_scratch_entries", size_t — both optional) of the run encodings' compressors: with it they take five or six small launches, one of which reads the column, and no
/// workgroup waits for another (kernels/compression/listed_runs.hpp; run lengths in 1-byte fields: always the chained kernel); `staged_per_wave`
/// (a multiple of 16, at most 512): the runs per 4096 elements the scratch holds — a wavefront of more is read a second time.
inline size_t run_scratch_entries(size_t length, unsigned element_size, unsigned staged_per_wave = runs::listed::staged_per_wave_at_most)
{
	return (size_t) runs::listed::scratch_layout(util::div_rounding_up((uint64_t) length, (uint64_t) runs::one_pass::wave_elements), element_size, staged_per_wave).entries;
}

namespace detail {

/// what the launches share: the scratch cut up; the first launch (the one that reads the column) and the last of phase 2 made by members
template <unsigned U>
struct listed_runs {
	uint64_t num_waves, num_blocks;
	unsigned staged_per_wave;
	uint64_t *headers, *bases, *previous_starts, *next_starts;
	uint64_t *block_sums, *block_bases, *block_first_starts, *block_last_starts, *block_next_starts, *block_previous_starts;
	uint16_t* offsets;
	uint_t<U>* values;
	launch_configuration_t per_wavefronts, relisting, per_blocks;

	listed_runs(stream::id_t stream, uint64_t* scratch, size_t scratch_entries, const uint_t<U>* input, uint64_t length)
	: num_waves(util::div_rounding_up(length, (uint64_t) runs::one_pass::wave_elements)),
	  num_blocks(util::div_rounding_up(num_waves, (uint64_t) runs::listed::block_waves)),
	  staged_per_wave(runs::listed::staged_per_wave_that_fits(scratch_entries, num_waves, U)),
	  per_wavefronts(tile_launch(util::div_rounding_up(num_waves, (uint64_t) waves_per_block))),
	  relisting(tile_launch(std::min<uint64_t>(util::div_rounding_up(num_waves, (uint64_t) waves_per_block), runs::listed::relisting_workgroups))),
	  per_blocks(tile_launch(num_blocks))
	{
		if (staged_per_wave == 0) { throw std::invalid_argument("\"run_scratch\" is too small for this column: see run_scratch_entries()"); }
		if (reinterpret_cast<uintptr_t>(scratch) % 16 != 0) { throw std::invalid_argument("\"run_scratch\" must be 16-byte aligned"); }
		if (num_blocks > 0x7FFFFFFFull) { throw std::invalid_argument("the column has more blocks of wavefronts than a grid has workgroups"); }
		const runs::listed::scratch_layout layout(num_waves, U, staged_per_wave);
		headers = scratch + layout.headers; bases = scratch + layout.bases; previous_starts = scratch + layout.previous_starts; next_starts = scratch + layout.next_starts;
		uint64_t* const blocks = scratch + layout.blocks;
		block_sums = blocks; block_bases = blocks + layout.blocks_stride; block_first_starts = blocks + 2 * layout.blocks_stride; block_last_starts = blocks + 3 * layout.blocks_stride;
		block_next_starts = blocks + 4 * layout.blocks_stride; block_previous_starts = blocks + 5 * layout.blocks_stride;
		offsets = reinterpret_cast<uint16_t*>(scratch + layout.offsets); values = reinterpret_cast<uint_t<U>*>(scratch + layout.values);
		cuda::enqueue_launch(reinterpret_cast<const void*>(runs::listed::list_runs<U>), stream, per_wavefronts, headers, offsets, values, input, length, (uint32_t) staged_per_wave);
	}

	void place_blocks(stream::id_t stream, uint64_t* grand_total) const
	{
		cuda::enqueue_launch(reinterpret_cast<const void*>(runs::listed::place_blocks<>), stream, launch_configuration_t(1, threads_per_block),
			block_bases, (const uint64_t*) block_sums, num_blocks, grand_total);
	}
};

/// (a caller that names the chain's test arguments wants the chain)
inline bool chain_is_asked_for(const cuda::registered::kernel_t::arguments_type& arguments)
{
	return optional_argument<uint32_t>(arguments, "chain_patience") || optional_argument<uint64_t>(arguments, "chain_fault_tile");
}

} // namespace detail

/// The chained kernels' grid: EXACTLY a workgroup per tile.  A smaller grid walking the tiles at grid stride deadlocks as soon as it is larger
/// than what the device holds at once: a resident workgroup's second tile waits for the first tile of a workgroup that cannot start
/// before a resident one ends (seen with 2048 workgroups for 16 384 tiles; 1024 — all resident — ran, no faster than one per tile).
inline launch_configuration_t chained_launch(uint64_t num_tiles)
{
	if (num_tiles > 0x7FFFFFFFull) { throw std::invalid_argument("the column has more tiles than a grid has workgroups"); }
	return launch_configuration_t((grid_dimension_t) (num_tiles < 1 ? 1 : num_tiles), threads_per_block);
}

inline launch_configuration_t elementwise_launch(uint64_t length)
{
	return tile_launch(util::div_rounding_up(length, (uint64_t) threads_per_block * 4));
}

// ---- Delta -------------------------------------------------------------------------------------------------------------
namespace delta {

template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);
};

template <unsigned I, unsigned U, unsigned C>
launch_configuration_t kernel_t<I, U, C>::resolve_launch_configuration(device::properties_t, device_function::attributes_t, arguments_type extra_arguments, launch_configuration_limits_t) const
{
	if (argument<size_t>(extra_arguments, "segment_length") == 0) { throw std::invalid_argument("The segment length must be a positive number"); }
	return elementwise_launch(argument<size_t>(extra_arguments, "length"));
}

template <unsigned I, unsigned U, unsigned C>
void kernel_t<I, U, C>::enqueue_launch_with(stream::id_t stream, const launch_configuration_t& launch_config, const arguments_type& arguments) const
{
	using size_type = size_type_by_index_size<I>;
	cuda::kernel::enqueue_launch(*this, stream, launch_config,
		argument<uint_t<C>*>(arguments, "compressed_input"), argument<uint_t<U>*>(arguments, "anchor_values"),
		argument<const uint_t<U>*>(arguments, "input"), (size_type) argument<uint_t<I>>(arguments, "length"),
		(size_type) argument<uint_t<I>>(arguments, "segment_length"),
		optional_argument<native_word_t*>(arguments, "unrepresentable_count").value_or(nullptr));
}

template <unsigned I, unsigned U, unsigned C>
const device_function_t kernel_t<I, U, C>::get_device_function() const { return reinterpret_cast<const void*>(compress<I, U, C>); }

} // namespace delta

// ---- fixed-width discard-zero-bytes --------------------------------------------------------------------------------------
namespace discard_zero_bytes {
namespace fixed_width {

template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);
};

template <unsigned I, unsigned U, unsigned C>
launch_configuration_t kernel_t<I, U, C>::resolve_launch_configuration(device::properties_t, device_function::attributes_t, arguments_type extra_arguments, launch_configuration_limits_t) const
{
	return elementwise_launch(argument<size_t>(extra_arguments, "length"));
}

template <unsigned I, unsigned U, unsigned C>
void kernel_t<I, U, C>::enqueue_launch_with(stream::id_t stream, const launch_configuration_t& launch_config, const arguments_type& arguments) const
{
	cuda::kernel::enqueue_launch(*this, stream, launch_config,
		argument<uint_t<C>*>(arguments, "compressed_input"), argument<const uint_t<U>*>(arguments, "input"),
		(size_type_by_index_size<I>) argument<uint_t<I>>(arguments, "length"),
		optional_argument<native_word_t*>(arguments, "unrepresentable_count").value_or(nullptr));
}

template <unsigned I, unsigned U, unsigned C>
const device_function_t kernel_t<I, U, C>::get_device_function() const { return reinterpret_cast<const void*>(compress<I, U, C>); }

} // namespace fixed_width

// ---- variable-width discard-zero-bytes -------------------------------------------------------------------------------------
namespace variable_width {

/// also: "compressed_data" uint8_t* (room for length * UncompressedSize bytes at worst), "packed_element_sizes", "position_anchors",
/// "segment_length", "min_represented_element_size" (unsigned), "bits_per_element_size" (unsigned), "total_bytes" uint64_t* (optional);
/// scratch "tile_totals": discard_zero_bytes_variable_scratch_entries(length, fields per container) entries
template <unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);
};

template <unsigned I, unsigned U, unsigned CS>
launch_configuration_t kernel_t<I, U, CS>::resolve_launch_configuration(device::properties_t, device_function::attributes_t, arguments_type extra_arguments, launch_configuration_limits_t) const
{
	if (argument<size_t>(extra_arguments, "segment_length") == 0) { throw std::invalid_argument("The segment length must be a positive number"); }
	return tile_launch(util::div_rounding_up(argument<size_t>(extra_arguments, "length"), (size_t) tile_elements));
}

template <unsigned I, unsigned U, unsigned CS>
void kernel_t<I, U, CS>::enqueue_launch_with(stream::id_t stream, const launch_configuration_t&, const arguments_type& arguments) const
{
	using size_type = size_type_by_index_size<I>;
	auto input      = argument<const uint_t<U>*>(arguments, "input");
	auto length     = (size_type) argument<uint_t<I>>(arguments, "length");
	auto segment    = (size_type) argument<uint_t<I>>(arguments, "segment_length");
	auto min_size   = argument<native_word_t>(arguments, "min_represented_element_size");
	auto bits       = argument<native_word_t>(arguments, "bits_per_element_size");
	auto totals     = argument<uint64_t*>(arguments, "tile_totals");
	if (bits == 0 || bits > 8 * CS) { throw std::invalid_argument("The size fields must be between 1 bit and a container wide"); }
	if (min_size > U) { throw std::invalid_argument("The minimum represented size exceeds the element size"); }
	if (length == 0) { return; }
	const uint64_t per = (8 * CS) / bits, num_tiles = util::div_rounding_up((uint64_t) length, (uint64_t) threads_per_block * per);
	const auto config = tile_launch(num_tiles);
	auto compressed_data = argument<uint8_t*>(arguments, "compressed_data");
	auto packed_sizes = argument<uint_t<CS>*>(arguments, "packed_element_sizes");
	auto anchors = argument<uint_t<I>*>(arguments, "position_anchors");
	auto unrepresentable = optional_argument<native_word_t*>(arguments, "unrepresentable_count").value_or(nullptr);
	auto total_bytes = optional_argument<uint64_t*>(arguments, "total_bytes").value_or(nullptr);
	const uint32_t patience = optional_argument<uint32_t>(arguments, "chain_patience").value_or((uint32_t) chained::default_patience);
	// The write pass by ROWS of 64 elements per wavefront where the shape allows — fields per container a power of two, a wavefront's bytes within
	// its staging area, segments of 64 elements or more (kernels/compression/discard_zero_bytes_variable.hpp) — and then in ONE launch, the tiles
	// chained by a decoupled look-back (the column read once; 2^28 64-bit elements: 2.66 ms in three launches with a thread per container
	// -> 1.4 ms by rows -> profiles/r04_compress_bench.txt).  Otherwise: count, offsets, write by a thread per container.
	const unsigned largest = largest_size_of(min_size, bits, U);
	const bool by_rows = compress_by_rows_applies((unsigned) per, largest, (uint64_t) segment);
	const shared_memory_size_t staging_bytes = (shared_memory_size_t) (waves_per_block * compress_by_rows_staging_words((unsigned) per, largest) * sizeof(uint32_t));
	if (by_rows && num_tiles <= 0x7FFFFFFFull) {
		const uint64_t num_links = util::div_rounding_up(num_tiles, (uint64_t) 4);   // (compress_by_rows: a workgroup — a link of the chain — takes four consecutive tiles)
		enqueue_zero(stream, totals, num_links * sizeof(uint64_t));
		auto chained_config = chained_launch(num_links);
		chained_config.dynamic_shared_memory_size = staging_bytes;
		cuda::enqueue_launch(reinterpret_cast<const void*>(compress_by_rows<I, U, CS, true>), stream, chained_config,
			compressed_data, packed_sizes, anchors, input, totals, segment, length, min_size, bits, unrepresentable, total_bytes, patience);
		return;
	}
	cuda::enqueue_launch(reinterpret_cast<const void*>(count_bytes<I, U, CS>), stream, config, totals, input, length, min_size, bits);
	enqueue_exclusive_offsets(stream, totals, num_tiles, total_bytes);
	if (by_rows) {
		auto write_config = config;
		write_config.dynamic_shared_memory_size = staging_bytes;
		cuda::enqueue_launch(reinterpret_cast<const void*>(compress_by_rows<I, U, CS, false>), stream, write_config,
			compressed_data, packed_sizes, anchors, input, totals, segment, length, min_size, bits, unrepresentable, static_cast<uint64_t*>(nullptr), patience);
		return;
	}
	cuda::enqueue_launch(reinterpret_cast<const void*>(compress<I, U, CS>), stream, config,
		compressed_data, packed_sizes, anchors, input, static_cast<const uint64_t*>(totals), segment, length, min_size, bits, unrepresentable);
}

template <unsigned I, unsigned U, unsigned CS>
const device_function_t kernel_t<I, U, CS>::get_device_function() const { return reinterpret_cast<const void*>(compress<I, U, CS>); }

} // namespace variable_width
} // namespace discard_zero_bytes

// ---- Frame of Reference, a constant model per segment ----------------------------------------------------------------------
namespace frame_of_reference {

template <unsigned IndexSize, typename Uncompressed, typename Compressed>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);
};

template <unsigned I, typename U, typename C>
launch_configuration_t kernel_t<I, U, C>::resolve_launch_configuration(device::properties_t, device_function::attributes_t, arguments_type extra_arguments, launch_configuration_limits_t) const
{
	auto segment_length = argument<size_t>(extra_arguments, "segment_length");
	if (segment_length == 0) { throw std::invalid_argument("The segment length must be a positive number"); }
	return tile_launch(util::div_rounding_up(argument<size_t>(extra_arguments, "length"), segment_length));
}

template <unsigned I, typename U, typename C>
void kernel_t<I, U, C>::enqueue_launch_with(stream::id_t stream, const launch_configuration_t& launch_config, const arguments_type& arguments) const
{
	using size_type = size_type_by_index_size<I>;
	cuda::kernel::enqueue_launch(*this, stream, launch_config,
		argument<C*>(arguments, "compressed_input"), reinterpret_cast<cuda::array<int, 1>*>(argument<int*>(arguments, "segment_model_coefficients")),
		argument<const U*>(arguments, "input"), (size_type) argument<uint_t<I>>(arguments, "length"),
		(size_type) argument<uint_t<I>>(arguments, "segment_length"),
		optional_argument<native_word_t*>(arguments, "unrepresentable_count").value_or(nullptr));
}

template <unsigned I, typename U, typename C>
const device_function_t kernel_t<I, U, C>::get_device_function() const { return reinterpret_cast<const void*>(compress_with_constant_model<I, U, C>); }

namespace linear {

/// as frame_of_reference::kernel_t, with a fitted line per segment: "segment_model_coefficients" int* holds two values per segment
template <unsigned IndexSize, typename Uncompressed, typename Compressed>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);
};

template <unsigned I, typename U, typename C>
launch_configuration_t kernel_t<I, U, C>::resolve_launch_configuration(device::properties_t, device_function::attributes_t, arguments_type extra_arguments, launch_configuration_limits_t) const
{
	auto segment_length = argument<size_t>(extra_arguments, "segment_length");
	if (segment_length == 0) { throw std::invalid_argument("The segment length must be a positive number"); }
	return tile_launch(util::div_rounding_up(argument<size_t>(extra_arguments, "length"), segment_length));
}

template <unsigned I, typename U, typename C>
void kernel_t<I, U, C>::enqueue_launch_with(stream::id_t stream, const launch_configuration_t& launch_config, const arguments_type& arguments) const
{
	using size_type = size_type_by_index_size<I>;
	cuda::kernel::enqueue_launch(*this, stream, launch_config,
		argument<C*>(arguments, "compressed_input"), reinterpret_cast<cuda::array<int, 2>*>(argument<int*>(arguments, "segment_model_coefficients")),
		argument<const U*>(arguments, "input"), (size_type) argument<uint_t<I>>(arguments, "length"),
		(size_type) argument<uint_t<I>>(arguments, "segment_length"),
		optional_argument<native_word_t*>(arguments, "unrepresentable_count").value_or(nullptr));
}

template <unsigned I, typename U, typename C>
const device_function_t kernel_t<I, U, C>::get_device_function() const { return reinterpret_cast<const void*>(compress_with_linear_model<I, U, C>); }

} // namespace linear
} // namespace frame_of_reference

// ---- dictionary encoding ---------------------------------------------------------------------------------------------------
namespace dictionary {

/// "dictionary_entries" (sorted ascending, distinct), "num_dictionary_entries" (size_type of the key width)
template <unsigned IndexSize, unsigned UncompressedSize, unsigned DictionaryIndexSize>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);
};

template <unsigned I, unsigned U, unsigned D>
launch_configuration_t kernel_t<I, U, D>::resolve_launch_configuration(device::properties_t, device_function::attributes_t, arguments_type extra_arguments, launch_configuration_limits_t) const
{
	return elementwise_launch(argument<size_t>(extra_arguments, "length"));
}

template <unsigned I, unsigned U, unsigned D>
void kernel_t<I, U, D>::enqueue_launch_with(stream::id_t stream, const launch_configuration_t& launch_config, const arguments_type& arguments) const
{
	cuda::kernel::enqueue_launch(*this, stream, launch_config,
		argument<uint_t<D>*>(arguments, "compressed_input"), argument<const uint_t<U>*>(arguments, "input"),
		argument<const uint_t<U>*>(arguments, "dictionary_entries"), (size_type_by_index_size<I>) argument<uint_t<I>>(arguments, "length"),
		(size_type_by_index_size<D>) argument<size_type_by_index_size<D>>(arguments, "num_dictionary_entries"),
		optional_argument<native_word_t*>(arguments, "unrepresentable_count").value_or(nullptr));
}

template <unsigned I, unsigned U, unsigned D>
const device_function_t kernel_t<I, U, D>::get_device_function() const { return reinterpret_cast<const void*>(encode<I, U, D>); }

} // namespace dictionary

namespace dictionary {
namespace construction {

/// The column's distinct values, sorted ascending.  Launch keys: "input", "length";
/// output "dictionary_entries" uint_t<U>* (room for 65 536); scratch, ZEROED by the caller: "slot_states" unsigned* (2^18 entries),
/// "status" unsigned* (2 entries: [0] = distinct values found — compare with the key width's limit —, [1] = entries written);
/// scratch: "slot_values" uint64_t* (2^18 entries), "unsorted_entries" uint_t<U>* (65 536 entries).
/// LARGER dictionaries (4- and 8-byte keys): "capacity" (unsigned: entries there is room for; more than 65 536) and "hash_slots" (unsigned: a power of two,
/// at least twice the capacity — four times keeps the probes short) size the arrays instead: "slot_states" / "slot_values" of hash_slots entries,
/// "dictionary_entries" of the capacity rounded up to a power of two ("unsorted_entries" is not used: the entries are sorted in place).
template <unsigned IndexSize, unsigned UncompressedSize>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);
};

template <unsigned I, unsigned U>
launch_configuration_t kernel_t<I, U>::resolve_launch_configuration(device::properties_t, device_function::attributes_t, arguments_type extra_arguments, launch_configuration_limits_t) const
{
	// (a grid of at most insert_blocks workgroups at grid stride: each fills its LDS cache of known values once for many elements)
	auto config = elementwise_launch(argument<size_t>(extra_arguments, "length"));
	if (config.grid_dimensions.x > insert_blocks) { config.grid_dimensions.x = insert_blocks; }
	return config;
}

template <unsigned I, unsigned U>
void kernel_t<I, U>::enqueue_launch_with(stream::id_t stream, const launch_configuration_t& launch_config, const arguments_type& arguments) const
{
	auto states   = argument<unsigned*>(arguments, "slot_states");
	auto values   = argument<uint64_t*>(arguments, "slot_values");
	auto status   = argument<unsigned*>(arguments, "status");
	auto length   = (size_type_by_index_size<I>) argument<uint_t<I>>(arguments, "length");
	const unsigned capacity = optional_argument<unsigned>(arguments, "capacity").value_or((unsigned) max_entries);
	const unsigned slots = optional_argument<unsigned>(arguments, "hash_slots").value_or((unsigned) hash_slots);
	// (at least two slots per entry for every capacity: a table that may fill completely makes every later insert probe all of it)
	if (capacity == 0 || (slots & (slots - 1)) != 0 || (uint64_t) slots < 2 * (uint64_t) capacity) {
		throw std::invalid_argument("dictionary construction: hash_slots must be a power of two, at least twice the capacity");
	}
	if (length == 0) { return; }
	cuda::kernel::enqueue_launch(*this, stream, launch_config, states, values, status, argument<const uint_t<U>*>(arguments, "input"), length, slots, capacity);
	auto* entries = argument<uint_t<U>*>(arguments, "dictionary_entries");
	const launch_configuration_t over_slots((grid_dimension_t) std::min<unsigned>(std::max<unsigned>((slots + threads_per_block - 1) / threads_per_block, 1u), 1u << 16), threads_per_block);
	if (capacity <= max_entries) {
		auto unsorted = argument<uint_t<U>*>(arguments, "unsorted_entries");
		cuda::enqueue_launch(reinterpret_cast<const void*>(collect_distinct<U>), stream, over_slots,
			unsorted, status, static_cast<const unsigned*>(states), static_cast<const uint64_t*>(values), capacity, slots);
		cuda::enqueue_launch(reinterpret_cast<const void*>(place_by_rank<U>), stream, launch_configuration_t(max_entries / threads_per_block, threads_per_block),
			entries, static_cast<const uint_t<U>*>(unsorted), static_cast<const unsigned*>(status), capacity);
		return;
	}
	// more than 65 536 entries: collected into the entries array itself, padded to a power of two with the largest value, sorted in place by a bitonic network
	unsigned padded = 1;
	while (padded < capacity) { padded <<= 1; }
	cuda::enqueue_launch(reinterpret_cast<const void*>(collect_distinct<U>), stream, over_slots,
		entries, status, static_cast<const unsigned*>(states), static_cast<const uint64_t*>(values), capacity, slots);
	const launch_configuration_t over_entries((grid_dimension_t) std::min<unsigned>(std::max<unsigned>(padded / 2 / threads_per_block, 1), 1u << 16), threads_per_block);
	const launch_configuration_t over_tiles((grid_dimension_t) std::min<unsigned>(std::max<unsigned>(padded / sorted_in_lds, 1), 1u << 16), threads_per_block);
	cuda::enqueue_launch(reinterpret_cast<const void*>(pad_for_sorting<U>), stream, over_entries, entries, static_cast<const unsigned*>(status), capacity, padded);
	cuda::enqueue_launch(reinterpret_cast<const void*>(bitonic_steps_in_lds<U>), stream, over_tiles, entries, padded, 0u, true);   // every stage up to sorted_in_lds
	for (unsigned k = 2 * sorted_in_lds; k <= padded && k != 0; k <<= 1) {
		for (unsigned j = k / 2; j >= sorted_in_lds; j >>= 1) {
			cuda::enqueue_launch(reinterpret_cast<const void*>(bitonic_step<U>), stream, over_entries, entries, padded, j, k);
		}
		cuda::enqueue_launch(reinterpret_cast<const void*>(bitonic_steps_in_lds<U>), stream, over_tiles, entries, padded, k, false);   // this stage's j < sorted_in_lds
	}
}

template <unsigned I, unsigned U>
const device_function_t kernel_t<I, U>::get_device_function() const { return reinterpret_cast<const void*>(insert_distinct<I, U>); }

} // namespace construction
} // namespace dictionary

// ---- incidence bitmaps --------------------------------------------------------------------------------------------------------
namespace incidence_bitmaps {

/// "keys" const unsigned char* (the column's dictionary keys), "bitmap_length" (32-bit words), "num_bitmaps" (unsigned),
/// output "incidence_bitmaps" unsigned* (num_bitmaps * round_up(bitmap_length, BitmapAlignmentInInts) words)
template <unsigned IndexSize, unsigned BitmapAlignmentInInts>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);
};

template <unsigned I, unsigned A>
launch_configuration_t kernel_t<I, A>::resolve_launch_configuration(device::properties_t, device_function::attributes_t, arguments_type extra_arguments, launch_configuration_limits_t) const
{
	return tile_launch(util::div_rounding_up(argument<size_t>(extra_arguments, "length"), (size_t) 64 * waves_per_block));
}

template <unsigned I, unsigned A>
void kernel_t<I, A>::enqueue_launch_with(stream::id_t stream, const launch_configuration_t& launch_config, const arguments_type& arguments) const
{
	auto num_bitmaps = argument<native_word_t>(arguments, "num_bitmaps");
	if (num_bitmaps > 256) { throw std::invalid_argument("At most 256 incidence bitmaps are supported"); }
	cuda::kernel::enqueue_launch(*this, stream, launch_config,
		argument<standard_bit_container_t*>(arguments, "incidence_bitmaps"), argument<const uint8_t*>(arguments, "keys"),
		(size_type_by_index_size<I>) argument<size_type_by_index_size<I>>(arguments, "bitmap_length"), num_bitmaps);
}

template <unsigned I, unsigned A>
const device_function_t kernel_t<I, A>::get_device_function() const { return reinterpret_cast<const void*>(compress<I, A>); }

} // namespace incidence_bitmaps

// ---- run-length and run-position encoding ------------------------------------------------------------------------------------
namespace run_length_encoding {

/// ("chain_patience" uint32_t, "chain_fault_tile" uint64_t — optional: how long a tile waits for the tiles before it, and a tile that plays dead for
/// a test; a chain that breaks leaves 2^64 - 1 in "num_element_runs": kernels/compression/chained.hpp, "Progress")
/// outputs "run_data", "run_lengths" (room for "capacity" runs — optional size_t, default `length`, the worst case; runs beyond it are
/// counted but not stored), "position_anchors", "intra_run_anchor_offsets", "num_element_runs" uint64_t* (device scalar, written: the TRUE
/// number of runs, so a caller who guessed too low can tell); scratch "tile_totals" (scratch_entries(length) entries).  ONE launch:
/// kernels/compression/runs.hpp, one_pass::encode_run_lengths.
template <unsigned IndexSize, unsigned UncompressedSize, unsigned RunLengthSize, unsigned RunIndexSize = IndexSize>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);
};

template <unsigned I, unsigned U, unsigned RL, unsigned RI>
launch_configuration_t kernel_t<I, U, RL, RI>::resolve_launch_configuration(device::properties_t, device_function::attributes_t, arguments_type extra_arguments, launch_configuration_limits_t) const
{
	if (argument<size_t>(extra_arguments, "segment_length") == 0) { throw std::invalid_argument("The segment length must be a positive number"); }
	return tile_launch(util::div_rounding_up(argument<size_t>(extra_arguments, "length"), (size_t) tile_elements));
}

template <unsigned I, unsigned U, unsigned RL, unsigned RI>
void kernel_t<I, U, RL, RI>::enqueue_launch_with(stream::id_t stream, const launch_configuration_t&, const arguments_type& arguments) const
{
	using size_type = size_type_by_index_size<I>;
	auto input     = argument<const uint_t<U>*>(arguments, "input");
	auto length    = (size_type) argument<uint_t<I>>(arguments, "length");
	auto segment   = (size_type) argument<uint_t<I>>(arguments, "segment_length");
	auto num_runs  = argument<uint64_t*>(arguments, "num_element_runs");
	if (length == 0) { return; }
	// with "run_scratch": three launches (below); otherwise ONE launch (kernels/compression/runs.hpp, one_pass) — the column is read once either way
	const uint64_t num_tiles = util::div_rounding_up((uint64_t) length, (uint64_t) runs::one_pass::tile);
	const uint64_t max_run_length = ~(uint64_t) 0 >> (64 - 8 * RL);   // the largest value a length field holds
	// "capacity" (optional, size_t): pieces the output arrays have room for — the true count is reported all the same, so that a caller
	// who sized by an estimate can tell and come back with bigger buffers
	const uint64_t capacity = optional_argument<size_t>(arguments, "capacity").value_or((size_t) length);
	if constexpr (RL >= 2) {
		const auto scratch = optional_argument<uint64_t*>(arguments, "run_scratch");
		if (scratch && !detail::chain_is_asked_for(arguments)) {   // three launches, no chain (kernels/compression/listed_runs.hpp)
			const detail::listed_runs<U> listed(stream, scratch.value(), argument<size_t>(arguments, "run_scratch_entries"), input, (uint64_t) length);
			cuda::enqueue_launch(reinterpret_cast<const void*>(runs::listed::summarize_wavefronts<true>), stream, listed.per_blocks,
				listed.bases, listed.previous_starts, listed.next_starts, listed.block_sums, listed.block_first_starts, listed.block_last_starts, (const uint64_t*) listed.headers, listed.num_waves);
			cuda::enqueue_launch(reinterpret_cast<const void*>(runs::listed::count_pieces<>), stream, listed.per_blocks,
				listed.bases, listed.block_sums, listed.block_next_starts, listed.block_previous_starts, (const uint64_t*) listed.next_starts,
				(const uint64_t*) listed.block_first_starts, (const uint64_t*) listed.block_last_starts, (const uint64_t*) listed.headers, listed.num_waves, listed.num_blocks,
				(uint64_t) length, max_run_length);
			listed.place_blocks(stream, num_runs);
			auto place = [&](const void* kernel, const launch_configuration_t& grid) {
				cuda::enqueue_launch(kernel, stream, grid,
					argument<uint_t<U>*>(arguments, "run_data"), argument<uint_t<RL>*>(arguments, "run_lengths"),
					argument<uint_t<RI>*>(arguments, "position_anchors"), argument<uint_t<RL>*>(arguments, "intra_run_anchor_offsets"), input,
					(const uint64_t*) listed.headers, (const uint64_t*) listed.bases, (const uint64_t*) listed.block_bases,
					(const uint64_t*) listed.previous_starts, (const uint64_t*) listed.next_starts, (const uint64_t*) listed.block_previous_starts, (const uint64_t*) listed.block_next_starts,
					(const uint16_t*) listed.offsets, (const uint_t<U>*) listed.values, (uint32_t) listed.staged_per_wave, segment, length, max_run_length, capacity);
			};
			place(reinterpret_cast<const void*>(runs::listed::place_run_lengths<I, U, RL, RI, false>), listed.per_wavefronts);
			place(reinterpret_cast<const void*>(runs::listed::place_run_lengths<I, U, RL, RI, true>), listed.relisting);
			return;
		}
	}
	auto totals = argument<uint64_t*>(arguments, "tile_totals");
	enqueue_zero(stream, totals, 2 * num_tiles * sizeof(uint64_t));
	cuda::enqueue_launch(reinterpret_cast<const void*>(runs::one_pass::encode_run_lengths<I, U, RL, RI>), stream, chained_launch(num_tiles),
		argument<uint_t<U>*>(arguments, "run_data"), argument<uint_t<RL>*>(arguments, "run_lengths"),
		argument<uint_t<RI>*>(arguments, "position_anchors"), argument<uint_t<RL>*>(arguments, "intra_run_anchor_offsets"), num_runs,
		input, totals, totals + num_tiles, segment, length, max_run_length, capacity,
		optional_argument<uint32_t>(arguments, "chain_patience").value_or((uint32_t) chained::default_patience),
		optional_argument<uint64_t>(arguments, "chain_fault_tile").value_or(chained::no_fault_tile));
}

template <unsigned I, unsigned U, unsigned RL, unsigned RI>
const device_function_t kernel_t<I, U, RL, RI>::get_device_function() const { return reinterpret_cast<const void*>(runs::one_pass::encode_run_lengths<I, U, RL, RI>); }

} // namespace run_length_encoding

namespace run_position_encoding {

/// outputs "run_data", "run_start_positions" (room for "capacity" runs — optional size_t, default `length`; runs beyond it are counted, not
/// stored), "position_anchors", "num_element_runs" uint64_t* (the true number of runs); scratch "tile_totals"
template <unsigned IndexSize, unsigned UncompressedSize, unsigned PositionOffsetSize, bool PositionsAreRelative, unsigned RunIndexSize = IndexSize>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);
};

template <unsigned I, unsigned U, unsigned P, bool R, unsigned RI>
launch_configuration_t kernel_t<I, U, P, R, RI>::resolve_launch_configuration(device::properties_t, device_function::attributes_t, arguments_type extra_arguments, launch_configuration_limits_t) const
{
	auto segment_length = argument<size_t>(extra_arguments, "segment_length");
	if (segment_length == 0) { throw std::invalid_argument("The segment length must be a positive number"); }
	if (R && P < 8 && (segment_length - 1) > (~(uint64_t) 0 >> (64 - 8 * (P < 8 ? P : 7)))) { throw std::invalid_argument("The segment is too long for relative positions of this width"); }
	return tile_launch(util::div_rounding_up(argument<size_t>(extra_arguments, "length"), (size_t) tile_elements));
}

template <unsigned I, unsigned U, unsigned P, bool R, unsigned RI>
void kernel_t<I, U, P, R, RI>::enqueue_launch_with(stream::id_t stream, const launch_configuration_t&, const arguments_type& arguments) const
{
	using size_type = size_type_by_index_size<I>;
	auto input   = argument<const uint_t<U>*>(arguments, "input");
	auto length  = (size_type) argument<uint_t<I>>(arguments, "length");
	auto segment = (size_type) argument<uint_t<I>>(arguments, "segment_length");
	if (length == 0) { return; }
	const auto scratch = optional_argument<uint64_t*>(arguments, "run_scratch");
	if (scratch && !detail::chain_is_asked_for(arguments)) {   // three launches, no chain (kernels/compression/listed_runs.hpp)
		const detail::listed_runs<U> listed(stream, scratch.value(), argument<size_t>(arguments, "run_scratch_entries"), input, (uint64_t) length);
		cuda::enqueue_launch(reinterpret_cast<const void*>(runs::listed::summarize_wavefronts<false>), stream, listed.per_blocks,
			listed.bases, (uint64_t*) nullptr, (uint64_t*) nullptr, listed.block_sums, (uint64_t*) nullptr, (uint64_t*) nullptr, (const uint64_t*) listed.headers, listed.num_waves);
		listed.place_blocks(stream, argument<uint64_t*>(arguments, "num_element_runs"));
		auto place = [&](const void* kernel, const launch_configuration_t& grid) {
			cuda::enqueue_launch(kernel, stream, grid,
				argument<uint_t<U>*>(arguments, "run_data"), argument<uint_t<P>*>(arguments, "run_start_positions"), argument<uint_t<RI>*>(arguments, "position_anchors"), input,
				(const uint64_t*) listed.headers, (const uint64_t*) listed.bases, (const uint64_t*) listed.block_bases, (const uint16_t*) listed.offsets, (const uint_t<U>*) listed.values,
				(uint32_t) listed.staged_per_wave, segment, length, (uint64_t) optional_argument<size_t>(arguments, "capacity").value_or((size_t) length));
		};
		place(reinterpret_cast<const void*>(runs::listed::place_run_positions<I, U, P, R, RI, false>), listed.per_wavefronts);
		place(reinterpret_cast<const void*>(runs::listed::place_run_positions<I, U, P, R, RI, true>), listed.relisting);
		return;
	}
	// ONE launch (kernels/compression/runs.hpp, one_pass): the column is read once
	const uint64_t num_tiles = util::div_rounding_up((uint64_t) length, (uint64_t) runs::one_pass::tile);
	const uint64_t num_links = util::div_rounding_up(num_tiles, (uint64_t) runs::one_pass::tiles_per_link);   // (a workgroup — a link of the chain — takes that many consecutive tiles)
	auto totals = argument<uint64_t*>(arguments, "tile_totals");
	enqueue_zero(stream, totals, num_links * sizeof(uint64_t));
	cuda::enqueue_launch(reinterpret_cast<const void*>(runs::one_pass::encode_run_positions<I, U, P, R, RI>), stream, chained_launch(num_links),
		argument<uint_t<U>*>(arguments, "run_data"), argument<uint_t<P>*>(arguments, "run_start_positions"),
		argument<uint_t<RI>*>(arguments, "position_anchors"), argument<uint64_t*>(arguments, "num_element_runs"), input, totals, segment, length,
		(uint64_t) optional_argument<size_t>(arguments, "capacity").value_or((size_t) length),
		optional_argument<uint32_t>(arguments, "chain_patience").value_or((uint32_t) chained::default_patience),
		optional_argument<uint64_t>(arguments, "chain_fault_tile").value_or(chained::no_fault_tile));
}

template <unsigned I, unsigned U, unsigned P, bool R, unsigned RI>
const device_function_t kernel_t<I, U, P, R, RI>::get_device_function() const { return reinterpret_cast<const void*>(runs::one_pass::encode_run_positions<I, U, P, R, RI>); }

} // namespace run_position_encoding

// ---- compressed patch positions -------------------------------------------------------------------------------------------------
namespace compressed_indices {

/// "positions" const uint_t<OutputIndexSize>* (ascending, distinct), "input_data_length", "anchoring_period"; outputs under the
/// compressed-indices scatter's names (room: `input_data_length` runs and input_data_length * OutputIndexSize * 2 + 8 pool bytes at
/// worst); "num_scatter_position_runs", "num_offset_bytes": uint64_t* device scalars; scratch "tile_totals": 2 * (blocks + 2) entries
template <unsigned OutputIndexSize, unsigned InputIndexSize, unsigned RunLengthSize = InputIndexSize>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);
};

template <unsigned O, unsigned I, unsigned R>
launch_configuration_t kernel_t<O, I, R>::resolve_launch_configuration(device::properties_t, device_function::attributes_t, arguments_type extra_arguments, launch_configuration_limits_t) const
{
	auto n = argument<size_t>(extra_arguments, "input_data_length");
	auto period = argument<size_t>(extra_arguments, "anchoring_period");
	if (n == 0) { throw std::invalid_argument("Zero-length scatters not currently supported"); }
	if (period == 0) { throw std::invalid_argument("The anchoring period must be a positive number"); }
	return tile_launch(util::div_rounding_up(util::div_rounding_up(n, period), (size_t) threads_per_block));
}

template <unsigned O, unsigned I, unsigned R>
void kernel_t<O, I, R>::enqueue_launch_with(stream::id_t stream, const launch_configuration_t& launch_config, const arguments_type& arguments) const
{
	using size_type = size_type_by_index_size<I>;
	auto positions = argument<const uint_t<O>*>(arguments, "positions");
	auto n         = (size_type) argument<size_type>(arguments, "input_data_length");
	auto period    = argument<uint_t<I>>(arguments, "anchoring_period");
	auto totals    = argument<uint64_t*>(arguments, "tile_totals");
	if (period == 0) { throw std::invalid_argument("The anchoring period must be a positive number"); }
	const uint64_t blocks = util::div_rounding_up((uint64_t) n, (uint64_t) period);
	uint64_t* block_runs = totals, * block_bytes = totals + blocks + 2;
	cuda::enqueue_launch(reinterpret_cast<const void*>(count_runs<O, I, R>), stream, launch_config, block_runs, block_bytes, positions, n, period);
	enqueue_exclusive_offsets(stream, block_runs, blocks, argument<uint64_t*>(arguments, "num_scatter_position_runs"));
	enqueue_exclusive_offsets(stream, block_bytes, blocks, optional_argument<uint64_t*>(arguments, "num_offset_bytes").value_or(nullptr));
	cuda::kernel::enqueue_launch(*this, stream, launch_config,
		argument<uint_t<R>*>(arguments, "scatter_position_run_lengths"), argument<unsigned char*>(arguments, "scatter_position_run_individual_offset_sizes"),
		argument<uint_t<O>*>(arguments, "scatter_position_run_baseline_values"), argument<uint_t<I>*>(arguments, "scatter_position_run_offsets_start_pos"),
		argument<unsigned char*>(arguments, "scatter_position_offset_bytes"), argument<uint_t<I>*>(arguments, "scatter_position_anchors"),
		positions, static_cast<const uint64_t*>(block_runs), static_cast<const uint64_t*>(block_bytes), n, period);
}

template <unsigned O, unsigned I, unsigned R>
const device_function_t kernel_t<O, I, R>::get_device_function() const { return reinterpret_cast<const void*>(write_runs<O, I, R>); }

} // namespace compressed_indices

// ---- patches: where a decoded column differs from the true one ---------------------------------------------------------------
namespace patches {

/// "decoded", "truth" (const uint_t<ElementSize>*), outputs "patch_positions" (ascending), "patch_values" (room for "capacity" patches —
/// optional size_t, default `length`; patches beyond it are counted, not stored), "num_patches" uint64_t* (the true number); scratch "tile_totals"
template <unsigned IndexSize, unsigned ElementSize>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);
};

template <unsigned I, unsigned E>
launch_configuration_t kernel_t<I, E>::resolve_launch_configuration(device::properties_t, device_function::attributes_t, arguments_type extra_arguments, launch_configuration_limits_t) const
{
	return tile_launch(util::div_rounding_up(argument<size_t>(extra_arguments, "length"), (size_t) tile_elements));
}

template <unsigned I, unsigned E>
void kernel_t<I, E>::enqueue_launch_with(stream::id_t stream, const launch_configuration_t&, const arguments_type& arguments) const
{
	using size_type = size_type_by_index_size<I>;
	auto decoded = argument<const uint_t<E>*>(arguments, "decoded");
	auto truth   = argument<const uint_t<E>*>(arguments, "truth");
	auto length  = (size_type) argument<uint_t<I>>(arguments, "length");
	auto totals  = argument<uint64_t*>(arguments, "tile_totals");
	if (length == 0) { return; }
	const uint64_t num_tiles = util::div_rounding_up((uint64_t) length, (uint64_t) tile_elements);
	const auto config = tile_launch(num_tiles);
	cuda::enqueue_launch(reinterpret_cast<const void*>(count_differences<I, E>), stream, config, totals, decoded, truth, length);
	enqueue_exclusive_offsets(stream, totals, num_tiles, argument<uint64_t*>(arguments, "num_patches"));
	cuda::enqueue_launch(reinterpret_cast<const void*>(write_differences<I, E>), stream, config,
		argument<uint_t<I>*>(arguments, "patch_positions"), argument<uint_t<E>*>(arguments, "patch_values"), decoded, truth,
		static_cast<const uint64_t*>(totals), length, (uint64_t) optional_argument<size_t>(arguments, "capacity").value_or((size_t) length));
}

template <unsigned I, unsigned E>
const device_function_t kernel_t<I, E>::get_device_function() const { return reinterpret_cast<const void*>(write_differences<I, E>); }

} // namespace patches

} // namespace compression
} // namespace kernels
} // namespace cuda

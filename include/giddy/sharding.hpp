// A column cut over the GPUs of a node — or into the chunks of a host-to-device pipeline — in C++.
//
// Every scheme is elementwise or anchored per fixed-length output segment, so a shard is a contiguous range of whole segments and
// decodes without talking to any other shard.  What a shard needs is a SLICE of each of the column's arrays
// and, for the variable-rate schemes, its anchors REBASED to those slices.  This header says which slices and which biases —
// the same cuts, bit for bit, as libgiddy_amd/sharding.py (tests/test_cpp_sharding.py compares them for every scheme at world
// sizes 1, 2, 3 and 8); giddy/kernels/data_layout/rebase.hpp holds the device kernel that rebases an anchor slice, and the
// C ABI (giddy.h: giddy_shard_*) carries both to callers who cannot instantiate templates.
//
// Where the reference is: it has no multi-GPU layer (last row); the anchor semantics the cuts rely on are its
// kernels' (discard_zero_bytes_variable.cuh:415-437: a segment's bytes begin at its anchor; run_length_encoding.cuh:271-326: a
// segment begins inside run anchors[s] at offset intra_run_anchor_offsets[s]; run_position_encoding.cuh:354: absolute positions
// index the whole output, which is why a shard's are biased by its first element).
#pragma once
#include <cstddef>
#include <cstdint>
#include <numeric>
#include <stdexcept>

namespace cuda {
namespace sharding {

enum : uint64_t { page_bytes = 64 * 1024 };   // shard boundaries fall on multiples of this many OUTPUT bytes where segments tile it

/// [first, end) of an array, in elements of that array
struct slice_t { uint64_t first = 0, end = 0; uint64_t size() const { return end - first; } };

/// Contiguous, balanced split of `num_segments` segments into `world_size` ranges whose boundaries are multiples of `granule`
/// segments (the last range takes the remainder).
inline slice_t segment_range(uint64_t num_segments, uint64_t rank, uint64_t world_size, uint64_t granule = 1)
{
	if (world_size == 0 || rank >= world_size || granule == 0) { throw std::invalid_argument("sharding: rank must be below a positive world size, the granule positive"); }
	const uint64_t units = (num_segments + granule - 1) / granule;
	slice_t r;
	r.first = std::min(units * rank / world_size * granule, num_segments);
	r.end = std::min(units * (rank + 1) / world_size * granule, num_segments);
	return r;
}

/// Segments per shard-boundary granule so that a shard's output begins a whole number of 64 KiB pages into the column's (the
/// wave-per-segment decoders write a workgroup's four segments into one such page) — only where segments tile a page;
/// `granule`: a granule the scheme needs anyway (the result is a multiple of it).
inline uint64_t page_granule(uint64_t segment_length, uint64_t output_itemsize, uint64_t granule = 1)
{
	const uint64_t segment_bytes = segment_length * output_itemsize;
	if (segment_bytes == 0 || page_bytes % segment_bytes != 0) { return granule; }
	return std::lcm(granule, page_bytes / segment_bytes);
}

/// A shard: output elements [elements.first, elements.end) = segments [segments.first, segments.end) of the column, and for every
/// array of the scheme the slice the shard reads.  Biases: what to subtract (wrapping, in the array's own width) from the
/// shard's slice of that array so that it refers to the shard's slices.
struct shard_t {
	slice_t segments, elements;
	// per-element / per-segment arrays (compressed_input; anchor_values, segment_model_coefficients, position_anchors, intra_run_anchor_offsets)
	//   are sliced by `elements` and `segments`
	slice_t units;                 // second-level array: compressed_data BYTES (discard-zero-bytes) or RUNS (run_data, run_lengths, run_start_positions)
	slice_t size_containers;       // packed_element_sizes (variable-width discard-zero-bytes)
	uint64_t anchor_bias = 0;      // position_anchors[segments] -= anchor_bias   (= units.first)
	uint64_t position_bias = 0;    // absolute run_start_positions[units] -= position_bias   (= elements.first)
	uint64_t length() const { return elements.size(); }
};

namespace detail {

inline shard_t by_segments(uint64_t length, uint64_t segment_length, uint64_t rank, uint64_t world_size, uint64_t granule)
{
	if (segment_length == 0) { throw std::invalid_argument("sharding: the segment length must be positive"); }
	const uint64_t num_segments = (length + segment_length - 1) / segment_length;
	shard_t s;
	s.segments = segment_range(num_segments, rank, world_size, granule);
	s.elements.first = std::min(s.segments.first * segment_length, length);
	s.elements.end = std::min(s.segments.end * segment_length, length);
	return s;
}

} // namespace detail

/// Dictionary, fixed-width discard-zero-bytes, Model: no anchors — split on multiples of `granule` elements (64 Ki: whole 64 KiB output
/// pages whatever the element size); side tables (the dictionary, the model's coefficients) go to every shard.
inline shard_t elementwise(uint64_t length, uint64_t rank, uint64_t world_size, uint64_t granule = page_bytes)
{
	return detail::by_segments(length, granule, rank, world_size, 1);
}

/// Frame of reference: compressed_input[elements], segment_model_coefficients[segments]
inline shard_t frame_of_reference(uint64_t length, uint64_t segment_length, uint64_t output_itemsize, uint64_t rank, uint64_t world_size)
{
	return detail::by_segments(length, segment_length, rank, world_size, page_granule(segment_length, output_itemsize));
}

/// Delta: compressed_input[elements], anchor_values[segments] (values, not offsets: nothing to rebase)
inline shard_t delta(uint64_t length, uint64_t segment_length, uint64_t output_itemsize, uint64_t rank, uint64_t world_size)
{
	return detail::by_segments(length, segment_length, rank, world_size, page_granule(segment_length, output_itemsize));
}

/// What the variable-rate cuts need to know of the column's position_anchors: its length and the values at two indices
/// (`at(i)` is called with i < count only) — a host array, or two scalar reads from the device (giddy_shard_* does the latter).
template <typename AnchorAt>
inline uint64_t anchor_or(AnchorAt&& at, uint64_t index, uint64_t count, uint64_t otherwise) { return index < count ? (uint64_t) at(index) : otherwise; }

/// Variable-width discard-zero-bytes: boundaries on multiples of lcm(segment_length, size fields per container) elements, so that a
/// shard begins on a whole container; compressed_data[units] (bytes, from the anchors), packed_element_sizes[size_containers],
/// position_anchors[segments] - anchor_bias.
template <typename AnchorAt>
inline shard_t discard_zero_bytes_variable(uint64_t length, uint64_t segment_length, unsigned bits_per_element_size, unsigned container_size,
	uint64_t output_itemsize, uint64_t num_anchors, uint64_t total_bytes, AnchorAt&& anchor_at, uint64_t rank, uint64_t world_size)
{
	if (bits_per_element_size == 0 || bits_per_element_size > 8 * container_size) { throw std::invalid_argument("sharding: size fields between 1 bit and a container wide"); }
	const uint64_t per = (8 * container_size) / bits_per_element_size;
	uint64_t granule = std::lcm(segment_length, per) / segment_length;
	granule = page_granule(segment_length, output_itemsize, granule);
	shard_t s = detail::by_segments(length, segment_length, rank, world_size, granule);
	s.units.first = anchor_or(anchor_at, s.segments.first, num_anchors, total_bytes);
	s.units.end = anchor_or(anchor_at, s.segments.end, num_anchors, total_bytes);
	s.size_containers.first = s.elements.first / per;
	s.size_containers.end = (s.elements.end + per - 1) / per;
	s.anchor_bias = s.segments.size() > 0 ? s.units.first : 0;
	return s;
}

/// Run-length encoding: run_data / run_lengths [units] — the run straddling the shard's end included —, position_anchors[segments]
/// - anchor_bias, intra_run_anchor_offsets[segments].  An empty shard (more ranks than segments) has empty slices.
template <typename AnchorAt>
inline shard_t run_length_encoding(uint64_t length, uint64_t segment_length, uint64_t output_itemsize, uint64_t num_anchors, uint64_t num_element_runs,
	AnchorAt&& anchor_at, uint64_t rank, uint64_t world_size)
{
	shard_t s = detail::by_segments(length, segment_length, rank, world_size, page_granule(segment_length, output_itemsize));
	if (s.segments.size() == 0) { return s; }
	s.units.first = (uint64_t) anchor_at(s.segments.first);
	s.units.end = s.segments.end < num_anchors ? (uint64_t) anchor_at(s.segments.end) + 1 : num_element_runs;
	s.anchor_bias = s.units.first;
	return s;
}

/// Run-position encoding: as run-length encoding; ABSOLUTE start positions index the whole output, a shard's are biased by its first element
template <typename AnchorAt>
inline shard_t run_position_encoding(uint64_t length, uint64_t segment_length, uint64_t output_itemsize, uint64_t num_anchors, uint64_t num_element_runs,
	bool positions_are_relative, AnchorAt&& anchor_at, uint64_t rank, uint64_t world_size)
{
	shard_t s = run_length_encoding(length, segment_length, output_itemsize, num_anchors, num_element_runs, anchor_at, rank, world_size);
	if (s.segments.size() > 0 && !positions_are_relative) { s.position_bias = s.elements.first; }
	return s;
}

} // namespace sharding
} // namespace cuda

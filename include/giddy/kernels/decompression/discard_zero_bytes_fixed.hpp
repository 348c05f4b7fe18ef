// Discard-zero-bytes (null suppression), fixed width:   out[p] = zero_extend(compressed_input[p])
//
// Kernel template name, parameter list and semantics as in the reference's
// src/kernels/decompression/discard_zero_bytes_fixed.cuh:108-124 (pad_high_bytes, :41-77; output is
// little-endian on every supported target).  Execution: a pure HBM stream — a lane widens G
// elements from one 2..8-byte load into one 16-byte store (G = 16/UncompressedSize), K steps per
// tile, tiles at grid stride.  HBM bytes per element: CompressedSize + UncompressedSize.
#pragma once
#include <hip/hip_runtime.h>

#include "giddy/common.hpp"
#include "giddy/kernels/launch_config_resolution_params.hpp"
#include "giddy/kernels/tuning.hpp"
#include "giddy/on_device/batch_pieces.hpp"
#include "giddy/on_device/sorted_patches.hpp"
#include "giddy/on_device/stream_tiles.hpp"
#include "giddy/on_device/wave64.hpp"
#include "giddy/util/math.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace discard_zero_bytes {
namespace fixed_width {

enum : serialization_factor_t { DefaultSerializationFactor = 32 };

template <unsigned UncompressedSize, unsigned CompressedSize, cuda::endianness_t OutputEndianness>
struct pad_high_bytes {
	using argument_type = uint_t<CompressedSize>;
	using result_type = uint_t<UncompressedSize>;
	static_assert(UncompressedSize > CompressedSize, "Uncompressed size isn't higher than Compressed size");
	static_assert(OutputEndianness == cuda::endianness_t::little, "only little-endian output is implemented (CDNA4 and x86-64 are little-endian)");
	__host__ __device__ result_type operator()(const argument_type& x) const { return (result_type) x; }
};

namespace detail {

/// the kernels' body (`AnyAlignment`: the build of stream_tiles::transform for arrays aligned for their elements only; `grid`: launch_grid.hpp)
template <unsigned UncompressedSize, unsigned CompressedSize, cuda::endianness_t UncompressedEndianness, bool AnyAlignment = false, typename Grid = launch_grid::whole>
__device__ __forceinline__ void decompress(
	uint_t<UncompressedSize>*      __restrict__  decompressed,
	const uint_t<CompressedSize>*  __restrict__  compressed_input,
	uint64_t                                     length,
	Grid                                         grid = Grid())
{
	using U = uint_t<UncompressedSize>;
	using C = uint_t<CompressedSize>;
	constexpr unsigned G = stream_tiles::group_size<U>::value;
	pad_high_bytes<UncompressedSize, CompressedSize, UncompressedEndianness> widen;
	stream_tiles::transform<C, U, stream_tiles::steps_per_tile, memory_access::non_temporal, AnyAlignment>(decompressed, compressed_input, length,
		[&](uint64_t, const memory_access::group<C, G>& in, memory_access::group<U, G>& out) {
			#pragma unroll
			for (unsigned j = 0; j < G; j++) { out[j] = widen(in[j]); }
		},
		[&](uint64_t, C x) { return widen(x); },
		stream_tiles::nothing_to_do(), stream_tiles::nothing_to_prepare(), grid);
}

} // namespace detail

template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize,
	cuda::endianness_t UncompressedEndianness = cuda::compilation_target_endianness,
	serialization_factor_t SerializationFactor = DefaultSerializationFactor>
__global__ void decompress(
	uint_t<UncompressedSize>*      __restrict__  decompressed,
	const uint_t<CompressedSize>*  __restrict__  compressed_input,
	size_type_by_index_size<IndexSize>           length)
{
	detail::decompress<UncompressedSize, CompressedSize, UncompressedEndianness>(decompressed, compressed_input, length);
}

/// The same for a column at an address aligned for its elements only, or inside a 128-byte line: the `AnyAlignment` build of stream_tiles::transform, launched
/// by the wrapper from the addresses
template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize,
	cuda::endianness_t UncompressedEndianness = cuda::compilation_target_endianness,
	serialization_factor_t SerializationFactor = DefaultSerializationFactor>
__global__ void decompress_at_any_alignment(
	uint_t<UncompressedSize>*      __restrict__  decompressed,
	const uint_t<CompressedSize>*  __restrict__  compressed_input,
	size_type_by_index_size<IndexSize>           length)
{
	detail::decompress<UncompressedSize, CompressedSize, UncompressedEndianness, true>(decompressed, compressed_input, length);
}

/// A piece of a batch launch (giddy/batch.hpp; launched as batch::decode_pieces<piece_body<...>>): `decompress`'s parameters out of the piece's record, its body
/// on the piece's part of the grid
template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize, cuda::endianness_t UncompressedEndianness = cuda::compilation_target_endianness, bool AnyAlignment = false>
struct piece_body {
	enum : unsigned { max_threads_per_block = 1024 };
	static __device__ __forceinline__ void run(const batch::piece_view piece, const launch_grid::part grid)
	{
		detail::decompress<UncompressedSize, CompressedSize, UncompressedEndianness, AnyAlignment>(
			piece.get<uint_t<UncompressedSize>*>(0), piece.get<const uint_t<CompressedSize>*>(1), piece.get<size_type_by_index_size<IndexSize>>(2), grid);
	}
};

namespace patched {

/// rows of an 8-byte column take 2 or 4 input bytes per lane: a chunk's input comes in with 16-byte loads through LDS then
/// (stream_tiles::chunk_input_storage; as the fused Dictionary's narrow keys)
template <unsigned UncompressedSize> constexpr unsigned stage_input_up_to() { return UncompressedSize == 8 ? 4 : 2; }
template <unsigned UncompressedSize, unsigned CompressedSize> constexpr bool stages_chunk_input()
{
	return stream_tiles::stages_chunk_input<uint_t<CompressedSize>, uint_t<UncompressedSize>, stage_input_up_to<UncompressedSize>()>();
}

/// Fixed-width discard-zero-bytes decompression and patching in one pass: what
/// `fixed_width::decompress` followed by `data_layout::scatter::scatter(decompressed, patch_values, patch_positions,
/// num_patches)` leaves in `decompressed`, for ASCENDING patch positions (equal positions: the last one wins) — narrow
/// storage for the elements that fit, patches for the few that do not.
template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize,
	cuda::endianness_t UncompressedEndianness = cuda::compilation_target_endianness,
	serialization_factor_t SerializationFactor = DefaultSerializationFactor, bool ManyPatches = false>
__global__ void decompress(
	uint_t<UncompressedSize>*        __restrict__  decompressed,
	const uint_t<CompressedSize>*    __restrict__  compressed_input,
	const uint_t<UncompressedSize>*  __restrict__  patch_values,
	const uint_t<IndexSize>*         __restrict__  patch_positions,
	size_type_by_index_size<IndexSize>             length,
	size_type_by_index_size<IndexSize>             num_patches)
{
	using U = uint_t<UncompressedSize>;
	using C = uint_t<CompressedSize>;
	constexpr unsigned G = stream_tiles::group_size<U>::value;
	__shared__ sorted_patches::window_storage<IndexSize, U> patch_windows;
	__shared__ sorted_patches::row_storage<U, ManyPatches ? 4 : 0, 16 / sizeof(U), IndexSize> patch_rows;
	__shared__ stream_tiles::chunk_input_storage<C, U, stages_chunk_input<UncompressedSize, CompressedSize>()> chunk_staging;
	pad_high_bytes<UncompressedSize, CompressedSize, UncompressedEndianness> widen;
	stream_tiles::transform_and_patch<C, U, stream_tiles::rows_in_flight, stage_input_up_to<UncompressedSize>()>(decompressed, compressed_input, length,
		[&](uint64_t, const memory_access::group<C, G>& in, memory_access::group<U, G>& out) {
			#pragma unroll
			for (unsigned j = 0; j < G; j++) { out[j] = widen(in[j]); }
		},
		[&](uint64_t, C x) { return widen(x); },
		sorted_patches::cursor_t<IndexSize, U, ManyPatches>(patch_values, patch_positions, num_patches, length, patch_windows, patch_rows),
		stream_tiles::nothing_to_do(), &chunk_staging);
}

} // namespace patched

template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize,
	cuda::endianness_t UncompressedEndianness = cuda::compilation_target_endianness,
	serialization_factor_t SerializationFactor = DefaultSerializationFactor>
class launch_config_resolution_params_t final : public kernels::launch_config_resolution_params_t {
public:
	using parent = kernels::launch_config_resolution_params_t;

	launch_config_resolution_params_t(
		device::properties_t device_properties_, device_function::attributes_t kernel_function_attributes_, size_t length_)
	: parent(device_properties_, kernel_function_attributes_)
	{
		constexpr unsigned elements_per_thread_per_tile = stream_tiles::elements_per_thread_per_tile<uint_t<UncompressedSize>>();
		grid_construction_resolution  = thread;
		length                        = util::div_rounding_up(length_, (size_t) elements_per_thread_per_tile);
		serialization_option          = auto_maximized;
		quanta.threads_in_block       = warp_size;
		upper_limits.threads_in_block = GIDDY_STREAM_THREADS_PER_BLOCK;
		keep_gpu_busy_factor          = 1024;  // in practice: one tile per workgroup — the dispatcher hands tiles out in address order, which HBM likes best
		default_serialization_factor  = DefaultSerializationFactor;
		// fewer resident workgroups, fewer interleaved write streams: +5..9 points of HBM peak (measured per instantiation, tools/occupancy_sweep.py)
		resident_blocks_per_multiprocessor_cap = CompressedSize == 1 && UncompressedSize >= 4 ? 5u : 3u;
	}

	launch_config_resolution_params_t(device::properties_t device_properties_, size_t length_)
	: launch_config_resolution_params_t(device_properties_,
		device_function_t(decompress<IndexSize, UncompressedSize, CompressedSize, UncompressedEndianness, SerializationFactor>).attributes(),
		length_) { }
};

} // namespace fixed_width
} // namespace discard_zero_bytes
} // namespace decompression
} // namespace kernels
} // namespace cuda

// Where a fused decompress-and-patch wrapper does NOT launch its fused kernel: the plain decoder followed by a scatter — the reference's own two
// launches (README.md:54-60; scatter.cuh:42-45), which is what the fused kernels are defined to equal — wherever those two are the faster way.
//
// Measured (tools/fused_regret.py, profiles/r06_fused_regret.txt: every family over 0.1 ... 50 % of patches, fused against the two launches on one stream).  The
// fused kernels win almost everywhere — from 1 % of patches on by a factor of 1.5 to 3: the scatter's read-modify-write of nearly every output line — except
//   * 2-byte outputs, whose rows of 512 elements carry twice the patches of a 4-byte row through VALU-bound kernels: bands of densities per family, below, and
//     from about a third of the elements patched on (the scatter is then nearly a stream of its own);
//   * a dictionary of 16 KiB or more staged in LDS beside the ManyPatches build's row areas (two workgroups to a CU, 128 registers): 28 % of the HBM peak at 2 %
//     of patches where the two launches reach 47.
// The choice is made from num_patches / length, the element size and the staged dictionary's bytes alone: the same call takes the same way every time.
#pragma once
#include "giddy/kernel_wrappers/common.hpp"
#include "giddy/kernels/data_layout/scatter.hpp"
#include "giddy/on_device/sorted_patches.hpp"

#include <memory>
#include <stdexcept>
#include <string>

namespace cuda {
namespace kernels {
namespace decompression {
namespace patched {

enum class family_t { delta, dictionary, discard_zero_bytes_fixed, frame_of_reference, model };

/// `staged_dictionary_bytes`: what the fused Dictionary kernel would keep in LDS (0: looked up through L2, or another family); `many_patches_build`: whether the
/// column is one for the ManyPatches build (sorted_patches::many_patches_per_row)
inline bool decode_then_scatter_is_faster(family_t family, unsigned element_size, uint64_t num_patches, uint64_t length, size_t staged_dictionary_bytes = 0, bool many_patches_build = false)
{
	if (length == 0 || num_patches == 0) { return false; }
	const double density = (double) num_patches / (double) length;
	if (family == family_t::dictionary && element_size == 8 && many_patches_build && staged_dictionary_bytes >= 16384) { return true; }
	// (... and just under the ManyPatches threshold, 1.3 %, beside a dictionary of 8 KiB or more: the walk that takes one patch at a time is at 55 % there, the two launches at 60)
	if (family == family_t::dictionary && element_size == 8 && !many_patches_build && staged_dictionary_bytes >= 8192 && density >= 0.008) { return true; }
	if (element_size != 2) { return false; }
	switch (family) {
	case family_t::delta:               return (density >= 0.0015 && density < 0.015) || density >= 0.15;   // (fused 53 / 46 / 41 % at 0.25 / 0.5 / 1 % against 65 / 59 / 50)
	case family_t::frame_of_reference:  return (density >= 0.004 && density < 0.015) || density >= 0.35;    // (46 against 52 at 1 %; 34 against 40 at 50 %)
	case family_t::model:               return density < 0.004 || density >= 0.35;                          // (64 against 76 at 0.1 %: the plain 2-byte Model is the fastest stream there is)
	case family_t::dictionary:          return density >= 0.35;                                             // (36 against 40 at 50 %)
	default:                            return false;                                                        // (fixed-width discard-zero-bytes: the fused kernel everywhere)
	}
}

/// The two launches on `stream`.  `PatchedWrapper`: the fused wrapper that calls — the plain one is the factory's, under the fused wrapper's key less "patched::"
/// (produced once per fused wrapper type: the plain decoders are not instantiated a second time in the fused wrappers' translation units); `arguments`: the fused
/// wrapper's own map (the plain wrapper reads its keys out of it); `plain_resolution_arguments`: what the plain wrapper's resolve_launch_configuration takes;
/// `fused_config`: the configuration the caller resolved for the fused kernel — a caller's block-size limit is read off it and kept.
template <typename PatchedWrapper, unsigned IndexSize, unsigned ElementSize>
inline void decode_then_scatter(stream::id_t stream, const launch_configuration_t& fused_config, const kernel_t::arguments_type& arguments,
	const kernel_t::arguments_type& plain_resolution_arguments, uint_t<ElementSize>* decompressed, const uint_t<ElementSize>* patch_values,
	const uint_t<IndexSize>* patch_positions, size_type_by_index_size<IndexSize> num_patches)
{
	static const std::unique_ptr<cuda::registered::kernel_t> plain_wrapper = [] {
		std::string key = cuda::registered::detail::factory_key_of(util::type_name<PatchedWrapper>());
		const std::string marker = "patched::";
		const auto at = key.find(marker);
		if (at == std::string::npos) { throw std::logic_error("not a fused wrapper's key: " + key); }
		key.erase(at, marker.length());
		return cuda::registered::kernel_t::produceSubclass(key);
	}();
	const cuda::registered::kernel_t& plain = *plain_wrapper;
	static const device_function::attributes_t plain_attributes = plain.get_device_function().attributes();
	const auto& properties = current_device_properties();
	launch_configuration_limits_t limits { nullopt, nullopt, nullopt };
	const auto fused_block = (grid_block_dimension_t) fused_config.block_dimensions.volume();
	if (fused_block < 256) { limits.block_size = fused_block; }   // (the fused kernels' own workgroups are 256 threads: fewer means the caller set a limit)
	const launch_configuration_t plain_config = plain.resolve_launch_configuration(properties, plain_attributes, plain_resolution_arguments, limits);
	plain.enqueue_launch_with(stream, plain_config, arguments);
	if (num_patches == 0) { return; }
	// scatter_ascending: of equal positions the last one wins, as in the fused kernels (the registered `scatter` wants distinct indices).  Four patches a lane, a
	// workgroup per tile; any grid is right (grid stride).
	const grid_block_dimension_t threads = limits.block_size ? std::max<grid_block_dimension_t>(warp_size, limits.block_size.value() / warp_size * warp_size) : 256;
	const uint64_t blocks = util::div_rounding_up((uint64_t) num_patches, (uint64_t) threads * cuda::kernels::scatter::detail::patches_per_lane);
	const launch_configuration_t scatter_config((grid_dimension_t) std::min<uint64_t>(blocks, 1u << 22), threads);
	cuda::enqueue_launch(reinterpret_cast<const void*>(cuda::kernels::scatter::scatter_ascending<IndexSize, ElementSize>), stream, scatter_config,
		decompressed, patch_values, patch_positions, (uint64_t) num_patches);
}

} // namespace patched
} // namespace decompression
} // namespace kernels
} // namespace cuda

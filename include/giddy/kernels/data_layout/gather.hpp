// Stand-alone gather for CDNA4:   reordered_data[p] = data[indices[p]],  p < num_indices
//
// Kernel template name, template parameters and parameter list as the reference's
// src/kernels/data_layout/gather.cuh:151-164 (launch parameters :166-232), which it keeps without a wrapper
// (SURVEY.md §8f.3); Dictionary decompression IS this kernel with the dictionary as `data` and the compressed
// column as `indices` (dictionary.cuh:61-75 forwards here), so the execution is the Dictionary decoder's
// (include/giddy/kernels/decompression/dictionary.hpp): `data` staged once per workgroup in LDS when the launch
// configuration reserved room for it, G indices per lane as one load, 16-byte stores contiguous across the
// wavefront; tables too large for LDS are looked up through L1/L2.  Unlike the reference — which computes but
// then disables its shared-memory caching (gather.cuh:202-205) — a table that fits LDS is staged by default.
#pragma once
#include "giddy/kernels/decompression/dictionary.hpp"

namespace cuda {
namespace kernels {
namespace gather {

using decompression::dictionary::get_default_serialization_factor;

template <unsigned OutputIndexSize, unsigned ElementSize, unsigned InputIndexSize>
__global__ void gather(
	uint_t<ElementSize>*           __restrict__  reordered_data,
	const uint_t<ElementSize>*     __restrict__  data,
	const uint_t<InputIndexSize>*  __restrict__  indices,
	size_type_by_index_size<InputIndexSize>      data_length,
	size_type_by_index_size<OutputIndexSize>     num_indices,
	bool                                         cache_data_in_shared_memory,
	serialization_factor_t                       serialization_factor) // accepted and ignored
{
	(void) serialization_factor;
	decompression::dictionary::detail::decompress<ElementSize, InputIndexSize>(
		reordered_data, indices, data, num_indices, data_length, cache_data_in_shared_memory, sorted_patches::none());
}

template <unsigned OutputIndexSize, unsigned ElementSize, unsigned InputIndexSize>
class launch_config_resolution_params_t final : public kernels::launch_config_resolution_params_t {
public:
	using parent = kernels::launch_config_resolution_params_t;
	using as_dictionary = decompression::dictionary::launch_config_resolution_params_t<OutputIndexSize, ElementSize, InputIndexSize>;

	launch_config_resolution_params_t(
		device::properties_t            device_properties_,
		size_t                          input_data_length,
		size_t                          num_indices,
		optional<bool>                  cache_input_data_in_shared_memory_override = nullopt,
		optional<shared_memory_size_t>  dynamic_shared_mem_limit = nullopt)
	: parent(device_properties_, device_function_t(gather<OutputIndexSize, ElementSize, InputIndexSize>).attributes(), dynamic_shared_mem_limit)
	{
		// the shape of the Dictionary decoder for a table of this size, for this kernel's own attributes
		as_dictionary shape(device_properties_, kernel_function_attributes, input_data_length, num_indices,
			cache_input_data_in_shared_memory_override, dynamic_shared_mem_limit);
		static_cast<parent&>(*this) = static_cast<const parent&>(shape);
	}
};

} // namespace gather
} // namespace kernels
} // namespace cuda

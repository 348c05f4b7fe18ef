// Kernel wrapper for the expansion of COMPRESSED patch positions into a plain ascending array (an extension;
// SURVEY.md §8f.1) — the position half of the reference's compressed-indices scatter
// (src/kernels/data_layout/compressed_indices_scatter.cuh:219-271), without the scattered writes.  The fused
// decoders (kernel_wrappers/decompression/patched/*.hpp) run it over their "patch_positions_scratch" when they are
// handed compressed positions.  Argument-map keys: those of the compressed-indices scatter wrapper
// (src/kernel_wrappers/data_layout/compressed_indices_scatter.cu:48-104) with "positions" in place of "target" and
// "data_to_scatter"; optionally "unsupported_patch_count" (unsigned*, zeroed by the caller): incremented once per patch
// of a run whose offset width is none of 8/16/32/64 bits — the scatter skips such runs, a plain array cannot say so.
#pragma once
#include "giddy/kernel_wrappers/common.hpp"
#include "giddy/kernels/data_layout/compressed_indices_scatter.hpp"

namespace cuda {
namespace kernels {
namespace scatter {
namespace compressed_indices {
namespace positions {

/// the launch shape of the expansion: a wavefront per anchored block of patches, four to a workgroup
inline launch_configuration_t launch_configuration(size_t num_patches, size_t anchoring_period)
{
	if (anchoring_period == 0) { throw std::invalid_argument("The anchoring period must be a positive number"); }
	const size_t patch_blocks = util::div_rounding_up(num_patches, anchoring_period);
	const size_t workgroups = util::div_rounding_up(patch_blocks, (size_t) 4);
	return launch_configuration_t((grid_dimension_t) (workgroups < (1u << 20) ? workgroups : (1u << 20)), 256);
}

/// Enqueues the expansion (nothing when there are no patches); the arrays as in the kernel's parameter list
template <unsigned OutputIndexSize, unsigned InputIndexSize, unsigned RunLengthSize = InputIndexSize>
inline void enqueue_expansion(stream::id_t stream, uint_t<OutputIndexSize>* positions, const cuda::kernel_t::arguments_type& arguments,
	size_type_by_index_size<InputIndexSize> num_patches)
{
	if (num_patches == 0) { return; }
	using input_index_type = uint_t<InputIndexSize>;
	auto run_lengths      = argument<const uint_t<RunLengthSize>*>(arguments, "scatter_position_run_lengths");
	auto offset_sizes     = argument<const unsigned char*>(arguments, "scatter_position_run_individual_offset_sizes");
	auto baseline_values  = argument<const uint_t<OutputIndexSize>*>(arguments, "scatter_position_run_baseline_values");
	auto offsets_start    = argument<const input_index_type*>(arguments, "scatter_position_run_offsets_start_pos");
	auto offset_bytes     = argument<const unsigned char*>(arguments, "scatter_position_offset_bytes");
	auto anchors          = argument<const input_index_type*>(arguments, "scatter_position_anchors");
	auto anchoring_period = argument<input_index_type>(arguments, "anchoring_period");
	auto num_runs         = argument<input_index_type>(arguments, "num_scatter_position_runs");
	// optional: a device counter (caller-zeroed) of the patches of runs whose offset width is none of 8/16/32/64 bits
	auto unsupported      = optional_argument<native_word_t*>(arguments, "unsupported_patch_count").value_or(nullptr);
	cuda::enqueue_launch(
		reinterpret_cast<const void*>(expand_positions<OutputIndexSize, InputIndexSize, RunLengthSize>), stream,
		launch_configuration(num_patches, anchoring_period),
		positions, run_lengths, offset_sizes, baseline_values, offsets_start, offset_bytes, anchors,
		anchoring_period, num_runs, num_patches, unsupported);
}

template <unsigned OutputIndexSize, unsigned InputIndexSize, unsigned RunLengthSize = InputIndexSize>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);
	using input_size_type = size_type_by_index_size<InputIndexSize>;
};

#define GIDDY_CIP_TEMPLATE template <unsigned OutputIndexSize, unsigned InputIndexSize, unsigned RunLengthSize>
#define GIDDY_CIP_KERNEL kernel_t<OutputIndexSize, InputIndexSize, RunLengthSize>

GIDDY_CIP_TEMPLATE
launch_configuration_t GIDDY_CIP_KERNEL::resolve_launch_configuration(
	device::properties_t, device_function::attributes_t, arguments_type extra_arguments, launch_configuration_limits_t) const
{
	auto input_data_length = argument<size_t>(extra_arguments, "input_data_length");
	auto anchoring_period  = argument<size_t>(extra_arguments, "anchoring_period");
	if (input_data_length == 0) { throw std::invalid_argument("Zero-length scatters not currently supported"); }
	return launch_configuration(input_data_length, anchoring_period);
}

GIDDY_CIP_TEMPLATE
void GIDDY_CIP_KERNEL::enqueue_launch_with(stream::id_t stream, const launch_configuration_t&, const arguments_type& arguments) const
{
	auto positions         = argument<uint_t<OutputIndexSize>*>(arguments, "positions");
	auto input_data_length = argument<input_size_type>(arguments, "input_data_length");
	enqueue_expansion<OutputIndexSize, InputIndexSize, RunLengthSize>(stream, positions, arguments, input_data_length);
}

GIDDY_CIP_TEMPLATE
const device_function_t GIDDY_CIP_KERNEL::get_device_function() const
{
	return reinterpret_cast<const void*>(expand_positions<OutputIndexSize, InputIndexSize, RunLengthSize>);
}

#undef GIDDY_CIP_TEMPLATE
#undef GIDDY_CIP_KERNEL

} // namespace positions
} // namespace compressed_indices
} // namespace scatter
} // namespace kernels
} // namespace cuda

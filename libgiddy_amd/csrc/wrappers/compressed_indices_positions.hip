// Instantiates and registers the expansion of compressed patch positions (an extension beyond the reference's
// registrations): keys "scatter::compressed_indices::positions::kernel_t<OutputIndexSize, InputIndexSize, RunLengthSize>".
#include "giddy/kernel_wrappers/data_layout/compressed_indices_positions.hpp"

namespace cuda {
namespace kernels {
namespace scatter {
namespace compressed_indices {
namespace positions {

template <unsigned OutputIndexSize>
static void register_for_output_index_size()
{
	kernel_t < OutputIndexSize, 1 >::registerInSubclassFactory();
	kernel_t < OutputIndexSize, 2 >::registerInSubclassFactory();
	kernel_t < OutputIndexSize, 4 >::registerInSubclassFactory();
	kernel_t < OutputIndexSize, 8 >::registerInSubclassFactory();
}

static_block {
	register_for_output_index_size<4>();
	register_for_output_index_size<8>();
}

} // namespace positions
} // namespace compressed_indices
} // namespace scatter
} // namespace kernels
} // namespace cuda

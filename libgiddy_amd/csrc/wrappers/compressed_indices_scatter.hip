// Instantiates and registers the compressed-indices scatter wrappers — the same 32 tuples as the
// reference's src/kernel_wrappers/data_layout/compressed_indices_scatter.cu:119-151.
#include "giddy/kernel_wrappers/data_layout/compressed_indices_scatter.hpp"

namespace cuda {
namespace kernels {
namespace scatter {
namespace compressed_indices {

template <unsigned OutputIndexSize, unsigned ElementSize>
static void register_for_input_index_sizes()
{
	kernel_t < OutputIndexSize, ElementSize, 1 >::registerInSubclassFactory();
	kernel_t < OutputIndexSize, ElementSize, 2 >::registerInSubclassFactory();
	kernel_t < OutputIndexSize, ElementSize, 4 >::registerInSubclassFactory();
	kernel_t < OutputIndexSize, ElementSize, 8 >::registerInSubclassFactory();
}

template <unsigned OutputIndexSize>
static void register_for_output_index_size()
{
	register_for_input_index_sizes<OutputIndexSize, 1>();
	register_for_input_index_sizes<OutputIndexSize, 2>();
	register_for_input_index_sizes<OutputIndexSize, 4>();
	register_for_input_index_sizes<OutputIndexSize, 8>();
}

static_block {
	register_for_output_index_size<4>();
	register_for_output_index_size<8>();
}

} // namespace compressed_indices
} // namespace scatter
} // namespace kernels
} // namespace cuda

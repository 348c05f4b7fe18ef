// Instantiates and registers the fixed-width DZB wrappers — the same 12 tuples as the reference's
// src/kernel_wrappers/decompression/discard_zero_bytes_fixed.cu:113-125.
#include "giddy/kernel_wrappers/decompression/discard_zero_bytes_fixed.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace discard_zero_bytes {
namespace fixed_width {

template <unsigned IndexSize>
static void register_for_index_size()
{
	//         IndexSize  UncompressedSize  CompressedSize
	kernel_t < IndexSize, 2,                1 >::registerInSubclassFactory();
	kernel_t < IndexSize, 4,                1 >::registerInSubclassFactory();
	kernel_t < IndexSize, 4,                2 >::registerInSubclassFactory();
	kernel_t < IndexSize, 8,                1 >::registerInSubclassFactory();
	kernel_t < IndexSize, 8,                2 >::registerInSubclassFactory();
	kernel_t < IndexSize, 8,                4 >::registerInSubclassFactory();
}

static_block {
	register_for_index_size<4>();
	register_for_index_size<8>();
}

} // namespace fixed_width
} // namespace discard_zero_bytes
} // namespace decompression
} // namespace kernels
} // namespace cuda

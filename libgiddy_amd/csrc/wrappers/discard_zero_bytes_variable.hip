// Instantiates and registers the variable-width DZB wrappers — the same 9 tuples as the reference's
// src/kernel_wrappers/decompression/discard_zero_bytes_variable.cu:109-119.
#include "giddy/kernel_wrappers/decompression/discard_zero_bytes_variable.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace discard_zero_bytes {
namespace variable_width {

static_block {
	//         IndexSize  Uncompressed   ElementSizesContainerSize
	kernel_t < 4,         2,             4 >::registerInSubclassFactory();
	kernel_t < 4,         4,             4 >::registerInSubclassFactory();
	kernel_t < 4,         8,             4 >::registerInSubclassFactory();
	kernel_t < 8,         2,             4 >::registerInSubclassFactory();
	kernel_t < 8,         4,             4 >::registerInSubclassFactory();
	kernel_t < 8,         8,             4 >::registerInSubclassFactory();
	kernel_t < 4,         2,             8 >::registerInSubclassFactory();
	kernel_t < 4,         4,             8 >::registerInSubclassFactory();
	kernel_t < 4,         8,             8 >::registerInSubclassFactory();
}

} // namespace variable_width
} // namespace discard_zero_bytes
} // namespace decompression
} // namespace kernels
} // namespace cuda

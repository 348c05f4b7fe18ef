// Instantiates and registers the Delta wrappers — the same 12 tuples as the reference's
// src/kernel_wrappers/decompression/delta.cu:94-106.
#include "giddy/kernel_wrappers/decompression/delta.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace delta {

template <unsigned IndexSize>
static void register_for_index_size()
{
	//         IndexSize   UncompressedSize   CompressedSize
	kernel_t < IndexSize,  2,                 1 >::registerInSubclassFactory();
	kernel_t < IndexSize,  4,                 1 >::registerInSubclassFactory();
	kernel_t < IndexSize,  4,                 2 >::registerInSubclassFactory();
	kernel_t < IndexSize,  8,                 1 >::registerInSubclassFactory();
	kernel_t < IndexSize,  8,                 2 >::registerInSubclassFactory();
	kernel_t < IndexSize,  8,                 4 >::registerInSubclassFactory();
}

static_block {
	register_for_index_size<4>();
	register_for_index_size<8>();
}

} // namespace delta
} // namespace decompression
} // namespace kernels
} // namespace cuda

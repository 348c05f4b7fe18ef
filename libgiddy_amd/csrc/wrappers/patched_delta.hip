// Instantiates and registers the fused Delta + patches wrappers (an extension beyond the reference's 163
// registrations): the tuples of the plain Delta wrappers, keys "decompression::delta::patched::kernel_t<...>".
#include "giddy/kernel_wrappers/decompression/patched/delta.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace delta {
namespace patched {

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

} // namespace patched
} // namespace delta
} // namespace decompression
} // namespace kernels
} // namespace cuda

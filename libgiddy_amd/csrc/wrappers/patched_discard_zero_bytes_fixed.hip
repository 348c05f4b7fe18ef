// Instantiates and registers the fused DZB-fixed + patches wrappers (an extension beyond the reference's 163
// registrations): the tuples of the plain wrappers, keys "decompression::discard_zero_bytes::fixed_width::patched::kernel_t<...>".
#include "giddy/kernel_wrappers/decompression/patched/discard_zero_bytes_fixed.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace discard_zero_bytes {
namespace fixed_width {
namespace patched {

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

} // namespace patched
} // namespace fixed_width
} // namespace discard_zero_bytes
} // namespace decompression
} // namespace kernels
} // namespace cuda

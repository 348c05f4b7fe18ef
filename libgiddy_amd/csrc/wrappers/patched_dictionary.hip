// Instantiates and registers the fused Dictionary + patches wrappers (an extension beyond the reference's 163
// registrations): the tuples of the plain Dictionary wrappers, keys "decompression::dictionary::patched::kernel_t<...>".
#include "giddy/kernel_wrappers/decompression/patched/dictionary.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace dictionary {
namespace patched {

template <unsigned IndexSize>
static void register_for_index_size()
{
	//         IndexSize   UncompressedSize  DictionaryIndexSize
	kernel_t < IndexSize,  1,                1 >::registerInSubclassFactory();
	kernel_t < IndexSize,  2,                1 >::registerInSubclassFactory();
	kernel_t < IndexSize,  2,                2 >::registerInSubclassFactory();
	kernel_t < IndexSize,  4,                1 >::registerInSubclassFactory();
	kernel_t < IndexSize,  4,                2 >::registerInSubclassFactory();
	kernel_t < IndexSize,  4,                4 >::registerInSubclassFactory();
	kernel_t < IndexSize,  8,                1 >::registerInSubclassFactory();
	kernel_t < IndexSize,  8,                2 >::registerInSubclassFactory();
	kernel_t < IndexSize,  8,                4 >::registerInSubclassFactory();
	kernel_t < IndexSize,  8,                8 >::registerInSubclassFactory();
}

static_block {
	register_for_index_size<4>();
	register_for_index_size<8>();
}

} // namespace patched
} // namespace dictionary
} // namespace decompression
} // namespace kernels
} // namespace cuda

// Instantiates and registers the Dictionary wrappers — the same 20 tuples as the reference's
// src/kernel_wrappers/decompression/dictionary.cu:113-139.
#include "giddy/kernel_wrappers/decompression/dictionary.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace dictionary {

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

} // namespace dictionary
} // namespace decompression
} // namespace kernels
} // namespace cuda

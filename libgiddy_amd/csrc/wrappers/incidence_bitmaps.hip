// Instantiates and registers the incidence-bitmaps wrappers — the same 5 tuples as the reference's
// src/kernel_wrappers/decompression/incidence_bitmaps.cu:96-100.
#include "giddy/kernel_wrappers/decompression/incidence_bitmaps.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace incidence_bitmaps {

static_block {
	//         IndexSize  UncompressedSize  BitmapAlignmentInInts  UseDictionary
	kernel_t < 4,         1,                2,                     true  >::registerInSubclassFactory();
	kernel_t < 4,         2,                2,                     true  >::registerInSubclassFactory();
	kernel_t < 4,         4,                2,                     true  >::registerInSubclassFactory();
	kernel_t < 4,         8,                2,                     true  >::registerInSubclassFactory();
	kernel_t < 4,         1,                2,                     false >::registerInSubclassFactory();
}

} // namespace incidence_bitmaps
} // namespace decompression
} // namespace kernels
} // namespace cuda

// Instantiates and registers the RLE wrappers — the same 18 tuples as the reference's
// src/kernel_wrappers/decompression/run_length_encoding.cu:107-128.
#include "giddy/kernel_wrappers/decompression/run_length_encoding.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace run_length_encoding {

static_block {
	//         UncompressedIndexSize   UncompressedSize  RunLengthSize
	kernel_t < 4,                      1,                1 >::registerInSubclassFactory();
	kernel_t < 4,                      2,                1 >::registerInSubclassFactory();
	kernel_t < 4,                      4,                1 >::registerInSubclassFactory();
	kernel_t < 4,                      8,                1 >::registerInSubclassFactory();

	kernel_t < 4,                      1,                2 >::registerInSubclassFactory();
	kernel_t < 4,                      2,                2 >::registerInSubclassFactory();
	kernel_t < 4,                      4,                2 >::registerInSubclassFactory();
	kernel_t < 4,                      8,                2 >::registerInSubclassFactory();

	kernel_t < 4,                      1,                4 >::registerInSubclassFactory();
	kernel_t < 4,                      2,                4 >::registerInSubclassFactory();
	kernel_t < 4,                      4,                4 >::registerInSubclassFactory();
	kernel_t < 4,                      8,                4 >::registerInSubclassFactory();

	kernel_t < 8,                      4,                1 >::registerInSubclassFactory();
	kernel_t < 8,                      4,                2 >::registerInSubclassFactory();
	kernel_t < 8,                      4,                4 >::registerInSubclassFactory();

	kernel_t < 8,                      1,                8 >::registerInSubclassFactory();
	kernel_t < 8,                      4,                8 >::registerInSubclassFactory();
	kernel_t < 8,                      8,                8 >::registerInSubclassFactory();
}

} // namespace run_length_encoding
} // namespace decompression
} // namespace kernels
} // namespace cuda

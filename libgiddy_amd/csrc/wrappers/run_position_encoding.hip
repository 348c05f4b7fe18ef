// Instantiates and registers the RPE wrappers — the same 10 tuples as the reference's
// src/kernel_wrappers/decompression/run_position_encoding.cu:124-136.
#include "giddy/kernel_wrappers/decompression/run_position_encoding.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace run_position_encoding {

static_block {
	constexpr bool relative = true;
	constexpr bool absolute = false;

	//         UncompressedIndexSize   Datum   RunPosition  PositionsAreRelative
	kernel_t < 4,                      1,      2,           relative >::registerInSubclassFactory();
	kernel_t < 4,                      2,      2,           relative >::registerInSubclassFactory();
	kernel_t < 4,                      4,      2,           relative >::registerInSubclassFactory();
	kernel_t < 4,                      8,      2,           relative >::registerInSubclassFactory();
	kernel_t < 8,                      4,      2,           relative >::registerInSubclassFactory();

	kernel_t < 4,                      1,      4,           absolute >::registerInSubclassFactory();
	kernel_t < 4,                      2,      4,           absolute >::registerInSubclassFactory();
	kernel_t < 4,                      4,      4,           absolute >::registerInSubclassFactory();
	kernel_t < 4,                      8,      4,           absolute >::registerInSubclassFactory();
	kernel_t < 8,                      4,      8,           absolute >::registerInSubclassFactory();
}

} // namespace run_position_encoding
} // namespace decompression
} // namespace kernels
} // namespace cuda

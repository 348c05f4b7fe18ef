// Instantiates and registers the device-side compressors (an extension: the reference has none) for the same type tuples the
// decompressors are registered with.
#include "giddy/kernel_wrappers/compression/compressors.hpp"

namespace cuda {
namespace kernels {
namespace compression {

template <unsigned IndexSize>
static void register_for_index_size()
{
	constexpr unsigned I = IndexSize;
	// Delta, fixed-width discard-zero-bytes: <IndexSize, UncompressedSize, CompressedSize>
	delta::kernel_t<I, 2, 1>::registerInSubclassFactory(); delta::kernel_t<I, 4, 1>::registerInSubclassFactory(); delta::kernel_t<I, 4, 2>::registerInSubclassFactory();
	delta::kernel_t<I, 8, 1>::registerInSubclassFactory(); delta::kernel_t<I, 8, 2>::registerInSubclassFactory(); delta::kernel_t<I, 8, 4>::registerInSubclassFactory();
	discard_zero_bytes::fixed_width::kernel_t<I, 2, 1>::registerInSubclassFactory(); discard_zero_bytes::fixed_width::kernel_t<I, 4, 1>::registerInSubclassFactory();
	discard_zero_bytes::fixed_width::kernel_t<I, 4, 2>::registerInSubclassFactory(); discard_zero_bytes::fixed_width::kernel_t<I, 8, 1>::registerInSubclassFactory();
	discard_zero_bytes::fixed_width::kernel_t<I, 8, 2>::registerInSubclassFactory(); discard_zero_bytes::fixed_width::kernel_t<I, 8, 4>::registerInSubclassFactory();
	// variable-width discard-zero-bytes: <IndexSize, UncompressedSize, ElementSizesContainerSize>
	discard_zero_bytes::variable_width::kernel_t<I, 2, 4>::registerInSubclassFactory(); discard_zero_bytes::variable_width::kernel_t<I, 4, 4>::registerInSubclassFactory();
	discard_zero_bytes::variable_width::kernel_t<I, 8, 4>::registerInSubclassFactory();
	// dictionary encoding: <IndexSize, UncompressedSize, DictionaryIndexSize>
	dictionary::kernel_t<I, 1, 1>::registerInSubclassFactory(); dictionary::kernel_t<I, 2, 1>::registerInSubclassFactory(); dictionary::kernel_t<I, 2, 2>::registerInSubclassFactory();
	dictionary::kernel_t<I, 4, 1>::registerInSubclassFactory(); dictionary::kernel_t<I, 4, 2>::registerInSubclassFactory(); dictionary::kernel_t<I, 4, 4>::registerInSubclassFactory();
	dictionary::kernel_t<I, 8, 1>::registerInSubclassFactory(); dictionary::kernel_t<I, 8, 2>::registerInSubclassFactory(); dictionary::kernel_t<I, 8, 4>::registerInSubclassFactory(); dictionary::kernel_t<I, 8, 8>::registerInSubclassFactory();
	// dictionary construction (distinct values, sorted; up to 65 536 of them): <IndexSize, UncompressedSize>
	dictionary::construction::kernel_t<I, 1>::registerInSubclassFactory(); dictionary::construction::kernel_t<I, 2>::registerInSubclassFactory();
	dictionary::construction::kernel_t<I, 4>::registerInSubclassFactory(); dictionary::construction::kernel_t<I, 8>::registerInSubclassFactory();
	// patches: <IndexSize, ElementSize>
	patches::kernel_t<I, 1>::registerInSubclassFactory(); patches::kernel_t<I, 2>::registerInSubclassFactory();
	patches::kernel_t<I, 4>::registerInSubclassFactory(); patches::kernel_t<I, 8>::registerInSubclassFactory();
}

static_block {
	register_for_index_size<4>();
	register_for_index_size<8>();
	discard_zero_bytes::variable_width::kernel_t<4, 2, 8>::registerInSubclassFactory();
	discard_zero_bytes::variable_width::kernel_t<4, 4, 8>::registerInSubclassFactory();
	discard_zero_bytes::variable_width::kernel_t<4, 8, 8>::registerInSubclassFactory();
	// Frame of Reference with a constant model: <IndexSize, Uncompressed, Compressed> (the decompressor's constant-model tuples + the 8-byte index)
	frame_of_reference::kernel_t<4, short, signed char>::registerInSubclassFactory();
	frame_of_reference::kernel_t<4, int, signed char>::registerInSubclassFactory();
	frame_of_reference::kernel_t<4, int, short>::registerInSubclassFactory();
	frame_of_reference::kernel_t<8, int, short>::registerInSubclassFactory();
	// ... and with a fitted line per segment (the decompressor's linear-model tuples)
	frame_of_reference::linear::kernel_t<4, short, signed char>::registerInSubclassFactory();
	frame_of_reference::linear::kernel_t<4, int, signed char>::registerInSubclassFactory();
	frame_of_reference::linear::kernel_t<4, int, short>::registerInSubclassFactory();
	frame_of_reference::linear::kernel_t<8, int, short>::registerInSubclassFactory();
	incidence_bitmaps::kernel_t<4, 2>::registerInSubclassFactory();   // <IndexSize, BitmapAlignmentInInts>
	// compressed patch positions: <OutputIndexSize, InputIndexSize, RunLengthSize> (what the fused decoders and the position expansion take)
	compressed_indices::kernel_t<4, 4>::registerInSubclassFactory(); compressed_indices::kernel_t<4, 8>::registerInSubclassFactory();
	compressed_indices::kernel_t<8, 4>::registerInSubclassFactory(); compressed_indices::kernel_t<8, 8>::registerInSubclassFactory();
	compressed_indices::kernel_t<4, 2>::registerInSubclassFactory(); compressed_indices::kernel_t<4, 1>::registerInSubclassFactory();
	// run-length: <IndexSize, UncompressedSize, RunLengthSize, RunIndexSize>, the decompressor's 18 tuples
	run_length_encoding::kernel_t<4, 1, 1>::registerInSubclassFactory(); run_length_encoding::kernel_t<4, 1, 2>::registerInSubclassFactory(); run_length_encoding::kernel_t<4, 1, 4>::registerInSubclassFactory();
	run_length_encoding::kernel_t<4, 2, 1>::registerInSubclassFactory(); run_length_encoding::kernel_t<4, 2, 2>::registerInSubclassFactory(); run_length_encoding::kernel_t<4, 2, 4>::registerInSubclassFactory();
	run_length_encoding::kernel_t<4, 4, 1>::registerInSubclassFactory(); run_length_encoding::kernel_t<4, 4, 2>::registerInSubclassFactory(); run_length_encoding::kernel_t<4, 4, 4>::registerInSubclassFactory();
	run_length_encoding::kernel_t<4, 8, 1>::registerInSubclassFactory(); run_length_encoding::kernel_t<4, 8, 2>::registerInSubclassFactory(); run_length_encoding::kernel_t<4, 8, 4>::registerInSubclassFactory();
	run_length_encoding::kernel_t<8, 4, 1>::registerInSubclassFactory(); run_length_encoding::kernel_t<8, 4, 2>::registerInSubclassFactory(); run_length_encoding::kernel_t<8, 4, 4>::registerInSubclassFactory();
	run_length_encoding::kernel_t<8, 1, 8>::registerInSubclassFactory(); run_length_encoding::kernel_t<8, 4, 8>::registerInSubclassFactory(); run_length_encoding::kernel_t<8, 8, 8>::registerInSubclassFactory();
	// run-position: <IndexSize, UncompressedSize, PositionOffsetSize, PositionsAreRelative, RunIndexSize>, the decompressor's 10 tuples
	run_position_encoding::kernel_t<4, 1, 2, true>::registerInSubclassFactory(); run_position_encoding::kernel_t<4, 2, 2, true>::registerInSubclassFactory();
	run_position_encoding::kernel_t<4, 4, 2, true>::registerInSubclassFactory(); run_position_encoding::kernel_t<4, 8, 2, true>::registerInSubclassFactory();
	run_position_encoding::kernel_t<8, 4, 2, true>::registerInSubclassFactory();
	run_position_encoding::kernel_t<4, 1, 4, false>::registerInSubclassFactory(); run_position_encoding::kernel_t<4, 2, 4, false>::registerInSubclassFactory();
	run_position_encoding::kernel_t<4, 4, 4, false>::registerInSubclassFactory(); run_position_encoding::kernel_t<4, 8, 4, false>::registerInSubclassFactory();
	run_position_encoding::kernel_t<8, 4, 8, false>::registerInSubclassFactory();
}

} // namespace compression
} // namespace kernels
} // namespace cuda

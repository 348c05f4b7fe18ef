// The fixed-rate compressors: Delta, fixed-width discard-zero-bytes, Frame of Reference with a constant model per
// segment, dictionary ENCODING against a given sorted dictionary.  One pass each; what cannot be represented in the
// requested width is counted in `*unrepresentable` (caller-zeroed, may be null) and stored truncated.
// Formats: libgiddy_amd/formats.py (delta, discard_zero_bytes_fixed, frame_of_reference(model="constant"), dictionary).
#pragma once
#include "giddy/array.hpp"
#include "giddy/kernels/compression/common.hpp"

namespace cuda {
namespace kernels {
namespace compression {

namespace delta {

/// compressed[i] = input[i] - input[i - 1] (wrapping; compressed[0] = 0), anchor_values[s] = input[s * L - 1] (anchor_values[0] =
/// input[0]): what decompression::delta::decompress turns back into `input` (delta.cuh:5-19: the anchor is the value BEFORE the segment)
template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize>
__global__ void __launch_bounds__(threads_per_block) compress(
	uint_t<CompressedSize>*          __restrict__  compressed_output,
	uint_t<UncompressedSize>*        __restrict__  anchor_values,
	const uint_t<UncompressedSize>*  __restrict__  input,
	size_type_by_index_size<IndexSize>             length,
	size_type_by_index_size<IndexSize>             segment_length,
	native_word_t*                   __restrict__  unrepresentable)
{
	using U = uint_t<UncompressedSize>;
	using C = uint_t<CompressedSize>;
	const uint64_t n = length, L = segment_length;
	unsigned bad = 0;
	for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
		const U previous = input[i == 0 ? 0 : i - 1];
		const U difference = (U) (input[i] - previous);
		compressed_output[i] = (C) difference;
		if ((U) (C) difference != difference) { bad++; }
		if (i % L == 0) { anchor_values[i / L] = previous; }
	}
	if (unrepresentable != nullptr && bad != 0) { atomicAdd(unrepresentable, bad); }
}

} // namespace delta

namespace discard_zero_bytes {
namespace fixed_width {

/// compressed[i] = the low CompressedSize bytes of input[i]
template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize>
__global__ void __launch_bounds__(threads_per_block) compress(
	uint_t<CompressedSize>*          __restrict__  compressed_output,
	const uint_t<UncompressedSize>*  __restrict__  input,
	size_type_by_index_size<IndexSize>             length,
	native_word_t*                   __restrict__  unrepresentable)
{
	using U = uint_t<UncompressedSize>;
	using C = uint_t<CompressedSize>;
	unsigned bad = 0;
	for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < length; i += (uint64_t) gridDim.x * blockDim.x) {
		const U value = input[i];
		compressed_output[i] = (C) value;
		if ((U) (C) value != value) { bad++; }
	}
	if (unrepresentable != nullptr && bad != 0) { atomicAdd(unrepresentable, bad); }
}

} // namespace fixed_width
} // namespace discard_zero_bytes

namespace frame_of_reference {

/// One workgroup per segment (segments at grid stride): the segment's reference value = the midpoint of its minimum and maximum
/// (floor of the mean of the two, as formats.frame_of_reference(model="constant") has it), then the signed offsets from it.
/// Uncompressed, Compressed: signed types, as in the decompressor's registrations (frame_of_reference.cu:132-142); coefficients int32.
template <unsigned IndexSize, typename Uncompressed, typename Compressed>
__global__ void __launch_bounds__(threads_per_block) compress_with_constant_model(
	Compressed*          __restrict__  compressed_output,
	cuda::array<int, 1>* __restrict__  segment_model_coefficients,
	const Uncompressed*  __restrict__  input,
	size_type_by_index_size<IndexSize> length,
	size_type_by_index_size<IndexSize> segment_length,
	native_word_t*       __restrict__  unrepresentable)
{
	__shared__ int64_t low[waves_per_block], high[waves_per_block];
	const uint64_t n = length, L = segment_length, num_segments = (n + L - 1) / L;
	unsigned bad = 0;
	for (uint64_t s = blockIdx.x; s < num_segments; s += gridDim.x) {
		const uint64_t start = s * L, end = start + L < n ? start + L : n;
		int64_t lowest = INT64_MAX, highest = INT64_MIN;
		for (uint64_t i = start + threadIdx.x; i < end; i += blockDim.x) {
			const int64_t v = (int64_t) input[i];
			lowest = v < lowest ? v : lowest; highest = v > highest ? v : highest;
		}
		#pragma unroll
		for (int offset = 32; offset >= 1; offset >>= 1) {
			const int64_t other_low = __shfl_xor(lowest, offset, 64), other_high = __shfl_xor(highest, offset, 64);
			lowest = other_low < lowest ? other_low : lowest; highest = other_high > highest ? other_high : highest;
		}
		__syncthreads();
		if (wave64::lane_id() == 0) { low[wave64::index_in_block()] = lowest; high[wave64::index_in_block()] = highest; }
		__syncthreads();
		for (unsigned w = 0; w < wave64::per_block() && w < waves_per_block; w++) {
			lowest = low[w] < lowest ? low[w] : lowest; highest = high[w] > highest ? high[w] : highest;
		}
		const int reference = (int) ((lowest + highest) >> 1);   // floor((min + max) / 2), then narrowed to int32 as the writer does
		if (threadIdx.x == 0) { segment_model_coefficients[s][0] = reference; }
		for (uint64_t i = start + threadIdx.x; i < end; i += blockDim.x) {
			const int64_t offset = (int64_t) input[i] - (int64_t) reference;
			compressed_output[i] = (Compressed) offset;
			if ((int64_t) (Compressed) offset != offset) { bad++; }
		}
	}
	if (unrepresentable != nullptr && bad != 0) { atomicAdd(unrepresentable, bad); }
}

/// The same with a LINE per segment: c1 = the least-squares slope over x = 0 .. n-1 rounded to the nearest integer (ties to even) in
/// exact integer arithmetic — (n Sxy - Sx Sy) / (n Sxx - Sx^2), 128-bit —, c0 = the midpoint of the residuals' extremes, offsets from
/// c0 + c1 x: what formats.frame_of_reference(model="linear") writes.  One workgroup per segment, three passes over it.
template <unsigned IndexSize, typename Uncompressed, typename Compressed>
__global__ void __launch_bounds__(threads_per_block) compress_with_linear_model(
	Compressed*          __restrict__  compressed_output,
	cuda::array<int, 2>* __restrict__  segment_model_coefficients,
	const Uncompressed*  __restrict__  input,
	size_type_by_index_size<IndexSize> length,
	size_type_by_index_size<IndexSize> segment_length,
	native_word_t*       __restrict__  unrepresentable)
{
	__shared__ __int128 partial_xy[threads_per_block];
	__shared__ int64_t partial_y[threads_per_block];
	__shared__ int64_t low[waves_per_block], high[waves_per_block];
	__shared__ int64_t slope_shared;
	const uint64_t n = length, L = segment_length, num_segments = (n + L - 1) / L;
	unsigned bad = 0;
	for (uint64_t s = blockIdx.x; s < num_segments; s += gridDim.x) {
		const uint64_t start = s * L, end = start + L < n ? start + L : n, count = end - start;
		// 1. Sy and Sxy, exactly
		__int128 sum_xy = 0; int64_t sum_y = 0;
		for (uint64_t x = threadIdx.x; x < count; x += blockDim.x) {
			const int64_t y = (int64_t) input[start + x];
			sum_y += y; sum_xy += (__int128) y * (int64_t) x;
		}
		__syncthreads();
		partial_xy[threadIdx.x] = sum_xy; partial_y[threadIdx.x] = sum_y;
		__syncthreads();
		if (threadIdx.x == 0) {
			__int128 sxy = 0, sy = 0;
			for (unsigned t = 0; t < blockDim.x && t < threads_per_block; t++) { sxy += partial_xy[t]; sy += partial_y[t]; }
			int64_t slope = 0;
			if (count >= 2) {
				const __int128 c = (__int128) count, sx = c * (c - 1) / 2, sxx = (c - 1) * c * (2 * c - 1) / 6;
				const __int128 numerator = c * sxy - sx * sy, denominator = c * sxx - sx * sx;   // denominator > 0
				__int128 quotient = numerator / denominator, remainder = numerator % denominator;
				if (remainder < 0) { quotient -= 1; remainder += denominator; }                    // floor division
				if (2 * remainder > denominator || (2 * remainder == denominator && (quotient & 1) != 0)) { quotient += 1; }
				slope = (int64_t) quotient;
			}
			slope_shared = slope;
		}
		__syncthreads();
		const int64_t slope = slope_shared;
		// 2. the residuals' extremes
		int64_t lowest = INT64_MAX, highest = INT64_MIN;
		for (uint64_t x = threadIdx.x; x < count; x += blockDim.x) {
			const int64_t residual = (int64_t) input[start + x] - slope * (int64_t) x;
			lowest = residual < lowest ? residual : lowest; highest = residual > highest ? residual : highest;
		}
		#pragma unroll
		for (int offset = 32; offset >= 1; offset >>= 1) {
			const int64_t other_low = __shfl_xor(lowest, offset, 64), other_high = __shfl_xor(highest, offset, 64);
			lowest = other_low < lowest ? other_low : lowest; highest = other_high > highest ? other_high : highest;
		}
		if (wave64::lane_id() == 0) { low[wave64::index_in_block()] = lowest; high[wave64::index_in_block()] = highest; }
		__syncthreads();
		for (unsigned w = 0; w < wave64::per_block() && w < waves_per_block; w++) {
			lowest = low[w] < lowest ? low[w] : lowest; highest = high[w] > highest ? high[w] : highest;
		}
		const int c0 = (int) ((lowest + highest) >> 1), c1 = (int) slope;   // (narrowed to int32 as the writer does)
		if (threadIdx.x == 0) { segment_model_coefficients[s][0] = c0; segment_model_coefficients[s][1] = c1; }
		// 3. offsets from the line
		for (uint64_t x = threadIdx.x; x < count; x += blockDim.x) {
			const int64_t offset = (int64_t) input[start + x] - ((int64_t) c0 + (int64_t) c1 * (int64_t) x);
			compressed_output[start + x] = (Compressed) offset;
			if ((int64_t) (Compressed) offset != offset) { bad++; }
		}
		__syncthreads();
	}
	if (unrepresentable != nullptr && bad != 0) { atomicAdd(unrepresentable, bad); }
}

} // namespace frame_of_reference

namespace dictionary {

/// compressed[i] = the index of input[i] in `dictionary_entries`, which is SORTED ascending and distinct (what
/// formats.dictionary builds with numpy.unique): a binary search per element.  A value that is not in the dictionary is counted
/// in `*unrepresentable` and gets the index of the first entry not below it.
template <unsigned IndexSize, unsigned UncompressedSize, unsigned DictionaryIndexSize>
__global__ void __launch_bounds__(threads_per_block) encode(
	uint_t<DictionaryIndexSize>*     __restrict__  compressed_output,
	const uint_t<UncompressedSize>*  __restrict__  input,
	const uint_t<UncompressedSize>*  __restrict__  dictionary_entries,
	size_type_by_index_size<IndexSize>             length,
	size_type_by_index_size<DictionaryIndexSize>   num_dictionary_entries,
	native_word_t*                   __restrict__  unrepresentable)
{
	using U = uint_t<UncompressedSize>;
	using count_type = typename std::conditional<(DictionaryIndexSize > 4), uint64_t, uint32_t>::type;
	enum : unsigned { staged_capacity = 16384 / sizeof(U) };   // dictionaries of up to 16 KiB are searched in LDS
	__shared__ U staged[staged_capacity];
	const count_type entries = (count_type) num_dictionary_entries;
	const bool in_lds = entries <= staged_capacity;   // (workgroup-uniform)
	if (in_lds) {
		for (unsigned i = threadIdx.x; i < entries; i += blockDim.x) { staged[i] = dictionary_entries[i]; }
		__syncthreads();
	}
	// the number of entries below `value`: every thread takes the same steps (no divergence), one look-up each; the two tables are
	// searched by separate code so that the LDS one is read with LDS loads (through one generic pointer both would be flat loads)
	auto entries_below = [entries](const auto* table, U value) {
		count_type lo = 0, left = entries;
		while (left > 1) {
			const count_type half = left / 2;
			lo += table[lo + half - 1] < value ? half : 0;
			left -= half;
		}
		if (left == 1 && table[lo] < value) { lo++; }
		return lo;
	};
	unsigned bad = 0;
	auto encode_one = [&](U value) {
		count_type lo; bool found;
		if (in_lds) { lo = entries_below(staged, value); found = lo < entries && staged[lo] == value; }
		else { lo = entries_below(dictionary_entries, value); found = lo < entries && dictionary_entries[lo] == value; }
		if (!found) { bad++; }
		return (uint_t<DictionaryIndexSize>) lo;
	};
	// G = 16 / sizeof(U) elements per thread per step: one 16-byte load, one store of G keys (element-sized stores of 1-byte keys are 64 bytes per
	// wavefront instruction)
	constexpr unsigned G = 16 / sizeof(U);
	const uint64_t n = length, global_thread = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x, num_threads = (uint64_t) gridDim.x * blockDim.x;
	const uint64_t num_groups = (memory_access::is_aligned(input, 16) && memory_access::is_aligned(compressed_output, G * DictionaryIndexSize)) ? n / G : 0;
	for (uint64_t g = global_thread; g < num_groups; g += num_threads) {
		const memory_access::group<U, G> values = memory_access::load_group<U, G>(input + g * G);
		memory_access::group<uint_t<DictionaryIndexSize>, G> keys;
		#pragma unroll
		for (unsigned j = 0; j < G; j++) { keys[j] = encode_one(values[j]); }
		memory_access::store_group<memory_access::plain_store>(compressed_output + g * G, keys);
	}
	for (uint64_t i = num_groups * G + global_thread; i < n; i += num_threads) { compressed_output[i] = encode_one(input[i]); }
	if (unrepresentable != nullptr && bad != 0) { atomicAdd(unrepresentable, bad); }
}

} // namespace dictionary

} // namespace compression
} // namespace kernels
} // namespace cuda

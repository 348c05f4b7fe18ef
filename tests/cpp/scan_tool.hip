// Driver for the un-anchored scan kernels (include/giddy/kernels/reduction/scan.hpp), used by
// tests/test_gpu_scan.py:   scan_tool <instantiation> <mode> <n> <segment_length> <grid> <offset> <in.bin> <out.bin>
//   mode: intervals  -> model::decompress_with_intervals; n outputs, segment_length = modeling period, in.bin = coefficient records
//         gather     -> data_layout gather; n = number of indices, segment_length = table length
//         two_phase  -> reduce_segments, then scan_using_segment_reductions; out.bin = results, out.bin.segments = phase-1 output
//         single     -> scan_single_segment
//   grid: 0 = as resolved by the launch parameters, otherwise imposed (any grid must be correct)
//   offset: elements by which input and output are displaced from their 256-byte-aligned allocations
// Prints "OK <milliseconds of the last of 5 timed repetitions>"; the numbers are checked in Python against numpy.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "cuda/api_wrappers.h"
#include "cuda/functors.hpp"
#include "kernels/data_layout/gather.cuh"
#include "kernels/decompression/model.cuh"
#include "kernels/reduction/scan.cuh"
#include "kernels/resolve_launch_configuration.h"

namespace scan = cuda::kernels::reduction::scan;
namespace f = cuda::functors;

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); std::exit(2); } } while (0)

template <typename T>
static std::vector<T> read_file(const std::string& path, size_t n)
{
	std::vector<T> v(n);
	FILE* fp = std::fopen(path.c_str(), "rb");
	if (!fp || std::fread(v.data(), sizeof(T), n, fp) != n) { std::fprintf(stderr, "cannot read %zu elements from %s\n", n, path.c_str()); std::exit(2); }
	std::fclose(fp);
	return v;
}

template <typename T>
static void write_file(const std::string& path, const std::vector<T>& v)
{
	FILE* fp = std::fopen(path.c_str(), "wb");
	if (!fp || std::fwrite(v.data(), sizeof(T), v.size(), fp) != v.size()) { std::fprintf(stderr, "cannot write %s\n", path.c_str()); std::exit(2); }
	std::fclose(fp);
}

template <unsigned IndexSize, typename Op, typename In, bool Inclusive, typename Pre>
static int run(const std::string& mode, size_t n, size_t segment_length, unsigned grid, size_t offset, const std::string& in_path, const std::string& out_path)
{
	using result_type = typename Op::result_type;
	using size_type = cuda::size_type_by_index_size<IndexSize>;
	auto host_in = read_file<In>(in_path, n);
	In* d_in_base; result_type* d_out_base; result_type* d_segments;
	HIP_OK(hipMalloc(&d_in_base, (n + offset + 64) * sizeof(In)));
	HIP_OK(hipMalloc(&d_out_base, (n + offset + 64) * sizeof(result_type)));
	In* d_in = d_in_base + offset; result_type* d_out = d_out_base + offset;
	if (n) { HIP_OK(hipMemcpy(d_in, host_in.data(), n * sizeof(In), hipMemcpyHostToDevice)); }
	HIP_OK(hipMemset(d_out_base, 0xCD, (n + offset + 64) * sizeof(result_type)));
	auto properties = cuda::device::current::get().properties();
	hipEvent_t begin, end; HIP_OK(hipEventCreate(&begin)); HIP_OK(hipEventCreate(&end));
	float ms = 0;
	std::vector<result_type> host_segments;

	if (mode == "two_phase") {
		typename scan::reduce_segments::launch_config_resolution_params_t<IndexSize, Op, In, Pre> reduce_params(properties, n, segment_length);
		auto reduce_config = cuda::kernels::resolve_launch_configuration(reduce_params);
		typename scan::scan_using_segment_reductions::launch_config_resolution_params_t<IndexSize, Op, In, Inclusive, Pre> scan_params(properties, n, segment_length);
		auto scan_config = cuda::kernels::resolve_launch_configuration(scan_params);
		if (grid) { reduce_config.grid_dimensions.x = grid; scan_config.grid_dimensions.x = grid; }
		size_t num_segments = (n + segment_length - 1) / segment_length;
		size_t entries = std::max<size_t>(num_segments, reduce_config.grid_dimensions.x) + 1;
		HIP_OK(hipMalloc(&d_segments, entries * sizeof(result_type)));
		for (int repetition = 0; repetition < 5; repetition++) {
			HIP_OK(hipEventRecord(begin, 0));
			if (reduce_config.grid_dimensions.x > 0) {
				cuda::launch(scan::reduce_segments::reduce_segments<IndexSize, Op, In, Pre>, reduce_config,
					d_segments, (const In*) d_in, (size_type) n, (size_type) segment_length);
			}
			if (scan_config.grid_dimensions.x > 0) {
				cuda::launch(scan::scan_using_segment_reductions::scan_using_segment_reductions<IndexSize, Op, In, Inclusive, Pre>, scan_config,
					d_out, (const result_type*) d_segments, (const In*) d_in, (size_type) n, (size_type) segment_length);
			}
			HIP_OK(hipEventRecord(end, 0));
			HIP_OK(hipEventSynchronize(end));
			HIP_OK(hipEventElapsedTime(&ms, begin, end));
		}
		host_segments.resize(num_segments);
		if (num_segments) { HIP_OK(hipMemcpy(host_segments.data(), d_segments, num_segments * sizeof(result_type), hipMemcpyDeviceToHost)); }
		write_file(out_path + ".segments", host_segments);
	}
	else {
		typename scan::scan_single_segment::launch_config_resolution_params_t<IndexSize, Op, In, Inclusive, Pre> params(properties, n);
		auto config = cuda::kernels::resolve_launch_configuration(params);
		if (grid) { config.grid_dimensions.x = grid; }   // surplus workgroups must do nothing
		for (int repetition = 0; repetition < 5; repetition++) {
			HIP_OK(hipEventRecord(begin, 0));
			cuda::launch(scan::scan_single_segment::scan_single_segment<IndexSize, Op, In, Inclusive, Pre>, config,
				d_out, (const In*) d_in, (size_type) n);
			HIP_OK(hipEventRecord(end, 0));
			HIP_OK(hipEventSynchronize(end));
			HIP_OK(hipEventElapsedTime(&ms, begin, end));
		}
	}
	HIP_OK(hipDeviceSynchronize());
	// results plus the 64 elements after them, which must still hold the fill pattern
	std::vector<result_type> host_out(n + 64);
	HIP_OK(hipMemcpy(host_out.data(), d_out, (n + 64) * sizeof(result_type), hipMemcpyDeviceToHost));
	write_file(out_path, host_out);
	std::printf("OK %.4f\n", ms);
	return 0;
}

/// mode "gather": n = number of indices, segment_length = length of the table, which holds data[i] = (i + 1) * 2654435761 truncated
template <unsigned OutputIndexSize, unsigned ElementSize, unsigned InputIndexSize>
static int run_gather(size_t num_indices, size_t data_length, unsigned grid, size_t offset, const std::string& in_path, const std::string& out_path)
{
	using element_type = cuda::uint_t<ElementSize>;
	using index_type = cuda::uint_t<InputIndexSize>;
	namespace gather = cuda::kernels::gather;
	auto host_indices = read_file<index_type>(in_path, num_indices);
	std::vector<element_type> host_data(data_length);
	for (size_t i = 0; i < data_length; i++) { host_data[i] = (element_type) ((uint64_t) (i + 1) * 2654435761ull); }
	index_type* d_indices_base; element_type* d_data; element_type* d_out_base;
	HIP_OK(hipMalloc(&d_indices_base, (num_indices + offset + 64) * sizeof(index_type)));
	HIP_OK(hipMalloc(&d_out_base, (num_indices + offset + 64) * sizeof(element_type)));
	HIP_OK(hipMalloc(&d_data, (data_length + 1) * sizeof(element_type)));
	index_type* d_indices = d_indices_base + offset; element_type* d_out = d_out_base + offset;
	if (num_indices) { HIP_OK(hipMemcpy(d_indices, host_indices.data(), num_indices * sizeof(index_type), hipMemcpyHostToDevice)); }
	HIP_OK(hipMemcpy(d_data, host_data.data(), data_length * sizeof(element_type), hipMemcpyHostToDevice));
	HIP_OK(hipMemset(d_out_base, 0xCD, (num_indices + offset + 64) * sizeof(element_type)));
	auto properties = cuda::device::current::get().properties();
	gather::launch_config_resolution_params_t<OutputIndexSize, ElementSize, InputIndexSize> params(properties, data_length, num_indices);
	auto config = cuda::kernels::resolve_launch_configuration(params);
	if (grid) { config.grid_dimensions.x = grid; }
	bool cache = config.dynamic_shared_memory_size >= data_length * ElementSize && config.dynamic_shared_memory_size > 0;
	hipEvent_t begin, end; HIP_OK(hipEventCreate(&begin)); HIP_OK(hipEventCreate(&end));
	float ms = 0;
	for (int repetition = 0; repetition < 5 && config.grid_dimensions.x > 0; repetition++) {
		HIP_OK(hipEventRecord(begin, 0));
		cuda::launch(gather::gather<OutputIndexSize, ElementSize, InputIndexSize>, config,
			d_out, (const element_type*) d_data, (const index_type*) d_indices,
			(cuda::size_type_by_index_size<InputIndexSize>) data_length, (cuda::size_type_by_index_size<OutputIndexSize>) num_indices,
			cache, gather::get_default_serialization_factor(cache));
		HIP_OK(hipEventRecord(end, 0));
		HIP_OK(hipEventSynchronize(end));
		HIP_OK(hipEventElapsedTime(&ms, begin, end));
	}
	HIP_OK(hipDeviceSynchronize());
	std::vector<element_type> host_out(num_indices + 64);
	HIP_OK(hipMemcpy(host_out.data(), d_out, (num_indices + 64) * sizeof(element_type), hipMemcpyDeviceToHost));
	write_file(out_path, host_out);
	std::printf("OK %.4f staged_in_lds=%d\n", ms, (int) cache);
	return 0;
}

/// mode "intervals": n outputs, segment_length = modeling period; in.bin holds one coefficient record (int32 x Dimension) per interval
template <unsigned IndexSize, typename Uncompressed, typename Model>
static int run_intervals(size_t n, size_t period, unsigned grid, size_t offset, const std::string& in_path, const std::string& out_path)
{
	namespace model = cuda::kernels::decompression::model;
	using coefficients_type = typename Model::coefficients_type;
	using size_type = cuda::size_type_by_index_size<IndexSize>;
	size_t num_intervals = (n + period - 1) / period;
	auto host_records = read_file<coefficients_type>(in_path, num_intervals);
	coefficients_type* d_records; Uncompressed* d_out_base;
	HIP_OK(hipMalloc(&d_records, (num_intervals + 1) * sizeof(coefficients_type)));
	HIP_OK(hipMalloc(&d_out_base, (n + offset + 64) * sizeof(Uncompressed)));
	Uncompressed* d_out = d_out_base + offset;
	if (num_intervals) { HIP_OK(hipMemcpy(d_records, host_records.data(), num_intervals * sizeof(coefficients_type), hipMemcpyHostToDevice)); }
	HIP_OK(hipMemset(d_out_base, 0xCD, (n + offset + 64) * sizeof(Uncompressed)));
	auto properties = cuda::device::current::get().properties();
	auto kernel = model::decompress_with_intervals<IndexSize, Uncompressed, Model>;
	model::launch_config_resolution_params_t<IndexSize, Uncompressed, Model> params(properties, cuda::device_function_t(kernel).attributes(), n);
	auto config = cuda::kernels::resolve_launch_configuration(params);
	if (grid) { config.grid_dimensions.x = grid; }
	hipEvent_t begin, end; HIP_OK(hipEventCreate(&begin)); HIP_OK(hipEventCreate(&end));
	float ms = 0;
	for (int repetition = 0; repetition < 5 && config.grid_dimensions.x > 0; repetition++) {
		HIP_OK(hipEventRecord(begin, 0));
		cuda::launch(kernel, config, d_out, (const coefficients_type*) d_records, (size_type) period, (size_type) 1, (size_type) n);
		HIP_OK(hipEventRecord(end, 0));
		HIP_OK(hipEventSynchronize(end));
		HIP_OK(hipEventElapsedTime(&ms, begin, end));
	}
	HIP_OK(hipDeviceSynchronize());
	std::vector<Uncompressed> host_out(n + 64);
	HIP_OK(hipMemcpy(host_out.data(), d_out, (n + 64) * sizeof(Uncompressed), hipMemcpyDeviceToHost));
	write_file(out_path, host_out);
	std::printf("OK %.4f\n", ms);
	return 0;
}

int main(int argc, char** argv)
{
	if (argc != 9) { std::fprintf(stderr, "usage: %s instantiation mode n segment_length grid offset in.bin out.bin\n", argv[0]); return 2; }
	std::string which = argv[1], mode = argv[2];
	size_t n = std::strtoull(argv[3], nullptr, 10), L = std::strtoull(argv[4], nullptr, 10);
	unsigned grid = (unsigned) std::strtoul(argv[5], nullptr, 10);
	size_t offset = std::strtoull(argv[6], nullptr, 10);
	std::string in = argv[7], out = argv[8];
	if (mode == "intervals") {
		namespace models = cuda::functors::unary::parametric_model;
		if (which == "intervals_4_int_linear")    { return run_intervals<4, int32_t, models::linear<4, int32_t>>(n, L, grid, offset, in, out); }
		if (which == "intervals_4_short_linear")  { return run_intervals<4, int16_t, models::linear<4, int32_t>>(n, L, grid, offset, in, out); }
		if (which == "intervals_8_int_constant")  { return run_intervals<8, int32_t, models::constant<4, int32_t>>(n, L, grid, offset, in, out); }
		std::fprintf(stderr, "unknown instantiation %s\n", which.c_str());
		return 2;
	}
	if (mode == "gather") {
		if (which == "gather_4_4_2") { return run_gather<4, 4, 2>(n, L, grid, offset, in, out); }
		if (which == "gather_4_1_1") { return run_gather<4, 1, 1>(n, L, grid, offset, in, out); }
		if (which == "gather_8_8_4") { return run_gather<8, 8, 4>(n, L, grid, offset, in, out); }
		if (which == "gather_4_2_4") { return run_gather<4, 2, 4>(n, L, grid, offset, in, out); }
		std::fprintf(stderr, "unknown instantiation %s\n", which.c_str());
		return 2;
	}
	using u16 = uint16_t; using u32 = uint32_t; using u64 = uint64_t; using i32 = int32_t; using i64 = int64_t;
	if (which == "plus_u32_u32_inclusive")       { return run<4, f::plus<u32>, u32, true,  f::identity<u32>>(mode, n, L, grid, offset, in, out); }
	if (which == "plus_u32_u32_exclusive")       { return run<4, f::plus<u32>, u32, false, f::identity<u32>>(mode, n, L, grid, offset, in, out); }
	if (which == "plus_u64_u16_inclusive")       { return run<4, f::plus<u64>, u16, true,  f::cast<u16, u64>>(mode, n, L, grid, offset, in, out); }   // Delta without anchors
	if (which == "plus_u64_u32_inclusive_idx8")  { return run<8, f::plus<u64>, u32, true,  f::cast<u32, u64>>(mode, n, L, grid, offset, in, out); }
	if (which == "plus_u16_u16_inclusive")       { return run<4, f::plus<u16>, u16, true,  f::identity<u16>>(mode, n, L, grid, offset, in, out); }
	if (which == "maximum_i32_i32_inclusive")    { return run<4, f::maximum<i32>, i32, true,  f::identity<i32>>(mode, n, L, grid, offset, in, out); }
	if (which == "maximum_i32_i32_exclusive")    { return run<4, f::maximum<i32>, i32, false, f::identity<i32>>(mode, n, L, grid, offset, in, out); }
	if (which == "minimum_i64_i64_inclusive")    { return run<8, f::minimum<i64>, i64, true,  f::identity<i64>>(mode, n, L, grid, offset, in, out); }
	if (which == "bit_or_u32_u32_inclusive")     { return run<4, f::bit_or<u32>, u32, true,  f::identity<u32>>(mode, n, L, grid, offset, in, out); }
	if (which == "plus_i64_absolute_i32")        { return run<4, f::plus<i64>, i32, true,  f::absolute<i64>>(mode, n, L, grid, offset, in, out); }
	if (which == "plus_f64_f64_inclusive")       { return run<4, f::plus<double>, double, true, f::identity<double>>(mode, n, L, grid, offset, in, out); }
	std::fprintf(stderr, "unknown instantiation %s\n", which.c_str());
	return 2;
}

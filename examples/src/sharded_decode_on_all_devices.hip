// A run-length column cut over the GPUs of a node, decoded shard by shard, gathered into one buffer — ONE process, in C++:
//
//   for every shard r of `world` (default 8):   device r % (devices present)
//     giddy/sharding.hpp           which slices of the column's arrays the shard reads, what its anchors are rebased by
//     hipMemcpyPeerAsync           those slices from the device that holds the column to the shard's device
//     giddy_enqueue_rebase         the shard's anchors made to refer to its own run slice, on its device
//     run_length_encoding::kernel_t   the decoder, through the wrapper's typed interface (mode of use B)
//     hipMemcpyPeerAsync           the decoded shard into its place in the contiguous output on device 0
//
// No collective, no host round trip on the data path (two scalar reads of the anchors per shard decide the cuts).  On a box with one GPU
// all the shards run on it — the flow and the checks are the same.  Prints OK when the gathered column equals the host's expansion.
//     make -C examples && examples/bin/sharded_decode_on_all_devices [world=8] [log2 elements=24]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "giddy.h"
#include "giddy/api/device.hpp"
#include "giddy/kernel_wrappers/decompression/run_length_encoding.hpp"
#include "giddy/sharding.hpp"

#define CHECK(call) do { hipError_t status_ = (call); if (status_ != hipSuccess) { std::printf("%s: %s\n", #call, hipGetErrorString(status_)); return 1; } } while (0)

int main(int argc, char** argv)
{
	const unsigned world = argc > 1 ? (unsigned) std::atoi(argv[1]) : 8;
	const uint64_t n = (uint64_t) 1 << (argc > 2 ? std::atoi(argv[2]) : 24);
	const uint32_t L = 4096;
	int devices = 0;
	CHECK(hipGetDeviceCount(&devices));
	if (devices == 0 || world == 0) { std::printf("no device\n"); return 1; }

	// a column of runs (lengths 1 .. 64) and its run-length format: values, 16-bit lengths, per segment the run it begins in and how far into it
	std::vector<uint32_t> column(n), run_data, anchors; std::vector<uint16_t> run_lengths, intra;
	uint64_t state = 88172645463325252ull;
	for (uint64_t i = 0; i < n; ) {
		state ^= state << 13; state ^= state >> 7; state ^= state << 17;
		const uint64_t length = std::min<uint64_t>(1 + state % 64, n - i);
		run_data.push_back((uint32_t) (state >> 32)); run_lengths.push_back((uint16_t) length);
		for (uint64_t j = 0; j < length; j++) { column[i + j] = run_data.back(); }
		i += length;
	}
	const uint64_t num_runs = run_data.size(), num_segments = (n + L - 1) / L;
	for (uint64_t s = 0, run = 0, run_start = 0; s < num_segments; s++) {
		while (run_start + run_lengths[run] <= s * L) { run_start += run_lengths[run]; run++; }
		anchors.push_back((uint32_t) run); intra.push_back((uint16_t) (s * L - run_start));
	}

	// the column lives on device 0
	uint32_t *d_data, *d_anchors, *d_out; uint16_t *d_lengths, *d_intra;
	CHECK(hipSetDevice(0));
	CHECK(hipMalloc(&d_data, num_runs * 4)); CHECK(hipMalloc(&d_lengths, num_runs * 2)); CHECK(hipMalloc(&d_anchors, num_segments * 4));
	CHECK(hipMalloc(&d_intra, num_segments * 2)); CHECK(hipMalloc(&d_out, n * 4));
	CHECK(hipMemcpy(d_data, run_data.data(), num_runs * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_lengths, run_lengths.data(), num_runs * 2, hipMemcpyHostToDevice));
	CHECK(hipMemcpy(d_anchors, anchors.data(), num_segments * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_intra, intra.data(), num_segments * 2, hipMemcpyHostToDevice));
	CHECK(hipMemset(d_out, 0, n * 4));

	namespace rle = cuda::kernels::decompression::run_length_encoding;
	rle::kernel_t<4, 4, 2> decoder;
	struct on_device { int device; hipStream_t stream; uint32_t *data, *anchors, *out; uint16_t *lengths, *intra; };
	std::vector<on_device> shards(world);
	for (unsigned r = 0; r < world; r++) {
		// the cut: the two anchors it needs are read from the device that holds them (host copies here, as the column was made on the host)
		const auto shard = cuda::sharding::run_length_encoding(n, L, sizeof(uint32_t), num_segments, num_runs, [&](uint64_t s) { return anchors[s]; }, r, world);
		on_device& d = shards[r];
		d.device = (int) (r % (unsigned) devices);
		CHECK(hipSetDevice(d.device));
		CHECK(hipStreamCreate(&d.stream));
		if (shard.length() == 0) { d.out = nullptr; continue; }
		const uint64_t runs = shard.units.size(), segments = shard.segments.size();
		CHECK(hipMalloc(&d.data, runs * 4 + 64)); CHECK(hipMalloc(&d.lengths, runs * 2 + 64)); CHECK(hipMalloc(&d.anchors, segments * 4)); CHECK(hipMalloc(&d.intra, segments * 2));
		CHECK(hipMalloc(&d.out, shard.length() * 4));
		CHECK(hipMemcpyPeerAsync(d.data, d.device, d_data + shard.units.first, 0, runs * 4, d.stream));
		CHECK(hipMemcpyPeerAsync(d.lengths, d.device, d_lengths + shard.units.first, 0, runs * 2, d.stream));
		CHECK(hipMemcpyPeerAsync(d.anchors, d.device, d_anchors + shard.segments.first, 0, segments * 4, d.stream));
		CHECK(hipMemcpyPeerAsync(d.intra, d.device, d_intra + shard.segments.first, 0, segments * 2, d.stream));
		if (giddy_enqueue_rebase(d.stream, d.anchors, d.anchors, segments, 4, shard.anchor_bias) != GIDDY_SUCCESS) { std::printf("rebase: %s\n", giddy_last_error()); return 1; }
		const auto properties = cuda::device::current::get().properties();
		const auto config = decoder.resolve_launch_configuration(properties, decoder.get_device_function().attributes(), shard.length(), L, cuda::launch_configuration_limits_t { cuda::nullopt, cuda::nullopt, cuda::nullopt });
		cuda::kernel::enqueue_launch(decoder, d.stream, config, d.out, (const uint32_t*) d.data, (const uint16_t*) d.lengths, (const uint32_t*) d.anchors, (const uint16_t*) d.intra,
			(uint32_t) L, (uint32_t) segments, (uint32_t) runs, (uint32_t) shard.length());
		// the one transfer a contiguous result costs: the shard into its place on device 0 (over xGMI between devices)
		CHECK(hipMemcpyPeerAsync(d_out + shard.elements.first, 0, d.out, d.device, shard.length() * 4, d.stream));
	}
	for (auto& d : shards) { CHECK(hipSetDevice(d.device)); CHECK(hipStreamSynchronize(d.stream)); }
	std::vector<uint32_t> got(n);
	CHECK(hipSetDevice(0));
	CHECK(hipMemcpy(got.data(), d_out, n * 4, hipMemcpyDeviceToHost));
	for (uint64_t i = 0; i < n; i++) { if (got[i] != column[i]) { std::printf("mismatch at %llu: %u, expected %u\n", (unsigned long long) i, got[i], column[i]); return 1; } }
	std::printf("OK: %llu elements, %u shards on %d device(s), gathered on device 0\n", (unsigned long long) n, world, devices);
	return 0;
}

// (debugging aid) fine::last_start_at_most against a thread's binary search, on random ascending arrays
#include "giddy/kernel_wrappers/common.hpp"
#include "giddy/kernels/decompression/run_anchors.hpp"
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
using namespace cuda::kernels::decompression::run_expansion;
__global__ void probe(const uint16_t* starts, const uint64_t* lows, const uint64_t* highs, const uint64_t* targets, uint64_t* out, unsigned count)
{
	for (unsigned q = blockIdx.x; q < count; q += gridDim.x) {
		const uint64_t r = fine::last_start_at_most(starts, lows[q], highs[q], targets[q], threadIdx.x);
		if (threadIdx.x == 0) { out[q] = r; }
	}
}
int main()
{
	const unsigned n = 50000, count = 20000;
	std::vector<uint16_t> starts(n); std::vector<uint64_t> lows(count), highs(count), targets(count), expect(count), got(count);
	srand(7);
	unsigned v = 0; for (unsigned i = 0; i < n; i++) { v += rand() % 3; starts[i] = (uint16_t) (v % 60000); }
	std::sort(starts.begin(), starts.end());
	for (unsigned q = 0; q < count; q++) {
		lows[q] = rand() % n; highs[q] = lows[q] + rand() % std::min<unsigned>(n - lows[q], 1 + rand() % 3000); targets[q] = rand() % 60000;
		uint64_t low = lows[q], high = highs[q];
		while (low < high) { const uint64_t middle = low + (high - low + 1) / 2; if (starts[middle] <= targets[q]) { low = middle; } else { high = middle - 1; } }
		expect[q] = low;
	}
	uint16_t* d_starts; uint64_t *d_lows, *d_highs, *d_targets, *d_out;
	hipMalloc(&d_starts, n * 2); hipMalloc(&d_lows, count * 8); hipMalloc(&d_highs, count * 8); hipMalloc(&d_targets, count * 8); hipMalloc(&d_out, count * 8);
	hipMemcpy(d_starts, starts.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(d_lows, lows.data(), count * 8, hipMemcpyHostToDevice);
	hipMemcpy(d_highs, highs.data(), count * 8, hipMemcpyHostToDevice); hipMemcpy(d_targets, targets.data(), count * 8, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(probe, dim3(256), dim3(64), 0, 0, d_starts, d_lows, d_highs, d_targets, d_out, count);
	hipMemcpy(got.data(), d_out, count * 8, hipMemcpyDeviceToHost);
	unsigned bad = 0;
	for (unsigned q = 0; q < count; q++) { if (got[q] != expect[q]) { if (bad++ < 5) { printf("query %u: low %lu high %lu target %lu: got %lu expected %lu\n", q, lows[q], highs[q], targets[q], got[q], expect[q]); } } }
	printf("%u of %u wrong\n", bad, count);
	return bad != 0;
}

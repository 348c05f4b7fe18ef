// development aid (GPU box): what HIP says about how many workgroups of the Delta decoders fit a compute unit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/occupancy_probe.hip -o /tmp/occupancy_probe && /tmp/occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include "giddy/kernels/decompression/delta.hpp"
template <unsigned I, unsigned U, unsigned C> void probe()
{
	int blocks = 0;
	auto kernel = cuda::kernels::decompression::delta::decompress<I, U, C>;
	hipFuncAttributes a{};
	(void) hipFuncGetAttributes(&a, reinterpret_cast<const void*>(kernel));
	for (int threads : { 256, 512, 1024 }) {
		(void) hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kernel, threads, 0);
		std::printf("delta<%u,%u,%u> %4d threads: %2d workgroups per CU (numRegs %d, static LDS %zu)\n", I, U, C, threads, blocks, a.numRegs, a.sharedSizeBytes);
	}
}
int main() { probe<4, 4, 2>(); probe<4, 4, 1>(); probe<4, 8, 2>(); probe<4, 8, 1>(); probe<8, 8, 4>(); probe<4, 2, 1>(); return 0; }

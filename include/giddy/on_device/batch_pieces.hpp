// Device side of giddy/batch.hpp: the ONE kernel template of a batch launch.  Workgroup w reads its entry of the block table
// and its piece's record — wave-uniform addresses in memory nobody writes during the launch: scalar loads, two dependent ones
// where an ordinary launch has one (its kernel arguments) — and runs the decoder's body on the piece as block `block_in_piece`
// of `blocks` (launch_grid::part).  `Body::run(piece, grid)` unpacks the record's slots into the body's typed parameters.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "giddy/batch.hpp"
#include "giddy/on_device/launch_grid.hpp"
#include "giddy/on_device/wave64.hpp"

namespace cuda {
namespace batch {

/// a 64-bit word of a table, through the scalar cache (the tables are written before the launch and only read by it)
__device__ __forceinline__ uint64_t uniform_word(const void* address)
{
	const uint64_t value = *reinterpret_cast<const uint64_t __attribute__((address_space(4)))*>(reinterpret_cast<uintptr_t>(address));
	return wave64::read_first_lane(value);   // (a no-op on a value the compiler already holds in scalar registers; says "uniform" where it could not tell)
}

/// a piece's record as its body reads it: get<T>(i) = the kernel's i-th parameter
struct piece_view {
	const piece_t* record;
	template <typename T>
	__device__ __forceinline__ T get(unsigned i) const
	{
		static_assert(sizeof(T) <= 8, "a launch parameter that does not fit a 64-bit slot");
		const uint64_t slot = uniform_word(&record->slot[i]);
		if constexpr (std::is_pointer<T>::value) {
			// A pointer made from an integer has no address space, and its loads would be FLAT ones (which count against the LDS counter as well): say
			// where it points first.  INPUT arrays (pointers to const) are named as CONSTANT memory — what `const T* __restrict__` kernel parameters tell
			// the compiler of an ordinary launch: nothing the kernel stores can change them, so a load at a wave-uniform address (a segment's anchor) is
			// a SCALAR load, which does not queue behind the wavefront's stores (Delta <4,8,2> in a batch: 62 % of the HBM peak with vector loads of the
			// anchors, against 77 % for the column's own launch).  The decoders never decode in place (their parameters are __restrict__).
			using pointee = typename std::remove_pointer<T>::type;
			if constexpr (std::is_const<pointee>::value) {
				typedef pointee __attribute__((address_space(4)))* constant_pointer;
				return (T) reinterpret_cast<constant_pointer>(static_cast<uintptr_t>(slot));
			}
			else {
				typedef pointee __attribute__((address_space(1)))* global_pointer;
				return (T) reinterpret_cast<global_pointer>(static_cast<uintptr_t>(slot));
			}
		}
		else if constexpr (std::is_same<T, bool>::value) { return (slot & 0xFF) != 0; }
		else if constexpr (std::is_integral<T>::value || std::is_enum<T>::value) { return (T) slot; }
		else { T value; __builtin_memcpy(&value, &slot, sizeof(T)); return value; }
	}
};

/// what the calling workgroup decodes: its piece's record and its place among the piece's blocks
struct located_piece { piece_view piece; launch_grid::part grid; };
__device__ __forceinline__ located_piece piece_of_this_block(const piece_t* __restrict__ pieces, const block_t* __restrict__ blocks)
{
	const uint64_t mine = uniform_word(blocks + blockIdx.x);            // { piece, block_in_piece }
	const piece_t* record = pieces + (uint32_t) mine;
	const uint64_t counts = uniform_word(&record->blocks);              // { blocks, first_block }
	return located_piece { piece_view { record }, launch_grid::part { (unsigned) (mine >> 32), (unsigned) (uint32_t) counts } };
}

template <typename Body>
__global__ void __launch_bounds__(Body::max_threads_per_block) decode_pieces(const piece_t* __restrict__ pieces, const block_t* __restrict__ blocks)
{
	const located_piece mine = piece_of_this_block(pieces, blocks);
	Body::run(mine.piece, mine.grid);
}

} // namespace batch
} // namespace cuda

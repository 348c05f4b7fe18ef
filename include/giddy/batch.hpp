// MANY small columns decoded by ONE launch.
//
// The reference's unit of work is one launch per column (src/cuda/kernel_wrapper.cuh:135-147); its premise is chunks of
// compressed columns crossing a link and being expanded where they arrive (README.md:15-24) — and chunks are small.  A
// launch costs the host 3-10 us and the device a ramp and a tail: on MI355X a column of 2^20 elements decodes at 5-20 %
// of the rate of a column of 2^28, one of 2^16 at 1 % (profiles/r05_size_sweep_round_start.txt).  A BATCH is a table of
// PIECES — each the parameter list of the wrapper's kernel for one column, as a direct launch would pass it — and a table
// of BLOCKS that deals the workgroups of one launch out to the pieces: workgroup w runs as "block blocks[w].block_in_piece
// of the pieces[blocks[w].piece].blocks blocks of that piece" (on_device/launch_grid.hpp: the decoders' bodies stride by
// that instead of blockIdx.x / gridDim.x).  Pieces are independent columns: any lengths (0 and 1 included), any segment
// lengths, each with the number of blocks its own launch configuration would have had.
//
//   cuda::batch::table_t table;
//   kernel.add_to_batch(table, kernel.resolve_launch_configuration(device, piece_resolution_arguments), piece_launch_arguments);   // per piece
//   cuda::batch::uploaded_t on_device(table, stream);          // one asynchronous copy of both tables (pinned staging)
//   kernel.enqueue_launch_of_batch(stream, on_device.view());  // ONE launch; as often as wanted while `on_device` lives
//
// Both tables may equally be WRITTEN BY THE DEVICE (a kernel that cuts a column it has just received): the layouts below
// are the contract, `giddy_enqueue_batch_of_device_tables` (giddy.h) the entry point that takes them as they lie.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <type_traits>
#include <utility>
#include <vector>

#include "giddy/api/types.hpp"

namespace cuda {
namespace batch {

enum : unsigned { slots_per_piece = 12, hints_per_batch = 4 };

/// One piece: parameter i of the kernel of the reference's signature (the wrapper's `launch` parameters, in order) in slot i,
/// zero-extended to 64 bits (pointers: the address; bool: 0 / 1; little-endian, so a value's bytes are the slot's first)
struct alignas(16) piece_t {
	uint64_t  slot[slots_per_piece];
	uint32_t  blocks;        // workgroups of the batch launch that decode this piece (0: none — an empty piece)
	uint32_t  first_block;   // ... of which the first is this workgroup of the launch
	uint64_t  reserved;
};
static_assert(sizeof(piece_t) == 112, "piece records are 112 bytes (giddy.h: giddy_batch_piece_t)");

/// workgroup w of the launch: block `block_in_piece` of piece `piece`
struct block_t { uint32_t piece, block_in_piece; };

template <typename T>
inline uint64_t as_slot(T value)
{
	static_assert(std::is_trivially_copyable<T>::value && sizeof(T) <= 8, "a launch parameter that does not fit a 64-bit slot");
	uint64_t slot = 0;
	std::memcpy(&slot, &value, sizeof(T));
	return slot;
}

/// what enqueue_launch_of_batch() takes: both tables where the device reads them, and what the launch needs to know of them
struct on_device_t {
	const piece_t*          pieces = nullptr;
	const block_t*          blocks = nullptr;
	uint32_t                num_pieces = 0, num_blocks = 0;
	grid_block_dimension_t  block_size = 0;
	shared_memory_size_t    dynamic_shared_memory_size = 0;
	uint64_t                hints[hints_per_batch] = { };   // per scheme (run decoders: outputs, runs, whether every run array is 16-byte aligned)
};

/// The tables as the host builds them.  add(): a piece with the blocks of ITS OWN resolved configuration; the launch takes the largest
/// block size and the most dynamic LDS any piece asked for (every decoder body is correct under any configuration: SURVEY.md section 8b).
class table_t {
public:
	template <typename... Parameters>
	void add(const launch_configuration_t& piece_configuration, Parameters... parameters)
	{
		static_assert(sizeof...(Parameters) <= slots_per_piece, "more launch parameters than a piece record has slots");
		piece_t piece { };
		unsigned i = 0;
		(..., (piece.slot[i++] = as_slot(parameters)));
		add_packed(piece_configuration, piece);
	}

	/// ... with the slots filled in by the caller
	void add_packed(const launch_configuration_t& piece_configuration, piece_t piece)
	{
		const size_t blocks = piece_configuration.grid_dimensions.volume();
		if (blocks_.size() + blocks > 0x7FFFFFFFull) { throw std::invalid_argument("a batch of more than 2^31 - 1 workgroups"); }
		piece.blocks = (uint32_t) blocks;
		piece.first_block = (uint32_t) blocks_.size();
		const uint32_t index = (uint32_t) pieces_.size();
		for (uint32_t b = 0; b < piece.blocks; b++) { blocks_.push_back(block_t { index, b }); }
		pieces_.push_back(piece);
		const grid_block_dimension_t block_size = (grid_block_dimension_t) piece_configuration.block_dimensions.volume();
		if (block_size > block_size_) { block_size_ = block_size; }
		if (piece_configuration.dynamic_shared_memory_size > dynamic_shared_memory_size_) { dynamic_shared_memory_size_ = piece_configuration.dynamic_shared_memory_size; }
	}

	void clear() { pieces_.clear(); blocks_.clear(); block_size_ = 0; dynamic_shared_memory_size_ = 0; for (auto& h : hints) { h = 0; } }

	const std::vector<piece_t>& pieces() const { return pieces_; }
	const std::vector<block_t>& blocks() const { return blocks_; }
	std::vector<piece_t>& pieces() { return pieces_; }
	grid_block_dimension_t block_size() const { return block_size_; }
	shared_memory_size_t dynamic_shared_memory_size() const { return dynamic_shared_memory_size_; }
	size_t bytes() const { return pieces_.size() * sizeof(piece_t) + blocks_.size() * sizeof(block_t); }

	uint64_t hints[hints_per_batch] = { };

private:
	std::vector<piece_t>    pieces_;
	std::vector<block_t>    blocks_;
	grid_block_dimension_t  block_size_ = 0;
	shared_memory_size_t    dynamic_shared_memory_size_ = 0;
};

/// PINNED host memory for the copies of tables to devices: a process-wide pool of blocks, each re-used only once the copy that was made from it has
/// completed (an event query, never a wait) — so a table may be re-built and uploaded again, or a function that uploaded one may return, while the
/// copy is still in flight.  The memory is portable, but a block's EVENT belongs to the device that was current when the block was made: take() hands a
/// block out only on that device (a process that drives several GPUs gets blocks per device), and a block whose event cannot be recorded is retired —
/// never a wait on the host.  Blocks live as long as the process.
class staging_pool {
public:
	struct block { void* memory = nullptr; size_t capacity = 0; hipEvent_t copied = nullptr; int device = -1; bool in_flight = false, taken = false, retired = false; };

	static staging_pool& instance() { static staging_pool pool; return pool; }

	/// a block of at least `bytes`, made on the current device, that no copy is reading: fill it, enqueue the copy, then copy_enqueued()
	block* take(size_t bytes)
	{
		int device = -1;
		throw_if_error(hipGetDevice(&device), "hipGetDevice for a batch's staging");
		std::lock_guard<std::mutex> lock(mutex_);
		for (auto& b : blocks_) {
			if (b->taken || b->retired || b->device != device) { continue; }
			if (b->in_flight) {
				const hipError_t status = hipEventQuery(b->copied);
				if (status == hipSuccess) { b->in_flight = false; }
				else if (status != hipErrorNotReady) { (void) hipGetLastError(); }
			}
			if (!b->in_flight && b->capacity >= bytes) { b->taken = true; return b.get(); }
		}
		std::unique_ptr<block> fresh(new block);
		fresh->capacity = bytes + bytes / 4 + 4096;
		fresh->device = device;
		throw_if_error(hipHostMalloc(&fresh->memory, fresh->capacity, hipHostMallocPortable), "hipHostMalloc of staging for a batch's tables");
		const hipError_t status = hipEventCreateWithFlags(&fresh->copied, hipEventDisableTiming);
		if (status != hipSuccess) { (void) hipHostFree(fresh->memory); throw_if_error(status, "hipEventCreate for a batch's staging"); }
		fresh->taken = true;
		blocks_.push_back(std::move(fresh));
		return blocks_.back().get();
	}

	/// the copy out of `b` has been enqueued on `stream` (of the device `b` was taken on): `b` is anybody's again once it has run.  If the event
	/// cannot be recorded, the block is never handed out again (its memory stays valid for the copy in flight and is not freed).
	void copy_enqueued(block* b, stream::id_t stream)
	{
		const bool recorded = hipEventRecord(b->copied, stream) == hipSuccess;
		if (!recorded) { (void) hipGetLastError(); }
		std::lock_guard<std::mutex> lock(mutex_);
		b->in_flight = recorded;
		b->retired = !recorded;
		b->taken = false;
	}

	/// (for tests) blocks made so far, and how many of them are retired
	std::pair<size_t, size_t> counts()
	{
		std::lock_guard<std::mutex> lock(mutex_);
		size_t retired = 0;
		for (auto& b : blocks_) { retired += b->retired ? 1 : 0; }
		return { blocks_.size(), retired };
	}

private:
	std::mutex mutex_;
	std::vector<std::unique_ptr<block>> blocks_;
};

/// bytes of device memory a table's copy takes
inline size_t device_bytes_of(const table_t& table) { return table.bytes(); }

/// `table` written to `host_memory` as it will lie at `device_memory` once copied there (both >= device_bytes_of(table), 16-byte aligned): for callers
/// that move several tables — or tables and something else — with ONE copy of their own
inline on_device_t place(void* host_memory, const void* device_memory, const table_t& table)
{
	const size_t pieces_bytes = table.pieces().size() * sizeof(piece_t), blocks_bytes = table.blocks().size() * sizeof(block_t);
	if (pieces_bytes != 0) { std::memcpy(host_memory, table.pieces().data(), pieces_bytes); }
	if (blocks_bytes != 0) { std::memcpy(static_cast<char*>(host_memory) + pieces_bytes, table.blocks().data(), blocks_bytes); }
	on_device_t view;
	view.pieces = static_cast<const piece_t*>(device_memory);
	view.blocks = reinterpret_cast<const block_t*>(static_cast<const char*>(device_memory) + pieces_bytes);
	view.num_pieces = (uint32_t) table.pieces().size();
	view.num_blocks = (uint32_t) table.blocks().size();
	view.block_size = table.block_size();
	view.dynamic_shared_memory_size = table.dynamic_shared_memory_size();
	for (unsigned i = 0; i < hints_per_batch; i++) { view.hints[i] = table.hints[i]; }
	return view;
}

/// `table` copied into `device_memory` (>= device_bytes_of(table), 16-byte aligned) asynchronously on `stream`, out of the staging pool
inline on_device_t upload_into(void* device_memory, const table_t& table, stream::id_t stream)
{
	const size_t bytes = device_bytes_of(table);
	if (bytes == 0) { return place(nullptr, device_memory, table); }
	staging_pool::block* from = staging_pool::instance().take(bytes);
	const on_device_t view = place(from->memory, device_memory, table);
	const hipError_t status = hipMemcpyAsync(device_memory, from->memory, bytes, hipMemcpyHostToDevice, stream);
	staging_pool::instance().copy_enqueued(from, stream);
	throw_if_error(status, "copy of a batch's tables to the device");
	return view;
}

/// A table's copy on the current device, in memory of its own: kept and re-used when a later table fits (a re-upload is ordered behind the earlier
/// launches by the stream; a caller that re-uploads on ANOTHER stream orders the two itself).  Lives until the launches that read it have run (as
/// a giddy_prepared_t does); the destructor frees device memory, which waits for the device.
class uploaded_t {
public:
	uploaded_t() = default;
	uploaded_t(const table_t& table, stream::id_t stream) { upload(table, stream); }
	uploaded_t(const uploaded_t&) = delete;
	uploaded_t& operator=(const uploaded_t&) = delete;
	~uploaded_t() { if (device_ != nullptr) { (void) hipFree(device_); } }

	void upload(const table_t& table, stream::id_t stream)
	{
		const size_t bytes = device_bytes_of(table);
		if (bytes > capacity_) {
			if (device_ != nullptr) { (void) hipFree(device_); device_ = nullptr; capacity_ = 0; }
			const size_t wanted = bytes + bytes / 4 + 4096;
			throw_if_error(hipMalloc(&device_, wanted), "hipMalloc of a batch's tables");
			capacity_ = wanted;
		}
		view_ = upload_into(device_, table, stream);
	}

	const on_device_t& view() const { return view_; }

private:
	void* device_ = nullptr;
	size_t capacity_ = 0;
	on_device_t view_;
};

} // namespace batch
} // namespace cuda

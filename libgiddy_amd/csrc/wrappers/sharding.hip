// giddy_shard / giddy_enqueue_rebase / giddy_decode_from_host (include/giddy.h): a column's cuts, the rebasing of a shard's anchors on
// the device, and the chunked host-to-device pipeline, for callers of the C ABI.  The cuts are include/giddy/sharding.hpp's.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../c_abi_internal.hpp"
#include "giddy/api/device.hpp"
#include "giddy/kernels/data_layout/rebase.hpp"
#include "giddy/sharding.hpp"
#include "giddy/stream_scratch.hpp"

namespace {

using giddy_internal::guarded;

/// an event for ordering only, destroyed with its scope (the runtime releases it once a pending record has completed)
struct ordering_event {
	hipEvent_t event = nullptr;
	ordering_event() { cuda::throw_if_error(hipEventCreateWithFlags(&event, hipEventDisableTiming), "hipEventCreateWithFlags"); }
	ordering_event(const ordering_event&) = delete;
	ordering_event& operator=(const ordering_event&) = delete;
	~ordering_event() { if (event != nullptr) { (void) hipEventDestroy(event); } }
	void record(hipStream_t stream) { cuda::throw_if_error(hipEventRecord(event, stream), "hipEventRecord"); }
	void make_wait(hipStream_t stream) { cuda::throw_if_error(hipStreamWaitEvent(stream, event, 0), "hipStreamWaitEvent"); }
};

/// `from`'s copies to and from `to` should go device to device (over xGMI) rather than staged through the host: checked, and switched on — ONCE per pair
/// and process (the answer is kept: the launch path does not repeat the two runtime calls).  Where the devices cannot reach each other directly (no
/// P2P: virtual machines, ACS / IOMMU set-ups, mixed hives) hipMemcpyPeerAsync still works — the runtime stages the copy through the host —, so the
/// gathers go on, slower, and say so once on stderr; GIDDY_REQUIRE_PEER_ACCESS=1 in the environment makes that an error instead (a benchmark that
/// must not silently measure the host's memory).  Returns whether the pair has direct access.
bool enable_peer_access(int from, int to)
{
	if (from == to) { return true; }
	static std::mutex mutex;
	static std::map<std::pair<int, int>, bool> known;
	static const bool required = [] { const char* v = std::getenv("GIDDY_REQUIRE_PEER_ACCESS"); return v != nullptr && v[0] == '1'; }();
	std::lock_guard<std::mutex> lock(mutex);
	auto found = known.find({ from, to });
	if (found == known.end()) {
		int can = 0;
		cuda::throw_if_error(hipDeviceCanAccessPeer(&can, from, to), "hipDeviceCanAccessPeer");
		if (can) {
			cuda::device::current::scoped_override_t<> on(from);
			const hipError_t status = hipDeviceEnablePeerAccess(to, 0);
			if (status == hipErrorPeerAccessAlreadyEnabled) { (void) hipGetLastError(); }
			else { cuda::throw_if_error(status, "hipDeviceEnablePeerAccess"); }
		}
		else if (!required) {
			std::fprintf(stderr, "libgiddy_amd: device %d cannot access device %d directly: copies between them are staged through the host (GIDDY_REQUIRE_PEER_ACCESS=1 makes this an error)\n", from, to);
		}
		found = known.emplace(std::make_pair(from, to), can != 0).first;
	}
	if (!found->second && required) {
		throw cuda::runtime_error(hipErrorPeerAccessUnsupported, "device " + std::to_string(from) + " cannot access device " + std::to_string(to) + " directly (GIDDY_REQUIRE_PEER_ACCESS=1): the gather would be staged through the host");
	}
	return found->second;
}

uint64_t anchor_value(const giddy_column_shape_t& shape, uint64_t index)
{
	if (shape.position_anchors == nullptr) { throw std::invalid_argument("this scheme's cuts read the column's position_anchors: none given"); }
	if (shape.anchor_itemsize != 1 && shape.anchor_itemsize != 2 && shape.anchor_itemsize != 4 && shape.anchor_itemsize != 8) { throw std::invalid_argument("anchors of 1, 2, 4 or 8 bytes"); }
	uint64_t value = 0;
	const char* at = static_cast<const char*>(shape.position_anchors) + index * shape.anchor_itemsize;
	if (shape.anchors_on_device) { cuda::throw_if_error(hipMemcpy(&value, at, shape.anchor_itemsize, hipMemcpyDeviceToHost), "hipMemcpy (an anchor)"); }
	else { std::memcpy(&value, at, shape.anchor_itemsize); }   // (little-endian host)
	return value;
}

cuda::sharding::shard_t cut(int scheme, const giddy_column_shape_t& shape, uint64_t rank, uint64_t world_size)
{
	namespace sh = cuda::sharding;
	auto at = [&](uint64_t i) { return anchor_value(shape, i); };
	switch (scheme) {
	case GIDDY_SCHEME_ELEMENTWISE:        return sh::elementwise(shape.length, rank, world_size);
	case GIDDY_SCHEME_FRAME_OF_REFERENCE: return sh::frame_of_reference(shape.length, shape.segment_length, shape.output_itemsize, rank, world_size);
	case GIDDY_SCHEME_DELTA:              return sh::delta(shape.length, shape.segment_length, shape.output_itemsize, rank, world_size);
	case GIDDY_SCHEME_DISCARD_ZERO_BYTES_VARIABLE:
		return sh::discard_zero_bytes_variable(shape.length, shape.segment_length, shape.bits_per_element_size, shape.size_container_size,
			shape.output_itemsize, shape.num_anchors, shape.num_units, at, rank, world_size);
	case GIDDY_SCHEME_RUN_LENGTH_ENCODING:
		return sh::run_length_encoding(shape.length, shape.segment_length, shape.output_itemsize, shape.num_anchors, shape.num_units, at, rank, world_size);
	case GIDDY_SCHEME_RUN_POSITION_ENCODING_RELATIVE:
	case GIDDY_SCHEME_RUN_POSITION_ENCODING_ABSOLUTE:
		return sh::run_position_encoding(shape.length, shape.segment_length, shape.output_itemsize, shape.num_anchors, shape.num_units,
			scheme == GIDDY_SCHEME_RUN_POSITION_ENCODING_RELATIVE, at, rank, world_size);
	default: throw std::invalid_argument("unknown giddy_scheme_t " + std::to_string(scheme));
	}
}

void enqueue_rebase(hipStream_t stream, void* out, const void* in, uint64_t count, uint32_t itemsize, uint64_t bias)
{
	if (count == 0) { return; }
	namespace rb = cuda::kernels::data_layout::rebase;
	const cuda::launch_configuration_t config((cuda::grid_dimension_t) std::min<uint64_t>((count + 255) / 256, 4096), 256);
	switch (itemsize) {
	case 1: cuda::enqueue_launch(reinterpret_cast<const void*>(rb::rebase<1>), stream, config, static_cast<uint8_t*>(out), static_cast<const uint8_t*>(in), count, (uint8_t) bias); break;
	case 2: cuda::enqueue_launch(reinterpret_cast<const void*>(rb::rebase<2>), stream, config, static_cast<uint16_t*>(out), static_cast<const uint16_t*>(in), count, (uint16_t) bias); break;
	case 4: cuda::enqueue_launch(reinterpret_cast<const void*>(rb::rebase<4>), stream, config, static_cast<uint32_t*>(out), static_cast<const uint32_t*>(in), count, (uint32_t) bias); break;
	case 8: cuda::enqueue_launch(reinterpret_cast<const void*>(rb::rebase<8>), stream, config, static_cast<uint64_t*>(out), static_cast<const uint64_t*>(in), count, (uint64_t) bias); break;
	default: throw std::invalid_argument("entries of 1, 2, 4 or 8 bytes");
	}
}

// ---- the pipeline ----------------------------------------------------------------------------------------------------------

/// compressed bytes a GROUP of consecutive chunks moves at least, where the wrapper decodes batches: the group's slices cross the link as ONE copy per array and
/// are decoded by ONE launch with a piece per chunk (GIDDY_PIPELINE_GROUP_BYTES in the environment, read once; 0: a copy and a launch per chunk, as before round 5)
uint64_t group_bytes_wanted()
{
	static const uint64_t bytes = [] {
		const char* v = std::getenv("GIDDY_PIPELINE_GROUP_BYTES");
		return v ? std::strtoull(v, nullptr, 10) : (uint64_t) 4 << 20;
	}();
	return bytes;
}

enum class sliced_by { nothing, elements, segments, units, size_containers, whole_table };

/// how an array argument of a decompression wrapper is cut (by its name in the wrappers' argument maps — SURVEY.md section 8b)
sliced_by slicing_of(const std::string& name)
{
	static const std::map<std::string, sliced_by> table = {
		{ "compressed_input", sliced_by::elements },
		{ "anchor_values", sliced_by::segments }, { "segment_model_coefficients", sliced_by::segments },
		{ "position_anchors", sliced_by::segments }, { "intra_run_anchor_offsets", sliced_by::segments },
		{ "segment_position_anchors", sliced_by::segments }, { "segment_position_anchor_intra_run_offsets", sliced_by::segments },
		{ "compressed_data", sliced_by::units }, { "run_data", sliced_by::units }, { "run_lengths", sliced_by::units }, { "run_start_positions", sliced_by::units },
		{ "packed_element_sizes", sliced_by::size_containers },
		{ "dictionary_entries", sliced_by::whole_table },
	};
	auto found = table.find(name);
	return found == table.end() ? sliced_by::nothing : found->second;
}

unsigned itemsize_of(int type)
{
	switch (type) {
	case GIDDY_PTR_U8: case GIDDY_PTR_I8: case GIDDY_CPTR_U8: case GIDDY_CPTR_I8: return 1;
	case GIDDY_PTR_U16: case GIDDY_PTR_I16: case GIDDY_CPTR_U16: case GIDDY_CPTR_I16: case GIDDY_CPTR_COEF_I16X1: return 2;
	case GIDDY_PTR_U32: case GIDDY_PTR_I32: case GIDDY_CPTR_U32: case GIDDY_CPTR_I32: case GIDDY_CPTR_COEF_I16X2: case GIDDY_CPTR_COEF_I32X1: return 4;
	case GIDDY_PTR_U64: case GIDDY_PTR_I64: case GIDDY_CPTR_U64: case GIDDY_CPTR_I64: case GIDDY_CPTR_COEF_I32X2: return 8;
	default: return 0;   // not a pointer
	}
}

bool is_scalar_count(const std::string& name, const cuda::sharding::shard_t& s, uint64_t& value)
{
	if (name == "length" || name == "total_uncompressed_length" || name == "uncompressed_length") { value = s.length(); return true; }
	if (name == "num_segments" || name == "num_anchors") { value = s.segments.size(); return true; }
	if (name == "num_element_runs") { value = s.units.size(); return true; }
	return false;
}

void decode_from_host(const char* key, int scheme, const giddy_column_shape_t& shape, const giddy_argument_t* arguments, size_t num_arguments,
	uint64_t num_chunks, void* const* streams, size_t num_streams)
{
	if (key == nullptr || arguments == nullptr || streams == nullptr || num_streams == 0 || num_chunks == 0) { throw std::invalid_argument("null key / arguments / streams, or no chunks"); }
	if (shape.anchors_on_device) { throw std::invalid_argument("the pipeline cuts a column that is in HOST memory: position_anchors on the host"); }
	auto wrapper = cuda::registered::kernel_t::produceSubclass(key);
	const auto properties = cuda::device::current::get().properties();
	const auto attributes = wrapper->get_device_function().attributes();
	hipStream_t first_stream = static_cast<hipStream_t>(streams[0]);

	// what each argument is, worked out once
	struct role { sliced_by how = sliced_by::nothing; unsigned itemsize = 0; bool anchors = false, positions = false, output = false, count = false; std::string name; };
	std::vector<role> roles(num_arguments);
	uint64_t dictionary_entries = 0;
	for (size_t i = 0; i < num_arguments; i++) {
		role& r = roles[i];
		r.name = arguments[i].name ? arguments[i].name : "";
		r.itemsize = itemsize_of(arguments[i].type);
		r.how = r.itemsize ? slicing_of(r.name) : sliced_by::nothing;
		r.anchors = r.name == "position_anchors" || r.name == "segment_position_anchors";
		r.positions = r.name == "run_start_positions";
		r.output = r.name == "decompressed";
		if (r.name == "num_dictionary_entries") { dictionary_entries = arguments[i].value.u64; }
	}
	std::vector<cuda::sharding::shard_t> shards;
	for (uint64_t c = 0; c < num_chunks; c++) { shards.push_back(cut(scheme, shape, c, num_chunks)); }

	// The column's per-segment arrays (anchors, anchor values, model records: one entry per segment — small) and its side tables go to the
	// device ONCE, whole, on the first stream; the anchors are rebased there chunk by chunk in one launch; the other streams wait for that.
	// A chunk then costs a copy per BULK array and its launch — at 256 chunks the runtime's ~3 us per enqueued operation is what bounds the
	// pipeline, not the link.
	const uint64_t num_segments_total = shape.segment_length ? (shape.length + shape.segment_length - 1) / shape.segment_length : 0;
	size_t whole_bytes = 256;
	std::vector<size_t> whole_offset(num_arguments, 0);
	for (size_t i = 0; i < num_arguments; i++) {
		const role& r = roles[i];
		size_t bytes = 0;
		if (r.how == sliced_by::segments) { bytes = num_segments_total * r.itemsize; }
		if (r.how == sliced_by::whole_table) { bytes = dictionary_entries * r.itemsize; }
		whole_offset[i] = whole_bytes;
		whole_bytes += (bytes + 255) / 256 * 256 + 256;
	}
	const size_t table_offset = whole_bytes;
	whole_bytes += num_chunks * sizeof(cuda::kernels::data_layout::rebase::chunk_of_entries) + 256;
	// (leased until every stream's last launch of this call is enqueued and the first stream is ordered behind them — see the end of this
	// function; an exception on the way gives it back too)
	cuda::stream_scratch::lease whole_block(first_stream);
	if (!whole_block.take(whole_bytes)) { throw cuda::runtime_error(hipErrorOutOfMemory, "no device memory for a column's per-segment arrays"); }
	void* whole = whole_block.memory();
	char* whole_base = static_cast<char*>(whole);
	// (the chunk table travels from PINNED memory that outlives this call — an asynchronous copy out of pageable memory may still be reading
	// it when this function has returned: one buffer per host thread AND device — a thread that drives several devices records each
	// device's event on that device's stream —, portable, reused once the previous call's copy out of it is known to be done)
	using chunk_entry = cuda::kernels::data_layout::rebase::chunk_of_entries;
	struct pinned_table { chunk_entry* memory = nullptr; size_t capacity = 0; hipEvent_t copied = nullptr; };
	static thread_local std::map<int, pinned_table> pinned_tables;
	pinned_table& pinned = pinned_tables[cuda::device::current::get().id()];
	if (pinned.copied != nullptr) { cuda::throw_if_error(hipEventSynchronize(pinned.copied), "hipEventSynchronize"); }
	else { cuda::throw_if_error(hipEventCreateWithFlags(&pinned.copied, hipEventDisableTiming), "hipEventCreate"); }
	if (pinned.capacity < num_chunks) {
		if (pinned.memory != nullptr) { (void) hipHostFree(pinned.memory); pinned.memory = nullptr; pinned.capacity = 0; }
		const size_t capacity = std::max<size_t>(2 * num_chunks, 1024);
		cuda::throw_if_error(hipHostMalloc(reinterpret_cast<void**>(&pinned.memory), capacity * sizeof(chunk_entry), hipHostMallocPortable), "hipHostMalloc (chunk table)");
		pinned.capacity = capacity;
	}
	chunk_entry* table = pinned.memory;
	for (uint64_t c = 0; c < num_chunks; c++) { table[c] = chunk_entry { shards[c].segments.first, shards[c].anchor_bias }; }
	bool any_bias = false;
	for (const auto& s : shards) { any_bias = any_bias || s.anchor_bias != 0; }
	for (size_t i = 0; i < num_arguments; i++) {
		const role& r = roles[i];
		if (r.how != sliced_by::segments && r.how != sliced_by::whole_table) { continue; }
		const size_t bytes = (r.how == sliced_by::segments ? num_segments_total : dictionary_entries) * r.itemsize;
		if (bytes) { cuda::throw_if_error(hipMemcpyAsync(whole_base + whole_offset[i], arguments[i].value.pointer, bytes, hipMemcpyHostToDevice, first_stream), "hipMemcpyAsync (a per-segment array)"); }
		if (r.anchors && any_bias && num_segments_total) {
			cuda::throw_if_error(hipMemcpyAsync(whole_base + table_offset, table, num_chunks * sizeof(chunk_entry), hipMemcpyHostToDevice, first_stream), "hipMemcpyAsync (chunk table)");
			namespace rb = cuda::kernels::data_layout::rebase;
			const cuda::launch_configuration_t config((cuda::grid_dimension_t) std::min<uint64_t>((num_segments_total + 255) / 256, 4096), 256);
			auto* chunks = reinterpret_cast<const rb::chunk_of_entries*>(whole_base + table_offset);
			void* entries = whole_base + whole_offset[i];
			switch (r.itemsize) {
			case 1: cuda::enqueue_launch(reinterpret_cast<const void*>(rb::rebase_by_chunks<1>), first_stream, config, static_cast<uint8_t*>(entries), chunks, num_chunks, num_segments_total); break;
			case 2: cuda::enqueue_launch(reinterpret_cast<const void*>(rb::rebase_by_chunks<2>), first_stream, config, static_cast<uint16_t*>(entries), chunks, num_chunks, num_segments_total); break;
			case 4: cuda::enqueue_launch(reinterpret_cast<const void*>(rb::rebase_by_chunks<4>), first_stream, config, static_cast<uint32_t*>(entries), chunks, num_chunks, num_segments_total); break;
			default: cuda::enqueue_launch(reinterpret_cast<const void*>(rb::rebase_by_chunks<8>), first_stream, config, static_cast<uint64_t*>(entries), chunks, num_chunks, num_segments_total); break;
			}
		}
	}
	cuda::throw_if_error(hipEventRecord(pinned.copied, first_stream), "hipEventRecord");

	// GROUPS of consecutive chunks.  A chunk used to cost a copy per bulk array and a launch — 3-4 us of host time per enqueued operation, so a run-length
	// column (ratio 21) cut into 256 / 1024 chunks decoded at 253 / 78 GB/s where 16 chunks reach 830 (profiles/r04_pcie_pipeline.txt).  The slices of
	// consecutive chunks are consecutive in the column's arrays (a run that straddles a cut belongs to both sides: the ranges overlap by it), so a group of
	// G chunks crosses the link as ONE copy per array and is decoded by ONE batch launch with a piece per chunk (giddy/batch.hpp) — the cuts, the rebased
	// anchors and every chunk's own launch configuration stay what they were.  G: enough chunks for `group_bytes_wanted()` compressed bytes, while there are
	// at least four groups per stream to overlap.  Absolute run positions are rebased per chunk by a launch of their own: those columns keep G = 1.
	uint64_t bulk_bytes = 0;
	bool rebases_positions = false;
	for (size_t i = 0; i < num_arguments; i++) {
		const role& r = roles[i];
		if (r.how == sliced_by::elements) { bulk_bytes += shape.length * r.itemsize; }
		if (r.how == sliced_by::units) { bulk_bytes += shape.num_units * r.itemsize; }
		if (r.how == sliced_by::size_containers) { bulk_bytes += shards.back().size_containers.end * r.itemsize; }
	}
	for (const auto& s : shards) { rebases_positions = rebases_positions || s.position_bias != 0; }
	uint64_t chunks_per_group = 1;
	if (wrapper->launches_batches() && !rebases_positions && group_bytes_wanted() != 0 && num_chunks > 1) {
		const uint64_t per_chunk = std::max<uint64_t>(1, bulk_bytes / num_chunks);
		chunks_per_group = std::max<uint64_t>(1, std::min<uint64_t>(group_bytes_wanted() / per_chunk, num_chunks / std::max<uint64_t>(1, 4 * num_streams)));
	}
	const uint64_t num_groups = (num_chunks + chunks_per_group - 1) / chunks_per_group;
	// a group's slices: [the first chunk's first entry, the last chunk's end) of every bulk array, each 256-byte aligned within ONE block per stream, leased for
	// the whole call — the groups of a stream follow each other through it in stream order.  (The wrapper's own scratch — chained Delta tiles, finer anchors
	// of long segments — and a group's batch tables are other blocks of the pool: none waits for another.)
	auto range_of = [&](const role& r, uint64_t first_chunk, uint64_t end_chunk) {
		auto slice_of = [&](const cuda::sharding::shard_t& shard) -> const cuda::sharding::slice_t& {
			return r.how == sliced_by::elements ? shard.elements : r.how == sliced_by::units ? shard.units : shard.size_containers;
		};
		uint64_t first = ~(uint64_t) 0, end = 0;
		for (uint64_t c = first_chunk; c < end_chunk; c++) {
			if (shards[c].length() == 0) { continue; }
			first = std::min<uint64_t>(first, slice_of(shards[c]).first); end = std::max<uint64_t>(end, slice_of(shards[c]).end);
		}
		return first > end ? cuda::sharding::slice_t { 0, 0 } : cuda::sharding::slice_t { first, end };
	};
	auto slices_of = [&](uint64_t first_chunk, uint64_t end_chunk, std::vector<size_t>& offset) {
		size_t total = 0;
		for (size_t i = 0; i < num_arguments; i++) {
			const role& r = roles[i];
			if (r.how != sliced_by::elements && r.how != sliced_by::units && r.how != sliced_by::size_containers) { continue; }
			offset[i] = total;
			total += (range_of(r, first_chunk, end_chunk).size() * r.itemsize + 255) / 256 * 256 + 256;   // (+ slack: the decoders' 16-byte loads may read up to the end of a slice's last chunk)
		}
		return total;
	};
	std::vector<size_t> most(num_streams, 256);
	{
		std::vector<size_t> offset(num_arguments, 0);
		for (uint64_t g = 0; g < num_groups; g++) {
			most[g % num_streams] = std::max(most[g % num_streams], slices_of(g * chunks_per_group, std::min(num_chunks, (g + 1) * chunks_per_group), offset));
		}
	}
	std::vector<std::unique_ptr<cuda::stream_scratch::lease>> blocks;
	for (size_t k = 0; k < num_streams && k < num_groups; k++) {
		blocks.emplace_back(new cuda::stream_scratch::lease(static_cast<hipStream_t>(streams[k])));
		if (!blocks.back()->take(most[k])) { throw cuda::runtime_error(hipErrorOutOfMemory, "no device memory for a chunk's slices"); }
	}
	// Every chunk's launch arguments, in the order of the chunks: the wrapper's argument map and the resolver's are built ONCE, a chunk rewrites the entries
	// that differ in place (no allocation per chunk), and the launch configuration is resolved again only when the chunk's length differs from the
	// previous chunk's.  `with_chunk(config, map)`: what to do with them.
	std::vector<std::vector<size_t>> offsets_of_group(num_groups, std::vector<size_t>(num_arguments, 0));
	std::vector<std::vector<uint64_t>> first_of_group(num_groups, std::vector<uint64_t>(num_arguments, 0));
	for (uint64_t g = 0; g < num_groups; g++) {
		const uint64_t first_chunk = g * chunks_per_group, end_chunk = std::min(num_chunks, (g + 1) * chunks_per_group);
		(void) slices_of(first_chunk, end_chunk, offsets_of_group[g]);
		for (size_t i = 0; i < num_arguments; i++) {
			const role& r = roles[i];
			if (r.how == sliced_by::elements || r.how == sliced_by::units || r.how == sliced_by::size_containers) { first_of_group[g][i] = range_of(r, first_chunk, end_chunk).first; }
		}
	}
	struct chunk_walk {
		std::vector<giddy_argument_t> chunk_arguments;
		cuda::kernel_t::arguments_type map, resolution;
		std::vector<cuda::any*> entry;
		cuda::launch_configuration_t config;
		uint64_t config_for_length = ~(uint64_t) 0;
	};
	chunk_walk walk;
	walk.chunk_arguments.assign(arguments, arguments + num_arguments);
	walk.map = giddy_internal::to_map(arguments, num_arguments);
	walk.entry.assign(num_arguments, nullptr);
	for (size_t i = 0; i < num_arguments; i++) { walk.entry[i] = &walk.map.at(roles[i].name); }
	// (launch-configuration resolution reads its sizes as size_t, under the names the wrappers' resolvers use)
	walk.resolution = walk.map;
	for (size_t i = 0; i < num_arguments; i++) {
		if (roles[i].itemsize == 0 && (roles[i].name == "segment_length" || roles[i].name == "num_dictionary_entries" || roles[i].name == "anchoring_period")) {
			walk.resolution[roles[i].name] = cuda::any((size_t) arguments[i].value.u64);
		}
	}
	auto arguments_of_chunk = [&](uint64_t g, uint64_t c) {   // -> walk.map / walk.config describe chunk c (of group g)
		const auto& shard = shards[c];
		char* const base = static_cast<char*>(blocks[g % num_streams]->memory());
		for (size_t i = 0; i < num_arguments; i++) {
			const role& r = roles[i];
			giddy_argument_t& a = walk.chunk_arguments[i];
			bool changed = false;
			if (r.how == sliced_by::elements || r.how == sliced_by::units || r.how == sliced_by::size_containers) {
				const auto& slice = r.how == sliced_by::elements ? shard.elements : r.how == sliced_by::units ? shard.units : shard.size_containers;
				a.value.pointer = base + offsets_of_group[g][i] + (slice.first - first_of_group[g][i]) * r.itemsize; changed = true;
			}
			else if (r.how == sliced_by::segments) { a.value.pointer = whole_base + whole_offset[i] + shard.segments.first * r.itemsize; changed = true; }
			else if (r.how == sliced_by::whole_table) { a.value.pointer = whole_base + whole_offset[i]; changed = true; }
			else if (r.output) { a.value.pointer = static_cast<char*>(arguments[i].value.pointer) + shard.elements.first * r.itemsize; changed = true; }
			else if (r.itemsize == 0) { uint64_t count; if (is_scalar_count(r.name, shard, count)) { a.value.u64 = count; changed = true; } }
			if (changed) { *walk.entry[i] = giddy_internal::to_any(a); }
		}
		if (shard.length() != walk.config_for_length) {
			walk.resolution["length"] = cuda::any((size_t) shard.length());
			walk.resolution["data_length"] = cuda::any((size_t) shard.length());
			walk.config = wrapper->resolve_launch_configuration(properties, attributes, walk.resolution, cuda::launch_configuration_limits_t { cuda::nullopt, cuda::nullopt, cuda::nullopt });
			walk.config_for_length = shard.length();
		}
	};
	// Groups of several chunks: every group's batch tables are built now and cross the link as ONE copy, with the per-segment arrays (a copy per group would
	// sit between the groups' bulk copies on the link: ~9 us each, measured) — a group then costs a copy per bulk array and its launch.
	std::vector<cuda::batch::on_device_t> tables_of_group(num_groups);
	cuda::stream_scratch::lease tables_block(first_stream);
	if (chunks_per_group > 1) {
		cuda::batch::table_t all;          // the groups' tables one after the other: pieces and blocks keep their own numbering within each group
		std::vector<cuda::batch::table_t> tables(num_groups);
		size_t bytes = 0;
		std::vector<size_t> at(num_groups, 0);
		for (uint64_t g = 0; g < num_groups; g++) {
			for (uint64_t c = g * chunks_per_group; c < std::min(num_chunks, (g + 1) * chunks_per_group); c++) {
				if (shards[c].length() == 0) { continue; }
				arguments_of_chunk(g, c);
				wrapper->add_to_batch(tables[g], walk.config, walk.map);
			}
			at[g] = bytes;
			bytes += (cuda::batch::device_bytes_of(tables[g]) + 255) / 256 * 256;
		}
		if (!tables_block.take(std::max<size_t>(bytes, 256))) { throw cuda::runtime_error(hipErrorOutOfMemory, "no device memory for the groups' batch tables"); }
		cuda::batch::staging_pool::block* staging = cuda::batch::staging_pool::instance().take(std::max<size_t>(bytes, 256));
		for (uint64_t g = 0; g < num_groups; g++) {
			tables_of_group[g] = cuda::batch::place(static_cast<char*>(staging->memory) + at[g], static_cast<char*>(tables_block.memory()) + at[g], tables[g]);
		}
		const hipError_t status = bytes ? hipMemcpyAsync(tables_block.memory(), staging->memory, bytes, hipMemcpyHostToDevice, first_stream) : hipSuccess;
		cuda::batch::staging_pool::instance().copy_enqueued(staging, first_stream);
		cuda::throw_if_error(status, "hipMemcpyAsync (the groups' batch tables)");
	}
	ordering_event uploaded;
	uploaded.record(first_stream);
	for (size_t k = 1; k < num_streams; k++) { uploaded.make_wait(static_cast<hipStream_t>(streams[k])); }

	for (uint64_t g = 0; g < num_groups; g++) {
		hipStream_t stream = static_cast<hipStream_t>(streams[g % num_streams]);
		char* const base = static_cast<char*>(blocks[g % num_streams]->memory());
		const uint64_t first_chunk = g * chunks_per_group, end_chunk = std::min(num_chunks, (g + 1) * chunks_per_group);
		// the group's slices of the bulk arrays: one copy each
		for (size_t i = 0; i < num_arguments; i++) {
			const role& r = roles[i];
			if (r.how != sliced_by::elements && r.how != sliced_by::units && r.how != sliced_by::size_containers) { continue; }
			const auto range = range_of(r, first_chunk, end_chunk);
			const size_t bytes = range.size() * r.itemsize;
			if (bytes) { cuda::throw_if_error(hipMemcpyAsync(base + offsets_of_group[g][i], static_cast<const char*>(arguments[i].value.pointer) + range.first * r.itemsize, bytes, hipMemcpyHostToDevice, stream), "hipMemcpyAsync (a group's slice)"); }
		}
		if (chunks_per_group > 1) {
			if (tables_of_group[g].num_blocks != 0) { wrapper->enqueue_launch_of_batch(stream, tables_of_group[g]); }
			continue;
		}
		const auto& shard = shards[first_chunk];   // (a group of one chunk: a launch of its own)
		if (shard.length() == 0) { continue; }
		arguments_of_chunk(g, first_chunk);
		for (size_t i = 0; i < num_arguments; i++) {   // (absolute run positions count from the column's first output: rebased where they have just arrived)
			if (roles[i].positions && shard.position_bias != 0) { enqueue_rebase(stream, walk.chunk_arguments[i].value.pointer, walk.chunk_arguments[i].value.pointer, shard.units.size(), roles[i].itemsize, shard.position_bias); }
		}
		wrapper->enqueue_launch_with(stream, walk.config, walk.map);
	}
	// the first stream runs on only when every stream has made its last launch of this call: whatever is enqueued on it next — the next
	// call's uploads into the same kept block among them — comes after all of this call's readers of the per-segment arrays
	for (size_t k = 1; k < num_streams; k++) {
		ordering_event done;
		done.record(static_cast<hipStream_t>(streams[k]));
		done.make_wait(first_stream);
	}
}

} // namespace

extern "C" {

int giddy_shard(int scheme, const giddy_column_shape_t* shape, uint64_t rank, uint64_t world_size, giddy_shard_t* shard)
{
	return guarded([&] {
		if (shape == nullptr || shard == nullptr) { throw std::invalid_argument("null shape or output shard"); }
		const auto s = cut(scheme, *shape, rank, world_size);
		*shard = giddy_shard_t { s.segments.first, s.segments.end, s.elements.first, s.elements.end, s.units.first, s.units.end,
			s.size_containers.first, s.size_containers.end, s.anchor_bias, s.position_bias };
	});
}

int giddy_enqueue_rebase(void* hip_stream, void* out, const void* in, uint64_t count, uint32_t itemsize, uint64_t bias)
{
	return guarded([&] {
		if (count != 0 && (out == nullptr || in == nullptr)) { throw std::invalid_argument("null array"); }
		enqueue_rebase(static_cast<hipStream_t>(hip_stream), out, in, count, itemsize, bias);
	});
}

int giddy_enqueue_gather_on_node(void* destination, int root_device, void* root_stream, const void* const* shards, const int* devices,
	void* const* hip_streams, const uint64_t* bytes, size_t num_shards)
{
	return guarded([&] {
		if (num_shards != 0 && (destination == nullptr || shards == nullptr || devices == nullptr || hip_streams == nullptr || bytes == nullptr)) { throw std::invalid_argument("null array"); }
		int current = 0;
		cuda::throw_if_error(hipGetDevice(&current), "hipGetDevice");
		char* at = static_cast<char*>(destination);
		for (size_t i = 0; i < num_shards; i++) {
			if (bytes[i] != 0) {
				if (shards[i] == nullptr) { throw std::invalid_argument("null shard"); }
				enable_peer_access(devices[i], root_device);
				cuda::throw_if_error(hipSetDevice(devices[i]), "hipSetDevice (a shard's device)");
				const auto stream = static_cast<hipStream_t>(hip_streams[i]);
				cuda::throw_if_error(hipMemcpyPeerAsync(at, root_device, shards[i], devices[i], bytes[i], stream), "hipMemcpyPeerAsync (a decoded shard to the root)");
				// the root's stream waits for this copy
				hipEvent_t copied;
				cuda::throw_if_error(hipEventCreateWithFlags(&copied, hipEventDisableTiming), "hipEventCreateWithFlags");
				hipError_t status = hipEventRecord(copied, stream);
				if (status == hipSuccess) { status = hipStreamWaitEvent(static_cast<hipStream_t>(root_stream), copied, 0); }
				(void) hipEventDestroy(copied);   // (released once the record has completed)
				cuda::throw_if_error(status, "ordering the root's stream behind a shard's copy");
			}
			at += bytes[i];
		}
		cuda::throw_if_error(hipSetDevice(current), "hipSetDevice (back)");
	});
}

int giddy_enqueue_gather_compressed_on_node(const char* key, int root_device, void* const* root_streams, size_t num_root_streams,
	const giddy_compressed_shard_t* shards, size_t num_shards)
{
	return guarded([&] {
		if (key == nullptr || root_streams == nullptr || num_root_streams == 0 || (num_shards != 0 && shards == nullptr)) { throw std::invalid_argument("null key / streams / shards"); }
		auto wrapper = cuda::registered::kernel_t::produceSubclass(key);
		cuda::device::current::scoped_override_t<> on_root(root_device);
		const auto properties = cuda::device::get(root_device).properties();
		const auto attributes = wrapper->get_device_function().attributes();
		for (size_t i = 0; i < num_shards; i++) {
			const giddy_compressed_shard_t& shard = shards[i];
			if (shard.num_arguments != 0 && (shard.arguments == nullptr || shard.bytes == nullptr)) { throw std::invalid_argument("a shard without arguments / sizes"); }
			const auto root_stream = static_cast<hipStream_t>(root_streams[i % num_root_streams]);
			// (GIDDY_SHARD_DECODE_IN_PLACE: no copy at all — the root's decoder reads the shard's arrays where they lie, over the link, as it reads
			// pinned host memory in libgiddy_amd.pipeline.decode_directly_from_pinned_host: peer access makes a peer's memory addressable)
			const bool in_place = (shard.flags & GIDDY_SHARD_DECODE_IN_PLACE) != 0;
			if (in_place && !enable_peer_access(root_device, shard.device)) {   // (a kernel cannot read a peer's memory through the host)
				throw cuda::runtime_error(hipErrorPeerAccessUnsupported, "GIDDY_SHARD_DECODE_IN_PLACE: device " + std::to_string(root_device) + " cannot access device " + std::to_string(shard.device) + " directly");
			}
			const bool moves = !in_place && (shard.device != root_device || (shard.flags & GIDDY_SHARD_COPY_EVEN_IF_LOCAL) != 0);
			// the root's stream waits for what the shard's device has enqueued on the shard's stream so far (its arrays being written, say)
			{
				cuda::device::current::scoped_override_t<> on_shard(shard.device);
				ordering_event ready;
				ready.record(static_cast<hipStream_t>(shard.hip_stream));
				ready.make_wait(root_stream);
			}
			std::vector<giddy_argument_t> arguments(shard.arguments, shard.arguments + shard.num_arguments);
			// the compressed arrays cross their link ONCE, into scratch of the root's stream (the next shard on this stream gets the same block,
			// in stream order), and are decoded from there
			cuda::stream_scratch::lease block(root_stream);
			if (moves) {
				enable_peer_access(root_device, shard.device);
				size_t total = 0;
				std::vector<size_t> offset(shard.num_arguments, 0);
				for (size_t a = 0; a < shard.num_arguments; a++) {
					if (shard.bytes[a] == 0) { continue; }
					offset[a] = total;
					total += (size_t) (shard.bytes[a] + 255) / 256 * 256 + 256;   // (+ slack: the decoders' 16-byte loads may read up to the end of an array's last chunk)
				}
				if (total != 0 && !block.take(total)) { throw cuda::runtime_error(hipErrorOutOfMemory, "no device memory on the root for a compressed shard"); }
				for (size_t a = 0; a < shard.num_arguments; a++) {
					if (shard.bytes[a] == 0) { continue; }
					char* at = static_cast<char*>(block.memory()) + offset[a];
					cuda::throw_if_error(hipMemcpyPeerAsync(at, root_device, arguments[a].value.pointer, shard.device, (size_t) shard.bytes[a], root_stream), "hipMemcpyPeerAsync (a compressed array to the root)");
					arguments[a].value.pointer = at;
				}
			}
			const auto map = giddy_internal::to_map(arguments.data(), arguments.size());
			const auto resolution = giddy_internal::to_map(shard.resolution_arguments, shard.num_resolution_arguments);
			const auto config = wrapper->resolve_launch_configuration(properties, attributes, resolution, cuda::launch_configuration_limits_t { cuda::nullopt, cuda::nullopt, cuda::nullopt });
			wrapper->enqueue_launch_with(root_stream, config, map);
		}
		// the first root stream runs on only when the others have made their last launch: work enqueued on it afterwards sees the whole column
		for (size_t k = 1; k < num_root_streams && k < num_shards; k++) {
			ordering_event done;
			done.record(static_cast<hipStream_t>(root_streams[k]));
			done.make_wait(static_cast<hipStream_t>(root_streams[0]));
		}
	});
}

int giddy_decode_from_host(const char* key, int scheme, const giddy_column_shape_t* shape, const giddy_argument_t* arguments, size_t num_arguments,
	uint64_t num_chunks, void* const* hip_streams, size_t num_streams)
{
	return guarded([&] {
		if (shape == nullptr) { throw std::invalid_argument("null shape"); }
		decode_from_host(key, scheme, *shape, arguments, num_arguments, num_chunks, hip_streams, num_streams);
	});
}

} // extern "C"

// Beyond the reference's three modes of use: MANY small columns decoded by ONE launch (giddy/batch.hpp).  The reference launches
// a kernel per column (src/cuda/kernel_wrapper.cuh:135-147); chunks that have just crossed a link are small, and a launch per
// chunk of 2^16 elements runs at 1 % of the rate of a 2^28-element column on MI355X.  Here: 300 dictionary-encoded columns of 0 ... 70 000
// elements each — the factory's type-erased wrapper as in mode A — added to one batch table, uploaded with one copy, decoded with one
// launch, every element checked on the host.
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <vector>

#include "cuda/api_wrappers.h"
#include "kernel_wrappers/registered_wrapper.h"
#include "giddy/batch.hpp"

using wrapper = cuda::registered::kernel_t;

int main()
{
	const char* const key = "decompression::dictionary::kernel_t<4u, 4u, 1u>";
	if (!wrapper::canProduce(key)) { std::cerr << "the library does not register " << key << "\n"; return EXIT_FAILURE; }
	auto kernel = wrapper::produceSubclass(key);
	if (!kernel->launches_batches()) { std::cerr << key << " does not launch batches\n"; return EXIT_FAILURE; }

	// the columns, one after the other in two big arrays (any placement would do: a piece is a set of pointers)
	const unsigned num_columns = 300;
	const uint8_t num_dictionary_entries = 200;
	std::vector<uint32_t> lengths(num_columns), starts(num_columns);
	uint32_t total = 0;
	for (unsigned c = 0; c < num_columns; c++) { starts[c] = total; lengths[c] = c == 0 ? 0 : c == 1 ? 1 : (c * 2654435761u) % 70001u; total += (lengths[c] + 15) / 16 * 16; }
	std::vector<uint8_t> h_keys(total);
	for (uint32_t i = 0; i < total; i++) { h_keys[i] = (uint8_t) ((i * 40503u + (i >> 7)) % num_dictionary_entries); }
	std::vector<uint32_t> h_dictionary(num_dictionary_entries);
	for (unsigned e = 0; e < num_dictionary_entries; e++) { h_dictionary[e] = 1000003u * e + 7u; }

	auto device = cuda::device::current::get();
	auto d_keys = cuda::memory::device::make_unique<uint8_t[]>(device, total);
	auto d_dictionary = cuda::memory::device::make_unique<uint32_t[]>(device, num_dictionary_entries);
	auto d_decompressed = cuda::memory::device::make_unique<uint32_t[]>(device, total);
	cuda::memory::copy(d_keys.get(), h_keys.data(), total);
	cuda::memory::copy(d_dictionary.get(), h_dictionary.data(), num_dictionary_entries * sizeof(uint32_t));
	{ const std::vector<uint32_t> zeros(total, 0u); cuda::memory::copy(d_decompressed.get(), zeros.data(), total * sizeof(uint32_t)); }

	// a piece per column: its own launch configuration (as if it were launched alone) and its own arguments — the names of mode A
	cuda::batch::table_t table;
	for (unsigned c = 0; c < num_columns; c++) {
		wrapper::arguments_type sizes;
		sizes["data_length"] = (size_t) lengths[c];
		sizes["num_dictionary_entries"] = (size_t) num_dictionary_entries;
		const auto configuration = kernel->resolve_launch_configuration(device.properties(), kernel->get_device_function().attributes(), sizes);
		uint32_t* decompressed = d_decompressed.get() + starts[c];
		const uint8_t* compressed_input = d_keys.get() + starts[c];
		const uint32_t* dictionary_entries = d_dictionary.get();
		uint32_t length = lengths[c];
		uint32_t num_entries = num_dictionary_entries;   // (size_type_by_index_size<1>: indices narrower than a register are promoted)
		wrapper::arguments_type piece = MAKE_KERNEL_ARGUMENTS(decompressed, compressed_input, dictionary_entries, length);
		piece["num_dictionary_entries"] = num_entries;
		kernel->add_to_batch(table, configuration, piece);
	}

	const auto stream = cuda::stream::default_stream_id;
	cuda::batch::uploaded_t on_device(table, stream);            // both tables: ONE asynchronous copy
	kernel->enqueue_launch_of_batch(stream, on_device.view());   // ONE launch for the 300 columns
	device.synchronize();

	std::vector<uint32_t> h_decompressed(total);
	cuda::memory::copy(h_decompressed.data(), d_decompressed.get(), total * sizeof(uint32_t));
	for (unsigned c = 0; c < num_columns; c++) {
		for (uint32_t i = 0; i < (lengths[c] + 15) / 16 * 16; i++) {
			const uint32_t expected = i < lengths[c] ? h_dictionary[h_keys[starts[c] + i]] : 0u;   // (nothing is written past a column's last element)
			if (h_decompressed[starts[c] + i] != expected) {
				std::cerr << "column " << c << " (" << lengths[c] << " elements), element " << i << ": got " << h_decompressed[starts[c] + i] << " instead of " << expected << "\n";
				return EXIT_FAILURE;
			}
		}
	}
	std::cout << "OK: " << num_columns << " columns, " << table.pieces().size() << " pieces and " << table.blocks().size() << " workgroups in one launch\n";
	return EXIT_SUCCESS;
}

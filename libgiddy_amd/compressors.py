"""Device-side compressors: columns that are resident on the GPU put into the formats the decompression kernels read,
without a host pass (an extension — the reference ships decompressors only, README.md:128-130 "Issue #3").

Same signatures and the same output dictionaries as the numpy writers of libgiddy_amd.formats — the parity target: these
functions' outputs equal that module's byte for byte (tests/test_gpu_compressors.py) — but inputs and outputs are torch
tensors on the device, and all computing is done by the library's HIP kernels through the C ABI (factory keys
"compression::<scheme>::kernel_t<...>", include/giddy/kernel_wrappers/compression/compressors.hpp).  torch allocates the
output and scratch buffers and reads back the few scalars that size them (bytes, runs, patches) — plumbing only.
Tensors stand for the unsigned bits they hold (torch has few unsigned dtypes): a uint64 column is an int64 tensor.
"""
import numpy as np
import torch

from . import registered as r
from . import schemes

_SIGNED = {1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}
_NAMES = {1: "signed char", 2: "short", 4: "int", 8: "long"}
TILE = 4096                                   # elements per tile of the variable-rate compressors (compression/common.hpp)


class Unrepresentable(ValueError):
    """some elements do not fit the requested widths (formats.py raises ValueError for the same columns)"""


def _size(x):
    return np.dtype(x).itemsize if not isinstance(x, int) else x


def _stream():
    return torch.cuda.current_stream().cuda_stream


LAST_SCRATCH = None   # (the last scratch block handed to a kernel: development probes look at what a debug build left there)


def _scratch(n, device):
    global LAST_SCRATCH
    LAST_SCRATCH = torch.empty(2 * (-(-n // TILE) + 2) + 2, dtype=torch.int64, device=device)
    return LAST_SCRATCH


LAST_RUN_SCRATCH = None
STAGED_PER_WAVE = 512   # runs per 4096 elements the run compressors' scratch has room for (a wavefront of more is read a second time); a multiple of 16, <= 512


def _run_scratch(n, usize, device, chain, staged_per_wave=None):
    """"run_scratch" / "run_scratch_entries" of the run compressors (kernel_wrappers/compression/compressors.hpp, run_scratch_entries()): with it
    they take three launches and no workgroup waits for another (kernels/compression/listed_runs.hpp); a caller who sets `chain` gets the one launch."""
    if chain:
        return {}
    staged = STAGED_PER_WAVE if staged_per_wave is None else staged_per_wave
    waves = -(-n // 4096)
    even = (waves + 1) // 2 * 2
    blocks = -(-waves // 1024)
    entries = 4 * even + 6 * ((blocks + 1) // 2 * 2) + (waves * staged * 2 + 15) // 16 * 2 + (waves * staged * usize + 15) // 16 * 2
    global LAST_RUN_SCRATCH
    LAST_RUN_SCRATCH = torch.empty(entries, dtype=torch.int64, device=device)
    return {"run_scratch": r.pointer(LAST_RUN_SCRATCH, "u64"), "run_scratch_entries": r.size_t(entries)}


def _launch(key, resolve, launch, counter=None, what="elements"):
    kernel = schemes.kernel(key)
    config = kernel.resolve_launch_configuration(r.CURRENT_DEVICE, resolve)
    kernel.enqueue_launch(_stream(), config, launch)
    if counter is not None:
        bad = int(counter[0])
        if bad:
            raise Unrepresentable("%d %s do not fit the requested widths" % (bad, what))


def _counter(device):
    return torch.zeros(1, dtype=torch.int32, device=device)


def delta(column, segment_length, compressed_dtype=np.uint16, index_size=4):
    n, usize, csize = column.numel(), column.element_size(), _size(compressed_dtype)
    out = {"compressed_input": torch.empty(n, dtype=_SIGNED[csize], device=column.device),
           "anchor_values": torch.empty(-(-n // segment_length), dtype=column.dtype, device=column.device), "length": n, "segment_length": segment_length}
    counter = _counter(column.device)
    if n:
        _launch("compression::delta::kernel_t<%du, %du, %du>" % (index_size, usize, csize),
                {"length": r.size_t(n), "segment_length": r.size_t(segment_length)},
                {"compressed_input": r.pointer(out["compressed_input"], "u%d" % (8 * csize)), "anchor_values": r.pointer(out["anchor_values"], "u%d" % (8 * usize)),
                 "input": r.const_pointer(column, "u%d" % (8 * usize)), "length": r.uint_t(index_size)(n), "segment_length": r.uint_t(index_size)(segment_length),
                 "unrepresentable_count": r.pointer(counter, "u32")}, counter, "deltas")
    return out


def discard_zero_bytes_fixed(column, compressed_dtype, index_size=4):
    n, usize, csize = column.numel(), column.element_size(), _size(compressed_dtype)
    out = {"compressed_input": torch.empty(n, dtype=_SIGNED[csize], device=column.device), "length": n}
    counter = _counter(column.device)
    if n:
        _launch("compression::discard_zero_bytes::fixed_width::kernel_t<%du, %du, %du>" % (index_size, usize, csize), {"length": r.size_t(n)},
                {"compressed_input": r.pointer(out["compressed_input"], "u%d" % (8 * csize)), "input": r.const_pointer(column, "u%d" % (8 * usize)),
                 "length": r.uint_t(index_size)(n), "unrepresentable_count": r.pointer(counter, "u32")}, counter)
    return out


def discard_zero_bytes_variable(column, segment_length, min_represented_element_size=1, bits_per_element_size=2, container_dtype=np.uint32, index_size=4):
    n, usize, csize = column.numel(), column.element_size(), _size(container_dtype)
    per = 8 * csize // bits_per_element_size
    device = column.device
    data = torch.empty(max(n * min(usize, min_represented_element_size + (1 << bits_per_element_size) - 1), 1), dtype=torch.uint8, device=device)
    out = {"packed_element_sizes": torch.empty(-(-n // per), dtype=_SIGNED[csize], device=device),
           "position_anchors": torch.empty(-(-n // segment_length), dtype=_SIGNED[index_size], device=device), "segment_length": segment_length, "length": n,
           "min_represented_element_size": min_represented_element_size, "bits_per_element_size": bits_per_element_size}
    counter, total = _counter(device), torch.zeros(1, dtype=torch.int64, device=device)
    scratch = torch.empty(-(-n // (256 * per)) + 4, dtype=torch.int64, device=device)
    if n:
        _launch("compression::discard_zero_bytes::variable_width::kernel_t<%du, %du, %du>" % (index_size, usize, csize),
                {"length": r.size_t(n), "segment_length": r.size_t(segment_length)},
                {"compressed_data": r.pointer(data, "u8"), "packed_element_sizes": r.pointer(out["packed_element_sizes"], "u%d" % (8 * csize)),
                 "position_anchors": r.pointer(out["position_anchors"], "u%d" % (8 * index_size)), "input": r.const_pointer(column, "u%d" % (8 * usize)),
                 "length": r.uint_t(index_size)(n), "segment_length": r.uint_t(index_size)(segment_length),
                 "min_represented_element_size": r.uint_t(4)(min_represented_element_size), "bits_per_element_size": r.uint_t(4)(bits_per_element_size),
                 "tile_totals": r.pointer(scratch, "u64"), "total_bytes": r.pointer(total, "u64"), "unrepresentable_count": r.pointer(counter, "u32")}, counter)
    compressed_bytes = int(total[0])
    if compressed_bytes == -1:   # (2^64 - 1 in the device scalar: kernels/compression/chained.hpp)
        raise ChainBroken("the chain of tiles broke: a tile was not heard of within its patience")
    out["compressed_data"] = data[:compressed_bytes]
    return out


def frame_of_reference(column, segment_length, compressed_dtype=np.int16, model="constant", index_size=4):
    """a constant model per segment (the midpoint of the segment's extremes) or a fitted line (exact least-squares slope, rounded; the
    midpoint of the residuals); `column`: a SIGNED tensor (int16 / int32)"""
    if model not in ("constant", "linear"):
        raise ValueError(model)
    n, usize, csize = column.numel(), column.element_size(), _size(compressed_dtype)
    nseg = -(-n // segment_length)
    dimension = 1 if model == "constant" else 2
    out = {"compressed_input": torch.empty(n, dtype=_SIGNED[csize], device=column.device),
           "segment_model_coefficients": torch.empty((nseg, dimension), dtype=torch.int32, device=column.device), "length": n, "segment_length": segment_length}
    counter = _counter(column.device)
    if n:
        _launch("compression::frame_of_reference::%skernel_t<%du, %s, %s>" % ("" if model == "constant" else "linear::", index_size, _NAMES[usize], _NAMES[csize]),
                {"length": r.size_t(n), "segment_length": r.size_t(segment_length)},
                {"compressed_input": r.pointer(out["compressed_input"], "i%d" % (8 * csize)), "segment_model_coefficients": r.pointer(out["segment_model_coefficients"], "i32"),
                 "input": r.const_pointer(column, "i%d" % (8 * usize)), "length": r.uint_t(index_size)(n), "segment_length": r.uint_t(index_size)(segment_length),
                 "unrepresentable_count": r.pointer(counter, "u32")}, counter, "offsets")
    return out


def dictionary_entries(column, limit=1 << 16, index_size=4):
    """the column's distinct values, sorted ascending (unsigned order) — numpy.unique on the device; at most `limit` of them (Unrepresentable beyond).
    Up to 65 536: a hash set of 2^18 slots, sorted by rank; more (4- and 8-byte keys): a hash set of four slots per entry of `limit`, the entries
    sorted in place by a bitonic network (include/giddy/kernels/compression/dictionary_construction.hpp)"""
    n, usize = column.numel(), column.element_size()
    device = column.device
    large = limit > (1 << 16)
    capacity = int(limit) if large else 1 << 16
    if capacity >= 1 << 29:
        raise ValueError("a dictionary of 2^29 entries or more is not built on the device")
    padded = 1 << max(0, (capacity - 1).bit_length())
    slots = max(1 << 18, 4 * padded) if large else 1 << 18
    states = torch.zeros(slots, dtype=torch.int32, device=device)
    status = torch.zeros(2, dtype=torch.int32, device=device)
    values = torch.empty(slots, dtype=torch.int64, device=device)
    entries = torch.empty(padded, dtype=column.dtype, device=device)
    if n:
        arguments = {"input": r.const_pointer(column, "u%d" % (8 * usize)), "length": r.uint_t(index_size)(n), "dictionary_entries": r.pointer(entries, "u%d" % (8 * usize)),
                     "slot_states": r.pointer(states, "u32"), "status": r.pointer(status, "u32"), "slot_values": r.pointer(values, "u64")}
        if large:
            arguments.update({"capacity": r.uint32(capacity), "hash_slots": r.uint32(slots)})
        else:
            unsorted = torch.empty(1 << 16, dtype=column.dtype, device=device)
            arguments["unsorted_entries"] = r.pointer(unsorted, "u%d" % (8 * usize))
        _launch("compression::dictionary::construction::kernel_t<%du, %du>" % (index_size, usize), {"length": r.size_t(n)}, arguments)
    found = int(status[0])
    if found > limit:
        raise Unrepresentable("too many distinct values for the key type")
    return entries[:found]


def dictionary(column, key_dtype=np.uint8, entries=None, index_size=4):
    """`entries`: the dictionary, sorted ascending and distinct, on the device.  Not given: the column's distinct values, found and
    sorted on the device (dictionary_entries())."""
    n, usize, ksize = column.numel(), column.element_size(), _size(key_dtype)
    if entries is None:
        limit = min(1 << (8 * ksize), max(1, n), 1 << 24) if ksize > 2 else 1 << (8 * ksize)
        if limit > (1 << 16):
            # Wide keys: the 65 536-entry construction FIRST (a hash set of 2^18 slots: 3 MiB) — most columns' dictionaries fit it.  Only a column that
            # overflows it pays for the large one: room for as many entries as the column could have, up to 2^24 — four hash slots per entry, i.e. up to
            # 2^26 slots = 256 MiB of states + 512 MiB of values, zero-filled, and the bitonic network over 2^24 entries.  (A dictionary of more than
            # 2^24 entries is no compression; pass `entries` for one, or for a column whose cardinality is known.)
            try:
                entries = dictionary_entries(column, limit=1 << 16, index_size=index_size)
            except Unrepresentable:
                entries = dictionary_entries(column, limit=limit, index_size=index_size)
        else:
            entries = dictionary_entries(column, limit=limit, index_size=index_size)
    if entries.numel() > (1 << (8 * ksize)):
        raise Unrepresentable("too many distinct values for the key type")
    out = {"compressed_input": torch.empty(n, dtype=_SIGNED[ksize], device=column.device), "dictionary_entries": entries, "length": n,
           "num_dictionary_entries": entries.numel()}
    counter = _counter(column.device)
    if n:
        _launch("compression::dictionary::kernel_t<%du, %du, %du>" % (index_size, usize, ksize), {"length": r.size_t(n)},
                {"compressed_input": r.pointer(out["compressed_input"], "u%d" % (8 * ksize)), "input": r.const_pointer(column, "u%d" % (8 * usize)),
                 "dictionary_entries": r.const_pointer(entries, "u%d" % (8 * usize)), "length": r.uint_t(index_size)(n),
                 "num_dictionary_entries": r.size_type_by_index_size(ksize)(entries.numel()), "unrepresentable_count": r.pointer(counter, "u32")},
                counter, "values (not in the dictionary)")
    return out


def incidence_bitmaps(column, alignment_in_ints=2, use_dictionary=True, index_size=4):
    """one bitmap per distinct value (at most 256); the column's length must be a multiple of 32"""
    n = column.numel()
    if n % 32:
        raise ValueError("the kernel decodes whole 32-bit containers")
    device = column.device
    if use_dictionary:
        encoded = dictionary(column, np.uint8, index_size=index_size)
        entries, keys = encoded["dictionary_entries"], encoded["compressed_input"]
    else:
        if column.element_size() != 1:
            raise ValueError("without a dictionary the values are the bitmap indices: one byte each")
        count = (int(column.to(torch.int64).bitwise_and(255).max()) + 1) if n else 0
        entries, keys = torch.arange(count, device=device).to(column.dtype), column
    words, num_bitmaps = n // 32, entries.numel()
    aligned = -(-words // alignment_in_ints) * alignment_in_ints
    bitmaps = torch.empty(num_bitmaps * aligned, dtype=torch.int32, device=device)
    if n:
        _launch("compression::incidence_bitmaps::kernel_t<%du, %du>" % (index_size, alignment_in_ints), {"length": r.size_t(n)},
                {"incidence_bitmaps": r.pointer(bitmaps, "u32"), "keys": r.const_pointer(keys, "u8"),
                 "bitmap_length": r.size_type_by_index_size(index_size)(words), "num_bitmaps": r.uint_t(4)(num_bitmaps)})
    return {"incidence_bitmaps": bitmaps, "dictionary_entries": entries, "bitmap_length": words, "num_bitmaps": num_bitmaps}


class CapacityExceeded(ValueError):
    """the column has more runs / patches than the caller's `capacity`; `.needed` is the true number (nothing was written out of bounds)"""
    def __init__(self, what, needed, capacity):
        super().__init__("%d %s, room for %d" % (needed, what, capacity))
        self.needed = needed


class ChainBroken(RuntimeError):
    """a one-launch compressor's tiles wait for the tiles before them (kernels/compression/chained.hpp); a tile that is not heard of within its
    patience gives up, every later tile with it, and the count the kernel leaves is 2^64 - 1: nothing was compressed, nothing hangs"""


def _chain_arguments(chain):
    """`chain`: None, or {"patience": polls, "fault_tile": t} (tests: a short patience, a tile that plays dead)"""
    if not chain:
        return {}
    out = {}
    if "patience" in chain:
        out["chain_patience"] = r.uint32(chain["patience"])
    if "fault_tile" in chain:
        out["chain_fault_tile"] = r.uint64(chain["fault_tile"])
    return out


def _count_of(runs, capacity):
    count = int(runs[0])
    if count == -1:   # (2^64 - 1 in the device scalar)
        raise ChainBroken("the chain of tiles broke: a tile was not heard of within its patience (kernels/compression/chained.hpp)")
    if count > capacity:
        raise CapacityExceeded("runs", count, capacity)
    return count


def run_length_encoding(column, segment_length, run_length_dtype=np.uint16, index_size=4, capacity=None, chain=None, staged_per_wave=None):
    """`capacity`: runs the output buffers have room for (default: as many as elements, the worst case).  The kernels are told, store nothing
    beyond it and still report the true number of runs: CapacityExceeded carries it."""
    n, usize, lsize = column.numel(), column.element_size(), _size(run_length_dtype)
    device = column.device
    capacity = capacity or max(n, 1)
    nseg = -(-n // segment_length)
    buffers = {"run_data": torch.empty(capacity, dtype=column.dtype, device=device), "run_lengths": torch.empty(capacity, dtype=_SIGNED[lsize], device=device),
               "position_anchors": torch.empty(nseg, dtype=_SIGNED[index_size], device=device), "intra_run_anchor_offsets": torch.empty(nseg, dtype=_SIGNED[lsize], device=device)}
    runs = torch.zeros(1, dtype=torch.int64, device=device)
    if n:
        _launch("compression::run_length_encoding::kernel_t<%du, %du, %du, %du>" % (index_size, usize, lsize, index_size),
                {"length": r.size_t(n), "segment_length": r.size_t(segment_length)},
                {"run_data": r.pointer(buffers["run_data"], "u%d" % (8 * usize)), "run_lengths": r.pointer(buffers["run_lengths"], "u%d" % (8 * lsize)),
                 "position_anchors": r.pointer(buffers["position_anchors"], "u%d" % (8 * index_size)),
                 "intra_run_anchor_offsets": r.pointer(buffers["intra_run_anchor_offsets"], "u%d" % (8 * lsize)),
                 "input": r.const_pointer(column, "u%d" % (8 * usize)), "length": r.uint_t(index_size)(n), "segment_length": r.uint_t(index_size)(segment_length),
                 "tile_totals": r.pointer(_scratch(n, device), "u64"),
                 "num_element_runs": r.pointer(runs, "u64"), "capacity": r.size_t(capacity), **_chain_arguments(chain),
                 **(_run_scratch(n, usize, device, chain, staged_per_wave) if lsize >= 2 else {})})
    count = _count_of(runs, capacity)
    return {"run_data": buffers["run_data"][:count], "run_lengths": buffers["run_lengths"][:count], "position_anchors": buffers["position_anchors"],
            "intra_run_anchor_offsets": buffers["intra_run_anchor_offsets"], "segment_length": segment_length, "num_segments": nseg,
            "num_element_runs": count, "length": n}


def run_position_encoding(column, segment_length, position_dtype=np.uint16, relative=True, index_size=4, capacity=None, chain=None, staged_per_wave=None):
    n, usize, psize = column.numel(), column.element_size(), _size(position_dtype)
    device = column.device
    capacity = capacity or max(n, 1)
    nseg = -(-n // segment_length)
    buffers = {"run_data": torch.empty(capacity, dtype=column.dtype, device=device), "run_start_positions": torch.empty(capacity, dtype=_SIGNED[psize], device=device),
               "position_anchors": torch.empty(nseg, dtype=_SIGNED[index_size], device=device)}
    runs = torch.zeros(1, dtype=torch.int64, device=device)
    if n:
        _launch("compression::run_position_encoding::kernel_t<%du, %du, %du, %s, %du>" % (index_size, usize, psize, "true" if relative else "false", index_size),
                {"length": r.size_t(n), "segment_length": r.size_t(segment_length)},
                {"run_data": r.pointer(buffers["run_data"], "u%d" % (8 * usize)), "run_start_positions": r.pointer(buffers["run_start_positions"], "u%d" % (8 * psize)),
                 "position_anchors": r.pointer(buffers["position_anchors"], "u%d" % (8 * index_size)),
                 "input": r.const_pointer(column, "u%d" % (8 * usize)), "length": r.uint_t(index_size)(n), "segment_length": r.uint_t(index_size)(segment_length),
                 "tile_totals": r.pointer(_scratch(n, device), "u64"), "num_element_runs": r.pointer(runs, "u64"), "capacity": r.size_t(capacity), **_chain_arguments(chain),
                 **_run_scratch(n, usize, device, chain, staged_per_wave)})
    count = _count_of(runs, capacity)
    return {"run_data": buffers["run_data"][:count], "run_start_positions": buffers["run_start_positions"][:count], "position_anchors": buffers["position_anchors"],
            "segment_length": segment_length, "num_anchors": nseg, "num_element_runs": count, "length": n}


def patches(decoded, truth, index_size=4, capacity=None):
    """positions (ascending) where `decoded` differs from `truth`, and the true values there"""
    n, esize = truth.numel(), truth.element_size()
    device = truth.device
    capacity = capacity or max(n, 1)
    positions = torch.empty(capacity, dtype=_SIGNED[index_size], device=device)
    values = torch.empty(capacity, dtype=truth.dtype, device=device)
    count = torch.zeros(1, dtype=torch.int64, device=device)
    if n:
        _launch("compression::patches::kernel_t<%du, %du>" % (index_size, esize), {"length": r.size_t(n)},
                {"patch_positions": r.pointer(positions, "u%d" % (8 * index_size)), "patch_values": r.pointer(values, "u%d" % (8 * esize)),
                 "decoded": r.const_pointer(decoded, "u%d" % (8 * esize)), "truth": r.const_pointer(truth, "u%d" % (8 * esize)),
                 "length": r.uint_t(index_size)(n), "tile_totals": r.pointer(_scratch(n, device), "u64"), "num_patches": r.pointer(count, "u64"),
                 "capacity": r.size_t(capacity)})
    k = int(count[0])
    if k > capacity:
        raise CapacityExceeded("patches", k, capacity)
    return {"indices": positions[:k], "data": values[:k], "data_length": k}


def compressed_indices(positions, anchoring_period, output_index_size=4, input_index_size=4, run_length_size=None):
    """ascending, distinct patch positions (a device tensor of output_index_size-byte integers) -> the compressed-indices arrays the
    compressed-indices scatter and the fused decoders take (without "data_to_scatter": the patch values are the caller's)"""
    run_length_size = run_length_size or input_index_size
    n = positions.numel()
    if n == 0:
        raise ValueError("zero-length scatters are rejected (compressed_indices_scatter.cuh:288-290)")
    device = positions.device
    O, I, R = output_index_size, input_index_size, run_length_size
    blocks = -(-n // anchoring_period)
    buffers = {"scatter_position_run_lengths": torch.empty(n, dtype=_SIGNED[R], device=device),
               "scatter_position_run_individual_offset_sizes": torch.empty(n, dtype=torch.uint8, device=device),
               "scatter_position_run_baseline_values": torch.empty(n, dtype=_SIGNED[O], device=device),
               "scatter_position_run_offsets_start_pos": torch.empty(n, dtype=_SIGNED[I], device=device),
               "scatter_position_offset_bytes": torch.empty(n * 2 * O + 8, dtype=torch.uint8, device=device),
               "scatter_position_anchors": torch.empty(blocks, dtype=_SIGNED[I], device=device)}
    sizes = torch.zeros(2, dtype=torch.int64, device=device)
    scratch = torch.empty(2 * (blocks + 2), dtype=torch.int64, device=device)
    launch = {name: r.pointer(t, "u8" if t.dtype == torch.uint8 else "u%d" % (8 * t.element_size())) for name, t in buffers.items()}
    launch.update({"positions": r.const_pointer(positions, "u%d" % (8 * O)), "input_data_length": r.size_type_by_index_size(I)(n),
                   "anchoring_period": r.uint_t(I)(anchoring_period), "tile_totals": r.pointer(scratch, "u64"),
                   "num_scatter_position_runs": r.pointer(sizes[0:1], "u64"), "num_offset_bytes": r.pointer(sizes[1:2], "u64")})
    _launch("compression::compressed_indices::kernel_t<%du, %du, %du>" % (O, I, R),
            {"input_data_length": r.size_t(n), "anchoring_period": r.size_t(anchoring_period)}, launch)
    runs, pool = int(sizes[0]), -(-int(sizes[1]) // 8) * 8
    out = {name: (t if name == "scatter_position_anchors" else t[: pool if name == "scatter_position_offset_bytes" else runs]) for name, t in buffers.items()}
    out.update({"anchoring_period": anchoring_period, "num_scatter_position_runs": runs, "input_data_length": n})
    return out

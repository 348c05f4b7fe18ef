"""-m gpu: every shard of a sharded column decoded by the HIP kernels (one GPU plays each rank in turn),
concatenation == the column.  Shard inputs are fresh device buffers, as they would be on their own GPU."""
import numpy as np
import pytest

import cases
from libgiddy_amd import formats, sharding

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_decode_on_gpu(world):
    import gpu_util as g
    from libgiddy_amd import schemes
    st = g.stream()
    n = 300000

    def run(shard_fn, fields, truth, launch, dtype):
        pieces = []
        for rank in range(world):
            sh = shard_fn(fields, rank, world)
            count = sh.stop - sh.start
            out = g.dev_empty(count, dtype)
            if count:
                launch(out, sh.fields)
            pieces.append(g.host(out, dtype, count))
        got = np.concatenate(pieces)
        assert (got == truth).all(), np.flatnonzero(got != truth)[:5]

    f, truth = cases.delta_case(n, 4096, np.uint64, np.uint16, seed=3)
    run(sharding.delta, f, truth, lambda out, s: schemes.delta(out, g.dev(s["compressed_input"]), g.dev(s["anchor_values"]), s["length"], 4096, stream=st), np.uint64)

    col, sizes = cases.dzbv_column(n, np.uint64, 1, 3, seed=4)
    f = formats.discard_zero_bytes_variable(col, 1024, 1, 3, np.uint32, 8, element_sizes=sizes)
    run(sharding.discard_zero_bytes_variable, f, col,
        lambda out, s: schemes.discard_zero_bytes_variable(out, g.dev(s["compressed_data"]), g.dev(s["packed_element_sizes"]), g.dev(s["position_anchors"]),
                                                           1024, s["length"], 1, 3, stream=st), np.uint64)

    col = cases.runs_column(n, 32, np.uint32, seed=5)
    f = formats.run_length_encoding(col, 4096, np.uint16)
    run(sharding.run_length_encoding, f, col,
        lambda out, s: schemes.run_length_encoding(out, g.dev(s["run_data"]), g.dev(s["run_lengths"]), g.dev(s["position_anchors"]),
                                                   g.dev(s["intra_run_anchor_offsets"]), 4096, s["num_segments"], s["num_element_runs"], s["length"], stream=st), np.uint32)
    for relative, pd in ((True, np.uint16), (False, np.uint32)):
        f = formats.run_position_encoding(col, 4096, pd, relative)
        run(lambda ff, r_, w: sharding.run_position_encoding(ff, r_, w, relative=relative), f, col,
            lambda out, s: schemes.run_position_encoding(out, g.dev(s["run_data"]), g.dev(s["run_start_positions"]), g.dev(s["position_anchors"]), 4096,
                                                         s["num_anchors"], s["num_element_runs"], s["length"], position_size=np.dtype(pd).itemsize,
                                                         relative=relative, stream=st), np.uint32)

    f = cases.dictionary_case(n, 256, np.uint32, np.uint8, seed=2)
    truth = f["dictionary_entries"][f["compressed_input"].astype(np.int64)]
    run(sharding.elementwise, f, truth,
        lambda out, s: schemes.dictionary(out, g.dev(s["compressed_input"]), g.dev(s["dictionary_entries"]), s["length"], 256, stream=st), np.uint32)


def test_pipelined_decode_from_pinned_host_memory():
    """libgiddy_amd.pipeline: chunks copied H2D on several streams while earlier chunks decode."""
    from libgiddy_amd import pipeline
    n = (1 << 22) + 12345
    gen = torch.Generator(); gen.manual_seed(3)
    keys_host = torch.randint(0, 256, (n,), dtype=torch.uint8, generator=gen).pin_memory()
    entries = (torch.arange(256, device="cuda") * 16777259 + 5).to(torch.int32)
    out = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    for s in pipeline.dictionary_from_host(keys_host, entries, out, chunk_elements=1 << 20, depth=3):
        s.synchronize()
    assert bool((out == entries[keys_host.cuda().long()]).all())


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mean_run", [3, 40, 3000])
def test_sharded_run_columns_in_long_segments(world, mean_run):
    """Shards of run-encoded columns whose segments are long enough for the decoders' finer anchors (kernels/decompression/run_anchors.hpp):
    a shard's run arrays begin mid-run and its anchors are rebased — the pre-pass must only ever take differences within a segment."""
    import gpu_util as g
    from libgiddy_amd import schemes
    st = g.stream()
    n, L = 1500000, 1 << 16
    col = cases.runs_column(n, mean_run, np.uint32, seed=50 + mean_run)
    for L in (1 << 16, 1 << 18):
        f = formats.run_length_encoding(col, L, np.uint16)
        pieces = []
        for rank in range(world):
            sh = sharding.run_length_encoding(f, rank, world)
            count = sh.stop - sh.start
            out = g.dev_empty(count, np.uint32)
            s = sh.fields
            if count:
                schemes.run_length_encoding(out, g.dev(s["run_data"]), g.dev(s["run_lengths"]), g.dev(s["position_anchors"]), g.dev(s["intra_run_anchor_offsets"]),
                                            L, s["num_segments"], s["num_element_runs"], s["length"], stream=st)
            pieces.append(g.host(out, np.uint32, count))
        got = np.concatenate(pieces)
        assert (got == col).all(), ("rle", L, np.flatnonzero(got != col)[:5])
        for relative, pd in ((True, np.uint16), (False, np.uint32)):
            if relative and L > 1 << 16: continue
            f = formats.run_position_encoding(col, L, pd, relative)
            pieces = []
            for rank in range(world):
                sh = sharding.run_position_encoding(f, rank, world, relative=relative)
                count = sh.stop - sh.start
                out = g.dev_empty(count, np.uint32)
                s = sh.fields
                if count:
                    schemes.run_position_encoding(out, g.dev(s["run_data"]), g.dev(s["run_start_positions"]), g.dev(s["position_anchors"]), L, s["num_anchors"],
                                                  s["num_element_runs"], s["length"], position_size=np.dtype(pd).itemsize, relative=relative, stream=st)
                pieces.append(g.host(out, np.uint32, count))
            got = np.concatenate(pieces)
            assert (got == col).all(), ("rpe", relative, L, np.flatnonzero(got != col)[:5])


@pytest.mark.parametrize("num_chunks,depth", [(1, 1), (5, 2), (16, 3)])
def test_pipelined_decode_of_every_sharding_scheme_from_host(num_chunks, depth):
    """libgiddy_amd.pipeline.decode_from_host: chunk = shard (anchors rebased on the host), copies and decodes overlapped
    on several streams; the result must be the column, for every scheme that shards"""
    from libgiddy_amd import pipeline
    n = 300000 + 77

    def check(scheme, fields, truth, **extra):
        out = torch.full((len(truth),), 0x5A, dtype=torch.uint8, device="cuda").repeat_interleave(truth.dtype.itemsize)
        out = out.view({1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[truth.dtype.itemsize])
        for s in pipeline.decode_from_host(scheme, pipeline.pin(fields), out, num_chunks=num_chunks, depth=depth, **extra):
            s.synchronize()
        got = out.cpu().numpy().view(truth.dtype)
        assert (got == truth).all(), (scheme, np.flatnonzero(got != truth)[:5])

    f, truth = cases.delta_case(n, 4096, np.uint64, np.uint16, seed=3)
    check("delta", f, truth)
    col, sizes = cases.dzbv_column(n, np.uint64, 1, 3, seed=4)
    check("discard_zero_bytes_variable", formats.discard_zero_bytes_variable(col, 1024, 1, 3, np.uint32, 8, element_sizes=sizes), col)
    col = cases.runs_column(n, 32, np.uint32, seed=5)
    check("run_length_encoding", formats.run_length_encoding(col, 4096, np.uint16), col)
    check("run_position_encoding", formats.run_position_encoding(col, 4096, np.uint16, True), col, relative=True)
    f = cases.dictionary_case(n, 256, np.uint32, np.uint8, seed=2)
    check("dictionary", dict(f, length=n), f["dictionary_entries"][f["compressed_input"].astype(np.int64)])
    f = cases.for_case(n, 4096, np.int32, np.int16, "linear", seed=6)
    from oracle import oracle
    check("frame_of_reference", f, oracle.frame_of_reference(f["compressed_input"], f["segment_model_coefficients"], n, 4096, np.int32, "linear"), model="linear")
    small = (cases.rng(7).integers(0, 1 << 20, size=n)).astype(np.uint64)
    check("discard_zero_bytes_fixed", formats.discard_zero_bytes_fixed(small, np.uint32), small)


@pytest.mark.parametrize("num_chunks,depth", [(1, 1), (5, 2), (64, 3)])
def test_native_pipeline_decodes_every_sharding_scheme_from_host(num_chunks, depth):
    """giddy_decode_from_host (the chunk loop in C++ behind the C ABI: cuts of include/giddy/sharding.hpp, slices copied into the stream's
    scratch, anchors and absolute positions rebased ON THE DEVICE, the wrapper's launch) gives the column for every scheme"""
    from libgiddy_amd import pipeline
    n = 300000 + 77

    def check(scheme, fields, truth, **extra):
        out = torch.full((len(truth),), 0x5A, dtype=torch.uint8, device="cuda").repeat_interleave(truth.dtype.itemsize)
        out = out.view({1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[truth.dtype.itemsize])
        pinned = pipeline.pin(fields)   # (kept until the streams are done: the copies out of it are asynchronous, and unlike torch's own copy_ they
        for s in pipeline.decode_from_host_native(scheme, pinned, out, num_chunks=num_chunks, depth=depth, **extra):   # are unknown to torch's pinned-memory cache)
            s.synchronize()
        del pinned
        got = out.cpu().numpy().view(truth.dtype)
        assert (got == truth).all(), (scheme, np.flatnonzero(got != truth)[:5])

    f, truth = cases.delta_case(n, 4096, np.uint64, np.uint16, seed=3)
    check("delta", f, truth)
    col, sizes = cases.dzbv_column(n, np.uint64, 1, 3, seed=4)
    check("discard_zero_bytes_variable", formats.discard_zero_bytes_variable(col, 1024, 1, 3, np.uint32, 8, element_sizes=sizes), col)
    col = cases.runs_column(n, 32, np.uint32, seed=5)
    check("run_length_encoding", formats.run_length_encoding(col, 4096, np.uint16), col)
    check("run_position_encoding", formats.run_position_encoding(col, 4096, np.uint16, True), col, relative=True)
    check("run_position_encoding", formats.run_position_encoding(col, 4096, np.uint32, False), col, relative=False)
    f = cases.dictionary_case(n, 256, np.uint32, np.uint8, seed=2)
    check("dictionary", dict(f, length=n), f["dictionary_entries"][f["compressed_input"].astype(np.int64)])
    f = cases.for_case(n, 4096, np.int32, np.int16, "linear", seed=6)
    from oracle import oracle
    check("frame_of_reference", f, oracle.frame_of_reference(f["compressed_input"], f["segment_model_coefficients"], n, 4096, np.int32, "linear"), model="linear")
    small = (cases.rng(7).integers(0, 1 << 20, size=n)).astype(np.uint64)
    check("discard_zero_bytes_fixed", formats.discard_zero_bytes_fixed(small, np.uint32), small)


def test_rebase_on_the_device():
    """giddy_enqueue_rebase: out[i] = in[i] - bias, wrapping in the entries' own width, in place or not, any count"""
    import ctypes
    from libgiddy_amd import _native as native
    stream = torch.cuda.current_stream().cuda_stream
    for dtype, np_dtype in ((torch.int8, np.uint8), (torch.int16, np.uint16), (torch.int32, np.uint32), (torch.int64, np.uint64)):
        for count in (0, 1, 255, 100001):
            host = cases.rng(count + 1).integers(0, 1 << (8 * np.dtype(np_dtype).itemsize - 1), size=count, dtype=np.uint64).astype(np_dtype)
            bias = int(host[count // 2]) + 7 if count else 5
            source = torch.from_numpy(host.view(np.dtype("i%d" % host.dtype.itemsize))).cuda()
            target = torch.zeros_like(source)
            assert native.lib.giddy_enqueue_rebase(stream, target.data_ptr(), source.data_ptr(), count, host.dtype.itemsize, bias) == 0, native.lib.giddy_last_error()
            assert native.lib.giddy_enqueue_rebase(stream, source.data_ptr(), source.data_ptr(), count, host.dtype.itemsize, bias) == 0
            torch.cuda.synchronize()
            expect = host - np_dtype(bias & ((1 << (8 * host.dtype.itemsize)) - 1))
            assert (target.cpu().numpy().view(np_dtype) == expect).all() and (source.cpu().numpy().view(np_dtype) == expect).all()
    assert native.lib.giddy_enqueue_rebase(stream, 0, 0, 5, 4, 1) != 0 and native.lib.giddy_enqueue_rebase(stream, 8, 8, 5, 3, 1) != 0


def test_pipelined_decode_with_large_anchor_arrays():
    """a column whose ANCHOR arrays are hundreds of KiB (2^22 elements in segments of 64): position anchors — rebased per chunk on
    the host — stay host arrays whatever their size; every array a chunk takes a plain slice of is pinned (by name: the bulk arrays,
    and Delta's anchor VALUES and the model records, which need no rebasing)"""
    from libgiddy_amd import pipeline
    n = 1 << 22

    def check(scheme, fields, truth, **extra):
        pinned = pipeline.pin(fields)
        assert all(isinstance(pinned[k], np.ndarray) for k in fields if "position_anchor" in k)
        assert all(isinstance(pinned[k], torch.Tensor) and pinned[k].is_pinned() for k in pipeline.BULK_ARRAYS if k in fields)
        out = torch.zeros(len(truth), dtype={4: torch.int32, 8: torch.int64}[truth.dtype.itemsize], device="cuda")
        out.fill_(-1)   # queued on the current stream: the pipeline's side streams must wait for it
        for s in pipeline.decode_from_host(scheme, pinned, out, num_chunks=7, depth=3, **extra):
            s.synchronize()
        got = out.cpu().numpy().view(truth.dtype)
        assert (got == truth).all(), (scheme, np.flatnonzero(got != truth)[:5])

    col = cases.runs_column(n, 8, np.uint32, seed=15)
    f = formats.run_length_encoding(col, 64, np.uint16)
    assert f["position_anchors"].nbytes >= 1 << 18
    check("run_length_encoding", f, col)
    col, sizes = cases.dzbv_column(n, np.uint64, 1, 2, seed=16)
    f = formats.discard_zero_bytes_variable(col, 64, 1, 2, np.uint32, 8, element_sizes=sizes)
    assert f["position_anchors"].nbytes >= 1 << 18
    check("discard_zero_bytes_variable", f, col)
    f, truth = cases.delta_case(n, 64, np.uint64, np.uint16, seed=17)
    check("delta", f, truth)


@pytest.mark.parametrize("world", [2, 8])
def test_device_resident_column_is_cut_without_a_host_round_trip(world):
    """sharding.* on DEVICE tensors: the shard's big arrays are views of the column's, the rebased anchors are computed on
    the device; every rank's shard decodes to its range of the column"""
    import gpu_util as g
    from libgiddy_amd import schemes
    st = g.stream()
    n = 1 << 20
    signed = {1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}

    def to_device(fields):
        return {k: (torch.from_numpy(np.ascontiguousarray(v).view({1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}[v.dtype.itemsize])).cuda()
                    if isinstance(v, np.ndarray) else v) for k, v in fields.items()}

    def check(shard_fn, fields, truth, launch):
        device_fields = to_device(fields)
        out = torch.zeros(len(truth), dtype=signed[truth.dtype.itemsize], device="cuda")
        for rank in range(world):
            sh = shard_fn(device_fields, rank, world)
            for k, v in sh.fields.items():
                assert not isinstance(v, torch.Tensor) or v.is_cuda, k
            if sh.stop > sh.start:
                launch(out[sh.start:sh.stop], sh.fields)
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(truth.dtype)
        assert (got == truth).all(), (shard_fn.__name__, np.flatnonzero(got != truth)[:5])

    col = cases.runs_column(n, 32, np.uint32, seed=21)
    check(sharding.run_length_encoding, formats.run_length_encoding(col, 4096, np.uint16), col,
          lambda out, s: schemes.run_length_encoding(out, s["run_data"], s["run_lengths"], s["position_anchors"], s["intra_run_anchor_offsets"],
                                                     s["segment_length"], s["num_segments"], s["num_element_runs"], s["length"], stream=st))
    check(lambda f, r_, w: sharding.run_position_encoding(f, r_, w, relative=False), formats.run_position_encoding(col, 4096, np.uint32, False), col,
          lambda out, s: schemes.run_position_encoding(out, s["run_data"], s["run_start_positions"], s["position_anchors"], s["segment_length"],
                                                       s["num_anchors"], s["num_element_runs"], s["length"], position_size=4, relative=False, stream=st))
    colv, sizes = cases.dzbv_column(n, np.uint64, 1, 2, seed=22)
    check(sharding.discard_zero_bytes_variable, formats.discard_zero_bytes_variable(colv, 1024, 1, 2, np.uint32, 8, element_sizes=sizes), colv,
          lambda out, s: schemes.discard_zero_bytes_variable(out, s["compressed_data"], s["packed_element_sizes"], s["position_anchors"], s["segment_length"],
                                                             s["length"], 1, 2, uncompressed_size=8, index_size=8, stream=st))
    f, truth = cases.delta_case(n, 4096, np.uint64, np.uint16, seed=23)
    check(sharding.delta, f, truth,
          lambda out, s: schemes.delta(out, s["compressed_input"], s["anchor_values"], s["length"], s["segment_length"], uncompressed_size=8, compressed_size=2, stream=st))


def test_gather_on_node_through_the_c_abi():
    """giddy_enqueue_gather_on_node with every "device" being device 0 (one GPU per box here): three decoded shards, each behind work on its own
    stream, land shard after shard in one buffer that the root's stream sees whole."""
    import ctypes
    from libgiddy_amd import _native
    lib = _native.lib
    streams = [torch.cuda.Stream() for _ in range(3)]
    root = torch.cuda.Stream()
    counts = [1000003, 0, 777]
    shards = []
    for k, (count, stream) in enumerate(zip(counts, streams)):
        with torch.cuda.stream(stream):
            shards.append(torch.arange(count, dtype=torch.int64, device="cuda") * (k + 3) + k)   # ("decoding" on the shard's stream)
    out = torch.empty(sum(counts), dtype=torch.int64, device="cuda")
    pointers = (ctypes.c_void_p * 3)(*[s.data_ptr() if s.numel() else None for s in shards])
    devices = (ctypes.c_int * 3)(0, 0, 0)
    handles = (ctypes.c_void_p * 3)(*[s.cuda_stream for s in streams])
    sizes = (ctypes.c_uint64 * 3)(*[8 * c for c in counts])
    status = lib.giddy_enqueue_gather_on_node(out.data_ptr(), 0, root.cuda_stream, pointers, devices, handles, sizes, 3)
    assert status == 0, _native.last_error()
    with torch.cuda.stream(root):
        total = out.sum()   # (enqueued on the root's stream: behind all three copies)
    root.synchronize()
    expect = torch.cat(shards)
    assert bool((out == expect).all()) and int(total) == int(expect.sum())


@pytest.mark.parametrize("num_chunks,depth", [(3, 2), (8, 3)])
def test_native_pipeline_with_long_segments(num_chunks, depth):
    """giddy_decode_from_host over columns whose wrappers take scratch of their OWN inside a chunk (round-3 advisor): chained Delta tiles
    (segments of 2^18), finer anchors for run decoders (segments of 65 536) and for discard-zero-bytes (segments of 4096) — the chunk's
    slices and the wrapper's scratch are two blocks of the pool (include/giddy/stream_scratch.hpp), neither waits for the other, and the
    pool does not grow from one chunk of a stream to the next."""
    from libgiddy_amd import pipeline, registered
    n = 1 << 21

    def check(scheme, fields, truth, **extra):
        out = torch.full((len(truth),), 0x5A, dtype=torch.uint8, device="cuda").repeat_interleave(truth.dtype.itemsize)
        out = out.view({1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[truth.dtype.itemsize])
        pinned = pipeline.pin(fields)
        # (every pass gets streams of its own from torch's pool of them, so a block is "this stream's own" only within a pass: the first two passes may still
        # add a block — which of the equal-sized idle blocks a request gets depends on the order the pool was filled in — after that nothing may be allocated)
        for repetition in range(4):
            for s in pipeline.decode_from_host_native(scheme, pinned, out, num_chunks=num_chunks, depth=depth, **extra):
                s.synchronize()
            if repetition == 1:
                held = registered.scratch_held_bytes()
        # (a request that finds its stream's blocks taken and the others' last use still pending allocates — there is no waiting on the launch path —, so a pass may
        # still add a block now and then; what must not happen is a block per chunk: 2 passes x 8 chunks of ~1 MiB each would more than double the pool)
        assert registered.scratch_held_bytes() <= held + held // 2, "later passes over the same column kept allocating scratch"
        del pinned
        got = out.cpu().numpy().view(truth.dtype)
        assert (got == truth).all(), (scheme, np.flatnonzero(got != truth)[:5])

    f, truth = cases.delta_case(n, 1 << 18, np.uint64, np.uint16, seed=13)
    check("delta", f, truth)
    col = cases.runs_column(n, 32, np.uint32, seed=15)
    check("run_length_encoding", formats.run_length_encoding(col, 65536, np.uint16), col)
    check("run_position_encoding", formats.run_position_encoding(col, 65536, np.uint16, True), col, relative=True)
    check("run_position_encoding", formats.run_position_encoding(col, 65536, np.uint32, False), col, relative=False)
    col, sizes = cases.dzbv_column(n, np.uint64, 1, 2, seed=14)
    check("discard_zero_bytes_variable", formats.discard_zero_bytes_variable(col, 4096, 1, 2, np.uint32, 8, element_sizes=sizes), col)


def test_gather_compressed_on_node_through_the_c_abi():
    """giddy_enqueue_gather_compressed_on_node with every "device" being device 0: three COMPRESSED shards of a discard-zero-bytes column and of
    a run-length column, each complete behind work on its own stream, are copied into the root's scratch (the copy is forced although the
    shards are local), decoded there into their ranges of one buffer, and the first root stream sees the whole column."""
    from libgiddy_amd import distributed, registered as r, schemes
    n, world = 300000 + 77, 3
    root_streams = [torch.cuda.Stream() for _ in range(2)]

    def on_device(fields, stream):
        with torch.cuda.stream(stream):
            return {k: (torch.from_numpy(np.ascontiguousarray(v).reshape(-1).view(np.uint8).copy()).cuda(non_blocking=True) if isinstance(v, np.ndarray) else v) for k, v in fields.items()}

    def gather(key, shard_function, fields, truth, arguments_of, resolution_of):
        out = torch.full((len(truth) * truth.dtype.itemsize,), 0x5A, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()   # (the fill is on torch's current stream, the decodes on streams of their own)
        shards, keep = [], []
        for rank in range(world):
            sh = shard_function(fields, rank, world)
            stream = torch.cuda.Stream()
            device_fields = on_device(sh.fields, stream)
            keep.append((stream, device_fields))
            arguments, nbytes = arguments_of(out[sh.start * truth.dtype.itemsize:], device_fields, sh.fields)
            shards.append({"device": 0, "stream": stream.cuda_stream, "arguments": arguments, "nbytes": nbytes,
                           "resolution": resolution_of(sh.fields), "copy_even_if_local": rank != 1, "decode_in_place": rank == 1})   # (one shard read where it lies)
        distributed.gather_compressed_on_node(key, 0, [s.cuda_stream for s in root_streams], shards)
        with torch.cuda.stream(root_streams[0]):
            got = out.clone()
        root_streams[0].synchronize()
        got = got.cpu().numpy().view(truth.dtype)
        assert (got == truth).all(), (key, np.flatnonzero(got != truth)[:5])

    col, sizes = cases.dzbv_column(n, np.uint64, 1, 2, seed=24)
    f = formats.discard_zero_bytes_variable(col, 1024, 1, 2, np.uint32, 8, element_sizes=sizes)

    def dzbv_arguments(out, d, s):
        arguments = {"decompressed": r.pointer(out, "u64"), "compressed_data": r.const_pointer(d["compressed_data"], "u8"),
                     "packed_element_sizes": r.const_pointer(d["packed_element_sizes"], "u32"), "position_anchors": r.const_pointer(d["position_anchors"], "u64"),
                     "segment_length": r.uint64(1024), "length": r.uint64(s["length"]), "min_represented_element_size": r.uint32(1), "bits_per_element_size": r.uint32(2)}
        return arguments, {k: d[k].numel() for k in ("compressed_data", "packed_element_sizes", "position_anchors")}
    gather(schemes.key_discard_zero_bytes_variable(8, 8, 4), sharding.discard_zero_bytes_variable, f, col, dzbv_arguments,
           lambda s: {"length": r.size_t(s["length"]), "segment_length": r.size_t(1024)})

    col = cases.runs_column(n, 32, np.uint32, seed=25)
    f = formats.run_length_encoding(col, 4096, np.uint16)

    def rle_arguments(out, d, s):
        arguments = {"decompressed": r.pointer(out, "u32"), "run_data": r.const_pointer(d["run_data"], "u32"), "run_lengths": r.const_pointer(d["run_lengths"], "u16"),
                     "position_anchors": r.const_pointer(d["position_anchors"], "u32"), "intra_run_anchor_offsets": r.const_pointer(d["intra_run_anchor_offsets"], "u16"),
                     "segment_length": r.uint32(4096), "num_segments": r.uint32(s["num_segments"]), "num_element_runs": r.uint32(s["num_element_runs"]), "length": r.uint32(s["length"])}
        return arguments, {k: d[k].numel() for k in ("run_data", "run_lengths", "position_anchors", "intra_run_anchor_offsets")}
    gather(schemes.key_run_length_encoding(4, 4, 2, 4), sharding.run_length_encoding, f, col, rle_arguments,
           lambda s: {"length": r.size_t(s["length"]), "segment_length": r.size_t(4096)})


def test_decode_directly_from_pinned_host_memory():
    """libgiddy_amd.pipeline.decode_directly_from_pinned_host: ONE launch per column, the kernels reading the compressed arrays out of pinned
    host memory over the link (no staging copy, no chunk loop) — every scheme of the pipelines, bit for bit."""
    from libgiddy_amd import pipeline
    from oracle import oracle
    n = 300000 + 77

    def check(scheme, fields, truth, **extra):
        out = torch.full((len(truth),), 0x5A, dtype=torch.uint8, device="cuda").repeat_interleave(truth.dtype.itemsize)
        out = out.view({1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[truth.dtype.itemsize])
        pinned = pipeline.pin_all(fields)
        pipeline.decode_directly_from_pinned_host(scheme, pinned, out, **extra)
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(truth.dtype)
        assert (got == truth).all(), (scheme, np.flatnonzero(got != truth)[:5])
        with pytest.raises(ValueError):
            pipeline.decode_directly_from_pinned_host(scheme, fields, out, **extra)   # (numpy arrays: not pinned)

    f, truth = cases.delta_case(n, 4096, np.uint64, np.uint16, seed=3)
    check("delta", f, truth)
    col, sizes = cases.dzbv_column(n, np.uint64, 1, 2, seed=4)
    check("discard_zero_bytes_variable", formats.discard_zero_bytes_variable(col, 1024, 1, 2, np.uint32, 8, element_sizes=sizes), col)
    col = cases.runs_column(n, 32, np.uint32, seed=5)
    check("run_length_encoding", formats.run_length_encoding(col, 4096, np.uint16), col)
    check("run_position_encoding", formats.run_position_encoding(col, 4096, np.uint16, True), col, relative=True)
    check("run_position_encoding", formats.run_position_encoding(col, 4096, np.uint32, False), col, relative=False)
    f = cases.dictionary_case(n, 256, np.uint32, np.uint8, seed=2)
    check("dictionary", dict(f, length=n), f["dictionary_entries"][f["compressed_input"].astype(np.int64)])
    f = cases.for_case(n, 4096, np.int32, np.int16, "linear", seed=6)
    check("frame_of_reference", f, oracle.frame_of_reference(f["compressed_input"], f["segment_model_coefficients"], n, 4096, np.int32, "linear"), model="linear")

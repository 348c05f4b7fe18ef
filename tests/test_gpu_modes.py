"""-m gpu: one decode per scheme, whichever of the reference's modes of use the caller takes.

The wrappers of Delta, run-length, run-position and variable-width discard-zero-bytes decoding pick among several kernels (chained tiles for
long Delta segments, builds of the run decoders for short / long runs, finer anchors for long segments, pieces of long columns).  That choice
lives in ONE typed `launch` per scheme, which the type-erased enqueue_launch (mode A), cuda::kernel::enqueue_launch on an instantiated
wrapper (mode B: examples/src/modes_of_use/instantiation_of_wrapper.cu:22-38) or on a base-type reference (src/cuda/kernel_wrapper.cuh:135-152)
and the free `<scheme>::launch<...>()` all reach; a direct launch of the kernel template (mode C: direct_use_of_kernel.cu:17-36) runs the kernel
of the reference's signature.  Here: every mode decodes the same device-resident column and must give mode A's bits (which
tests/test_gpu_parity_*.py and the workloads' own verify() hold against the oracle)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

MODES = ["B", "C", "C+", "B*"]


def _column(scheme, n, L, mean_run=32):
    import workloads as w
    device = torch.device("cuda", 0)
    if scheme.startswith("delta<4,8,2>"):
        work = w.DeltaPatched(n, device, segment_length=L)
        return {"compressed_input": work.deltas, "anchor_values": work.anchors, "length": n, "segment_length": L}, work
    if scheme.startswith("delta<4,4,1>"):
        deltas = torch.randint(0, 256, (n,), dtype=torch.int32, device=device).to(torch.uint8)
        sums = torch.cumsum(deltas.to(torch.int64), 0)
        anchors = torch.cat([torch.zeros(1, dtype=torch.int64, device=device), sums[L - 1:-1:L]]).add_(12345).to(torch.int32)

        class work:   # (a bare column: delta<4,4,1> is not among bench.py's workloads)
            output = torch.empty(n, dtype=torch.int32, device=device)

            @staticmethod
            def verify(oracle):
                from oracle import oracle as o
                expect = o.delta(deltas.cpu().numpy(), anchors.cpu().numpy().view(np.uint32), L)
                assert (expect == work.output.cpu().numpy().view(np.uint32)).all()
        return {"compressed_input": deltas, "anchor_values": anchors, "length": n, "segment_length": L}, work
    if scheme.startswith("rle"):
        work = w.RunLength(n, device, segment_length=L, mean_run=mean_run)
    elif scheme.startswith("rpe"):
        work = w.RunPosition(n, device, segment_length=L, mean_run=mean_run, relative="true" in scheme)
    else:
        work = w.DiscardZeroBytesVariable(n, device, segment_length=L)
    return work.column_fields(), work


CASES = [
    # (scheme, log2 n, segment length, mean run) — ordinary segments, and the shapes the wrappers' other kernels exist for
    ("delta<4,8,2>", 22, 4096, 0), ("delta<4,8,2>", 22, 1 << 18, 0), ("delta<4,8,2>", 22, 1 << 22, 0),
    ("delta<4,4,1>", 22, 4096, 0), ("delta<4,4,1>", 22, 1 << 20, 0),
    ("rle<4,4,2>", 22, 4096, 32), ("rle<4,4,2>", 22, 4096, 3), ("rle<4,4,2>", 23, 65536, 32), ("rle<4,4,2>", 23, 1 << 20, 4),
    ("rpe<4,4,2,true>", 22, 4096, 32), ("rpe<4,4,2,true>", 22, 4096, 3), ("rpe<4,4,2,true>", 23, 65536, 32),
    ("rpe<4,4,4,false>", 22, 4096, 32), ("rpe<4,4,4,false>", 23, 1 << 20, 32), ("rpe<4,4,4,false>", 22, 4096, 3),
    ("dzbv<8,8,4>", 22, 1024, 0), ("dzbv<8,8,4>", 22, 1 << 14, 0), ("dzbv<8,8,4>", 22, 1000, 0),
]


@pytest.mark.parametrize("scheme,log2n,L,mean_run", CASES)
def test_every_mode_decodes_what_mode_a_decodes(scheme, log2n, L, mean_run):
    import modes
    from oracle import oracle
    n = 1 << log2n
    fields, work = _column(scheme, n, L, mean_run)
    stream = torch.cuda.current_stream().cuda_stream
    out = work.output
    modes.launcher(scheme, fields, out, "A")(stream)
    if scheme.startswith("delta<4,8,2>"):
        torch.cuda.synchronize()
        from oracle import oracle as o
        expect = o.delta(work.deltas.cpu().numpy().view(np.uint16), work.anchors.cpu().numpy().view(np.uint64), L)
        assert (expect == out.cpu().numpy().view(np.uint64)).all()
    else:
        work.verify(oracle)
    reference = out.clone()
    for mode in MODES:
        out.fill_(-1)
        modes.launcher(scheme, fields, out, mode)(stream)
        torch.cuda.synchronize()
        assert bool((out == reference).all()), "%s, segments of %d: mode %s differs from mode A" % (scheme, L, mode)


def test_a_parameter_of_the_wrong_size_is_refused_through_the_base_type():
    """cuda::kernel::enqueue_launch(const cuda::kernel_t&, ...) re-types the values for the wrapper's typed launch: a value whose size is not the
    parameter's in its place is an error, not a silent misread (the reference reinterprets the kernel's address with the caller's types)."""
    import subprocess
    program = os.path.join(ROOT, "tests", "cpp", "bin", "packed_parameters")
    if not os.path.exists(program):
        import __graft_entry__ as entry
        entry.build_cpp_test_tools()
    result = subprocess.run([program], capture_output=True, text=True, timeout=120)
    assert result.returncode == 0 and "packed_parameters OK" in result.stdout, result.stdout + result.stderr


def test_attributes_cover_every_kernel_the_configuration_may_be_launched_with():
    """get_device_function().attributes() of a run decoder's wrapper = the limits all its builds accept (cuda::device_function_t's alternates):
    asked of the runtime here, kernel by kernel, through tests/cpp/bin/packed_parameters --attributes"""
    import subprocess
    program = os.path.join(ROOT, "tests", "cpp", "bin", "packed_parameters")
    if not os.path.exists(program):
        import __graft_entry__ as entry
        entry.build_cpp_test_tools()
    result = subprocess.run([program, "--attributes"], capture_output=True, text=True, timeout=120)
    assert result.returncode == 0 and "attributes OK" in result.stdout, result.stdout + result.stderr


def test_scratch_pool_reserve_forget_trim():
    """include/giddy/stream_scratch.hpp behind giddy_reserve_scratch / giddy_release_scratch / giddy_trim: a long-segment decode on a stream
    that reserved its scratch allocates nothing; forgetting the stream and trimming gives the memory back"""
    import modes
    from libgiddy_amd import registered
    registered.trim()
    assert registered.scratch_held_bytes() == 0
    n, L = 1 << 22, 1 << 20
    fields, work = _column("rle<4,4,2>", n, L, 32)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        registered.reserve_scratch(side.cuda_stream, 8 << 20)
        held = registered.scratch_held_bytes()
        assert held >= 8 << 20
        run = modes.launcher("rle<4,4,2>", fields, work.output, "A")
        for _ in range(3):
            run(side.cuda_stream)
        assert registered.scratch_held_bytes() == held, "a launch on a stream with reserved scratch allocated more"
        side.synchronize()
        from oracle import oracle
        work.verify(oracle)
        registered.release_scratch(side.cuda_stream)
    assert registered.trim() == held
    assert registered.scratch_held_bytes() == 0


def test_prepared_launches_keep_their_map_and_take_updates():
    """giddy_prepare / giddy_prepared_update / giddy_enqueue_prepared: the argument map is built once, a loop rewrites the entries that change — here the
    two halves of a Dictionary column decoded by ONE prepared launch whose pointers and length are moved between the launches."""
    from libgiddy_amd import registered as r, schemes
    n = 1 << 20
    keys = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")
    entries = (torch.arange(256, dtype=torch.int32, device="cuda") * 2654435 + 17)
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    kernel = schemes.kernel(schemes.key_dictionary(4, 4, 1))
    half = n // 2
    config = kernel.resolve_launch_configuration(r.CURRENT_DEVICE, {"num_dictionary_entries": r.size_t(256), "data_length": r.size_t(half)})
    prepared = kernel.prepare({"decompressed": r.pointer(out, "u32"), "compressed_input": r.const_pointer(keys, "u8"), "dictionary_entries": r.const_pointer(entries, "u32"),
                               "length": r.size_type_by_index_size(4)(half), "num_dictionary_entries": r.size_type_by_index_size(1)(256)})
    stream = torch.cuda.current_stream().cuda_stream
    prepared.enqueue(stream, config)
    prepared.update({"decompressed": r.pointer(out[half:], "u32"), "compressed_input": r.const_pointer(keys[half:], "u8")})
    prepared.enqueue(stream, config)
    torch.cuda.synchronize()
    assert bool((out == entries[keys.to(torch.int64)]).all())
    with pytest.raises(Exception):
        prepared.update({"length": r.uint64(half)})   # (accepted into the map ...)
        prepared.enqueue(stream, config)              # ... and refused by the wrapper: it holds another type than the wrapper casts to


def test_every_mode_decodes_a_column_at_an_odd_element_offset():
    """Delta <4,4,1> with output and deltas one element past a 16-byte boundary: modes A, B, the base-type launch and the free launch<>() reach the wrapper's
    typed launch(), which takes the any-alignment build (kernels/decompression/delta.hpp, decompress_at_any_alignment); mode C runs the kernel of the reference's
    signature, element by element there — the same bits from all of them"""
    import modes
    from oracle import oracle as o
    device = torch.device("cuda", 0)
    n, L = 1 << 22, 4096
    deltas_all = torch.randint(0, 256, (n + 1,), dtype=torch.int32, device=device).to(torch.uint8)
    deltas = deltas_all[1:]
    sums = torch.cumsum(deltas.to(torch.int64), 0)
    anchors = torch.cat([torch.zeros(1, dtype=torch.int64, device=device), sums[L - 1:-1:L]]).add_(777).to(torch.int32)
    out_all = torch.empty(n + 1, dtype=torch.int32, device=device)
    out = out_all[1:]
    assert out.data_ptr() % 16 == 4 and deltas.data_ptr() % 4 == 1
    fields = {"compressed_input": deltas, "anchor_values": anchors, "length": n, "segment_length": L}
    stream = torch.cuda.current_stream().cuda_stream
    expect = o.delta(deltas.cpu().numpy(), anchors.cpu().numpy().view(np.uint32), L)
    for mode in ["A"] + MODES:
        out_all.fill_(-1)
        modes.launcher("delta<4,4,1>", fields, out, mode)(stream)
        torch.cuda.synchronize()
        got = out_all.cpu().numpy().view(np.uint32)
        assert got[0] == 0xFFFFFFFF, "mode %s wrote before the output" % mode
        assert (got[1:] == expect).all(), "mode %s" % mode

"""The measured streaming ceiling of this box (tests/cpp/stream_ceiling.hip): the library's own tile loop with the identity in place of a decoder,
as a plain copy (4 B in : 4 B out) and in the decoders' traffic shapes (1 B : 4 B Dictionary<4,4,1>, 2 B : 4 B Frame of reference, 2 B : 8 B Delta / DZB-fixed <4,8,2>).  GB/s of bytes read + written, by HIP events on the launch stream."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "cpp", "bin", "libstream_ceiling.so")
_shim = None


def build_shim():
    os.makedirs(os.path.dirname(SHIM), exist_ok=True)
    subprocess.check_call(["hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "stream_ceiling.hip"), "-o", SHIM])


def shim():
    """never a compiler from a process that has initialised the GPU: absent -> FileNotFoundError (__graft_entry__.build() builds it)"""
    global _shim
    if _shim is None:
        if not os.path.exists(SHIM):
            raise FileNotFoundError("tests/cpp/bin/libstream_ceiling.so is not built (python -c 'import __graft_entry__ as g; g.build_cpp_test_tools()')")
        _shim = ctypes.CDLL(SHIM)
        for f in (_shim.stream_copy, _shim.stream_widen, _shim.stream_widen_2_to_4, _shim.stream_widen_2_to_8):
            f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int]
            f.restype = ctypes.c_int
    return _shim


# (reserved LDS per workgroup -> resident workgroups per CU; store policy) the loop is tried with: the decoders' own launch shapes
# (0: whatever fits — 8 workgroups of 256 threads a CU; 26 624: 6 a CU, the headline's launch; 39 936: 4; 53 248: 3; 79 872: 2 — a copy's reads and writes want FEWER
# workgroups in flight than a decoder's: 5.6 TB/s unthrottled, 6.1 at 3 a CU)
SHAPES = [(0, 1), (0, 2), (26624, 2), (39936, 2), (53248, 2), (79872, 2)]


def measure(log2n=30, repetitions=10):
    """{"copy_GBps", "widen_1_to_4_GBps", ...}: for each traffic shape the BEST of the launch shapes the decoders use (residency throttle x store policy),
    each the median of `repetitions` launches over 2^log2n elements"""
    import torch
    s = shim()
    n = 1 << log2n
    stream = torch.cuda.current_stream().cuda_stream
    results = {}
    for name, fn, in_bytes, out_dtype in (("copy", s.stream_copy, 4, torch.int32), ("widen_1_to_4", s.stream_widen, 1, torch.int32),
                                          ("widen_2_to_4", s.stream_widen_2_to_4, 2, torch.int32), ("widen_2_to_8", s.stream_widen_2_to_8, 2, torch.int64)):
        out = torch.empty(n, dtype=out_dtype, device="cuda")
        out_bytes = out.element_size()
        source = torch.randint(0, 127, (n * in_bytes,), dtype=torch.int8, device="cuda")
        by_shape = {}
        for lds, stores in SHAPES:
            for _ in range(3):
                status = fn(out.data_ptr(), source.data_ptr(), n, stream, lds, stores)
                assert status == 0, status
            torch.cuda.synchronize()
            events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(repetitions)]
            for begin, end in events:
                begin.record(); fn(out.data_ptr(), source.data_ptr(), n, stream, lds, stores); end.record()
            torch.cuda.synchronize()
            ms = sorted(b.elapsed_time(e) for b, e in events)[repetitions // 2]
            by_shape["lds=%d,stores=%s" % (lds, "sc1_nt" if stores == 2 else "nt")] = round((in_bytes + out_bytes) * n / ms / 1e6, 1)
        # (the identity is checked once: what was timed is a copy)
        narrow = {1: torch.int8, 2: torch.int16, 4: torch.int32}[in_bytes]
        expected = source.view(narrow)[: 1 << 20].to(out_dtype)
        assert bool((out[: 1 << 20] == expected).all())
        best = max(by_shape, key=by_shape.get)
        results[name + "_GBps"] = by_shape[best]
        results[name + "_best_shape"] = best
        results[name + "_by_shape"] = by_shape
        del source, out
    torch.cuda.empty_cache()
    return results


if __name__ == "__main__":
    print(measure())

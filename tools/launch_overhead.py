"""What one launch costs on the HOST (development aid): the type-erased path — Python -> ctypes -> giddy_enqueue_launch -> argument map ->
wrapper -> hipLaunchKernel — timed over many asynchronous launches of a tiny column, and the same wrapper call from C++ alone (tests/cpp).
    python tools/launch_overhead.py"""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from libgiddy_amd import schemes, registered as r, _native as native

dev = torch.device("cuda", 0)
n = 4096
keys = torch.zeros(n, dtype=torch.uint8, device=dev); entries = torch.arange(256, dtype=torch.int32, device=dev); out = torch.empty(n, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
config = schemes.dictionary(out, keys, entries, n, 256, stream=stream)
torch.cuda.synchronize()


def per_call(fn, calls=20000):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(calls): fn()
    host = time.perf_counter() - t
    torch.cuda.synchronize()
    return host / calls * 1e6, (time.perf_counter() - t) / calls * 1e6


host_us, total_us = per_call(lambda: schemes.dictionary(out, keys, entries, n, 256, stream=stream, launch_config=config))
print("schemes.dictionary(...)  [Python: key cache, argument constructors, marshalling, ctypes, C++ map + launch]: %.2f us of host time per call (%.2f us per call until the queue is drained)" % (host_us, total_us))
k = schemes.kernel(schemes.key_dictionary(4, 4, 1))
arguments = {"decompressed": r.pointer(out, "u32"), "compressed_input": r.const_pointer(keys, "u8"), "dictionary_entries": r.const_pointer(entries, "u32"),
             "length": r.size_type_by_index_size(4)(n), "num_dictionary_entries": r.size_type_by_index_size(1)(256)}
host_us, total_us = per_call(lambda: k.enqueue_launch(stream, config, arguments))
print("kernel_t.enqueue_launch  [marshalling + ctypes + C++]: %.2f us per call (%.2f)" % (host_us, total_us))
array, keep = r._marshal(arguments)
c_config = config._c()
handle, lib, n_args = k._handle, native.lib, len(arguments)
host_us, total_us = per_call(lambda: lib.giddy_enqueue_launch(handle, ctypes.c_void_p(stream), ctypes.byref(c_config), array, n_args))
print("giddy_enqueue_launch     [ctypes + C++: tagged arguments -> std::unordered_map<string, any> -> wrapper -> hipLaunchKernel]: %.2f us per call (%.2f)" % (host_us, total_us))
x = torch.empty(1, device=dev)
host_us, total_us = per_call(lambda: x.add_(1))
print("for scale: torch's x.add_(1) on a 1-element tensor: %.2f us per call (%.2f)" % (host_us, total_us))

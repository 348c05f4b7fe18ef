"""Which headers a decoder kernel is made of, and a fingerprint of them: a committed HBM-traffic measurement
(profiles/hbm_traffic.json) is only quoted by bench.py while the kernel it was taken on has not changed."""
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INCLUDE = os.path.join(ROOT, "include", "giddy")
SHARED = ["common.hpp", "on_device/wave64.hpp", "on_device/stream_tiles.hpp", "on_device/sorted_patches.hpp", "kernels/tuning.hpp"]
BY_KERNEL = {   # substring of the kernel name -> its own headers
    "dictionary::": ["kernels/decompression/dictionary.hpp"],
    "frame_of_reference::": ["kernels/decompression/frame_of_reference.hpp", "functors/model_functors.hpp"],
    "delta::": ["kernels/decompression/delta.hpp"],
    "run_length_encoding::": ["kernels/decompression/run_length_encoding.hpp", "kernels/decompression/run_expansion.hpp"],
    "run_position_encoding::": ["kernels/decompression/run_position_encoding.hpp", "kernels/decompression/run_expansion.hpp"],
    "variable_width::": ["kernels/decompression/discard_zero_bytes_variable.hpp"],
    "fixed_width::": ["kernels/decompression/discard_zero_bytes_fixed.hpp"],
    "model::": ["kernels/decompression/model.hpp", "functors/model_functors.hpp"],
    "incidence_bitmaps::": ["kernels/decompression/incidence_bitmaps.hpp"],
}


def files_of(kernel):
    own = [files for key, files in BY_KERNEL.items() if kernel.startswith(key) or ("::" + key) in ("::" + kernel)]
    return sorted(set(SHARED + (own[0] if own else [])))


def code_of(text):
    """a header's text without comments and with white space collapsed: what the compiler sees of it, near enough"""
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return re.sub(r"\s+", " ", text).strip()


def fingerprint(kernel):
    """sha256 (16 hex digits) over the CODE of the headers `kernel` is compiled from (comments and spacing do not count)"""
    h = hashlib.sha256()
    for name in files_of(kernel):
        path = os.path.join(INCLUDE, name)
        h.update(name.encode())
        h.update(code_of(open(path, encoding="utf-8").read()).encode() if os.path.exists(path) else b"(absent)")
    return h.hexdigest()[:16]

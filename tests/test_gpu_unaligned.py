"""-m gpu: base pointers that are element-aligned but NOT 16-byte aligned (a column handed over at an odd
element offset): every decoder must fall back to its element-wise paths and still be bit-exact, without
touching the element before the output or anything past its end (guard bands)."""
import numpy as np
import pytest

import cases
from libgiddy_amd import formats

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _odd(g, array):
    """device copy of `array` placed one element past a 16-byte boundary; returns (keepalive, address)"""
    array = np.ascontiguousarray(array)
    item = array.dtype.itemsize
    buf = g.dev(np.concatenate([np.zeros(1, array.dtype), array.reshape(-1)]))
    return buf, buf.data_ptr() + item


def _odd_out(g, n, dtype):
    item = np.dtype(dtype).itemsize
    buf = g.dev_empty(n + 1, dtype)
    return buf, buf.data_ptr() + item


def _read(g, buf, dtype, n):
    full = g.host(buf, dtype, n + 1)
    assert full[0] == np.frombuffer(bytes([0xCD] * np.dtype(dtype).itemsize), dtype=dtype)[0], "wrote before the output"
    return full[1:]


def test_unaligned_bases():
    import gpu_util as g
    from libgiddy_amd import schemes
    st = g.stream()
    n = 70001

    f, truth = cases.delta_case(n, 4096, np.uint64, np.uint16, seed=1)
    keep_in, in_ptr = _odd(g, f["compressed_input"])
    out, out_ptr = _odd_out(g, n, np.uint64)
    schemes.delta(out_ptr, in_ptr, g.dev(f["anchor_values"]), n, 4096, stream=st)
    assert (_read(g, out, np.uint64, n) == truth).all()

    f = cases.dictionary_case(n, 256, np.uint32, np.uint8, seed=2)
    keep_in, in_ptr = _odd(g, f["compressed_input"])
    out, out_ptr = _odd_out(g, n, np.uint32)
    schemes.dictionary(out_ptr, in_ptr, g.dev(f["dictionary_entries"]), n, 256, stream=st)
    assert (_read(g, out, np.uint32, n) == f["dictionary_entries"][f["compressed_input"].astype(np.int64)]).all()

    col = cases.runs_column(n, 32, np.uint32, seed=3)
    f = formats.run_length_encoding(col, 4096, np.uint16)
    out, out_ptr = _odd_out(g, n, np.uint32)
    keep_l, len_ptr = _odd(g, f["run_lengths"])
    keep_d, data_ptr = _odd(g, f["run_data"])
    schemes.run_length_encoding(out_ptr, data_ptr, len_ptr, g.dev(f["position_anchors"]), g.dev(f["intra_run_anchor_offsets"]), 4096,
                                f["num_segments"], f["num_element_runs"], n, stream=st)
    assert (_read(g, out, np.uint32, n) == col).all()
    f = formats.run_position_encoding(col, 4096, np.uint16, True)
    out, out_ptr = _odd_out(g, n, np.uint32)
    keep_p, pos_ptr = _odd(g, f["run_start_positions"])
    schemes.run_position_encoding(out_ptr, g.dev(f["run_data"]), pos_ptr, g.dev(f["position_anchors"]), 4096, f["num_anchors"],
                                  f["num_element_runs"], n, stream=st)
    assert (_read(g, out, np.uint32, n) == col).all()

    colv, sizes = cases.dzbv_column(n, np.uint64, 1, 2, seed=4)
    f = formats.discard_zero_bytes_variable(colv, 1024, 1, 2, np.uint32, 8, element_sizes=sizes)
    out, out_ptr = _odd_out(g, n, np.uint64)
    keep_b, bytes_ptr = _odd(g, f["compressed_data"])          # compressed bytes at an odd address
    schemes.discard_zero_bytes_variable(out_ptr, bytes_ptr, g.dev(f["packed_element_sizes"]), g.dev(f["position_anchors"]), 1024, n, 1, 2, stream=st)
    assert (_read(g, out, np.uint64, n) == colv).all()

    pool = cases.rng(5).choice(1 << 31, size=9, replace=False).astype(np.uint32)
    colb = pool[cases.rng(6).integers(0, 9, size=32 * 2001)]
    f = formats.incidence_bitmaps(colb, 2)
    out, out_ptr = _odd_out(g, len(colb), np.uint32)
    schemes.incidence_bitmaps(out_ptr, g.dev(f["incidence_bitmaps"]), g.dev(f["dictionary_entries"]), f["bitmap_length"], f["num_bitmaps"], stream=st)
    assert (_read(g, out, np.uint32, len(colb)) == colb).all()

    c = cases.rng(7).integers(0, 65535, size=n, endpoint=True).astype(np.uint16)
    keep_in, in_ptr = _odd(g, c)
    out, out_ptr = _odd_out(g, n, np.uint64)
    schemes.discard_zero_bytes_fixed(out_ptr, in_ptr, n, uncompressed_size=8, compressed_size=2, stream=st)
    assert (_read(g, out, np.uint64, n) == c).all()

    out, out_ptr = _odd_out(g, n, np.int32)
    schemes.model(out_ptr, np.array([7, -3], np.int32), n, stream=st)
    assert (_read(g, out, np.int32, n) == (7 - 3 * np.arange(n)).astype(np.int32)).all()

"""Helpers for the -m gpu parity tests: numpy <-> device byte buffers (torch is only the allocator).

Output buffers come with guard bands: dev_empty() hands out the middle of a larger allocation whose
first and last GUARD bytes hold a known pattern, and host() checks that the kernel left them alone —
so every parity test also detects out-of-bounds writes next to the output."""
import numpy as np
import torch

GUARD = 512
_PATTERN = 0xA7
_guarded = {}   # data_ptr of the handed-out slice -> (whole allocation, payload bytes)


def dev(array):
    """host numpy array -> device byte buffer (torch.uint8), 256-byte aligned by the allocator"""
    array = np.ascontiguousarray(array)
    if array.nbytes == 0:
        return torch.zeros(16, dtype=torch.uint8, device="cuda")
    return torch.from_numpy(array.reshape(-1).view(np.uint8).copy()).cuda()


def dev_empty(n, dtype, fill=0xCD):
    """device output buffer of n elements, pre-filled with a poison byte, between two guard bands"""
    nbytes = max(int(n) * np.dtype(dtype).itemsize, 16)
    padded = (nbytes + 15) // 16 * 16
    whole = torch.full((padded + 2 * GUARD,), _PATTERN, dtype=torch.uint8, device="cuda")
    payload = whole[GUARD: GUARD + padded]
    payload.fill_(fill)
    if len(_guarded) > 256:
        _guarded.clear()
    _guarded[payload.data_ptr()] = (whole, padded)
    return payload


def host(buffer, dtype, n):
    torch.cuda.synchronize()
    dtype = np.dtype(dtype)
    entry = _guarded.get(buffer.data_ptr())
    if entry is not None:
        whole, padded = entry
        guards = torch.cat([whole[:GUARD], whole[GUARD + padded:]])
        assert bool((guards == _PATTERN).all()), "the kernel wrote outside its output buffer"
        used = int(n) * dtype.itemsize
        if used >= 16:   # bytes past the last element (padding to 16) must keep the poison value
            assert bool((buffer[used:] == buffer[used:].new_full((1,), 0xCD)).all()) or padded == used, "write past the last element"
    return buffer.cpu().numpy()[: int(n) * dtype.itemsize].view(dtype).copy()


def stream():
    return torch.cuda.current_stream().cuda_stream

"""Helpers for the -m gpu parity tests: numpy <-> device byte buffers (torch is only the allocator)."""
import numpy as np
import torch


def dev(array):
    """host numpy array -> device byte buffer (torch.uint8), 256-byte aligned by the allocator"""
    array = np.ascontiguousarray(array)
    if array.nbytes == 0:
        return torch.zeros(16, dtype=torch.uint8, device="cuda")
    return torch.from_numpy(array.reshape(-1).view(np.uint8).copy()).cuda()


def dev_empty(n, dtype, fill=0xCD):
    """device output buffer of n elements, pre-filled with a poison byte"""
    nbytes = max(int(n) * np.dtype(dtype).itemsize, 16)
    return torch.full((nbytes,), fill, dtype=torch.uint8, device="cuda")


def host(buffer, dtype, n):
    torch.cuda.synchronize()
    dtype = np.dtype(dtype)
    return buffer.cpu().numpy()[: int(n) * dtype.itemsize].view(dtype).copy()


def stream():
    return torch.cuda.current_stream().cuda_stream

"""TEST INFRASTRUCTURE ONLY: Engine backend that places buffers in host memory and calls the host-compiled kernels
(tests/hipemu).  Lets `-m "not gpu"` tests run the product's plan-materialisation code end to end on tiny inputs."""
import ctypes as C

import numpy as np
import torch

from yolov5_amd import _lib

from .emu import aligned, emu

_NP = {torch.float16: np.float16, torch.float32: np.float32, torch.uint8: np.uint8}


class EmuBackend:
    def __init__(self):
        self.lib = emu()

    def empty(self, shape, dtype):
        return aligned(shape, _NP[dtype], 0)

    def from_torch(self, t):
        a = aligned(tuple(t.shape), _NP[t.dtype])
        a[...] = t.detach().cpu().numpy()
        return a

    def ptr(self, h):
        return h.ctypes.data

    def to_torch(self, h):
        return torch.from_numpy(np.array(h))

    def view_torch(self, h):
        return torch.from_numpy(h)  # shares memory with the emulated device buffer

    def zero_(self, h):
        h[...] = 0

    def stream(self):
        return None

    def input(self, x):
        x = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else x
        code = {np.dtype(np.float16): _lib.Y5_F16, np.dtype(np.float32): _lib.Y5_F32, np.dtype(np.uint8): _lib.Y5_U8}[x.dtype]
        a = aligned(x.shape, x.dtype)
        a[...] = x
        return a, a.ctypes.data, code

"""TEST INFRASTRUCTURE ONLY: Engine backend that places buffers in host memory and calls the host-compiled kernels
(tests/hipemu).  Lets `-m "not gpu"` tests run the product's plan-materialisation code end to end on tiny inputs."""
import ctypes as C

import numpy as np
import torch

from yolov5_amd import _lib

from .emu import aligned, emu

_NP = {torch.float16: np.float16, torch.float32: np.float32, torch.uint8: np.uint8, torch.float64: np.float64}


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

    def assign(self, h, t):
        h[...] = t.detach().cpu().numpy()

    def stream(self):
        return None

    def input(self, x):
        x = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else x
        code = {np.dtype(np.float16): _lib.Y5_F16, np.dtype(np.float32): _lib.Y5_F32, np.dtype(np.uint8): _lib.Y5_U8}[x.dtype]
        a = aligned(x.shape, x.dtype)
        a[...] = x
        return a, a.ctypes.data, code


class TorchEmuBackend:
    """Same role, buffers are torch CPU tensors (64-byte aligned by torch's allocator): what the `_lib.use_test_library` seam installs so
    that `model(x)`, ComputeLoss, HipSGD, the train / detect loops ... run unchanged on CPU tensors against the host-compiled kernels."""

    direct = True

    def __init__(self):
        self.lib = emu()

    def empty(self, shape, dtype):
        n = int(np.prod(shape)) if len(shape) else 1
        es = torch.empty((), dtype=dtype).element_size()
        with torch.inference_mode(False):      # as _HipBackend.empty: ordinary tensors also under torch.inference_mode()
            raw = torch.zeros(n * es + 256, dtype=torch.uint8)
            off = (-raw.data_ptr()) % 256      # the C-ABI wants 256-byte aligned workspaces / 16-byte aligned tensors
            return raw[off:off + n * es].view(dtype).reshape(tuple(shape))

    def from_torch(self, t):
        return t.detach().cpu().contiguous().clone()

    def ptr(self, h):
        return h.data_ptr()

    def to_torch(self, h):
        return h

    def view_torch(self, h):
        return h

    def zero_(self, h):
        h.zero_()

    def assign(self, h, t):
        h.copy_(t)

    def stream(self):
        return None

    def input(self, x):
        code = {torch.float16: _lib.Y5_F16, torch.float32: _lib.Y5_F32, torch.uint8: _lib.Y5_U8}.get(x.dtype)
        if code is None:
            raise TypeError(f"unsupported input dtype {x.dtype}")
        x = x.contiguous()
        return x, x.data_ptr(), code


def install():
    """Route the whole yolov5_amd Python layer to the host-compiled kernels for CPU tensors (tests only)."""
    _lib.use_test_library(emu(), TorchEmuBackend)


def uninstall():
    _lib.use_test_library(None, None)

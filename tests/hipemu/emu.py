"""TEST INFRASTRUCTURE ONLY: ctypes handle on liby5emu.so (product kernel sources compiled for the host on the
fiber-based HIP emulator).  Lets `-m "not gpu"` tests exercise the real kernel code on tiny shapes."""
import ctypes as C
import os
import subprocess

import numpy as np

from yolov5_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
_emu = None


def emu():
    global _emu
    if _emu is None:
        import fcntl

        os.makedirs(os.path.join(HERE, "_build"), exist_ok=True)
        with open(os.path.join(HERE, "_build", ".lock"), "w") as lk:  # xdist workers: one (incremental) build at a time
            fcntl.flock(lk, fcntl.LOCK_EX)
            out = subprocess.run([os.path.join(HERE, "build.sh")], capture_output=True, text=True)
        if out.returncode != 0:
            raise RuntimeError("hipemu build failed:\n" + out.stdout + out.stderr)
        _emu = _lib.bind(C.CDLL(os.path.join(HERE, "_build", "liby5emu.so")))
    return _emu


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def aligned(shape, dtype, fill=None):
    """numpy array whose data pointer is 256-byte aligned (the C-ABI requires 16 B)."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    raw = np.zeros(n + 256, dtype=np.uint8)
    off = (-raw.ctypes.data) % 256
    a = raw[off:off + n].view(dtype).reshape(shape)
    if fill is not None:
        a[...] = fill
    return a

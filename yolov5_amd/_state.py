"""Process-wide bookkeeping shared by the engines.

`weights_epoch` counts events that change model weights or BatchNorm running statistics WITHOUT going through torch's
tensor version counters: the fused optimizer / EMA kernels (csrc/optim.hip) and the train-mode BatchNorm kernel
(csrc/bn_kernels.h) write through raw device pointers.  Every cached eval plan remembers the epoch its packed filters were
built from; `BaseModel._forward_once` re-packs them (Engine.refresh_weights) when the epoch moved -- the reference always
runs on the live parameters (models/yolo.py:160-170), so must we.
"""
weights_epoch = 0


def bump_weights_epoch():
    global weights_epoch
    weights_epoch += 1
    return weights_epoch


cu_budget = 0  # CUs the persistent kernels currently size their grids for (0 = all); mirrors csrc/core.hip's value for host-side grid arithmetic


def set_cu_budget(lib, n):
    global cu_budget
    n = max(int(n), 0)
    if n != cu_budget:
        rc = lib.y5_set_cu_budget(n)
        if rc != 0:
            raise RuntimeError("y5_set_cu_budget failed")
        cu_budget = n

"""Device letterbox at detect.py scale: 64 frames of 1280x720 uint8 -> (64, 3, 640, 640) fp16 (resize + border + CHW + RGB + /255)
in one launch, timed with events on the launch stream; the CPU oracle (restated cv2 path, numpy) timed beside it on a bounded
sample.  Prints one JSON line with the HBM roofline of the launch (algorithmic bytes = source bytes + output bytes)."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import thirdparty as tp  # noqa: E402  (checker + CPU baseline only)
from yolov5_amd.augmentations import letterbox_batch, letterbox_geometry  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B, h0, w0 = 64, 720, 1280
    rng = np.random.default_rng(0)
    frames = [rng.integers(0, 256, (h0, w0, 3), dtype=np.uint8) for _ in range(4)]
    ims = [torch.from_numpy(frames[i % 4]).to(dev) for i in range(B)]
    x, _ = letterbox_batch(ims, 640, auto=False, dtype=torch.float16, swap_rb=True)
    torch.cuda.synchronize()
    g = letterbox_geometry((h0, w0), 640, auto=False)
    t0 = time.perf_counter()
    refs = []
    for f in frames[:2]:
        r = tp.cv2_resize(f, g["new_unpad"], interpolation=1)
        r = tp.cv2_copy_make_border(r, g["top"], g["bottom"], g["left"], g["right"], 0, value=(114, 114, 114))
        refs.append(torch.from_numpy(np.ascontiguousarray(r.transpose(2, 0, 1)[::-1])).half() / 255)
    cpu_ms = (time.perf_counter() - t0) * 1e3 / 2
    assert torch.equal(x[0].cpu(), refs[0]) and torch.equal(x[1].cpu(), refs[1])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    for _ in range(3):
        letterbox_batch(ims, 640, auto=False, dtype=torch.float16, swap_rb=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        letterbox_batch(ims, 640, auto=False, dtype=torch.float16, swap_rb=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    kernel_ms = launch_only_ms(ims, g, x)
    # rows of the source the bilinear taps touch: all of them at this ratio (scale 2.0 -> area path reads every pixel)
    src_bytes = B * h0 * w0 * 3
    out_bytes = B * 3 * 640 * 640 * 2
    print(json.dumps({"op": "letterbox_batch", "frames": B, "src": [h0, w0], "dst": [640, 640], "ms_per_batch": round(ms, 4),
                      "images_per_s": round(B / ms * 1e3, 1), "algorithmic_GB": round((src_bytes + out_bytes) / 1e9, 4),
                      "kernel_ms": round(kernel_ms, 4), "kernel_GB_per_s": round((src_bytes + out_bytes) / kernel_ms / 1e6, 1),
                      "kernel_hbm_frac_of_8TBps": round((src_bytes + out_bytes) / kernel_ms / 1e6 / 8000, 3),
                      "cpu_oracle_ms_per_image": round(cpu_ms, 2), "cpu_cores": 1}))


def launch_only_ms(ims, g, out, n=50):
    """The launch alone (job table resident), HIP events on the launch stream."""
    import ctypes as C

    from yolov5_amd import _lib

    jobs = (_lib.LetterboxJob * len(ims))()
    for j, im in zip(jobs, ims):
        j.src, j.h0, j.w0, j.stride = im.data_ptr(), im.shape[0], im.shape[1], im.stride(0)
        j.nw, j.nh, j.top, j.left = g["new_unpad"][0], g["new_unpad"][1], g["top"], g["left"]
    table = torch.frombuffer(bytearray(jobs), dtype=torch.uint8).to(out.device)
    lib = _lib.lib()
    st = C.c_void_p(torch.cuda.current_stream(out.device).cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    args = (C.c_void_p(table.data_ptr()), len(ims), out.shape[2], out.shape[3], 114, 1, C.c_void_p(out.data_ptr()), _lib.Y5_F16, 1, 1, st)
    for _ in range(3):
        lib.y5_letterbox_batch(*args)
    e0.record()
    for _ in range(n):
        lib.y5_letterbox_batch(*args)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def second():
    """1920x1080 -> 640x360 (+ border): the fixed-point bilinear path (scale 3: two of every three source rows are read)."""
    dev = torch.device("cuda:0")
    B, h0, w0 = 64, 1080, 1920
    rng = np.random.default_rng(1)
    frames = [rng.integers(0, 256, (h0, w0, 3), dtype=np.uint8) for _ in range(2)]
    ims = [torch.from_numpy(frames[i % 2]).to(dev) for i in range(B)]
    x, _ = letterbox_batch(ims, 640, auto=False, dtype=torch.float16, swap_rb=True)
    g = letterbox_geometry((h0, w0), 640, auto=False)
    r = tp.cv2_copy_make_border(tp.cv2_resize(frames[0], g["new_unpad"], interpolation=1), g["top"], g["bottom"], g["left"], g["right"], 0,
                                value=(114, 114, 114))
    assert torch.equal(x[0].cpu(), torch.from_numpy(np.ascontiguousarray(r.transpose(2, 0, 1)[::-1])).half() / 255)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        letterbox_batch(ims, 640, auto=False, dtype=torch.float16, swap_rb=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(json.dumps({"op": "letterbox_batch", "frames": B, "src": [h0, w0], "dst": [640, 640], "path": "bilinear", "ms_per_batch": round(ms, 4),
                      "kernel_ms": round(launch_only_ms(ims, g, x), 4),
                      "images_per_s": round(B / ms * 1e3, 1)}))


if __name__ == "__main__":
    main()
    second()

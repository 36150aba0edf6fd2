"""GPU parity of the device letterbox (HIP y5_letterbox_batch through yolov5_amd.augmentations) vs the fixture produced by the
reference's own `letterbox` (tests/golden/letterbox.npz; cv2 arithmetic restated, see tests/test_emu_letterbox.py) and, at
detect.py sizes, vs the oracle's restated cv2 path."""
import os

import numpy as np
import pytest
import torch

from oracle import thirdparty as tp
from oracle.make_golden import LETTERBOX_CASES, letterbox_image

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "letterbox.npz"))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("name", list(LETTERBOX_CASES))
def test_letterbox_vs_reference_golden(name, dev):
    from yolov5_amd.augmentations import letterbox, letterbox_batch

    _, kw = LETTERBOX_CASES[name]
    im = torch.from_numpy(letterbox_image(name)).to(dev)
    out, ratio, pad = letterbox(im, **kw)
    meta = G[name + "_meta"]
    assert out.dtype == torch.uint8 and np.array_equal(out.cpu().numpy(), G[name])
    assert tuple(ratio) == (meta[0], meta[1]) and (float(pad[0]), float(pad[1])) == (meta[2], meta[3])
    if not kw.get("auto", True):
        x, shapes = letterbox_batch([im, im], kw["new_shape"], auto=False, scaleFill=kw.get("scaleFill", False), scaleup=kw.get("scaleup", True),
                                    dtype=torch.float16, swap_rb=True)
        want = torch.from_numpy(np.ascontiguousarray(G[name].transpose(2, 0, 1)[::-1])).half() / 255
        assert torch.equal(x[0].cpu(), want) and torch.equal(x[1].cpu(), want)
        assert shapes[0][0] == tuple(im.shape[:2]) and shapes[0][1][0] == tuple(ratio) and shapes[0][1][1] == tuple(pad)


def test_letterbox_batch_detect_sizes_vs_oracle(dev):
    """A mixed batch of camera-sized frames -> (B, 3, 640, 640) fp16, against letterbox restated on the CPU (oracle cv2 layer)."""
    from yolov5_amd.augmentations import letterbox_batch, letterbox_geometry

    rng = np.random.default_rng(8)
    sizes = [(1080, 1920), (720, 1280), (1280, 960), (480, 640), (640, 640), (375, 500)]
    ims = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    x, shapes = letterbox_batch([torch.from_numpy(i).to(dev) for i in ims], 640, auto=False, dtype=torch.float16, swap_rb=True)
    assert x.shape == (len(ims), 3, 640, 640)
    xc = x.cpu()
    for i, im in enumerate(ims):
        g = letterbox_geometry(im.shape[:2], 640, auto=False)
        r = im if (im.shape[1], im.shape[0]) == g["new_unpad"] else tp.cv2_resize(im, g["new_unpad"], interpolation=1)
        ref = tp.cv2_copy_make_border(r, g["top"], g["bottom"], g["left"], g["right"], 0, value=(114, 114, 114))
        want = torch.from_numpy(np.ascontiguousarray(ref.transpose(2, 0, 1)[::-1])).half() / 255
        assert torch.equal(xc[i], want), i
        assert shapes[i] == (tuple(im.shape[:2]), (g["ratio"], g["pad"]))


def test_letterbox_needs_gpu_uint8_hwc(dev):
    from yolov5_amd.augmentations import letterbox

    with pytest.raises(RuntimeError):
        letterbox(torch.zeros(8, 8, 3, dtype=torch.uint8))
    with pytest.raises(ValueError):
        letterbox(torch.zeros(8, 8, 3, device=dev))
    with pytest.raises(NotImplementedError):
        letterbox(torch.zeros(8, 8, 3, dtype=torch.uint8, device=dev), color=(1, 2, 3))

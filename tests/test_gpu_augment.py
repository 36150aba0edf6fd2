"""GPU (-m gpu): y5_mosaic_batch on the MI355X against the reference-generated batches of tests/golden/augment.npz (bit-identical
pixels and labels, like the emulator twin tests/test_emu_augment.py), and a full-size batch (64 x 640^2 from 1280x720 frames)
against the oracle restatement on a few images + timing."""
import os
import random
import time

import numpy as np
import pytest
import torch

from oracle import augment_oracle as ao

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "augment.npz"))
HYP = dict(ao.HYP_AUG, degrees=5.0, shear=2.0, flipud=0.3)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_mosaic_batch_matches_reference_golden(seed):
    from yolov5_amd.dataloaders import draw_sample, mosaic_batch

    dev = torch.device("cuda:0")
    s = int(G["s"])
    ims, labs = ao.synthetic_dataset(6, seed=3)
    ims_t = [torch.from_numpy(im).to(dev) for im in ims]
    labs = [lb.astype(np.float32) for lb in labs]
    draws = []
    for index in (seed % 6, (seed + 3) % 6):
        random.seed(seed * 10 + index)
        np.random.seed(seed * 10 + index)
        draws.append(draw_sample(index, 6, s, HYP))
    imgs, targets = mosaic_batch(ims_t, labs, draws, s, HYP, dtype=torch.uint8)
    assert np.array_equal(imgs.cpu().numpy(), G[f"img{seed}"])
    np.testing.assert_array_equal(targets.numpy(), G[f"lab{seed}"])


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_mixed_mosaic_and_letterbox_branches_match_reference_golden(seed):
    """hyp['mosaic'] = 0.5: mosaic samples and letterbox-branch samples (dataloaders.py:710-733) in one launch, against the reference's own
    __getitem__ / collate_fn (tests/golden/augment_mixed.npz) -- the emulator twin is tests/test_emu_augment.py."""
    from yolov5_amd.dataloaders import draw_sample, mosaic_batch

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "augment_mixed.npz"))
    dev = torch.device("cuda:0")
    s = int(g["s"])
    hyp = dict(HYP, mosaic=0.5)
    ims, labs = ao.synthetic_dataset(6, seed=3)
    ims_t = [torch.from_numpy(im).to(dev) for im in ims]
    labs = [lb.astype(np.float32) for lb in labs]
    draws = []
    for index in (seed % 6, (seed + 2) % 6, (seed + 4) % 6):
        random.seed(seed * 10 + index)
        np.random.seed(seed * 10 + index)
        draws.append(draw_sample(index, 6, s, hyp))
    assert [d["mosaic"] for d in draws] == list(g[f"mosaic{seed}"])
    imgs, targets = mosaic_batch(ims_t, labs, draws, s, hyp, dtype=torch.uint8)
    assert np.array_equal(imgs.cpu().numpy(), g[f"img{seed}"])
    np.testing.assert_array_equal(targets.numpy(), g[f"lab{seed}"])


@pytest.mark.parametrize("seed", [21, 22, 23, 24])
def test_mixup_matches_reference_golden(seed):
    """hyp['mixup'] = 0.5 on the MI355X (the double-precision blend + uint8 truncation of utils/augmentations.py:231 inside the kernel) against the
    reference's own __getitem__ / collate_fn (tests/golden/augment_mixup.npz); emulator twin: tests/test_emu_augment.py."""
    from yolov5_amd.dataloaders import draw_sample, mosaic_batch

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "augment_mixup.npz"))
    dev = torch.device("cuda:0")
    s = int(g["s"])
    hyp = dict(HYP, mixup=0.5)
    ims, labs = ao.synthetic_dataset(6, seed=3)
    ims_t = [torch.from_numpy(im).to(dev) for im in ims]
    labs = [lb.astype(np.float32) for lb in labs]
    draws = []
    for index in (seed % 6, (seed + 2) % 6, (seed + 4) % 6):
        random.seed(seed * 10 + index)
        np.random.seed(seed * 10 + index)
        draws.append(draw_sample(index, 6, s, hyp))
    imgs, targets = mosaic_batch(ims_t, labs, draws, s, hyp, dtype=torch.uint8)
    assert np.array_equal(imgs.cpu().numpy(), g[f"img{seed}"])
    np.testing.assert_array_equal(targets.numpy(), g[f"lab{seed}"])


def test_full_size_batch_vs_oracle_and_rate():
    from yolov5_amd.dataloaders import draw_sample, mosaic_batch

    dev = torch.device("cuda:0")
    s, n, B = 640, 24, 64
    ims, labs = ao.synthetic_dataset(n, seed=5, sizes=((720, 1280), (1280, 720), (640, 640), (480, 640), (1080, 1920), (375, 500)))
    labs = [lb.astype(np.float32) for lb in labs]
    ims_t = [torch.from_numpy(im).to(dev) for im in ims]
    hyp = dict(ao.HYP_AUG)
    random.seed(11); np.random.seed(11)
    draws = [draw_sample(i % n, n, s, hyp) for i in range(B)]
    imgs, targets = mosaic_batch(ims_t, labs, draws, s, hyp, dtype=torch.uint8)
    torch.cuda.synchronize()
    for b in (0, 17, 63):                                        # the oracle takes ~1 s per 640^2 sample
        e_img, e_lab = ao.mosaic_sample(ims, labs, draws[b], s, hyp)
        assert np.array_equal(imgs[b].cpu().numpy(), e_img), b
        got = targets[targets[:, 0] == b][:, 1:].numpy()
        np.testing.assert_array_equal(got, e_lab[:, 1:])
    t0 = time.time()
    for _ in range(5):
        mosaic_batch(ims_t, labs, draws, s, hyp, dtype=torch.float16, normalize=True)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / 5 * 1e3
    print(f"\\n[mosaic] {B} x 3x{s}x{s} fp16 from {n} frames: {ms:.2f} ms per batch incl. host geometry + labels = {B / ms * 1e3:.0f} img/s")

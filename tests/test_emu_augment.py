"""CPU (emulator): the device training-input pipeline (yolov5_amd/dataloaders.py + csrc/augment.hip: mosaic, random_perspective,
HSV, flips, CHW / RGB, collate) against the batches the REFERENCE's own LoadImagesAndLabels.__getitem__ / collate_fn produced
(tests/golden/augment.npz, oracle/make_golden.py:gen_augment) with the random draws reproduced from the same seeds: pixels
bit-identical, labels identical."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import augment_oracle as ao
from tests.hipemu import backend as emu_backend

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "augment.npz"))
HYP = dict(ao.HYP_AUG, degrees=5.0, shear=2.0, flipud=0.3)


@pytest.fixture(autouse=True)
def _seam():
    emu_backend.install()
    yield
    emu_backend.uninstall()


def _dataset():
    ims, labs = ao.synthetic_dataset(6, seed=3)
    return [torch.from_numpy(im) for im in ims], [lb.astype(np.float32) for lb in labs], ims


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_mosaic_batch_matches_reference_golden(seed):
    from yolov5_amd.dataloaders import draw_sample, mosaic_batch

    s = int(G["s"])
    ims_t, labs, _ = _dataset()
    draws = []
    for index in (seed % 6, (seed + 3) % 6):
        random.seed(seed * 10 + index)            # the generators the reference consumed (oracle/make_golden.py:gen_augment)
        np.random.seed(seed * 10 + index)
        draws.append(draw_sample(index, 6, s, HYP))
    imgs, targets = mosaic_batch(ims_t, labs, draws, s, HYP, dtype=torch.uint8)
    assert imgs.shape == (2, 3, s, s) and imgs.dtype == torch.uint8
    assert np.array_equal(imgs.numpy(), G[f"img{seed}"])
    assert targets.shape == G[f"lab{seed}"].shape
    np.testing.assert_array_equal(targets.numpy(), G[f"lab{seed}"])
    # fp16 / 255 output = the a0 contract of the model (train.py:375 `.float() / 255` on the same uint8 values)
    half, _ = mosaic_batch(ims_t, labs, draws, s, HYP, dtype=torch.float16, normalize=True)
    assert torch.equal(half, (imgs.float() / 255).half())


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_mixed_mosaic_and_letterbox_branches_match_reference_golden(seed):
    """hyp['mosaic'] = 0.5 (dataloaders.py:701): samples that lose the gate take the letterbox branch (:710-733) -- one tile on an s x s canvas,
    labels shifted by the float half-borders, random_perspective with border (0, 0) -- inside the same launch as the mosaics of the batch.
    Against the reference's own __getitem__ / collate_fn (tests/golden/augment_mixed.npz): pixels bit-identical, labels identical."""
    from yolov5_amd.dataloaders import draw_sample, mosaic_batch

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "augment_mixed.npz"))
    s = int(g["s"])
    hyp = dict(HYP, mosaic=0.5)
    ims_t, labs, _ = _dataset()
    draws = []
    for index in (seed % 6, (seed + 2) % 6, (seed + 4) % 6):
        random.seed(seed * 10 + index)
        np.random.seed(seed * 10 + index)
        draws.append(draw_sample(index, 6, s, hyp))
    assert [d["mosaic"] for d in draws] == list(g[f"mosaic{seed}"])
    imgs, targets = mosaic_batch(ims_t, labs, draws, s, hyp, dtype=torch.uint8)
    assert np.array_equal(imgs.numpy(), g[f"img{seed}"])
    assert targets.shape == g[f"lab{seed}"].shape
    np.testing.assert_array_equal(targets.numpy(), g[f"lab{seed}"])


@pytest.mark.parametrize("seed", [21, 22, 23, 24])
def test_mixup_matches_reference_golden(seed):
    """hyp['mixup'] = 0.5 (dataloaders.py:707-708, utils/augmentations.py:225-233): a second mosaic -- its own draws, its own random_perspective --
    blended into the first in uint8 inside the same launch (job.mix_job / mix_r), labels concatenated; the Beta(32, 32) ratio is drawn from numpy's
    generator right after the partner's draws.  Against the reference's own __getitem__ / collate_fn (tests/golden/augment_mixup.npz)."""
    from yolov5_amd.dataloaders import draw_sample, mosaic_batch

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "augment_mixup.npz"))
    s = int(g["s"])
    hyp = dict(HYP, mixup=0.5)
    ims_t, labs, _ = _dataset()
    draws = []
    for index in (seed % 6, (seed + 2) % 6, (seed + 4) % 6):
        random.seed(seed * 10 + index)
        np.random.seed(seed * 10 + index)
        draws.append(draw_sample(index, 6, s, hyp))
    assert any(d.get("partner") is not None for d in draws)
    imgs, targets = mosaic_batch(ims_t, labs, draws, s, hyp, dtype=torch.uint8)
    assert np.array_equal(imgs.numpy(), g[f"img{seed}"])
    assert targets.shape == g[f"lab{seed}"].shape
    np.testing.assert_array_equal(targets.numpy(), g[f"lab{seed}"])


def test_draws_follow_the_reference_order_and_loader_shapes():
    from yolov5_amd.dataloaders import MosaicLoader, draw_sample

    s = 64
    random.seed(7); np.random.seed(7)
    d = draw_sample(2, 6, s, HYP)
    ref = ao.reference_draws(7, 2, 6, s, HYP)
    for k in ("yc", "xc", "indices", "angle", "scale", "shear", "translate", "flipud", "fliplr"):
        assert d[k] == ref[k], k
    assert np.array_equal(d["hsv"], ref["hsv"])
    ims_t, labs, _ = _dataset()
    random.seed(0); np.random.seed(0)
    loader = MosaicLoader(ims_t, labs, img_size=s, batch_size=4, hyp=HYP, dtype=torch.float16)
    assert len(loader) == 2
    batches = list(loader)
    assert [tuple(b[0].shape) for b in batches] == [(4, 3, s, s), (2, 3, s, s)]
    for imgs, targets, paths, _ in batches:
        assert imgs.dtype == torch.float16 and float(imgs.max()) <= 1.0 and float(imgs.min()) >= 0.0
        assert targets.shape[1] == 6 and set(targets[:, 0].tolist()) <= set(range(imgs.shape[0]))
        assert ((targets[:, 2:] >= 0) & (targets[:, 2:] <= 1)).all() and len(paths) == imgs.shape[0]
    # two ranks of a DDP run see disjoint halves of the epoch's permutation
    r0 = MosaicLoader(ims_t, labs, img_size=s, batch_size=8, hyp=HYP, rank=0, world_size=2, seed=3)
    r1 = MosaicLoader(ims_t, labs, img_size=s, batch_size=8, hyp=HYP, rank=1, world_size=2, seed=3)
    p0, p1 = sum((b[2] for b in r0), []), sum((b[2] for b in r1), [])
    assert len(p0) == len(p1) == 3 and not set(p0) & set(p1)

"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (yolov5_amd/).

Deterministic, platform-independent data generator (splitmix64 on numpy uint64) used to create
the synthetic weights / images / predictions / targets that the golden fixtures were made from.
Because the stream depends only on (seed, element index), the GPU box can regenerate the exact
inputs without shipping them; only the (small) expected outputs are committed under tests/golden/.
"""
from __future__ import annotations

import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def key(name: str, seed: int = 0) -> int:
    """Stable 64-bit stream key from a tensor name and a seed."""
    return (zlib.crc32(name.encode()) | (seed << 32)) & 0xFFFFFFFFFFFFFFFF


def uniform(shape, lo=0.0, hi=1.0, *, name="x", seed=0) -> np.ndarray:
    """float32 array ~ U[lo, hi) with 24 random mantissa bits per element (exact in fp32)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([key(name, seed)], dtype=np.uint64))[0]
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + base
    bits = _splitmix64(idx) >> np.uint64(40)  # 24 bits
    u = bits.astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def integers(shape, lo, hi, *, name="i", seed=0) -> np.ndarray:
    """int64 array ~ U{lo..hi-1}."""
    u = uniform(shape, 0.0, 1.0, name=name, seed=seed).astype(np.float64)
    return np.minimum((lo + np.floor(u * (hi - lo))).astype(np.int64), hi - 1)


def fill_state_dict(sd: dict, seed: int = 0) -> dict:
    """Deterministic replacement values for every tensor of a YOLOv5 state_dict (name -> shape kept).

    conv weights ~ U(-b, b), b = sqrt(6 / fan_in)   (keeps SiLU activations O(1) through 60 layers)
    bn.weight ~ U(0.8, 1.2); bn.bias, running_mean ~ U(-0.1, 0.1); running_var ~ U(0.8, 1.2)
    conv bias (fused convs, Detect.m) ~ U(-0.1, 0.1) except Detect objectness/class logits which get a
    wider U(-3, 1) so that a realistic fraction of rows passes the NMS confidence threshold.
    Returns {name: np.ndarray}; integer buffers (num_batches_tracked) -> 0; `anchors` are left untouched.
    """
    out = {}
    for name, t in sd.items():
        shape = tuple(t.shape)
        if name.endswith("num_batches_tracked"):
            out[name] = np.zeros(shape, dtype=np.int64)
        elif name.endswith("anchors") or name.endswith("anchor_grid"):
            out[name] = None  # keep
        elif name.endswith("running_var"):
            out[name] = uniform(shape, 0.8, 1.2, name=name, seed=seed)
        elif name.endswith("running_mean"):
            out[name] = uniform(shape, -0.1, 0.1, name=name, seed=seed)
        elif ".bn." in name and name.endswith("weight"):
            out[name] = uniform(shape, 0.8, 1.2, name=name, seed=seed)
        elif name.endswith("bias"):
            if ".m." in name and ".bn." not in name and len(shape) == 1 and ".cv" not in name:
                out[name] = uniform(shape, -3.0, 1.0, name=name, seed=seed)  # Detect head bias
            else:
                out[name] = uniform(shape, -0.1, 0.1, name=name, seed=seed)
        elif name.endswith("weight") and len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            b = float(np.sqrt(6.0 / fan_in))
            out[name] = uniform(shape, -b, b, name=name, seed=seed)
        else:
            out[name] = uniform(shape, -0.1, 0.1, name=name, seed=seed)
    return out


def condition_state_dict(vals: dict, bn_stats=None, head_affine=None, bn_gamma_scale=None) -> dict:
    """Make a fill_state_dict network well conditioned at FULL depth / resolution (the detset_* fixtures):
    * `bn_stats` = (means, vars): float32 vectors holding, concatenated in state_dict order, the running_mean / running_var of every
      BatchNorm.  oracle/make_golden.py:gen_detset sets them to the batch statistics of the fixture's own input (one train-mode
      pass of the reference with momentum 1), i.e. every layer's activations are normalised the way a trained network's are --
      with the raw U(-0.1, 0.1) / U(0.8, 1.2) statistics the deep models saturate (yolov5x: 12 stacked shortcuts per C3) and the
      head sees almost no position-dependent signal;
    * `head_affine` = [(scale_l, bias_l)] per pyramid level, each (na*no,) float32: row r of the Detect / Segment 1x1 filter
      `<last>.m.<l>.weight` is multiplied by scale_l[r] and the bias is REPLACED by bias_l (measured on the reference so that box /
      objectness / class logits are spread like a trained head's: a ranking among near-tied scores is not a parity test).
    * `bn_gamma_scale` (round 3): every BatchNorm weight is multiplied by it BEFORE the statistics are calibrated.  A random-init SiLU
      stack with unit-variance pre-activations sits at the edge of chaos: fp16 storage noise grows with depth until, at the head, the
      class argmax and the confidence of a third to all of the detections differ between the reference's OWN fp16 and fp32 forwards
      (round-2 fixtures: 48 % / 97 % unpaired).  Trained networks are not like that.  Smaller pre-activations (std 0.07-0.2) keep the
      stack in the ordered regime -- the reference agrees with itself (<= 3 % unpaired) and a detection-level comparison has power.
    Works in place on {name: np.ndarray | None} and returns it."""
    import re

    if bn_gamma_scale is not None:
        for k, v in list(vals.items()):
            if v is not None and ".bn." in k and k.endswith("weight"):
                vals[k] = (v * np.float32(bn_gamma_scale)).astype(np.float32)

    if bn_stats is not None:
        off = {"running_mean": 0, "running_var": 0}
        for k, v in list(vals.items()):
            for which, src in (("running_mean", bn_stats[0]), ("running_var", bn_stats[1])):
                if k.endswith(which):
                    n = int(v.size)
                    vals[k] = np.asarray(src[off[which]:off[which] + n], dtype=np.float32).reshape(v.shape).copy()
                    off[which] += n
        assert off["running_mean"] == len(bn_stats[0]) and off["running_var"] == len(bn_stats[1]), "bn_stats length mismatch"
    if head_affine is not None:
        for k, v in list(vals.items()):
            if v is None:
                continue
            mw = re.search(r"^model\.\d+\.m\.(\d+)\.weight$", k)
            mb = re.search(r"^model\.\d+\.m\.(\d+)\.bias$", k)
            if mw and v.ndim == 4:
                sc = np.asarray(head_affine[int(mw.group(1))][0], dtype=np.float32)
                vals[k] = (v * sc[:, None, None, None]).astype(np.float32)
            elif mb and v.ndim == 1:
                vals[k] = np.asarray(head_affine[int(mb.group(1))][1], dtype=np.float32).copy()
    return vals


def scene(shape, *, seed=0, nrect=24) -> np.ndarray:
    """Synthetic (bs, 3, H, W) float32 images in [0, 1] with STRUCTURE: a per-image colour gradient, `nrect` random rectangles of
    random colour and a little noise.  (Uniform noise looks the same at every position to a convolutional net: its head outputs
    barely vary over the grid.)"""
    bs, c, H, W = shape
    yy = np.linspace(0.0, 1.0, H, dtype=np.float32)[None, :, None]
    xx = np.linspace(0.0, 1.0, W, dtype=np.float32)[None, None, :]
    out = np.empty(shape, dtype=np.float32)
    for b in range(bs):
        g = uniform((c, 3), 0.0, 1.0, name="scene_grad", seed=seed * 1000 + b)
        img = g[:, 0, None, None] * 0.4 + g[:, 1, None, None] * 0.3 * yy + g[:, 2, None, None] * 0.3 * xx
        r = uniform((nrect, 4), 0.0, 1.0, name="scene_rect", seed=seed * 1000 + b)
        col = uniform((nrect, c), 0.0, 1.0, name="scene_col", seed=seed * 1000 + b)
        for k in range(nrect):
            cx, cy, w, h = r[k]
            w, h = 0.04 + 0.3 * w * w, 0.04 + 0.3 * h * h
            x0, x1 = int(max(0.0, cx - w / 2) * W), int(min(1.0, cx + w / 2) * W)
            y0, y1 = int(max(0.0, cy - h / 2) * H), int(min(1.0, cy + h / 2) * H)
            img[:, y0:y1, x0:x1] = col[k][:, None, None]
        out[b] = img
    out += (uniform(shape, -0.03, 0.03, name="scene_noise", seed=seed)).astype(np.float32)
    return np.clip(out, 0.0, 1.0)


def synth_predictions(bs, n, no, *, obj_pow=8, seed=0, img=640.0) -> np.ndarray:
    """Synthetic Detect output (bs, n, no) float32 as SURVEY 8d defines for the NMS benchmark.

    xy ~ U(0,img); wh ~ U(4,104); obj = u**obj_pow; cls ~ U(0,1); extra (mask) columns ~ U(-1,1).
    """
    p = uniform((bs, n, no), 0.0, 1.0, name=f"pred{obj_pow}", seed=seed)
    p[..., 0:2] *= img
    p[..., 2:4] = 4.0 + 100.0 * p[..., 2:4]
    p[..., 4] = p[..., 4] ** obj_pow
    return p


def synth_targets(bs, per_img, nc=80, *, seed=0) -> np.ndarray:
    """Synthetic training targets (nt,6) float32 [img, cls, x, y, w, h]."""
    nt = bs * per_img
    t = np.zeros((nt, 6), dtype=np.float32)
    t[:, 0] = np.repeat(np.arange(bs), per_img)
    t[:, 1] = integers((nt,), 0, nc, name="tcls", seed=seed)
    t[:, 2:4] = uniform((nt, 2), 0.1, 0.9, name="txy", seed=seed)
    t[:, 4:6] = uniform((nt, 2), 0.02, 0.32, name="twh", seed=seed)
    return t

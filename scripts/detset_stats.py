"""Prints the error statistics of the HIP forward (+ NMS) against the reference fixtures tests/golden/detset_*.npz (fp32 and fp16):
the numbers the tolerances of tests/test_gpu_configs.py were set from.  GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tests import detset
from yolov5_amd.general import non_max_suppression
from yolov5_amd.yolo import DetectionModel, SegmentationModel

dev = torch.device("cuda:0")
for name in detset.CASES:
    g, cfg, x, seed, seg = detset.load(name)
    model = detset.CASES[name][0]
    for half in (False, True):
        m = (SegmentationModel if seg else DetectionModel)(model + ".yaml")
        m.load_state_dict(detset.state_dict(name, g, fused=False))
        m = m.eval().fuse()
        m = (m.half() if half else m.float()).to(dev)
        z = m(x.half().to(dev) if half else x.to(dev))[0]
        rs = int(g["row_stride"])
        rows = z.float().cpu().numpy().reshape(-1, z.shape[-1])[::rs]
        ref = g["z_rows"]
        d = np.abs(rows - ref)
        size = np.maximum(ref[:, 2], ref[:, 3])[:, None]
        rel_box = d[:, :4] / (size + 8.0)
        k = np.unravel_index(d.argmax(), d.shape)
        print(f"{name} {'fp16' if half else 'fp32'}: max|d| {d.max():.4g} at col {k[1]} ref {ref[k]:.5g}; box abs mean {d[:, :4].mean():.4g} q999 {np.quantile(d[:, :4], .999):.4g} max {d[:, :4].max():.4g}; "
              f"box rel mean {rel_box.mean():.3g} q999 {np.quantile(rel_box, .999):.3g} max {rel_box.max():.3g}; conf mean {d[:, 4:85].mean():.3g} q999 {np.quantile(d[:, 4:85], .999):.3g} max {d[:, 4:85].max():.3g}"
              + (f"; mask max {d[:, 85:].max():.3g} (|ref| max {np.abs(ref[:, 85:]).max():.3g})" if seg else ""))
        relerr = d / (1e-4 * np.maximum(np.abs(ref), 1.0) + 2e-4)
        print(f"   worst fp32-tolerance ratio {relerr.max():.3g}")
        if half:
            conf, iou, max_det = float(g["nms"][0]), float(g["nms"][1]), int(g["nms"][2])
            dets = non_max_suppression(z, conf, iou, max_det=max_det, nm=32 if seg else 0)
            for i, dd in enumerate(dets):
                r = g[f"det{i}"]
                a = detset.agreement(r, dd.cpu().numpy(), conf)
                a2 = detset.agreement(r, dd.cpu().numpy(), conf, box_atol=1e9, conf_atol=1.0)
                print(f"   img {i}: ref {len(r)} got {len(dd)} agreement {a}; class-only unmatched {a2['unmatched_ref']}/{a2['unmatched_got']}")

#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): images/sec at 640 px, yolov5s, bs=64 per GPU, fp16, synthetic data.

One "step" = the whole inference hot path over one batch that is already resident in HBM:
    NCHW fp16 batch -> HIP forward (backbone + PANet neck + Detect decode) -> HIP non_max_suppression.
N GPUs: one process per GPU (torch.distributed, backend nccl = RCCL), every rank runs the same per-GPU batch
(images are independent: weak scaling, no data-path collective -- DESIGN.md section "multi-GPU").

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (conv implicit-GEMM kernel,
MFMA bound) and `cpu_baseline` (the CPU oracle = port of the reference path, timed on this box's host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBS = 8000.0      # HBM3E, same table (about 6.3 TB/s is achievable in practice)


def build_model(name, dev, half=True):
    from yolov5_amd.yolo import DetectionModel, SegmentationModel

    torch.manual_seed(0)
    m = (SegmentationModel if name.endswith("-seg") else DetectionModel)(name + ".yaml").eval().fuse()
    m = m.half() if half else m.float()
    return m.to(dev)


def calibrate_head(model, x, obj_frac=0.04, conf=0.25):
    """Random-init weights give objectness ~0.007 and class scores ~0.01 everywhere (models/yolo.py:323-326 bias
    init), so NMS would see no candidates.  Shift the Detect biases so that ~`obj_frac` of the rows have
    obj > 0.5 and the typical best-class score is ~0.7: a few hundred candidates per image pass
    obj*cls > 0.25 (SURVEY 8d 'realistic' NMS load) and overlapping boxes get suppressed."""
    import math

    det = model.model[-1]
    z = model(x)[0].float()
    obj = z[..., 4].flatten()
    q = obj.kthvalue(max(int(obj.numel() * (1 - obj_frac)), 1)).values.clamp(1e-6, 1 - 1e-6)
    nc = det.nc
    cls = z[..., 5:5 + nc].max(-1).values.flatten()  # (a Segment head carries raw mask coefficients after the classes)
    qc = cls.median().clamp(1e-6, 1 - 1e-6)
    logit = lambda v: math.log(float(v) / (1 - float(v)))  # noqa: E731
    with torch.no_grad():
        for mi in det.m:
            b = mi.bias.view(det.na, -1)
            b[:, 4] += logit(0.5) - logit(q)
            b[:, 5:5 + nc] += logit(0.7) - logit(qc)
    model.invalidate_engine()


def conv_flops(engine):
    """Algorithmic FLOPs per conv launch: 2 * B*OH*OW * C2 * C1*kh*kw with the TRUE (unpadded) channel counts."""
    out = []
    B = engine.spec.B
    for i, op in enumerate(engine.spec.ops):
        if op["op"] != "conv":
            continue
        fl = 0
        for m in op["mods"]:
            cv = m.conv if hasattr(m, "conv") else m
            fl += 2 * B * op["y"].H * op["y"].W * cv.out_channels * cv.in_channels * cv.kernel_size[0] * cv.kernel_size[1]
        out.append((i, fl))
    return out


def conv_bytes(engine, esize=2):
    """Algorithmic HBM bytes per conv launch (SURVEY.md 8(d) "per-layer unfused conv traffic"): input read once + output
    written once (+ the residual read of a Bottleneck add) + filter read once, true channel counts, `esize` bytes/element.
    The input of the NCHW stem is its 3-channel image.  yolov5s bs=64 640^2: 7.8 GB per forward."""
    out = []
    B = engine.spec.B
    for i, op in enumerate(engine.spec.ops):
        if op["op"] != "conv":
            continue
        x, y = op["x"], op["y"]
        cin = sum((m.conv if hasattr(m, "conv") else m).in_channels for m in op["mods"][:1])
        by = B * x.H * x.W * cin + B * y.H * y.W * sum((m.conv if hasattr(m, "conv") else m).out_channels for m in op["mods"])
        if op.get("res") is not None:
            by += B * y.H * y.W * y.C
        for m in op["mods"]:
            cv = m.conv if hasattr(m, "conv") else m
            by += cv.out_channels * cv.in_channels * cv.kernel_size[0] * cv.kernel_size[1]
        out.append((i, by * esize))
    return out


def pmc_traffic(a, conv_by):
    """HBM bytes of the conv launches of one forward from the memory-side L2 counters.  They cannot be collected from inside this
    process: scripts/pmc_forward.sh runs the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the same forward and the
    result is committed under profiles/pmc/ (FETCH_SIZE doubled: gfx950 correction of MI355X_MICROARCH.md "HBM").  Only
    reported for the configuration it was measured on."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc", "r01_pmc_forward.json")
    if not (a.model == "yolov5s" and a.batch == 64 and a.imgsz == 640 and os.path.isfile(path)):
        return None
    try:
        with open(path) as f:
            d = json.load(f)
        gb = float(d["conv_traffic_gb_per_forward"])
    except (OSError, ValueError, KeyError):
        return None
    return {"gbytes_per_step": round(gb, 3), "vs_algorithmic": round(gb * 1e9 / conv_by, 3) if conv_by else None,
            "source": "profiles/pmc/r01_pmc_forward.json (scripts/pmc_forward.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, FETCH x2)"}


def usable_cores():
    """Host cores this process may really use: affinity mask and cgroup CPU quota, capped at 64 threads (oneDNN/OpenMP
    stop scaling -- and thrash -- far below the 256 logical CPUs the GPU box advertises)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(seconds_budget=20.0):
    """The oracle (CPU port of the reference forward + NMS, torch-CPU fp32, all host cores) on a bounded sample:
    yolov5s fused, 8 images of 3x640x640, forward + NMS per iteration."""
    from oracle import detgen, yolo_oracle as yo

    cores = usable_cores()
    torch.set_num_threads(cores)
    cfg = yo.model_cfg("yolov5s")
    sd = yo.det_state_dict(cfg, 0, fused=True)
    bs = 8
    x = torch.from_numpy(detgen.uniform((bs, 3, 640, 640), 0.0, 1.0, name="img", seed=0))
    with torch.no_grad():
        yo.model_forward(cfg, sd, x[:1])  # warm-up
        t0 = time.time()
        n = 0
        while True:
            z = yo.model_forward(cfg, sd, x)[0]
            yo.non_max_suppression(z.numpy(), 0.25, 0.45, max_det=1000)
            n += bs
            if time.time() - t0 > seconds_budget or n >= 64:
                break
        dt = time.time() - t0
    return {"value": round(n / dt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"oracle/yolo_oracle.py forward+NMS, yolov5s fused fp32, {n} images of 3x640x640 (batches of {bs}), {dt:.1f} s"}


def train_probe(name, batch, imgsz, dev, world, steps=6, warmup=2):
    """Secondary measurement (BASELINE config 3 per-GPU shape): training step = train-mode forward + ComputeLoss + backward
    (+ bucketed RCCL gradient all-reduce when world > 1) + SGD, fp16 compute with fp32 master weights, synthetic data."""
    from yolov5_amd.loss import ComputeLoss
    from yolov5_amd.torch_utils import ModelEMA, smart_DDP, smart_optimizer
    from yolov5_amd.yolo import DetectionModel

    torch.manual_seed(0)
    m = DetectionModel(name + ".yaml").to(dev).train()
    m.hyp = {"box": 0.05, "cls": 0.5, "cls_pw": 1.0, "obj": 1.0, "obj_pw": 1.0, "anchor_t": 4.0, "fl_gamma": 0.0, "label_smoothing": 0.0}
    compute_loss = ComputeLoss(m)
    model = smart_DDP(m) if world > 1 else m
    opt = smart_optimizer(m, "SGD", lr=0.01, momentum=0.937, decay=5e-4)  # HipSGD: 3 groups, fused multi-tensor step
    ema = ModelEMA(m)
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.rand((batch, 3, imgsz, imgsz), generator=g).half().to(dev)
    nt = batch * 8
    t = torch.cat((torch.randint(0, batch, (nt, 1), generator=g).float(), torch.randint(0, 80, (nt, 1), generator=g).float(),
                   torch.rand((nt, 2), generator=g) * 0.8 + 0.1, torch.rand((nt, 2), generator=g) * 0.3 + 0.02), 1).to(dev)
    scale = 1024.0

    def step():
        pred = model(x)
        loss, _ = compute_loss(pred, t)
        if world > 1:
            loss = loss * world  # train.py:404-405
        opt.zero_grad(set_to_none=True)
        (loss * scale).backward()
        # train.py:413-421 scaler.unscale_ + clip_grad_norm_(10.0) + optimizer step + ema.update, fused (csrc/optim.hip)
        opt.step_fused(inv_scale=1.0 / scale, max_norm=10.0, ema=ema, model=m)
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    del m, model, opt
    torch.cuda.empty_cache()
    return {"images_per_sec": round(batch * world * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 2), "steps": steps,
            "workload": f"{name} train step, {batch} img/GPU 3x{imgsz}x{imgsz}, {nt} targets: forward(train BN) + ComputeLoss + backward"
                        f"{' + RCCL all-reduce' if world > 1 else ''} + SGD; fp16 compute / fp32 masters", "loss": round(float(loss.detach()), 4)}


def pipeline_probe(model, batch, dev, nm, iters=10):
    """Secondary measurement: detect.py / val.py around the hot path, everything on the device -- `batch` uint8 1280x720 BGR frames
    resident in HBM -> letterbox + CHW + RGB + /255 (one launch) -> forward -> NMS (padded result, no host sync) -> scale_boxes of
    all images (one launch) -> validation matching against synthetic labels (one launch) -> the counts' D2H copy."""
    from yolov5_amd.augmentations import letterbox_batch
    from yolov5_amd.general import non_max_suppression, scale_boxes_batch
    from yolov5_amd.metrics import match_batch

    g = torch.Generator(device="cpu").manual_seed(7)
    frames = [torch.randint(0, 256, (720, 1280, 3), generator=g, dtype=torch.uint8).to(dev) for _ in range(4)]
    ims = [frames[i % 4] for i in range(batch)]
    nt = batch * 7
    targets = torch.cat((torch.randint(0, batch, (nt, 1), generator=g).float(), torch.randint(0, 80, (nt, 1), generator=g).float(),
                         torch.rand((nt, 2), generator=g) * 500 + 70, torch.rand((nt, 2), generator=g) * 150 + 20), 1).to(dev)
    iouv = torch.linspace(0.5, 0.95, 10, device=dev)

    def step():
        x, shapes = letterbox_batch(ims, 640, auto=False, dtype=torch.float16, swap_rb=True)
        out, cnt = non_max_suppression(model(x)[0], 0.25, 0.45, max_det=300, nm=nm, padded=True)
        correct = match_batch(out, cnt, targets, shapes, iouv)
        scale_boxes_batch((640, 640), out, cnt, [s[0] for s in shapes], [s[1] for s in shapes])
        return cnt.tolist(), correct

    for _ in range(3):
        counts, _ = step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / iters * 1e3
    return {"images_per_sec": round(batch / ms * 1e3, 1), "ms_per_batch": round(ms, 4), "detections_per_img": round(sum(counts) / batch, 1),
            "workload": f"{batch} uint8 1280x720 frames in HBM -> letterbox/CHW/RGB//255 -> forward -> NMS (max_det 300) -> val matching "
                        "(10 IoU thresholds) -> scale_boxes; one host sync per batch"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--model", default="yolov5s")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary training-step measurement")
    ap.add_argument("--train", action="store_true", help="run the training-step probe also under torchrun (N > 1: adds the RCCL gradient "
                    "all-reduce; by default the multi-GPU run measures the headline inference metric only, scripts/train_bench.py is the DDP entry)")
    ap.add_argument("--op-table", default="", help="write the per-op timing table (JSON) to this path")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from yolov5_amd.general import non_max_suppression

    model = build_model(a.model, dev)
    model.model[-1].export = True  # AutoShape mode: return (z,) only (models/common.py:866)
    nm = getattr(model.model[-1], "nm", 0)  # Segment head (C5: yolov5s-seg): 32 mask coefficients ride through NMS
    g = torch.Generator(device="cpu").manual_seed(rank)
    x = torch.rand((a.batch, 3, a.imgsz, a.imgsz), generator=g).half().to(dev)
    calibrate_head(model, x)

    def step():
        z = model(x)[0]
        return non_max_suppression(z, 0.25, 0.45, max_det=1000, nm=nm)

    for _ in range(max(a.warmup, 1)):
        det = step()
    ncand = sum(int(d.shape[0]) for d in det) / len(det)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- per-kernel timing on the launch stream (HIP events inside y5_plan_time_range) -----------------
    torch.cuda.synchronize(dev)
    eng_top = next(iter(model._engines.values()))
    parts = getattr(eng_top, "parts", 1)  # SplitEngine: `parts` sub-batch plans on separate streams; per-kernel figures come from one of them
    eng = eng_top.engines[0] if parts > 1 else eng_top
    t_f0 = time.perf_counter()
    for _ in range(10):
        model(x)
    torch.cuda.synchronize(dev)
    fwd_ms = (time.perf_counter() - t_f0) / 10 * 1e3
    z = model(x)[0]
    torch.cuda.synchronize(dev)
    t_n0 = time.perf_counter()
    for _ in range(10):
        non_max_suppression(z, 0.25, 0.45, max_det=1000, nm=nm)
    torch.cuda.synchronize(dev)
    nms_ms = (time.perf_counter() - t_n0) / 10 * 1e3
    ops = eng.time_ops(iters=10)
    fl = dict(conv_flops(eng))
    if eng._stem is not None:
        fl[eng._stem] = fl[1]  # the fused NCHW stem op computes spec op 1 (0.Conv)
    timed = list(zip(eng.timed_order, ops))  # (plan index, (name, ms)) in execution order
    conv_ms = sum(ms for i, (name, ms) in timed if i in fl)
    conv_fl = sum(fl[i] for i, _ in timed if i in fl)
    other_ms = sum(ms for i, (name, ms) in timed if i not in fl)
    achieved = conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    by = dict(conv_bytes(eng))
    if eng._stem is not None:
        by[eng._stem] = by[1]
    conv_by = sum(by[i] for i, _ in timed if i in by)
    achieved_bw = conv_by / (conv_ms * 1e-3) / 1e9 if conv_ms > 0 else 0.0  # GB/s
    nconv = sum(1 for i, _ in timed if i in fl)
    images_per_plan = eng.spec.B
    if a.op_table and rank == 0:
        cfg_of = {}
        ci = iter(eng.conv_cfgs)
        for i, op in enumerate(eng.spec.ops):
            if op["op"] == "conv":
                cfg_of[i] = next(ci)
        table = [{"op": name, "cfg": cfg_of.get(i), "ms": round(ms, 5), "gflop": round(fl.get(i, 0) / 1e9, 3),
                  "tflops": round(fl.get(i, 0) / (ms * 1e-3) / 1e12, 1) if ms > 0 and i in fl else None}
                 for i, (name, ms) in timed]
        with open(a.op_table, "w") as f:
            json.dump(table, f, indent=1)

    # ---- secondary: the detect.py pipeline around the hot path (SURVEY 8(f) rank 1 + 3 rows) ---------------------------------
    pipeline = None
    if world == 1 and a.imgsz == 640 and not a.no_train:
        try:
            pipeline = pipeline_probe(model, a.batch, dev, nm)
        except Exception as e:  # the headline metric must not depend on the secondary probe
            pipeline = {"error": f"{type(e).__name__}: {e}"}

    train = None
    if not a.no_train and (world == 1 or a.train):
        try:
            del model, eng, eng_top
            torch.cuda.empty_cache()
            train = train_probe(a.model, a.batch, a.imgsz, dev, world)
        except Exception as e:  # the headline metric must not depend on the secondary probe
            train = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        imgs = a.batch * world * a.steps
        res = {
            "metric": "images/sec at 640px (yolov5s bs=64), forward+NMS", "value": round(imgs / dt, 1), "unit": "images/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{a.model} inference bs={a.batch}/GPU 3x{a.imgsz}x{a.imgsz} fp16: HIP forward (backbone+neck+Detect) + HIP NMS "
                                   "(conf 0.25, iou 0.45, max_det 1000); random-init weights, Detect biases calibrated to a realistic NMS load",
                       "global_batch": a.batch * world, "parallelism": f"replicas x{world} (images independent, no collective)"},
            "forward_ms": round(fwd_ms, 4), "nms_ms": round(nms_ms, 4), "nms_us_per_img": round(nms_ms * 1e3 / a.batch, 2),
            "forward_images_per_sec": round(a.batch / (fwd_ms * 1e-3), 1), "detections_per_img": round(ncand, 1),
            # arithmetic intensity of the conv stack at this config = algorithmic flops / algorithmic bytes (135 flop/B for
            # yolov5s bs=64 640^2) is below the ridge (2500 TF / 8 TB/s = 312 flop/B): the stack as a whole is HBM-bound;
            # the MFMA view of the same launches is kept beside it
            "roofline": {"bound": "hbm", "kernel": "y5_conv_{igemm,pw,k3,stem}_kernel (all conv launches of one forward)",
                         "achieved": round(achieved_bw, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved_bw / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(a, conv_by * parts),
                         "plans_per_step": parts, "images_per_plan": images_per_plan,
                         "algorithmic_gbytes_per_step": round(conv_by * parts / 1e9, 3), "algorithmic_gflop_per_step": round(conv_fl * parts / 1e9, 1),
                         "arithmetic_intensity_flop_per_byte": round(conv_fl / conv_by, 1) if conv_by else None,
                         "mfma_achieved_tflops": round(achieved, 2), "mfma_peak_tflops": MFMA_PEAK_TFLOPS,
                         "mfma_frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                         # durations are per launch in isolation (HIP events); with plans_per_step > 1 the plans overlap in time
                         "conv_ms_per_step": round(conv_ms * parts, 4), "launches_per_step": nconv * parts,
                         "other_kernels_ms_per_step": round(other_ms * parts, 4)},
        }
        if pipeline is not None:
            res["pipeline"] = pipeline
        if train is not None:
            res["train"] = train
        if not a.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): images/sec at 640 px, yolov5s, bs=64 per GPU, fp16, synthetic data.

One "step" = the whole inference hot path over one batch that is already resident in HBM:
    NCHW fp16 batch -> HIP forward (backbone + PANet neck + Detect decode) -> HIP non_max_suppression.
N GPUs: one process per GPU (torch.distributed, backend nccl = RCCL), every rank runs the same per-GPU batch
(images are independent: weak scaling, no data-path collective -- DESIGN.md section "multi-GPU").

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes itself under torch.distributed.run
(one rank per GPU, 127.0.0.1 rendezvous); under a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE as usual.  At N > 1 the line
also carries the DDP TRAINING step (`train`: the only path with a collective -- bucketed RCCL all-reduce of the gradient arena
overlapped with backward, train.py:404-405 semantics) with `rccl_ranks` and the all-reduce bytes per step.

Protocol: >= 10 warm-up + >= 50 timed steps; `value` = images of the K timed steps / wall time between two
barrier + synchronize fences (max over ranks); per-step device-event times give median / p10 / p90 beside it.  The K steps run through
`DetectPipeline` (same kernels; the host's wait for batch i's per-image counts is taken after batch i+1 is queued, the NMS chain of batch i
runs on a high-priority side stream beside forward i+1, the last batch is collected inside the timed region); `--sequential` times the plain loop, and whichever is not the headline is reported as
`alt_step_mode` from a second timed run of the same K steps.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (all conv launches of one forward, durations
measured IN SITU: one HIP event between consecutive plan ops of a real forward) and `cpu_baseline` (the CPU oracle = port of
the reference path, timed on this box's host cores BEFORE the GPU work starts).
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
        subs = [op] if op["op"] == "conv" else [op["cv1"], op["cv2"]] if op["op"] == "bneck" else []
        if not subs:
            continue
        fl = 0
        for sub in subs:
            for m in sub["mods"]:
                cv = m.conv if hasattr(m, "conv") else m
                fl += 2 * B * sub["y"].H * sub["y"].W * cv.out_channels * cv.in_channels * cv.kernel_size[0] * cv.kernel_size[1]
        out.append((i, fl))
    return out


def conv_bytes(engine, esize=2):
    """Algorithmic HBM bytes per conv launch (SURVEY.md 8(d) "per-layer unfused conv traffic"): input read once + output
    written once (+ the residual read of a Bottleneck add) + filter read once, true channel counts, `esize` bytes/element.
    The input of the NCHW stem is its 3-channel image.  yolov5s bs=64 640^2: 7.3 GB per forward.  A fused launch (Bottleneck: cv1 +
    cv2 + residual in one pass) is charged the sum of the layers it replaces -- the figure stays the per-layer one whatever the plan
    fuses, so that `achieved` measures time, not accounting."""
    out = []
    B = engine.spec.B

    def one(op, residual):
        x, y = op["x"], op["y"]
        cin = sum((m.conv if hasattr(m, "conv") else m).in_channels for m in op["mods"][:1])
        cout = sum((m.conv if hasattr(m, "conv") else m).out_channels for m in op["mods"])
        by = B * x.H * x.W * cin + B * y.H * y.W * cout
        if residual:
            by += B * y.H * y.W * cout
        for m in op["mods"]:
            cv = m.conv if hasattr(m, "conv") else m
            by += cv.out_channels * cv.in_channels * cv.kernel_size[0] * cv.kernel_size[1]
        return by

    for i, op in enumerate(engine.spec.ops):
        if op["op"] == "conv":
            out.append((i, one(op, op.get("res") is not None) * esize))
        elif op["op"] == "bneck":
            out.append((i, (one(op["cv1"], False) + one(op["cv2"], op["add"])) * esize))
    return out


def pmc_traffic(a, conv_by):
    """HBM bytes of the conv launches of one forward from the memory-side L2 counters.  They cannot be collected from inside this
    process: scripts/pmc_forward.sh runs the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the same forward and the
    result is committed under profiles/pmc/ (FETCH_SIZE doubled: gfx950 correction of MI355X_MICROARCH.md "HBM").  Only
    reported for the configuration it was measured on."""
    import glob

    cands = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc", "r*_pmc_forward.json")))
    if not (a.model == "yolov5s" and a.batch == 64 and a.imgsz == 640 and cands):
        return None
    path = cands[-1]  # newest round
    try:
        with open(path) as f:
            d = json.load(f)
        gb = float(d["conv_traffic_gb_per_forward"])
    except (OSError, ValueError, KeyError):
        return None
    return {"gbytes_per_step": round(gb, 3), "vs_algorithmic": round(gb * 1e9 / conv_by, 3) if conv_by else None,
            "source": f"profiles/pmc/{os.path.basename(path)} (scripts/pmc_forward.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, FETCH x2)"}


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


def cpu_baseline(seconds_budget=10.0):
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


def measured_ceilings(dev):
    """What THIS box sustains, measured now: MFMA throughput and shader clock under back-to-back 32x32x16 fp16 MFMAs (y5_probe_mfma, no memory
    traffic) and device-to-device copy bandwidth (read + write counted once each) -- the ceilings the practical per-layer floor is computed from."""
    import ctypes as C

    from yolov5_amd import _lib

    lib = _lib.lib()
    scratch = torch.empty(4 << 20, dtype=torch.uint8, device=dev)
    tf, ghz = C.c_float(0), C.c_float(0)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(lib.y5_probe_mfma(C.c_void_p(scratch.data_ptr()), scratch.numel(), 20000, C.byref(tf), C.byref(ghz), st), lib)
    a = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize(dev)
    copy_gbs = 10 * 2 * a.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b
    torch.cuda.empty_cache()
    return {"mfma_sustained_tflops": round(tf.value, 1), "shader_clock_ghz_under_mfma": round(ghz.value, 3), "copy_gbytes_per_s": round(copy_gbs, 1)}


def _pct(v, q):
    v = sorted(v)
    return v[min(len(v) - 1, max(0, int(round(q * (len(v) - 1)))))]


def event_times(fn, iters, dev):
    """Per-iteration device time (ms) of `fn` from HIP events on torch's current stream (the stream every yolov5_amd launch uses)."""
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    evs[0].record()
    for i in range(iters):
        fn()
        evs[i + 1].record()
    torch.cuda.synchronize(dev)
    return [evs[i].elapsed_time(evs[i + 1]) for i in range(iters)]


def stats(ms):
    return {"median": round(_pct(ms, 0.5), 4), "p10": round(_pct(ms, 0.1), 4), "p90": round(_pct(ms, 0.9), 4), "n": len(ms)}


def gpu_state_probe(fn, dev, seconds=1.6):
    """Clocks and socket power WHILE `fn` (one forward) runs in a loop: two `rocm-smi --json` samples taken by a side thread.  The same
    binary runs 15 % apart on different MI355X boxes of the pool; this records what the box was doing (box-to-box variance)."""
    import re
    import subprocess
    import threading

    samples = []

    def sample():
        for _ in range(2):
            time.sleep(0.45)
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
                card = next(iter(json.loads(out).values()))
                num = lambda k: float(re.sub(r"[^0-9.]", "", str(card.get(k, "")) or "0") or 0)  # noqa: E731
                samples.append({"sclk_mhz": num("sclk clock speed:"), "mclk_mhz": num("mclk clock speed:"), "fclk_mhz": num("fclk clock speed:"),
                                "power_w": num("Current Socket Graphics Package Power (W)")})
            except Exception as e:  # the probe must never break the benchmark
                samples.append({"error": f"{type(e).__name__}: {e}"})

    th = threading.Thread(target=sample)
    th.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds or th.is_alive():
        fn()
        n += 1
        if n % 16 == 0:
            torch.cuda.synchronize(dev)
        if time.perf_counter() - t0 > 20:
            break
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    th.join()
    props = torch.cuda.get_device_properties(dev)
    return {"forward_ms_sustained": round(dt / max(n, 1) * 1e3, 4), "samples": samples, "device": props.name, "compute_units": props.multi_processor_count,
            "hbm_gb": round(props.total_memory / 2 ** 30, 1)}


TRAIN_GFLOP_PER_IMG = {"yolov5s": 49.3}  # forward + data gradient + weight gradient = 3 x 16.43 GFLOP at 640^2


def train_probe(name, batch, imgsz, dev, world, steps=20, warmup=5):
    """BASELINE config 3 per-GPU shape: one training step = train-mode forward (batch-statistics BN) + ComputeLoss + backward
    (+ bucketed RCCL gradient all-reduce overlapped with backward when world > 1, loss * WORLD_SIZE as train.py:404-405) +
    unscale / clip / SGD-Nesterov / EMA (fused), fp16 compute with fp32 master weights, synthetic data."""
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
    g = torch.Generator(device="cpu").manual_seed(1 + int(os.environ.get("RANK", 0)))
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
        torch.cuda.synchronize(dev)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(steps):
        loss = step()
        evs[i + 1].record()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    per = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
    if world > 1:
        tt = torch.tensor([dt], device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ar_bytes = nbuckets = None
    if world > 1:
        ar_bytes = sum((b.hi - b.lo) * 4 for b in model.buckets)
        nbuckets = len(model.buckets)
    del m, model, opt
    torch.cuda.empty_cache()
    ips = batch * world * steps / dt
    out = {"images_per_sec": round(ips, 1), "ms_per_step": round(dt / steps * 1e3, 2), "steps": steps, "warmup": warmup,
           "step_ms": stats(per), "rccl_ranks": dist.get_world_size() if world > 1 else 1,
           "allreduce_bytes_per_step": ar_bytes, "allreduce_buckets": nbuckets,
           "workload": f"{name} train step, {batch} img/GPU 3x{imgsz}x{imgsz}, {nt} targets: forward(train BN) + ComputeLoss + backward"
                       f"{' + RCCL all-reduce (overlapped)' if world > 1 else ''} + clip + SGD + EMA; fp16 compute / fp32 masters",
           "loss": round(float(loss.detach()), 4)}
    gf = TRAIN_GFLOP_PER_IMG.get(name)
    if gf and imgsz == 640:
        tf = ips / world * gf / 1e3  # per GPU
        out["roofline"] = {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4),
                           "traffic": None, "algorithmic_gflop_per_img": gf,
                           "note": "whole training step (all kernels + host), per GPU: 3 x forward conv FLOPs / step time"}
    return out


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


def respawn_under_torchrun(n):
    """`bench.py --gpus N` outside a launcher: become `python -m torch.distributed.run --nproc-per-node N bench.py ...`."""
    import socket

    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--model", default="yolov5s")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sequential", action="store_true", help="time the plain per-step loop (host sync inside every step) instead of DetectPipeline")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step measurement")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the detect/val pipeline measurement")
    ap.add_argument("--train", action="store_true", help="(kept for compatibility: the training step is measured at every N unless --no-train)")
    ap.add_argument("--op-table", default="", help="write the per-op timing table (JSON: in-situ and isolated ms, %% of bound) to this path")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        respawn_under_torchrun(a.gpus)
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus and rank == 0:
        print(f"[bench] --gpus {a.gpus} but the launcher started {world} rank(s): reporting n_gpus={world}", file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs a GPU"

    # ---- CPU baseline first: the GPU is idle while the host cores run the oracle, the rest of the run is GPU work ------------
    cpu = None
    if not a.no_cpu_baseline and world == 1:
        cpu = cpu_baseline()
        torch.set_num_threads(min(8, usable_cores()))

    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

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

    # The timed region: EXACTLY a.steps steps, every one a full forward + NMS + per-image result lists on the host side.  Default: the steps run
    # through yolov5_amd.detect_loop.DetectPipeline -- same kernels, same order, one stream; the host's only wait (the per-image counts of batch i)
    # is taken AFTER batch i+1 has been queued, so the GPU does not idle while the host wakes up, builds the lists and launches the next forward
    # (~140 us of a 2.7 ms step in the rocprofv3 trace).  The last batch is collected (flush) before the closing fence.  --sequential times the
    # plain loop `non_max_suppression(model(x)[0])` instead; the other of the two is measured right after and reported as `alt_step_mode`.
    # (DetectPipeline also moves the NMS chain of batch i to a high-priority side stream, where it overlaps forward i+1: detect_loop.py.)
    from yolov5_amd.detect_loop import DetectPipeline

    pipe = DetectPipeline(model, 0.25, 0.45, max_det=1000, nm=nm)

    def timed_run(pipelined):
        if pipelined:  # untimed: the caching allocator grows to the pipeline's steady state (three generations of NMS buffers alive: queued, waited-for, held by the caller)
            for _ in range(max(6, a.warmup)):
                r = pipe.submit(x)
            r = pipe.flush()
        fence()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
        t0 = time.perf_counter()
        evs[0].record()
        ndet = 0
        host_t = [t0]
        for i in range(a.steps):
            r = pipe.submit(x) if pipelined else step()
            ndet += 0 if r is None else len(r)
            evs[i + 1].record()
            host_t.append(time.perf_counter())
        if pipelined:
            ndet += len(pipe.flush())
        fence()
        dt = time.perf_counter() - t0
        assert ndet == a.steps * a.batch, (ndet, a.steps, a.batch)  # every batch of the timed region delivered its per-image results inside it
        if os.environ.get("Y5_BENCH_DEBUG") and rank == 0:
            d = [round((host_t[i + 1] - host_t[i]) * 1e3, 2) for i in range(a.steps)]
            print(f"[bench debug] pipelined={pipelined} wall {dt * 1e3:.2f} ms, host ms per step: {d}", file=sys.stderr)
        return dt, [evs[i].elapsed_time(evs[i + 1]) for i in range(a.steps)]

    dt, step_ms = timed_run(not a.sequential)
    alt_dt, _ = timed_run(a.sequential)
    if world > 1:
        t = torch.tensor([dt, alt_dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, alt_dt = float(t[0].item()), float(t[1].item())

    # ---- forward / NMS alone: device events, >= 50 iterations ------------------------------------------------------------------
    torch.cuda.synchronize(dev)
    eng_top = next(iter(model._engines.values()))
    parts = getattr(eng_top, "parts", 1)  # SplitEngine: `parts` sub-batch plans on separate streams; per-kernel figures come from one of them
    eng = eng_top.engines[0] if parts > 1 else eng_top
    n_it = max(50, a.steps)
    fwd = event_times(lambda: model(x), n_it, dev)
    z = model(x)[0]
    nms = event_times(lambda: non_max_suppression(z, 0.25, 0.45, max_det=1000, nm=nm), n_it, dev)
    nms_dev = event_times(lambda: non_max_suppression(z, 0.25, 0.45, max_det=1000, nm=nm, padded=True), n_it, dev)  # device side only: no host sync, as inside DetectPipeline
    fwd_ms, nms_ms = _pct(fwd, 0.5), _pct(nms, 0.5)

    # ---- per-kernel timing: IN SITU (one eager forward, a HIP event between consecutive ops, median of 9 passes) is what the
    # roofline uses; the isolated figure (10 back-to-back launches of one op on warm buffers) is printed beside it ---------------
    model(x)
    torch.cuda.synchronize(dev)
    insitu = eng.profile_ops(iters=9)
    ops = eng.time_ops(iters=10)
    fl = dict(conv_flops(eng))
    if eng._stem is not None:
        fl[eng._stem] = fl[1]  # the fused NCHW stem op computes spec op 1 (0.Conv)
    by = dict(conv_bytes(eng))
    if eng._stem is not None:
        by[eng._stem] = by[1]
    timed = list(zip(eng.timed_order, ops, insitu))  # (plan index, (name, isolated ms), (name, in-situ ms)) in execution order
    assert all(o[0] == s[0] for _, o, s in timed)
    conv_ms = sum(s[1] for i, o, s in timed if i in fl)
    conv_ms_iso = sum(o[1] for i, o, s in timed if i in fl)
    conv_fl = sum(fl[i] for i, _, _ in timed if i in fl)
    other_ms = sum(s[1] for i, o, s in timed if i not in fl)
    conv_by = sum(by[i] for i, _, _ in timed if i in by)
    # The roofline's time base is the CONSERVATIVE one: the event-timed forward of the real (graph) execution minus the non-conv ops, which is what
    # the conv kernel durations of a rocprofv3 trace of this command add up to (profiles/: scripts/rocprof_frac.py); the in-situ per-op sum (one
    # eager pass with an event between ops) runs 2-4 % below it and is kept as `conv_ms_per_step_in_situ`.
    conv_ms_in_situ = conv_ms
    conv_ms = max(conv_ms_in_situ, fwd_ms - other_ms) if parts == 1 else conv_ms_in_situ
    achieved = conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    achieved_bw = conv_by / (conv_ms * 1e-3) / 1e9 if conv_ms > 0 else 0.0  # GB/s
    nconv = sum(1 for i, _, _ in timed if i in fl)
    images_per_plan = eng.spec.B
    # practical per-layer floor: every conv launch at the ceilings this box just showed (copy bandwidth, sustained MFMA rate)
    ceil = None
    if rank == 0:
        try:
            ceil = measured_ceilings(dev)
            floor_ms = sum(max(by[i] / (ceil["copy_gbytes_per_s"] * 1e9), fl[i] / (ceil["mfma_sustained_tflops"] * 1e12)) for i, _, _ in timed if i in fl) * 1e3
            nominal_ms = sum(max(by[i] / (HBM_PEAK_GBS * 1e9), fl[i] / (MFMA_PEAK_TFLOPS * 1e12)) for i, _, _ in timed if i in fl) * 1e3
            ceil.update(per_layer_floor_ms=round(floor_ms, 4), per_layer_floor_ms_at_datasheet_peaks=round(nominal_ms, 4),
                        conv_ms_over_floor=round(conv_ms / floor_ms, 3) if floor_ms > 0 else None,
                        note="floor = sum over conv launches of max(algorithmic bytes / measured copy bandwidth, flops / measured sustained MFMA rate); "
                             "the data-sheet figures (8 TB/s, 2.5 PFLOP/s at 2.4 GHz) are not what the part sustains")
        except Exception as e:  # a probe must never take the headline down
            ceil = {"error": f"{type(e).__name__}: {e}"}
    if a.op_table and rank == 0:
        cfg_of = {}
        ci = iter(eng.conv_cfgs)
        for i, op in enumerate(eng.spec.ops):
            if op["op"] == "conv":
                cfg_of[i] = next(ci)
            elif op["op"] == "bneck":
                cfg_of[i] = "bneck"
        table = []
        for i, (name, iso), (_, ms) in timed:
            row = {"op": name, "cfg": cfg_of.get(i), "ms": round(ms, 5), "ms_isolated": round(iso, 5), "gflop": round(fl.get(i, 0) / 1e9, 3)}
            if i in fl and ms > 0:
                t_hbm = by[i] / (HBM_PEAK_GBS * 1e9) * 1e3
                t_mfma = fl[i] / (MFMA_PEAK_TFLOPS * 1e12) * 1e3
                row.update(tflops=round(fl[i] / (ms * 1e-3) / 1e12, 1), gbytes_per_s=round(by[i] / (ms * 1e-3) / 1e9, 1),
                           bound="hbm" if t_hbm >= t_mfma else "mfma", bound_ms=round(max(t_hbm, t_mfma), 5),
                           pct_of_bound=round(100.0 * max(t_hbm, t_mfma) / ms, 1))
            table.append(row)
        with open(a.op_table, "w") as f:
            json.dump(table, f, indent=1)

    gpu_state = gpu_state_probe(lambda: model(x), dev) if rank == 0 else None

    # ---- secondary: the detect.py pipeline around the hot path (SURVEY 8(f) rank 1 + 3 rows) ---------------------------------
    pipeline = None
    if world == 1 and a.imgsz == 640 and not a.no_pipeline:
        try:
            pipeline = pipeline_probe(model, a.batch, dev, nm)
        except Exception as e:  # the headline metric must not depend on the secondary probe
            pipeline = {"error": f"{type(e).__name__}: {e}"}

    train = None
    if not a.no_train:
        try:
            del model, eng, eng_top
            torch.cuda.empty_cache()
            train = train_probe(a.model, a.batch, a.imgsz, dev, world)
        except Exception as e:  # the headline metric must not depend on the secondary probe
            if world > 1:
                raise  # (a rank that fails alone would leave the others in a collective)
            train = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        imgs = a.batch * world * a.steps
        res = {
            "metric": "images/sec at 640px (yolov5s bs=64), forward+NMS", "value": round(imgs / dt, 1), "unit": "images/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{a.model} inference bs={a.batch}/GPU 3x{a.imgsz}x{a.imgsz} fp16: HIP forward (backbone+neck+Detect) + HIP NMS "
                                   "(conf 0.25, iou 0.45, max_det 1000) + per-image result lists; random-init weights, Detect biases calibrated to a realistic NMS load",
                       "step_mode": "sequential: non_max_suppression(model(x)[0]) per step, host sync inside every step" if a.sequential else
                                    "DetectPipeline: the kernels of the sequential step; the host waits for batch i's counts after batch i+1 is queued (one-deep "
                                    "deferred collect, last batch flushed inside the timed region) and the NMS chain of batch i runs on a high-priority side stream "
                                    "beside forward i+1",
                       "global_batch": a.batch * world, "parallelism": f"replicas x{world} (images independent, no data-path collective); "
                                                                       "the DDP training step with its RCCL all-reduce is the `train` object"},
            "alt_step_mode": {"mode": "DetectPipeline" if a.sequential else "sequential", "ms_per_step": round(alt_dt / a.steps * 1e3, 4),
                              "images_per_sec": round(imgs / alt_dt, 1)},
            "step_ms": stats(step_ms), "forward_ms": round(fwd_ms, 4), "forward_ms_stats": stats(fwd), "nms_ms": round(nms_ms, 4),
            "nms_ms_stats": stats(nms), "nms_us_per_img": round(nms_ms * 1e3 / a.batch, 2),
            "nms_device_ms": round(_pct(nms_dev, 0.5), 4), "nms_device_us_per_img": round(_pct(nms_dev, 0.5) * 1e3 / a.batch, 2),
            "forward_images_per_sec": round(a.batch / (fwd_ms * 1e-3), 1), "detections_per_img": round(ncand, 1),
            # arithmetic intensity of the conv stack at this config = algorithmic flops / algorithmic bytes (144 flop/B for
            # yolov5s bs=64 640^2) is below the ridge (2500 TF / 8 TB/s = 312 flop/B): the stack as a whole is HBM-bound;
            # the MFMA view of the same launches is kept beside it
            "roofline": {"bound": "hbm", "kernel": "y5_conv_{igemm,h3,pw,k3,stem,bneck}_kernel (all conv launches of one forward)",
                         "achieved": round(achieved_bw, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved_bw / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(a, conv_by * parts),
                         "timing": "max(event-timed forward of the real graph execution minus the non-conv ops, in-situ per-op sum): the former is what the conv "
                                   "kernel durations of a rocprofv3 trace of this command add up to; in situ = HIP event between consecutive ops of one "
                                   "eager forward on the launch stream, median of 9 passes, minus the cost of the event record itself",
                         "conv_ms_per_step_in_situ": round(conv_ms_in_situ * parts, 4),
                         "plans_per_step": parts, "images_per_plan": images_per_plan,
                         "algorithmic_gbytes_per_step": round(conv_by * parts / 1e9, 3), "algorithmic_gflop_per_step": round(conv_fl * parts / 1e9, 1),
                         "arithmetic_intensity_flop_per_byte": round(conv_fl / conv_by, 1) if conv_by else None,
                         "mfma_achieved_tflops": round(achieved, 2), "mfma_peak_tflops": MFMA_PEAK_TFLOPS,
                         "mfma_frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                         "conv_ms_per_step": round(conv_ms * parts, 4), "conv_ms_per_step_isolated": round(conv_ms_iso * parts, 4),
                         "launches_per_step": nconv * parts, "other_kernels_ms_per_step": round(other_ms * parts, 4),
                         "measured_ceilings": ceil},
        }
        if gpu_state is not None:
            res["gpu_state"] = gpu_state
        if pipeline is not None:
            res["pipeline"] = pipeline
        if train is not None:
            res["train"] = train
        if cpu is not None:
            res["cpu_baseline"] = cpu
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

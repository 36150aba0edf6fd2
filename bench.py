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
        subs = [op] if op["op"] == "conv" else [op["cv1"], op["cv2"]] if op["op"] == "bneck" else [op["cv1"]] if op["op"] == "sppf_front" else []
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
        elif op["op"] == "sppf_front":   # (charged as the cv1 layer it contains: the pools were never part of the conv figure)
            out.append((i, one(op["cv1"], False) * esize))
    return out


def kernel_src_hash():
    """sha256 (first 16 hex digits) over the kernel sources and the plan builder: what a committed PMC measurement is valid for (the GPU box
    has no .git, so a commit id cannot be checked there; scripts/pmc_forward.sh writes the same hash into its result)."""
    import glob
    import hashlib

    root = os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "yolov5_amd", "csrc", "*.h")) + glob.glob(os.path.join(root, "yolov5_amd", "csrc", "*.hip"))
                    + [os.path.join(root, "yolov5_amd", "engine.py")]):
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def _pmc_plan_mismatch(d, timed_families):
    """None when the counter file was measured on a plan with the kernel families x launches of the timed plan; else what differs."""
    have = d.get("plan_families")
    if have is None:
        return {"reason": "the counter file carries no plan (measured before round 6)"}
    if timed_families is None or have == timed_families:
        return None
    keys = sorted(set(have) | set(timed_families))
    delta = {k: [have.get(k, 0), timed_families.get(k, 0)] for k in keys if have.get(k, 0) != timed_families.get(k, 0)}
    # One near-tie layer that the tuner gives to another family on another box (boxes differ by 5 % in clocks: one plan hash across boxes is not achievable)
    # moves one launch from one family to another and < 1 % of the traffic; that much is tolerated and REPORTED (`plan_delta`).  Anything more -- a family
    # that exists on one side only, more than one swapped layer -- is another plan: the counter figures are withheld.
    if set(have) == set(timed_families) and sum(abs(a - b) for a, b in delta.values()) <= 2:
        d["_plan_delta"] = delta
        return None
    return {"reason": "the counters were collected on another plan", "family: [profiled, timed] launches": delta}


def pmc_traffic(a, conv_by, timed_families=None):
    """HBM bytes of the conv launches of one forward from the memory-side L2 counters.  They cannot be collected from inside this
    process: scripts/pmc_forward.sh runs the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the same forward and the
    result is committed under profiles/pmc/ (FETCH_SIZE doubled: gfx950 correction of MI355X_MICROARCH.md "HBM").  Only
    reported for the configuration it was measured on AND for the kernel sources it was measured with: the file carries the hash of
    csrc/ + engine.py (kernel_src_hash); a measurement of other sources is reported as stale, with null bytes."""
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
    src = f"profiles/pmc/{os.path.basename(path)} (scripts/pmc_forward.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, FETCH x2)"
    have, want = d.get("kernel_src_sha16"), kernel_src_hash()
    if have != want:
        return {"gbytes_per_step": None, "vs_algorithmic": None, "stale": True, "measured_with_kernel_src_sha16": have, "current_kernel_src_sha16": want,
                "stale_gbytes_per_step": round(gb, 3), "source": src}
    mis = _pmc_plan_mismatch(d, timed_families)
    if mis is not None:
        return {"gbytes_per_step": None, "vs_algorithmic": None, "other_plan": mis, "other_plan_gbytes_per_step": round(gb, 3), "source": src}
    return {"gbytes_per_step": round(gb, 3), "vs_algorithmic": round(gb * 1e9 / conv_by, 3) if conv_by else None, "kernel_src_sha16": have,
            "plan_families": d.get("plan_families"), "plan_delta": d.get("_plan_delta"), "source": src}


def pmc_mfma_busy(a, timed_families=None):
    """Counter-based matrix-core utilisation of the conv launches of one forward: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles),
    collected by scripts/pmc_issue_mix.sh (rocprofv3 --kernel-trace --pmc, two SQ passes) and committed under profiles/pmc/ -- the figure
    north_star asks for beside the FLOP-derived fraction.  Same validity rule as `traffic`: only for the configuration and the kernel sources
    (kernel_src_hash) it was measured with; otherwise reported as stale with a null value."""
    import glob

    cands = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc", "r*_pmc_issue_mix.json")))
    if not (a.model == "yolov5s" and a.batch == 64 and a.imgsz == 640 and cands):
        return None, {}
    path = cands[-1]
    try:
        with open(path) as f:
            d = json.load(f)
        frac = float(d["stack"]["mfma_busy_frac"])
    except (OSError, ValueError, KeyError, TypeError):
        return None, {}
    src = f"profiles/pmc/{os.path.basename(path)} (scripts/pmc_issue_mix.sh: rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, ...)"
    have, want = d.get("kernel_src_sha16"), kernel_src_hash()
    per_kernel = {r["kernel"]: (r.get("mfma_busy_frac"), r.get("launches_per_forward")) for r in d.get("kernels", []) if r.get("conv")}
    if have != want:
        return {"value": None, "stale": True, "stale_value": round(frac, 4), "measured_with_kernel_src_sha16": have, "current_kernel_src_sha16": want, "source": src}, {}
    mis = _pmc_plan_mismatch(d, timed_families)
    if mis is not None:
        return {"value": None, "other_plan": mis, "other_plan_value": round(frac, 4), "source": src}, {}
    return {"value": round(frac, 4), "kernel_src_sha16": have, "plan_families": d.get("plan_families"), "plan_delta": d.get("_plan_delta"), "source": src,
            "definition": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles) over the conv launches of one forward (inside the kernels: no launch gaps)"}, per_kernel


def conv_family(cfg):
    """Kernel family a convolution configuration id launches (csrc/conv.hip id space)."""
    if isinstance(cfg, str):
        return {"front": "y5_conv_front_kernel", "bneck": "y5_conv_bneck_kernel", "h3b": "y5_conv_h3b_kernel", "sppf": "y5_sppf_cv1_pool_kernel"}.get(cfg, "y5_conv_" + cfg + "_kernel")
    if cfg is None or cfg < 0:
        return None
    if 84 <= cfg < 88 or cfg == 56 or 14 <= cfg < 22:   # (88 / 89 = the virtual Upsample + Concat loader: implicit-GEMM instantiations)
        return "y5_conv_pw_kernel"
    if 78 <= cfg < 84 or 30 <= cfg < 35:
        return "y5_conv_k3_kernel"
    if 61 <= cfg < 78 or 90 <= cfg < 93:
        return "y5_conv_h3_kernel"
    if 93 <= cfg < 95:
        return "y5_conv_pwk_kernel"
    if cfg == 95:
        return "y5_conv_g8_kernel"
    if cfg == 96:
        return "y5_conv_g8n_kernel"
    return "y5_conv_igemm_kernel"


def plan_hash(plan):
    import hashlib

    return hashlib.sha256(json.dumps([[n, c] for n, c in plan]).encode()).hexdigest()[:16]


def plan_families(plan):
    """{kernel family: launches per forward} of a plan table ([op name, configuration] in launch order): the kernel SET a counter measurement belongs to.
    Conv-class ops by their configuration's family (conv_family); fused heads by the kernel that carries them; everything else by its op kind."""
    fam = {}
    for n, c in plan:
        if n.startswith("conv+decode:"):
            k = "y5_conv_pw_head_kernel" if conv_family(c) == "y5_conv_pw_kernel" else "y5_conv_headk_kernel"
        elif "(fused)" in n:
            continue   # carried by another launch
        elif c is not None and not (isinstance(c, int) and c < 0):
            k = conv_family(c)
        else:
            k = n.split(":")[0].split("(")[0]
        fam[k] = fam.get(k, 0) + 1
    return dict(sorted(fam.items()))


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
    """The oracle (CPU port of the reference forward + NMS, torch-CPU fp32, all host cores) on a bounded sample: yolov5s fused, 8 images of
    3x640x640 per iteration.  `value` = forward + NMS together (the unit of the headline); the two legs are also timed separately, because they
    are not equally representative of the reference: the forward is torch-CPU's own conv / SiLU kernels (what the reference runs on a CPU), the
    NMS leg is the oracle's numpy restatement whose greedy stage is a Python loop -- slower than torchvision's C++ `nms` (absent from this image)."""
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
        n, t_fwd, t_nms = 0, 0.0, 0.0
        while True:
            a0 = time.time()
            z = yo.model_forward(cfg, sd, x)[0]
            a1 = time.time()
            yo.non_max_suppression(z.numpy(), 0.25, 0.45, max_det=1000)
            a2 = time.time()
            t_fwd, t_nms, n = t_fwd + (a1 - a0), t_nms + (a2 - a1), n + bs
            if time.time() - t0 > seconds_budget or n >= 64:
                break
        dt = time.time() - t0
    return {"value": round(n / dt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "forward_images_per_sec": round(n / t_fwd, 3), "nms_us_per_img": round(t_nms / n * 1e6, 1),
            "nms_note": "numpy restatement with a Python greedy loop on random-init rows (few candidates): NOT torchvision's C++ nms; read the forward figure "
                        "as the reference's CPU speed and this one as the checker's",
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


def dev_sync(dev):
    if torch.device(dev).type == "cuda":
        torch.cuda.synchronize(dev)


def fence(dev, world):
    """The contract's bracket around a timed region: device drained, all ranks arrived, device drained again."""
    dev_sync(dev)
    if world > 1:
        dist.barrier()
        dev_sync(dev)


def reduce_max(vals, dev, world):
    """MAX over ranks of a list of python floats (the job's time is its slowest rank's)."""
    if world <= 1:
        return list(vals)
    t = torch.tensor(list(vals), device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t.tolist()]


class StepClock:
    """Per-step device times: HIP events on a GPU, host clock on the CPU emulator (harness dry run)."""

    def __init__(self, dev, n):
        self.cuda = torch.device(dev).type == "cuda"
        self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)] if self.cuda else []
        self.t = []

    def mark(self, i):
        if self.cuda:
            self.ev[i].record()
        else:
            self.t.append(time.perf_counter())

    def ms(self):
        if self.cuda:
            return [self.ev[i].elapsed_time(self.ev[i + 1]) for i in range(len(self.ev) - 1)]
        return [(self.t[i + 1] - self.t[i]) * 1e3 for i in range(len(self.t) - 1)]


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


REF_SUSTAINED_TFLOPS = 1950.0   # the reference box of `value_at_ref_clock` (the middle of what the pool's boxes sustain: 1900 .. 2001 measured in rounds 4 / 5)
TRAIN_GFLOP_PER_IMG = {"yolov5s": 49.3}  # forward + data gradient + weight gradient = 3 x 16.43 GFLOP at 640^2


def train_probe(name, batch, imgsz, dev, world, steps=20, warmup=5, exchange_group=False):
    """BASELINE config 3 per-GPU shape: one training step = train-mode forward (batch-statistics BN) + ComputeLoss + backward
    (+ bucketed RCCL gradient all-reduce overlapped with backward when world > 1, loss * WORLD_SIZE as train.py:404-405) +
    unscale / clip / SGD-Nesterov / EMA (fused), fp16 compute with fp32 master weights, synthetic data.
    exchange_group (N = 1 on a GPU): the model is wrapped by smart_DDP over a ONE-rank `nccl` group, so every bucket of the
    gradient arena really goes through RCCL's launch / wait path (AVG over one rank = identity); reported beside the plain step."""
    from yolov5_amd.loss import ComputeLoss
    from yolov5_amd.torch_utils import ModelEMA, smart_DDP, smart_optimizer
    from yolov5_amd.yolo import DetectionModel

    torch.manual_seed(0)
    m = DetectionModel(name + ".yaml").to(dev).train()
    m.hyp = {"box": 0.05, "cls": 0.5, "cls_pw": 1.0, "obj": 1.0, "obj_pw": 1.0, "anchor_t": 4.0, "fl_gamma": 0.0, "label_smoothing": 0.0}
    compute_loss = ComputeLoss(m)
    ddp = world > 1 or exchange_group
    model = smart_DDP(m) if ddp else m
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
    fence(dev, world)
    clk = StepClock(dev, steps)
    t0 = time.perf_counter()
    clk.mark(0)
    for i in range(steps):
        loss = step()
        clk.mark(i + 1)
    fence(dev, world)
    dt = time.perf_counter() - t0
    per = clk.ms()
    dt = reduce_max([dt], dev, world)[0]
    ar_bytes = nbuckets = None
    if ddp:
        ar_bytes = sum((b.hi - b.lo) * 4 for b in model.buckets)
        nbuckets = len(model.buckets)
    del m, model, opt
    if torch.device(dev).type == "cuda":
        torch.cuda.empty_cache()
    ips = batch * world * steps / dt
    out = {"images_per_sec": round(ips, 1), "ms_per_step": round(dt / steps * 1e3, 2), "steps": steps, "warmup": warmup,
           "step_ms": stats(per), "rccl_ranks": dist.get_world_size() if ddp else 1,
           "allreduce_bytes_per_step": ar_bytes, "allreduce_buckets": nbuckets,
           "workload": f"{name} train step, {batch} img/GPU 3x{imgsz}x{imgsz}, {nt} targets: forward(train BN) + ComputeLoss + backward"
                       f"{' + RCCL all-reduce (overlapped)' if ddp else ''} + clip + SGD + EMA; fp16 compute / fp32 masters",
           "loss": round(float(loss.detach()), 4)}
    gf = TRAIN_GFLOP_PER_IMG.get(name)
    if gf and imgsz == 640:
        tf = ips / world * gf / 1e3  # per GPU
        out["roofline"] = {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4),
                           "traffic": None, "algorithmic_gflop_per_img": gf,
                           "note": "whole training step (all kernels + host), per GPU: 3 x forward conv FLOPs / step time"}
    return out


def train_with_exchange(name, batch, imgsz, dev, steps=20, warmup=5):
    """N = 1 on a GPU: the plain training step (the figure `train.ms_per_step` has always meant: train.py without DDP at one GPU) and the SAME step
    through smart_DDP over a one-rank RCCL group -- the exchange step of the data-parallel path (bucketed all-reduce of the flat gradient arena
    launched from inside the backward plan, utils/torch_utils.py:61-70) with everything but the wire: bucket bookkeeping, RCCL kernel launches on
    its own stream, the event hand-over to the compute stream.  `exposed_us` = step with the exchange - plain step (BASELINE.md section 4 "us exposed")."""
    plain = train_probe(name, batch, imgsz, dev, 1, steps, warmup)
    own_group = False
    try:
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            own_group = True
        ex = train_probe(name, batch, imgsz, dev, 1, steps, warmup, exchange_group=True)
        plain["rccl_ranks"] = ex["rccl_ranks"]
        plain["allreduce_bytes_per_step"] = ex["allreduce_bytes_per_step"]
        plain["allreduce_buckets"] = ex["allreduce_buckets"]
        plain["exchange"] = {"backend": "nccl (RCCL), one rank", "ms_per_step": ex["ms_per_step"], "step_ms": ex["step_ms"],
                             "exposed_us": round((ex["step_ms"]["median"] - plain["step_ms"]["median"]) * 1e3, 1),
                             "loss": ex["loss"],
                             "note": "same step through smart_DDP over a 1-rank RCCL group: real bucket launches / waits, no wire.  A one-rank all-reduce launches NO "
                                     "kernel (rocprofv3): the exposed time is the fixed cost of bringing a second hardware queue into play, not CU starvation "
                                     "(DESIGN.md section 6; Y5_DDP_SYNC=all issues the collectives on the compute stream: 0.00-0.02 ms).  No 1->8 curve has been "
                                     "measured (the driver's 8-GPU tier was unavailable in rounds 1-4)"}
    except Exception as e:  # the plain figure stands; say why the exchange leg is missing
        plain["exchange"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    finally:
        if own_group and dist.is_initialized():
            dist.destroy_process_group()
    return plain


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



def nms_distributions(dev, batch=64, iters=20):
    """NMS alone on the three synthetic prediction distributions of SURVEY 8(d) / BASELINE.md section 4, fp16 (batch, 25200, 85):
    xy ~ U(0,640), wh ~ U(4,104), cls ~ U(0,1), obj = u^8 ("dense": ~16 % of the rows pass obj > 0.25), obj = u^64 ("realistic": ~2 %), and the
    val.py setting conf 0.001 / iou 0.6 / multi_label / max_det 300 on the dense rows (candidate list capped at max_nms = 30000).
    us per image, end to end (host sync included) and device side only (padded result, no sync)."""
    from yolov5_amd.general import non_max_suppression

    g = torch.Generator(device="cpu").manual_seed(0)
    out = {}
    base = torch.rand((batch, 25200, 85), generator=g)
    base[..., 0:2] *= 640.0
    base[..., 2:4] = 4.0 + 100.0 * base[..., 2:4]
    u = base[..., 4].clone()
    for name, pw, kw in (("dense_u8", 8, dict(conf_thres=0.25, iou_thres=0.45, max_det=1000)),
                         ("realistic_u64", 64, dict(conf_thres=0.25, iou_thres=0.45, max_det=1000)),
                         ("val_conf0.001_multilabel", 8, dict(conf_thres=0.001, iou_thres=0.6, max_det=300, multi_label=True))):
        p = base.clone()
        p[..., 4] = u ** pw
        p = p.half().to(dev)
        det = non_max_suppression(p, **kw)
        e2e = event_times(lambda: non_max_suppression(p, **kw), iters, dev)
        devs = event_times(lambda: non_max_suppression(p, padded=True, **kw), iters, dev)
        cand = float((p[..., 4].float() > kw["conf_thres"]).sum()) / batch
        out[name] = {"us_per_img": round(_pct(e2e, 0.5) * 1e3 / batch, 2), "device_us_per_img": round(_pct(devs, 0.5) * 1e3 / batch, 2),
                     "rows_over_obj_threshold_per_img": round(cand, 1), "detections_per_img": round(sum(len(d) for d in det) / batch, 1), **{k: v for k, v in kw.items()}}
        del p
    torch.cuda.empty_cache()
    return out


def config_probe(name, batch, imgsz, dev, steps=20):
    """Secondary lines for BASELINE configs C4 (yolov5x bs=16 1280^2) and C5 (yolov5s-seg bs=32 640^2 incl. process_mask per image): the same
    pipelined forward + NMS step as the headline on that model, its forward alone, and the MFMA fraction of its conv stack."""
    from yolov5_amd.detect_loop import DetectPipeline
    from yolov5_amd.general import non_max_suppression

    model = build_model(name, dev)
    model.model[-1].export = True
    nm = getattr(model.model[-1], "nm", 0)
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.rand((batch, 3, imgsz, imgsz), generator=g).half().to(dev)
    calibrate_head(model, x)
    masks = None
    if nm:
        from yolov5_amd.segment import process_mask_batch

        mask_dtype = [torch.float32]

        def masks(protos, dets):  # segment/predict.py:161-172: per image, upsampled masks of its detections (float32 0/1 as the reference) -- one launch per batch
            return process_mask_batch(protos, dets, (imgsz, imgsz), upsample=True, out_dtype=mask_dtype[0])

    pipe = DetectPipeline(model, 0.25, 0.45, max_det=300 if nm else 1000, nm=nm)
    for _ in range(6):
        r = pipe.submit(x)
    r = pipe.flush()
    ndet = sum(len(d) for d in r) / batch
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        r = pipe.submit(x)
        if nm and r is not None:
            masks(pipe.protos, r)   # prototypes of the batch just collected
    r = pipe.flush()
    if nm:
        masks(pipe.protos, r)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    dt_u8 = None
    if nm:   # the same step with uint8 masks (4x fewer bytes than the reference's float32 0/1 tensors): reported beside, never as the config's figure
        mask_dtype[0] = torch.uint8
        for _ in range(2):
            r = pipe.submit(x)
        r = pipe.flush()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            r = pipe.submit(x)
            if r is not None:
                masks(pipe.protos, r)
        masks(pipe.protos, pipe.flush())
        torch.cuda.synchronize(dev)
        dt_u8 = time.perf_counter() - t0
        mask_dtype[0] = torch.float32
    fwd = event_times(lambda: model(x), 30, dev)
    z = model(x)[0]
    nms = event_times(lambda: non_max_suppression(z, 0.25, 0.45, max_det=300 if nm else 1000, nm=nm), 30, dev)
    eng = next(iter(model._engines.values()))
    fl = sum(f for _, f in conv_flops(eng))
    fwd_ms = _pct(fwd, 0.5)
    out = {"workload": f"{name} inference bs={batch} 3x{imgsz}x{imgsz} fp16: forward + NMS" + (" + process_mask(upsample, float32 masks as the reference) of every image, one launch" if nm else "") + ", DetectPipeline",
           "images_per_sec": round(batch * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4), "forward_ms": round(fwd_ms, 4),
           "nms_us_per_img": round(_pct(nms, 0.5) * 1e3 / batch, 2), "detections_per_img": round(ndet, 1),
           "algorithmic_gflop_per_step": round(fl / 1e9, 1), "forward_mfma_tflops": round(fl / (fwd_ms * 1e-3) / 1e12, 1),
           "forward_mfma_frac": round(fl / (fwd_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)}
    if dt_u8 is not None:
        out["mask_gbytes_per_step_float32"] = round(ndet * batch * imgsz * imgsz * 4 / 1e9, 2)
        out["uint8_masks"] = {"images_per_sec": round(batch * steps / dt_u8, 1), "ms_per_step": round(dt_u8 / steps * 1e3, 4)}
    del model, pipe, eng
    torch.cuda.empty_cache()
    return out


def self_check(model, x, nm, dev, pick=None):
    """UNTIMED parity check of the very plan the timed region ran: images {0, B/2-1, B-1} of the bench batch --
    forward rows against the CPU oracle's fp32 forward inside the oracle's own fp16-storage envelope (x1.5), HIP NMS of the HIP z == oracle
    NMS of the same z bit for bit, DetectPipeline == the sequential step bit for bit.  Never raises: the result rides on the JSON line."""
    import numpy as np

    from oracle import yolo_oracle as yo
    from yolov5_amd.detect_loop import DetectPipeline
    from yolov5_amd.general import non_max_suppression

    B = x.shape[0]
    pick = sorted({0, max(B // 2 - 1, 0), B - 1}) if pick is None else pick
    cfg = {"yolov5s": "yolov5s", "yolov5n": "yolov5n", "yolov5x": "yolov5x", "yolov5m": "yolov5m", "yolov5l": "yolov5l"}.get
    out = {"images": pick}
    z = model(x)[0]
    zc = z[pick].float().cpu().numpy()
    kw = dict(conf_thres=0.25, iou_thres=0.45, max_det=1000, nm=nm)
    dets = non_max_suppression(z, **kw)
    exp = yo.non_max_suppression(zc, 0.25, 0.45, max_det=1000, nm=nm)
    out["nms_bit_exact_vs_oracle"] = bool(all(np.array_equal(dets[i].cpu().numpy(), e) for i, e in zip(pick, exp)))
    out["detections_checked"] = int(sum(len(e) for e in exp))
    pipe = DetectPipeline(model, 0.25, 0.45, max_det=1000, nm=nm)
    x2 = x.flip(0).contiguous()
    seq = [non_max_suppression(model(b)[0], **kw) for b in (x, x2)]
    got = [pipe.submit(x), pipe.submit(x2), pipe.flush()][1:]
    out["pipeline_equals_sequential"] = bool(all(torch.equal(u, v) for s_, p_ in zip(seq, got) for u, v in zip(s_, p_)))
    try:  # forward rows: the oracle needs the model's weights in its own (fused) layout
        sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
        name = getattr(model, "_bench_name", None)
        ocfg = yo.model_cfg(name)
        with torch.no_grad():
            xs = x[pick].float().cpu()
            o32 = yo.model_forward(ocfg, sd, xs)[0].numpy()
            sdh = {k: (v.half() if v.dtype.is_floating_point else v) for k, v in sd.items()}
            o16 = yo.model_forward(ocfg, sdh, xs.half())[0].half().float().numpy()   # (fp16 OUTPUT too, as `model.half()` returns it: at 640 px a coordinate's ulp is 0.25-0.5 px)
        no = zc.shape[-1]

        def errs(a, ref):
            d = np.abs(a.astype(np.float64) - ref.astype(np.float64)).reshape(-1, no)
            r = ref.reshape(-1, no)
            size = np.maximum(r[:, 2], r[:, 3])[:, None].astype(np.float64) + 8.0
            return d[:, :4] / size, d[:, 4:]
        hb, hc = errs(zc, o32)
        yb, yc = errs(o16, o32)
        out.update(box_rel_err_mean=float(f"{hb.mean():.3g}"), box_rel_err_mean_oracle_fp16=float(f"{yb.mean():.3g}"),
                   score_err_mean=float(f"{hc.mean():.3g}"), score_err_mean_oracle_fp16=float(f"{yc.mean():.3g}"),
                   forward_within_fp16_envelope=bool(hb.mean() <= 1.5 * yb.mean() + 1e-6 and hc.mean() <= 1.5 * yc.mean() + 1e-6
                                                     and np.quantile(hb, 0.999) <= 1.5 * np.quantile(yb, 0.999) + 1e-4
                                                     and np.quantile(hc, 0.999) <= 1.5 * np.quantile(yc, 0.999) + 1e-4))
    except Exception as e:
        out["forward_check_error"] = f"{type(e).__name__}: {e}"
    out["ok"] = bool(out.get("nms_bit_exact_vs_oracle") and out.get("pipeline_equals_sequential") and out.get("forward_within_fp16_envelope", False))
    return out


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


def dry_run_tail(a, model, x, step, nm, dev, rank, world):
    """--dry-run-emu: the N-rank skeleton of main() without a GPU -- exactly K timed steps between fences, max over ranks, the DDP training
    probe (smart_DDP over gloo, loss * WORLD_SIZE, bucketed all-reduce of the gradient arena), rank 0 prints ONE JSON line with the keys the
    driver parses.  Sequential step only (DetectPipeline needs HIP streams)."""
    fence(dev, world)
    clk = StepClock(dev, a.steps)
    t0 = time.perf_counter()
    clk.mark(0)
    ndet = 0
    for i in range(a.steps):
        ndet += len(step())
        clk.mark(i + 1)
    fence(dev, world)
    dt = time.perf_counter() - t0
    assert ndet == a.steps * a.batch
    dt = reduce_max([dt], dev, world)[0]
    train = None
    if not a.no_train:
        del model
        train = train_probe(a.model, a.batch, a.imgsz, dev, world, steps=1, warmup=1)
    if rank == 0:
        imgs = a.batch * world * a.steps
        print(json.dumps({"metric": "images/sec at 640px (yolov5s bs=64), forward+NMS", "value": round(imgs / dt, 3), "unit": "images/sec", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f16", "data": "synthetic",
                          "config": {"workload": f"HARNESS DRY RUN on the CPU emulator over gloo: {a.model} bs={a.batch}/rank 3x{a.imgsz}x{a.imgsz} -- not a measurement",
                                     "global_batch": a.batch * world, "parallelism": f"replicas x{world}"},
                          "step_ms": stats(clk.ms()), "train": train, "dry_run": True}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


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
    ap.add_argument("--dry-run-emu", action="store_true", help="HARNESS TEST ONLY (tests/test_bench_harness.py): the rank logic of this file on the CPU "
                    "emulator over gloo with a tiny model -- respawn, rank env, fence, max-over-ranks, DDP train probe, rank-0 JSON; the numbers mean nothing")
    ap.add_argument("--no-configs", action="store_true", help="skip the secondary C4 / C5 lines and the NMS distributions")
    ap.add_argument("--no-selfcheck", action="store_true", help="skip the untimed parity check of the timed plan against the CPU oracle")
    ap.add_argument("--op-table", default="", help="write the per-op timing table (JSON: in-situ and isolated ms, %% of bound) to this path")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        respawn_under_torchrun(a.gpus)
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus and rank == 0:
        print(f"[bench] --gpus {a.gpus} but the launcher started {world} rank(s): reporting n_gpus={world}", file=sys.stderr)
    emu = a.dry_run_emu
    assert emu or torch.cuda.is_available(), "bench.py needs a GPU"

    # ---- CPU baseline first: the GPU is idle while the host cores run the oracle, the rest of the run is GPU work ------------
    cpu = None
    if not a.no_cpu_baseline and world == 1 and not emu:
        cpu = cpu_baseline()
        torch.set_num_threads(min(8, usable_cores()))

    if emu:
        from tests.hipemu import backend as emu_backend

        emu_backend.install()  # CPU tensors -> host-compiled kernels (the tests' seam); collectives over gloo
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if emu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from yolov5_amd.general import non_max_suppression

    model = build_model(a.model, dev)
    model._bench_name = a.model
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

    if emu:
        return dry_run_tail(a, model, x, step, nm, dev, rank, world)

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
        fence(dev, world)
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
        fence(dev, world)
        dt = time.perf_counter() - t0
        assert ndet == a.steps * a.batch, (ndet, a.steps, a.batch)  # every batch of the timed region delivered its per-image results inside it
        if os.environ.get("Y5_BENCH_DEBUG") and rank == 0:
            d = [round((host_t[i + 1] - host_t[i]) * 1e3, 2) for i in range(a.steps)]
            print(f"[bench debug] pipelined={pipelined} wall {dt * 1e3:.2f} ms, host ms per step: {d}", file=sys.stderr)
        return dt, [evs[i].elapsed_time(evs[i + 1]) for i in range(a.steps)]

    dt, step_ms = timed_run(not a.sequential)
    alt_dt, _ = timed_run(a.sequential)
    dt, alt_dt = reduce_max([dt, alt_dt], dev, world)

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
    by = dict(conv_bytes(eng))
    if eng._stem is not None:
        fl[eng._stem] = fl[1]  # the fused NCHW stem op computes spec op 1 (0.Conv)
        by[eng._stem] = by[1]
    if getattr(eng, "_front", None) is not None:
        # the fused front computes spec ops 1..3 (0.Conv, 1.Conv, 2.C3.cv1+cv2): charged the per-layer sum, like every fused launch
        fl[eng._front] = fl[1] + fl[2] + fl[3]
        by[eng._front] = by[1] + by[2] + by[3]
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
    # backbone = layers 0..9 of the yaml (models/yolov5s.yaml:21-33): every launch up to and including 9.SPPF.cv2, in execution order
    bb_ms = bb_fl = 0.0
    bb_done = False
    for i, o, s_ in timed:
        if bb_done:
            break
        bb_ms += s_[1]
        bb_fl += fl.get(i, 0)
        bb_done = "9.SPPF.cv2" in s_[0]
    backbone = {"ms": round(bb_ms * parts, 4), "gflop": round(bb_fl * parts / 1e9, 1), "mfma_tflops": round(bb_fl / (bb_ms * 1e-3) / 1e12, 1) if bb_ms > 0 else None,
                "mfma_frac": round(bb_fl / (bb_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4) if bb_ms > 0 else None,
                "target_mfma_frac": 0.70, "note": "layers 0-9 (stem .. SPPF), in-situ per-op times incl. the pooling launch"} if bb_done else None
    # the DOMINANT kernel: conv launches grouped by (family, configuration id) = one kernel instantiation; the one with the
    # largest in-situ time per forward is named in roofline.kernel with ITS OWN achieved / frac, the stack-wide figure stays beside it as stack_frac
    cfg_of_all = {i: c for i, (_n, c) in enumerate(eng.plan_table())}
    groups = {}
    for i, o, s_ in timed:
        if i not in fl or fl[i] <= 0 or s_[1] <= 0.002:   # (fused-away ops keep a ~0 ms placeholder row)
            continue
        key = (conv_family(cfg_of_all.get(i)), cfg_of_all.get(i))
        gsum = groups.setdefault(key, [0.0, 0.0, 0.0, 0])
        gsum[0] += s_[1]; gsum[1] += fl[i]; gsum[2] += by.get(i, 0); gsum[3] += 1
    dominant = None
    if groups:
        (fam, cid), (gms, gfl, gby, gn) = max(groups.items(), key=lambda kv: kv[1][0])
        tile = ""
        if isinstance(cid, int):
            try:
                import ctypes as _C
                from yolov5_amd import _lib as _l
                bm, bn, bk = _C.c_int(0), _C.c_int(0), _C.c_int(0)
                _l.lib().y5_conv_cfg_info(cid, _C.byref(bm), _C.byref(bn), _C.byref(bk))
                tile = f", {bm.value}x{bn.value} tile, {bk.value}-byte K rows"
            except Exception:
                tile = ""
        dominant = {"name": f"{fam} (configuration {cid}{tile})", "family": fam, "cfg": cid, "launches_per_step": gn * parts, "ms_per_step": round(gms * parts, 4),
                    "share_of_conv_time": round(gms / conv_ms_in_situ, 3) if conv_ms_in_situ > 0 else None,
                    "achieved_tflops": round(gfl / (gms * 1e-3) / 1e12, 1), "frac": round(gfl / (gms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                    "hbm_gbytes_per_s": round(gby / (gms * 1e-3) / 1e9, 1), "hbm_frac": round(gby / (gms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    # which roof bounds THIS group: its algorithmic bytes at 8 TB/s against its FLOPs at 2.5 PFLOP/s (round 6: the dominant group is no longer
                    # an implicit-GEMM configuration -- the 8-phase family split that group up -- but the fused Bottlenecks of P2 / P3, 160 flop/B: HBM-bound)
                    "bound": "hbm" if gby / (HBM_PEAK_GBS * 1e9) >= gfl / (MFMA_PEAK_TFLOPS * 1e12) else "mfma",
                    "arithmetic_intensity_flop_per_byte": round(gfl / gby, 1) if gby else None}
    try:  # which plan the tuner built on this box: two lines are comparable only when this hash agrees
        import hashlib
        plan_sha16 = hashlib.sha256(json.dumps([[n, c] for n, c in eng.plan_table()]).encode()).hexdigest()[:16]
    except Exception:
        plan_sha16 = None
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
        cfg_of = {i: c for i, (_n, c) in enumerate(eng.plan_table())}
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

    selfcheck = None
    if rank == 0 and not a.no_selfcheck:
        try:
            selfcheck = self_check(model, x, nm, dev)
            selfcheck["plan"] = [[n, c] for n, c in eng.plan_table()]
        except Exception as e:
            selfcheck = {"ok": False, "error": f"{type(e).__name__}: {e}"}

    nms_dist = configs = None
    if world == 1 and not a.no_configs and a.model == "yolov5s" and a.batch == 64 and a.imgsz == 640:
        try:
            nms_dist = nms_distributions(dev)
        except Exception as e:
            nms_dist = {"error": f"{type(e).__name__}: {e}"}
        configs = {}
        for key, (nm_, bs_, sz_) in {"C4": ("yolov5x", 16, 1280), "C5": ("yolov5s-seg", 32, 640)}.items():
            try:
                configs[key] = config_probe(nm_, bs_, sz_, dev)
            except Exception as e:  # the headline metric must not depend on a secondary probe
                configs[key] = {"error": f"{type(e).__name__}: {e}"}

    try:   # (the model is released below: what its engines' in-situ refinement swapped is read here)
        insitu_swaps = [list(s) for e in getattr(model, "_engines", {}).values() for s in getattr(e, "insitu_swaps", [])]
    except Exception:
        insitu_swaps = None
    train = None
    if not a.no_train:
        try:
            del model, eng, eng_top
            torch.cuda.empty_cache()
            if world == 1 and not emu and torch.device(dev).type == "cuda":
                train = train_with_exchange(a.model, a.batch, a.imgsz, dev)   # + the exchange step through a one-rank RCCL group
            else:
                train = train_probe(a.model, a.batch, a.imgsz, dev, world)
        except Exception as e:  # the headline metric must not depend on the secondary probe
            if world > 1:
                raise  # (a rank that fails alone would leave the others in a collective)
            train = {"error": f"{type(e).__name__}: {e}"}
    try:
        timed_families = plan_families(eng.plan_table())
    except Exception:
        timed_families = None
    mfma_busy = pmc_mfma_busy(a, timed_families) if rank == 0 else (None, {})
    dom_busy = None
    if rank == 0 and dominant and mfma_busy[1]:
        # the PMC file names kernels by their (mangled or demangled) symbol: match family + the instantiation's share of launches
        # several instantiations of the family were profiled: the dominant one is the one with (nearly) the same number of launches per forward
        fam_rows = {k: v for k, v in mfma_busy[1].items() if dominant["family"].replace("_kernel", "") in k}
        if fam_rows:
            k_best = min(fam_rows, key=lambda k: abs((fam_rows[k][1] or 0) - dominant["launches_per_step"] / max(parts, 1)))
            dom_busy = {"value": fam_rows[k_best][0], "kernel": k_best[:120], "launches_per_forward": fam_rows[k_best][1]}
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
            # roofline.frac is the fraction the contract defines: conv FLOPs / conv kernel time / the dense
            # fp16 MFMA peak.  The HBM view of the same launches (per-layer algorithmic bytes / the same time / 8 TB/s) is kept beside it: with
            # per-layer execution the stack's arithmetic intensity (144 flop/B for yolov5s bs=64 640^2) is below the ridge (312 flop/B), so
            # per-layer it is the HBM figure that says how close each launch is to ITS bound -- but the contract prices the stack against MFMA.
            # roofline.kernel / achieved / frac: the DOMINANT kernel instantiation (largest time per forward) with its own figure; stack_* : all conv
            # launches of one forward (the number rounds 1-3 reported as `frac`); mfma_busy_frac: the counter-based utilisation of the same launches
            "roofline": {"bound": dominant["bound"] if dominant else "mfma", "kernel": dominant["name"] if dominant else "y5_conv_*_kernel (all conv launches of one forward)",
                         "achieved": (dominant["hbm_gbytes_per_s"] if dominant["bound"] == "hbm" else dominant["achieved_tflops"]) if dominant else round(achieved, 2),
                         "peak": (HBM_PEAK_GBS if dominant["bound"] == "hbm" else MFMA_PEAK_TFLOPS) if dominant else MFMA_PEAK_TFLOPS,
                         "unit": ("GB/s" if dominant["bound"] == "hbm" else "TFLOP/s") if dominant else "TFLOP/s",
                         "frac": (dominant["hbm_frac"] if dominant["bound"] == "hbm" else dominant["frac"]) if dominant else round(achieved / MFMA_PEAK_TFLOPS, 4),
                         "frac_definition": "the dominant kernel group (largest in-situ time per forward) against the roof that bounds IT: algorithmic bytes of its launches / their "
                                            "in-situ time / 8000 GB/s when its arithmetic intensity is below the ridge (312 flop/B), else its conv FLOPs / time / 2500 TFLOP/s "
                                            "(dominant_kernel carries both views); stack_frac = conv FLOPs of ALL conv launches of one forward / their time / 2500 TFLOP/s "
                                            "(rounds 1-3 reported that one as frac; rounds 4-5 the MFMA view of an implicit-GEMM group)",
                         "dominant_kernel": dominant,
                         "stack_kernels": "y5_conv_{front,igemm,h3,h3b,pw,k3,stem,bneck}_kernel (all conv launches of one forward)",
                         "stack_achieved": round(achieved, 2), "stack_frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                         "mfma_busy_frac": mfma_busy[0], "mfma_busy_frac_dominant_kernel": dom_busy,
                         "traffic": pmc_traffic(a, conv_by * parts, timed_families),
                         "whole_step_frac": round(a.batch * world * a.steps / dt / world * conv_fl * parts / a.batch / 1e12 / MFMA_PEAK_TFLOPS, 4),
                         "backbone_l0_9": backbone,
                         "hbm_achieved_gbytes_per_s": round(achieved_bw, 1), "hbm_peak_gbytes_per_s": HBM_PEAK_GBS, "hbm_frac": round(achieved_bw / HBM_PEAK_GBS, 4),
                         "timing": "max(event-timed forward of the real graph execution minus the non-conv ops, in-situ per-op sum): the former is what the conv "
                                   "kernel durations of a rocprofv3 trace of this command add up to; in situ = HIP event between consecutive ops of one "
                                   "eager forward on the launch stream, median of 9 passes, minus the cost of the event record itself",
                         "conv_ms_per_step_in_situ": round(conv_ms_in_situ * parts, 4),
                         "plans_per_step": parts, "images_per_plan": images_per_plan,
                         "algorithmic_gbytes_per_step": round(conv_by * parts / 1e9, 3), "algorithmic_gflop_per_step": round(conv_fl * parts / 1e9, 1),
                         "arithmetic_intensity_flop_per_byte": round(conv_fl / conv_by, 1) if conv_by else None,
                         "conv_ms_per_step": round(conv_ms * parts, 4), "conv_ms_per_step_isolated": round(conv_ms_iso * parts, 4),
                         "launches_per_step": nconv * parts, "other_kernels_ms_per_step": round(other_ms * parts, 4),
                         "measured_ceilings": ceil},
        }
        try:  # which plan the tuner built on this box: two lines are comparable only when this hash agrees
            res["config"]["plan_sha16"] = plan_sha16
            res["config"]["kernel_src_sha16"] = kernel_src_hash()
        except Exception:
            pass
        if selfcheck is not None:
            res["selfcheck"] = selfcheck
        if nms_dist is not None:
            res["nms_distributions"] = nms_dist
        if configs is not None:
            res["configs"] = configs
        if gpu_state is not None:
            res["gpu_state"] = gpu_state
        # Box normalisation: boxes of the pool sustain 1.90 .. 2.00 PFLOP/s of back-to-back MFMAs (a clock / power-cap property, measured
        # by this run's own probe); `value` stays the raw throughput, `value_at_ref_clock` = value x (1950 / mfma_sustained_tflops), the factor clamped to +-10 %.
        # Round-over-round claims quote the normalised figure; the contract's `value` is the raw one.
        try:
            sus = float(ceil["mfma_sustained_tflops"])
            fac = min(1.10, max(0.90, REF_SUSTAINED_TFLOPS / sus))
            res["value_at_ref_clock"] = round(res["value"] * fac, 1)
            res["ref_clock"] = {"ref_mfma_sustained_tflops": REF_SUSTAINED_TFLOPS, "this_box_mfma_sustained_tflops": sus, "factor": round(fac, 4),
                                "shader_clock_ghz_under_mfma": ceil.get("shader_clock_ghz_under_mfma")}
            # the same FLOPs against what THIS box sustains on back-to-back MFMAs (MFMA-dense code clocks the part to 0.65-0.75 of the
            # 2.4 GHz the 2.5 PFLOP/s peak assumes); `frac` / `stack_frac` keep the contract's 2.5 PFLOP/s denominator
            if dominant:
                res["roofline"]["frac_of_sustained"] = round(float(dominant["achieved_tflops"]) / sus, 4)   # (MFMA view of the dominant group)
            res["roofline"]["stack_frac_of_sustained"] = round(float(res["roofline"]["stack_achieved"]) / sus, 4)
            sm = [q.get("sclk_mhz") for q in (gpu_state or {}).get("samples", []) if isinstance(q, dict) and q.get("sclk_mhz")]
            res["config"]["gpu_state"] = {"sclk_mhz": max(sm) if sm else None, "mfma_sustained_tflops": sus}
        except Exception:
            res["value_at_ref_clock"] = None
        try:   # where the plan's tile choices came from: the database shipped with the kernels (yolov5_amd/tune_db.json, engine._load_tune_cache) or races run here
            from yolov5_amd import engine as _eng

            res["config"]["tile_choices"] = {"from_shipped_db": _eng.TUNE_STATS["db_entries"], "races_run_here": _eng.TUNE_STATS["races"],
                                             "in_situ_swaps": insitu_swaps}
        except Exception:
            pass
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

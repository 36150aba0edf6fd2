"""Inference engine: turns a YOLOv5 module tree (yolov5_amd.yolo / yolov5_amd.common classes) into a static list
of HIP kernel launches (`PlanSpec`), then materialises it on one MI355X as a C-side execution plan
(include/yolov5_hip.h: y5_plan_*).  Replaces the Python layer walk of the reference's
`BaseModel._forward_once` (models/yolo.py:160-170) for eval-mode forward.

Design:
  * activations are NHWC; every tensor is a channel slice (buffer, c_off, C) with pixel stride = buffer width, so
    `torch.cat` never runs: producers write straight into their slice of the consumer's concat buffer
    (C3 common.py:246, SPPF :340, Concat :453), and `nn.Upsample` is a second, 2x-replicated store of the producer.
  * C3's cv1 and cv2 (same input) are one GEMM with N = 2*c_; Bottleneck's residual add is the conv epilogue.
  * BN is folded into the conv at plan build (utils/torch_utils.py:224-254 semantics), bias + SiLU fused.
  * one host call replays the whole forward (y5_plan_run_range) on the caller's current HIP stream.

`build_plan_spec` is device independent (pure shape/graph logic, unit-tested on CPU); `Engine` needs a GPU and the
built library and raises otherwise -- there is no CPU execution path.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

import torch
from torch import nn

from . import _lib
from .packing import fuse_conv_bn_weights, pack_conv_weight, pack_stem_weight


@dataclass
class Buf:
    id: int
    H: int
    W: int
    C: int
    name: str = ""


@dataclass(frozen=True)
class TRef:
    """Channel slice [c_off, c_off+C) of NHWC buffer `buf` (pixel stride = that buffer's C)."""
    buf: int
    c_off: int
    C: int
    H: int
    W: int


@dataclass
class PlanSpec:
    B: int
    in_ch: int
    in_hw: tuple
    bufs: list = field(default_factory=list)
    ops: list = field(default_factory=list)
    outputs: dict = field(default_factory=dict)

    def new_buf(self, H, W, C, name=""):
        b = Buf(len(self.bufs), H, W, C, name)
        self.bufs.append(b)
        return TRef(b.id, 0, C, H, W)

    def ld(self, t: TRef) -> int:
        return self.bufs[t.buf].C


def _slice(t: TRef, off: int, c: int) -> TRef:
    assert 0 <= off and off + c <= t.C
    return TRef(t.buf, t.c_off + off, c, t.H, t.W)


def _conv_out_hw(h, w, k, s, p):
    return (h + 2 * p[0] - k[0]) // s[0] + 1, (w + 2 * p[1] - k[1]) // s[1] + 1


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class _Planner:
    """Walks the module tree once and emits PlanSpec ops."""

    def __init__(self, model, B, ch, H, W, want_raw=False, fuse_bneck=False, virtual_up=False):
        from . import common, yolo

        self.cm, self.yo = common, yolo
        self.model = model
        self.spec = PlanSpec(B, ch, (H, W))
        self.want_raw = want_raw
        self.fuse_bneck = fuse_bneck  # fp16 plans: Bottlenecks of 32-channel C3 blocks as one launch (csrc/conv_bneck.h)
        # fp16 plans: `nn.Upsample(2) -> Concat -> C3` (models/yolov5s.yaml:36-38,41-43) without the 2x replica -- the C3's cv1+cv2 GEMM reads the
        # low-resolution tensor for the first half of K (csrc/conv_igemm.h UP2, configuration ids 88 / 89)
        self.virtual_up = virtual_up

    # ---- leaf emitters ----------------------------------------------------------------------------
    def conv(self, mods, x: TRef, dest: TRef | None = None, res: TRef | None = None, up2: TRef | None = None,
             act=True, name="", view=None, c2_store=None, split=None, emit=True, x_up=None):
        """mods: list of Conv-like modules stacked along output channels (same input, same k/s/p).
        split = (n, hi): output channels [0, n) go to `dest` (n channels wide), channels [n, c2) to the slice `hi`."""
        m0 = mods[0]
        cv = m0.conv if hasattr(m0, "conv") else m0
        k, s, p = _pair(cv.kernel_size), _pair(cv.stride), _pair(cv.padding)
        c2 = sum((m.conv if hasattr(m, "conv") else m).out_channels for m in mods)
        oh, ow = _conv_out_hw(x.H, x.W, k, s, p)
        cst = c2 if c2_store is None else c2_store
        if dest is None:
            dest = self.spec.new_buf(oh, ow, cst if split is None else split[0], name)
        if split is None:
            assert dest.C == cst and (dest.H, dest.W) == (oh, ow), (name, dest, cst, oh, ow)
        else:
            assert up2 is None and res is None and dest.C == split[0] and split[1].C == cst - split[0], (name, dest, split)
            up2 = split[1]
        op = dict(op="conv", mods=mods, x=x, y=dest, res=res, y2=up2, k=k, s=s, p=p, act=act, c2=c2, c2_store=cst, name=name, view=view,
                  split_n=0 if split is None else split[0])
        if x_up is not None:   # (low-resolution TRef, channels): input channels [0, c_up) are read from it at (oh >> 1, ow >> 1)
            assert k == (1, 1) and s == (1, 1) and p == (0, 0) and res is None and (up2 is None or split is not None), name
            op["x_up"], op["up_c"] = x_up   # (TRef at the top level of the op: the "no other reader of this buffer" scans of the fusion races see it)
        if emit:
            self.spec.ops.append(op)
            return dest
        return op

    def bottleneck(self, m, x: TRef, dest: TRef, tmp: TRef):
        """Bottleneck (common.py:164-181) with e=1.0; dest may alias x (in-place residual)."""
        t = self.conv([m.cv1], x, tmp, name="b.cv1")
        return self.conv([m.cv2], t, dest, res=x if m.add else None, name="b.cv2")

    def _bneck_fusable(self, m, c_, x):
        # c_ = 32: four waves, stage + t (conv_bneck.h); c_ = 64 (round 4): eight waves with t aliased onto the stage -- Y5_FUSED_BNECK64 = 0 keeps
        # the two-launch form there
        # c_ = 128 (round 5): the 1x1 as a GEMM-1 phase of the halo-resident 3x3 (conv_h3b.h), any H x W -- Y5_FUSED_BNECK128 = 0 keeps two launches
        ok_c = (c_ == 32 or (c_ == 64 and os.environ.get("Y5_FUSED_BNECK64", "1") != "0")
                or (c_ == 128 and os.environ.get("Y5_FUSED_BNECK128", "1") != "0"))
        if not self.fuse_bneck or not ok_c or (c_ != 128 and (x.H % 4 or x.W % 8)) or not len(m.m):
            return False
        #  conv_h3b.h runs ONE workgroup per CU on 10 x 20 output tiles and was measured at bs 64 only: with fewer tiles than half the CUs of an
        # MI355X (small batches) the two-launch form, whose kernels the tuner picks per shape, is kept -- Y5_FUSED_BNECK128 = force overrides
        if c_ == 128 and os.environ.get("Y5_FUSED_BNECK128", "1") != "force" and self.spec.B * (-(-x.H // 10)) * (-(-x.W // 20)) < 128:
            return False
        for b in m.m:
            c1, c2 = b.cv1.conv, b.cv2.conv
            if not (c1.in_channels == c1.out_channels == c2.in_channels == c2.out_channels == c_ and _pair(c1.kernel_size) == (1, 1)
                    and _pair(c2.kernel_size) == (3, 3) and _pair(c2.stride) == (1, 1) and _pair(c2.padding) == (1, 1)):
                return False
        return True

    def c3(self, m, x: TRef, dest: TRef | None, up2=None, name="C3", x_up=None):
        c_ = m.cv1.conv.out_channels
        cat = self.spec.new_buf(x.H, x.W, 2 * c_, name + ".cat")
        if self._bneck_fusable(m, c_, x):
            # cv1's half of the GEMM lands in a buffer of its own (split store), every Bottleneck is ONE launch that reads one buffer
            # and writes another (a tile needs its neighbours' input pixels: not in place), the last one writes its slice of `cat`
            ping = [self.spec.new_buf(x.H, x.W, c_, name + ".a0")]
            if len(m.m) > 1:
                ping.append(self.spec.new_buf(x.H, x.W, c_, name + ".a1"))
            self.conv([m.cv1, m.cv2], x, ping[0], name=name + ".cv1+cv2", split=(c_, _slice(cat, c_, c_)), x_up=x_up)
            src = ping[0]
            for i, b in enumerate(m.m):
                dst = _slice(cat, 0, c_) if i == len(m.m) - 1 else ping[(i + 1) % 2]
                self.spec.ops.append(dict(op="bneck", x=src, y=dst, add=bool(b.add), name=f"{name}.m{i}",
                                          cv1=self.conv([b.cv1], src, dst, name="b.cv1", emit=False),
                                          cv2=self.conv([b.cv2], src, dst, name="b.cv2", emit=False)))
                src = dst
            return self.conv([m.cv3], cat, dest, up2=up2, name=name + ".cv3")
        self.conv([m.cv1, m.cv2], x, cat, name=name + ".cv1+cv2", x_up=x_up)  # one GEMM, N = 2*c_
        a = _slice(cat, 0, c_)
        if len(m.m):
            tmp = self.spec.new_buf(x.H, x.W, c_, name + ".tmp")
            for b in m.m:
                self.bottleneck(b, a, a, tmp)
        return self.conv([m.cv3], cat, dest, up2=up2, name=name + ".cv3")

    def sppf(self, m, x: TRef, dest, up2=None, name="SPPF"):
        c_ = m.cv1.conv.out_channels
        k = m.m.kernel_size if isinstance(m.m.kernel_size, int) else m.m.kernel_size[0]
        cat = self.spec.new_buf(x.H, x.W, 4 * c_, name + ".cat")
        cv1 = m.cv1.conv
        # fp16 inference plans (round 5): cv1 + the three pools as ONE launch (csrc/conv_sppf.h: a workgroup owns an image's H x W pixels for 64 output
        # channels and pools its own GEMM result in LDS) -- Y5_FUSED_SPPF = 0 keeps cv1 and y5_sppf_pool apart
        #  the fused launch has B x c_ / 64 workgroups (4 at batch 1 for yolov5s): below 64 the two-launch form is kept -- Y5_FUSED_SPPF = force overrides
        enough = self.spec.B * (c_ // 64) >= 64 or os.environ.get("Y5_FUSED_SPPF", "1") == "force"
        if (self.fuse_bneck and os.environ.get("Y5_FUSED_SPPF", "1") != "0" and enough and x.H * x.W <= 416 and c_ % 64 == 0 and cv1.in_channels % 32 == 0
                and _pair(cv1.kernel_size) == (1, 1) and _pair(cv1.stride) == (1, 1) and _pair(cv1.padding) == (0, 0) and k % 2 == 1):
            self.spec.ops.append(dict(op="sppf_front", x=x, y=_slice(cat, 0, c_), buf=cat, C=c_, k=k, name=name + ".cv1+pool",
                                      cv1=self.conv([m.cv1], x, _slice(cat, 0, c_), name=name + ".cv1", emit=False)))
        else:
            self.conv([m.cv1], x, _slice(cat, 0, c_), name=name + ".cv1")
            self.spec.ops.append(dict(op="sppf_pool", buf=cat, C=c_, k=k))
        return self.conv([m.cv2], cat, dest, up2=up2, name=name + ".cv2")

    def proto(self, m, x: TRef):
        """Proto (common.py:1104-1117): cv3(cv2(up2x(cv1(x)))); the upsample is cv1's replicated store."""
        c_ = m.cv1.conv.out_channels
        up = self.spec.new_buf(2 * x.H, 2 * x.W, c_, "proto.up")
        self.conv([m.cv1], x, None, up2=up, name="proto.cv1")
        t = self.conv([m.cv2], up, None, name="proto.cv2")
        return self.conv([m.cv3], t, None, name="proto.cv3")

    # ---- whole model ----------------------------------------------------------------------------------
    def run(self):
        cm, yo, spec = self.cm, self.yo, self.spec
        layers = list(self.model.model)
        n = len(layers)
        H, W = spec.in_hw

        def src_ids(i, m):
            f = m.f
            fl = [f] if isinstance(f, int) else list(f)
            return [i - 1 if j == -1 else (j if j >= 0 else i + j) for j in fl]

        # pass 1: shapes
        shp = []
        for i, m in enumerate(layers):
            ins = [(spec.in_ch, H, W)] if i == 0 else [shp[j] for j in src_ids(i, m)]
            c, h, w = ins[0]
            if isinstance(m, cm.Conv):
                cv = m.conv
                oh, ow = _conv_out_hw(h, w, _pair(cv.kernel_size), _pair(cv.stride), _pair(cv.padding))
                shp.append((cv.out_channels, oh, ow))
            elif isinstance(m, cm.C3):
                shp.append((m.cv3.conv.out_channels, h, w))
            elif isinstance(m, cm.SPPF):
                shp.append((m.cv2.conv.out_channels, h, w))
            elif isinstance(m, nn.Upsample):
                if float(m.scale_factor) != 2.0 or m.mode != "nearest":
                    raise NotImplementedError("only nn.Upsample(scale_factor=2, mode='nearest') is supported")
                shp.append((c, 2 * h, 2 * w))
            elif isinstance(m, cm.Concat):
                if m.d != 1:
                    raise NotImplementedError("Concat along a dimension other than channels")
                shp.append((sum(s[0] for s in ins), h, w))
            elif isinstance(m, yo.Detect):
                shp.append(None)
            else:
                raise NotImplementedError(f"layer {i}: {type(m).__name__} has no HIP implementation")

        # pass 2: concat homes + fusable upsamples
        consumers = {i: [] for i in range(n)}
        for i, m in enumerate(layers):
            if i:
                for j in src_ids(i, m):
                    consumers[j].append(i)
        cat_bufs, home = {}, {}
        for i, m in enumerate(layers):
            if isinstance(m, cm.Concat):
                c, h, w = shp[i]
                cat_bufs[i] = spec.new_buf(h, w, c, f"cat{i}")
                off = 0
                for j in src_ids(i, m):
                    if j not in home:
                        home[j] = _slice(cat_bufs[i], off, shp[j][0])
                    off += shp[j][0]
        fused_up = {}  # producer layer -> upsample layer
        for i, m in enumerate(layers):
            if isinstance(m, nn.Upsample):
                j = src_ids(i, m)[0]
                if isinstance(layers[j], (cm.Conv, cm.C3, cm.SPPF)):
                    fused_up[j] = i

        # virtual upsample: Upsample u (fused into producer j) -> first source of Concat c -> only reader of c is a C3 k that is not the fused-Bottleneck form
        virt = {}      # upsample layer u -> (concat layer c, C3 layer k)
        virt_cat = {}  # C3 layer k -> (producer layer j, channels)
        if self.virtual_up:
            for j, u in fused_up.items():
                cu, hu, wu = shp[u]
                if len(consumers[u]) != 1 or cu % 64 or hu % 2 or wu % 2:
                    continue
                c = consumers[u][0]
                if not isinstance(layers[c], cm.Concat) or src_ids(c, layers[c])[0] != u or len(consumers[c]) != 1:
                    continue
                k = consumers[c][0]
                mk = layers[k]
                if not isinstance(mk, cm.C3) or src_ids(k, mk) != [c] or shp[c][0] <= cu or (shp[c][0] - cu) % 64:
                    continue
                c_k = mk.cv1.conv.out_channels
                if self.fuse_bneck and c_k == 32:   # that C3 stores its cv1 / cv2 halves separately (split store): keep the replica
                    continue
                virt[u] = (c, k)
                virt_cat[k] = (j, cu)

        # pass 3: emit
        out = {}
        x0 = spec.new_buf(H, W, 4 if spec.in_ch <= 4 else (spec.in_ch + 7) // 8 * 8, "input_nhwc")
        spec.ops.append(dict(op="to_nhwc", dst=x0, C=spec.in_ch))
        for i, m in enumerate(layers):
            ins = [x0] if i == 0 else [out[j] for j in src_ids(i, m)]
            dest = home.get(i)
            up2 = None
            if i in fused_up:
                u = fused_up[i]
                c, h, w = shp[u]
                if u in virt:
                    out[u] = "virtual"   # no replica: the C3 behind the Concat reads this layer's own output (emitted below)
                else:
                    up2 = home.get(u) or spec.new_buf(h, w, c, f"up{u}")
                    out[u] = up2
            if isinstance(m, cm.Conv):
                view = None
                if i == 0:
                    view = "first"
                out[i] = self.conv([m], ins[0], dest, up2=up2, name=f"{i}.Conv", view=view)
            elif isinstance(m, cm.C3):
                x_up = None
                if i in virt_cat:
                    j, cu = virt_cat[i]
                    x_up = (out[j], cu)
                kw = {} if x_up is None else {"x_up": x_up}   # (subclassed planners without the virtual form never see the keyword)
                out[i] = self.c3(m, ins[0], dest, up2, name=f"{i}.C3", **kw)
            elif isinstance(m, cm.SPPF):
                out[i] = self.sppf(m, ins[0], dest, up2, name=f"{i}.SPPF")
            elif isinstance(m, nn.Upsample):
                if i not in out:
                    c, h, w = shp[i]
                    d = dest or spec.new_buf(h, w, c, f"up{i}")
                    spec.ops.append(dict(op="upsample", src=ins[0], dst=d))
                    out[i] = d
            elif isinstance(m, cm.Concat):
                buf = cat_bufs[i]
                off = 0
                for t, j in zip(ins, src_ids(i, m)):
                    if isinstance(t, str):   # "virtual": the upsampled replica is never materialised (its slice of the buffer stays unwritten)
                        off += shp[j][0]
                        continue
                    want = _slice(buf, off, t.C)
                    if t != want:
                        spec.ops.append(dict(op="copy", src=t, dst=want))
                    off += t.C
                out[i] = buf
            elif isinstance(m, yo.Detect):
                self.detect(m, ins)
        # opt-in since round 2: the A/B on MI355X (scripts/ab_head_branch.sh) shows no gain any more, and without overlapping launches the
        # per-kernel durations of a rocprofv3 trace add up to the forward
        if _lib.experimental("head_branch"):
            self._schedule_heads()
        return spec

    def _schedule_heads(self):
        """The Detect convolution + decode of every pyramid level but the last (and the Segment prototype branch) depend only on
        one feature map and feed nothing but the final output (models/yolo.py:83-108,141-149 run them after the whole neck only
        because Detect is the last module).  They are moved right behind the op that completes their input and marked `side`: the
        plan runs them on its second stream (y5_plan_set_branch) -- HBM-bound head work overlapped with the tail of the neck."""
        ops = self.spec.ops
        groups = {}  # group key -> op indices (in order); keys: "proto", 0, 1, ... (pyramid level)
        for j, op in enumerate(ops):
            if "head" in op:
                groups.setdefault(op["head"], []).append(j)
        levels = [k for k in groups if k != "proto"]
        if len(levels) < 2:
            return
        last = max(levels)
        head_ops = {j for idx in groups.values() for j in idx}

        def writes(op, t):
            outs = [op.get(k) for k in ("y", "y2", "dst", "buf")]
            return any(hasattr(o, "buf") and o.buf == t.buf and o.c_off < t.c_off + t.C and t.c_off < o.c_off + o.C for o in outs)

        moved, after = set(), {}
        for key, idx in groups.items():
            if key == last:
                continue
            x = ops[idx[0]]["x"] if "x" in ops[idx[0]] else None
            if x is None:
                continue
            prod = max((p for p in range(idx[0]) if p not in head_ops and writes(ops[p], x)), default=None)
            if prod is None:
                continue
            after.setdefault(prod, []).extend(idx)
            moved.update(idx)
        if not moved:
            return
        new = []
        for j, op in enumerate(ops):
            if j in moved:
                continue
            new.append(op)
            for k in after.get(j, ()):
                ops[k]["side"] = True
                new.append(ops[k])
        self.spec.ops[:] = new

    def detect(self, m, xs):
        spec = self.spec
        B = spec.B
        no, na = m.no, m.na
        nm = getattr(m, "nm", 0) if isinstance(m, self.yo.Segment) else 0
        nrows = sum(na * x.H * x.W for x in xs)
        spec.outputs["z"] = dict(shape=(B, nrows, no))
        if isinstance(m, self.yo.Segment):
            n0 = len(spec.ops)
            p = self.proto(m.proto, xs[0])
            spec.ops.append(dict(op="to_nchw", src=p, out="proto"))
            spec.outputs["proto"] = dict(shape=(B, p.C, p.H, p.W))
            for op in spec.ops[n0:]:
                op["head"] = "proto"
        row_off = 0
        for i, x in enumerate(xs):
            npad = (na * no + 31) // 32 * 32
            n0 = len(spec.ops)
            lg = self.conv([m.m[i]], x, None, act=False, name=f"detect.m{i}", c2_store=npad)
            raw = None
            if self.want_raw:
                raw = f"raw{i}"
                spec.outputs[raw] = dict(shape=(B, na, x.H, x.W, no))
            spec.ops.append(dict(op="decode", x=lg, level=i, ny=x.H, nx=x.W, na=na, no=no, nm=nm, row_off=row_off,
                                 nrows=nrows, raw=raw))
            for op in spec.ops[n0:]:
                op["head"] = i
            row_off += na * x.H * x.W


def build_plan_spec(model, B, ch, H, W, want_raw=False, fuse_bneck=False, virtual_up=False) -> PlanSpec:
    """Device-independent kernel schedule for `model` (a yolov5_amd.yolo.BaseModel) at input (B,ch,H,W)."""
    return _Planner(model, B, ch, H, W, want_raw, fuse_bneck, virtual_up).run()


# ----------------------------------------------------------------------------------------------------------
def folded_weights(mods):
    """Stack Conv-like modules along output channels; fold BN (eval statistics) when present. fp32 (w, b)."""
    ws, bs = [], []
    for m in mods:
        if hasattr(m, "conv"):
            cv = m.conv
            if hasattr(m, "bn") and m.bn is not None:
                bn = m.bn
                w, b = fuse_conv_bn_weights(cv.weight.detach(), None if cv.bias is None else cv.bias.detach(), bn.weight.detach(),
                                            bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
            else:
                w = cv.weight.detach().float()
                b = torch.zeros(cv.out_channels, device=w.device) if cv.bias is None else cv.bias.detach().float()
        else:  # bare nn.Conv2d (Detect.m[i])
            w = m.weight.detach().float()
            b = torch.zeros(m.out_channels, device=w.device) if m.bias is None else m.bias.detach().float()
        ws.append(w)
        bs.append(b)
    return torch.cat(ws, 0), torch.cat(bs, 0)


_TUNE_CACHE = {}  # conv descriptor (shape/dtype/strides) -> fastest tile configuration id, per process
_LAST_RACE = [None, None]  # key and (best, runner-up) of the latest autotune_conv call (Engine._refine_in_situ)
_K3PW_MARK = -779          # key prefix: which build of the 3x3 + pointwise fusion (configuration 34 / 81) won the race for that shape
_HEAD_MARK = -778          # key prefix: which build of the fused Detect head (configuration 56 / 87) won the race for that shape
_INSITU_MARK = -777        # key suffix (_INSITU_MARK, plan index): the configuration the in-situ refinement settled on for that op of that plan


SK_CFGS = frozenset(range(57, 61))  # stream-K configurations: they share one registered workspace (include/yolov5_hip.h)


def autotune_conv(lib, d, ptrs, st, exclude=()):
    """Measure-don't-guess tile selection: HIP-event timing of every kernel configuration on the real buffers
    (cached per descriptor for the life of the process).  ptrs = (x, w, bias, residual, y, y2) as c_void_p / None.
    exclude: configuration ids that must not be chosen for this launch (stream-K on ops that overlap with others in time)."""
    key = tuple(getattr(d, f) for f, _ in _lib.ConvDesc._fields_ if f not in ("cfg", "max_blocks")) + tuple(p is None or p.value is None for p in ptrs) \
        + (bool(exclude),)
    _load_tune_cache()
    rank = int(os.environ.get("Y5_TUNE_RANK", "0"))  # 1: the RUNNER-UP of every race (parity tests cover the plans a near-tie could select)
    hit = _TUNE_CACHE.get(key)
    _LAST_RACE[:] = [key, hit]
    if hit is not None and (rank == 0 or hit[1] >= 0):
        return hit[1] if rank else hit[0]
    ncfg = lib.y5_conv_num_cfgs() if d.dtype == _lib.Y5_F16 else 4
    ms = C.c_float(0)
    best, best_ms, second, second_ms = -1, float("inf"), -1, float("inf")
    bm, bn, kb = C.c_int(0), C.c_int(0), C.c_int(0)
    iters = int(os.environ.get("Y5_AUTOTUNE_ITERS", "5"))
    skip = set()
    for part in os.environ.get("Y5_AUTOTUNE_SKIP", "").split(","):  # e.g. "14-21,30-34": keep kernel families out of the race
        if part:
            lo, _, hi = part.partition("-")
            skip.update(range(int(lo), int(hi or lo) + 1))
    if not _lib.experimental("h3_s2"):
        # the halo-resident 3x3 at STRIDE 2 (ids 61-77 on stride-2 layers, the small-tile ids 90-92) lost every race of round 5
        # (profiles/r05/r05_ab_h3_stride2.log): at parity, reachable by id, no longer timed at every plan build
        skip.update(range(90, 93))
        if d.SH == 2 and d.KH == 3:
            skip.update(range(61, 78))
    timed = []
    TUNE_STATS["races"] += 1
    for cfg in range(ncfg):
        if cfg in skip or cfg in exclude:
            continue
        lib.y5_conv_cfg_info(cfg, C.byref(bm), C.byref(bn), C.byref(kb))
        if bn.value >= 2 * d.Npad and bn.value > 32:
            continue  # more than half of the tile's channels would be padding
        d.cfg = cfg
        rc = lib.y5_conv2d_time(C.byref(d), *ptrs, iters, st, C.byref(ms))
        if rc != 0:
            continue  # configuration not applicable to this shape
        timed.append((ms.value, cfg))
    if not timed:
        _lib.check(-2, lib)
    # Reproducible choices at near ties (three runs of one build gave three plans): the front-runners of the coarse pass are timed
    # again with 4x the iterations, in ascending id order, and a configuration only displaces one with a LOWER id when it is faster by more than the
    # hysteresis (Y5_TUNE_HYST, 2 %) -- launch-to-launch noise of these timings is about 1 %
    hyst = float(os.environ.get("Y5_TUNE_HYST", "0.02"))
    timed.sort()
    finals = sorted(cfg for t, cfg in timed[:4] if t <= timed[0][0] * 1.10)
    for cfg in finals:
        d.cfg = cfg
        if lib.y5_conv2d_time(C.byref(d), *ptrs, 4 * iters, st, C.byref(ms)) != 0:
            continue
        t = ms.value
        if t < best_ms * (1.0 - hyst):
            second, second_ms = best, best_ms
            best, best_ms = cfg, t
        elif t < second_ms:
            second, second_ms = cfg, t
    if best < 0:
        best = timed[0][1]
    if second < 0 and len(timed) > 1:
        second = next(cfg for t, cfg in timed if cfg != best)
    if second < 0:
        second = best  # a single applicable configuration: the runner-up plan keeps it
    _TUNE_CACHE[key] = (best, second)
    _LAST_RACE[:] = [key, (best, second)]
    _save_tune_cache()
    return second if rank else best


_TUNE_FILE_STATE = {"loaded": False}


def _tune_cache_path():
    """Where the per-layer tile choices persist: $Y5_TUNE_CACHE (a json path; "0" / "off" disables), default
    ~/.cache/yolov5_amd/tune_<library version>_<kernel configurations>.json -- later processes (plan build in production, rectangular
    validation shapes seen before, clean rocprof traces) skip the timing launches."""
    path = os.environ.get("Y5_TUNE_CACHE")
    if path is not None:
        return None if path in ("", "0", "off") else path
    try:
        lib = _lib.lib()
        tag = f"{lib.y5_version()}_{lib.y5_conv_num_cfgs()}"
    except Exception:
        return None
    return os.path.join(os.path.expanduser("~"), ".cache", "yolov5_amd", f"tune_{tag}.json")


def _lib_stamp():
    """sha256 (16 hex digits) of the kernel library the choices were timed with: a cache written by another build of the kernels is ignored (a tile
    choice is a property of the binary -- this round's bias-in-LDS episode changed which configuration wins three layers without changing any id)."""
    if "stamp" not in _TUNE_FILE_STATE:
        import hashlib

        try:
            with open(_lib.LIB_PATH, "rb") as f:
                _TUNE_FILE_STATE["stamp"] = hashlib.sha256(f.read()).hexdigest()[:16]
        except OSError:
            _TUNE_FILE_STATE["stamp"] = "unknown"
    return _TUNE_FILE_STATE["stamp"]


TUNE_DB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tune_db.json")
TUNE_STATS = {"db_entries": 0, "races": 0}   # entries taken from the shipped database / races run by this process (bench.py reports both)


def _read_tune_file(path):
    """{key tuple: (best, runner-up)} of a tile-choice file written for THIS build of the kernel library, else {}."""
    try:
        import json

        with open(path) as f:
            d = json.load(f)
        if d.pop("__lib_sha16__", None) != _lib_stamp():
            return {}
        out = {}
        for k, v in d.items():
            v = (int(v), -1) if not isinstance(v, (list, tuple)) else (int(v[0]), int(v[1]))  # (best, runner-up); older files: best only
            out[tuple(int(x) if x not in ("True", "False") else x == "True" for x in k.split(","))] = v
        return out
    except (OSError, ValueError):
        return {}


def _load_tune_cache():
    """Tile choices known before any race: the user's cache file (_tune_cache_path) and -- only when no cache was named explicitly with Y5_TUNE_CACHE --
    the database shipped beside the library (yolov5_amd/tune_db.json: the races and in-situ decisions of the benchmarked plans, written on an MI355X by
    scripts/r6_final.sh with the committed kernels).  Like the cache it is bound to the library's hash: after any kernel change it is ignored and every
    choice is raced again on the spot.  With it, a bench run executes exactly the plan whose counter files are committed under profiles/pmc/.
    Y5_TUNE_DB=0 ignores it."""
    if _TUNE_FILE_STATE["loaded"]:
        return
    path = _tune_cache_path()
    if os.environ.get("Y5_TUNE_CACHE") is None and os.environ.get("Y5_TUNE_DB", "1") != "0":
        _TUNE_FILE_STATE["loaded"] = True
        db = _read_tune_file(TUNE_DB_PATH)
        TUNE_STATS["db_entries"] = len(db)
        _TUNE_CACHE.update(db)
    if not path:
        return
    _TUNE_FILE_STATE["loaded"] = True
    _TUNE_CACHE.update(_read_tune_file(path))


def _save_tune_cache():
    path = _tune_cache_path()
    if not path:
        return
    try:
        import json

        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "w") as f:
            json.dump({"__lib_sha16__": _lib_stamp(), **{",".join(str(x) for x in k): list(v) for k, v in _TUNE_CACHE.items()}}, f)
        os.replace(tmp, path)  # atomic: several ranks tune at once
    except OSError:
        pass


def default_backend(device):
    """The GPU backend -- or, under the test seam of _lib.use_test_library, the host-emulation backend that goes with it."""
    if _lib._test_backend is not None and torch.device(device).type == "cpu":
        return _lib._test_backend()
    return _HipBackend(device)


_SK_WS = {}  # device -> workspace tensor registered with the library (kept alive for the life of the process)


def _ensure_sk_workspace(be, lib, st):
    """Register the stream-K scratch (64 MiB of the 288 GB) for this device once; every plan of the process shares it."""
    dev = str(getattr(be, "device", "cpu"))
    if dev in _SK_WS or not _lib.experimental("streamk"):
        return
    nbytes = int(lib.y5_conv_sk_workspace_bytes())
    ws = be.empty((nbytes + 256,), torch.uint8)
    off = (-be.ptr(ws)) % 256
    _lib.check(lib.y5_conv_set_sk_workspace(C.c_void_p(be.ptr(ws) + off), nbytes, st), lib)
    _SK_WS[dev] = ws


class _HipBackend:
    """Device memory + stream provider of the Engine: PyTorch-ROCm caching allocator and current HIP stream.
    (The Engine takes it as a parameter so that tests can drive the very same plan-materialisation code against
    the host-compiled kernels of tests/hipemu; the product only ever constructs this GPU backend.)"""

    direct = True  # buffers ARE torch tensors: parameters / outputs are used in place through data_ptr()

    # time every workgroup-tile configuration per conv layer at plan build and keep the fastest (Y5_AUTOTUNE=0: heuristic tiles)
    autotune = os.environ.get("Y5_AUTOTUNE", "1") != "0"

    def __init__(self, device):
        if not torch.cuda.is_available() or torch.device(device).type != "cuda":
            raise RuntimeError("yolov5_amd.Engine needs a ROCm GPU (MI355X); there is no CPU execution path")
        self.lib = _lib.lib()
        self.device = torch.device(device)
        if os.environ.get("Y5_AUTOTUNE", "1") == "0":
            self.autotune = False

    def empty(self, shape, dtype):
        # outputs handed to the caller stay ORDINARY tensors under torch.inference_mode() (detect.py / val.py run inside
        # smart_inference_mode): an inference tensor has no version counter, and the objectness-hint tag relies on it
        with torch.inference_mode(False):
            return torch.empty(shape, dtype=dtype, device=self.device)

    def from_torch(self, t):
        return t.to(self.device).contiguous()

    def ptr(self, h):
        return h.data_ptr()

    def to_torch(self, h):
        return h

    def view_torch(self, h):
        """torch tensor sharing the buffer's memory (device buffers ARE torch tensors here)."""
        return h

    def zero_(self, h):
        h.zero_()

    def assign(self, h, t):
        h.copy_(t)

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def input(self, x):
        if not x.is_cuda:
            raise RuntimeError("yolov5_amd: input tensor must live on the GPU (no CPU path)")
        code = {torch.float16: _lib.Y5_F16, torch.float32: _lib.Y5_F32, torch.uint8: _lib.Y5_U8}.get(x.dtype)
        if code is None:
            raise TypeError(f"unsupported input dtype {x.dtype}")
        x = x.contiguous()
        return x, x.data_ptr(), code


class Engine:
    """A materialised PlanSpec on one GPU.  `engine(x)` -> dict of output tensors (engine-owned buffers, valid
    until the next call).  Raises RuntimeError without a GPU / without libyolov5_hip.so."""

    def __init__(self, model, x_shape, dtype: torch.dtype, device, want_raw=False, backend=None, spec=None, outputs=None, raw_views=None):
        if dtype not in (torch.float16, torch.float32):
            raise TypeError(f"Engine dtype must be float16 or float32, got {dtype}")
        self.be = backend if backend is not None else default_backend(device)
        self.lib = self.be.lib
        self.dtype = dtype
        self.dt = _lib.Y5_F16 if dtype == torch.float16 else _lib.Y5_F32
        self.es = 2 if dtype == torch.float16 else 4
        B, ch, H, W = x_shape
        self.x_shape = tuple(x_shape)
        self.spec = spec if spec is not None else build_plan_spec(
            model, B, ch, H, W, want_raw, fuse_bneck=dtype == torch.float16 and os.environ.get("Y5_FUSED_BNECK", "1") != "0",
            virtual_up=dtype == torch.float16 and not _lib.disabled("virtual_up"))
        det = getattr(model, "model", [None])[-1] if model is not None else None
        self._det = det if det is not None and hasattr(det, "anchors") else None
        self._anchor_ops = []  # (plan op index, pyramid level) of every op that holds anchor sizes
        if self._det is not None:
            self.stride_t = [float(s) for s in det.stride]
            self.anchors = det.anchors.detach().float().cpu().clone()
        self._keep = []  # device tensors referenced by raw pointers inside the C plan
        self._conv_bufs = []  # (op, packed filter, bias, stem filter, stem bias) of every conv op -> refresh_weights()
        self.bufs = [self.be.empty((B, b.H, b.W, b.C), dtype) for b in self.spec.bufs]
        # The raw (bs, na, ny, nx, no) head tensors of eval mode (models/yolo.py:96-98) are the Detect convs' NHWC outputs seen
        # through another index order: on the GPU they are returned as strided VIEWS of those plan buffers instead of being
        # written a second time by the decode kernel (274 MB per 64 images at 640^2).  Same shape and values; not contiguous;
        # valid until the next forward like every other output.
        self.raw_views = (getattr(self.be, "direct", False) and outputs is None and not _lib.disabled("raw_view")) \
            if raw_views is None else raw_views
        self.outputs = {}
        for name, o in self.spec.outputs.items():
            if self.raw_views and name.startswith("raw"):
                continue  # filled in by the decode op below
            if outputs is not None:  # caller-owned output buffers (SplitEngine: batch slices of one tensor)
                t = outputs[name]
                if tuple(t.shape) != tuple(o["shape"]) or t.dtype != dtype or not t.is_contiguous():
                    raise ValueError(f"engine output {name}: expected contiguous {tuple(o['shape'])} {dtype}")
                self.outputs[name] = t
            else:
                self.outputs[name] = self.be.empty(o["shape"], dtype)
        # Objectness plane beside z (one value per prediction row, written by the Detect decode / fused head): the NMS filter reads it instead
        # of the 85-value rows and fetches only the rows it cannot exclude (general.non_max_suppression picks it up from the z tensor it is
        # attached to).  Engine-owned outputs on the device only; Y5_DISABLE=obj_hint switches it off.
        self._hint = ("z" in self.outputs and outputs is None and getattr(self.be, "direct", False) and not _lib.disabled("obj_hint"))
        if self._hint:
            zs = self.spec.outputs["z"]["shape"]
            self._hint_shape = (zs[0], zs[1])  # (neither a spec output nor a key of self.outputs: it travels on the z tensor)
            self._hint_t = self.be.empty(self._hint_shape, dtype)
        self.plan = C.c_void_p(self.lib.y5_plan_create())
        self.op_names = []
        self.conv_cfgs = []
        self._insitu = []      # plain convolution ops with a runner-up configuration: candidates of _refine_in_situ (first forward)
        self._first_op = None
        self._stem = None      # plan index of the fused NCHW stem op (conv_stem.h), appended after the regular ops
        self._stem_args = None
        self._front = None     # plan index of the fused backbone front (conv_front.h: stem + 1.Conv + 2.C3.cv1+cv2), appended after the stem op
        self._front_args = None
        with torch.no_grad():
            self._fused_heads = set()
            for k, op in enumerate(self.spec.ops):
                self._cur = k
                self._add(op)
                if op.get("side"):
                    _lib.check(self.lib.y5_plan_set_branch(self.plan, self.lib.y5_plan_size(self.plan) - 1, 1), self.lib)
            if self._stem_args is not None:
                wp, bp, c2, npad, y = self._stem_args
                self._stem = self.lib.y5_plan_size(self.plan)
                _lib.check(self.lib.y5_plan_add_conv_stem(self.plan, None, B, H, W, C.c_void_p(self.be.ptr(wp)), C.c_void_p(self.be.ptr(bp)),
                                                          c2, npad, self._ptr(y), self._ld(y)), self.lib)
                self.op_names.append(self.op_names[1] + "[stem,nchw]")
                self._add_front(B, H, W)
        self._graph = False
        self._use_graph = os.environ.get("Y5_GRAPH", "1") == "1" and isinstance(self.be, _HipBackend)
        # Fresh z / proto (/ raw copies) per call, as the reference returns new tensors from every forward (models/yolo.py:115):
        # the plan's output pointers are re-pointed at newly allocated tensors (no copy) and captured graphs are cached per
        # binding -- in a steady loop the caching allocator hands the same blocks back and the same graph replays.  The strided
        # raw VIEWS of eval mode stay views of plan buffers (valid until the next forward; see DetectionModel.forward).
        self.fresh_outputs = outputs is None and not _lib.disabled("fresh_outputs")
        self._bound = {k: self.be.ptr(v) for k, v in self.outputs.items() if not (self.raw_views and k.startswith("raw"))}
        # The objectness plane is NOT re-allocated per call: it is an engine-owned side channel guarded by a forward counter (the tag on z names the
        # forward it belongs to; general.non_max_suppression drops a hint whose forward is no longer the engine's latest).  Re-pointing it per call
        # made the binding key (z, plane) wander through more pointer pairs than the graph cache holds whenever several generations of NMS buffers
        # were alive (DetectPipeline): a graph re-capture (~2.8 ms) on every forward.
        # Two planes, used alternately: the NMS of forward i may still be reading its plane on another stream while forward i+1 writes the other
        # (DetectPipeline); bindings stay periodic (plane k x a handful of z blocks), well inside the graph cache.
        self._hint_state = [0, 0]
        self._hint_k = 0
        self._hint_seq = 0
        if self._hint:
            self._hint_ring = [self._hint_t, self.be.empty(self._hint_shape, dtype)]
            self._bound["obj_hint"] = self.be.ptr(self._hint_t)

    def __del__(self):
        try:
            if getattr(self, "plan", None):
                self.lib.y5_plan_destroy(self.plan)
                self.plan = None
        except Exception:
            pass

    # -- helpers ---------------------------------------------------------------------------------------------
    def _ptr(self, t: TRef | None):
        if t is None:
            return None
        return C.c_void_p(self.be.ptr(self.bufs[t.buf]) + t.c_off * self.es)

    def _ld(self, t: TRef):
        return self.spec.bufs[t.buf].C

    def _add(self, op):
        lib, B, kind = self.lib, self.spec.B, op["op"]
        if kind == "to_nhwc":
            d = op["dst"]
            self._first_op = (d, op["C"])
            rc = lib.y5_plan_add_nchw_to_nhwc(self.plan, None, self.dt, self._ptr(d), self.dt, B, op["C"], d.H, d.W, self._ld(d), 1.0)
            self.op_names.append("to_nhwc")
        elif kind == "conv" and getattr(self, "_k3pw_skip", -1) == self._cur:
            rc = lib.y5_plan_add_nop(self.plan)  # multiplied inside the 3x3 in front of it (y5_plan_add_conv_k3pw)
            self.conv_cfgs.append(-2)
            self.op_names.append("conv:" + op["name"] + "(fused)")
        elif kind == "conv":
            rc = self._add_conv(op)
        elif kind == "bneck":
            rc = self._add_bneck(op)
        elif kind == "sppf_front":
            rc = self._add_sppf_front(op)
        elif kind == "sppf_pool":
            b = op["buf"]
            rc = lib.y5_plan_add_sppf_pool(self.plan, self._ptr(b), self.dt, B, b.H, b.W, op["C"], self._ld(b), op["k"])
            self.op_names.append("sppf_pool")
        elif kind == "upsample":
            s, d = op["src"], op["dst"]
            rc = lib.y5_plan_add_upsample2x(self.plan, self._ptr(s), self.dt, self._ptr(d), B, s.H, s.W, s.C, self._ld(s), self._ld(d))
            self.op_names.append("upsample2x")
        elif kind == "copy":
            s, d = op["src"], op["dst"]
            rc = lib.y5_plan_add_copy_slice(self.plan, self._ptr(s), self.dt, self._ptr(d), B * s.H * s.W, s.C, self._ld(s), self._ld(d))
            self.op_names.append("copy_slice")
        elif kind == "decode" and op["level"] in self._fused_heads:
            rc = lib.y5_plan_add_nop(self.plan)  # decoded inside the level's convolution (y5_plan_add_detect_head)
            self.op_names.append(f"decode{op['level']}(fused)")
        elif kind == "decode":
            x, i = op["x"], op["level"]
            apx = (self.anchors[i] * self.stride_t[i]).reshape(-1).tolist()
            arr = (C.c_float * len(apx))(*apx)
            raw = None
            if op["raw"] and self.raw_views:
                t = self.be.view_torch(self.bufs[x.buf])  # (B, ny, nx, ld) logits, channel a*no + o
                self.outputs[op["raw"]] = t[..., x.c_off:x.c_off + op["na"] * op["no"]].unflatten(-1, (op["na"], op["no"])).permute(0, 3, 1, 2, 4)
            elif op["raw"]:
                raw = self.outputs[op["raw"]]
            self._anchor_ops.append((lib.y5_plan_size(self.plan), i))
            rc = lib.y5_plan_add_detect_decode(self.plan, self._ptr(x), self.dt, B, op["ny"], op["nx"], op["na"], op["no"], op["nm"],
                                               self._ld(x), self.stride_t[i], arr, C.c_void_p(self.be.ptr(self.outputs["z"])), self.dt,
                                               op["nrows"], op["row_off"], C.c_void_p(self.be.ptr(raw)) if raw is not None else None)
            if rc == 0 and self._hint:
                rc = lib.y5_plan_set_obj_hint(self.plan, lib.y5_plan_size(self.plan) - 1, C.c_void_p(self.be.ptr(self._hint_t)))
            self.op_names.append(f"decode{i}")
        elif kind == "to_nchw":
            s = op["src"]
            o = self.outputs[op["out"]]
            rc = lib.y5_plan_add_nhwc_to_nchw(self.plan, self._ptr(s), self.dt, C.c_void_p(self.be.ptr(o)), B, s.C, s.H, s.W, self._ld(s))
            self.op_names.append("to_nchw")
        else:
            raise ValueError(kind)
        _lib.check(rc, lib)

    def _conv_weights(self, op):
        """Packed filter / bias of one conv op from the LIVE module parameters (BN folded with the current running statistics):
        (view geometry, main pack, stem pack or None).  Called at plan build and again by refresh_weights()."""
        x, res, y2 = op["x"], op["res"], op["y2"]
        w, b = folded_weights(op["mods"])
        (kh, kw), (sh, sw), (ph, pw) = op["k"], op["s"], op["p"]
        H, W, C1, ldx = x.H, x.W, x.C, self._ld(x)
        stem = None
        if op["view"] == "first":
            cin = w.shape[1]
            if (self.dtype == torch.float16 and cin == 3 and (kh, kw, sh, sw, ph, pw) == (6, 6, 2, 2, 2, 2) and op["act"] and res is None
                    and y2 is None and W % 64 == 0 and H % 2 == 0 and w.shape[0] <= 64 and w.shape[0] % 8 == 0
                    and not _lib.disabled("stem")):
                stem = pack_stem_weight(w, b) + (int(w.shape[0]),)
            wfull = torch.zeros((w.shape[0], C1, kh, kw), device=w.device)
            wfull[:, :cin] = w
            w = wfull
            if self.dtype == torch.float16 and C1 == 4 and kw % 2 == 0 and sw % 2 == 0 and pw % 2 == 0 and W % 2 == 0:
                # k6 s2 p2 stem: NHWC4 pixels pair up into 16-byte pieces -> conv over (H, W/2, 8), kernel (kh, kw/2)
                W, C1, ldx, kw, sw, pw = W // 2, 8, 8, kw // 2, sw // 2, pw // 2
        else:
            assert w.shape[1] == C1, (op["name"], w.shape, C1)
        main = pack_conv_weight(w, b, self.dtype)
        return (H, W, C1, ldx, kh, kw, sh, sw, ph, pw), main, stem

    def refresh_weights(self):
        """Re-pack every filter / bias from the live parameters into the plan's EXISTING device buffers (same shapes, same
        pointers: captured graphs stay valid).  Needed whenever weights or BatchNorm running statistics changed behind torch's
        version counters -- the fused optimizer, the fused EMA and the train-mode BatchNorm kernel write through raw pointers
        (yolo.BaseModel tracks that with `_state.weights_epoch`)."""
        with torch.no_grad():
            for op, wp, bp, swp, sbp in self._conv_bufs:
                _, main, stem = self._conv_weights(op)
                self.be.assign(wp, main[0])
                self.be.assign(bp, main[1])
                if swp is not None:
                    self.be.assign(swp, stem[0])
                    self.be.assign(sbp, stem[1])
            new_anchors = self._det.anchors.detach().float().cpu().clone() if self._det is not None else None
            if new_anchors is not None and self._anchor_ops and not torch.equal(new_anchors, self.anchors):
                self.anchors = new_anchors
                for idx, lvl in self._anchor_ops:
                    apx = (self.anchors[lvl] * self.stride_t[lvl]).reshape(-1).tolist()
                    _lib.check(self.lib.y5_plan_set_anchors(self.plan, idx, (C.c_float * len(apx))(*apx), len(apx)), self.lib)
                # graph kernel nodes hold their arguments by value: graphs captured so far carry the old anchors
                self._graph = False
                self._graph_gen = getattr(self, "_graph_gen", 0) + 1

    def _add_sppf_front(self, op):
        """SPPF.cv1 + the three max pools as one launch (y5_sppf_cv1_pool_fwd); the filter is packed like any conv's and re-packed by refresh_weights()."""
        sub, x, cat = op["cv1"], op["x"], op["buf"]
        _, (wp, bp, _K, Kpad, _Npad), _ = self._conv_weights(sub)
        wp, bp = self.be.from_torch(wp), self.be.from_torch(bp)
        self._keep += [wp, bp]
        self._conv_bufs.append((sub, wp, bp, None, None))
        self.op_names.append("sppf_front:" + op["name"])
        return self.lib.y5_plan_add_sppf_cv1_pool(self.plan, self._ptr(x), self._ld(x), C.c_void_p(self.be.ptr(wp)), C.c_void_p(self.be.ptr(bp)), Kpad,
                                                  self._ptr(cat), self._ld(cat), self.spec.B, x.H, x.W, x.C, op["C"], op["k"], 1 if sub["act"] else 0)

    def _add_bneck(self, op):
        """Fused Bottleneck (y5_bottleneck_fwd): both filters packed like ordinary convs; refresh_weights() re-packs them too.  When it is the last
        Bottleneck of a C3 and the op behind it is that C3's cv3, both can run as ONE launch (y5_bottleneck_cv3_fwd): the Bottleneck's result
        stays in LDS.  Y5_FUSED_CV3 = 0: never, 1: whenever the shapes fit, auto (default): both forms are timed at plan build."""
        x, y = op["x"], op["y"]
        packs = []
        for sub in (op["cv1"], op["cv2"]):
            _, (wp, bp, _K, Kpad, _Npad), _ = self._conv_weights(sub)
            wp, bp = self.be.from_torch(wp), self.be.from_torch(bp)
            self._keep += [wp, bp]
            self._conv_bufs.append((sub, wp, bp, None, None))
            packs.append((wp, bp, Kpad))
        (w1, b1, k1), (w2, b2, k2) = packs
        base = (self._ptr(x), self._ld(x), C.c_void_p(self.be.ptr(w1)), C.c_void_p(self.be.ptr(b1)), k1, C.c_void_p(self.be.ptr(w2)), C.c_void_p(self.be.ptr(b2)), k2)
        tail = (self.spec.B, x.H, x.W, x.C, int(op["add"]))
        cv3 = self._fused_cv3_args(op, base, tail)
        if cv3 is not None:
            self._k3pw_skip = self._cur + 1
            self.op_names.append(("bneck128+cv3:" if x.C == 128 else "bneck+cv3:") + op["name"] + "+" + cv3["name"])
            return self.lib.y5_plan_add_bottleneck_cv3(self.plan, *base, *cv3["args"], *tail)
        self.op_names.append(("bneck128:" if x.C == 128 else "bneck:") + op["name"])
        return self.lib.y5_plan_add_bottleneck(self.plan, *base, self._ptr(y), self._ld(y), *tail)

    def _fused_cv3_args(self, op, base, tail):
        mode = os.environ.get("Y5_FUSED_CV3", "auto")
        nxt_i = self._cur + 1
        x, y = op["x"], op["y"]
        if x.C == 128:   # c_ = 128 (conv_h3b.h, CV3 form): its own switch, same meaning.  Default OFF: measured slower than the two-launch form on every box
            # (97-100 us against 54 + 36, profiles/r05/r05_ab_cv3_128.log), and under a profiler's serialised launches the plan-build race picked
            # it anyway -- a PMC pass must measure the plan the bench runs
            mode = "1" if _lib.experimental("cv3_128") else "0"
        if mode == "0" or self.dt != _lib.Y5_F16 or x.C not in (32, 128) or nxt_i >= len(self.spec.ops):
            return None
        nxt = self.spec.ops[nxt_i]
        if not (nxt["op"] == "conv" and _pair(nxt["k"]) == (1, 1) and _pair(nxt["s"]) == (1, 1) and _pair(nxt["p"]) == (0, 0) and nxt["res"] is None
                and nxt["y2"] is None and not nxt.get("split_n") and not nxt.get("side") and nxt["c2_store"] <= 2 * x.C and nxt["c2_store"] % 8 == 0):
            return None
        cat = nxt["x"]
        if not (cat.buf == y.buf and cat.c_off == y.c_off and cat.C == 2 * y.C and y.C == x.C):  # this Bottleneck writes the first half of cv3's input
            return None
        for k, o in enumerate(self.spec.ops):  # the Bottleneck's own output must have no other reader
            if k in (self._cur, nxt_i):
                continue
            for v in o.values():
                if isinstance(v, TRef) and v.buf == y.buf and v.c_off < y.c_off + y.C and y.c_off < v.c_off + v.C and k > self._cur:
                    return None
        if mode != "1" and not getattr(self.be, "autotune", False):
            return None
        (_g, (wp3, bp3, _K3, Kpad3, Npad3), stem3) = self._conv_weights(nxt)
        if stem3 is not None or Npad3 > 2 * x.C:
            return None
        wp3, bp3 = self.be.from_torch(wp3), self.be.from_torch(bp3)
        y2 = _slice(cat, y.C, y.C)
        out = nxt["y"]
        args = (self._ptr(y2), self._ld(y2), C.c_void_p(self.be.ptr(wp3)), C.c_void_p(self.be.ptr(bp3)), Kpad3, nxt["c2_store"], 1 if nxt["act"] else 0,
                self._ptr(out), self._ld(out))
        lib, st = self.lib, self._stream()
        fused = C.c_void_p(lib.y5_plan_create())
        try:
            if lib.y5_plan_add_bottleneck_cv3(fused, *base, *args, *tail) != 0:
                return None
            ms_f = C.c_float(0)
            if lib.y5_plan_time_range(fused, 0, 1, 1 if mode == "1" else 10, st, C.byref(ms_f)) != 0:
                return None
            if mode != "1":
                (H3, W3, C13, ldx3, *_r) = _g
                d3 = _lib.ConvDesc(dtype=self.dt, B=self.spec.B, H=H3, W=W3, C1=C13, ldx=ldx3, OH=out.H, OW=out.W, C2=nxt["c2_store"], ldy=self._ld(out), KH=1, KW=1,
                                   SH=1, SW=1, PH=0, PW=0, act=1 if nxt["act"] else 0, Kpad=Kpad3, Npad=Npad3, ldr=0, ld2=0, cfg=-1, max_blocks=0, split_n=0)
                ptrs3 = (self._ptr(cat), args[2], args[3], None, self._ptr(out), None)
                d3.cfg = self._autotune_conv(d3, ptrs3, exclude=SK_CFGS)
                two = C.c_void_p(lib.y5_plan_create())
                try:
                    _lib.check(lib.y5_plan_add_bottleneck(two, *base, self._ptr(y), self._ld(y), *tail), lib)
                    _lib.check(lib.y5_plan_add_conv(two, C.byref(d3), *ptrs3), lib)
                    ms_t = C.c_float(0)
                    _lib.check(lib.y5_plan_time_range(two, 0, 2, 10, st, C.byref(ms_t)), lib)
                finally:
                    lib.y5_plan_destroy(two)
                if not ms_f.value < ms_t.value:
                    return None
        finally:
            lib.y5_plan_destroy(fused)
        self._keep += [wp3, bp3]
        self._conv_bufs.append((nxt, wp3, bp3, None, None))
        return dict(args=args, name=nxt["name"])

    def _add_conv(self, op):
        x, y, res, y2 = op["x"], op["y"], op["res"], op["y2"]
        (H, W, C1, ldx, kh, kw, sh, sw, ph, pw), (wp, bp, K, Kpad, Npad), stem = self._conv_weights(op)
        swp = sbp = None
        if stem is not None:
            swp, sbp, snpad, sc2 = stem
            swp, sbp = self.be.from_torch(swp), self.be.from_torch(sbp)
            self._keep += [swp, sbp]
            self._stem_args = (swp, sbp, sc2, snpad, y)
        if self.dtype == torch.float16 and C1 % 8:
            raise NotImplementedError(f"conv {op['name']}: fp16 needs input channels in multiples of 8 (got {C1})")
        wp, bp = self.be.from_torch(wp), self.be.from_torch(bp)
        self._keep += [wp, bp]
        self._conv_bufs.append((op, wp, bp, swp, sbp))
        c2s = op["c2_store"]
        d = _lib.ConvDesc(dtype=self.dt, B=self.spec.B, H=H, W=W, C1=C1, ldx=ldx, OH=y.H, OW=y.W, C2=c2s, ldy=self._ld(y),
                          KH=kh, KW=kw, SH=sh, SW=sw, PH=ph, PW=pw, act=1 if op["act"] else 0, Kpad=Kpad, Npad=Npad,
                          ldr=self._ld(res) if res is not None else 0, ld2=self._ld(y2) if y2 is not None else 0, cfg=-1, max_blocks=0,
                          split_n=op.get("split_n", 0))
        if d.split_n:
            d.ldy = self._ld(y)
        assert Npad >= c2s
        if op.get("x_up") is not None:   # virtual Upsample + Concat: the low-resolution tensor rides in the residual slot (include/yolov5_hip.h, up_c)
            lo, c_up = op["x_up"], op["up_c"]
            assert res is None and lo.H * 2 == H and lo.W * 2 == W and lo.C >= c_up
            d.up_c, d.ld_up = c_up, self._ld(lo)
            res = lo
        ptrs = (self._ptr(x), C.c_void_p(self.be.ptr(wp)), C.c_void_p(self.be.ptr(bp)), self._ptr(res), self._ptr(y), self._ptr(y2))
        if getattr(self.be, "autotune", False):
            # stream-K kernels combine split tiles through ONE registered workspace: not for ops that run beside others (side stream)
            # (measured, scripts/streamk_bench.py: at yolov5s bs=64 sizes the slab round trip costs more than the tail it removes --
            # 60-119 us against 42-69 us for the plain tiles -- so they only enter the race when asked for: Y5_EXPERIMENTAL=streamk)
            d.cfg = self._autotune_conv(d, ptrs, exclude=SK_CFGS if (op.get("side") or not _lib.experimental("streamk")) else ())
            race = tuple(_LAST_RACE)   # (key, (winner, runner-up)) of THIS op's race: the fusion checks below may run further races
        else:
            race = (None, None)
        k3pw = self._fused_k3pw_args(op, d, ptrs)
        if k3pw is not None:
            self._k3pw_skip = self._cur + 1
            self.conv_cfgs.append(int(d.cfg))
            self.op_names.append("conv+pw:" + op["name"] + "+" + k3pw["name"])
            return self.lib.y5_plan_add_conv_k3pw(self.plan, C.byref(d), ptrs[0], ptrs[1], ptrs[2], *k3pw["args"])
        head = self._fused_head_args(op, d, ptrs)
        if head is not None:
            self._fused_heads.add(head["level"])
            self._anchor_ops.append((self.lib.y5_plan_size(self.plan), head["level"]))
            self.conv_cfgs.append(head["cfg"])
            self.op_names.append("conv+decode:" + op["name"])
            rc = self.lib.y5_plan_add_detect_head(self.plan, C.byref(d), ptrs[0], ptrs[1], ptrs[2], *head["args"])
            if rc == 0 and self._hint:
                rc = self.lib.y5_plan_set_obj_hint(self.plan, self.lib.y5_plan_size(self.plan) - 1, C.c_void_p(self.be.ptr(self._hint_t)))
            return rc
        if race[1] is not None and int(d.cfg) == race[1][0] and not op.get("side"):
            self._insitu.append(dict(idx=self.lib.y5_plan_size(self.plan), slot=len(self.conv_cfgs), key=race[0], best=race[1][0], second=race[1][1]))
        self.conv_cfgs.append(int(d.cfg))
        self.op_names.append("conv:" + op["name"])
        return self.lib.y5_plan_add_conv(self.plan, C.byref(d), self._ptr(x), C.c_void_p(self.be.ptr(wp)), C.c_void_p(self.be.ptr(bp)),
                                         self._ptr(res), self._ptr(y), self._ptr(y2))

    def _fused_k3pw_args(self, op, d, ptrs):
        """`Conv(32, 64, 3, 2)` + the pointwise convolution that is its only reader (yolov5s: 1.Conv -> 2.C3.cv1+cv2) as ONE launch
        (csrc/conv_k3.h PW2, y5_conv_k3pw_fwd): the 3x3's output never reaches HBM.  Y5_FUSED_K3PW = 0: never, 1: whenever the shapes fit,
        auto (default): both forms are timed on the real buffers at plan build and the faster one is kept."""
        mode = os.environ.get("Y5_FUSED_K3PW", "auto")
        nxt_i = self._cur + 1
        if mode == "0" or self.dt != _lib.Y5_F16 or nxt_i >= len(self.spec.ops) or op.get("side"):
            return None
        nxt = self.spec.ops[nxt_i]
        y = op["y"]
        if not (_pair(op["k"]) == (3, 3) and _pair(op["s"]) == (2, 2) and _pair(op["p"]) == (1, 1) and op["act"] and op["res"] is None and op["y2"] is None and not op.get("split_n")
                and d.C1 == 32 and d.C2 == 64 and d.Npad == 64 and y.H % 4 == 0 and y.W % 8 == 0):
            return None
        if not (nxt["op"] == "conv" and _pair(nxt["k"]) == (1, 1) and _pair(nxt["s"]) == (1, 1) and _pair(nxt["p"]) == (0, 0) and nxt["res"] is None and nxt["c2_store"] <= 64
                and nxt["c2_store"] % 8 == 0 and not nxt.get("side") and nxt.get("x_up") is None and nxt["x"].buf == y.buf and nxt["x"].c_off == y.c_off and nxt["x"].C == d.C2):
            return None
        if nxt["y2"] is not None and not nxt.get("split_n"):
            return None  # an upsampled replica behind the pointwise layer: not built
        for k, o in enumerate(self.spec.ops):  # the 3x3's output must have no other reader
            if k in (self._cur, nxt_i):
                continue
            for v in o.values():
                if isinstance(v, TRef) and v.buf == y.buf:
                    return None
        if mode != "1" and not getattr(self.be, "autotune", False):
            return None
        (_g, (wp2, bp2, _K2, Kpad2, Npad2), stem2) = self._conv_weights(nxt)
        if stem2 is not None or Npad2 != 64:
            return None
        wp2, bp2 = self.be.from_torch(wp2), self.be.from_torch(bp2)
        c3 = nxt["c2_store"]
        split = nxt.get("split_n") or c3
        y1, y2 = nxt["y"], nxt["y2"]
        args = (C.c_void_p(self.be.ptr(wp2)), C.c_void_p(self.be.ptr(bp2)), c3, Npad2, Kpad2, 1 if nxt["act"] else 0, self._ptr(y1), self._ld(y1),
                self._ptr(y2), self._ld(y2) if y2 is not None else 0, split)
        lib, st = self.lib, self._stream()
        d3 = _lib.ConvDesc.from_buffer_copy(d)
        best_cfg, ms_f = None, C.c_float(0)
        kkey = (_K3PW_MARK, int(d.dtype), int(d.B), int(d.H), int(d.W), int(d.ldx), int(c3))   # (the race between the two builds is kept like any other)
        _load_tune_cache()
        known = _TUNE_CACHE.get(kkey)
        for cand in ((known[0],) if known is not None and known[0] in (34, 81) else (34, 81)):  # four waves with two stages each / eight waves with one stage each
            d3.cfg = cand
            fused = C.c_void_p(lib.y5_plan_create())
            try:
                if lib.y5_plan_add_conv_k3pw(fused, C.byref(d3), ptrs[0], ptrs[1], ptrs[2], *args) != 0:
                    continue
                ms = C.c_float(0)
                if lib.y5_plan_time_range(fused, 0, 1, 1 if (mode == "1" and known is not None) else 10, st, C.byref(ms)) != 0:
                    continue
                if best_cfg is None or ms.value < ms_f.value:
                    best_cfg, ms_f = cand, C.c_float(ms.value)
            finally:
                lib.y5_plan_destroy(fused)
        if best_cfg is None:
            return None
        if known is None:
            _TUNE_CACHE[kkey] = (best_cfg, -1)
            _save_tune_cache()
        d3.cfg = best_cfg
        if mode != "1":
            (H2, W2, C12, ldx2, *_r) = _g
            d2 = _lib.ConvDesc(dtype=self.dt, B=self.spec.B, H=H2, W=W2, C1=C12, ldx=ldx2, OH=y1.H, OW=y1.W, C2=c3, ldy=self._ld(y1), KH=1, KW=1, SH=1, SW=1,
                               PH=0, PW=0, act=1 if nxt["act"] else 0, Kpad=Kpad2, Npad=Npad2, ldr=0, ld2=self._ld(y2) if y2 is not None else 0, cfg=-1,
                               max_blocks=0, split_n=nxt.get("split_n", 0))
            ptrs2 = (self._ptr(nxt["x"]), args[0], args[1], None, self._ptr(y1), self._ptr(y2))
            d2.cfg = self._autotune_conv(d2, ptrs2, exclude=SK_CFGS)
            two = C.c_void_p(lib.y5_plan_create())
            try:
                _lib.check(lib.y5_plan_add_conv(two, C.byref(d), *ptrs), lib)
                _lib.check(lib.y5_plan_add_conv(two, C.byref(d2), *ptrs2), lib)
                ms_t = C.c_float(0)
                _lib.check(lib.y5_plan_time_range(two, 0, 2, 10, st, C.byref(ms_t)), lib)
            finally:
                lib.y5_plan_destroy(two)
            if not ms_f.value < ms_t.value:
                return None
        d.cfg = d3.cfg
        self._keep += [wp2, bp2]
        self._conv_bufs.append((nxt, wp2, bp2, None, None))
        # what the fused backbone front (stem + this 3x3 + this 1x1 in one launch, _add_front) needs beside the stem's arguments
        self._front_args = dict(w1=ptrs[1], b1=ptrs[2], C1=int(d.C2), Npad1=int(d.Npad), Kpad1=int(d.Kpad), act1=int(d.act), tail=args,
                                name=op["name"] + "+" + nxt["name"], x_is_stem=op["x"], idx=self._cur)
        return dict(args=args, name=nxt["name"])

    def _add_front(self, B, H, W):
        """0.Conv (NCHW stem) + 1.Conv + the pointwise layer behind it as ONE launch (csrc/conv_front.h, y5_conv_front_fwd): available when the plan
        already runs the stem from NCHW and `1.Conv + 2.C3.cv1+cv2` as one launch (their shapes are the front kernel's).  Y5_FUSED_FRONT = 0: never,
        1: whenever the library accepts the shape, auto (default): timed against the two launches it replaces at plan build, faster form kept.
        The op is appended behind the stem op; __call__ runs it instead of plan ops 1..3 for fp16 NCHW batches."""
        mode = os.environ.get("Y5_FUSED_FRONT", "auto")
        fa = self._front_args
        if mode == "0" or fa is None or self._stem_args is None or self.dt != _lib.Y5_F16 or fa.get("idx") != 2:
            return
        if mode != "1" and not getattr(self.be, "autotune", False):
            return
        swp, sbp, sc2, snpad, sy = self._stem_args
        if fa["x_is_stem"].buf != sy.buf or sc2 != 32 or snpad != 32:
            return
        lib, st = self.lib, self._stream()
        w2, b2, c3, npad2, kpad2, act2, y, ldy, y2, ld2, split = fa["tail"]
        args = (None, B, H, W, C.c_void_p(self.be.ptr(swp)), C.c_void_p(self.be.ptr(sbp)), sc2, fa["w1"], fa["b1"], fa["C1"], fa["Npad1"], fa["Kpad1"], fa["act1"],
                w2, b2, c3, npad2, kpad2, act2, y, ldy, y2, ld2, split)
        if mode != "1":
            # race on the real buffers: a scratch input of the plan's shape stands in for the caller's batch
            xs = self.be.empty((B, 3, H, W), torch.float16)
            tmp = C.c_void_p(lib.y5_plan_create())
            try:
                if lib.y5_plan_add_conv_front(tmp, C.c_void_p(self.be.ptr(xs)), *args[1:]) != 0:
                    return
                ms_f, ms_s, ms_k = C.c_float(0), C.c_float(0), C.c_float(0)
                if lib.y5_plan_time_range(tmp, 0, 1, 10, st, C.byref(ms_f)) != 0:
                    return
                _lib.check(lib.y5_plan_set_input(self.plan, self._stem, C.c_void_p(self.be.ptr(xs))), lib)
                _lib.check(lib.y5_plan_time_range(self.plan, self._stem, self._stem + 1, 10, st, C.byref(ms_s)), lib)
                _lib.check(lib.y5_plan_time_range(self.plan, 2, 3, 10, st, C.byref(ms_k)), lib)
            finally:
                lib.y5_plan_destroy(tmp)
            if not ms_f.value < ms_s.value + ms_k.value:
                return
        if lib.y5_plan_add_conv_front(self.plan, *args) != 0:
            return  # shape outside the kernel's range: the two-launch form stays
        self._front = lib.y5_plan_size(self.plan) - 1
        self.op_names.append("front:0.Conv+" + fa["name"])

    def _fused_head_args(self, op, d, ptrs):
        """Detect convolution of one level + its decode as ONE launch (csrc/head.hip) when only `z` is wanted (export mode: no raw
        tensors) and the shape fits the kernel.  Y5_FUSED_HEAD = 0: never, 1: whenever the library accepts the shape, auto (default):
        both forms are timed on the real buffers at plan build, like the tile autotuner, and the faster one is kept."""
        mode = os.environ.get("Y5_FUSED_HEAD", "auto")
        if mode == "0" or not op["name"].startswith("detect.m") or self.dt != _lib.Y5_F16 or self._cur + 1 >= len(self.spec.ops):
            return None
        dec = self.spec.ops[self._cur + 1]
        if dec["op"] != "decode" or dec["raw"] or dec["na"] != 3 or dec["no"] != 85 or dec["nm"] or "z" not in self.outputs:
            return None
        if mode != "1" and not getattr(self.be, "autotune", False):
            return None
        if d.C1 > 128 and _lib.disabled("head_deep"):
            return None   # A/B switch: the K-streamed fused head of the deep levels (csrc/conv_headk.h) off
        lvl = dec["level"]
        apx = (self.anchors[lvl] * self.stride_t[lvl]).reshape(-1).tolist()
        arr = (C.c_float * 6)(*apx)
        zp = C.c_void_p(self.be.ptr(self.outputs["z"]))
        args = (dec["ny"], dec["nx"], self.stride_t[lvl], arr, zp, dec["nrows"], dec["row_off"])
        lib, st = self.lib, self._stream()
        # two builds of the fused kernel: four waves x two stages (cfg 56) and eight waves x one stage (cfg 87); timed, faster kept
        # (the choice between the two is kept in the tile-choice cache like any other race: timed once per shape, the same in every later plan)
        best, tuned = None, int(d.cfg)
        hkey = (_HEAD_MARK, int(d.dtype), int(d.B), int(d.H), int(d.W), int(d.C1), int(d.C2), int(d.ldx))
        _load_tune_cache()
        known = _TUNE_CACHE.get(hkey)
        for hc in ((known[0],) if known is not None and known[0] in (56, 87) else (56, 87)):
            d.cfg = hc
            one = C.c_void_p(lib.y5_plan_create())
            try:
                ms_h = C.c_float(0)
                if (lib.y5_plan_add_detect_head(one, C.byref(d), ptrs[0], ptrs[1], ptrs[2], *args) == 0
                        and lib.y5_plan_time_range(one, 0, 1, 1 if (mode == "1" and known is not None) else 10, st, C.byref(ms_h)) == 0
                        and (best is None or ms_h.value < best[0])):
                    best = (ms_h.value, hc)
            finally:
                lib.y5_plan_destroy(one)
        d.cfg = tuned
        if best is None:
            return None  # shape not supported by the fused kernel
        if known is None:
            _TUNE_CACHE[hkey] = (best[1], -1)
            _save_tune_cache()
        if mode != "1":
            two = C.c_void_p(lib.y5_plan_create())
            try:
                x = dec["x"]
                _lib.check(lib.y5_plan_add_conv(two, C.byref(d), *ptrs), lib)
                _lib.check(lib.y5_plan_add_detect_decode(two, self._ptr(x), self.dt, self.spec.B, dec["ny"], dec["nx"], 3, 85, 0, self._ld(x),
                                                         self.stride_t[lvl], arr, zp, self.dt, dec["nrows"], dec["row_off"], None), lib)
                ms_t = C.c_float(0)
                _lib.check(lib.y5_plan_time_range(two, 0, 2, 10, st, C.byref(ms_t)), lib)
            finally:
                lib.y5_plan_destroy(two)
            if not best[0] < ms_t.value:
                return None
        d.cfg = best[1]
        self._keep.append(arr)
        return dict(level=lvl, args=args, cfg=best[1])

    def _autotune_conv(self, d, ptrs, exclude=()):
        _ensure_sk_workspace(self.be, self.lib, self._stream())
        return autotune_conv(self.lib, d, ptrs, self._stream(), exclude)

    # -- execution ---------------------------------------------------------------------------------------------
    def _stream(self):
        return self.be.stream()

    def __call__(self, x, outputs=None):
        """outputs: optional caller-owned destination tensors {name: contiguous tensor of the plan's output shape} for this call."""
        if tuple(x.shape) != self.x_shape:
            raise ValueError(f"engine built for input {self.x_shape}, got {tuple(x.shape)}")
        x, xptr, src_dt = self.be.input(x)
        d, cin = self._first_op
        st = self._stream()
        B = self.spec.B
        n = self.lib.y5_plan_size(self.plan)
        if self.fresh_outputs or outputs is not None:
            self._rebind_fresh(n, outputs)
        if self._stem is not None and src_dt == _lib.Y5_F16:
            # fp16 NCHW batch: the stem conv reads it in place (no NHWC repack pass), then the plan continues at op 2 -- or, with the fused
            # front (stem + 1.Conv + 2.C3.cv1+cv2 in one launch), at op 4
            self._stem_active = True
            head = self._front if self._front is not None else self._stem
            body0 = 4 if self._front is not None else 2
            _lib.check(self.lib.y5_plan_set_input(self.plan, head, C.c_void_p(xptr)), self.lib)
            _lib.check(self.lib.y5_plan_run_range(self.plan, head, head + 1, st), self.lib)
            if self._insitu and not self._graph:
                self._refine_in_situ(body0, self._stem)
            if self._use_graph:
                # the body only touches plan-owned buffers and the bound outputs: replayed as ONE hipGraph launch (captured on first
                # use of every output binding)
                if not self._graph:
                    if self.lib.y5_plan_capture_range(self.plan, body0, self._stem, st) == 0:
                        self._graph = True
                    else:  # capture refused by the runtime: same kernels, launched one by one
                        self._use_graph = False
            if self._use_graph:
                _lib.check(self.lib.y5_plan_launch_graph(self.plan, st), self.lib)
            else:
                _lib.check(self.lib.y5_plan_run_range(self.plan, body0, self._stem, st), self.lib)
            return self._tag_hint()
        self._stem_active = False
        scale = 1.0 / 255.0 if src_dt == _lib.Y5_U8 else 1.0  # train.py:379 / detect.py:209: uint8 images -> 0..1
        _lib.check(self.lib.y5_nchw_to_nhwc(C.c_void_p(xptr), src_dt, self._ptr(d), self.dt, B, cin, d.H, d.W, self._ld(d),
                                            scale, st), self.lib)
        if self._insitu:
            self._refine_in_situ(1, n if self._stem is None else self._stem)
        _lib.check(self.lib.y5_plan_run_range(self.plan, 1, n if self._stem is None else self._stem, st), self.lib)
        return self._tag_hint()

    def _tag_hint(self):
        """Hang the objectness plane on the z tensor it was written with (general.non_max_suppression looks for it there and checks that z is
        still the tensor of this forward: same object, unchanged version counter)."""
        if self._hint:
            z = self.outputs["z"]
            self._hint_seq += 1
            self._hint_state[self._hint_k] = self._hint_seq
            # a caller-provided inference tensor cannot prove that it was not edited in place (no version counter): no hint for it,
            # the NMS filter then reads the rows themselves
            if not z.is_inference():
                z._y5_obj_hint = (self._hint_t, z._version, z.data_ptr(), self._hint_state, self._hint_k, self._hint_seq)
        return self.outputs

    def _rebind_fresh(self, n, outputs=None):
        """Point the plan at newly allocated (or caller-provided) output tensors and select the graph captured for that binding."""
        changed = False
        for name, old in list(self._bound.items()):
            if name == "obj_hint":  # the other plane of the ring
                self._hint_k ^= 1
                self._hint_t = self._hint_ring[self._hint_k]
                new = self.be.ptr(self._hint_t)
                _lib.check(self.lib.y5_plan_rebind_output(self.plan, 0, n, C.c_void_p(old), C.c_void_p(new)), self.lib)
                self._bound[name] = new
                changed = True
                continue
            if outputs is not None:
                t = outputs[name]
                if tuple(t.shape) != tuple(self.spec.outputs[name]["shape"]) or not t.is_contiguous():
                    raise ValueError(f"engine output {name}: expected contiguous {tuple(self.spec.outputs[name]['shape'])}")
            else:
                t = self.be.empty(self.spec.outputs[name]["shape"], self.dtype)
            new = self.be.ptr(t)
            self.outputs[name] = t
            if new != old:
                _lib.check(self.lib.y5_plan_rebind_output(self.plan, 0, n, C.c_void_p(old), C.c_void_p(new)), self.lib)
                self._bound[name] = new
                changed = True
        if changed or not self._graph:
            key = hash((getattr(self, "_graph_gen", 0),) + tuple(sorted(self._bound.items()))) & 0xFFFFFFFFFFFFFFFF
            self._graph = self.lib.y5_plan_select_graph(self.plan, key) == 1

    def _refine_in_situ(self, lo, hi):
        """The tuner races configurations on ISOLATED back-to-back launches (warm caches, no neighbours); inside the forward a launch starts behind another
        kernel's tail with cold caches, and configurations differ in how much that costs them -- a near tie flipped 21.Conv from id 96 (38.6 us in situ) to id 40
        (49.2 us in situ; 38.7 vs 39.3 isolated: profiles/r06/r06_ab_tapseq_runtime_walker.log).  So, once per plan and before its graph is captured, the
        runner-up of every plain convolution is timed IN its place: one in-situ profile of ops [lo, hi) with every winner, one with every runner-up, a
        runner-up replaces a winner when it is more than 3 % faster where it runs -- in that profile AND in a closing profile of the mix, which also has to be no
        slower as a whole.
        Decisions persist in the tile-choice cache (key + (_INSITU_MARK, plan index)): later processes -- clean rocprof traces, production -- apply them without
        timing.  Y5_DISABLE=insitu_tune switches it off; runner-up plans (Y5_TUNE_RANK=1) are left alone."""
        cands = [c for c in self._insitu if lo <= c["idx"] < hi and c["second"] >= 0 and c["second"] != c["best"]]
        self._insitu = []
        if not cands or _lib.disabled("insitu_tune") or int(os.environ.get("Y5_TUNE_RANK", "0")):
            return
        lib, st = self.lib, self._stream()

        def apply(c, cfg):
            _lib.check(lib.y5_plan_set_conv_cfg(self.plan, c["idx"], cfg), lib)
            self.conv_cfgs[c["slot"]] = cfg

        stored = [_TUNE_CACHE.get(c["key"] + (_INSITU_MARK, c["idx"])) for c in cands]
        self.insitu_timed = False
        if all(s is not None for s in stored):
            for c, s in zip(cands, stored):
                if s[0] != c["best"]:
                    apply(c, s[0])
            self.insitu_swaps = [(self.op_names[c["idx"]], c["best"], s[0]) for c, s in zip(cands, stored) if s[0] != c["best"]]
            return
        self.insitu_timed = True

        def profile():
            buf = (C.c_float * (hi - lo))()
            _lib.check(lib.y5_plan_profile_range(self.plan, lo, hi, 9, st, buf), lib)
            return [float(v) for v in buf]

        t_best = profile()
        for c in cands:
            apply(c, c["second"])
        try:
            t_second = profile()
        except RuntimeError:   # a runner-up the plan cannot launch in place: the winners stay
            for c in cands:
                apply(c, c["best"])
            return
        swapped = []
        for c in cands:
            k = c["idx"] - lo
            if t_second[k] < 0.97 * t_best[k]:
                swapped.append(c)
            else:
                apply(c, c["best"])
        if swapped:
            # second opinion: the mix profiled as a whole -- a swap stays only if it is faster in THIS measurement too (two independent medians: launch noise
            # alone passes the 3 % bar once in a while, not twice), and nothing stays if the range as a whole got slower
            t_mix = profile()
            whole = sum(t_mix) <= sum(t_best)
            for c in list(swapped):
                if not whole or t_mix[c["idx"] - lo] >= 0.97 * t_best[c["idx"] - lo]:
                    apply(c, c["best"])
                    swapped.remove(c)
        for c in cands:
            _TUNE_CACHE[c["key"] + (_INSITU_MARK, c["idx"])] = (c["second"] if c in swapped else c["best"], -1)
        self.insitu_swaps = [(self.op_names[c["idx"]], c["best"], c["second"]) for c in swapped]
        _save_tune_cache()

    def plan_table(self):
        """[(op name, tile configuration id | "stem" | "bneck" | None)] in launch order: what the autotuner / the fusion races chose for this
        plan (logged by bench.py --op-table and tests/test_gpu_plans.py so that every measured number names the plan it belongs to)."""
        ci = iter(self.conv_cfgs)
        out = []
        for n in self.op_names:
            if n.endswith("[stem,nchw]"):
                out.append((n, "stem"))
            elif n.startswith("front:"):
                out.append((n, "front"))
            elif n.startswith(("conv:", "conv+pw:", "conv+decode:")):
                out.append((n, next(ci, None)))
            elif n.startswith("sppf_front:"):
                out.append((n, "sppf"))     # csrc/conv_sppf.h
            elif n.startswith("bneck128"):
                out.append((n, "h3b"))      # csrc/conv_h3b.h (the 128-channel Bottleneck: GEMM-1 phase + halo-resident 3x3)
            elif n.startswith("bneck"):
                out.append((n, "bneck"))
            else:
                out.append((n, None))
        return out

    def profile_ops(self, iters=5):
        """In-situ per-op HIP-event timing: [(name, ms)] in execution order, every op timed in its real position of one eager
        forward (median of `iters` passes) -- what bench.py's roofline is computed from.  Call after at least one forward."""
        st = self._stream()
        n = self.lib.y5_plan_size(self.plan)
        res = []
        if self._stem is not None and getattr(self, "_stem_active", False):
            ranges = [(self._front, self._front + 1), (4, self._stem)] if self._front is not None else [(self._stem, self._stem + 1), (2, self._stem)]
        else:
            ranges = [(1, n if self._stem is None else self._stem)]
        for lo, hi in ranges:
            buf = (C.c_float * (hi - lo))()
            _lib.check(self.lib.y5_plan_profile_range(self.plan, lo, hi, iters, st, buf), self.lib)
            res += [(self.op_names[lo + k], float(buf[k])) for k in range(hi - lo)]
        return res

    def time_ops(self, iters=20):
        """Per-op HIP-event timing (ms per launch) on the current stream: [(name, ms)] -- used by bench.py."""
        st = self._stream()
        n = self.lib.y5_plan_size(self.plan)
        res = []
        ms = C.c_float(0)
        if self._stem is not None and getattr(self, "_stem_active", False):
            # as executed by __call__ for an fp16 batch
            order = [self._front] + list(range(4, self._stem)) if self._front is not None else [self._stem] + list(range(2, self._stem))
        else:
            order = list(range(1, n if self._stem is None else self._stem))
        for i in order:
            _lib.check(self.lib.y5_plan_time_range(self.plan, i, i + 1, iters, st, C.byref(ms)), self.lib)
            res.append((self.op_names[i], ms.value / iters))
        self.timed_order = order
        return res


_SPLIT_STREAMS = {}


class SplitEngine:
    """The batch cut into `parts` equal sub-batches, each with its own plan on its own HIP stream, writing batch slices of
    one set of output tensors.  Every layer is a persistent kernel whose prologue burst, last epilogue and tail round leave
    most CUs idle for 10-25 % of its duration; with two independent plans in flight the hardware fills
    those holes with the other plan's workgroups (yolov5s bs=64: 2.86 ms vs 3.04 ms).  Images are independent in eval mode
    (BatchNorm is folded), so the result is the same function of the input."""

    def __init__(self, model, x_shape, dtype, device, want_raw=False, parts=2):
        B = x_shape[0]
        if B % parts:
            raise ValueError("SplitEngine: batch not divisible")
        self.parts, self.x_shape, self.sub_b = parts, tuple(x_shape), B // parts
        sub_shape = (self.sub_b,) + tuple(x_shape[1:])
        spec = build_plan_spec(model, self.sub_b, x_shape[1], x_shape[2], x_shape[3], want_raw)
        self._out_shapes = {name: (B,) + tuple(o["shape"][1:]) for name, o in spec.outputs.items()}
        self.dtype = dtype
        self.outputs = {name: torch.empty(shp, dtype=dtype, device=device) for name, shp in self._out_shapes.items()}
        self.engines = []
        for i in range(parts):
            views = {name: t[i * self.sub_b:(i + 1) * self.sub_b] for name, t in self.outputs.items()}
            self.engines.append(Engine(model, sub_shape, dtype, device, want_raw=want_raw, spec=spec if i == 0 else None, outputs=views,
                                       raw_views=False))
        self.device = device
        # HIP multiplexes streams onto a few hardware queues and two streams on one queue run one after the other (measured:
        # 2.92 ms vs 3.55 ms for the same two plans depending on the pair the stream pool hands out, 3.27 ms once eight
        # streams exist) -- so one fixed set of side streams per device is created once and shared by every SplitEngine
        key = (str(device), parts)
        if key not in _SPLIT_STREAMS:
            _SPLIT_STREAMS[key] = [torch.cuda.Stream(device) for _ in range(parts)]
        self.streams = _SPLIT_STREAMS[key]

    def __call__(self, x):
        if tuple(x.shape) != self.x_shape:
            raise ValueError(f"engine built for input {self.x_shape}, got {tuple(x.shape)}")
        cur = torch.cuda.current_stream(self.device)
        if not _lib.disabled("fresh_outputs"):  # new result tensors per call (see Engine.fresh_outputs)
            self.outputs = {name: torch.empty(shp, dtype=self.dtype, device=self.device) for name, shp in self._out_shapes.items()}
        for i, (eng, s) in enumerate(zip(self.engines, self.streams)):
            s.wait_stream(cur)
            views = {name: t[i * self.sub_b:(i + 1) * self.sub_b] for name, t in self.outputs.items()}
            with torch.cuda.stream(s):
                eng(x[i * self.sub_b:(i + 1) * self.sub_b], outputs=views)
        for s in self.streams:
            cur.wait_stream(s)
        return self.outputs

    def refresh_weights(self):
        for eng in self.engines:
            eng.refresh_weights()

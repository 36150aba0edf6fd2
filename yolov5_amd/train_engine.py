"""Training engine: train-mode forward (conv -> batch-stat BN -> SiLU, models/common.py:82-88) and the full backward of
the YOLOv5 graph as HIP kernel launches, exposed to torch.autograd as ONE Function so that
`pred = model(imgs); loss, _ = compute_loss(pred, targets); scaler.scale(loss).backward()` (train.py:401-410) works
unchanged: autograd sees model parameters -> pred, the kernels do everything in between.

Per Conv block, forward:  z = conv(x) [y5_conv2d_fwd, no bias/act] -> y = [res +] silu(bn_batch(z)) [y5_bn_silu_fwd]
                backward: dz, dgamma, dbeta = y5_bn_silu_bwd(dy, z); dW = y5_conv2d_wgrad(x, dz); dx (+)= dgrad(dz, W)
Concat is free (producers write channel slices of the consumer's buffer; the gradient buffer mirrors that layout),
nn.Upsample / SPPF pools / Bottleneck shortcuts have their own small backward kernels (train_misc.hip).

Differences from the inference plan (engine.py): no in-place residuals and C3's cv1/cv2 are separate launches, because
backward needs every block input intact.  Compute dtype is fp16 with fp32 master weights (AMP semantics): filters are
re-packed on the device from the live fp32 parameters at every step (y5_pack_conv_weight / y5_pack_dgrad_weight).
Gradients can be streamed to a sink (HipDDP, torch_utils.py) the moment their kernels are queued.  No CPU path.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib, _state
from .engine import TRef, _Planner, _slice, autotune_conv, default_backend
from .packing import round_up
from .train_ops import _axis_classes


class _TrainPlanner(_Planner):
    """Same graph walk as the inference planner; C3 un-fused and without buffer aliasing."""

    def c3(self, m, x: TRef, dest, up2=None, name="C3"):
        c_ = m.cv1.conv.out_channels
        n = len(m.m)
        cat = self.spec.new_buf(x.H, x.W, 2 * c_, name + ".cat")
        a = self.conv([m.cv1], x, _slice(cat, 0, c_) if n == 0 else None, name=name + ".cv1")
        self.conv([m.cv2], x, _slice(cat, c_, c_), name=name + ".cv2")
        for i, b in enumerate(m.m):
            t = self.conv([b.cv1], a, None, name=f"{name}.m{i}.cv1")
            a = self.conv([b.cv2], t, _slice(cat, 0, c_) if i == n - 1 else None, res=a if b.add else None, name=f"{name}.m{i}.cv2")
        return self.conv([m.cv3], cat, dest, up2=up2, name=name + ".cv3")


NO_RESIDUAL_CFGS = frozenset(list(range(14, 22)) + [56] + list(range(84, 88)) + [93, 94])   # conv.hip: configurations without a residual / accumulate path


def _vp(v):
    return C.c_void_p(v)


class TrainEngine:
    """Materialised training plan for one (batch, resolution) on one GPU."""

    def __init__(self, model, x_shape, device, backend=None, dtype=torch.float16):
        """dtype: torch.float16 = AMP semantics (fp16 activations / gradients, fp32 master weights and accumulators: the production
        path); torch.float32 = the whole step in fp32 (exact-fp32 MFMA forward / data gradient, plain fp32 weight gradient): the
        reference-precision mode the end-to-end gradient check against the oracle's autograd runs in."""
        if dtype not in (torch.float16, torch.float32):
            raise TypeError("TrainEngine dtype must be float16 or float32")
        self.dtype, self.dt, self.es = dtype, (_lib.Y5_F16 if dtype == torch.float16 else _lib.Y5_F32), (2 if dtype == torch.float16 else 4)
        self.be = backend if backend is not None else default_backend(device)
        self.lib = self.be.lib
        self.model = model
        # Y5_DETERMINISTIC=1 (or torch.use_deterministic_algorithms(True), as the reference's init_seeds(deterministic=True) sets, utils/general.py):
        # weight gradients are reduced over their pixel-range splits in a fixed order instead of with fp32 atomics -- the one order-dependent
        # step of the training plan; a few % slower (one more small launch per layer)
        self.deterministic = os.environ.get("Y5_DETERMINISTIC", "0") == "1" or torch.are_deterministic_algorithms_enabled()
        self._wg_ws, self._wg_ws_bytes = None, 0
        B, ch, H, W = x_shape
        self.x_shape = tuple(x_shape)
        self.spec = _TrainPlanner(model, B, ch, H, W, want_raw=True).run()
        if any(o["op"] in ("copy", "upsample") for o in self.spec.ops):
            raise NotImplementedError("training plan: standalone Concat copy / Upsample ops are not supported")
        f16 = dtype  # (activation / gradient storage type of this plan)
        self.bufs = [self.be.empty((B, b.H, b.W, b.C), f16) for b in self.spec.bufs]
        self.gbufs = [self.be.empty((B, b.H, b.W, b.C), f16) for b in self.spec.bufs]
        self.params = list(model.parameters())
        self._pidx = {id(p): i for i, p in enumerate(self.params)}
        # flat fp32 gradient arena, laid out in the order the backward plan produces the gradients (reverse registration
        # order): kernels write parameter gradients straight into it, HipDDP all-reduces contiguous ranges of it, the fused
        # optimizer reads it -- no per-parameter copies.  Slots are padded to 64 floats (Detect's 255-channel bias
        # gradient is produced 256 wide).
        self.goff, off = {}, 0
        for i in reversed(range(len(self.params))):
            self.goff[i] = off
            off += round_up(self.params[i].numel(), 64)
        self.gtotal = off
        self.gflat = self.be.empty((max(off, 64),), torch.float32)
        self.be.zero_(self.gflat)
        self.raw = {}
        self.convs = []
        self._dw_total = 0
        self._keep = []
        self.grad_sink = None   # HipDDP (torch_utils.py): receives every parameter gradient as soon as it is queued
        self.debug_hook = None  # tests only: hook(kind, st, info) behind every weight- / data-gradient launch of the backward plan (tests/test_gpu_train.py)
        max_z, max_ws = 0, 0
        for op in self.spec.ops:
            if op["op"] == "conv":
                m = op["mods"][0]
                assert len(op["mods"]) == 1
                cv = m.conv if hasattr(m, "conv") else m
                has_bn = hasattr(m, "bn")
                if hasattr(m, "conv") and not has_bn:
                    raise NotImplementedError("training needs the un-fused model (Conv blocks with their BatchNorm); do not call fuse()")
                y = op["y"]
                st = dict(op=op, mod=m, cv=cv, has_bn=has_bn)
                if has_bn:
                    c2 = cv.out_channels
                    st["z"] = self.be.empty((B, y.H, y.W, c2), f16)
                    for k in ("mean", "invstd"):
                        st[k] = self.be.empty((c2,), torch.float32)
                    max_z = max(max_z, B * y.H * y.W * c2)
                    max_ws = max(max_ws, self.lib.y5_bn_workspace_bytes(c2, B * y.H * y.W))
                else:
                    st["dbias"] = self.be.empty((op["c2_store"],), torch.float32)
                    max_ws = max(max_ws, self.lib.y5_bn_workspace_bytes(op["c2_store"], B * y.H * y.W))
                self._alloc_conv(st)
                op["_st"] = st
                self.convs.append(st)
            elif op["op"] == "decode":
                self.raw[op["level"]] = None  # allocated per forward: the caller owns the returned maps (models/yolo.py:98)
        self.dz = self.be.empty((max(max_z, 8),), f16)
        self.dwflat = self.be.empty((max(self._dw_total, 64),), torch.float32)  # packed fp32 dW accumulators of all convs
        self.ws = self.be.empty((max(max_ws, 256),), torch.uint8)
        self.ws_bytes = max(max_ws, 256)
        # BatchNorm statistics from the convolution's epilogue (y5_conv2d_fwd_stats: one partial row per workgroup, <= 8 workgroups per CU x 256 channels);
        # Y5_DISABLE=bn_fused_stats keeps the separate statistics pass everywhere
        self.fused_stats = self.dtype == torch.float16 and not _lib.disabled("bn_fused_stats")
        ncu = 256
        if getattr(self.be, "device", None) is not None and torch.cuda.is_available():
            ncu = max(256, torch.cuda.get_device_properties(self.be.device).multi_processor_count)
        self.stats_ws_bytes = ncu * 8 * 2 * 256 * 4   # one [2][<= 256 channels] fp32 row per workgroup, <= 8 workgroups per CU
        self.stats_ws = self.be.empty((self.stats_ws_bytes,), torch.uint8) if self.fused_stats else None

    # ---- helpers ---------------------------------------------------------------------------------------------------
    def _ptr(self, t: TRef, grad=False):
        bufs = self.gbufs if grad else self.bufs
        return self.be.ptr(bufs[t.buf]) + t.c_off * self.es

    def _ld(self, t: TRef):
        return self.spec.bufs[t.buf].C

    def _gptr(self, param):
        """Device address of a parameter's slot in the gradient arena."""
        return self.be.ptr(self.gflat) + self.goff[self._pidx[id(param)]] * 4

    def _f32(self, t):
        """Device pointer of an fp32 parameter / buffer (the emulated backend gets a host copy that is kept alive)."""
        if getattr(self.be, "direct", False):
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise TypeError("training expects contiguous fp32 master parameters")
            return t.data_ptr()
        h = self.be.from_torch(t.detach().float())
        self._keep.append(h)
        return self.be.ptr(h)

    def _geom(self, st):
        """Forward geometry of a conv op (with the paired-pixel view of the 3-channel stem)."""
        op = st["op"]
        x = op["x"]
        (kh, kw), (sh, sw), (ph, pw) = op["k"], op["s"], op["p"]
        H, W, C1, ldx = x.H, x.W, x.C, self._ld(x)
        paired = self.dtype == torch.float16 and op["view"] == "first" and C1 == 4 and kw % 2 == 0 and sw % 2 == 0 and pw % 2 == 0 and W % 2 == 0
        if paired:
            W, C1, ldx, kw, sw, pw = W // 2, 8, 8, kw // 2, sw // 2, pw // 2
        return dict(H=H, W=W, C1=C1, ldx=ldx, k=(kh, kw), s=(sh, sw), p=(ph, pw), paired=paired)

    def _alloc_conv(self, st):
        """Persistent per-convolution buffers: packed forward filter, bias, weight-gradient accumulator and the packed
        sub-filters of the data-gradient parity classes (geometry is fixed; contents are rewritten every step)."""
        be, op, cv = self.be, st["op"], st["cv"]
        x = op["x"]
        c2, c1, kh, kw = cv.weight.shape
        c1v = x.C if op["view"] == "first" else c1           # channels of the NHWC input view (stem: 3 -> 4)
        K = kh * kw * c1v
        st["c1v"], st["K"], st["Kpad"], st["Npad"] = c1v, K, round_up(K, 64), round_up(c2, 32)
        st["wp"] = be.empty((st["Npad"], st["Kpad"]), self.dtype)
        st["bp"] = be.empty((st["Npad"],), torch.float32)
        be.zero_(st["bp"])
        st["dw_off"] = self._dw_total  # slot in the packed weight-gradient arena (one memset per backward pass)
        self._dw_total += round_up(st["Npad"] * st["Kpad"], 64)
        st["c2s"] = c2 if st["has_bn"] else op["c2_store"]
        st["fcfg"] = -1
        # 0.Conv (k6 s2 p2 on the 3-channel image): the forward runs in the NCHW stem kernel without bias / activation (y5_conv_stem_fwd_raw) instead of
        # a table-gather launch of the general kernel on the NHWC copy (369 -> ~185 us at bs 64); the NHWC copy stays for the weight gradient.
        # Y5_DISABLE=train_stem keeps the general kernel.  (fp16 plan, fp16 NCHW input, even H, W % 64 == 0, <= 64 output channels.)
        _, _, Hi, Wi = self.x_shape
        st["stem_w"] = None
        if (op["view"] == "first" and self.dtype == torch.float16 and not _lib.disabled("train_stem") and st["has_bn"]
                and (c1, kh, kw) == (3, 6, 6) and tuple(op["s"]) == (2, 2) and tuple(op["p"]) == (2, 2) and c2 % 8 == 0 and c2 <= 64
                and Hi % 2 == 0 and Wi % 64 == 0):
            st["stem_np"] = round_up(c2, 32)
            st["stem_w"] = be.empty((st["stem_np"] * 144,), torch.float16)
        subs = []
        if op["view"] != "first":
            (sh, sw), (ph, pw) = op["s"], op["p"]
            for rh, th, padh, nh in _axis_classes(kh, sh, ph, x.H):
                for rw, tw, padw, nw in _axis_classes(kw, sw, pw, x.W):
                    if nh == 0 or nw == 0:
                        continue
                    if not th or not tw:
                        raise NotImplementedError("dgrad: parity class without taps (kernel smaller than stride)")
                    K2 = len(th) * len(tw) * st["c2s"]
                    Kp2, Np2 = round_up(K2, 64), round_up(c1, 32)
                    subs.append(dict(rh=rh, rw=rw, nh=nh, nw=nw, th=(C.c_int * len(th))(*th), tw=(C.c_int * len(tw))(*tw), nth=len(th),
                                     ntw=len(tw), pad=(padh, padw), Kpad=Kp2, Npad=Np2, w=be.empty((Np2, Kp2), self.dtype), cfg={}))
            st["zb"] = be.empty((round_up(c1, 32),), torch.float32)
            be.zero_(st["zb"])
        st["subs"] = subs

    def _wgrad_launch(self, d, x_ptr, dz_ptr, ld_dz, dw_ptr, stm):
        """One weight-gradient launch in the form this engine runs: atomics, or (deterministic) per-split slabs + ordered reduction."""
        lib, be = self.lib, self.be
        if self.deterministic and self.dt == _lib.Y5_F16:
            need = int(lib.y5_conv2d_wgrad_ws_bytes(C.byref(d), ld_dz))
            if need < 0:
                _lib.check(-1, lib)
            if self._wg_ws is None or self._wg_ws_bytes < need:
                self._wg_ws_bytes = max(need, 1 << 20)
                self._wg_ws = be.empty((self._wg_ws_bytes // 4 + 64,), torch.float32)
            wsp = be.ptr(self._wg_ws)
            wsp += (-wsp) % 16
            return lib.y5_conv2d_wgrad_det(C.byref(d), _vp(x_ptr), _vp(dz_ptr), ld_dz, _vp(dw_ptr), _vp(wsp), need, stm)
        return lib.y5_conv2d_wgrad(C.byref(d), _vp(x_ptr), _vp(dz_ptr), ld_dz, _vp(dw_ptr), stm)

    def _tune_wgrad(self, d, x_ptr, dz_ptr, ld_dz, dw_ptr, dw_bytes, stm):
        """(kernel family, pixel-range split count) of one weight-gradient launch (csrc/wgrad.hip, wgrad3.h: grid = filter tiles x splits), chosen
        like the forward tiles -- by timing on the real buffers, in the form that will run (atomic / deterministic), once per geometry (persisted
        with the tile choices).  Splits: the library default suits the P1/P2 layers; at P4/P5, where a filter tile is 128 x 128 and the pixel
        range short, half as many splits halve the atomic traffic (scripts/wgrad_bench.py --split-factors: 60 -> 37 us for 512->256 @20^2).
        Family (3x3 layers): the patch-staged kernel (cfg 3 / 341) against the general gather kernel (cfg 1) -- 250 -> 164 us for 0->1.Conv's
        gradient, 209 -> 99 us for 2.C3.m.0.cv2, the general kernel keeps P4/P5 (profiles/r03/r03_wgrad3_ab.log).  Returns (cfg, max_blocks)."""
        mode = os.environ.get("Y5_WGRAD_SPLITS", "auto")  # auto | <n>: fixed split count for every layer (0 = library default)
        fam = int(os.environ.get("Y5_WGRAD_CFG", "-2"))    # -2 = timed | -1 library heuristic | 1 | 3 | 3xy
        k3 = d.KH == 3 and d.KW == 3 and d.PH == 1 and d.PW == 1 and d.SH == d.SW and d.SH in (1, 2)
        if fam >= 3 and fam != 6 and not k3:
            fam = 1
        if fam == 6:
            fam = -1   # (the library picks the stem kernel by geometry)
        if mode != "auto":
            return (fam if fam != -2 else -1), int(mode)
        if not getattr(self.be, "autotune", False) or self.dt != _lib.Y5_F16:
            return (fam if fam != -2 else -1), 0
        from .engine import _TUNE_CACHE, _load_tune_cache, _save_tune_cache
        key = (-7002 - 1000 * _state.cu_budget, int(self.deterministic), fam, d.B, d.H, d.W, d.C1, d.ldx, d.OH, d.OW, d.C2, ld_dz, d.KH, d.KW, d.SH, d.SW, d.Kpad, d.Npad)
        _load_tune_cache()
        if key in _TUNE_CACHE:
            v = _TUNE_CACHE[key][0]
            return (v >> 20) - 1, v & 0xFFFFF
        lib = self.lib
        K = d.KH * d.KW * d.C1
        ncu = torch.cuda.get_device_properties(self.be.device).multi_processor_count
        if 0 < _state.cu_budget < ncu:
            ncu = _state.cu_budget   # (HipDDP reserves CUs for the collective during backward: split counts are chosen for the CUs the plan may fill)
        fams = []
        if fam in (-2, -1, 1):
            fams.append((-1 if fam == -1 else 1, -(-d.C2 // (128 if d.C2 >= 128 else 64)) * -(-K // (128 if K >= 128 else 64)), (1, 1.5, 2, 3, 4, 6)))
        if fam == -2 and d.C2 >= 128 and K >= 128 and d.B * d.OH * d.OW <= 64 * 40 * 40:
            # few pixels, large filter: atomic traffic = splits x filter; the same kernel on 64 x 64 / 64 x 128 tiles fills the chip with fewer splits
            for cfg, tn, tk in ((111, 64, 64), (112, 64, 128)):
                fams.append((cfg, -(-d.C2 // tn) * -(-K // tk), (1, 1.5, 2, 3)))
        elif 100 <= fam < 200:
            tn, tk = 64 * min((fam - 100) // 10, 2 if d.C2 >= 128 else 1), 64 * min((fam - 100) % 10, 2 if K >= 128 else 1)
            fams.append((fam, -(-d.C2 // tn) * -(-K // tk), (1, 1.5, 2, 3, 4, 6)))
        if fam == -2 and d.KH == 6 and d.KW == 3 and d.SH == 2 and d.SW == 1 and d.PH == 2 and d.PW == 1 and d.C1 == 8 and d.ldx == 8:
            fams.append((6, -(-d.C2 // 32), (1, 2, 3, 4)))   # 0.Conv on the paired-pixel view: the stem kernel (517 -> 252 us, scripts/wgrad_bench.py --stem)
        if k3 and (fam == -2 or fam >= 3):
            for cfg in ((3, 341) if fam == -2 else (fam,)):
                cn, cc = ((cfg - 300) // 10, (cfg - 300) % 10) if cfg >= 300 else (4, 2)
                nt, ct = min(cn, 4 if d.C2 > 64 else 2 if d.C2 > 32 else 1), min(cc, 2 if d.C1 > 32 else 1)
                if cfg >= 300 and (nt, ct) == (4 if d.C2 > 64 else 2 if d.C2 > 32 else 1, 2 if d.C1 > 32 else 1):
                    continue   # the cap changes nothing for this layer
                fams.append((cfg, -(-d.C2 // (32 * nt)) * -(-d.C1 // (32 * ct)), (0.75, 1, 1.5, 2, 3)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best, best_ms = (-1, 0), float("inf")
        for cfg, tiles, factors in fams:
            for mb in sorted({max(1, (int(f * ncu) + tiles - 1) // tiles) for f in factors}):
                d.cfg, d.max_blocks = cfg, mb
                _lib.check(self._wgrad_launch(d, x_ptr, dz_ptr, ld_dz, dw_ptr, stm), lib)
                e0.record()
                for _ in range(3):
                    self._wgrad_launch(d, x_ptr, dz_ptr, ld_dz, dw_ptr, stm)
                e1.record()
                e1.synchronize()
                ms = e0.elapsed_time(e1)
                if ms < best_ms:
                    best, best_ms = (cfg, mb), ms
        _lib.check(lib.y5_memset_zero(_vp(dw_ptr), dw_bytes, stm), lib)  # the timing launches accumulated into this layer's dW
        _TUNE_CACHE[key] = ((best[0] + 1) << 20 | best[1], -1)   # (choice, no runner-up): the cache's value layout (engine.autotune_conv)
        _save_tune_cache()
        return best

    # ---- forward ---------------------------------------------------------------------------------------------------
    def forward(self, x):
        lib, be, B = self.lib, self.be, self.spec.B
        if tuple(x.shape) != self.x_shape:
            raise ValueError(f"training engine built for input {self.x_shape}, got {tuple(x.shape)}")
        stm = be.stream()
        self._keep = []
        self._nbt = []
        self._fwd_seq = getattr(self, "_fwd_seq", 0) + 1  # backward checks that its saved activations are still this forward's
        _state.bump_weights_epoch()  # BatchNorm running statistics are updated through raw pointers below
        x, xptr, src_dt = be.input(x)
        self._x_nchw = (xptr, src_dt)
        self._run_jobs(0, stm)  # every forward filter: fp32 master weights -> packed fp16, one launch
        for op in self.spec.ops:
            kind = op["op"]
            if kind == "to_nhwc":
                d = op["dst"]
                scale = 1.0 / 255.0 if src_dt == _lib.Y5_U8 else 1.0
                _lib.check(lib.y5_nchw_to_nhwc(_vp(xptr), src_dt, _vp(self._ptr(d)), self.dt, B, op["C"], d.H, d.W, self._ld(d), scale, stm), lib)
            elif kind == "conv":
                self._fwd_conv(op["_st"], stm)
            elif kind == "sppf_pool":
                b = op["buf"]
                _lib.check(lib.y5_sppf_pool(_vp(self._ptr(b)), self.dt, B, b.H, b.W, op["C"], self._ld(b), op["k"], stm), lib)
            elif kind == "decode":
                lg = op["x"]
                self.raw[op["level"]] = be.empty((B, op["na"], op["ny"], op["nx"], op["no"]), self.dtype)
                if self.dtype == torch.float16:
                    _lib.check(lib.y5_nhwc_to_raw(_vp(self._ptr(lg)), _vp(be.ptr(self.raw[op["level"]])), B, op["ny"] * op["nx"], op["na"], op["no"],
                                                  self._ld(lg), stm), lib)
                else:
                    _lib.check(lib.y5_train_glue_f32(0, _vp(self._ptr(lg)), _vp(be.ptr(self.raw[op["level"]])), B, op["ny"] * op["nx"], op["na"], op["no"],
                                                     self._ld(lg), 0, 0, stm), lib)
            else:
                raise NotImplementedError(kind)
        if self._nbt:
            torch._foreach_add_(self._nbt, 1)  # BatchNorm2d.num_batches_tracked of every layer: one launch instead of 57
        return [self.raw[i] for i in sorted(self.raw)]

    def _fwd_conv(self, st, stm):
        lib, be, B = self.lib, self.be, self.spec.B
        op, cv = st["op"], st["cv"]
        x, y, res, y2 = op["x"], op["y"], op["res"], op["y2"]
        g = self._geom(st)
        c2, c1, kh, kw = cv.weight.shape
        Kpad, Npad = st["Kpad"], st["Npad"]
        if not st["has_bn"] and cv.bias is not None:
            if getattr(be, "direct", False):
                st["bp"][:c2].copy_(cv.bias.detach())
            else:
                st["bp"][:c2] = cv.bias.detach().float().numpy()
        if st["has_bn"]:
            out_ptr, ldo = be.ptr(st["z"]), c2
        else:
            out_ptr, ldo = self._ptr(y), self._ld(y)
        d = _lib.ConvDesc(dtype=self.dt, B=B, H=g["H"], W=g["W"], C1=g["C1"], ldx=g["ldx"], OH=y.H, OW=y.W, C2=st["c2s"], ldy=ldo,
                          KH=g["k"][0], KW=g["k"][1], SH=g["s"][0], SW=g["s"][1], PH=g["p"][0], PW=g["p"][1], act=0, Kpad=Kpad, Npad=Npad,
                          ldr=0, ld2=0, cfg=st["fcfg"], max_blocks=0)
        ptrs = (_vp(self._ptr(x)), _vp(be.ptr(st["wp"])), _vp(be.ptr(st["bp"])), None, _vp(out_ptr), None)
        if st["fcfg"] < 0 and getattr(be, "autotune", False):
            st["fcfg"] = d.cfg = autotune_conv(lib, d, ptrs, stm)
        xp, xdt = getattr(self, "_x_nchw", (0, -1))
        stats_rows = 0
        if st["stem_w"] is not None and xdt == _lib.Y5_F16 and xp % 16 == 0:
            _lib.check(lib.y5_conv_stem_fwd_raw(_vp(xp), B, self.x_shape[2], self.x_shape[3], _vp(be.ptr(st["stem_w"])), c2, st["stem_np"], _vp(out_ptr), ldo,
                                                0, stm), lib)
        else:
            if self.fused_stats and st["has_bn"] and st.get("stats_ok", True) and self._sync_world(st["mod"].bn) == 1:
                # the streaming kernels of P1-P3 leave sum z / sum z^2 per workgroup beside z: no statistics pass over z (models/common.py:82-88)
                rows = C.c_int(0)
                rc = lib.y5_conv2d_fwd_stats(C.byref(d), ptrs[0], ptrs[1], ptrs[2], ptrs[4], _vp(be.ptr(self.stats_ws)), self.stats_ws_bytes, C.byref(rows), stm)
                if _lib.experimental("stats_debug") and "stats_ok" not in st:
                    print(f"[stats] {op['name']:16s} cfg {d.cfg:3d} {'fused' if rc == 0 else 'separate pass'}  z {B * y.H * y.W * c2 * 2 / 1e6:.0f} MB", flush=True)
                if rc in (_lib.Y5_ERR_UNSUPPORTED, _lib.Y5_ERR_WORKSPACE):
                    # this layer's configuration is not a streaming kernel -- or its grid leaves more partial rows than the workspace holds (a part with
                    # more CUs / occupancy than it was sized for): separate pass, decided once
                    st["stats_ok"] = False
                else:
                    _lib.check(rc, lib)
                    st["stats_ok"], stats_rows = True, rows.value
            if not stats_rows:
                _lib.check(lib.y5_conv2d_fwd(C.byref(d), *ptrs, stm), lib)
        if st["has_bn"]:
            bn = st["mod"].bn
            npix = B * y.H * y.W
            st["gamma"], st["beta"] = self._f32(bn.weight), self._f32(bn.bias)
            rm, rv = self._running(bn)
            world = self._sync_world(bn)
            if stats_rows:
                _lib.check(lib.y5_bn_silu_fwd_from_partials(_vp(be.ptr(st["z"])), self.dt, npix, c2, c2, _vp(st["gamma"]), _vp(st["beta"]), float(bn.eps),
                                                            float(bn.momentum if bn.momentum is not None else 0.1), rm, rv, _vp(be.ptr(st["mean"])),
                                                            _vp(be.ptr(st["invstd"])), _vp(be.ptr(self.stats_ws)), stats_rows,
                                                            _vp(self._ptr(res)) if res is not None else None, self._ld(res) if res is not None else 0,
                                                            _vp(self._ptr(y)), self._ld(y), stm), lib)
            elif world > 1:
                # SyncBatchNorm (train.py:269-271): per-channel sum z / sum z^2 of this rank, all-reduce(SUM) over the ranks (2 C doubles; torch's
                # SyncBatchNorm all-gathers mean / invstd / count per layer at the same point), statistics of the GLOBAL batch, then the apply pass
                sums = st.get("sums")
                if sums is None:
                    sums = st["sums"] = be.empty((2 * c2,), torch.float64)
                _lib.check(lib.y5_bn_stats(_vp(be.ptr(st["z"])), self.dt, npix, c2, c2, _vp(be.ptr(sums)), _vp(be.ptr(self.ws)), self.ws_bytes, stm), lib)
                torch.distributed.all_reduce(be.view_torch(sums), group=bn.process_group)   # (the same rank set as the image count)
                _lib.check(lib.y5_bn_silu_fwd_from_sums(_vp(be.ptr(st["z"])), self.dt, npix, c2, c2, _vp(st["gamma"]), _vp(st["beta"]), float(bn.eps),
                                                        float(bn.momentum if bn.momentum is not None else 0.1), rm, rv, _vp(be.ptr(st["mean"])),
                                                        _vp(be.ptr(st["invstd"])), _vp(be.ptr(sums)), y.H * y.W * self._sync_images(bn, world),
                                                        _vp(self._ptr(res)) if res is not None else None, self._ld(res) if res is not None else 0,
                                                        _vp(self._ptr(y)), self._ld(y), stm), lib)
            else:
                _lib.check(lib.y5_bn_silu_fwd(_vp(be.ptr(st["z"])), self.dt, npix, c2, c2, _vp(st["gamma"]), _vp(st["beta"]), float(bn.eps),
                                              float(bn.momentum if bn.momentum is not None else 0.1), rm, rv, _vp(be.ptr(st["mean"])),
                                              _vp(be.ptr(st["invstd"])), _vp(self._ptr(res)) if res is not None else None,
                                              self._ld(res) if res is not None else 0, _vp(self._ptr(y)), self._ld(y), _vp(be.ptr(self.ws)),
                                              self.ws_bytes, stm), lib)
            self._running_done(bn)
        if y2 is not None:
            _lib.check(lib.y5_upsample2x(_vp(self._ptr(y)), self.dt, _vp(self._ptr(y2)), B, y.H, y.W, y.C, self._ld(y), self._ld(y2), stm), lib)

    @staticmethod
    def _sync_world(bn):
        """World size over which a BatchNorm layer's statistics are shared: > 1 only for torch.nn.SyncBatchNorm modules (what
        `SyncBatchNorm.convert_sync_batchnorm(model)` of train.py:269-271 leaves behind) inside an initialised process group."""
        if isinstance(bn, torch.nn.SyncBatchNorm) and torch.distributed.is_available() and torch.distributed.is_initialized():
            return torch.distributed.get_world_size(bn.process_group) if bn.process_group is not None else torch.distributed.get_world_size()
        return 1

    def _sync_images(self, bn, world):
        """Images over which a SyncBatchNorm layer's statistics run: the sum of the ranks' batch sizes, exchanged ONCE per forward (torch's
        SyncBatchNorm all-gathers the per-rank counts; `npix * world` was only right while every rank held the same number of images, which a
        loader without padding does not guarantee).  The backward of the same step reuses the forward's figure."""
        tot = getattr(self, "_sync_total", None)
        if tot is None or tot[0] != self._fwd_seq:
            t = torch.tensor([float(self.spec.B)], dtype=torch.float64, device=self.be.view_torch(self.gflat).device)
            torch.distributed.all_reduce(t, group=bn.process_group)
            tot = self._sync_total = (self._fwd_seq, int(round(float(t.item()))))
        return tot[1]

    def _running(self, bn):
        """Device pointers of the BatchNorm running statistics (updated in place by the kernel)."""
        if not bn.track_running_stats or bn.running_mean is None:
            return None, None
        if getattr(self.be, "direct", False):
            return _vp(bn.running_mean.data_ptr()), _vp(bn.running_var.data_ptr())
        rm, rv = self.be.from_torch(bn.running_mean.detach().float()), self.be.from_torch(bn.running_var.detach().float())
        self._run_tmp = (rm, rv)
        return _vp(self.be.ptr(rm)), _vp(self.be.ptr(rv))

    def _running_done(self, bn):
        if not bn.track_running_stats or bn.running_mean is None:
            return
        if not getattr(self.be, "direct", False):  # host-emulated (numpy) backend: copy the updated statistics back
            rm, rv = self._run_tmp
            bn.running_mean.copy_(self.be.to_torch(rm))
            bn.running_var.copy_(self.be.to_torch(rv))
        if bn.num_batches_tracked is not None:
            self._nbt.append(bn.num_batches_tracked)

    # ---- backward --------------------------------------------------------------------------------------------------
    def backward(self, dps):
        lib, be, B = self.lib, self.be, self.spec.B
        stm = be.stream()
        written = {}

        def is_written(t: TRef):
            iv = written.get(t.buf, [])
            lo, hi = t.c_off, t.c_off + t.C
            covered = any(a <= lo and hi <= b for a, b in iv)
            if not covered and any(a < hi and lo < b for a, b in iv):
                raise RuntimeError("training plan: partially written gradient slice")
            return covered

        def mark(t: TRef):
            iv = written.setdefault(t.buf, [])
            iv.append((t.c_off, t.c_off + t.C))
            iv.sort()
            merged = [iv[0]]
            for a, b in iv[1:]:
                if a <= merged[-1][1]:
                    merged[-1] = (merged[-1][0], max(merged[-1][1], b))
                else:
                    merged.append((a, b))
            written[t.buf] = merged

        sink = self.grad_sink
        if sink is not None:
            sink.begin(self)
        try:
            return self._backward_plan(dps, stm, is_written, mark, sink)
        except BaseException:
            if sink is not None:
                sink.abort()   # (a reserved CU budget must not outlive a failed backward: ADVICE r5)
            raise

    def _backward_plan(self, dps, stm, is_written, mark, sink):
        lib, be, B = self.lib, self.be, self.spec.B
        params = self.params
        _lib.check(lib.y5_memset_zero(_vp(be.ptr(self.dwflat)), self._dw_total * 4, stm), lib)  # every conv's dW accumulator at once
        self._run_jobs(1, stm)  # every data-gradient sub-filter, one launch

        class _G(list):  # grads[i] = True: the gradient's kernels are queued -> tell the sink (bucketed all-reduce overlaps the rest)
            def __setitem__(s2, i, t):
                list.__setitem__(s2, i, t)
                if sink is not None and t is not None:
                    sink.grad_ready(i)

        grads = _G([None] * len(params))
        hold = []
        self._unpack_later = []
        self._unpack_pending, self._unpack_bucket = [], None
        for op in reversed(self.spec.ops):
            kind = op["op"]
            if kind == "decode":
                lg = op["x"]
                dph, dpp, _ = be.input(dps[op["level"]])
                hold.append(dph)
                if self.dtype == torch.float16:
                    _lib.check(lib.y5_raw_to_nhwc(_vp(dpp), _vp(self._ptr(lg, True)), B, op["ny"] * op["nx"], op["na"], op["no"], self._ld(lg), stm), lib)
                else:
                    _lib.check(lib.y5_train_glue_f32(1, _vp(dpp), _vp(self._ptr(lg, True)), B, op["ny"] * op["nx"], op["na"], op["no"], self._ld(lg), 0, 0, stm), lib)
                mark(lg)
            elif kind == "sppf_pool":
                b = op["buf"]
                if not is_written(b):
                    raise RuntimeError("training plan: SPPF gradient buffer not produced")
                if self.dtype == torch.float16:
                    _lib.check(lib.y5_sppf_pool_bwd(_vp(self._ptr(b)), _vp(self._ptr(b, True)), B, b.H, b.W, op["C"], self._ld(b), self._ld(b), op["k"], stm), lib)
                else:
                    _lib.check(lib.y5_train_glue_f32(4, _vp(self._ptr(b)), _vp(self._ptr(b, True)), B, b.H, b.W, op["C"], self._ld(b), self._ld(b), op["k"], stm), lib)
            elif kind == "conv":
                self._bwd_conv(op["_st"], stm, is_written, mark, grads, hold)
            elif kind == "to_nhwc":
                pass
            else:
                raise NotImplementedError(kind)
        if self._unpack_pending:
            self._flush_unpack(stm, grads)
        if self._unpack_later:
            self._run_jobs(2, stm)  # packed fp32 dW of every conv -> parameter layout in the gradient arena, one launch
            for i in self._unpack_later:
                grads[i] = True
        if sink is not None:
            sink.finish(grads)
        flat = be.view_torch(self.gflat)  # (fresh view tensors: autograd's AccumulateGrad can adopt them without a copy)
        return [None if g is None else flat[self.goff[i]:self.goff[i] + p.numel()].view(p.shape) for i, (p, g) in enumerate(zip(params, grads))]

    def _flush_unpack(self, stm, grads):
        """Unpack the packed weight gradients of the layers collected for one gradient bucket (one launch), then report them to the sink."""
        sts, self._unpack_pending, self._unpack_bucket = self._unpack_pending, [], None
        self._run_jobs(2, stm, convs=sts)
        for st in sts:
            grads[self._pidx[id(st["cv"].weight)]] = True

    def _run_jobs(self, kind, stm, convs=None):
        """One y5_filter_jobs launch: kind 0 = pack every forward filter, 1 = pack every data-gradient sub-filter, 2 = unpack every
        weight gradient (convs: only those of these layers -- one launch per gradient bucket under HipDDP).  The device-resident job table is
        rebuilt only when a pointer changed."""
        import numpy as np

        lib, be = self.lib, self.be
        cache = self.__dict__.setdefault("_job_tables", {})
        sel = self.convs if convs is None else convs
        ckey = kind if convs is None else (kind, tuple(id(st) for st in convs))
        ent = cache.get(ckey)
        wkey = tuple(self._f32(st["cv"].weight) for st in sel)  # (all other pointers are engine-owned buffers)
        if ent is None or ent[0] != wkey:
            rows = []
            for st, wptr in zip(sel, wkey):
                cv = st["cv"]
                c2, c1, kh, kw = cv.weight.shape
                if kind == 0:
                    rows.append((wptr, be.ptr(st["wp"]), st["Npad"] * st["Kpad"], 0, c2, c1, kh, kw, st["c1v"], 0, st["Kpad"], st["Npad"], 0, 0, (), ()))
                    if st["stem_w"] is not None:   # the stem kernel's own filter layout ([Npad][144], y5_filter_jobs kind 3)
                        rows.append((wptr, be.ptr(st["stem_w"]), st["stem_np"] * 144, 3, c2, c1, kh, kw, 0, 0, 144, st["stem_np"], 0, 0, (), ()))
                elif kind == 2:
                    rows.append((be.ptr(self.dwflat) + st["dw_off"] * 4, self._gptr(cv.weight), c2 * c1 * kh * kw, 2, c2, c1, kh, kw, st["c1v"], 0,
                                 st["Kpad"], st["Npad"], 0, 0, (), ()))
                else:
                    for sub in st["subs"]:
                        rows.append((wptr, be.ptr(sub["w"]), sub["Npad"] * sub["Kpad"], 1, c2, c1, kh, kw, 0, st["c2s"], sub["Kpad"], sub["Npad"],
                                     sub["nth"], sub["ntw"], tuple(sub["th"]), tuple(sub["tw"])))
            if not rows:
                return
            arr = (_lib.FilterJob * len(rows))()
            for j, r in zip(arr, rows):
                (j.src, j.dst, j.total, j.kind, j.C2, j.C1, j.KH, j.KW, j.C1_view, j.C2_view, j.Kpad, j.Npad, j.nth, j.ntw) = r[:14]
                for q, v in enumerate(r[14]):
                    j.th[q] = v
                for q, v in enumerate(r[15]):
                    j.tw[q] = v
                j.reserved = 1 if (self.dtype == torch.float32 and j.kind != 2) else 0  # fp32 plan: packed filters are fp32
            tab = be.from_torch(torch.from_numpy(np.frombuffer(arr, dtype=np.uint8).copy()))
            ent = (wkey, tab, len(rows), max(r[2] for r in rows))
            cache[ckey] = ent
        _lib.check(lib.y5_filter_jobs(_vp(be.ptr(ent[1])), ent[2], ent[3], stm), lib)

    def _bwd_conv(self, st, stm, is_written, mark, grads, hold):
        lib, be, B = self.lib, self.be, self.spec.B
        op, cv, m = st["op"], st["cv"], st["mod"]
        x, y, res, y2 = op["x"], op["y"], op["res"], op["y2"]
        npix = B * y.H * y.W
        c2, c1, kh, kw = cv.weight.shape
        c2s = st["c2s"]
        if y2 is not None:
            if self.dtype == torch.float16:
                _lib.check(lib.y5_upsample2x_bwd(_vp(self._ptr(y2, True)), _vp(self._ptr(y, True)), B, y.H, y.W, y.C, self._ld(y2), self._ld(y),
                                                 1 if is_written(y) else 0, stm), lib)
            else:
                _lib.check(lib.y5_train_glue_f32(2, _vp(self._ptr(y2, True)), _vp(self._ptr(y, True)), B, y.H, y.W, y.C, self._ld(y2), self._ld(y),
                                                 1 if is_written(y) else 0, stm), lib)
            mark(y)
        if not is_written(y):
            raise RuntimeError(f"training plan: gradient of {op['name']} output was never produced")
        if st["has_bn"]:
            if res is not None:
                if self.dtype == torch.float16:
                    _lib.check(lib.y5_add_slice(_vp(self._ptr(y, True)), _vp(self._ptr(res, True)), npix, y.C, self._ld(y), self._ld(res),
                                                1 if is_written(res) else 0, stm), lib)
                else:
                    _lib.check(lib.y5_train_glue_f32(3, _vp(self._ptr(y, True)), _vp(self._ptr(res, True)), npix, 1, 1, y.C, self._ld(y), self._ld(res),
                                                     1 if is_written(res) else 0, stm), lib)
                mark(res)
            world = self._sync_world(m.bn)
            if world > 1:
                # SyncBatchNorm backward: this rank's dgamma / dbeta are the parameter gradients (averaged with all the others later); dz needs the
                # sums over the GLOBAL batch -- all-reduce(SUM) of a copy, as torch's SyncBatchNorm does with (sum_dy, sum_dy_xmu)
                _lib.check(lib.y5_bn_bwd_stats(_vp(self._ptr(y, True)), self._ld(y), _vp(be.ptr(st["z"])), c2, self.dt, npix, c2, _vp(st["gamma"]),
                                               _vp(st["beta"]), _vp(be.ptr(st["mean"])), _vp(be.ptr(st["invstd"])), _vp(self._gptr(m.bn.weight)),
                                               _vp(self._gptr(m.bn.bias)), _vp(be.ptr(self.ws)), self.ws_bytes, stm), lib)
                gs = st.get("gsum")
                if gs is None:
                    gs = st["gsum"] = be.empty((2 * c2,), torch.float32)
                ow, ob = self.goff[self._pidx[id(m.bn.weight)]], self.goff[self._pidx[id(m.bn.bias)]]
                gst, gft = be.view_torch(gs), be.view_torch(self.gflat)
                gst[:c2].copy_(gft[ow:ow + c2])
                gst[c2:].copy_(gft[ob:ob + c2])
                torch.distributed.all_reduce(gst, group=m.bn.process_group)
                _lib.check(lib.y5_bn_silu_bwd_from_sums(_vp(self._ptr(y, True)), self._ld(y), _vp(be.ptr(st["z"])), c2, self.dt, npix, c2, _vp(st["gamma"]),
                                                        _vp(st["beta"]), _vp(be.ptr(st["mean"])), _vp(be.ptr(st["invstd"])), _vp(be.ptr(gs)),
                                                        _vp(be.ptr(gs) + 4 * c2), y.H * y.W * self._sync_images(m.bn, world), _vp(be.ptr(self.dz)), c2, stm), lib)
            else:
                _lib.check(lib.y5_bn_silu_bwd(_vp(self._ptr(y, True)), self._ld(y), _vp(be.ptr(st["z"])), c2, self.dt, npix, c2,
                                              _vp(st["gamma"]), _vp(st["beta"]), _vp(be.ptr(st["mean"])), _vp(be.ptr(st["invstd"])),
                                              _vp(be.ptr(self.dz)), c2, _vp(self._gptr(m.bn.weight)), _vp(self._gptr(m.bn.bias)),
                                              _vp(be.ptr(self.ws)), self.ws_bytes, stm), lib)
            dz_ptr, ld_dz = be.ptr(self.dz), c2
            grads[self._pidx[id(m.bn.weight)]] = True
            grads[self._pidx[id(m.bn.bias)]] = True
        else:
            dz_ptr, ld_dz = self._ptr(y, True), self._ld(y)
            dbias = self._gptr(cv.bias) if cv.bias is not None else be.ptr(st["dbias"])  # (arena slots are 64-float padded: c2s fits)
            _lib.check(lib.y5_channel_sum(_vp(dz_ptr), self.dt, npix, c2s, ld_dz, _vp(dbias), _vp(be.ptr(self.ws)), self.ws_bytes, stm), lib)
            if cv.bias is not None:
                grads[self._pidx[id(cv.bias)]] = True
        # weight gradient: packed fp32 accumulator -> parameter layout
        g = self._geom(st)
        Kpad, Npad = st["Kpad"], st["Npad"]
        d = _lib.ConvDesc(dtype=self.dt, B=B, H=g["H"], W=g["W"], C1=g["C1"], ldx=g["ldx"], OH=y.H, OW=y.W, C2=c2s, ldy=ld_dz,
                          KH=g["k"][0], KW=g["k"][1], SH=g["s"][0], SW=g["s"][1], PH=g["p"][0], PW=g["p"][1], act=0, Kpad=Kpad, Npad=Npad,
                          cfg=-1, max_blocks=0)
        dw_ptr = be.ptr(self.dwflat) + st["dw_off"] * 4
        if "wg_choice" not in st:
            st["wg_choice"] = self._tune_wgrad(d, self._ptr(x), dz_ptr, ld_dz, dw_ptr, Npad * Kpad * 4, stm)
        d.cfg, d.max_blocks = st["wg_choice"]
        # (deterministic: fixed-order reduction of the pixel-range splits through a workspace, csrc/wgrad.hip DET -- bit-identical gradients run to run)
        _lib.check(self._wgrad_launch(d, self._ptr(x), dz_ptr, ld_dz, dw_ptr, stm), lib)
        if self.debug_hook is not None:
            self.debug_hook("wgrad", st, dict(dz_ptr=dz_ptr, ld_dz=ld_dz, geom=g, choice=st["wg_choice"]))
        if self.grad_sink is not None:
            # a gradient sink (HipDDP) wants every gradient as early as possible -- but one unpack launch per LAYER was 56 extra launches per step
            # (most of the "exposed exchange" of a one-rank group, profiles/r05/r05_ddp_exchange_trace.log): the packed gradients of the layers whose
            # weights share a bucket are unpacked by ONE job launch when the bucket's last layer has queued its weight gradient
            widx = self._pidx[id(cv.weight)]
            bk = self.grad_sink.bucket_of(widx)
            if self._unpack_bucket is not None and bk is not self._unpack_bucket:
                self._flush_unpack(stm, grads)
            self._unpack_bucket = bk
            self._unpack_pending.append(st)
        else:
            self._unpack_later.append(self._pidx[id(cv.weight)])
        # data gradient: one forward launch per parity class on the re-packed sub-filter
        if not st["subs"]:
            return
        acc = is_written(x)
        dense = tuple(op["s"]) == (1, 1)
        if self.debug_hook is not None:
            self.debug_hook("pre_dgrad", st, dict(dz_ptr=dz_ptr, ld_dz=ld_dz, acc=acc))
        for sub in st["subs"]:
            dd = _lib.ConvDesc(dtype=self.dt, B=B, H=y.H, W=y.W, C1=c2s, ldx=ld_dz, OH=sub["nh"], OW=sub["nw"], C2=x.C, ldy=self._ld(x),
                               KH=sub["nth"], KW=sub["ntw"], SH=1, SW=1, PH=sub["pad"][0], PW=sub["pad"][1], act=0, Kpad=sub["Kpad"],
                               Npad=sub["Npad"], ldr=self._ld(x) if acc else 0, ld2=0, cfg=sub["cfg"].get(acc, -1), max_blocks=0,
                               out_mul_h=0 if dense else op["s"][0], out_mul_w=0 if dense else op["s"][1], out_off_h=sub["rh"],
                               out_off_w=sub["rw"], out_H=0 if dense else x.H, out_W=0 if dense else x.W)
            gx = _vp(self._ptr(x, True))
            ptrs = (_vp(dz_ptr), _vp(be.ptr(sub["w"])), _vp(be.ptr(st["zb"])), gx if acc else None, gx, None)
            if acc not in sub["cfg"] and getattr(be, "autotune", False):
                if not acc:
                    sub["cfg"][acc] = dd.cfg = autotune_conv(lib, dd, ptrs, stm)
                else:
                    # Timing replays the launch, and this one ACCUMULATES into its output: until round 6 such launches (every input with a second reader:
                    # C3's shared input, the stride-2 parity classes after the first ...) ran the untuned default tile.  The race is run on the
                    # non-accumulating form of the same launch (no residual read) with the gradient buffer saved and restored around it -- once per plan.
                    gt = be.view_torch(self.gbufs[x.buf])
                    keep = gt.clone()
                    dn = _lib.ConvDesc.from_buffer_copy(dd)
                    dn.ldr, dn.cfg = 0, -1
                    # (the streaming pointwise families 14-21 / 56 / 84-87 and the K-streamed 93 / 94 have no residual path: out of this race)
                    sub["cfg"][acc] = dd.cfg = autotune_conv(lib, dn, (ptrs[0], ptrs[1], ptrs[2], None, ptrs[4], None), stm, exclude=NO_RESIDUAL_CFGS)
                    gt.copy_(keep)
                    del keep
            rc = lib.y5_conv2d_fwd(C.byref(dd), *ptrs, stm)
            if rc == _lib.Y5_ERR_UNSUPPORTED and acc and dd.cfg >= 0:   # a configuration chosen on the non-accumulating form that refuses the residual: default tile
                sub["cfg"][acc] = dd.cfg = -1
                rc = lib.y5_conv2d_fwd(C.byref(dd), *ptrs, stm)
            _lib.check(rc, lib)
        if self.debug_hook is not None:
            self.debug_hook("dgrad", st, dict(dz_ptr=dz_ptr, ld_dz=ld_dz, acc=acc, cfgs=[sub["cfg"].get(acc) for sub in st["subs"]]))
        mark(x)


class _TrainFn(torch.autograd.Function):
    """parameters -> raw head outputs, with the HIP backward plan as the gradient."""

    @staticmethod
    def forward(ctx, eng, x, *params):
        ctx.eng = eng
        outs = eng.forward(x)
        ctx.seq = eng._fwd_seq
        return tuple(eng.be.to_torch(o) for o in outs)

    @staticmethod
    def backward(ctx, *dps):
        eng = ctx.eng
        if ctx.seq != eng._fwd_seq:
            # activations, BatchNorm statistics and the raw head maps live in engine-owned buffers that the later forward overwrote
            # (the reference would keep one autograd graph per forward alive; one training plan holds exactly one)
            raise RuntimeError("yolov5_amd: backward() of a training forward that is no longer the model's latest one -- "
                               "call backward before the next train-mode forward at this input shape")
        # gradient accumulation (a second backward before zero_grad): .grad tensors that alias the arena would be overwritten
        # by this pass before autograd adds to them -- give those their own storage first, and hand autograd copies
        lo = eng.be.ptr(eng.gflat)
        hi = lo + eng.gtotal * 4
        accumulating = False
        for p in eng.params:
            if p.grad is not None:
                accumulating = True
                if lo <= p.grad.data_ptr() < hi:
                    p.grad = p.grad.clone()
        grads = eng.backward([d.contiguous() for d in dps])
        if accumulating:
            grads = [None if g is None else g.clone() for g in grads]
        return (None, None, *grads)


def train_forward(model, x):
    """Train-mode `BaseModel._forward_once` (models/yolo.py:160-170): list of (bs, na, ny, nx, no) fp16 tensors."""
    # fp32 images -> the fp32 plan (reference-precision mode: train.py without AMP); fp16 / uint8 images -> the fp16 (AMP) plan
    dtype = torch.float32 if x.dtype == torch.float32 else torch.float16
    key = (tuple(x.shape), str(x.device), dtype)
    cache = model.__dict__.setdefault("_train_engines", {})
    eng = cache.get(key)
    if eng is None:
        cache.clear()
        eng = TrainEngine(model, tuple(x.shape), x.device, dtype=dtype)
        cache[key] = eng
    eng.grad_sink = model.__dict__.get("_ddp_sink")
    return list(_TrainFn.apply(eng, x, *eng.params))

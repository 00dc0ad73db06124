"""Host-side layers: ``torch.autograd.Function`` wrappers whose forward/backward are sequences of C-ABI
HIP kernels (cddmsl_amd.hip).  This mirrors the reference's own native-op convention
(``autograd.Function`` over ``_C.<op>_forward/_backward``, detectron2/layers/roi_align_rotated.py:11-47).

Conventions
  * activations are NHWC in the compute dtype T (bf16 throughput path / f32 parity path);
  * parameters are f32 masters with the reference's names; conv weights are ``channels_last`` OIHW
    (= [Cout][KH][KW][Cin] in memory), so the kernels read them in place;
  * weight gradients are accumulated by the wgrad kernels straight into ``param.grad`` (f32, same
    memory layout as the parameter) -- the Functions return ``None`` for parameter inputs.
"""
import os
import weakref

import torch

from . import hip

_STEP = [0]  # bumped by the optimizer; keys the per-step cache of prepared (cast/transposed) weights


def bump_weight_version():
    _STEP[0] += 1


_LOAD_GEN = [0]  # bumped whenever a state dict is loaded into a module of this package: caches of DERIVED tensors (folded
#                  FrozenBN affines, padded stem weights, block-parameter tuples) key on it; caches of prepared parameter
#                  copies additionally key on the parameter's ``_version`` (load_state_dict copies in place: same data_ptr)


def note_weights_loaded(*_):
    _LOAD_GEN[0] += 1
    _STEP[0] += 1


def load_generation():
    return _LOAD_GEN[0]


_TOUCHED = set()   # ids of parameters that received a gradient this step (SGD skips grad-less parameters)


_TOUCH_HOOK = [None]   # callable(param) run at every gradient-buffer hand-out (engine.GradBuckets: all-reduce overlapped with backward)


def _grad_buf(p):
    """The buffer a weight-gradient kernel accumulates into.  Every write of a parameter gradient on the hot path goes through
    here, immediately before the launch that writes it: the one place that knows the order in which gradients are produced."""
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.preserve_format)
    _TOUCHED.add(id(p))
    if _TOUCH_HOOK[0] is not None:
        _TOUCH_HOOK[0](p)
    return p.grad


def take_touched():
    t = set(_TOUCHED)
    _TOUCHED.clear()
    return t


def _ohwi(w):
    """[Cout,Cin,KH,KW] channels_last parameter (or a 2-D linear weight) -> contiguous [Cout,KH,KW,Cin] view."""
    if w.dim() == 2:
        return w.view(w.shape[0], 1, 1, w.shape[1])
    v = w.permute(0, 2, 3, 1)
    assert v.is_contiguous(), "conv weights must be channels_last"
    return v


class PreparedWeight:
    """Per-step cache of the T-dtype forward weights and the flipped/transposed (and BN-scaled) dgrad weights.

    Trainable weights register themselves; the first ``get`` of a step refreshes ALL registered weights that were used in the
    previous step with one ``cddmsl_weight_prep_multi`` launch into their persistent buffers (instead of ~110 small launches
    scattered through the forward pass).  Frozen weights are prepared once, individually."""

    _live = weakref.WeakSet()
    _table = None            # (signature, device int64 table, dtype)

    def __init__(self, param, scale=None, frozen=False):
        self.param, self.scale, self.frozen = param, scale, frozen
        self._key = None
        self._wf = self._wd = None
        self._dtype, self._used = None, False
        if not frozen:
            PreparedWeight._live.add(self)

    def _sig(self):
        return (self.param.data_ptr(), 0 if self.scale is None else self.scale.data_ptr(), self._wf.data_ptr(),
                0 if self._wd is None else self._wd.data_ptr())

    @staticmethod
    def _refresh_all(dtype):
        """one launch for every trainable weight that has buffers of this dtype and was used last step"""
        items = [pw for pw in PreparedWeight._live if pw._wf is not None and pw._dtype == dtype and pw._used
                 and pw._key is not None and pw._key[2] == pw.param.data_ptr()]   # (same storage; the refresh re-reads the master whatever its version)
        if not items:
            return
        sig = tuple(pw._sig() for pw in items)
        if PreparedWeight._table is None or PreparedWeight._table[0] != sig or PreparedWeight._table[2] != dtype:
            rows = []
            for pw in items:
                Cout, KH, KW, Cin = _ohwi(pw.param).shape
                rows.append(list(pw._sig()) + [Cout, KH, KW, Cin])
            from ._lib import to_device_async
            PreparedWeight._table = (sig, to_device_async(torch.tensor(rows, dtype=torch.int64), items[0].param.device), dtype)
        hip.weight_prep_multi(PreparedWeight._table[1], len(items), dtype)
        for pw in items:
            pw._key = (dtype, _STEP[0], pw.param.data_ptr(), pw.param._version)
            pw._used = False

    def get(self, dtype, need_dgrad=True):
        key = (dtype, -1 if self.frozen else _STEP[0], self.param.data_ptr(), self.param._version)
        if key != self._key and not self.frozen and self._wf is not None and self._dtype == dtype:
            PreparedWeight._refresh_all(dtype)        # first stale weight of the step: bring every registered weight up to date
        if key != self._key or (need_dgrad and self._wd is None):
            wf, wd = hip.weight_prep(_ohwi(self.param.detach()), self.scale, dtype, True, need_dgrad)
            self._wf, self._wd, self._key, self._dtype = wf, (wd if need_dgrad else None), key, dtype
        self._used = True
        return self._wf, self._wd


# ------------------------------------------------------------------------------------------------
# OCP e4m3 (fp8) forward GEMMs -- BASELINE.json configs[4]; no counterpart in the reference (AMP off, config/defaults.py:697)
# ------------------------------------------------------------------------------------------------
class Fp8Slot:
    """one tensor's scaling state (views into ``Fp8Scales.buf``).  ``calibrated`` (host flag): the first time a tensor is
    quantised its scale is taken from the tensor itself (one extra reduction, no host sync) -- a gradient of magnitude 1e-6 would
    otherwise flush to zero under the initial scale of 1, and so would everything computed from it, so that a chain of L
    layers needed L steps to find its scales.  Until then no producer writes the e4m3 copy on the consumer's behalf."""
    __slots__ = ("scale", "deq", "amax", "calibrated")

    def __init__(self, scale, deq, amax):
        self.scale, self.deq, self.amax, self.calibrated = scale, deq, amax, False

    def calibrate(self, t):
        if not self.calibrated:
            amax = t.detach().abs().amax().float().view(1)
            ok = torch.isfinite(amax) & (amax > 0)          # (a non-finite tensor keeps scale 1 and raises the table's flag)
            sc = torch.where(ok, hip.FP8_MAX / amax.clamp(1e-30, 3.0e38), torch.ones_like(amax))
            self.scale.copy_(sc)
            self.deq.copy_(1.0 / sc)
            FP8_SCALES.note_nonfinite(~torch.isfinite(amax))
            self.calibrated = True

    def emit(self):
        """(scale, amax) for a producing launch -- only once this slot has a scale of its own"""
        return (self.scale, self.amax) if self.calibrated else None


class Fp8Scales:
    """Per-tensor delayed scaling, entirely on the device: slot = (scale, 1/scale, 64 words of running max|x|).  A quantising
    launch (the quantise kernel, or the epilogue of the convolution / RoIAlign launch that produces the tensor) reads its slot's
    scale and max-es the tensor's magnitude into the slot's words (atomics spread by block); ``roll()`` (once per optimizer
    step) turns the recorded maxima into the next step's scales.  No host round trip anywhere; first use runs with scale 1."""

    CAP, W = 512, 66

    def __init__(self):
        self.buf, self.used = None, 0
        self.nonfinite = None     # device flag [1]: some quantised tensor held an Inf / NaN (the e4m3 copy itself saturates at +-448,
        #                           so the loss may stay finite: engine._write_metrics reads this flag next to its NaN / Inf check)

    def note_nonfinite(self, bad):
        if self.nonfinite is None or self.nonfinite.device != bad.device:
            self.nonfinite = torch.zeros(1, device=bad.device, dtype=torch.bool)
        self.nonfinite |= bad.view(-1).any()

    def slot(self, device):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:       # ("cuda" names the current device: compare like with like)
            device = torch.device("cuda", torch.cuda.current_device())
        if self.buf is None or self.buf.device != device:
            self.buf = torch.zeros(self.CAP, self.W, device=device, dtype=torch.float32)
            self.buf[:, :2] = 1.0
            self.used = 0
        assert self.used < self.CAP
        row = self.buf[self.used]
        self.used += 1
        return Fp8Slot(row[0:1], row[1:2], row[2:])   # scale [1], dequantisation factor [1], amax words [64]

    def roll(self, margin=1.0):
        if self.buf is None or self.used == 0:
            return
        b = self.buf[: self.used]
        amax = b[:, 2:].amax(dim=1)                  # (NaN propagates through amax; the kernels record Inf / NaN as such: csrc/common.h absmax_bits)
        fin = torch.isfinite(amax)
        self.note_nonfinite(~fin)
        new = torch.where(fin & (amax > 0), (hip.FP8_MAX * margin) / amax.clamp(1e-30, 3.0e38), b[:, 0])   # a non-finite maximum keeps the previous scale
        b[:, 0] = new
        b[:, 1] = 1.0 / new
        b[:, 2:] = 0.0


FP8_SCALES = Fp8Scales()


def fp8_weight(pw):
    """PreparedWeight -> (e4m3 forward weights [Cout,KH,KW,Cin] uint8, dequantisation factor [1] f32), per-tensor scale from the
    weight's own current maximum; cached like the bf16 copies (refreshed after every optimizer step / load)."""
    key = (-1 if pw.frozen else _STEP[0], pw.param.data_ptr(), pw.param._version, _LOAD_GEN[0])
    c = getattr(pw, "_fp8", None)
    if c is None or c[0] != key:
        w = _ohwi(pw.param.detach()).contiguous()
        amax = w.abs().amax().clamp_min(1e-20).view(1)
        w8 = hip.quantize_fp8(w, hip.FP8_MAX / amax)
        pw._fp8 = c = (key, w8, amax / hip.FP8_MAX)
    return c[1], c[2]


def fp8_act_slot(pw):
    """the delayed-scaling slot of the activation tensor that feeds this weight's convolution"""
    sl = getattr(pw, "_fp8_slot", None)
    if sl is None or sl.scale.device != pw.param.device:
        pw._fp8_slot = sl = FP8_SCALES.slot(pw.param.device)
    return sl


def _fp8_eligible(pw, in_shape, pad=0):
    Cout, KH, KW, Cin = _ohwi(pw.param).shape
    N, H, W, _ = in_shape
    M = N * (H + 2 * pad - KH + 1) * (W + 2 * pad - KW + 1)
    return hip.conv_fwd_fp8_ok(M, Cin, Cout, KH, KW, pad)


def fp8_emit_for(consumer_pw, out_shape, pad, fp8, producer_cout=None, producer_taps=1):
    """(scale, amax) of the slot of the convolution that will read a tensor of ``out_shape`` through ``consumer_pw`` -- handed to
    the PRODUCING launch so that its epilogue writes the e4m3 copy -- or None (consumer not on the e4m3 kernel, or the producer's
    launch cannot carry a second output)."""
    if not fp8 or consumer_pw is None or not _fp8_eligible(consumer_pw, out_shape, pad):
        return None
    M = out_shape[0] * out_shape[1] * out_shape[2]
    if producer_cout is not None and not hip.conv_emit8_ok(M, producer_cout, producer_taps, 1):
        return None
    return fp8_act_slot(consumer_pw).emit()


def fp8_copy(t, slot):
    """the e4m3 copy of ``t`` under ``slot``'s scale: the one its producer wrote (``t._fp8``), else one quantisation pass -- whose
    result stays attached to ``t`` for the tensor's other consumers (a gradient feeds an input-gradient AND a weight-gradient GEMM)"""
    made = getattr(t, "_fp8", None)
    if made is not None and made[1] == slot.scale.data_ptr():
        return made[0]
    slot.calibrate(t)
    t8 = hip.quantize_fp8(t, slot.scale, slot.amax)
    t._fp8 = (t8, slot.scale.data_ptr())
    return t8


def wgrad_auto(x, dy, pw, scale, pad, out, fp8=False, x8=None, dy_slot=None, min_m=None, taps_only=True):
    """Weight gradient of a stride-1 convolution: ``hip.conv_wgrad`` on the bf16 / f32 tensors, or -- fp8 configuration, shapes
    ``hip.conv_wgrad_fp8_ok`` takes -- the e4m3 kernel on the activation's copy kept from the forward pass (``x8``, made under
    ``fp8_act_slot(pw)``) and the gradient's copy under ``dy_slot`` (the one its input-gradient convolution reads); both
    dequantisation factors ride in the per-channel scale."""
    Cout, KH, KW, Cin = _ohwi(pw.param).shape
    if fp8 and x8 is not None and dy_slot is not None and dy.dtype == torch.bfloat16 \
            and hip.conv_wgrad_fp8_ok(dy.numel() // Cout, Cin, Cout, KH, KW, pad, min_m, taps_only):
        d8 = fp8_copy(dy, dy_slot)
        deq = fp8_act_slot(pw).deq * dy_slot.deq
        eff = deq * scale if scale is not None else deq.expand(Cout).contiguous()
        return hip.conv_wgrad_fp8(x8, d8, (Cout, KH, KW, Cin), eff, pad=pad, out=out)
    return hip.conv_wgrad(x, dy, (Cout, KH, KW, Cin), scale, pad=pad, out=out)


def _fp8_wgrad1x1_wanted(pw, M, fp8):
    """fp8 configuration: does the weight gradient of the 1x1 convolution ``pw`` over M rows take the e4m3 kernel (so that the
    launches producing its two operands should write their e4m3 copies)?"""
    if not fp8 or os.environ.get("CDDMSL_FP8_WGRAD_1X1", "1") == "0":
        return False
    Cout, KH, KW, Cin = _ohwi(pw.param).shape
    return KH == 1 and KW == 1 and hip.conv_wgrad_fp8_ok(M, Cin, Cout, 1, 1, 0, None, False)


def _fp8_made_for(t, pw):
    """the e4m3 copy of ``t`` made under the activation slot of ``pw``'s convolution (by its producer or by ``conv_fwd_auto``), or None"""
    made = getattr(t, "_fp8", None)
    sl = getattr(pw, "_fp8_slot", None)
    return made[0] if made is not None and sl is not None and made[1] == sl.scale.data_ptr() else None


def conv_fwd_auto(x, pw, scale=None, bias=None, fp8=False, emit8=None, **kw):
    """``hip.conv_fwd`` on the prepared bf16 / f32 weights, or -- with ``fp8`` on shapes the e4m3 kernel takes and wins on
    (``hip.conv_fwd_fp8_ok``: MFMA-bound in bf16) -- the e4m3 kernel on the input's e4m3 copy: the one its producer wrote
    (``x._fp8``, made with this convolution's slot) or, failing that, one quantisation pass; both dequantisation factors ride in
    the epilogue's per-channel scale.  ``emit8``: (scale, amax) for an e4m3 copy of the OUTPUT (``fp8_emit_for``).
    kw: residual, relu, relu_mask, pad."""
    T = x.dtype
    if fp8 and T == torch.bfloat16 and kw.get("stride", 1) == 1 and not kw.get("out_f32", False) and kw.get("out_spec") is None:
        Cout = _ohwi(pw.param).shape[0]
        pad = kw.get("pad", 0)
        if _fp8_eligible(pw, x.shape, pad):
            sl = fp8_act_slot(pw)
            x8 = fp8_copy(x, sl)
            w8, d_w = fp8_weight(pw)
            eff = (sl.deq * d_w) * scale if scale is not None else (sl.deq * d_w).expand(Cout).contiguous()
            return hip.conv_fwd_fp8(x8, w8, eff, bias, kw.get("residual"), kw.get("relu", False), kw.get("relu_mask"), pad, emit8=emit8)
    wf, _ = pw.get(T, need_dgrad=False)
    if emit8 is not None:
        Cout, KH, KW, Cin = _ohwi(pw.param).shape
        N, H, W, _ = x.shape
        pad = kw.get("pad", 0)
        M = N * (H + 2 * pad - KH + 1) * (W + 2 * pad - KW + 1)
        if not (T == torch.bfloat16 and hip.conv_emit8_ok(M, Cout, KH, KW, (Cin * 2 // 16) % 8 == 0) and kw.get("stride", 1) == 1):
            emit8 = None
    return hip.conv_fwd(x, wf, scale, bias, emit8=emit8, **kw)


def fp8_weight_d(pw, wd):
    """e4m3 copy of the prepared input-gradient weights ``wd`` ([Cin, KH, KW, Cout]: flipped, transposed, FrozenBN-scaled) and
    its dequantisation factor; cached per step like ``fp8_weight``"""
    key = (-1 if pw.frozen else _STEP[0], wd.data_ptr(), pw.param._version, _LOAD_GEN[0])
    c = getattr(pw, "_fp8d", None)
    if c is None or c[0] != key:
        amax = wd.abs().amax().float().clamp_min(1e-20).view(1)
        pw._fp8d = c = (key, hip.quantize_fp8(wd, hip.FP8_MAX / amax), amax / hip.FP8_MAX)
    return c[1], c[2]


def fp8_slot_of(owner, name):
    """a delayed-scaling slot attached to ``owner`` (a PreparedWeight or BlockParams) under ``name``"""
    sl = getattr(owner, name, None)
    dev = (owner.param if hasattr(owner, "param") else owner.w[0]).device
    if sl is None or sl.scale.device != dev:
        sl = FP8_SCALES.slot(dev)
        setattr(owner, name, sl)
    return sl


def dgrad_auto(g, pw, fp8=False, slot=None, emit8=None, **kw):
    """Input gradient of a convolution = ``hip.conv_fwd`` of the output gradient with the prepared (flipped / transposed /
    FrozenBN-scaled) weights; with ``fp8`` on shapes the e4m3 kernel takes and wins on, both operands in e4m3: the gradient's
    copy comes from its producer (``g._fp8``, made with ``slot``) or one quantisation pass, the weights' from ``fp8_weight_d``.
    ``slot``: the gradient tensor's delayed-scaling slot (shared by all its consumers).  kw: pad, residual, relu_mask,
    residual_pooled."""
    T = g.dtype
    _, wd = pw.get(T, True)
    if fp8 and T == torch.bfloat16 and slot is not None and not kw.get("residual_pooled", False):
        Cout_d, KH, KW, Cin_d = wd.shape
        pad = kw.get("pad", 0)
        N, H, W, _ = g.shape
        M = N * (H + 2 * pad - KH + 1) * (W + 2 * pad - KW + 1)
        if hip.conv_fwd_fp8_ok(M, Cin_d, Cout_d, KH, KW, pad):
            g8 = fp8_copy(g, slot)
            wd8, d_w = fp8_weight_d(pw, wd)
            return hip.conv_fwd_fp8(g8, wd8, (slot.deq * d_w).expand(Cout_d).contiguous(), None, kw.get("residual"), False, kw.get("relu_mask"),
                                    pad, emit8=emit8)
    if emit8 is not None:
        Cout_d, KH, KW, Cin_d = wd.shape
        N, H, W, _ = g.shape
        pad = kw.get("pad", 0)
        M = N * (H + 2 * pad - KH + 1) * (W + 2 * pad - KW + 1)
        if not (T == torch.bfloat16 and hip.conv_emit8_ok(M, Cout_d, KH, KW, (Cin_d * 2 // 16) % 8 == 0) and not kw.get("residual_pooled", False)):
            emit8 = None
    return hip.conv_fwd(g, wd, emit8=emit8, **kw)


def _fp8_dgrad_wanted(pw, g_shape, pad, fp8):
    """would ``dgrad_auto`` run the e4m3 kernel for a gradient of ``g_shape`` through ``pw``?"""
    if not fp8:
        return False
    Cout, KH, KW, Cin = _ohwi(pw.param).shape          # the input-gradient conv maps Cout -> Cin channels
    N, H, W, _ = g_shape
    M = N * (H + 2 * pad - KH + 1) * (W + 2 * pad - KW + 1)
    return hip.conv_fwd_fp8_ok(M, Cout, Cin, KH, KW, pad)


def cat_prepared(weights, dtype):
    """Concatenate several [Ni,K] / 1x1 weights into one [sum Ni,1,1,K] GEMM operand (fused heads)."""
    return torch.cat([hip.weight_prep(_ohwi(w.detach()), None, dtype, True, False)[0] for w in weights], dim=0)


# ------------------------------------------------------------------------------------------------
# generic conv / linear with bias (+ReLU)   -- RPN head, projections, projector, mapper linears
# ------------------------------------------------------------------------------------------------
class ConvFn(torch.autograd.Function):
    """y = relu?(conv(x, W) + b).  x NHWC T; W f32 master; optional f32 output.  ``train_w`` False = frozen
    weights (input gradient only)."""

    @staticmethod
    def forward(ctx, x, anchor, pw, bias, stride, pad, relu, out_f32, train_w, residual=None, fp8=False):
        y = conv_fwd_auto(x, pw, None, None if bias is None else bias.detach(), fp8, residual=None if residual is None else residual.detach(),
                          relu=relu, stride=stride, pad=pad, out_f32=out_f32)
        ctx.pw, ctx.bias, ctx.cfg = pw, bias, (stride, pad, relu, out_f32, train_w)
        # fp8 configuration: where the forward ran on the e4m3 kernel, its input copy is kept for the weight gradient
        ctx.fp8 = bool(fp8)
        ctx.save_for_backward(x, y if relu else None, _fp8_made_for(x, pw) if fp8 else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, x8 = ctx.saved_tensors
        stride, pad, relu, out_f32, train_w = ctx.cfg
        # (fp8 configuration, only where the forward took the e4m3 kernel: one e4m3 copy of the gradient feeds both gradient GEMMs)
        f8 = ctx.fp8 and x8 is not None and stride == 1
        g_slot = fp8_slot_of(ctx.pw, "_fp8_g") if f8 else None
        T = x.dtype
        dy_in = dy
        dy = dy.contiguous()
        if relu:
            dy = hip.relu_bwd(dy, y if y.dtype == T else y.to(T))
        elif dy.dtype != T:
            dy = dy.to(T)
        w = ctx.pw.param
        if train_w:
            if f8:
                wgrad_auto(x, dy, ctx.pw, None, pad, _ohwi(_grad_buf(w)), True, x8, g_slot, min_m=65536)
            else:
                hip.conv_wgrad(x, dy, _ohwi(w).shape, None, stride=stride, pad=pad, out=_ohwi(_grad_buf(w)))
            if ctx.bias is not None:
                hip.colsum(dy.view(-1, dy.shape[-1]), out=_grad_buf(ctx.bias))
        dx = None
        if ctx.needs_input_grad[0]:
            assert stride == 1, "input gradient of a strided conv is not on the hot path"
            _, wd = ctx.pw.get(T, need_dgrad=True)
            KH = _ohwi(w).shape[1]
            Kd, rows = dy.shape[-1], dy.numel() // dy.shape[-1]
            if KH == 1 and _ohwi(w).shape[2] == 1 and pad == 0 and rows <= 1024 and Kd >= 8192 and Kd % (16 * 64) == 0:
                # few rows, very long reduction (the mapper's 1024 -> 40*768 linear: 544 x 1024 outputs over K = 30720 would be
                # 12 tiles on 256 CUs): 16 reduction slices as one batched launch, f32 partial sums added afterwards
                S, N = 16, x.shape[-1]
                part = torch.empty(S, rows, N, device=dy.device, dtype=torch.float32)
                hip.gemm_nt_batched(dy, wd, part, rows, N, Kd // S, Kd, Kd, N, S, Kd // S, Kd // S, rows * N)
                dx = part.sum(0).to(T).view(x.shape)
            elif f8:
                dx = dgrad_auto(dy, ctx.pw, True, g_slot, pad=KH - 1 - pad)
            else:
                dx = hip.conv_fwd(dy, wd, stride=1, pad=KH - 1 - pad)
        # y = conv + residual (no ReLU with a residual): the residual's gradient is the incoming one, as it came
        return dx, None, None, None, None, None, None, None, None, (dy_in if ctx.needs_input_grad[9] else None), None


class FrozenMlpFn(torch.autograd.Function):
    """y (f32) = residual + W2 relu(W1 x + b1) + b2 with frozen weights (the mapper's MlpTransformer, clipcap.py:39-56) as ONE
    node: the ReLU backward rides in the epilogue of fc2's input-gradient GEMM (relu_mask = the saved hidden activations)
    instead of a pass of its own over the [rows, hidden] tensor."""

    @staticmethod
    def forward(ctx, x, pw1, b1, pw2, b2, residual):
        T = x.dtype
        w1, _ = pw1.get(T, need_dgrad=False)
        w2, _ = pw2.get(T, need_dgrad=False)
        M, K = x.shape
        h = hip.conv_fwd(x.view(1, 1, M, K), w1, None, b1.detach(), relu=True)
        y = hip.conv_fwd(h, w2, None, b2.detach(), residual.detach().view(1, 1, M, -1), out_f32=True)
        ctx.pw = (pw1, pw2)
        ctx.save_for_backward(h)
        return y.view(M, -1)

    @staticmethod
    def backward(ctx, dy):
        (h,) = ctx.saved_tensors
        T = h.dtype
        M = h.shape[2]
        g = dy.contiguous()
        if g.dtype != T:
            g = g.to(T)
        _, w2d = ctx.pw[1].get(T, need_dgrad=True)
        _, w1d = ctx.pw[0].get(T, need_dgrad=True)
        dh = hip.conv_fwd(g.view(1, 1, M, -1), w2d, relu_mask=h)
        dx = hip.conv_fwd(dh, w1d).view(M, -1) if ctx.needs_input_grad[0] else None
        return dx, None, None, None, None, (dy if ctx.needs_input_grad[5] else None)


def frozen_mlp(x2d, pw1, b1, pw2, b2, residual):
    return FrozenMlpFn.apply(x2d.contiguous(), pw1, b1, pw2, b2, residual.contiguous())


def conv(x, pw, bias=None, stride=1, pad=0, relu=False, out_f32=False, train_w=True, residual=None, fp8=False):
    """``residual`` (same shape as the output; f32 with ``out_f32`` on the bf16 path) is added in the GEMM epilogue.
    ``fp8``: forward on e4m3 operands where the shape qualifies (``conv_fwd_auto``), and then both gradient GEMMs where theirs do
    (``wgrad_auto`` / ``dgrad_auto``: the RPN's 3x3 convolution)."""
    anchor = pw.param if train_w else None
    assert residual is None or not relu
    return ConvFn.apply(x, anchor, pw, bias, stride, pad, relu, out_f32, train_w, residual, fp8)


def linear(x2d, pw, bias=None, relu=False, out_f32=False, train_w=True, residual=None):
    M, K = x2d.shape
    y = conv(x2d.contiguous().view(1, 1, M, K), pw, bias, 1, 0, relu, out_f32, train_w,
             None if residual is None else residual.contiguous().view(1, 1, M, -1))
    return y.view(M, -1)


class FusedHeadsFn(torch.autograd.Function):
    """Several 1x1 heads over the same input as ONE GEMM (N = sum of head widths, zero-padded to a multiple of 8 so
    every row of y / dy is whole 16-byte chunks): the RPN's objectness (A) + anchor-delta (4A) heads, rpn.py:173-176."""

    @staticmethod
    def forward(ctx, x, anchor, heads, out_f32):
        T = x.dtype
        K = heads[0][0].shape[1] if heads[0][0].dim() == 2 else heads[0][0].shape[1]
        ws = [w.detach().reshape(w.shape[0], -1) for w, _ in heads]
        n = sum(w.shape[0] for w in ws)
        npad = (n + 7) // 8 * 8
        wcat = torch.zeros(npad, 1, 1, ws[0].shape[1], device=x.device, dtype=torch.float32)
        wcat[:n, 0, 0] = torch.cat(ws, dim=0)
        bcat = torch.zeros(npad, device=x.device, dtype=torch.float32)
        bcat[:n] = torch.cat([b.detach() for _, b in heads])
        wf, wd = hip.weight_prep(wcat, None, T, True, True)
        y = hip.conv_fwd(x, wf, None, bcat, out_f32=out_f32)
        ctx.heads, ctx.wd, ctx.npad = heads, wd, npad
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        T = x.dtype
        dy = dy.contiguous()
        if dy.dtype != T:
            dy = dy.to(T)
        K = x.shape[-1]
        dw = hip.conv_wgrad(x, dy, (ctx.npad, 1, 1, K))
        db = hip.colsum(dy.view(-1, ctx.npad))
        off = 0
        for w, b in ctx.heads:
            n = w.shape[0]
            _grad_buf(w).view(n, -1).add_(dw[off:off + n].view(n, -1)) if w.dim() == 2 else \
                _ohwi(_grad_buf(w)).add_(dw[off:off + n])
            _grad_buf(b).add_(db[off:off + n])
            off += n
        dx = hip.conv_fwd(dy, ctx.wd) if ctx.needs_input_grad[0] else None
        return dx, None, None, None


def fused_heads(x, heads, out_f32=True):
    """heads: list of (weight [Ni,K,1,1] or [Ni,K], bias [Ni]).  Returns y [..., Npad]; slice per head."""
    return FusedHeadsFn.apply(x, heads[0][0], heads, out_f32)


# ------------------------------------------------------------------------------------------------
# CLIP residual stage: a chain of Bottlenecks as ONE autograd node (fused FrozenBN/ReLU/avgpool/residual)
# ------------------------------------------------------------------------------------------------
class BlockParams:
    """Weights + folded FrozenBN affine of one Bottleneck (clip_backbone.py:14-70)."""

    def __init__(self, w1, w2, w3, wd, bn1, bn2, bn3, bnd, stride, frozen, fp8=False):
        self.stride, self.frozen, self.fp8 = stride, frozen, fp8
        self.bn = (bn1, bn2, bn3, bnd)  # each = (scale, bias) f32
        self.w = (w1, w2, w3, wd)
        self.pw = tuple(None if w is None else PreparedWeight(w, b[0], frozen) for w, b in zip(self.w, self.bn))


def _block_forward(x, bp, save, px_given=None, next_pw=None, out_spec=None):
    # (next_pw: conv1 of the block that reads this block's output -- fp8 configuration: its e4m3 copy is written here)
    """Bottleneck forward (clip_backbone.py:57-70).  AvgPool2d(stride) runs as its own HBM-bound kernel: fusing it into
    the GEMM A-loader (kernel option pool=1, kept and tested) halves the MFMA rate of this kernel structure."""
    (s1, b1), (s2, b2), (s3, b3), bnd = bp.bn
    pool = bp.stride > 1
    f8 = bp.fp8
    o1_shape = (x.shape[0], x.shape[1], x.shape[2], _ohwi(bp.w[0]).shape[0])
    o1 = conv_fwd_auto(x, bp.pw[0], s1, b1, f8, emit8=fp8_emit_for(bp.pw[1], o1_shape, 1, f8), relu=True)     # (conv2 reads o1's e4m3 copy)
    x8 = _fp8_made_for(x, bp.pw[0]) if f8 and save else None                                # (conv1's e4m3 input copy, for its weight gradient)
    M2 = x.shape[0] * x.shape[1] * x.shape[2]
    # conv3's weight gradient on e4m3 operands needs o2's copy: written by conv2's epilogue under conv3's activation slot
    e8o2 = fp8_act_slot(bp.pw[2]).emit() if save and not pool and _fp8_wgrad1x1_wanted(bp.pw[2], M2, f8) else None
    o2 = conv_fwd_auto(o1, bp.pw[1], s2, b2, f8, emit8=e8o2, relu=True, pad=1)
    o2_8 = _fp8_made_for(o2, bp.pw[2]) if e8o2 is not None else None
    p2 = hip.avgpool2_fwd(o2) if pool else o2
    if pool and px_given is not None and tuple(px_given.shape) == (x.shape[0], x.shape[1] // 2, x.shape[2] // 2, x.shape[3]):
        px = px_given                              # the producer of x pooled it on the way (roi_align with_pooled)
    else:
        px = hip.avgpool2_fwd(x) if pool else x
    if bp.pw[3] is not None:
        idn = conv_fwd_auto(px, bp.pw[3], bnd[0], bnd[1], f8)
    else:
        idn = x
    out_shape = (p2.shape[0], p2.shape[1], p2.shape[2], _ohwi(bp.w[2]).shape[0])
    out = conv_fwd_auto(p2, bp.pw[2], s3, b3, f8, emit8=fp8_emit_for(next_pw, out_shape, 0, f8), residual=idn, relu=True, out_spec=out_spec)
    return out, ((o1, o2, p2 if pool else None, px if pool else None, (x8, _fp8_made_for(o1, bp.pw[1]) if f8 else None, o2_8)) if save else None)


def _block_backward(gs, x, o1, o2, p2, px, bp, need_dx, mask_x, prev_bp=None, c8=(None, None, None)):
    """gs = dL/d(pre-ReLU sum) of this block (already masked by out>0).  Returns dL/dx, masked by x>0 when
    ``mask_x`` (x is the previous block's post-ReLU output) so it is directly the previous block's ``gs``.
    fp8 configuration: the input-gradient convolutions with a long reduction (conv3's, conv2's, the downsample conv's) run on
    e4m3 operands (``dgrad_auto``); ``prev_bp`` = the block that receives the returned gradient as ITS ``gs`` (its e4m3 copy is
    then written by this block's last launch)."""
    T = x.dtype
    (s1, _), (s2, _), (s3, _), bnd = bp.bn
    pool = bp.stride > 1
    f8 = bp.fp8 and T == torch.bfloat16
    w1p, w2p, w3p, wdp = bp.w
    shp = lambda w: _ohwi(w).shape
    gs_slot = fp8_slot_of(bp, "_fp8_gs") if f8 else None
    d2_slot = fp8_slot_of(bp.pw[1], "_fp8_g") if f8 else None
    x8, o1_8, o2_8 = c8                               # e4m3 copies kept from the forward pass (fp8 configuration): inputs of conv1, conv2, conv3
    M2 = o2.shape[0] * o2.shape[1] * o2.shape[2]
    if o2_8 is not None and not pool:
        wgrad_auto(o2, gs, bp.pw[2], s3, 0, _ohwi(_grad_buf(w3p)), f8, o2_8, gs_slot, taps_only=False)
    else:
        hip.conv_wgrad(p2 if pool else o2, gs, shp(w3p), s3, out=_ohwi(_grad_buf(w3p)))
    d1_slot = fp8_slot_of(bp.pw[0], "_fp8_g") if f8 and x8 is not None and _fp8_wgrad1x1_wanted(bp.pw[0], M2, f8) else None
    if pool:
        dp2 = dgrad_auto(gs, bp.pw[2], f8, gs_slot)
        dpre2 = hip.avgpool2_bwd(dp2, tuple(o2.shape), mask=o2, emit8=d2_slot.emit() if f8 and _fp8_dgrad_wanted(bp.pw[1], o2.shape, 1, f8) else None)
    else:
        e8 = d2_slot.emit() if f8 and _fp8_dgrad_wanted(bp.pw[1], o2.shape, 1, f8) else None
        dpre2 = dgrad_auto(gs, bp.pw[2], f8, gs_slot, emit8=e8, relu_mask=o2)
    # (conv2's input gradient first: it leaves the e4m3 copy of dpre2 attached, which the fp8 weight gradient reads as well)
    dpre1 = dgrad_auto(dpre2, bp.pw[1], f8, d2_slot, pad=1, relu_mask=o1, emit8=d1_slot.emit() if d1_slot is not None else None)
    wgrad_auto(o1, dpre2, bp.pw[1], s2, 1, _ohwi(_grad_buf(w2p)), f8, o1_8, d2_slot)
    if d1_slot is not None:
        wgrad_auto(x, dpre1, bp.pw[0], s1, 0, _ohwi(_grad_buf(w1p)), f8, x8, d1_slot, taps_only=False)
    else:
        hip.conv_wgrad(x, dpre1, shp(w1p), s1, out=_ohwi(_grad_buf(w1p)))
    if wdp is not None:
        hip.conv_wgrad(px if pool else x, gs, shp(wdp), bnd[0], out=_ohwi(_grad_buf(wdp)))
    if not need_dx:
        return None
    pooled = False
    if wdp is not None:
        dxb = dgrad_auto(gs, bp.pw[3], f8, gs_slot)
        if pool:
            # the downsample path's input gradient stays at pooled resolution: conv1's dgrad epilogue adds a quarter of each
            # row's pooled pixel (AvgPool2d backward fused; saves writing and re-reading the full-resolution tensor)
            pooled = x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0 and hip.pooled_residual_ok(dxb)
            if not pooled:
                dxb = hip.avgpool2_bwd(dxb, tuple(x.shape))
    else:
        dxb = gs
    e8 = None
    if f8 and prev_bp is not None and prev_bp.fp8 and not pooled and _fp8_dgrad_wanted(prev_bp.pw[2], x.shape, 0, True):
        e8 = fp8_slot_of(prev_bp, "_fp8_gs").emit()  # the returned gradient is prev_bp's gs: write its e4m3 copy here
    return dgrad_auto(dpre1, bp.pw[0], False, None, emit8=e8, residual=dxb, relu_mask=x if mask_x else None, residual_pooled=pooled)


class ResStageFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, blocks, out_grad_premasked=False, px0=None, out_spec=None):
        ctx.out_grad_premasked = out_grad_premasked
        saved, o1_8s = [x], []
        cur = x
        for bi, bp in enumerate(blocks):
            cur, mids = _block_forward(cur, bp, True, px0 if bi == 0 else None, blocks[bi + 1].pw[0] if bi + 1 < len(blocks) else None,
                                       out_spec if bi + 1 == len(blocks) else None)
            saved += [mids[0], mids[1], mids[2], mids[3], cur]
            o1_8s += list(mids[4])
        ctx.blocks = blocks
        ctx.save_for_backward(*saved, *o1_8s)         # (fp8 configuration: three e4m3 input copies per block, for the weight gradients)
        return cur

    @staticmethod
    def backward(ctx, g):
        blocks = ctx.blocks
        saved, o1_8s = ctx.saved_tensors[:-3 * len(blocks)], ctx.saved_tensors[-3 * len(blocks):]
        need_dx = ctx.needs_input_grad[0]
        # mask by the stage output's ReLU -- unless the one consumer of the output has done it (res_stage_attnpool)
        gs = g.contiguous() if ctx.out_grad_premasked else hip.relu_bwd(g.contiguous(), saved[-1])
        for i in range(len(blocks) - 1, -1, -1):
            x, o1, o2, p2, px = saved[5 * i: 5 * i + 5]
            gs = _block_backward(gs, x, o1, o2, p2, px, blocks[i], need_dx or i > 0, mask_x=i > 0, prev_bp=blocks[i - 1] if i > 0 else None,
                                 c8=tuple(o1_8s[3 * i: 3 * i + 3]))
        return gs, None, None, None, None, None


def res_stage(x, blocks, frozen, out_grad_premasked=False, out_spec=None):
    """Runs a residual stage.  Frozen stages (FREEZE_AT) and no-grad calls keep no activations.  ``out_spec`` (hip.OutSpec): the
    stage output is written into one half of a buffer shared with a second pass (no concatenation copy later)."""
    px0 = getattr(x, "_pooled2", None)               # 2x2-pooled copy of x supplied by its producer (roi_align)
    if frozen or not torch.is_grad_enabled():
        cur = x
        for bi, bp in enumerate(blocks):
            cur, _ = _block_forward(cur, bp, False, px0 if bi == 0 else None, blocks[bi + 1].pw[0] if bi + 1 < len(blocks) else None,
                                    out_spec if bi + 1 == len(blocks) else None)
        return cur
    return ResStageFn.apply(x, blocks[0].w[0], blocks, out_grad_premasked, px0, out_spec)


class StackHalvesFn(torch.autograd.Function):
    """``torch.cat([a, b])`` for two tensors that already ARE the two halves of ``full`` (hip.OutSpec): no copy forward, two
    views of the gradient backward."""

    @staticmethod
    def forward(ctx, a, b, full):
        n = a.shape[0]
        assert a.data_ptr() == full.data_ptr() and b.data_ptr() == full[n:].data_ptr() and full.shape[0] == 2 * n and a.shape == b.shape
        ctx.n = n
        return full.view(full.shape)

    @staticmethod
    def backward(ctx, g):
        return g[:ctx.n], g[ctx.n:], None


def stack_halves(a, b, full):
    return StackHalvesFn.apply(a, b, full)


def res_stage_attnpool(x, blocks, frozen, ap):
    """layer4 -> AttentionPool2d (clip_roi_heads.py:160-165) as one composition: the pool's backward writes its input gradient
    already masked by the stage's output ReLU (the map it pooled IS that output), so the stage skips its own relu_bwd pass
    over the [K,7,7,2048] tensor.  Only valid because nothing else consumes the stage output -- which this function owns."""
    fuse = not frozen and torch.is_grad_enabled()
    y = res_stage(x, blocks, frozen, out_grad_premasked=fuse)
    return AttnPoolFn.apply(y, None if ap.frozen else ap.q_w, ap, fuse)


# ------------------------------------------------------------------------------------------------
# RoIAlign -> CLIP layer4 with the first block's conv1 moved in front of the pooling
# ------------------------------------------------------------------------------------------------
def _roi_block0_forward(feat, rois, bp, out_size, scale, sr, extra, next_pw=None):
    """First Bottleneck of the RoI head's layer4 (clip_roi_heads.py:113-115 -> clip_backbone.py:57-70) on the pooled crops, WITHOUT
    the crops: RoIAlign is a linear map over pixels and conv1 / the downsample conv are 1x1 (linear over channels), so

        relu(bn1(conv1(roi_align(x))))    = relu(s1 * roi_align(conv1(x)) + b1)          (conv1 runs once per image pixel: 66 400
        avgpool2(roi_align(x))  (-> downsample conv)  written directly, pooled           rows instead of 1.6 M crop rows at 8192 RoIs)

    The [K,14,14,1024] crop tensor (3.3 GB) is never written or read; the 512-channel o1 (half of it) is what conv2 needs
    anyway.  Returns (o1, o2, p2, px, out, e4m3 copy of o1 or None)."""
    T = feat.dtype
    (s1, b1), (s2, b2), (s3, b3), bnd = bp.bn
    w1, _ = bp.pw[0].get(T, False)
    f8 = bp.fp8
    K, E = rois.shape[0], (0 if extra is None else extra.shape[0])
    z = hip.conv_fwd(feat, w1)                                                              # conv1 on the feature map, no affine yet
    o1_shape = (K + E, out_size, out_size, z.shape[-1])
    e8 = fp8_emit_for(bp.pw[1], o1_shape, 1, f8) if E == 0 else None                       # (appended maps are copied in afterwards: quantise then)
    o1 = hip.roi_align_forward_affine(z, rois, out_size, out_size, scale, sr, True, s1, b1, relu=True, extra_rows=E, emit8=e8)
    px = hip.roi_align_forward_affine(feat, rois, out_size, out_size, scale, sr, True, pooled_only=True, extra_rows=E)
    if E:                                            # maps of the crops' geometry riding behind them (the 224x224 crops' res4)
        o1[K:].copy_(hip.conv_fwd(extra, w1, s1, b1, relu=True))
        px[K:].copy_(hip.avgpool2_fwd(extra))
    o2 = conv_fwd_auto(o1, bp.pw[1], s2, b2, f8, relu=True, pad=1)
    p2 = hip.avgpool2_fwd(o2)
    idn = conv_fwd_auto(px, bp.pw[3], bnd[0], bnd[1], f8)
    out_shape = (p2.shape[0], p2.shape[1], p2.shape[2], _ohwi(bp.w[2]).shape[0])
    out = conv_fwd_auto(p2, bp.pw[2], s3, b3, f8, emit8=fp8_emit_for(next_pw, out_shape, 0, f8), residual=idn, relu=True)
    return o1, o2, p2, px, out, (_fp8_made_for(o1, bp.pw[1]) if f8 else None)


class RoIStageFn(torch.autograd.Function):
    """pooler (poolers.py:190-229) + ``backbone.layer4`` (clip_roi_heads.py:113-115) as ONE autograd node, first block as in
    ``_roi_block0_forward``.  Backward of that block: the gradient of o1 goes back through the RoIAlign gather at 512 channels
    to the feature-map-level conv1 (its weight gradient is a 66 400-row reduction, its input gradient a 66 400-row GEMM), the
    downsample path's gradient through the gather on the POOLED 7x7 grid (a quarter of the bytes)."""

    @staticmethod
    def forward(ctx, feat, anchor, rois, roi_start, blocks, out_size, scale, sr, out_grad_premasked, extra):
        feat = feat.contiguous()
        nxt = lambda i: blocks[i + 1].pw[0] if i + 1 < len(blocks) else None
        o1, o2, p2, px, cur, o1_8 = _roi_block0_forward(feat, rois, blocks[0], out_size, scale, sr, extra, nxt(0))
        saved, o1_8s = [feat, rois, roi_start, extra, o1, o2, p2, px, cur], [None, o1_8, None]
        for bi, bp in enumerate(blocks[1:], start=1):
            cur, mids = _block_forward(cur, bp, True, None, nxt(bi))
            saved += [mids[0], mids[1], mids[2], mids[3], cur]
            o1_8s += list(mids[4])
        ctx.blocks, ctx.meta = blocks, (out_size, scale, sr, out_grad_premasked)
        ctx.save_for_backward(*saved, *o1_8s)         # (fp8 configuration: three e4m3 input copies per block, for the weight gradients)
        return cur

    @staticmethod
    def backward(ctx, g):
        blocks = ctx.blocks
        saved, o1_8s = ctx.saved_tensors[:-3 * len(blocks)], ctx.saved_tensors[-3 * len(blocks):]
        out_size, scale, sr, premasked = ctx.meta
        feat, rois, roi_start, extra = saved[:4]
        st = saved[4:]                               # per block: o1, o2, p2, px, out
        gs = g.contiguous() if premasked else hip.relu_bwd(g.contiguous(), st[-1])
        for i in range(len(blocks) - 1, 0, -1):
            o1, o2, p2, px = st[5 * i: 5 * i + 4]
            gs = _block_backward(gs, st[5 * i - 1], o1, o2, p2, px, blocks[i], True, mask_x=True, prev_bp=blocks[i - 1], c8=tuple(o1_8s[3 * i: 3 * i + 3]))
        bp = blocks[0]
        o1, o2, p2, px = st[0:4]
        T = o1.dtype
        (s1, _), (s2, _), (s3, _), bnd = bp.bn
        w1p, w2p, w3p, wdp = bp.w
        shp = lambda w: _ohwi(w).shape
        K, E = rois.shape[0], (0 if extra is None else extra.shape[0])
        f8 = bp.fp8 and T == torch.bfloat16
        gs_slot = fp8_slot_of(bp, "_fp8_gs") if f8 else None
        d2_slot = fp8_slot_of(bp.pw[1], "_fp8_g") if f8 else None
        hip.conv_wgrad(p2, gs, shp(w3p), s3, out=_ohwi(_grad_buf(w3p)))
        e8 = d2_slot.emit() if f8 and _fp8_dgrad_wanted(bp.pw[1], o2.shape, 1, f8) else None   # (the e4m3 copy conv2's two gradient GEMMs read)
        dpre2 = hip.avgpool2_bwd(dgrad_auto(gs, bp.pw[2], f8, gs_slot), tuple(o2.shape), mask=o2, emit8=e8)
        dpre1 = dgrad_auto(dpre2, bp.pw[1], f8, d2_slot, pad=1, relu_mask=o1)                # [K+E,14,14,planes] wrt bn1's output
        wgrad_auto(o1, dpre2, bp.pw[1], s2, 1, _ohwi(_grad_buf(w2p)), f8, o1_8s[1], d2_slot)
        hip.conv_wgrad(px, gs, shp(wdp), bnd[0], out=_ohwi(_grad_buf(wdp)))
        dxb = dgrad_auto(gs, bp.pw[3], f8, gs_slot)                                          # [K+E,7,7,C] wrt the pooled crops
        # back across the pooling: gather at `planes` channels, then conv1's gradients on the feature map (s1 rides in the
        # weight-gradient scale and in the prepared input-gradient weights, as for every conv of a stage)
        N, H, W, C = feat.shape
        dz = hip.roi_align_backward(dpre1[:K], rois, roi_start, (N, H, W, dpre1.shape[-1]), scale, sr, True)
        hip.conv_wgrad(feat, dz, shp(w1p), s1, out=_ohwi(_grad_buf(w1p)))
        _, w1d = bp.pw[0].get(T, True)
        dfeat = dextra = None
        if E:
            hip.conv_wgrad(extra, dpre1[K:], shp(w1p), s1, out=_ohwi(_grad_buf(w1p)))
            if ctx.needs_input_grad[9]:
                dextra = hip.conv_fwd(dpre1[K:], w1d, residual=hip.avgpool2_bwd(dxb[K:].contiguous(), tuple(extra.shape)))
        if ctx.needs_input_grad[0]:
            r = hip.roi_align_backward(dxb[:K], rois, roi_start, (N, H, W, C), scale, sr, True, pooled=True)
            dfeat = hip.conv_fwd(dz, w1d, residual=r)
        return dfeat, None, None, None, None, None, None, None, None, dextra


def roi_stage_supported(blocks):
    """the commuted form needs a 1x1 conv1 and a pooled (stride-2) downsample path in the first block: CLIP's layer4"""
    bp = blocks[0]
    return bp.stride == 2 and bp.w[3] is not None and tuple(bp.w[0].shape[2:]) == (1, 1) and tuple(bp.w[3].shape[2:]) == (1, 1)


def roi_stage(feat, rois, roi_start, blocks, frozen, out_size, scale, sr, extra=None, out_grad_premasked=False):
    assert rois.dim() == 2 and rois.size(1) == 5 and out_size % 2 == 0                     # layers/roi_align.py:55
    assert extra is None or tuple(extra.shape[1:]) == (out_size, out_size, feat.shape[3])
    if frozen or not torch.is_grad_enabled():
        nxt = lambda i: blocks[i + 1].pw[0] if i + 1 < len(blocks) else None
        cur = _roi_block0_forward(feat.contiguous(), rois, blocks[0], out_size, scale, sr, extra, nxt(0))[4]
        for bi, bp in enumerate(blocks[1:], start=1):
            cur, _ = _block_forward(cur, bp, False, None, nxt(bi))
        return cur
    return RoIStageFn.apply(feat, blocks[0].w[0], rois, roi_start, blocks, out_size, scale, sr, out_grad_premasked, extra)


def roi_stage_attnpool(feat, rois, roi_start, blocks, frozen, ap, out_size, scale, sr, extra=None):
    """RoIAlign -> layer4 -> AttentionPool2d (clip_roi_heads.py:160-165); see ``res_stage_attnpool`` for the ReLU fusion"""
    fuse = not frozen and torch.is_grad_enabled()
    y = roi_stage(feat, rois, roi_start, blocks, frozen, out_size, scale, sr, extra, out_grad_premasked=fuse)
    return AttnPoolFn.apply(y, None if ap.frozen else ap.q_w, ap, fuse)


# ------------------------------------------------------------------------------------------------
# RoIAlign (layers/roi_align.py:7-65, poolers.py:190-229)
# ------------------------------------------------------------------------------------------------
class RoIAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rois, roi_start, out_size, scale, sampling_ratio, aligned, with_pooled=False, extra=None):
        K, E = rois.shape[0], (0 if extra is None else extra.shape[0])
        out = hip.roi_align_forward(x, rois, out_size, out_size, scale, sampling_ratio, aligned, with_pooled=with_pooled, extra_rows=E)
        ctx.save_for_backward(rois, roi_start)
        ctx.meta = (tuple(x.shape), scale, sampling_ratio, aligned, K)
        y = out[0] if with_pooled else out
        if E:                                        # maps appended behind the crops: the stage that follows runs once over both
            y[K:].copy_(extra)
            if with_pooled:
                out[1][K:].copy_(hip.avgpool2_fwd(extra.contiguous()))
        if with_pooled:
            ctx.mark_non_differentiable(out[1])      # a by-product: the consumer (ResStageFn) differentiates through y alone
        return out

    @staticmethod
    def backward(ctx, dy, *unused):
        rois, roi_start = ctx.saved_tensors
        shape, scale, sr, aligned, K = ctx.meta
        dy = dy.contiguous()
        dx = hip.roi_align_backward(dy[:K], rois, roi_start, shape, scale, sr, aligned)
        dextra = dy[K:] if (dy.shape[0] > K and ctx.needs_input_grad[8]) else None
        return dx, None, None, None, None, None, None, None, dextra


def roi_align(x, rois, roi_start, out_size, scale, sampling_ratio, aligned=True, with_pooled=False, extra=None):
    """``with_pooled`` (even out_size): the result carries its 2x2-average-pooled copy as ``y._pooled2`` -- the first block of
    a stride-2 residual stage picks it up instead of pooling the map again (res_stage).  ``extra`` [E, out, out, C]: maps of
    the crops' geometry appended behind them (rows K..K+E of the result), written into the one output buffer."""
    assert rois.dim() == 2 and rois.size(1) == 5  # layers/roi_align.py:55
    assert extra is None or tuple(extra.shape[1:]) == (out_size, out_size, x.shape[3])
    if with_pooled and out_size % 2 == 0:
        y, yp = RoIAlignFn.apply(x, rois, roi_start, out_size, scale, sampling_ratio, aligned, True, extra)
        y._pooled2 = yp
        return y
    return RoIAlignFn.apply(x, rois, roi_start, out_size, scale, sampling_ratio, aligned, False, extra)


# ------------------------------------------------------------------------------------------------
# AttentionPool2d (clip_backbone.py:73-107), query-0-only
# ------------------------------------------------------------------------------------------------
class AttnPoolParams:
    def __init__(self, pos, q_w, q_b, k_w, k_b, v_w, v_b, c_w, c_b, heads, frozen=False):
        self.pos, self.heads, self.frozen = pos, heads, frozen
        self.q_w, self.q_b, self.k_w, self.k_b, self.v_w, self.v_b, self.c_w, self.c_b = q_w, q_b, k_w, k_b, v_w, v_b, c_w, c_b
        self.pq = PreparedWeight(q_w, None, frozen)
        self.pk = PreparedWeight(k_w, None, frozen)
        self.pv = PreparedWeight(v_w, None, frozen)
        self.pc = PreparedWeight(c_w, None, frozen)


class AttnPoolFn(torch.autograd.Function):
    """Query-0-only attention pool, reassociated so that no per-token K/V projection is ever formed:

        scores[h,t] = q0_h . (Wk_h tok_t + bk_h)  =  (Wk_h^T q0_h) . tok_t  + const_h      (const drops out of softmax)
        out_h       = sum_t p[h,t] (Wv_h tok_t + bv_h)  =  Wv_h (sum_t p[h,t] tok_t) + bv_h

    i.e. u = per-head GEMM [K,64]x[64,C], scores = per-region GEMM [H,C]x[C,T], z = per-region GEMM [H,T]x[T,C],
    o = per-head GEMM [K,C]x[C,64]: ~20x fewer FLOPs than projecting K and V for all 50 tokens (the reference projects
    q, k, v for all tokens and keeps token 0, clip_backbone.py:83-107).  All products run on the batched MFMA GEMMs;
    the 50-wide softmax and its backward are fp32 elementwise glue.  Token rows per region are padded 50 -> 56
    (whole 16-byte chunks for the transposed products); pad rows are zero."""

    TP = 56

    @staticmethod
    def forward(ctx, x, anchor, ap, premask_input_grad=False):
        """x [K,h,w,C] NHWC T with h*w+1 == len(pos)  ->  [K, out_dim] f32.  ``premask_input_grad``: x is a ReLU output whose
        producer expects dL/dx already zeroed where x <= 0 (res_stage_attnpool)."""
        ctx.relu_src = x if premask_input_grad else None
        T = x.dtype
        K, h, w, C = x.shape
        P, TP, H = h * w, AttnPoolFn.TP, ap.heads
        D = C // H
        assert ap.pos.shape[0] == P + 1 <= TP, "attention pool needs a 7x7 map (clip_backbone.py:86)"
        dev = x.device
        # the fused input-gradient epilogue (cddmsl_attnpool_dx) reads the pooled map's ReLU mask as one 64-bit word per column
        fused_dx = hip.attnpool_dx_ok(K, H, P, TP, C, T) and bool(ctx.needs_input_grad[0]) and os.environ.get("CDDMSL_ATTNPOOL_DX", "1") != "0"
        mbits = None
        if fused_dx and premask_input_grad:
            tok, mbits = hip.attn_tokens_fwd_mask(x.view(K, P, C), ap.pos.detach(), TP)       # [K,TP,C], [K,C]
        else:
            tok = hip.attn_tokens_fwd(x.view(K, P, C), ap.pos.detach(), TP)                  # [K,TP,C]
        wq, _ = ap.pq.get(T, False)
        wk, wkT = ap.pk.get(T, True)                                                          # [C,1,1,C]: rows = hd ; rows = c
        wv, wvT = ap.pv.get(T, True)
        wc, _ = ap.pc.get(T, False)
        q0 = hip.conv_fwd(tok.view(K, 1, TP, C), wq, None, ap.q_b.detach(), stride=TP).view(K, C)
        # zu[:, :H] = dZ (backward), zu[:, H:] = U : one buffer so the backward's dtok product is a single GEMM
        zu = torch.empty((K, 2 * H, C), device=dev, dtype=T)
        # U[k,h,:] = q0[k, hD:(h+1)D] @ Wk[hD:(h+1)D, :]        (per-head, contraction 64)
        hip.gemm_nt_batched(q0, wkT, zu, K, C, D, C, C, 2 * H * C, H, D, D, C, c_off=H * C)
        # S[k] = U[k] (H x C) . tok[k]^T (C x TP)
        S = torch.empty((K, H, TP), device=dev, dtype=torch.float32)
        hip.gemm_nt_batched(zu, tok, S, H, TP, C, C, C, TP, K, 2 * H * C, TP * C, H * TP, a_off=H * C)
        p, pT = hip.attnpool_softmax_fwd(S, P + 1, D ** -0.5, T)                              # [K,H,P+1] f32 ; [K,TP,H] T
        # Z[k] (H x C) = P[k] (H x TP) . tok[k] (TP x C)      (reduction over token rows)
        z = torch.empty((K, H, C), device=dev, dtype=T)
        hip.gemm_tn_batched(pT, tok, z, TP, H, C, H, C, C, K, TP * H, TP * C, H * C)
        # o[k, hD:(h+1)D] = Z[k,h,:] @ Wv[hD:(h+1)D, :]^T    (+ bv)
        o = torch.empty((K, C), device=dev, dtype=T)
        hip.gemm_nt_batched(z, wv, o, K, D, C, H * C, C, C, H, C, D * C, D)
        o = o + ap.v_b.detach().to(T)
        out = hip.conv_fwd(o.view(1, 1, K, C), wc, None, ap.c_b.detach(), out_f32=True).view(K, -1)
        ctx.ap = ap
        ctx.fused_dx = fused_dx
        ctx.save_for_backward(tok, q0, zu, p, z, o, mbits)
        ctx.shape = (K, h, w, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        ap = ctx.ap
        tok, q0, zu, p, z, o, mbits = ctx.saved_tensors
        K, h, w, C = ctx.shape
        P, TP, H = h * w, AttnPoolFn.TP, ap.heads
        D = C // H
        T = tok.dtype
        dev = tok.device
        train = not ap.frozen
        dout_t = dout.contiguous().to(T)
        if train:
            hip.conv_wgrad(o.view(1, 1, K, C), dout_t.view(1, 1, K, -1), _ohwi(ap.c_w).shape, out=_ohwi(_grad_buf(ap.c_w)))
            hip.colsum(dout_t, out=_grad_buf(ap.c_b))
        _, wcd = ap.pc.get(T, True)
        wk, wkT = ap.pk.get(T, True)
        wv, wvT = ap.pv.get(T, True)
        do = hip.conv_fwd(dout_t.view(1, 1, K, -1), wcd).view(K, C)
        if train:
            hip.colsum(do, out=_grad_buf(ap.v_b))
            # dWv[hD:(h+1)D, :] += dO[:, hD:(h+1)D]^T @ Z[:,h,:]      (reduction over regions, f32 atomics)
            hip.gemm_tn_batched(do, z, _grad_buf(ap.v_w), K, D, C, C, H * C, C, H, D, C, D * C, accumulate=True)
        # dZ[k,h,:] = dO[k, hD:(h+1)D] @ Wv[hD:(h+1)D, :]  -> zu[:, :H]
        hip.gemm_nt_batched(do, wvT, zu, K, C, D, C, C, 2 * H * C, H, D, D, C)
        # dP[k] = dZ[k] . tok[k]^T ; softmax backward in fp32
        dP = torch.empty((K, H, TP), device=dev, dtype=torch.float32)
        hip.gemm_nt_batched(zu, tok, dP, H, TP, C, C, C, TP, K, 2 * H * C, TP * C, H * TP)
        # ds = p * (dp - sum p dp) * scale, as dsT [K,TP,H] and stacked under p as pds [K,2H,TP] (one kernel)
        dsT, pds = hip.attnpool_softmax_bwd(p, dP, D ** -0.5, T)
        # dU[k] (H x C) = dS[k] (H x TP) . tok[k]
        du = torch.empty((K, H, C), device=dev, dtype=T)
        hip.gemm_tn_batched(dsT, tok, du, TP, H, C, H, C, C, K, TP * H, TP * C, H * C)
        # dq0[k, hD:(h+1)D] = dU[k,h,:] @ Wk[hD:(h+1)D, :]^T
        dq0 = torch.empty((K, C), device=dev, dtype=T)
        hip.gemm_nt_batched(du, wk, dq0, K, D, C, H * C, C, C, H, C, D * C, D)
        if train:
            # dWk[hD:(h+1)D, :] += q0[:, hD:(h+1)D]^T @ dU[:,h,:]   (k_proj.bias has an exactly-zero gradient)
            hip.gemm_tn_batched(q0, du, _grad_buf(ap.k_w), K, D, C, C, H * C, C, H, D, C, D * C, accumulate=True)
            _grad_buf(ap.k_b)
            hip.conv_wgrad(tok.view(K, 1, TP, C), dq0.view(K, 1, 1, C), _ohwi(ap.q_w).shape, stride=TP, out=_ohwi(_grad_buf(ap.q_w)))
            hip.colsum(dq0, out=_grad_buf(ap.q_b))
        _, wqd = ap.pq.get(T, True)
        gpos = _grad_buf(ap.pos) if train else None
        want_dx = ctx.needs_input_grad[0]
        if ctx.fused_dx and want_dx:
            # dtok[k] = [P[k]; dS[k]]^T . [dZ[k]; U[k]] is never stored: the product's epilogue adds the mean token's share
            # (its own row + the query path's g0 = Wq^T dq0, over the 49 pixels), applies the map's ReLU mask and keeps the
            # positional embedding's gradient (the unmasked column sums) -- one pass instead of product + read-back pass
            g0 = hip.conv_fwd(dq0.view(1, 1, K, C), wqd, out_f32=True).view(K, C)
            if mbits is None:
                mbits = torch.full((K, C), -1, device=dev, dtype=torch.int64)
            dx = hip.attnpool_dx(pds, zu, g0, mbits, P, gpos)
            return dx.view(K, h, w, C), None, None, None
        # dtok[k] (TP x C) = [P[k]; dS[k]]^T (TP x 2H) . [dZ[k]; U[k]] (2H x C)     (reduction over the 2H stacked rows)
        dtok = torch.empty((K, TP, C), device=dev, dtype=T)
        hip.gemm_tn_batched(pds, zu, dtok, 2 * H, TP, C, TP, C, C, K, 2 * H * TP, 2 * H * C, TP * C)
        dtok[:, 0, :] += hip.conv_fwd(dq0.view(1, 1, K, C), wqd).view(K, C)
        # one pass over dtok: the positional embedding's gradient (column sums per token row) and the map's gradient
        dx = None
        if want_dx or train:
            dx = hip.attn_tokens_bwd(dtok, P, ctx.relu_src if want_dx else None, gpos, want_dx)
        return (dx.view(K, h, w, C) if want_dx else None), None, None, None


def attnpool(x, ap):
    return AttnPoolFn.apply(x, None if ap.frozen else ap.q_w, ap)


# ------------------------------------------------------------------------------------------------
# fp32 heads: cosine-logit classifier (fast_rcnn.py:546-572) and contrastive loss (rcnn.py:308-317)
# ------------------------------------------------------------------------------------------------
_FP8_CONST = {}


def _fp8_const(value, device):
    k = (float(value), str(device))
    if k not in _FP8_CONST:
        _FP8_CONST[k] = torch.tensor([float(value)], device=device, dtype=torch.float32)
    return _FP8_CONST[k]


def fp8_unit_rows(wn):
    """e4m3 copy of unit-norm rows at the fixed scale 448 (the format's whole range)"""
    return hip.quantize_fp8(wn.contiguous(), _fp8_const(hip.FP8_MAX, wn.device))


class CosineLogitsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, wn, temperature, fp8=False, wn8=None):
        x = x.contiguous()
        if fp8 and x.shape[1] % 64 == 0 and wn.shape[0] <= 32:
            # the region x text-embedding contraction on e4m3 operands (BASELINE.json configs[4]): rows are unit vectors, so a fixed
            # scale of 448 uses the format's whole range; f32 accumulation, 1 / (448^2 T) applied to the f32 result.  The
            # backward is the exact-f32 one (straight-through: the quantisation is not differentiated).
            xn, inv = hip.l2norm_fwd(x, 1e-12)
            unit = _fp8_const(hip.FP8_MAX, x.device)
            if wn8 is None:          # (callers that own the embeddings pass their cached copy: no process-wide cache keyed by address)
                wn8 = fp8_unit_rows(wn)
            dot = hip.fp8_dot_nt(hip.quantize_fp8(xn, unit), wn8, _fp8_const(1.0 / (hip.FP8_MAX * hip.FP8_MAX * temperature), x.device))
            scores = torch.cat([dot, torch.zeros(x.shape[0], 1, device=x.device, dtype=torch.float32)], dim=1)
        else:
            scores, inv = hip.cosine_logits_fwd(x, wn, temperature)
        ctx.save_for_backward(x, wn, inv)
        ctx.t = temperature
        return scores

    @staticmethod
    def backward(ctx, ds):
        x, wn, inv = ctx.saved_tensors
        return hip.cosine_logits_bwd(ds, x.contiguous(), wn, inv, ctx.t), None, None, None, None


def cosine_logits(x, wn, temperature, fp8=False, wn8=None):
    return CosineLogitsFn.apply(x, wn, temperature, fp8, wn8)


class ContrastiveFn(torch.autograd.Function):
    """0.5*(CE(S, arange) + CE(S^T, arange)), S = norm(a) norm(b)^T; a, b f32 [n, d]."""

    @staticmethod
    def forward(ctx, a, b):
        an, ia = hip.l2norm_fwd(a.contiguous(), 0.0)
        bn, ib = hip.l2norm_fwd(b.contiguous(), 0.0)
        S = hip.linear_fwd(an, bn)                       # exact-f32 MFMA contraction
        loss, rl, cl = hip.contrastive_fwd(S)
        ctx.save_for_backward(an, ia, bn, ib, S, rl, cl)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        an, ia, bn, ib, S, rl, cl = ctx.saved_tensors
        dS = hip.contrastive_bwd(S, rl, cl, g)
        n = dS.shape[0]
        pad = (-n) % 4                                   # GEMM rows are whole 16-byte chunks: zero-pad the contraction
        P = (lambda t: torch.nn.functional.pad(t, (0, pad))) if pad else (lambda t: t)
        dan = hip.linear_fwd(P(dS).contiguous(), P(bn.t()).contiguous())
        dbn = hip.linear_fwd(P(dS.t()).contiguous(), P(an.t()).contiguous())
        return hip.l2norm_bwd(dan, an, ia), hip.l2norm_bwd(dbn, bn, ib)


def contrastive_loss(a, b):
    return ContrastiveFn.apply(a, b)


class LayerNormFn(torch.autograd.Function):
    """Frozen-affine LayerNorm of the mapper: f32 residual stream in, compute-dtype GEMM operand out (clipcap.py:97-100)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, out_dtype):
        y, mean, rstd = hip.layernorm_fwd(x.contiguous(), gamma, beta, out_dtype)
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        return hip.layernorm_bwd(dy, x.contiguous(), gamma, mean, rstd), None, None, None


def layer_norm(x2d, gamma, beta, out_dtype):
    return LayerNormFn.apply(x2d, gamma.detach(), beta.detach(), out_dtype)


class LayerNormSkipFn(torch.autograd.Function):
    """Pre-norm residual helper: returns (LN(x), x).  The second output is x itself, to be used by the residual add; in the
    backward both gradients arrive at this one node, so ``dx = d_skip + LN'(d_ln)`` is ONE kernel (LN backward accumulating
    into the skip gradient) instead of the LN backward plus autograd's separate accumulation add."""

    @staticmethod
    def forward(ctx, x, gamma, beta, out_dtype):
        x = x.contiguous()
        y, mean, rstd = hip.layernorm_fwd(x, gamma, beta, out_dtype)
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        x, gamma, mean, rstd = ctx.saved_tensors
        if dskip is None:
            return hip.layernorm_bwd(dy, x, gamma, mean, rstd), None, None, None
        # In place on the incoming skip gradient: its only other consumer is the sublayer branch hanging off the same residual
        # add, and that branch has necessarily run already (its result is the ``dy`` in hand).
        acc = dskip.contiguous()
        if dy is None:
            return acc, None, None, None
        return hip.layernorm_bwd(dy, x, gamma, mean, rstd, accumulate_into=acc), None, None, None


def layer_norm_skip(x2d, gamma, beta, out_dtype):
    return LayerNormSkipFn.apply(x2d, gamma.detach(), beta.detach(), out_dtype)


class FocalCEFn(torch.autograd.Function):
    """mean_i[ CE_i * (1 - p_t)^gamma * (bg_weight if target == bg else 1) ]   (fast_rcnn.py:624-644)"""

    @staticmethod
    def forward(ctx, logits, target, gamma, bg_class, bg_weight):
        logits = logits.contiguous()
        row, probs = hip.focal_ce_fwd(logits, target, gamma, bg_class, bg_weight)
        ctx.save_for_backward(logits, target, probs)
        ctx.meta = (gamma, bg_class, bg_weight)
        return row.mean()

    @staticmethod
    def backward(ctx, g):
        logits, target, probs = ctx.saved_tensors
        gamma, bg_class, bg_weight = ctx.meta
        return hip.focal_ce_bwd(logits, target, probs, g / logits.shape[0], gamma, bg_class, bg_weight), None, None, None, None


def focal_cross_entropy(logits, target, gamma, bg_class, bg_weight):
    return FocalCEFn.apply(logits, target, float(gamma or 0.0), int(bg_class), float(1.0 if bg_weight is None else bg_weight))


class RpnLossesFn(torch.autograd.Function):
    """(loss_rpn_cls, loss_rpn_loc) of RPN.losses (rpn.py:365-429) over the SAMPLED anchors only -- their index lists are known --
    as one kernel per direction instead of a dense BCE over all N x A anchors and ~35 gather / arithmetic launches."""

    @staticmethod
    def forward(ctx, logits, deltas, pos, neg, midx, gt, gt_off, anchors, weights, inv_norm):
        logits, deltas = logits.contiguous(), deltas.contiguous()
        ctx.save_for_backward(logits, deltas, pos, neg, midx, gt, gt_off, anchors)
        ctx.cfg = (weights, inv_norm)
        return hip.rpn_losses(logits, deltas, pos, neg, midx, gt, gt_off, anchors, weights, inv_norm)

    @staticmethod
    def backward(ctx, g):
        logits, deltas, pos, neg, midx, gt, gt_off, anchors = ctx.saved_tensors
        dl, dd = hip.rpn_losses(logits, deltas, pos, neg, midx, gt, gt_off, anchors, ctx.cfg[0], ctx.cfg[1], gout=g)
        return dl, dd, None, None, None, None, None, None, None, None


def rpn_losses(logits, deltas, pos, neg, midx, gt, gt_off, anchors, weights, inv_norm):
    return RpnLossesFn.apply(logits, deltas, pos, neg, midx, gt, gt_off, anchors, tuple(weights), float(inv_norm))


class BoxL1Fn(torch.autograd.Function):
    """sum over the foreground rows of |class-specific deltas - get_deltas(proposal, gt box)| * inv_norm (fast_rcnn.py:646-689)"""

    @staticmethod
    def forward(ctx, deltas, fg, cls, src, tgt, weights, inv_norm):
        deltas = deltas.contiguous()
        ctx.save_for_backward(deltas, fg, cls, src, tgt)
        ctx.cfg = (weights, inv_norm)
        return hip.box_l1(deltas, fg, cls, src, tgt, weights, inv_norm).view(())

    @staticmethod
    def backward(ctx, g):
        deltas, fg, cls, src, tgt = ctx.saved_tensors
        return hip.box_l1(deltas, fg, cls, src, tgt, ctx.cfg[0], ctx.cfg[1], gout=g), None, None, None, None, None, None


def box_l1(deltas, fg, cls, src, tgt, weights, inv_norm):
    return BoxL1Fn.apply(deltas, fg, cls, src.contiguous(), tgt.contiguous(), tuple(weights), float(inv_norm))


class MeanPoolFn(torch.autograd.Function):
    """box_features.mean(dim=[2, 3]) of Res5ROIHeads (roi_heads.py:487): [K,h,w,C] T -> [K,C] f32"""

    @staticmethod
    def forward(ctx, x):
        K, h, w, C = x.shape
        ctx.meta = (K, h, w, C, x.dtype)
        return hip.meanpool_fwd(x.contiguous().view(K, h * w, C))

    @staticmethod
    def backward(ctx, dy):
        K, h, w, C, T = ctx.meta
        return hip.meanpool_bwd(dy, h * w, T).view(K, h, w, C)


def mean_pool(x_nhwc):
    return MeanPoolFn.apply(x_nhwc)


class SmallAttnFn(torch.autograd.Function):
    """softmax(QK^T * scale) V of the ClipCap mapper (clipcap.py:59-83) as one fused bf16 MFMA kernel per direction;
    q [n*t, d], kv [n*t, 2d] are the projection outputs as they come (heads are column blocks, no permutes)."""

    @staticmethod
    def forward(ctx, q, kv, t, heads, scale):
        ctx.save_for_backward(q, kv)
        ctx.cfg = (t, heads, scale)
        return hip.attn_small_fwd(q, kv, t, heads, scale)

    @staticmethod
    def backward(ctx, do):
        q, kv = ctx.saved_tensors
        t, heads, scale = ctx.cfg
        dq, dkv = hip.attn_small_bwd(q, kv, do, t, heads, scale)
        return dq, dkv, None, None, None


def small_attention(q, kv, t, heads, scale):
    return SmallAttnFn.apply(q, kv, t, heads, scale)


class LastTokenAttnFn(torch.autograd.Function):
    """softmax(q K^T * scale) V for ONE query row per sequence (the mapper's last layer, evaluated for the token ``v2l`` keeps):
    q [n, d], kv [n*t, 2d] bf16 as the projections emit them -> o [n, d]; one kernel per direction instead of ~14 torch ops."""

    @staticmethod
    def forward(ctx, q, kv, t, heads, scale):
        o, p = hip.attn_last_fwd(q, kv, t, heads, scale)
        ctx.save_for_backward(q, kv, p)
        ctx.cfg = (t, heads, scale)
        return o

    @staticmethod
    def backward(ctx, do):
        q, kv, p = ctx.saved_tensors
        t, heads, scale = ctx.cfg
        dq, dkv = hip.attn_last_bwd(q, kv, do.to(q.dtype), p, t, heads, scale)
        return dq, dkv, None, None, None


def last_token_attention(q, kv, t, heads, scale):
    return LastTokenAttnFn.apply(q.contiguous(), kv.contiguous(), t, heads, scale)


class SmallAttnQkvFn(torch.autograd.Function):
    """The same on the output of ONE fused q|k|v projection [n*t, 3d]: the backward hands back one [n*t, 3d] gradient, so
    the projection's input gradient is one GEMM and no add of two partial gradients."""

    @staticmethod
    def forward(ctx, qkv, t, heads, scale):
        ctx.save_for_backward(qkv)
        ctx.cfg = (t, heads, scale)
        return hip.attn_small_fwd_qkv(qkv, t, heads, scale)

    @staticmethod
    def backward(ctx, do):
        (qkv,) = ctx.saved_tensors
        t, heads, scale = ctx.cfg
        return hip.attn_small_bwd_qkv(qkv, do, t, heads, scale), None, None, None


def small_attention_qkv(qkv, t, heads, scale):
    return SmallAttnQkvFn.apply(qkv, t, heads, scale)

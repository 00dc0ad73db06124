"""Thin functional wrappers over the C-ABI HIP library: shape checks on the host, raw pointers down.

Every function here requires CUDA (=HIP) tensors and enqueues on torch's current stream.  No
fallbacks: a CPU tensor or a missing library raises ``HipLibraryError``.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_long, c_void_p

import torch

from ._lib import check, lib, ptr, require_cuda, stream_ptr

DT = {torch.bfloat16: 0, torch.float32: 1}

_sigs_done = False


def _L():
    global _sigs_done
    L = lib()
    if not _sigs_done:
        vp, ci, cf = c_void_p, c_int, c_float
        L.cddmsl_last_kernel.argtypes = []
        L.cddmsl_plan_only.argtypes = [ci]
        L.cddmsl_conv_fwd.argtypes = [vp] * 7 + [ci] * 16 + [vp]
        L.cddmsl_conv_wgrad.argtypes = [vp] * 4 + [ci] * 12 + [vp]
        L.cddmsl_set_workspace.argtypes = [vp, c_long]
        L.cddmsl_weight_prep.argtypes = [vp] * 4 + [ci] * 5 + [vp]
        L.cddmsl_weight_prep_multi.argtypes = [vp, ci, ci, vp]
        L.cddmsl_preprocess.argtypes = [vp, vp] + [ci] * 6 + [vp, vp, ci, ci, vp]
        L.cddmsl_maxpool3s2_fwd.argtypes = [vp, vp] + [ci] * 5 + [vp]
        L.cddmsl_upsample_zero2.argtypes = [vp] * 4 + [ci] * 5 + [vp]
        L.cddmsl_meanpool_fwd.argtypes = [vp, vp, c_long, ci, ci, ci, vp]
        L.cddmsl_meanpool_bwd.argtypes = [vp, vp, c_long, ci, ci, ci, vp]
        L.cddmsl_preprocess224.argtypes = [vp, vp] + [ci] * 11 + [vp, vp, ci, vp]
        L.cddmsl_preprocess_batch.argtypes = [vp, vp, vp, ci, vp, ci, ci, ci, vp, vp, ci, ci, vp]
        L.cddmsl_preprocess224_batch.argtypes = [vp, vp, vp, ci, vp] + [ci] * 8 + [vp, vp, ci, vp]
        L.cddmsl_avgpool2_fwd.argtypes = [vp, vp] + [ci] * 5 + [vp]
        L.cddmsl_avgpool2_bwd.argtypes = [vp] * 4 + [ci] * 5 + [vp]
        L.cddmsl_avgpool2_bwd_q8.argtypes = [vp] * 4 + [ci] * 4 + [vp] * 4
        L.cddmsl_attn_tokens_fwd.argtypes = [vp] * 3 + [ci] * 5 + [vp]
        L.cddmsl_attn_tokens_bwd.argtypes = [vp] * 4 + [ci] * 5 + [vp]
        L.cddmsl_attnpool_softmax_fwd.argtypes = [vp] * 3 + [c_long, ci, ci, ci, cf, ci, vp]
        L.cddmsl_attnpool_softmax_bwd.argtypes = [vp] * 4 + [c_long, ci, ci, ci, cf, ci, vp]
        L.cddmsl_gemm_nt_batched.argtypes = [vp] * 4 + [ci] * 7 + [c_long] * 3 + [ci, ci, vp]
        L.cddmsl_gemm_tn_batched.argtypes = [vp] * 3 + [ci] * 7 + [c_long] * 3 + [ci, ci, vp]
        L.cddmsl_relu_bwd.argtypes = [vp, vp, vp, c_long, ci, ci, vp]
        L.cddmsl_colsum.argtypes = [vp, vp, c_long, ci, ci, ci, vp]
        L.cddmsl_sgd_clip_step.argtypes = [vp] * 4 + [ci, vp] + [cf] * 4 + [ci, vp]
        L.cddmsl_roi_align_forward.argtypes = [vp] * 5 + [ci] * 7 + [cf, ci, ci, ci, vp]
        L.cddmsl_roi_align_backward.argtypes = [vp] * 7 + [ci] * 7 + [cf, ci, ci, ci, vp]
        L.cddmsl_nms_anyorder.argtypes = [vp] * 4 + [ci, cf, vp, vp, vp]
        L.cddmsl_roi_align_nchw_anyorder.argtypes = [vp] * 3 + [ci] * 7 + [cf, ci, ci, ci, vp, vp, vp]
        L.cddmsl_roi_align_backward_nchw_anyorder.argtypes = [vp] * 3 + [ci] * 7 + [cf, ci, ci, ci, vp, vp, vp]
        L.cddmsl_quantize_fp8.argtypes = [vp] * 4 + [c_long, ci, vp]
        L.cddmsl_conv_fwd_fp8.argtypes = [vp] * 7 + [ci] * 10 + [vp] * 4
        L.cddmsl_conv_wgrad_fp8_ok.argtypes = [ci] * 6
        L.cddmsl_conv_wgrad_fp8.argtypes = [vp] * 4 + [ci] * 9 + [vp]
        L.cddmsl_conv_fwd_q8.argtypes = [vp] * 7 + [ci] * 10 + [vp] * 4
        L.cddmsl_fp8_dot_nt.argtypes = [vp] * 4 + [ci] * 4 + [vp]
        L.cddmsl_roi_align_forward_affine.argtypes = [vp] * 6 + [ci] * 8 + [cf, ci, ci, ci] + [vp] * 4
        L.cddmsl_roi_align_backward_pooled.argtypes = [vp] * 7 + [ci] * 7 + [cf, ci, ci, ci, vp]
        L.cddmsl_anchors.argtypes = [vp, vp, ci, ci, ci, cf, cf, vp]
        L.cddmsl_sort_desc.argtypes = [vp] * 5 + [ci, ci, vp, vp, vp]
        L.cddmsl_rpn_decode.argtypes = [vp] * 6 + [ci] * 5 + [cf] * 8 + [vp]
        L.cddmsl_nms.argtypes = [vp] * 5 + [ci, ci, cf, ci, vp]
        L.cddmsl_iou_match.argtypes = [vp, ci, vp, ci, vp, vp, vp, ci, cf, cf, ci, ci, ci, ci, vp]
        L.cddmsl_iou_match_batched.argtypes = [vp] * 7 + [ci] * 5 + [cf, cf] + [ci] * 4 + [vp]
        L.cddmsl_l2norm_fwd.argtypes = [vp, vp, vp, c_long, ci, cf, vp]
        L.cddmsl_l2norm_bwd.argtypes = [vp, vp, vp, vp, c_long, ci, vp]
        L.cddmsl_cosine_logits_fwd.argtypes = [vp] * 4 + [c_long, ci, ci, cf, cf, vp]
        L.cddmsl_cosine_logits_bwd.argtypes = [vp] * 5 + [c_long, ci, ci, cf, ci, vp]
        L.cddmsl_layernorm_fwd.argtypes = [vp] * 6 + [c_long, ci, cf, ci, vp]
        L.cddmsl_layernorm_bwd.argtypes = [vp] * 6 + [c_long, ci, ci, ci, vp]
        L.cddmsl_focal_ce_fwd.argtypes = [vp] * 4 + [c_long, ci, cf, ci, cf, vp]
        L.cddmsl_focal_ce_bwd.argtypes = [vp] * 5 + [c_long, ci, cf, ci, cf, vp]
        L.cddmsl_attn_small_fwd.argtypes = [vp] * 4 + [ci] * 8 + [cf, ci, vp]
        L.cddmsl_attn_small_bwd.argtypes = [vp] * 7 + [ci] * 8 + [cf, ci, vp]
        L.cddmsl_attn_tokens_fwd_mask.argtypes = [vp] * 4 + [ci] * 5 + [vp]
        L.cddmsl_attnpool_dx.argtypes = [vp] * 6 + [ci] * 6 + [vp]
        L.cddmsl_rpn_losses.argtypes = [vp, vp, vp, ci, vp, ci, vp, vp, vp, vp, c_long, cf, cf, cf, cf, cf, vp, vp, vp, vp, vp]
        L.cddmsl_box_l1.argtypes = [vp, ci, vp, ci, vp, vp, vp, cf, cf, cf, cf, cf, vp, vp, vp, vp]
        L.cddmsl_attn_last_fwd.argtypes = [vp] * 4 + [ci] * 8 + [cf, ci, vp]
        L.cddmsl_attn_last_bwd.argtypes = [vp] * 6 + [ci] * 8 + [cf, ci, vp]
        L.cddmsl_contrastive_fwd.argtypes = [vp] * 4 + [ci, ci, vp]
        L.cddmsl_contrastive_bwd.argtypes = [vp] * 5 + [ci, ci, vp]
        _sigs_done = True
    return L


class _Profiler:
    """Optional per-kernel timing with HIP events recorded on the launch stream (torch's current stream is the
    stream every kernel of this library is enqueued on).  Used by bench.py for the roofline object."""

    def __init__(self):
        self.on = False
        self.only = None
        self.events = []
        self.shapes = []

    def enable(self, only=None):
        """``only``: a set of profiler row names -- launches of other kernels record no events at all (an event pair per
        launch on ~1000 launches costs ~5 % of the step; inside bench.py's timed region only the dominant kernel is timed)."""
        self.on, self.only, self.events, self.shapes = True, (set(only) if only else None), [], []

    def begin(self, name=None, plan=None):
        """``name``: the row this launch will be booked under, if known before the launch; ``plan``: a callable that runs
        the entry point in plan-only mode (no launch) for entry points whose kernel is chosen inside the library."""
        if not self.on:
            return None
        if self.only is not None:
            if name is None and plan is not None:
                L = _L()
                L.cddmsl_plan_only(1)
                try:
                    plan()
                finally:
                    L.cddmsl_plan_only(0)
                name = _CONV_KERNEL.get(L.cddmsl_last_kernel())
            if name not in self.only:
                return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, e0, name, flops=0.0, shape=None, nbytes=0.0):
        if e0 is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.events.append((name, flops, e0, e1, nbytes))
        if shape is not None:
            self.shapes.append((name, shape, flops, e0, e1, nbytes))

    def by_shape(self):
        """(name, shape) -> [launches, ms, flops, algorithmic bytes]; call before collect()."""
        torch.cuda.synchronize()
        out = {}
        for name, shape, flops, e0, e1, nbytes in self.shapes:
            d = out.setdefault((name, shape), [0, 0.0, 0.0, 0.0])
            d[0] += 1
            d[1] += e0.elapsed_time(e1)
            d[2] += flops
            d[3] += nbytes
        return out

    def collect(self):
        torch.cuda.synchronize()
        out = {}
        for name, flops, e0, e1, nbytes in self.events:
            d = out.setdefault(name, {"flops": 0.0, "ms": 0.0, "launches": 0, "bytes": 0.0})
            d["flops"] += flops
            d["bytes"] += nbytes
            d["ms"] += e0.elapsed_time(e1)
            d["launches"] += 1
        self.on, self.only, self.events = False, None, []
        return out


PROFILE = _Profiler()
# cddmsl_last_kernel() ids -> profiler row names (one row per KERNEL, so the roofline object describes one kernel)
_CONV_KERNEL = {10: "k_conv_fwd256_fp8", 11: "k_conv_fwd2", 1: "k_conv_fwd", 2: "k_conv_fwd_reg", 3: "k_conv_fwd256", 4: "k_conv_wgrad", 5: "k_conv_wgrad_dma", 6: "k_wgrad256", 7: "k_gemm_tn_stream", 8: "k_conv3x3_small", 9: "k_gemm_tn_small", 12: "k_wgrad256_fp8"}


def _timed(name):
    def deco(fn):
        def wrapper(*a, **k):
            e0 = PROFILE.begin(name)
            r = fn(*a, **k)
            PROFILE.end(e0, name)
            return r
        wrapper.__name__, wrapper.__doc__ = fn.__name__, fn.__doc__
        return wrapper
    return deco


def _dt(t):
    if t.dtype not in DT:
        raise TypeError(f"unsupported dtype {t.dtype}: the kernels compute in bf16 or f32")
    return DT[t.dtype]


def _conv_input_pixels(N, H, W, Ho, Wo, KH, KW, stride, pool):
    """input pixels a convolution reads at least once (rows x columns its taps cover, per image)"""
    if pool:
        return N * min(H, 2 * Ho) * min(W, 2 * Wo)
    rows = min(H, Ho * min(KH, stride) if stride > KH else (Ho - 1) * stride + KH)
    cols = min(W, Wo * min(KW, stride) if stride > KW else (Wo - 1) * stride + KW)
    return N * rows * cols


class OutSpec:
    """Where a convolution puts its result when two launches' outputs have to end up adjacent (rows [0, N) and [N, 2N) of ONE
    buffer) without a concatenation copy: the first launch (``full`` None) allocates the double buffer, writes its half and
    records it; the second (``full`` set) writes rows [N, 2N).  Anything that does not fit leaves ``full`` / ``used`` untouched
    and the caller falls back to ``torch.cat``."""

    def __init__(self):
        self.full, self.used = None, 0


def conv_fwd(x, w, scale=None, bias=None, residual=None, relu=False, relu_mask=None, stride=1, pad=0,
             pool=False, out_f32=False, residual_pooled=False, emit8=None, out_spec=None):
    """x NHWC [N,H,W,Cin]; w [Cout,KH,KW,Cin] (same dtype).  Returns NHWC [N,Ho,Wo,Cout].
    y = relu?(acc*scale[n] + bias[n] + residual), zeroed where relu_mask <= 0 (ReLU backward).
    pool=True: 1x1 conv over the 2x2 average-pooled input (AvgPool2d(2) fused into the loader).
    residual_pooled=True: ``residual`` is [N, Ho//2, Wo//2, Cout] and every output pixel adds a quarter of its pooled pixel
    (the backward of AvgPool2d(2) fused into the epilogue; pooled tensor < 2 GiB -- see ``pooled_residual_ok``)."""
    require_cuda(x, w, scale, bias, residual, relu_mask)
    ensure_workspace(x.device)                       # (the split-K tail of badly quantised launches goes through it)
    assert x.dim() == 4 and w.dim() == 4 and x.is_contiguous() and w.is_contiguous()
    assert x.dtype == w.dtype
    N, H, W, Cin = x.shape
    Cout, KH, KW, Cin2 = w.shape
    assert Cin == Cin2, (x.shape, w.shape)
    if pool:
        Ho, Wo = H // 2, W // 2
    else:
        Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    ydt = torch.float32 if out_f32 else x.dtype
    if out_spec is not None and out_spec.full is None:
        out_spec.full = torch.empty((2 * N, Ho, Wo, Cout), device=x.device, dtype=ydt)
        y, out_spec.used = out_spec.full[:N], 1
    elif out_spec is not None and out_spec.used == 1 and tuple(out_spec.full.shape) == (2 * N, Ho, Wo, Cout) and out_spec.full.dtype == ydt:
        y, out_spec.used = out_spec.full[N:], 2
    else:
        y = torch.empty((N, Ho, Wo, Cout), device=x.device, dtype=ydt)
    for v in (scale, bias):
        assert v is None or (v.dtype == torch.float32 and v.numel() == Cout and v.is_contiguous())
    res_f32 = residual is not None and residual.dtype == torch.float32 and x.dtype != torch.float32
    if res_f32:       # f32 residual stream on the bf16 kernels: f32 output, vector epilogue
        assert out_f32 and relu_mask is None and not pool and Cout % 8 == 0
    if residual_pooled:
        assert residual is not None and not res_f32 and residual.dtype == x.dtype and residual.is_contiguous()
        assert tuple(residual.shape) == (N, Ho // 2, Wo // 2, Cout) and pooled_residual_ok(residual), residual.shape
    for v in (residual, relu_mask):
        assert v is None or (v is residual and residual_pooled) or (
            (v.dtype == x.dtype or (v is residual and res_f32)) and v.is_contiguous() and v.numel() == y.numel())
    y8 = None
    if emit8 is not None:     # fp8 configuration: (scale [1], amax [64]) of the consuming convolution's slot -> e4m3 copy of y, as y._fp8
        assert x.dtype == torch.bfloat16 and not out_f32 and not pool and not residual_pooled and not res_f32
        y8 = torch.empty(y.shape, device=x.device, dtype=torch.uint8)

    def launch():
        if y8 is not None:
            return _L().cddmsl_conv_fwd_q8(ptr(x), ptr(w), ptr(y), ptr(scale), ptr(bias), ptr(residual), ptr(relu_mask),
                                           N, H, W, Cin, Cout, KH, KW, stride, pad, int(relu), ptr(y8), ptr(emit8[0]), ptr(emit8[1]), stream_ptr())
        return _L().cddmsl_conv_fwd(ptr(x), ptr(w), ptr(y), ptr(scale), ptr(bias), ptr(residual), ptr(relu_mask),
                                    N, H, W, Cin, Cout, KH, KW, stride, pad, int(pool), Cout, Cout, Cout,
                                    int(relu), int(out_f32) | (2 if res_f32 else 0) | (4 if residual_pooled else 0), _dt(x), stream_ptr())
    e0 = PROFILE.begin(plan=launch) if PROFILE.on else None
    st = launch()
    check(st, "cddmsl_conv_fwd")
    PROFILE.end(e0, _CONV_KERNEL.get(_L().cddmsl_last_kernel(), "conv_fwd") if e0 is not None else "conv_fwd",
                2.0 * N * Ho * Wo * Cout * KH * KW * Cin,    # algorithmic 2*M*N*K
                (N * Ho * Wo, Cout, KH * KW * Cin, KH, int(pool), stride),
                # algorithmic HBM bytes: every operand once -- of x, the pixels the taps reach (a strided 1x1 convolution, e.g. the
                # attention pool's query projection over every 56th token row, reads one input row per output row, not the tensor)
                nbytes=float(_conv_input_pixels(N, H, W, Ho, Wo, KH, KW, stride, bool(pool)) * Cin * x.element_size()
                             + w.numel() * w.element_size() + y.numel() * y.element_size()
                             + (residual.numel() * residual.element_size() if residual is not None else 0)
                             + (relu_mask.numel() * relu_mask.element_size() if relu_mask is not None else 0)))
    if y8 is not None:
        y._fp8 = (y8, emit8[0].data_ptr())
    return y


def conv_emit8_ok(M, Cout, KH, KW, Cin_chunks_ok=True):
    """can a bf16 launch carry the e4m3 second output? (the 256x256 kernel's epilogue writes it: whole 256-wide tiles, >= 160 tiles)"""
    return Cout % 256 == 0 and KH * KW <= 31 and ((M + 255) // 256) * (Cout // 256) >= 160 and Cin_chunks_ok and os.environ.get("CDDMSL_GEMM256", "1") != "0"


# ------------------------------------------------------------------------------------------------ e4m3 (fp8) configuration
FP8_MAX = 448.0      # largest finite OCP e4m3fn magnitude


@_timed("quantize_fp8")
def quantize_fp8(x, scale=None, amax=None):
    """x (bf16 or f32, contiguous, numel % 8 == 0) -> uint8 tensor of OCP e4m3 bytes = sat(x * scale[0]).  ``scale`` / ``amax``:
    1-element f32 device tensors; max|x| is atomically max-ed into ``amax`` (delayed scaling: next step's scale)."""
    require_cuda(x, scale, amax)
    assert x.is_contiguous() and x.dtype in (torch.bfloat16, torch.float32) and x.numel() % 8 == 0
    y = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    check(_L().cddmsl_quantize_fp8(ptr(x), ptr(y), ptr(scale), ptr(amax), x.numel(), 0 if x.dtype == torch.bfloat16 else 1, stream_ptr()),
          "cddmsl_quantize_fp8")
    return y


def conv_fwd_fp8_ok(M, Cin, Cout, KH, KW, pad):
    """shapes the e4m3 kernel takes AND wins on: whole 256-wide tiles, enough of them to fill the chip, a reduction long enough to
    be MFMA-bound in bf16 (short reductions are bound by their output traffic, which fp8 operands do not shrink)"""
    min_tiles = int(os.environ.get("CDDMSL_FP8_MIN_TILES", "160"))      # (tests lower both to reach the kernel at small sizes)
    min_k = int(os.environ.get("CDDMSL_FP8_MIN_K", "2048"))
    return Cin % 128 == 0 and Cout % 256 == 0 and KH * KW <= 31 and 2 * pad <= KH - 1 and ((M + 255) // 256) * (Cout // 256) >= min_tiles \
        and KH * KW * Cin >= min_k


def conv_fwd_fp8(x8, w8, scale=None, bias=None, residual=None, relu=False, relu_mask=None, pad=0, out_f32=False, emit8=None):
    """x8 uint8 (e4m3) NHWC [N,H,W,Cin]; w8 uint8 [Cout,KH,KW,Cin].  y (bf16, or f32) = epilogue(sum x8 * w8): ``scale`` must carry
    the two dequantisation factors (x the FrozenBN scale).  stride 1."""
    require_cuda(x8, w8, scale, bias, residual, relu_mask)
    assert x8.dtype == torch.uint8 and w8.dtype == torch.uint8 and x8.is_contiguous() and w8.is_contiguous()
    N, H, W, Cin = x8.shape
    Cout, KH, KW, Cin2 = w8.shape
    assert Cin == Cin2
    Ho, Wo = H + 2 * pad - KH + 1, W + 2 * pad - KW + 1
    y = torch.empty((N, Ho, Wo, Cout), device=x8.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
    for v in (scale, bias):
        assert v is None or (v.dtype == torch.float32 and v.numel() == Cout and v.is_contiguous())
    for v in (residual, relu_mask):
        assert v is None or (v.dtype == torch.bfloat16 and v.is_contiguous() and v.numel() == y.numel())

    y8 = torch.empty(y.shape, device=x8.device, dtype=torch.uint8) if emit8 is not None else None
    assert emit8 is None or not out_f32

    def launch():
        return _L().cddmsl_conv_fwd_fp8(ptr(x8), ptr(w8), ptr(y), ptr(scale), ptr(bias), ptr(residual), ptr(relu_mask),
                                        N, H, W, Cin, Cout, KH, KW, pad, int(relu), int(out_f32), ptr(y8),
                                        ptr(emit8[0]) if emit8 else c_void_p(0), ptr(emit8[1]) if emit8 else c_void_p(0), stream_ptr())
    e0 = PROFILE.begin(name="k_conv_fwd256_fp8") if PROFILE.on else None
    check(launch(), "cddmsl_conv_fwd_fp8")
    PROFILE.end(e0, "k_conv_fwd256_fp8", 2.0 * N * Ho * Wo * Cout * KH * KW * Cin, (N * Ho * Wo, Cout, KH * KW * Cin, KH, 0, 1),
                nbytes=float(x8.numel() + w8.numel() + y.numel() * y.element_size() + (residual.numel() * 2 if residual is not None else 0)
                             + (relu_mask.numel() * 2 if relu_mask is not None else 0)))
    if y8 is not None:
        y._fp8 = (y8, emit8[0].data_ptr())
    return y


@_timed("fp8_dot_nt")
def fp8_dot_nt(a8, b8, alpha=None):
    """a8 [R,K], b8 [N,K] uint8 (e4m3), N <= 32, K % 64 == 0 -> [R,N] f32 = alpha[0] * a . b^T"""
    require_cuda(a8, b8, alpha)
    R, K = a8.shape
    N = b8.shape[0]
    assert a8.dtype == torch.uint8 and b8.dtype == torch.uint8 and a8.is_contiguous() and b8.is_contiguous() and b8.shape[1] == K
    c = torch.empty((R, N), device=a8.device, dtype=torch.float32)
    check(_L().cddmsl_fp8_dot_nt(ptr(a8), ptr(b8), ptr(c), ptr(alpha), R, N, K, N, stream_ptr()), "cddmsl_fp8_dot_nt")
    return c


def pooled_residual_ok(t):
    """the fused AvgPool2d(2)-backward residual is buffer-addressed from the tensor base: below 2 GiB, rows of whole 16-B chunks"""
    return t.numel() * t.element_size() < 2 ** 31 and t.shape[-1] % 8 == 0


def linear_fwd(x, w, scale=None, bias=None, residual=None, relu=False, relu_mask=None, out_f32=False):
    """x [M,K] @ w[N,K]^T with the conv epilogue (a 1x1 conv over M 'pixels')."""
    M, K = x.shape
    y = conv_fwd(x.view(1, 1, M, K), w.view(w.shape[0], 1, 1, K), scale, bias,
                 None if residual is None else residual.view(1, 1, M, -1), relu,
                 None if relu_mask is None else relu_mask.view(1, 1, M, -1), out_f32=out_f32)
    return y.view(M, w.shape[0])


_WORKSPACE = {}
WORKSPACE_BYTES = 288 << 20      # the largest split reduction of the 16 x 800x1333 step needs 258 MiB (1 008 blocks x 256 KiB)


def ensure_workspace(device):
    """Registers (once per device) the scratch the weight-gradient kernels' split reductions go through: see cddmsl_set_workspace."""
    key = torch.device(device).index or 0
    if key not in _WORKSPACE:
        ws = torch.empty(WORKSPACE_BYTES, dtype=torch.uint8, device=device)
        check(_L().cddmsl_set_workspace(ptr(ws), ctypes.c_long(ws.numel())), "cddmsl_set_workspace")
        _WORKSPACE[key] = ws
    return _WORKSPACE[key]


def conv_wgrad(x, dy, w_shape, scale=None, stride=1, pad=0, pool=False, out=None):
    """dW[Cout,KH,KW,Cin] (f32) += scale[n] * sum_m dY[m,n] * im2col(x)[m,k].  x NHWC, dy NHWC."""
    require_cuda(x, dy, scale, out)
    Cout, KH, KW, Cin = w_shape
    N, H, W, Cin2 = x.shape
    assert Cin == Cin2 and x.dtype == dy.dtype and x.is_contiguous() and dy.is_contiguous()
    assert dy.shape[-1] == Cout and dy.numel() // Cout == (N * (H // 2) * (W // 2) if pool else
                                                          N * ((H + 2 * pad - KH) // stride + 1) * ((W + 2 * pad - KW) // stride + 1))
    if out is None:
        out = torch.zeros(w_shape, device=x.device, dtype=torch.float32)
    assert out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == tuple(w_shape)
    ensure_workspace(x.device)
    def launch():
        return _L().cddmsl_conv_wgrad(ptr(x), ptr(dy), ptr(out), ptr(scale), N, H, W, Cin, Cout, KH, KW, stride, pad,
                                      int(pool), Cout, _dt(x), stream_ptr())
    e0 = PROFILE.begin(plan=launch) if PROFILE.on else None
    st = launch()
    check(st, "cddmsl_conv_wgrad")
    PROFILE.end(e0, _CONV_KERNEL.get(_L().cddmsl_last_kernel(), "conv_wgrad") if e0 is not None else "conv_wgrad",
                2.0 * (dy.numel() // Cout) * Cout * KH * KW * Cin,
                (dy.numel() // Cout, Cout, KH * KW * Cin, KH, int(pool), stride),
                # algorithmic HBM bytes: both operands once (a long reduction into a small dW is bound by reading them, not by the MFMAs)
                nbytes=float(_conv_input_pixels(N, H, W, (H // 2) if pool else (H + 2 * pad - KH) // stride + 1,
                                                 (W // 2) if pool else (W + 2 * pad - KW) // stride + 1, KH, KW, stride, bool(pool)) * Cin * x.element_size()
                             + dy.numel() * dy.element_size() + out.numel() * 4))
    return out


def conv_wgrad_fp8_ok(M, Cin, Cout, KH, KW, pad, min_m=None, taps_only=True):
    """shapes the e4m3 weight-gradient kernel takes AND wins on: whole 256 x 256 output tiles of a "same" convolution and a reduction
    long enough to fill the chip.  ``taps_only``: 1x1 layers only where the caller says their e4m3 copies come for free (written by the
    producing epilogues: the RoI head's conv1 / conv3); the kernel itself runs ~2x the bf16 one on every shape measured."""
    # (default: the RoI head's 3x3 layers; a caller whose copies cost nothing extra passes its own bound; tests lower it to reach the kernel at small sizes)
    min_m = int(os.environ.get("CDDMSL_FP8_WGRAD_MIN_M", "300000" if min_m is None else str(min_m)))
    return os.environ.get("CDDMSL_FP8_WGRAD", "1") != "0" and (KH * KW > 1 or not taps_only) and M >= min_m \
        and bool(_L().cddmsl_conv_wgrad_fp8_ok(Cin, Cout, KH, KW, pad, Cout))


def conv_wgrad_fp8(x8, dy8, w_shape, scale, pad=0, out=None):
    """dW[Cout,KH,KW,Cin] (f32) += scale[n] * sum_m dy8[m,n] * im2col(x8)[m,k]; x8, dy8 uint8 (e4m3) NHWC, stride 1, "same" padding.
    ``scale`` must carry the two dequantisation factors (x the FrozenBN scale)."""
    require_cuda(x8, dy8, scale, out)
    Cout, KH, KW, Cin = w_shape
    N, H, W, Cin2 = x8.shape
    assert Cin == Cin2 and x8.dtype == torch.uint8 and dy8.dtype == torch.uint8 and x8.is_contiguous() and dy8.is_contiguous()
    assert dy8.shape[-1] == Cout and dy8.numel() // Cout == N * H * W and 2 * pad == KH - 1
    assert scale is not None and scale.dtype == torch.float32 and scale.numel() == Cout and scale.is_contiguous()
    if out is None:
        out = torch.zeros(w_shape, device=x8.device, dtype=torch.float32)
    assert out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == tuple(w_shape)
    ensure_workspace(x8.device)
    def launch():
        return _L().cddmsl_conv_wgrad_fp8(ptr(x8), ptr(dy8), ptr(out), ptr(scale), N, H, W, Cin, Cout, KH, KW, pad, Cout, stream_ptr())
    e0 = PROFILE.begin(name="k_wgrad256_fp8") if PROFILE.on else None
    check(launch(), "cddmsl_conv_wgrad_fp8")
    PROFILE.end(e0, "k_wgrad256_fp8", 2.0 * (N * H * W) * Cout * KH * KW * Cin, (N * H * W, Cout, KH * KW * Cin, KH, 0, 1),
                nbytes=float(x8.numel() + dy8.numel() + out.numel() * 4))
    return out


@_timed("weight_prep")
def weight_prep(w_master, scale, dtype, want_fwd=True, want_dgrad=True):
    """f32 master [Cout,KH,KW,Cin] -> (fwd weights, dgrad weights [Cin,KH,KW,Cout] flipped, *scale[cout])."""
    require_cuda(w_master, scale)
    assert w_master.dtype == torch.float32 and w_master.is_contiguous() and w_master.dim() == 4
    Cout, KH, KW, Cin = w_master.shape
    wf = torch.empty((Cout, KH, KW, Cin), device=w_master.device, dtype=dtype) if want_fwd else None
    wd = torch.empty((Cin, KH, KW, Cout), device=w_master.device, dtype=dtype) if want_dgrad else None
    st = _L().cddmsl_weight_prep(ptr(w_master), ptr(scale), ptr(wf), ptr(wd), Cout, KH, KW, Cin, DT[dtype], stream_ptr())
    check(st, "cddmsl_weight_prep")
    return wf, wd


@_timed("weight_prep")
def weight_prep_multi(table, count, dtype):
    """``table``: int64 device tensor [count, 8] = {w, scale, wf, wd pointers, Cout, KH, KW, Cin} -> all copies in one launch"""
    require_cuda(table)
    assert table.dtype == torch.int64 and table.is_contiguous() and table.shape == (count, 8)
    check(_L().cddmsl_weight_prep_multi(ptr(table), count, DT[dtype], stream_ptr()), "cddmsl_weight_prep_multi")


# ------------------------------------------------------------------------------------------------ elementwise
def _f3(v):
    return (c_float * 3)(*[float(x) for x in v])


def _image_table(images_u8):
    """host arrays (device pointers, heights, widths) of a list of CHW images: the batched preprocessing entries' image table"""
    import ctypes
    n = len(images_u8)
    ptrs = (ctypes.c_void_p * n)(*[im.data_ptr() for im in images_u8])
    hs = (ctypes.c_int * n)(*[int(im.shape[1]) for im in images_u8])
    ws = (ctypes.c_int * n)(*[int(im.shape[2]) for im in images_u8])
    return ptrs, hs, ws


@_timed("preprocess")
def preprocess(images_u8, Hp, Wp, mean, std, dtype, Cp=None, div255=True):
    """list of u8 CHW device tensors -> normalised, zero-padded NHWC [N,Hp,Wp,Cp] (rcnn.py:758-768).
    div255: CLIP models take x/255 (rcnn.py:87-91); stock models normalise raw 0-255 pixels."""
    require_cuda(*images_u8)
    Cp = Cp or (8 if dtype == torch.bfloat16 else 4)
    out = torch.empty((len(images_u8), Hp, Wp, Cp), device=images_u8[0].device, dtype=dtype)
    m, s = _f3(mean), _f3(std)
    for im in images_u8:
        assert im.dtype == torch.uint8 and im.dim() == 3 and im.shape[0] == 3 and im.is_contiguous()
    ptrs, hs, ws = _image_table(images_u8)
    check(_L().cddmsl_preprocess_batch(ptrs, hs, ws, len(images_u8), ptr(out), Hp, Wp, Cp, m, s, int(div255), DT[dtype], stream_ptr()),
          "cddmsl_preprocess_batch")
    return out


@_timed("preprocess224")
def preprocess224(images_u8, Hp, Wp, mean, std, dtype, size=224, Cp=None):
    """rcnn.py:161-179: /255 -> pad to (Hp,Wp) -> bicubic short side `size` -> center crop -> normalise; NHWC out."""
    require_cuda(*images_u8)
    Cp = Cp or (8 if dtype == torch.bfloat16 else 4)
    if Wp <= Hp:
        RW, RH = size, int(size * Hp / Wp)
    else:
        RH, RW = size, int(size * Wp / Hp)
    top, left = int(round((RH - size) / 2.0)), int(round((RW - size) / 2.0))
    out = torch.empty((len(images_u8), size, size, Cp), device=images_u8[0].device, dtype=dtype)
    m, s = _f3(mean), _f3(std)
    for im in images_u8:
        assert im.dtype == torch.uint8 and im.dim() == 3 and im.is_contiguous()
    ptrs, hs, ws = _image_table(images_u8)
    check(_L().cddmsl_preprocess224_batch(ptrs, hs, ws, len(images_u8), ptr(out), Hp, Wp, RH, RW, top, left, size, Cp,
                                           m, s, DT[dtype], stream_ptr()), "cddmsl_preprocess224_batch")
    return out


@_timed("avgpool2_fwd")
def avgpool2_fwd(x):
    require_cuda(x)
    N, H, W, C = x.shape
    y = torch.empty((N, H // 2, W // 2, C), device=x.device, dtype=x.dtype)
    check(_L().cddmsl_avgpool2_fwd(ptr(x), ptr(y), N, H, W, C, _dt(x), stream_ptr()), "cddmsl_avgpool2_fwd")
    return y


@_timed("avgpool2_bwd")
def avgpool2_bwd(dy, in_shape, mask=None, add=None, emit8=None):
    """dx = up(dy)/4 (+ add), zeroed where mask <= 0.  in_shape = (N,H,W,C) of the pooled tensor's input.  ``emit8`` (scale, amax),
    bf16 only: the pass also writes dx's e4m3 copy (attached as ``dx._fp8``) and records max|dx|."""
    require_cuda(dy, mask, add)
    N, H, W, C = in_shape
    assert dy.is_contiguous() and tuple(dy.shape) == (N, H // 2, W // 2, C)
    dx = torch.empty(in_shape, device=dy.device, dtype=dy.dtype)
    if emit8 is not None and dy.dtype == torch.bfloat16:
        y8 = torch.empty(in_shape, device=dy.device, dtype=torch.uint8)
        check(_L().cddmsl_avgpool2_bwd_q8(ptr(dy), ptr(mask), ptr(add), ptr(dx), N, H, W, C, ptr(y8), ptr(emit8[0]), ptr(emit8[1]), stream_ptr()),
              "cddmsl_avgpool2_bwd_q8")
        dx._fp8 = (y8, emit8[0].data_ptr())
        return dx
    check(_L().cddmsl_avgpool2_bwd(ptr(dy), ptr(mask), ptr(add), ptr(dx), N, H, W, C, _dt(dy), stream_ptr()), "cddmsl_avgpool2_bwd")
    return dx


@_timed("attn_tokens_fwd")
def attn_tokens_fwd(x, pos, tp=None):
    """x [K,P,C] (T), pos [P+1,C] f32 -> tok [K,TP,C] (TP >= P+1 token rows per region; pad rows are zero)"""
    require_cuda(x, pos)
    K, P, C = x.shape
    tp = tp or P + 1
    assert pos.dtype == torch.float32 and tuple(pos.shape) == (P + 1, C) and x.is_contiguous() and pos.is_contiguous()
    tok = torch.empty((K, tp, C), device=x.device, dtype=x.dtype)
    check(_L().cddmsl_attn_tokens_fwd(ptr(x), ptr(pos), ptr(tok), K, P, tp, C, _dt(x), stream_ptr()), "cddmsl_attn_tokens_fwd")
    return tok


@_timed("attn_tokens_fwd")
def attn_tokens_fwd_mask(x, pos, tp=None):
    """the same, plus mbits [K,C] int64: bit p of a column's word = (x[k,p,col] > 0) -- the pooled map's ReLU mask for ``attnpool_dx``"""
    require_cuda(x, pos)
    K, P, C = x.shape
    tp = tp or P + 1
    assert pos.dtype == torch.float32 and tuple(pos.shape) == (P + 1, C) and x.is_contiguous() and pos.is_contiguous() and P <= 64
    tok = torch.empty((K, tp, C), device=x.device, dtype=x.dtype)
    mbits = torch.empty((K, C), device=x.device, dtype=torch.int64)
    check(_L().cddmsl_attn_tokens_fwd_mask(ptr(x), ptr(pos), ptr(tok), ptr(mbits), K, P, tp, C, _dt(x), stream_ptr()), "cddmsl_attn_tokens_fwd_mask")
    return tok, mbits


def rpn_losses(logits, deltas, pos, neg, midx, gt, gt_off, anchors, weights, inv_norm, gout=None):
    """forward: -> [2] f32 (loss_rpn_cls, loss_rpn_loc); backward (gout [2]): -> (dlogits, ddeltas), zero except at the sampled anchors"""
    require_cuda(logits, deltas, pos, neg, midx, gt, gt_off, anchors)
    for t in (logits, deltas, gt, anchors):
        assert t.dtype == torch.float32 and t.is_contiguous()
    for t in (pos, neg, midx, gt_off):
        assert t.dtype == torch.int64 and t.is_contiguous()
    A = anchors.shape[0]
    assert deltas.numel() == 4 * logits.numel() and midx.numel() == logits.numel()
    args = (ptr(logits), ptr(deltas), ptr(pos), pos.numel(), ptr(neg), neg.numel(), ptr(midx), ptr(gt), ptr(gt_off), ptr(anchors), A,
            *[float(w) for w in weights], float(inv_norm))
    if gout is None:
        out = torch.empty(2, device=logits.device, dtype=torch.float32)
        check(_L().cddmsl_rpn_losses(*args, ptr(out), None, None, None, stream_ptr()), "cddmsl_rpn_losses")
        return out
    dl, dd = torch.zeros_like(logits), torch.zeros_like(deltas)
    g = gout.contiguous().float()
    check(_L().cddmsl_rpn_losses(*args, None, ptr(g), ptr(dl), ptr(dd), stream_ptr()), "cddmsl_rpn_losses(backward)")
    return dl, dd


def box_l1(deltas, fg, cls, src, tgt, weights, inv_norm, gout=None):
    """forward: -> [1] f32 sum_{fg} |deltas[r, 4 cls[r]..+4] - get_deltas(src[r], tgt[r])| * inv_norm; backward (gout [1]): -> ddeltas"""
    require_cuda(deltas, fg, src, tgt)
    assert deltas.dtype == torch.float32 and deltas.dim() == 2 and deltas.is_contiguous() and fg.dtype == torch.int64 and fg.is_contiguous()
    assert src.dtype == tgt.dtype == torch.float32 and src.is_contiguous() and tgt.is_contiguous() and (cls is None or (cls.dtype == torch.int64 and cls.is_contiguous()))
    args = (ptr(deltas), deltas.shape[1], ptr(fg), fg.numel(), ptr(cls), ptr(src), ptr(tgt), *[float(w) for w in weights], float(inv_norm))
    if gout is None:
        out = torch.empty(1, device=deltas.device, dtype=torch.float32)
        check(_L().cddmsl_box_l1(*args, ptr(out), None, None, stream_ptr()), "cddmsl_box_l1")
        return out
    dd = torch.zeros_like(deltas)
    g = gout.contiguous().float().view(1)
    check(_L().cddmsl_box_l1(*args, None, ptr(g), ptr(dd), stream_ptr()), "cddmsl_box_l1(backward)")
    return dd


def attnpool_dx_ok(K, H, P, TP, C, dtype):
    return dtype == torch.bfloat16 and P == 49 and TP == 56 and 2 * H <= 64 and (2 * H) % 8 == 0 and C % 128 == 0 and 0 < K <= 65535 * 8


@_timed("attnpool_dx")
def attnpool_dx(pds, zu, g0, mbits, P, gpos=None):
    """pds [K,2H,TP] = [p ; ds], zu [K,2H,C] = [dZ ; U] (bf16), g0 [K,C] f32, mbits [K,C] int64 -> dx [K,P,C] bf16 (masked),
    gpos [P+1,C] f32 accumulated when given: the token-gradient product with its epilogue fused (cddmsl_attnpool_dx)"""
    require_cuda(pds, zu, g0, mbits)
    K, H2, TP = pds.shape
    C = zu.shape[2]
    assert pds.dtype == zu.dtype == torch.bfloat16 and g0.dtype == torch.float32 and mbits.dtype == torch.int64
    assert pds.is_contiguous() and zu.is_contiguous() and g0.is_contiguous() and mbits.is_contiguous() and tuple(zu.shape) == (K, H2, C)
    assert tuple(g0.shape) == (K, C) and tuple(mbits.shape) == (K, C) and (gpos is None or (gpos.dtype == torch.float32 and tuple(gpos.shape) == (P + 1, C) and gpos.is_contiguous()))
    dx = torch.empty((K, P, C), device=pds.device, dtype=pds.dtype)
    check(_L().cddmsl_attnpool_dx(ptr(pds), ptr(zu), ptr(g0), ptr(mbits), ptr(dx), ptr(gpos), K, H2, P, TP, C, 0, stream_ptr()), "cddmsl_attnpool_dx")
    return dx


@_timed("attn_tokens_bwd")
def attn_tokens_bwd(dtok, P, relu_mask=None, gpos=None, want_dx=True):
    """dtok [K,TP,C] -> dx [K,P,C], zeroed where relu_mask [K,P,C] <= 0; gpos (f32 [P+1,C], accumulated into) += the
    per-token column sums of dtok (the positional embedding's gradient) in the same pass"""
    require_cuda(dtok, relu_mask, gpos)
    K, TP, C = dtok.shape
    assert dtok.is_contiguous()
    assert relu_mask is None or (relu_mask.is_contiguous() and relu_mask.dtype == dtok.dtype and relu_mask.numel() == K * P * C)
    assert gpos is None or (gpos.is_contiguous() and gpos.dtype == torch.float32 and tuple(gpos.shape) == (P + 1, C))
    assert want_dx or gpos is not None
    dx = torch.empty((K, P, C), device=dtok.device, dtype=dtok.dtype) if want_dx else None
    check(_L().cddmsl_attn_tokens_bwd(ptr(dtok), ptr(relu_mask), ptr(dx), ptr(gpos), K, P, TP, C, _dt(dtok), stream_ptr()), "cddmsl_attn_tokens_bwd")
    return dx


@_timed("attnpool_softmax")
def attnpool_softmax_fwd(S, P1, scale, dtype):
    """S [K,H,TP] f32 -> (p [K,H,P1] f32 = softmax(S[..., :P1] * scale), pT [K,TP,H] dtype with zero rows past P1)"""
    require_cuda(S)
    K, H, TP = S.shape
    assert S.dtype == torch.float32 and S.is_contiguous() and P1 <= TP
    p = torch.empty((K, H, P1), device=S.device, dtype=torch.float32)
    pT = torch.empty((K, TP, H), device=S.device, dtype=dtype)
    check(_L().cddmsl_attnpool_softmax_fwd(ptr(S), ptr(p), ptr(pT), K, H, P1, TP, float(scale), DT[dtype], stream_ptr()), "cddmsl_attnpool_softmax_fwd")
    return p, pT


@_timed("attnpool_softmax")
def attnpool_softmax_bwd(p, dP, scale, dtype):
    """p [K,H,P1] f32, dP [K,H,TP] f32 -> (dsT [K,TP,H], pds [K,2H,TP] = [p ; ds]) in dtype, zero past P1"""
    require_cuda(p, dP)
    K, H, P1 = p.shape
    TP = dP.shape[2]
    assert p.dtype == dP.dtype == torch.float32 and p.is_contiguous() and dP.is_contiguous() and dP.shape[:2] == (K, H)
    dsT = torch.empty((K, TP, H), device=p.device, dtype=dtype)
    pds = torch.empty((K, 2 * H, TP), device=p.device, dtype=dtype)
    check(_L().cddmsl_attnpool_softmax_bwd(ptr(p), ptr(dP), ptr(dsT), ptr(pds), K, H, P1, TP, float(scale), DT[dtype], stream_ptr()), "cddmsl_attnpool_softmax_bwd")
    return dsT, pds


def _eptr(t, elem_offset=0):
    return ctypes.c_void_p(t.data_ptr() + elem_offset * t.element_size())


@_timed("gemm_nt_batched")
def gemm_nt_batched(a, w, c, M, N, K, lda, ldb, ldc, batch, sa, sw, sc, a_off=0, w_off=0, c_off=0, bias=None):
    """C_b[m][n] = sum_k A_b[m][k] * B_b[n][k] for b < batch; raw strided views of the tensors a, w, c (element offsets /
    strides).  c's dtype decides f32 vs compute-dtype output."""
    require_cuda(a, w, c, bias)
    assert a.dtype == w.dtype and c.dtype in (a.dtype, torch.float32)
    out_f32 = int(c.dtype == torch.float32 and a.dtype != torch.float32)
    check(_L().cddmsl_gemm_nt_batched(_eptr(a, a_off), _eptr(w, w_off), _eptr(c, c_off), ptr(bias), M, N, K, lda, ldb, ldc, batch,
                                      sa, sw, sc, out_f32, _dt(a), stream_ptr()), "cddmsl_gemm_nt_batched")


@_timed("gemm_tn_batched")
def gemm_tn_batched(a, b, out, M, N, K, lda, ldb, ldo, batch, sa, sb, so, a_off=0, b_off=0, o_off=0, accumulate=False):
    """out_b[n][k] (+)= sum_m A_b[m][n] * B_b[m][k].  accumulate=True: f32 atomic adds into `out` (f32); otherwise a plain
    store in out's dtype (f32 or the compute dtype)."""
    require_cuda(a, b, out)
    assert a.dtype == b.dtype
    if accumulate:
        assert out.dtype == torch.float32
        mode = 0
    else:
        mode = 1 if (out.dtype == torch.float32 and a.dtype != torch.float32) else (1 if out.dtype == torch.float32 else 2)
        assert out.dtype in (a.dtype, torch.float32)
    check(_L().cddmsl_gemm_tn_batched(_eptr(a, a_off), _eptr(b, b_off), _eptr(out, o_off), M, N, K, lda, ldb, ldo, batch, sa, sb, so,
                                      mode, _dt(a), stream_ptr()), "cddmsl_gemm_tn_batched")


@_timed("relu_bwd")
def relu_bwd(g, y):
    """dx = g * (y > 0) in y's dtype; g may be f32 while y is bf16."""
    require_cuda(g, y)
    assert g.is_contiguous() and y.is_contiguous() and g.numel() == y.numel()
    g_f32 = int(g.dtype == torch.float32 and y.dtype != torch.float32)
    assert g_f32 or g.dtype == y.dtype
    dx = torch.empty_like(y)
    check(_L().cddmsl_relu_bwd(ptr(g), ptr(y), ptr(dx), y.numel(), g_f32, _dt(y), stream_ptr()), "cddmsl_relu_bwd")
    return dx


@_timed("colsum")
def colsum(x2d, period=1, out=None):
    """f32 column sums of x [rows, cols] (rows folded modulo `period`)."""
    require_cuda(x2d, out)
    rows, cols = x2d.shape
    assert x2d.is_contiguous()
    if out is None:
        out = torch.zeros((period, cols) if period > 1 else (cols,), device=x2d.device, dtype=torch.float32)
    check(_L().cddmsl_colsum(ptr(x2d), ptr(out), rows, cols, period, _dt(x2d), stream_ptr()), "cddmsl_colsum")
    return out


def same_layout(a, b):
    """Same shape and the same element order in memory (strides of size-1 dims are irrelevant)."""
    return a.shape == b.shape and all(sa == sb for sa, sb, n in zip(a.stride(), b.stride(), a.shape) if n > 1)


@_timed("sgd_clip_step")
def sgd_clip_step(params, grads, moms, norm_ws, lr, momentum, wd, clip, first_step):
    """Fused per-parameter grad-norm clip + SGD(momentum, wd) over a list of f32 tensors (solver/build.py:59-130)."""
    n = len(params)
    if n == 0:
        return
    require_cuda(*params, *grads, *moms)
    for p, g, m in zip(params, grads, moms):
        assert p.dtype == g.dtype == m.dtype == torch.float32
        assert same_layout(p, g) and same_layout(p, m), "param/grad/momentum must share a memory layout"
    P = (c_void_p * n)(*[p.data_ptr() for p in params])
    G = (c_void_p * n)(*[g.data_ptr() for g in grads])
    M = (c_void_p * n)(*[m.data_ptr() for m in moms])
    S = (c_long * n)(*[p.numel() for p in params])
    check(_L().cddmsl_sgd_clip_step(P, G, M, S, n, ptr(norm_ws), lr, momentum, wd, clip, int(first_step), stream_ptr()),
          "cddmsl_sgd_clip_step")


# ------------------------------------------------------------------------------------------------ RoIAlign
@_timed("roi_align_forward")
def roi_align_forward(x, rois, ph, pw, spatial_scale, sampling_ratio, aligned, dbg_grid=None, with_pooled=False, extra_rows=0):
    """x NHWC [N,H,W,C]; rois [K,5] f32 (batch_idx,x0,y0,x1,y1) -> [K,ph,pw,C]  (layers/roi_align.py:49-65).
    ``with_pooled``: also returns AvgPool2d(2) of the result [K,ph/2,pw/2,C] (bit-identical to avgpool2_fwd of it).
    ``extra_rows``: the outputs are allocated with that many more (unwritten) leading rows behind the K crops, for a caller that
    appends maps of the same geometry without a concatenation copy."""
    require_cuda(x, rois)
    assert rois.dim() == 2 and rois.size(1) == 5 and rois.dtype == torch.float32 and rois.is_contiguous()
    N, H, W, C = x.shape
    K = rois.shape[0]
    y = torch.empty((K + extra_rows, ph, pw, C), device=x.device, dtype=x.dtype)
    yp = torch.empty((K + extra_rows, ph // 2, pw // 2, C), device=x.device, dtype=x.dtype) if with_pooled else None
    check(_L().cddmsl_roi_align_forward(ptr(x), ptr(rois), ptr(y), ptr(yp), ptr(dbg_grid), N, C, H, W, K, ph, pw, spatial_scale,
                                         sampling_ratio, int(aligned), _dt(x), stream_ptr()), "cddmsl_roi_align_forward")
    return (y, yp) if with_pooled else y


@_timed("roi_align_forward")
def roi_align_forward_affine(x, rois, ph, pw, spatial_scale, sampling_ratio, aligned, scale=None, bias=None, relu=False,
                             pooled_only=False, extra_rows=0, emit8=None):
    """RoIAlign of a map that already went through a 1x1 convolution (layers.RoIStageFn):
    ``pooled_only=False``: y [K,ph,pw,C] = relu?(scale * roi_align(x) + bias) (per channel, f32 scale / bias);
    ``pooled_only=True``: only AvgPool2d(2) of the crops, [K,ph/2,pw/2,C] (no affine) -- the full-resolution crops are never written."""
    require_cuda(x, rois)
    assert rois.dim() == 2 and rois.size(1) == 5 and rois.dtype == torch.float32 and rois.is_contiguous()
    assert (scale is None) == (bias is None) and not (pooled_only and scale is not None)
    N, H, W, C = x.shape
    K = rois.shape[0]
    if pooled_only:
        y, yp = None, torch.empty((K + extra_rows, ph // 2, pw // 2, C), device=x.device, dtype=x.dtype)
    else:
        y, yp = torch.empty((K + extra_rows, ph, pw, C), device=x.device, dtype=x.dtype), None
        assert scale is None or (scale.dtype == torch.float32 and scale.numel() == C and bias.numel() == C and scale.is_contiguous() and bias.is_contiguous())
    y8 = None
    if emit8 is not None:     # e4m3 copy of the crops for the convolution that consumes them (fp8 configuration), as y._fp8
        assert not pooled_only and x.dtype == torch.bfloat16
        y8 = torch.empty(y.shape, device=x.device, dtype=torch.uint8)
    check(_L().cddmsl_roi_align_forward_affine(ptr(x), ptr(rois), ptr(y), ptr(yp), ptr(scale), ptr(bias), int(relu), N, C, H, W, K,
                                                ph, pw, spatial_scale, sampling_ratio, int(aligned), _dt(x), ptr(y8),
                                                ptr(emit8[0]) if emit8 else c_void_p(0), ptr(emit8[1]) if emit8 else c_void_p(0), stream_ptr()),
          "cddmsl_roi_align_forward_affine")
    if y8 is not None:
        y._fp8 = (y8, emit8[0].data_ptr())
    return yp if pooled_only else y


@_timed("roi_align_backward")
def roi_align_backward(dy, rois, roi_start, in_shape, spatial_scale, sampling_ratio, aligned, pooled=False):
    """dy [K,ph,pw,C] -> dx NHWC in_shape.  rois must be grouped by image; roi_start int32 [N+1] prefix offsets.
    ``pooled``: dy is the gradient of AvgPool2d(2) of the crops (the RoIAlign grid is 2ph x 2pw)."""
    require_cuda(dy, rois, roi_start)
    N, H, W, C = in_shape
    K, ph, pw, _ = dy.shape
    assert roi_start.dtype == torch.int32 and roi_start.numel() == N + 1 and dy.is_contiguous()
    dx = torch.empty(in_shape, device=dy.device, dtype=dy.dtype)
    ay = workspace("roi_ay", max(K, 1) * H * ph * 4, dy.device)
    ax = workspace("roi_ax", max(K, 1) * W * pw * 4, dy.device)
    fp = workspace("roi_fp", max(K, 1) * 16, dy.device)
    fn = _L().cddmsl_roi_align_backward_pooled if pooled else _L().cddmsl_roi_align_backward
    check(fn(ptr(dy), ptr(rois), ptr(roi_start), ptr(dx), ptr(ay), ptr(ax), ptr(fp), N, C, H, W, K,
             ph, pw, spatial_scale, sampling_ratio, int(aligned), _dt(dy), stream_ptr()), "cddmsl_roi_align_backward")
    return dx


def roi_align_nchw(input, rois, output_size, spatial_scale, sampling_ratio, aligned):
    """torchvision.ops.roi_align as layers/roi_align.py:58-65 calls it: input [N,C,H,W], rois [K,5] in any order -> [K,C,ph,pw]"""
    require_cuda(input, rois)
    assert rois.dim() == 2 and rois.size(1) == 5 and rois.dtype == torch.float32 and rois.is_contiguous() and input.is_contiguous()
    N, C, H, W = input.shape
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    K = rois.shape[0]
    out = torch.zeros((K, C, ph, pw), device=input.device, dtype=input.dtype)
    nbytes = ctypes.c_size_t(0)
    args = (N, C, H, W, K, ph, pw, spatial_scale, sampling_ratio, int(aligned), _dt(input))
    check(_L().cddmsl_roi_align_nchw_anyorder(ptr(input), ptr(rois), ptr(out), *args, None, ctypes.byref(nbytes), stream_ptr()), "cddmsl_roi_align_nchw_anyorder(size)")
    ws = workspace("roi_nchw", max(nbytes.value, 1), input.device)
    check(_L().cddmsl_roi_align_nchw_anyorder(ptr(input), ptr(rois), ptr(out), *args, ptr(ws), ctypes.byref(nbytes), stream_ptr()), "cddmsl_roi_align_nchw_anyorder")
    return out


def roi_align_backward_nchw(grad, rois, input_shape, spatial_scale, sampling_ratio, aligned):
    """its backward: grad [K,C,ph,pw] -> grad_input [N,C,H,W]"""
    require_cuda(grad, rois)
    assert rois.dtype == torch.float32 and rois.is_contiguous() and grad.is_contiguous()
    N, C, H, W = input_shape
    K, _, ph, pw = grad.shape
    dx = torch.zeros(tuple(input_shape), device=grad.device, dtype=grad.dtype)
    nbytes = ctypes.c_size_t(0)
    args = (N, C, H, W, K, ph, pw, spatial_scale, sampling_ratio, int(aligned), _dt(grad))
    check(_L().cddmsl_roi_align_backward_nchw_anyorder(ptr(grad), ptr(rois), ptr(dx), *args, None, ctypes.byref(nbytes), stream_ptr()), "cddmsl_roi_align_backward_nchw_anyorder(size)")
    ws = workspace("roi_nchw_b", max(nbytes.value, 1), grad.device)
    check(_L().cddmsl_roi_align_backward_nchw_anyorder(ptr(grad), ptr(rois), ptr(dx), *args, ptr(ws), ctypes.byref(nbytes), stream_ptr()), "cddmsl_roi_align_backward_nchw_anyorder")
    return dx


# ------------------------------------------------------------------------------------------------ boxes
def anchors(cell, Hf, Wf, stride, offset):
    require_cuda(cell)
    A = cell.shape[0]
    out = torch.empty((Hf * Wf * A, 4), device=cell.device, dtype=torch.float32)
    check(_L().cddmsl_anchors(ptr(cell), ptr(out), Hf, Wf, A, stride, offset, stream_ptr()), "cddmsl_anchors")
    return out


_WS = {}


def workspace(key, nbytes, device):
    """Persistent scratch buffers (NMS masks, sort temp storage ...): allocated once per (key, size) and reused, so the
    hot loop never goes back to the allocator for its 100+ MB workspaces."""
    k = (key, str(device))
    buf = _WS.get(k)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 16), device=device, dtype=torch.uint8)
        _WS[k] = buf
    return buf


@_timed("sort_desc")
def sort_desc(keys):
    """Stable descending sort of each row of keys [N,total] f32 -> (sorted keys, int32 order)."""
    require_cuda(keys)
    N, total = keys.shape
    assert keys.dtype == torch.float32 and keys.is_contiguous()
    dev = keys.device
    offs = torch.arange(0, (N + 1) * total, total, device=dev, dtype=torch.int32)
    keys_out = torch.empty_like(keys)
    idx = torch.empty((N, total), device=dev, dtype=torch.int32)
    order = torch.empty((N, total), device=dev, dtype=torch.int32)
    nbytes = ctypes.c_size_t(0)
    check(_L().cddmsl_sort_desc(ptr(keys), ptr(keys_out), ptr(idx), ptr(order), ptr(offs), N, total, None, ctypes.byref(nbytes),
                                stream_ptr()), "cddmsl_sort_desc(size)")
    ws = workspace("sort", nbytes.value, dev)
    check(_L().cddmsl_sort_desc(ptr(keys), ptr(keys_out), ptr(idx), ptr(order), ptr(offs), N, total, ptr(ws), ctypes.byref(nbytes),
                                stream_ptr()), "cddmsl_sort_desc")
    return keys_out, order


@_timed("nms")
def nms_anyorder(boxes, scores, iou_threshold):
    """torchvision.ops.nms(boxes [K,4], scores [K], thr) -> (keep int64 [K] (kept indices, descending score, then -1), nkeep int32 [1])"""
    require_cuda(boxes, scores)
    K = boxes.shape[0]
    assert boxes.dtype == scores.dtype == torch.float32 and boxes.is_contiguous() and scores.is_contiguous() and scores.numel() == K
    keep = torch.empty(K, device=boxes.device, dtype=torch.int64)
    nkeep = torch.zeros(1, device=boxes.device, dtype=torch.int32)
    nbytes = ctypes.c_size_t(0)
    check(_L().cddmsl_nms_anyorder(ptr(boxes), ptr(scores), ptr(keep), ptr(nkeep), K, iou_threshold, None, ctypes.byref(nbytes), stream_ptr()),
          "cddmsl_nms_anyorder(size)")
    ws = workspace("nms_any", max(nbytes.value, 1), boxes.device)
    check(_L().cddmsl_nms_anyorder(ptr(boxes), ptr(scores), ptr(keep), ptr(nkeep), K, iou_threshold, ptr(ws), ctypes.byref(nbytes), stream_ptr()),
          "cddmsl_nms_anyorder")
    return keep, nkeep


@_timed("rpn_decode")
def rpn_decode(order, deltas, cell, img_hw, Hf, Wf, topk, stride, offset, weights, scale_clamp, min_size):
    """Decode + clip the top-k sorted anchors of every image -> boxes [N,topk,4] f32, valid u8 [N,topk]."""
    require_cuda(order, deltas, cell, img_hw)
    N = deltas.shape[0]
    A = cell.shape[0]
    assert deltas.dtype == torch.float32 and deltas.is_contiguous() and img_hw.dtype == torch.int32
    boxes = torch.empty((N, topk, 4), device=deltas.device, dtype=torch.float32)
    valid = torch.empty((N, topk), device=deltas.device, dtype=torch.uint8)
    check(_L().cddmsl_rpn_decode(ptr(order), ptr(deltas), ptr(cell), ptr(img_hw), ptr(boxes), ptr(valid), N, Hf, Wf, A, topk,
                                  stride, offset, *[float(w) for w in weights], scale_clamp, min_size, stream_ptr()),
          "cddmsl_rpn_decode")
    return boxes, valid


@_timed("nms")
def nms(boxes, valid, thr, max_keep):
    """boxes [N,n,4] f32 score-descending, valid u8 [N,n] -> keep int32 [N,max_keep] (positions), nkeep int32 [N]."""
    require_cuda(boxes, valid)
    N, n, _ = boxes.shape
    assert boxes.dtype == torch.float32 and boxes.is_contiguous() and valid.dtype == torch.uint8 and valid.is_contiguous()
    nw = (n + 63) // 64
    mask = workspace("nms_mask", max(N * n * nw, 1) * 8, boxes.device)
    keep = torch.full((N, max_keep), -1, device=boxes.device, dtype=torch.int32)
    nkeep = torch.zeros(N, device=boxes.device, dtype=torch.int32)
    check(_L().cddmsl_nms(ptr(boxes), ptr(valid), ptr(mask), ptr(keep), ptr(nkeep), N, n, thr, max_keep, stream_ptr()), "cddmsl_nms")
    return keep, nkeep


@_timed("iou_match")
def iou_match(gt, preds, thresholds, labels, allow_low_quality, out_matches=None, out_labels=None):
    """Fused pairwise_iou + Matcher for one image: gt [G,4], preds [P,4] -> (matches int64 [P], labels int8 [P]).
    ``out_*``: optional contiguous destination rows (callers that batch several images write into slices of one tensor)."""
    require_cuda(gt, preds)
    G, P = gt.shape[0], preds.shape[0]
    assert gt.dtype == preds.dtype == torch.float32 and gt.is_contiguous() and preds.is_contiguous()
    matches = torch.empty(P, device=preds.device, dtype=torch.int64) if out_matches is None else out_matches
    lab = torch.empty(P, device=preds.device, dtype=torch.int8) if out_labels is None else out_labels
    assert matches.dtype == torch.int64 and lab.dtype == torch.int8 and matches.is_contiguous() and lab.is_contiguous()
    assert matches.numel() == P and lab.numel() == P
    best = torch.empty(max(G, 1), device=preds.device, dtype=torch.int32)
    nthr = len(thresholds)
    t0 = float(thresholds[0])
    t1 = float(thresholds[1]) if nthr > 1 else 0.0
    l = list(labels) + [0]
    check(_L().cddmsl_iou_match(ptr(gt), G, ptr(preds), P, ptr(matches), ptr(lab), ptr(best), nthr, t0, t1, l[0], l[1], l[2],
                                 int(allow_low_quality), stream_ptr()), "cddmsl_iou_match")
    return matches, lab


@_timed("iou_match")
def iou_match_batched(gt_list, preds, pred_counts, thresholds, labels, allow_low_quality):
    """Fused pairwise_iou + Matcher for ALL images in one launch pair.  ``gt_list``: per-image [G_i, 4] f32 tensors;
    ``preds``: [P, 4] shared by every image (``pred_counts`` None; returns [N, P] tensors) or the images' predictions
    concatenated [sum P_i, 4] with ``pred_counts`` = [P_i] (returns [sum P_i] tensors)."""
    from ._lib import to_device_async
    require_cuda(preds, *gt_list)
    dev = preds.device
    N = len(gt_list)
    ng = [int(g.shape[0]) for g in gt_list]
    total_g = sum(ng)
    gt = torch.cat(gt_list).float().contiguous() if total_g else torch.zeros((1, 4), device=dev)
    gt_off = to_device_async(torch.tensor([0] + ng, dtype=torch.int32).cumsum(0).to(torch.int32), dev)
    assert preds.dtype == torch.float32 and preds.is_contiguous()
    if pred_counts is None:
        P, pred_off, shape = preds.shape[0], None, (N, preds.shape[0])
    else:
        P, shape = max(pred_counts) if pred_counts else 0, (sum(pred_counts),)
        pred_off = to_device_async(torch.tensor([0] + list(pred_counts), dtype=torch.int32).cumsum(0).to(torch.int32), dev)
    matches = torch.empty(shape, device=dev, dtype=torch.int64)
    lab = torch.empty(shape, device=dev, dtype=torch.int8)
    best = torch.empty(max(total_g, 1), device=dev, dtype=torch.int32)
    nthr = len(thresholds)
    t0 = float(thresholds[0])
    t1 = float(thresholds[1]) if nthr > 1 else 0.0
    l = list(labels) + [0]
    check(_L().cddmsl_iou_match_batched(ptr(gt), ptr(gt_off), ptr(preds), ptr(pred_off), ptr(matches), ptr(lab), ptr(best), N, P,
                                         max(ng) if ng else 0, total_g, nthr, t0, t1, l[0], l[1], l[2], int(allow_low_quality),
                                         stream_ptr()), "cddmsl_iou_match_batched")
    return matches, lab


# ------------------------------------------------------------------------------------------------ attention pool / losses
def l2norm_fwd(x, eps):
    require_cuda(x)
    R, D = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    y = torch.empty_like(x)
    inv = torch.empty(R, device=x.device, dtype=torch.float32)
    check(_L().cddmsl_l2norm_fwd(ptr(x), ptr(y), ptr(inv), R, D, eps, stream_ptr()), "cddmsl_l2norm_fwd")
    return y, inv


def l2norm_bwd(dy, y, inv):
    require_cuda(dy, y, inv)
    dx = torch.empty_like(y)
    check(_L().cddmsl_l2norm_bwd(ptr(dy.contiguous()), ptr(y), ptr(inv), ptr(dx), y.shape[0], y.shape[1], stream_ptr()), "cddmsl_l2norm_bwd")
    return dx


def cosine_logits_fwd(x, wn, temperature, eps=1e-12):
    require_cuda(x, wn)
    R, D = x.shape
    Kc = wn.shape[0]
    assert x.dtype == wn.dtype == torch.float32 and x.is_contiguous() and wn.is_contiguous()
    scores = torch.empty((R, Kc + 1), device=x.device, dtype=torch.float32)
    inv = torch.empty(R, device=x.device, dtype=torch.float32)
    check(_L().cddmsl_cosine_logits_fwd(ptr(x), ptr(wn), ptr(scores), ptr(inv), R, D, Kc, temperature, eps, stream_ptr()),
          "cddmsl_cosine_logits_fwd")
    return scores, inv


def cosine_logits_bwd(ds, x, wn, inv, temperature, dx=None):
    require_cuda(ds, x, wn, inv, dx)
    R, D = x.shape
    acc = dx is not None
    if dx is None:
        dx = torch.empty_like(x)
    check(_L().cddmsl_cosine_logits_bwd(ptr(ds.contiguous()), ptr(x), ptr(wn), ptr(inv), ptr(dx), R, D, wn.shape[0], temperature, int(acc),
                                         stream_ptr()), "cddmsl_cosine_logits_bwd")
    return dx


def contrastive_fwd(S):
    require_cuda(S)
    n = S.shape[0]
    assert S.dtype == torch.float32 and S.shape == (n, n) and S.is_contiguous()
    rl = torch.empty(n, device=S.device, dtype=torch.float32)
    cl = torch.empty(n, device=S.device, dtype=torch.float32)
    loss = torch.empty(1, device=S.device, dtype=torch.float32)
    check(_L().cddmsl_contrastive_fwd(ptr(S), ptr(rl), ptr(cl), ptr(loss), n, n, stream_ptr()), "cddmsl_contrastive_fwd")
    return loss, rl, cl


def contrastive_bwd(S, rl, cl, gloss):
    require_cuda(S, rl, cl, gloss)
    n = S.shape[0]
    dS = torch.empty_like(S)
    check(_L().cddmsl_contrastive_bwd(ptr(S), ptr(rl), ptr(cl), ptr(gloss.reshape(1).float().contiguous()), ptr(dS), n, n, stream_ptr()),
          "cddmsl_contrastive_bwd")
    return dS


@_timed("layernorm")
def layernorm_fwd(x, gamma, beta, out_dtype, eps=1e-5):
    """x [R,D] f32 -> (y [R,D] out_dtype, mean [R], rstd [R])"""
    require_cuda(x, gamma, beta)
    R, D = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    y = torch.empty((R, D), device=x.device, dtype=out_dtype)
    mean = torch.empty(R, device=x.device, dtype=torch.float32)
    rstd = torch.empty(R, device=x.device, dtype=torch.float32)
    check(_L().cddmsl_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), R, D, eps, DT[out_dtype], stream_ptr()),
          "cddmsl_layernorm_fwd")
    return y, mean, rstd


@_timed("layernorm")
def layernorm_bwd(dy, x, gamma, mean, rstd, accumulate_into=None):
    """dx = LN'(dy); with ``accumulate_into`` (f32 [R,D], contiguous) the result is ADDED to that tensor in place and it is returned"""
    require_cuda(dy, x, gamma, mean, rstd, accumulate_into)
    R, D = x.shape
    dy = dy.contiguous()
    if accumulate_into is None:
        dx, acc = torch.empty_like(x), 0
    else:
        assert accumulate_into.dtype == torch.float32 and accumulate_into.is_contiguous() and accumulate_into.shape == x.shape
        dx, acc = accumulate_into, 1
    check(_L().cddmsl_layernorm_bwd(ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), R, D, acc, _dt(dy), stream_ptr()),
          "cddmsl_layernorm_bwd")
    return dx


def _attn_views(q, kv, t, heads):
    assert q.dim() == 2 and kv.dim() == 2 and q.is_contiguous() and kv.is_contiguous() and q.dtype == kv.dtype == torch.bfloat16
    R, d = q.shape
    assert kv.shape == (R, 2 * d) and R % t == 0 and d % heads == 0
    return R // t, d, d // heads


@_timed("attn_small")
def attn_small_fwd(q, kv, t, heads, scale):
    """q [n*t, d], kv [n*t, 2d] (keys | values), bf16 -> o [n*t, d] = softmax(q_h k_h^T * scale) v_h per (sequence, head)"""
    require_cuda(q, kv)
    n, d, dh = _attn_views(q, kv, t, heads)
    o = torch.empty_like(q)
    vptr = c_void_p(kv.data_ptr() + d * 2)
    check(_L().cddmsl_attn_small_fwd(ptr(q), ptr(kv), vptr, ptr(o), n, t, heads, dh, d, 2 * d, 2 * d, d, float(scale), 0, stream_ptr()),
          "cddmsl_attn_small_fwd")
    return o


@_timed("attn_last")
def attn_last_fwd(q, kv, t, heads, scale):
    """one query row per sequence: q [n, d], kv [n*t, 2d] (keys | values) bf16 -> (o [n, d] bf16, p [n, heads, t] f32)"""
    require_cuda(q, kv)
    n, d = q.shape
    assert q.dtype == kv.dtype == torch.bfloat16 and q.is_contiguous() and kv.is_contiguous() and tuple(kv.shape) == (n * t, 2 * d)
    o = torch.empty((n, d), device=q.device, dtype=q.dtype)
    p = torch.empty((n, heads, t), device=q.device, dtype=torch.float32)
    check(_L().cddmsl_attn_last_fwd(ptr(q), ptr(kv), ptr(o), ptr(p), n, t, heads, d // heads, d, 2 * d, d, d, float(scale), 0, stream_ptr()),
          "cddmsl_attn_last_fwd")
    return o, p


@_timed("attn_last")
def attn_last_bwd(q, kv, do, p, t, heads, scale):
    """-> (dq [n, d], dkv [n*t, 2d]) bf16"""
    require_cuda(q, kv, do, p)
    n, d = q.shape
    do = do.contiguous()
    assert do.shape == (n, d) and do.dtype == q.dtype and p.is_contiguous()
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    check(_L().cddmsl_attn_last_bwd(ptr(q), ptr(kv), ptr(do), ptr(p), ptr(dq), ptr(dkv), n, t, heads, d // heads, d, 2 * d, d, d, float(scale), 0,
                                    stream_ptr()), "cddmsl_attn_last_bwd")
    return dq, dkv


@_timed("attn_small")
def attn_small_fwd_qkv(qkv, t, heads, scale):
    """qkv [n*t, 3d] (queries | keys | values: ONE projection GEMM's output), bf16 -> o [n*t, d]"""
    require_cuda(qkv)
    assert qkv.dim() == 2 and qkv.is_contiguous() and qkv.dtype == torch.bfloat16 and qkv.shape[1] % 3 == 0 and qkv.shape[0] % t == 0
    R, d = qkv.shape[0], qkv.shape[1] // 3
    o = torch.empty((R, d), device=qkv.device, dtype=qkv.dtype)
    base = qkv.data_ptr()
    check(_L().cddmsl_attn_small_fwd(c_void_p(base), c_void_p(base + d * 2), c_void_p(base + 4 * d), ptr(o), R // t, t, heads, d // heads,
                                     3 * d, 3 * d, 3 * d, d, float(scale), 0, stream_ptr()), "cddmsl_attn_small_fwd")
    return o


@_timed("attn_small")
def attn_small_bwd_qkv(qkv, do, t, heads, scale):
    """-> dqkv [n*t, 3d] bf16 (the three gradients side by side: ONE input-gradient GEMM follows)"""
    require_cuda(qkv, do)
    R, d = qkv.shape[0], qkv.shape[1] // 3
    do = do.contiguous()
    assert do.shape == (R, d) and do.dtype == qkv.dtype
    dqkv = torch.empty_like(qkv)
    base, gbase = qkv.data_ptr(), dqkv.data_ptr()
    check(_L().cddmsl_attn_small_bwd(c_void_p(base), c_void_p(base + d * 2), c_void_p(base + 4 * d), ptr(do), c_void_p(gbase),
                                     c_void_p(gbase + d * 2), c_void_p(gbase + 4 * d), R // t, t, heads, d // heads, 3 * d, 3 * d, 3 * d, d,
                                     float(scale), 0, stream_ptr()), "cddmsl_attn_small_bwd")
    return dqkv


@_timed("attn_small")
def attn_small_bwd(q, kv, do, t, heads, scale):
    """-> (dq [n*t, d], dkv [n*t, 2d]) bf16"""
    require_cuda(q, kv, do)
    n, d, dh = _attn_views(q, kv, t, heads)
    do = do.contiguous()
    assert do.shape == q.shape and do.dtype == q.dtype
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    vptr, dvptr = c_void_p(kv.data_ptr() + d * 2), c_void_p(dkv.data_ptr() + d * 2)
    check(_L().cddmsl_attn_small_bwd(ptr(q), ptr(kv), vptr, ptr(do), ptr(dq), ptr(dkv), dvptr, n, t, heads, dh, d, 2 * d, 2 * d, d,
                                     float(scale), 0, stream_ptr()), "cddmsl_attn_small_bwd")
    return dq, dkv


def focal_ce_fwd(logits, target, gamma, bg_class, bg_weight):
    require_cuda(logits, target)
    R, C = logits.shape
    assert logits.dtype == torch.float32 and logits.is_contiguous() and target.dtype == torch.int64
    row = torch.empty(R, device=logits.device, dtype=torch.float32)
    probs = torch.empty_like(logits)
    check(_L().cddmsl_focal_ce_fwd(ptr(logits), ptr(target), ptr(row), ptr(probs), R, C, gamma, bg_class, bg_weight, stream_ptr()),
          "cddmsl_focal_ce_fwd")
    return row, probs


def focal_ce_bwd(logits, target, probs, gscale, gamma, bg_class, bg_weight):
    require_cuda(logits, target, probs, gscale)
    d = torch.empty_like(logits)
    check(_L().cddmsl_focal_ce_bwd(ptr(logits), ptr(target), ptr(probs), ptr(gscale.reshape(1).float().contiguous()), ptr(d), logits.shape[0],
                                   logits.shape[1], gamma, bg_class, bg_weight, stream_ptr()), "cddmsl_focal_ce_bwd")
    return d


def maxpool3s2_fwd(x):
    """F.max_pool2d(x, 3, 2, 1) on NHWC"""
    require_cuda(x)
    N, H, W, C = x.shape
    y = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), device=x.device, dtype=x.dtype)
    check(_L().cddmsl_maxpool3s2_fwd(ptr(x), ptr(y), N, H, W, C, _dt(x), stream_ptr()), "cddmsl_maxpool3s2_fwd")
    return y


def upsample_zero2(t, in_shape, mask=None, add=None):
    """dx[:, ::2, ::2] = t, zero elsewhere (+ add), zeroed where mask <= 0: input gradient of a stride-2 1x1 conv."""
    require_cuda(t, mask, add)
    N, H, W, C = in_shape
    assert t.is_contiguous() and tuple(t.shape) == (N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C)
    dx = torch.empty(in_shape, device=t.device, dtype=t.dtype)
    check(_L().cddmsl_upsample_zero2(ptr(t), ptr(mask), ptr(add), ptr(dx), N, H, W, C, _dt(t), stream_ptr()), "cddmsl_upsample_zero2")
    return dx


def meanpool_fwd(x):
    """x [K,P,C] (T) -> [K,C] f32"""
    require_cuda(x)
    K, P, C = x.shape
    y = torch.empty((K, C), device=x.device, dtype=torch.float32)
    check(_L().cddmsl_meanpool_fwd(ptr(x), ptr(y), K, P, C, _dt(x), stream_ptr()), "cddmsl_meanpool_fwd")
    return y


def meanpool_bwd(dy, P, dtype):
    require_cuda(dy)
    K, C = dy.shape
    dx = torch.empty((K, P, C), device=dy.device, dtype=dtype)
    check(_L().cddmsl_meanpool_bwd(ptr(dy.contiguous().float()), ptr(dx), K, P, C, DT[dtype], stream_ptr()), "cddmsl_meanpool_bwd")
    return dx

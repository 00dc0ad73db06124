"""Thin functional wrappers over the C-ABI HIP library: shape checks on the host, raw pointers down.

Every function here requires CUDA (=HIP) tensors and enqueues on torch's current stream.  No
fallbacks: a CPU tensor or a missing library raises ``HipLibraryError``.
"""
import ctypes
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
        L.cddmsl_conv_fwd.argtypes = [vp] * 7 + [ci] * 16 + [vp]
        L.cddmsl_conv_wgrad.argtypes = [vp] * 4 + [ci] * 12 + [vp]
        L.cddmsl_weight_prep.argtypes = [vp] * 4 + [ci] * 5 + [vp]
        for name in dir(L):
            pass
        _sigs_done = True
    return L


def _dt(t):
    if t.dtype not in DT:
        raise TypeError(f"unsupported dtype {t.dtype}: the kernels compute in bf16 or f32")
    return DT[t.dtype]


def conv_fwd(x, w, scale=None, bias=None, residual=None, relu=False, relu_mask=None, stride=1, pad=0,
             pool=False, out_f32=False):
    """x NHWC [N,H,W,Cin]; w [Cout,KH,KW,Cin] (same dtype).  Returns NHWC [N,Ho,Wo,Cout].
    y = relu?(acc*scale[n] + bias[n] + residual), zeroed where relu_mask <= 0 (ReLU backward).
    pool=True: 1x1 conv over the 2x2 average-pooled input (AvgPool2d(2) fused into the loader)."""
    require_cuda(x, w, scale, bias, residual, relu_mask)
    assert x.dim() == 4 and w.dim() == 4 and x.is_contiguous() and w.is_contiguous()
    assert x.dtype == w.dtype
    N, H, W, Cin = x.shape
    Cout, KH, KW, Cin2 = w.shape
    assert Cin == Cin2, (x.shape, w.shape)
    if pool:
        Ho, Wo = H // 2, W // 2
    else:
        Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    y = torch.empty((N, Ho, Wo, Cout), device=x.device, dtype=torch.float32 if out_f32 else x.dtype)
    for v in (scale, bias):
        assert v is None or (v.dtype == torch.float32 and v.numel() == Cout and v.is_contiguous())
    for v in (residual, relu_mask):
        assert v is None or (v.dtype == x.dtype and v.is_contiguous() and v.numel() == y.numel())
    st = _L().cddmsl_conv_fwd(ptr(x), ptr(w), ptr(y), ptr(scale), ptr(bias), ptr(residual), ptr(relu_mask),
                              N, H, W, Cin, Cout, KH, KW, stride, pad, int(pool), Cout, Cout, Cout,
                              int(relu), int(out_f32), _dt(x), stream_ptr())
    check(st, "cddmsl_conv_fwd")
    return y


def linear_fwd(x, w, scale=None, bias=None, residual=None, relu=False, relu_mask=None, out_f32=False):
    """x [M,K] @ w[N,K]^T with the conv epilogue (a 1x1 conv over M 'pixels')."""
    M, K = x.shape
    y = conv_fwd(x.view(1, 1, M, K), w.view(w.shape[0], 1, 1, K), scale, bias,
                 None if residual is None else residual.view(1, 1, M, -1), relu,
                 None if relu_mask is None else relu_mask.view(1, 1, M, -1), out_f32=out_f32)
    return y.view(M, w.shape[0])


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
    st = _L().cddmsl_conv_wgrad(ptr(x), ptr(dy), ptr(out), ptr(scale), N, H, W, Cin, Cout, KH, KW, stride, pad,
                                int(pool), Cout, _dt(x), stream_ptr())
    check(st, "cddmsl_conv_wgrad")
    return out


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

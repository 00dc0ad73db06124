"""GPU parity: implicit-GEMM conv fwd / dgrad / wgrad (C-ABI) vs ATen CPU fp32 (the oracle's conv)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


CASES = [
    # N, H, W, Cin, Cout, K, stride, pad
    (2, 9, 13, 64, 96, 3, 1, 1),
    (1, 14, 14, 128, 256, 1, 1, 0),
    (3, 7, 5, 32, 40, 3, 1, 1),
    (2, 16, 20, 8, 32, 3, 2, 1),
    (1, 50, 83, 256, 136, 1, 1, 0),
]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_dgrad_wgrad(case, dtype, tol):
    from cddmsl_amd import hip
    N, H, W, Cin, Cout, K, s, p = case
    x = _rand((N, Cin, H, W), 1).to(dtype).float()
    w = (_rand((Cout, Cin, K, K), 2) * (Cin * K * K) ** -0.5).to(dtype).float()
    scale = torch.rand(Cout, generator=torch.Generator().manual_seed(3)) + 0.5
    bias = _rand((Cout,), 4, 0.1)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    conv = F.conv2d(xr, wr, stride=s, padding=p)
    res = _rand(tuple(conv.shape), 5).to(dtype).float()
    y_ref = F.relu(conv * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1) + res)
    dy = _rand(tuple(conv.shape), 6).to(dtype).float()
    y_ref.backward(dy)

    dev = "cuda"
    xg = _nhwc(x).to(dev, dtype)
    w_ohwi = w.permute(0, 2, 3, 1).contiguous().to(dev)
    wf, wd = hip.weight_prep(w_ohwi, scale.to(dev), dtype)
    y = hip.conv_fwd(xg, wf, scale.to(dev), bias.to(dev), _nhwc(res).to(dev, dtype), relu=True, stride=s, pad=p)
    y_cpu = y.float().cpu().permute(0, 3, 1, 2)
    err = (y_cpu - y_ref.detach()).abs().max() / y_ref.detach().abs().max()
    assert err < tol, f"fwd err {err}"

    # backward of relu(bn(conv)+res): dpre = dy * (y>0); wgrad uses scale in its epilogue, dgrad in its weights
    dyg = _nhwc(dy).to(dev, dtype)
    dpre = (dyg.float() * (y.float() > 0)).to(dtype)
    dw = hip.conv_wgrad(xg, dpre, (Cout, K, K, Cin), scale.to(dev), stride=s, pad=p)
    dw_cpu = dw.cpu().permute(0, 3, 1, 2)
    # reference with the GPU's own relu mask (bf16 rounding can flip y>0 at the boundary)
    mask = (y_cpu > 0).float()
    xr2 = x.clone().requires_grad_(True)
    wr2 = w.clone().requires_grad_(True)
    c2 = F.conv2d(xr2, wr2, stride=s, padding=p)
    (c2 * scale.view(1, -1, 1, 1)).backward(dy * mask)
    errw = (dw_cpu - wr2.grad).abs().max() / wr2.grad.abs().max()
    assert errw < tol, f"wgrad err {errw}"
    if s == 1:
        dx = hip.conv_fwd(dpre, wd, stride=1, pad=K - 1 - p)
        dx_cpu = dx.float().cpu().permute(0, 3, 1, 2)
        errx = (dx_cpu - xr2.grad).abs().max() / xr2.grad.abs().max()
        assert errx < tol, f"dgrad err {errx}"


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_conv_pool_fused(dtype, tol):
    """AvgPool2d(2) + 1x1 conv (clip_backbone.py:64-65) fused into the A-loader, odd sizes floor."""
    from cddmsl_amd import hip
    N, H, W, Cin, Cout = 2, 15, 21, 64, 160
    x = _rand((N, Cin, H, W), 11).to(dtype).float().requires_grad_(True)
    w = (_rand((Cout, Cin, 1, 1), 12) * Cin ** -0.5).to(dtype).float().requires_grad_(True)
    y_ref = F.conv2d(F.avg_pool2d(x, 2), w)
    dy = _rand(tuple(y_ref.shape), 13).to(dtype).float()
    y_ref.backward(dy)
    xg = _nhwc(x.detach()).cuda().to(dtype)
    wf, _ = hip.weight_prep(w.detach().permute(0, 2, 3, 1).contiguous().cuda(), None, dtype)
    y = hip.conv_fwd(xg, wf, pool=True)
    assert y.shape == (N, H // 2, W // 2, Cout)
    err = (y.float().cpu().permute(0, 3, 1, 2) - y_ref.detach()).abs().max() / y_ref.abs().max()
    assert err < tol
    dw = hip.conv_wgrad(xg, _nhwc(dy).cuda().to(dtype), (Cout, 1, 1, Cin), None, pool=True)
    errw = (dw.cpu().permute(0, 3, 1, 2) - w.grad).abs().max() / w.grad.abs().max()
    assert errw < tol


def test_linear_tails():
    """M, N, K tails (not multiples of the 128x128x64 tile) incl. f32 output from bf16 inputs."""
    from cddmsl_amd import hip
    for M, N, K in [(1, 21, 1024), (130, 75, 1032), (257, 129, 8), (700, 264, 200)]:
        x = _rand((M, K), 21)
        w = _rand((N, K), 22) * K ** -0.5
        b = _rand((N,), 23)
        ref = F.linear(x, w, b)
        y = hip.linear_fwd(x.cuda(), w.cuda(), None, b.cuda())
        assert (y.cpu() - ref).abs().max() < 2e-5 * max(1.0, float(ref.abs().max()))
        yb = hip.linear_fwd(x.cuda().bfloat16(), w.cuda().bfloat16(), None, b.cuda(), out_f32=True)
        assert yb.dtype == torch.float32
        assert (yb.cpu() - ref).abs().max() < 3e-2 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_batched_gemms(dtype, tol):
    """The strided/batched NT and TN GEMM entry points used by the reassociated attention pool."""
    from cddmsl_amd import hip
    g = torch.Generator().manual_seed(5)
    # NT over "heads": C[:, h*N:(h+1)*N] = A[:, h*Kd:(h+1)*Kd] @ W[h*N:(h+1)*N, :Kd]^T with strided operands
    M, Hh, Kd, N = 70, 3, 64, 40
    A = torch.randn(M, Hh * Kd, generator=g).to(dtype)
    W = torch.randn(Hh * N, Kd, generator=g).to(dtype)
    C = torch.zeros(M, Hh * N, dtype=torch.float32).cuda()
    hip.gemm_nt_batched(A.cuda(), W.cuda(), C, M, N, Kd, Hh * Kd, Kd, Hh * N, Hh, Kd, N * Kd, N)
    ref = torch.cat([A.float()[:, h * Kd:(h + 1) * Kd] @ W.float()[h * N:(h + 1) * N].t() for h in range(Hh)], dim=1)
    assert (C.cpu() - ref).abs().max() < tol * max(1.0, float(ref.abs().max()))
    # NT over "regions": S[b] = U[b] (32 x 128) @ T[b]^T (56 x 128)
    B = 5
    U = torch.randn(B, 32, 128, generator=g).to(dtype)
    Tk = torch.randn(B, 56, 128, generator=g).to(dtype)
    S = torch.zeros(B, 32, 56, dtype=torch.float32).cuda()
    hip.gemm_nt_batched(U.cuda(), Tk.cuda(), S, 32, 56, 128, 128, 128, 56, B, 32 * 128, 56 * 128, 32 * 56)
    ref = torch.bmm(U.float(), Tk.float().transpose(1, 2))
    assert (S.cpu() - ref).abs().max() < tol * max(1.0, float(ref.abs().max()))
    # TN direct store: Z[b] (32 x 128) = P[b]^T-ish: sum_m A[b][m][n] * X[b][m][k]
    Pm = torch.randn(B, 56, 32, generator=g).to(dtype)
    Z = torch.zeros(B, 32, 128, dtype=dtype).cuda()
    hip.gemm_tn_batched(Pm.cuda(), Tk.cuda(), Z, 56, 32, 128, 32, 128, 128, B, 56 * 32, 56 * 128, 32 * 128)
    ref = torch.bmm(Pm.float().transpose(1, 2), Tk.float())
    assert (Z.float().cpu() - ref).abs().max() < tol * max(1.0, float(ref.abs().max()))
    # the same product over many "regions" runs on the streaming kernel (runs of batches per block, DMA pipeline
    # carried across batch boundaries): 203 batches (ragged last run), 56-row and 100-row (two tiles) reductions
    for Bn, Mm in ((203, 56), (77, 100)):
        Pm = torch.randn(Bn, Mm, 32, generator=g).to(dtype)
        Tn = torch.randn(Bn, Mm, 256, generator=g).to(dtype)
        Z = torch.zeros(Bn, 32, 256, dtype=dtype).cuda()
        hip.gemm_tn_batched(Pm.cuda(), Tn.cuda(), Z, Mm, 32, 256, 32, 256, 256, Bn, Mm * 32, Mm * 256, 32 * 256)
        ref = torch.bmm(Pm.float().transpose(1, 2), Tn.float())
        assert (Z.float().cpu() - ref).abs().max() < tol * max(1.0, float(ref.abs().max()))
    # one-tile reductions with a narrow output take the compact three-stage ring (bf16): every row-group count of the
    # template (N = 8..64), a 3-row reduction, bf16 / f32 stores and f32 accumulation, runs shorter than the ring
    if dtype == torch.bfloat16:
        for Bn, Mm, Nn, Kk in ((150, 64, 56, 384), (64, 3, 8, 128), (201, 50, 64, 256), (90, 33, 24, 128), (67, 17, 40, 128),
                               (129, 64, 48, 256), (70, 9, 16, 128)):
            Pm = torch.randn(Bn, Mm, Nn, generator=g).to(dtype)
            Tn = torch.randn(Bn, Mm, Kk, generator=g).to(dtype)
            ref = torch.bmm(Pm.float().transpose(1, 2), Tn.float())
            Z = torch.zeros(Bn, Nn, Kk, dtype=dtype).cuda()
            hip.gemm_tn_batched(Pm.cuda(), Tn.cuda(), Z, Mm, Nn, Kk, Nn, Kk, Kk, Bn, Mm * Nn, Mm * Kk, Nn * Kk)
            assert hip._L().cddmsl_last_kernel() == 9
            assert (Z.float().cpu() - ref).abs().max() < tol * max(1.0, float(ref.abs().max())), (Bn, Mm, Nn, Kk)
            Zf = torch.full((Bn, Nn, Kk), 2.0, dtype=torch.float32).cuda()
            hip.gemm_tn_batched(Pm.cuda(), Tn.cuda(), Zf, Mm, Nn, Kk, Nn, Kk, Kk, Bn, Mm * Nn, Mm * Kk, Nn * Kk)
            assert (Zf.cpu() - ref).abs().max() < 1e-3 * max(1.0, float(ref.abs().max())), (Bn, Mm, Nn, Kk)
            hip.gemm_tn_batched(Pm.cuda(), Tn.cuda(), Zf, Mm, Nn, Kk, Nn, Kk, Kk, Bn, Mm * Nn, Mm * Kk, Nn * Kk, accumulate=True)
            assert (Zf.cpu() - 2 * ref).abs().max() < 2e-3 * max(1.0, float(ref.abs().max())), (Bn, Mm, Nn, Kk)
    # TN accumulate (f32 atomics, many row tiles, split over blocks): dW[h] += A[:, h]^T @ X[:, h]
    Mr = 1500
    Ar = torch.randn(Mr, Hh * 64, generator=g).to(dtype)
    Xr = torch.randn(Mr, Hh * 128, generator=g).to(dtype)
    out = torch.ones(Hh * 64, 128, dtype=torch.float32).cuda()
    hip.gemm_tn_batched(Ar.cuda(), Xr.cuda(), out, Mr, 64, 128, Hh * 64, Hh * 128, 128, Hh, 64, 128, 64 * 128, accumulate=True)
    ref = 1.0 + torch.cat([Ar.float()[:, h * 64:(h + 1) * 64].t() @ Xr.float()[:, h * 128:(h + 1) * 128] for h in range(Hh)], dim=0)
    assert (out.cpu() - ref).abs().max() < tol * max(1.0, float(ref.abs().max()))


CASES256 = [
    # N, H, W, Cin, Cout, K, pad      (Cout % 256 == 0, Cin a whole number of K-tiles)
    (2, 19, 23, 64, 256, 3, 1),       # ragged M (874 rows), 9 taps, one K-tile per tap (bf16)
    (1, 30, 33, 128, 512, 1, 0),      # two column tiles, 2 K-tiles
    (3, 14, 14, 192, 256, 3, 1),      # odd number of K-tiles (27)
    (1, 20, 20, 64, 256, 1, 0),       # a single K-tile (bf16): prologue-only pipeline
    (4, 28, 28, 256, 768, 3, 1),
]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("case", CASES256)
def test_conv_fwd_256_tile_kernel(case, dtype, tol, monkeypatch):
    """The 256x256 ping-pong kernel (forced with CDDMSL_GEMM256=2) against ATen fp32 and against the 128x128 kernel:
    FrozenBN scale/bias + residual + ReLU forward, and the masked dgrad form."""
    from cddmsl_amd import hip
    N, H, W, Cin, Cout, K, p = case
    x = _rand((N, Cin, H, W), 21).to(dtype).float()
    w = (_rand((Cout, Cin, K, K), 22) * (Cin * K * K) ** -0.5).to(dtype).float()
    scale = torch.rand(Cout, generator=torch.Generator().manual_seed(23)) + 0.5
    bias = _rand((Cout,), 24, 0.1)
    conv = F.conv2d(x, w, padding=p)
    res = _rand(tuple(conv.shape), 25).to(dtype).float()
    y_ref = F.relu(conv * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1) + res)
    dev = "cuda"
    xg = _nhwc(x).to(dev, dtype)
    wf, wd = hip.weight_prep(w.permute(0, 2, 3, 1).contiguous().to(dev), scale.to(dev), dtype)
    out = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("CDDMSL_GEMM256", mode)
        y = hip.conv_fwd(xg, wf, scale.to(dev), bias.to(dev), _nhwc(res).to(dev, dtype), relu=True, stride=1, pad=p)
        # dgrad form: input = a [.., Cout] gradient, weights = wd (Cin outputs ... only legal for the 256 kernel if Cin % 256 == 0)
        msk = hip.conv_fwd(xg, wf, relu_mask=_nhwc(res).to(dev, dtype), stride=1, pad=p)
        torch.cuda.synchronize()
        out[mode] = (y.float().cpu(), msk.float().cpu())
    y_cpu = out["2"][0].permute(0, 3, 1, 2)
    err = (y_cpu - y_ref).abs().max() / y_ref.abs().max()
    assert err < tol, f"fwd err {err}"
    m_ref = conv * (res > 0)
    errm = (out["2"][1].permute(0, 3, 1, 2) - m_ref).abs().max() / m_ref.abs().max()
    assert errm < tol, f"masked err {errm}"
    # the two kernels accumulate K in the same order (chunk by chunk, fp32): results agree to rounding of the store dtype
    for a, b in zip(out["0"], out["2"]):
        assert (a - b).abs().max() <= tol * a.abs().max()


@pytest.mark.parametrize("case", [(1, 66560, 1, 512, 256, 1, 0), (13, 64, 80, 320, 256, 3, 1), (1, 66304, 1, 1024, 1024, 1, 0)])
@pytest.mark.parametrize("flags", [(False, False), (True, False), (True, True)])
def test_conv_fwd_256_tail_tiles_split_along_k(case, flags, monkeypatch):
    """260 row panels on 256 CUs: the main launch of the 256x256 kernel stops at the last full round of workgroups and the
    leftover tiles are computed split along K (raw accumulators through the workspace, k_conv_split_reduce applies the
    epilogue) -- against ATen fp32, and against the unsplit launch (CDDMSL_TAIL_SPLIT=0) to one bf16 rounding of the output
    (same products; the K-slices are summed in a different order).  1x1 and 3x3 (5 K-tiles per tap, slices of 3: they start and end inside taps)."""
    from cddmsl_amd import hip
    res_on, msk_on = flags
    N, H, W, Cin, Cout, K, p = case
    dev, dtype = "cuda", torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(61)
    x = torch.randn(N, H, W, Cin, device=dev, generator=g).to(dtype)
    w = (torch.randn(Cout, K, K, Cin, device=dev, generator=g) * (Cin * K * K) ** -0.5).to(dtype)
    scale = torch.rand(Cout, device=dev, generator=g) + 0.5
    bias = torch.randn(Cout, device=dev, generator=g) * 0.1
    res = torch.randn(N, H, W, Cout, device=dev, generator=g).to(dtype) if res_on else None
    msk = torch.randn(N, H, W, Cout, device=dev, generator=g).to(dtype) if msk_on else None
    hip.ensure_workspace(dev)
    monkeypatch.setenv("CDDMSL_GEMM256", "2")
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("CDDMSL_TAIL_SPLIT", mode)
        out[mode] = hip.conv_fwd(x, w, scale, bias, res, relu=not msk_on, relu_mask=msk, stride=1, pad=p)
        assert hip._L().cddmsl_last_kernel() == 3
    torch.cuda.synchronize()
    a, b = out["0"].float(), out["1"].float()
    assert not torch.equal(a[-1], b[-1]) or K == 1            # (the tail tiles did take the other path: last rows differ in rounding)
    assert (a - b).abs().max() <= 2.0 ** -7 * a.abs().max()
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=p).permute(0, 2, 3, 1) * scale + bias
    if res_on:
        ref = ref + res.float()
    ref = ref * (msk.float() > 0) if msk_on else ref.clamp_min(0)
    err = (b - ref).abs().max() / ref.abs().max()
    assert err < 2e-2, err


CASES_FWD2 = [
    # N, H, W, Cin, Cout, K, pad      (Cout % 128 == 0, Cin a whole number of 32-element K-tiles in bf16 / 16 in f32)
    (2, 19, 23, 32, 128, 3, 1),       # ragged M (874 rows), one K-tile per tap (bf16)
    (1, 30, 33, 96, 384, 1, 0),       # three column tiles, 3 K-tiles
    (3, 14, 14, 160, 128, 3, 1),      # 45 K-tiles
    (1, 20, 20, 32, 256, 1, 0),       # a single K-tile (bf16)
    (2, 16, 16, 64, 128, 1, 0),       # two K-tiles: the short prologue paths
]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("case", CASES_FWD2)
def test_conv_fwd_two_workgroup_kernel(case, dtype, tol, monkeypatch):
    """The 256x128 two-workgroup kernel (k_conv_fwd2, forced with CDDMSL_FWD2=2) against ATen fp32 and -- bit for bit: the same
    products accumulated in the same order -- against the kernels it replaces (CDDMSL_FWD2=0): FrozenBN scale/bias + residual +
    ReLU forward, and the masked input-gradient form."""
    from cddmsl_amd import hip
    N, H, W, Cin, Cout, K, p = case
    x = _rand((N, Cin, H, W), 51).to(dtype).float()
    w = (_rand((Cout, Cin, K, K), 52) * (Cin * K * K) ** -0.5).to(dtype).float()
    scale = torch.rand(Cout, generator=torch.Generator().manual_seed(53)) + 0.5
    bias = _rand((Cout,), 54, 0.1)
    conv = F.conv2d(x, w, padding=p)
    res = _rand(tuple(conv.shape), 55).to(dtype).float()
    y_ref = F.relu(conv * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1) + res)
    dev = "cuda"
    xg = _nhwc(x).to(dev, dtype)
    wf, _ = hip.weight_prep(w.permute(0, 2, 3, 1).contiguous().to(dev), scale.to(dev), dtype)
    out = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("CDDMSL_FWD2", mode)
        y = hip.conv_fwd(xg, wf, scale.to(dev), bias.to(dev), _nhwc(res).to(dev, dtype), relu=True, stride=1, pad=p)
        kern = hip._L().cddmsl_last_kernel()
        msk = hip.conv_fwd(xg, wf, relu_mask=_nhwc(res).to(dev, dtype), stride=1, pad=p)
        plain = hip.conv_fwd(xg, wf, stride=1, pad=p)
        torch.cuda.synchronize()
        assert (kern == 11) == (mode == "2"), (mode, kern)
        out[mode] = (y.cpu(), msk.cpu(), plain.cpu())
    err = (out["2"][0].float().permute(0, 3, 1, 2) - y_ref).abs().max() / y_ref.abs().max()
    assert err < tol, f"fwd err {err}"
    m_ref = conv * (res > 0)
    errm = (out["2"][1].float().permute(0, 3, 1, 2) - m_ref).abs().max() / m_ref.abs().max()
    assert errm < tol, f"masked err {errm}"
    for a, b in zip(out["0"], out["2"]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("flags", [(False, False), (True, False), (False, True), (True, True)])
def test_conv_fwd_256_persistent_form_is_bit_equal(flags, monkeypatch):
    """Short reductions (K <= 512, bf16, no taps) with more tiles than CUs run the PERSISTENT form of the 256x256 kernel (one workgroup
    per CU walking its tiles, CDDMSL_PERSIST=1 = default): every epilogue operand set against ATen fp32, and bit-equal to the
    one-tile-per-workgroup grid (CDDMSL_PERSIST=0).  1 300 tiles: five tiles per workgroup, a ragged last row panel (M = 83 003) and
    a ragged last round."""
    from cddmsl_amd import hip
    res_on, msk_on = flags
    N, H, W, Cin, Cout = 1, 83003, 1, 128, 1024
    dev, dtype = "cuda", torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(31)
    x = torch.randn(N, H, W, Cin, device=dev, generator=g).to(dtype)
    w = (torch.randn(Cout, 1, 1, Cin, device=dev, generator=g) * Cin ** -0.5).to(dtype)
    scale = torch.rand(Cout, device=dev, generator=g) + 0.5
    bias = torch.randn(Cout, device=dev, generator=g) * 0.1
    res = torch.randn(N, H, W, Cout, device=dev, generator=g).to(dtype) if res_on else None
    msk = torch.randn(N, H, W, Cout, device=dev, generator=g).to(dtype) if msk_on else None
    monkeypatch.setenv("CDDMSL_GEMM256", "2")
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("CDDMSL_PERSIST", mode)
        out[mode] = hip.conv_fwd(x, w, scale, bias, res, relu=not msk_on, relu_mask=msk, stride=1, pad=0)
        assert hip._L().cddmsl_last_kernel() == 3
    torch.cuda.synchronize()
    assert torch.equal(out["0"], out["1"])
    ref = (x.float().view(-1, Cin) @ w.float().view(Cout, Cin).t()) * scale + bias
    if res_on:
        ref = ref + res.float().view(-1, Cout)
    if msk_on:
        ref = ref * (msk.float().view(-1, Cout) > 0)
    else:
        ref = ref.clamp_min(0)
    err = (out["1"].float().view(-1, Cout) - ref).abs().max() / ref.abs().max()
    assert err < 2e-2, err


@pytest.mark.parametrize("g256", ["0", "2"])
def test_f32_residual_in_the_bf16_gemm_epilogue(g256, monkeypatch):
    """y (f32) = x (bf16) @ W^T + bias + residual (f32): the mapper's residual stream stays f32 and its add rides in the GEMM
    epilogue (out_f32 bit 1), on the 128x128 and on the 256x256 kernel; ragged M, with and without ReLU-free bias."""
    from cddmsl_amd import hip
    monkeypatch.setenv("CDDMSL_GEMM256", g256)
    g = torch.Generator().manual_seed(71)
    for M, K, N in ((1000, 768, 768), (333, 1536, 768), (2560, 768, 256)):
        x = torch.randn(M, K, generator=g).bfloat16()
        w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16()
        b = torch.randn(N, generator=g) * 0.1
        r = torch.randn(M, N, generator=g) * 3.0
        ref = x.float() @ w.float().t() + b + r
        y = hip.conv_fwd(x.cuda().view(1, 1, M, K), w.cuda().view(N, 1, 1, K), None, b.cuda(), r.cuda().view(1, 1, M, N), out_f32=True)
        assert hip._L().cddmsl_last_kernel() == (3 if g256 == "2" else 1)
        assert y.dtype == torch.float32
        err = (y.view(M, N).cpu() - ref).abs().max()
        assert err < 1e-4 * ref.abs().max() + 2e-2 * (x.float() @ w.float().t()).abs().max() * 2 ** -8, float(err)
    # the autograd wrapper: the residual's gradient is the incoming gradient
    from cddmsl_amd import layers
    M, K, N = 512, 768, 768
    wp = torch.nn.Parameter((torch.randn(N, K, generator=g) * K ** -0.5).cuda())
    pw = layers.PreparedWeight(wp, None, frozen=True)
    x = torch.randn(M, K, generator=g).cuda().bfloat16().requires_grad_(True)
    r = torch.randn(M, N, generator=g).cuda().requires_grad_(True)
    y = layers.linear(x, pw, None, out_f32=True, train_w=False, residual=r)
    gy = torch.randn(M, N, generator=g).cuda()
    y.backward(gy)
    assert torch.equal(r.grad, gy)
    dx_ref = gy.bfloat16().float() @ wp.detach().bfloat16().float()
    assert (x.grad.float() - dx_ref).abs().max() < 2e-2 * dx_ref.abs().max()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("g256", ["0", "2"])
def test_pooled_residual_is_avgpool_backward(g256, dtype, tol, monkeypatch):
    """out_f32 bit 2: the residual operand is a 2x2-average-pooled gradient and every output pixel adds a quarter of its
    pooled pixel -- conv1's dgrad with the downsample path's gradient (clip_backbone.py:45-52,57-70) without materialising
    avgpool2_bwd's output.  Must equal the two-kernel form bit for bit (x 0.25 is exact), on both kernels, even and odd
    sizes (an odd size's last row / column has no pooled pixel), with and without the ReLU mask."""
    from cddmsl_amd import hip
    monkeypatch.setenv("CDDMSL_GEMM256", g256)
    g = torch.Generator().manual_seed(81)
    for (N, H, W, Cin, Cout) in ((40, 14, 14, 64, 256), (3, 15, 21, 128, 256), (2, 50, 84, 64, 512)):
        x = torch.randn(N, H, W, Cin, generator=g).to(dtype).cuda()
        w = (torch.randn(Cout, 1, 1, Cin, generator=g) * Cin ** -0.5).to(dtype).cuda()
        gp = torch.randn(N, H // 2, W // 2, Cout, generator=g).to(dtype).cuda()
        msk = torch.randn(N, H, W, Cout, generator=g).to(dtype).cuda()
        up = hip.avgpool2_bwd(gp, (N, H, W, Cout))
        for m in (None, msk):
            two = hip.conv_fwd(x, w, residual=up, relu_mask=m)
            one = hip.conv_fwd(x, w, residual=gp, relu_mask=m, residual_pooled=True)
            assert hip._L().cddmsl_last_kernel() == (3 if g256 == "2" else 1)
            assert torch.equal(one, two), (N, H, W, m is not None)
        ref = (x.float().view(-1, Cin) @ w.float().view(Cout, Cin).t()).view(N, H, W, Cout)
        ref[:, : H // 2 * 2, : W // 2 * 2] += 0.25 * gp.float().repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
        assert (one.float() * 0 + hip.conv_fwd(x, w, residual=gp, residual_pooled=True).float() - ref).abs().max() < tol * ref.abs().max()


def test_conv3x3_streaming_kernel_64_channels_and_relu_mask():
    """The streaming 3x3 kernel at 8 chunks per pixel (the 64 -> 64 layers of res2, bf16): FrozenBN + ReLU forward, and the
    input-gradient form (no affine, output zeroed where the ReLU mask is <= 0), ragged sizes, vs ATen fp32."""
    from cddmsl_amd import hip
    dtype, tol = torch.bfloat16, 2e-2
    for (N, H, W) in ((2, 37, 53), (1, 200, 333)):
        x = _rand((N, 64, H, W), 91).to(dtype).float()
        w = (_rand((64, 64, 3, 3), 92) * (64 * 9) ** -0.5).to(dtype).float()
        scale = torch.rand(64, generator=torch.Generator().manual_seed(93)) + 0.5
        bias = _rand((64,), 94, 0.1)
        msk = _rand((N, 64, H, W), 95).to(dtype).float()
        conv = F.conv2d(x, w, padding=1)
        wf, _ = hip.weight_prep(w.permute(0, 2, 3, 1).contiguous().cuda(), None, dtype, True, False)
        y = hip.conv_fwd(_nhwc(x).cuda().to(dtype), wf, scale.cuda(), bias.cuda(), relu=True, pad=1)
        assert hip._L().cddmsl_last_kernel() == 8
        ref = F.relu(conv * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1))
        assert (y.float().cpu().permute(0, 3, 1, 2) - ref).abs().max() < tol * ref.abs().max()
        ym = hip.conv_fwd(_nhwc(x).cuda().to(dtype), wf, relu_mask=_nhwc(msk).cuda().to(dtype), pad=1)
        assert hip._L().cddmsl_last_kernel() == 8
        refm = conv * (msk > 0)
        assert (ym.float().cpu().permute(0, 3, 1, 2) - refm).abs().max() < tol * refm.abs().max()
        assert bool((ym.float().cpu().permute(0, 3, 1, 2)[msk <= 0] == 0).all())


CASES_W256 = [
    # N, H, W, Cin, Cout, K, pad     (Cout % 256 == 0, KH*KW*Cin % 256 == 0, Cin % 64 == 0)
    (3, 14, 14, 256, 256, 3, 1),     # M = 588: ragged last reduction tile, 9 taps
    (2, 9, 11, 256, 512, 1, 0),      # 1x1: two n tiles
    (5, 7, 7, 512, 256, 3, 1),       # RoI-head geometry (7x7 images), 18 k tiles
    (1, 5, 5, 256, 256, 3, 1),       # M = 25: a single partial reduction tile
    (9, 14, 14, 512, 512, 3, 1),
]


@pytest.mark.parametrize("case", CASES_W256)
def test_conv_wgrad_256_tile_kernel(case, monkeypatch):
    """bf16 wgrad on the 256x256 ping-pong kernel (forced) vs ATen fp32 and vs the 128x128 LDS-DMA kernel."""
    from cddmsl_amd import hip
    N, H, W, Cin, Cout, K, p = case
    dtype, tol = torch.bfloat16, 2e-2
    x = _rand((N, Cin, H, W), 31).to(dtype).float().requires_grad_(False)
    dy = _rand((N, Cout, H, W), 32).to(dtype).float()
    scale = torch.rand(Cout, generator=torch.Generator().manual_seed(33)) + 0.5
    w = torch.zeros(Cout, Cin, K, K, requires_grad=True)
    (F.conv2d(x, w, padding=p) * scale.view(1, -1, 1, 1)).backward(dy)
    dev = "cuda"
    xg, dyg = _nhwc(x).to(dev, dtype), _nhwc(dy).to(dev, dtype)
    out = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("CDDMSL_GEMM256", mode)
        dw = hip.conv_wgrad(xg, dyg, (Cout, K, K, Cin), scale.to(dev), stride=1, pad=p)
        torch.cuda.synchronize()
        out[mode] = dw.cpu().permute(0, 3, 1, 2)
    ref = w.grad
    for mode in ("0", "2"):
        err = (out[mode] - ref).abs().max() / ref.abs().max()
        assert err < tol, (mode, float(err))
    # same products, fp32 accumulation in a different split order
    assert (out["0"] - out["2"]).abs().max() <= 1e-4 * ref.abs().max()


@pytest.mark.parametrize("g256", ["0", "2"])
@pytest.mark.parametrize("case", [(8, 28, 28, 256, 256, 3, 1), (4, 40, 37, 128, 512, 1, 0), (9, 14, 14, 512, 512, 3, 1)])
def test_conv_wgrad_split_reduction_through_workspace(case, g256, monkeypatch):
    """Split reductions of both bf16 weight-gradient kernels through the registered workspace (partial tiles + k_wgrad_reduce, the
    default) against f32 atomics (CDDMSL_WGRAD_WS=0) and ATen fp32, accumulating INTO a non-zero dW as the training step does;
    the workspace form is deterministic: two runs are bit-equal."""
    from cddmsl_amd import hip
    N, H, W, Cin, Cout, K, p = case
    dtype = torch.bfloat16
    x = _rand((N, Cin, H, W), 41).to(dtype).float()
    dy = _rand((N, Cout, H, W), 42).to(dtype).float()
    scale = torch.rand(Cout, generator=torch.Generator().manual_seed(43)) + 0.5
    w = torch.zeros(Cout, Cin, K, K, requires_grad=True)
    (F.conv2d(x, w, padding=p) * scale.view(1, -1, 1, 1)).backward(dy)
    base = _rand((Cout, K, K, Cin), 44)
    dev = "cuda"
    xg, dyg = _nhwc(x).to(dev, dtype), _nhwc(dy).to(dev, dtype)
    monkeypatch.setenv("CDDMSL_GEMM256", g256)
    out = {}
    for mode in ("0", "1", "1b"):
        monkeypatch.setenv("CDDMSL_WGRAD_WS", mode[0])
        dw = base.to(dev).clone()
        hip.conv_wgrad(xg, dyg, (Cout, K, K, Cin), scale.to(dev), stride=1, pad=p, out=dw)
        torch.cuda.synchronize()
        out[mode] = dw.cpu()
    assert hip._L().cddmsl_last_kernel() == (6 if g256 == "2" and Cout % 256 == 0 and (K * K * Cin) % 256 == 0 else 5)
    ref = base + w.grad.permute(0, 2, 3, 1)
    for mode in ("0", "1"):
        err = (out[mode] - ref).abs().max() / ref.abs().max()
        assert err < 2e-2, (mode, float(err))
    assert (out["0"] - out["1"]).abs().max() <= 1e-4 * ref.abs().max()
    assert torch.equal(out["1"], out["1b"])


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("case", [(2, 37, 53, 3, 32, 2), (1, 40, 61, 32, 32, 1), (3, 21, 30, 32, 64, 1), (1, 5, 7, 3, 64, 1)])
def test_conv3x3_small_channel_streaming_kernel(case, dtype, tol):
    """The register-weight streaming kernel of the CLIP stem (Cin 3 -> padded pixel, or 32; Cout 32 / 64; stride 1 / 2;
    ragged sizes) vs ATen fp32, FrozenBN + ReLU epilogue."""
    from cddmsl_amd import hip
    N, H, W, Cin, Cout, s = case
    x = _rand((N, Cin, H, W), 41).to(dtype).float()
    w = (_rand((Cout, Cin, 3, 3), 42) * (Cin * 9) ** -0.5).to(dtype).float()
    scale = torch.rand(Cout, generator=torch.Generator().manual_seed(43)) + 0.5
    bias = _rand((Cout,), 44, 0.1)
    ref = F.relu(F.conv2d(x, w, stride=s, padding=1) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1))
    cp = Cin if Cin >= 8 else (8 if dtype == torch.bfloat16 else 4)        # the 3-channel input is padded to a 16-byte pixel
    xg = torch.zeros(N, H, W, cp)
    xg[..., :Cin] = x.permute(0, 2, 3, 1)
    wg = torch.zeros(Cout, 3, 3, cp)
    wg[..., :Cin] = w.permute(0, 2, 3, 1)
    if dtype == torch.float32 and Cin == 32:
        pytest.skip("f32 with 32 channels is 8 chunks per pixel: the tile kernel's fast path, not this kernel")
    wf, _ = hip.weight_prep(wg.cuda(), None, dtype, True, False)
    y = hip.conv_fwd(xg.cuda().to(dtype), wf, scale.cuda(), bias.cuda(), relu=True, stride=s, pad=1)
    assert hip._L().cddmsl_last_kernel() == 8
    err = (y.float().cpu().permute(0, 3, 1, 2) - ref).abs().max() / ref.abs().max()
    assert err < tol, float(err)


@pytest.mark.parametrize("case", [(2, 37, 53, 256, 64), (1, 50, 83, 64, 64), (3, 5, 7, 256, 64)])
def test_conv1x1_to_64_channels_on_the_streaming_kernel(case, monkeypatch):
    """res2's 1x1 layers with 64 output channels (64 -> 64, 256 -> 64; clip_backbone.py:57-70) run on the streaming kernel's one-tap
    instantiation (a 128-column GEMM tile would be half empty): vs ATen fp32 with the FrozenBN + ReLU epilogue, ragged tile counts, and
    bit-equal to the GEMM kernel it replaces (same products, same k order inside an MFMA step; CDDMSL_SMALL_1X1=0 selects it)."""
    from cddmsl_amd import hip
    N, H, W, Cin, Cout = case
    x = _rand((N, H, W, Cin), 51).bfloat16()
    w = (_rand((Cout, 1, 1, Cin), 52) * Cin ** -0.5)
    scale = torch.rand(Cout, generator=torch.Generator().manual_seed(53)) + 0.5
    bias = _rand((Cout,), 54, 0.1)
    wf, _ = hip.weight_prep(w.cuda(), None, torch.bfloat16, True, False)
    y = hip.conv_fwd(x.cuda(), wf, scale.cuda(), bias.cuda(), relu=True)
    assert hip._L().cddmsl_last_kernel() == 8
    ref = F.relu(x.float().view(-1, Cin) @ wf.float().cpu().view(Cout, Cin).t() * scale + bias).view(N, H, W, Cout)
    err = (y.float().cpu() - ref).abs().max() / ref.abs().max()
    assert err < 6e-3, float(err)
    monkeypatch.setenv("CDDMSL_SMALL_1X1", "0")
    y0 = hip.conv_fwd(x.cuda(), wf, scale.cuda(), bias.cuda(), relu=True)
    assert hip._L().cddmsl_last_kernel() != 8
    assert float((y.float() - y0.float()).abs().max()) <= 2e-2 * float(ref.abs().max())

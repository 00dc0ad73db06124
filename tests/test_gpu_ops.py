"""GPU parity (through the C-ABI) of the non-GEMM hot-path kernels against the CPU oracle."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
T = torch.tensor


def _rand(shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


# ------------------------------------------------------------------ RoIAlign
@pytest.mark.parametrize("aligned", [False, True])
def test_roi_align_kat(aligned):
    """tests/layers/test_roi_align.py:14-47 table, through the HIP kernel (C padded 1 -> 4 channels)."""
    from cddmsl_amd import hip
    k = json.load(open(os.path.join(G, "kat.json")))["roi_align_5x5"]
    x = torch.arange(25, dtype=torch.float32).reshape(1, 5, 5, 1).repeat(1, 1, 1, 4).cuda()
    rois = T([[0.0] + [float(v) for v in k["box"]]]).cuda()
    out = hip.roi_align_forward(x, rois, 4, 4, 1.0, 0, aligned)
    exp = T(k["aligned_true" if aligned else "aligned_false"])
    assert torch.allclose(out[0, :, :, 0].cpu(), exp)


def test_roi_align_empty():
    from cddmsl_amd import hip
    x = torch.rand(1, 5, 5, 4).cuda()
    out = hip.roi_align_forward(x, T([[0.0, 3.0, 3.0, 3.0, 3.0]]).cuda(), 7, 7, 1.0, 0, True)
    assert out.shape == (1, 7, 7, 4) and (out == 0).all()
    assert hip.roi_align_forward(x, torch.zeros(0, 5).cuda(), 7, 7, 1.0, 0, True).shape == (0, 7, 7, 4)
    dx = hip.roi_align_backward(torch.ones(1, 7, 7, 4).cuda(), T([[0.0, 3.0, 3.0, 3.0, 3.0]]).cuda(),
                                T([0, 1], dtype=torch.int32).cuda(), (1, 5, 5, 4), 1.0, 0, True)
    assert (dx == 0).all()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1.5e-2)])
def test_roi_align_vs_oracle(dtype, tol):
    from cddmsl_amd import hip
    from oracle import ops as oo
    N, C, H, W, K = 3, 64, 13, 21, 40
    g = torch.Generator().manual_seed(7)
    x = _rand((N, C, H, W), 1).to(dtype).float()
    b = torch.sort(torch.randint(0, N, (K,), generator=g)).values
    x0 = torch.rand(K, generator=g) * W * 16 * 0.9 - 10
    y0 = torch.rand(K, generator=g) * H * 16 * 0.9 - 10
    rois = torch.stack([b.float(), x0, y0, x0 + torch.rand(K, generator=g) * W * 12, y0 + torch.rand(K, generator=g) * H * 12], 1)
    rois[5, 3:] = rois[5, 1:3]  # empty box
    xr = x.clone().requires_grad_(True)
    ref = oo.roi_align(xr, rois, 14, 1 / 16, 0, True)
    dy = _rand(tuple(ref.shape), 3).to(dtype).float()
    ref.backward(dy)
    dbg = torch.zeros(K, 2, dtype=torch.int32).cuda()
    out = hip.roi_align_forward(_nhwc(x).cuda().to(dtype), rois.cuda(), 14, 14, 1 / 16, 0, True, dbg)
    assert (out.float().cpu().permute(0, 3, 1, 2) - ref.detach()).abs().max() < tol * max(1.0, float(ref.abs().max()))
    # bit-exact integer sample-grid assignment (ceil(roi/14) per axis)
    rh = (rois[:, 4] * (1 / 16) - 0.5) - (rois[:, 2] * (1 / 16) - 0.5)
    rw = (rois[:, 3] * (1 / 16) - 0.5) - (rois[:, 1] * (1 / 16) - 0.5)
    assert torch.equal(dbg.cpu()[:, 0], torch.ceil(rh / 14).int()) and torch.equal(dbg.cpu()[:, 1], torch.ceil(rw / 14).int())
    start = torch.searchsorted(b, torch.arange(N + 1)).int().cuda()
    dx = hip.roi_align_backward(_nhwc(dy).cuda().to(dtype), rois.cuda(), start, (N, H, W, C), 1 / 16, 0, True)
    assert (_nchw(dx.float().cpu()) - xr.grad).abs().max() < tol * max(1.0, float(xr.grad.abs().max()))


# ------------------------------------------------------------------ preprocess / pooling
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 1e-2)])
def test_preprocess(dtype, tol):
    from cddmsl_amd import hip
    from oracle import model as om
    cfg = om.Cfg()
    g = torch.Generator().manual_seed(3)
    imgs = [torch.randint(0, 256, (3, 37, 53), generator=g, dtype=torch.uint8), torch.randint(0, 256, (3, 41, 49), generator=g, dtype=torch.uint8)]
    ref, sizes = om.preprocess_image(cfg, [{"image": i} for i in imgs])
    out = hip.preprocess([i.cuda() for i in imgs], 41, 53, cfg.pixel_mean, cfg.pixel_std, dtype)
    assert (out[..., :3].float().cpu().permute(0, 3, 1, 2) - ref).abs().max() < tol * 3
    assert (out[..., 3:] == 0).all()
    ref2 = om.preprocess_image_train(cfg, [{"image": i} for i in imgs], "image")
    # torchvision Resize is un-vendored: bicubic definition = ATen upsample_bicubic2d(align_corners=False) ("parity unpinned")
    imgs2 = [torch.randint(0, 256, (3, 260, 300), generator=g, dtype=torch.uint8), torch.randint(0, 256, (3, 250, 330), generator=g, dtype=torch.uint8)]
    ref2 = om.preprocess_image_train(cfg, [{"image": i} for i in imgs2], "image")
    out2 = hip.preprocess224([i.cuda() for i in imgs2], 260, 330, cfg.pixel_mean, cfg.pixel_std, dtype)
    assert out2.shape[:3] == (2, 224, 224)
    assert (out2[..., :3].float().cpu().permute(0, 3, 1, 2) - ref2).abs().max() < max(tol * 5, 2e-5)
    # the per-image entry points of the C-ABI (one launch per image) write the same bytes as the batched ones, and a batch larger than
    # one launch's image table (32) is split correctly
    from cddmsl_amd.hip import _L, _f3, DT, ptr, stream_ptr
    cu = [i.cuda() for i in imgs]
    one = torch.empty_like(out)
    m, sd = _f3(cfg.pixel_mean), _f3(cfg.pixel_std)
    for n, im in enumerate(cu):
        assert _L().cddmsl_preprocess(ptr(im), ptr(one), n, im.shape[1], im.shape[2], 41, 53, out.shape[-1], m, sd, 1, DT[dtype], stream_ptr()) == 0
    assert torch.equal(one, out)
    cu2 = [i.cuda() for i in imgs2]
    one2 = torch.empty_like(out2)
    for n, im in enumerate(cu2):
        assert _L().cddmsl_preprocess224(ptr(im), ptr(one2), n, im.shape[1], im.shape[2], 260, 330, 224, int(224 * 330 / 260), 0,
                                         int(round((int(224 * 330 / 260) - 224) / 2.0)), 224, out2.shape[-1], m, sd, DT[dtype], stream_ptr()) == 0
    assert torch.equal(one2, out2)
    many = hip.preprocess(cu * 20, 41, 53, cfg.pixel_mean, cfg.pixel_std, dtype)           # 40 images: two launches
    assert torch.equal(many, out.repeat(20, 1, 1, 1))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 1e-2)])
def test_avgpool(dtype, tol):
    from cddmsl_amd import hip
    x = _rand((2, 16, 9, 11), 5).to(dtype).float()
    y = hip.avgpool2_fwd(_nhwc(x).cuda().to(dtype))
    assert (_nchw(y.float().cpu()) - F.avg_pool2d(x, 2)).abs().max() < tol
    xr = x.clone().requires_grad_(True)
    out = F.relu(xr)
    dy = _rand((2, 16, 4, 5), 6).to(dtype).float()
    add = _rand((2, 16, 9, 11), 7).to(dtype).float()
    F.avg_pool2d(out, 2).backward(dy)
    # dx = (up(dy)/4 + add) * (x > 0)
    dx = hip.avgpool2_bwd(_nhwc(dy).cuda().to(dtype), (2, 9, 11, 16), _nhwc(x).cuda().to(dtype), _nhwc(add).cuda().to(dtype))
    ref = xr.grad + add * (x > 0)
    assert (_nchw(dx.float().cpu()) - ref).abs().max() < tol * 4


# ------------------------------------------------------------------ boxes: bit-exact integer stages
def test_anchors_iou_match_bitexact():
    from cddmsl_amd import hip
    from oracle import ops as oo
    cell = oo.cell_anchors()
    a_ref = oo.grid_anchors(7, 11, 16)
    a = hip.anchors(cell.cuda(), 7, 11, 16.0, 0.0)
    assert torch.equal(a.cpu(), a_ref)
    g = np.load(os.path.join(G, "ref_boxes.npz"))
    gt, pr = T(g["gt"]), T(g["pr"])
    m1, l1 = hip.iou_match(gt.cuda(), pr.cuda(), [0.3, 0.7], [0, -1, 1], True)
    m2, l2 = hip.iou_match(gt.cuda(), pr.cuda(), [0.5], [0, 1], False)
    assert np.array_equal(m1.cpu().numpy(), g["match_rpn"]) and np.array_equal(l1.cpu().numpy(), g["label_rpn"])
    assert np.array_equal(m2.cpu().numpy(), g["match_roi"]) and np.array_equal(l2.cpu().numpy(), g["label_roi"])
    # larger random case vs oracle, incl. empty gt
    gen = torch.Generator().manual_seed(9)
    gt2 = torch.rand(7, 4, generator=gen) * 600
    gt2[:, 2:] = gt2[:, :2] + 30 + torch.rand(7, 2, generator=gen) * 300
    an = oo.grid_anchors(50, 83, 16)
    mo, lo = oo.matcher(oo.pairwise_iou(gt2, an), [0.3, 0.7], [0, -1, 1], True)
    mg, lg = hip.iou_match(gt2.cuda(), an.cuda(), [0.3, 0.7], [0, -1, 1], True)
    assert torch.equal(mg.cpu(), mo) and torch.equal(lg.cpu(), lo)
    me, le = hip.iou_match(torch.zeros(0, 4).cuda(), an[:100].cuda(), [0.3, 0.7], [0, -1, 1], True)
    assert (me == 0).all() and (le == 0).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_roi_align_pooled_byproduct_is_avgpool_of_the_result(dtype):
    """roi_align_forward(with_pooled=True): the crops are the same as without it, and the second output is bit-identical to
    avgpool2_fwd of the crops (it is formed from the rounded outputs in the pooling kernel's order)."""
    from cddmsl_amd import hip
    g = torch.Generator().manual_seed(17)
    N, H, W, C, K = 2, 25, 31, 64, 37
    x = torch.randn(N, H, W, C, generator=g).to(dtype).cuda()
    b = torch.rand(K, 4, generator=g)
    x0, y0 = b[:, 0] * 300, b[:, 1] * 250
    rois = torch.stack([torch.randint(0, N, (K,), generator=g).float(), x0, y0, x0 + 8 + b[:, 2] * 200, y0 + 8 + b[:, 3] * 150], dim=1)
    rois = rois[rois[:, 0].argsort(stable=True)].contiguous().cuda()
    for sr in (0, 2):
        y = hip.roi_align_forward(x, rois, 14, 14, 1.0 / 16, sr, True)
        y2, yp = hip.roi_align_forward(x, rois, 14, 14, 1.0 / 16, sr, True, with_pooled=True)
        if dtype == torch.float32:
            assert torch.equal(y, y2)
        else:    # crops alone come from the row-sliding bf16 kernel (same sums, another order): one rounding apart at most
            assert float((y.float() - y2.float()).abs().max()) <= 2.0 ** -7 * float(y2.float().abs().max())
        assert torch.equal(yp, hip.avgpool2_fwd(y2))


def test_frozen_mlp_node_matches_torch():
    """layers.frozen_mlp (the mapper's MlpTransformer as one node: f32 residual in fc2's epilogue, ReLU backward in fc2's
    input-gradient epilogue) vs fp32 torch on the same bf16-rounded operands: output, input gradient, residual gradient."""
    from cddmsl_amd import layers
    g = torch.Generator().manual_seed(23)
    M, D, Hd = 700, 768, 1536
    w1 = torch.nn.Parameter((torch.randn(Hd, D, generator=g) * D ** -0.5).cuda())
    w2 = torch.nn.Parameter((torch.randn(D, Hd, generator=g) * Hd ** -0.5).cuda())
    b1, b2 = (torch.randn(Hd, generator=g) * 0.1).cuda(), (torch.randn(D, generator=g) * 0.1).cuda()
    x = torch.randn(M, D, generator=g).cuda().bfloat16().requires_grad_(True)
    res = torch.randn(M, D, generator=g).cuda().requires_grad_(True)
    y = layers.frozen_mlp(x, layers.PreparedWeight(w1, None, frozen=True), b1, layers.PreparedWeight(w2, None, frozen=True), b2, res)
    gy = torch.randn(M, D, generator=g).cuda()
    y.backward(gy)
    xr = x.detach().float().requires_grad_(True)
    h = torch.relu(xr @ w1.detach().bfloat16().float().t() + b1).bfloat16().float()
    yr = h @ w2.detach().bfloat16().float().t() + b2 + res.detach()
    yr.backward(gy.bfloat16().float())
    assert y.dtype == torch.float32 and (y - yr).abs().max() < 2e-2 * yr.abs().max()
    assert torch.equal(res.grad, gy)
    assert (x.grad.float() - xr.grad).abs().max() < 3e-2 * xr.grad.abs().max()


def test_sort_decode_nms_bitexact():
    from cddmsl_amd import hip
    from oracle import ops as oo
    from oracle import model as om
    cfg = om.Cfg()
    N, Hf, Wf, A = 2, 12, 17, 15
    total = Hf * Wf * A
    logits = _rand((N, total), 1)
    logits[0, 5] = logits[0, 900]  # tie -> lower index first
    deltas = _rand((N, total, 4), 2, 0.4)
    deltas[1, 3, 2] = 50.0  # scale clamp
    sizes = [(180, 260), (192, 272)]
    anchors = oo.grid_anchors(Hf, Wf, 16)
    topk = 1000
    prop = oo.apply_deltas(deltas.reshape(-1, 4), anchors.unsqueeze(0).expand(N, -1, -1).reshape(-1, 4), cfg.rpn_bbox_weights).view(N, -1, 4)
    cfg2 = om.Cfg(rpn_pre_nms_topk=topk, rpn_post_nms_topk=300)
    rec = {}
    ref = om.find_top_rpn_proposals(cfg2, prop, logits, sizes, True, rec)

    keys, order = hip.sort_desc(logits.cuda())
    srt = torch.sort(logits, descending=True, dim=1, stable=True)
    assert torch.equal(order.cpu().long(), srt.indices) and torch.equal(keys.cpu(), srt.values)
    img_hw = T(sizes, dtype=torch.int32).cuda()
    boxes, valid = hip.rpn_decode(order, deltas.cuda(), oo.cell_anchors().cuda(), img_hw, Hf, Wf, topk, 16.0, 0.0,
                                  cfg.rpn_bbox_weights, oo.SCALE_CLAMP, 0.0)
    # decode is fp32 with expf: compare to the oracle's boxes within 1e-4 px, then NMS on the ORACLE's boxes bit-exactly
    for n in range(N):
        ob = oo.clip_boxes(prop[n][srt.indices[n, :topk]], sizes[n])
        assert (boxes[n].cpu() - ob).abs().max() < 2e-3
        assert torch.equal(valid[n].cpu() == 1, oo.nonempty(ob))
    oboxes = torch.stack([oo.clip_boxes(prop[n][srt.indices[n, :topk]], sizes[n]) for n in range(N)]).cuda()
    ovalid = torch.stack([oo.nonempty(oo.clip_boxes(prop[n][srt.indices[n, :topk]], sizes[n])) for n in range(N)]).to(torch.uint8).cuda()
    keep, nkeep = hip.nms(oboxes, ovalid, 0.7, 300)
    for n in range(N):
        # oracle keep indexes the nonempty-filtered list; map positions in the sorted list through the filter
        pos = torch.nonzero(ovalid[n].cpu() == 1, as_tuple=True)[0]
        assert int(nkeep[n]) == len(rec["nms_keep"][n])
        assert torch.equal(keep[n, : int(nkeep[n])].cpu().long(), pos[rec["nms_keep"][n]])


# ------------------------------------------------------------------ losses
def test_cosine_logits_and_contrastive():
    from cddmsl_amd import hip, synthetic
    from oracle import model as om
    cfg = om.Cfg()
    sd = synthetic.make_state_dict(0)
    x = _rand((37, 1024), 1).requires_grad_(True)
    scores, _ = om.box_predictor(sd, cfg, x)
    ds = _rand(tuple(scores.shape), 2)
    scores.backward(ds)
    # bbox_pred path excluded: zero its contribution
    x2 = x.detach().clone().requires_grad_(True)
    nx = F.normalize(x2, dim=1)
    s2 = torch.cat((nx @ F.normalize(sd["roi_heads.box_predictor.cls_score.weight"], dim=1).t(), nx @ torch.zeros(1024, 1)), 1) / cfg.cls_temp
    s2.backward(ds)
    wn = F.normalize(sd["roi_heads.box_predictor.cls_score.weight"], dim=1).cuda()
    sg, inv = hip.cosine_logits_fwd(x.detach().cuda(), wn, cfg.cls_temp)
    assert (sg.cpu() - s2.detach()).abs().max() < 1e-4 * float(s2.abs().max())
    assert (sg[:, -1] == 0).all()
    dx = hip.cosine_logits_bwd(ds.cuda(), x.detach().cuda(), wn, inv, cfg.cls_temp)
    assert (dx.cpu() - x2.grad).abs().max() < 1e-4 * float(x2.grad.abs().max())

    a = _rand((48, 256), 3).requires_grad_(True)
    b = _rand((48, 256), 4).requires_grad_(True)
    loss = om.symmetric_ce(a, b)
    loss.backward()
    an, ia = hip.l2norm_fwd(a.detach().cuda(), 0.0)
    bn, ib = hip.l2norm_fwd(b.detach().cuda(), 0.0)
    S = hip.linear_fwd(an, bn)
    lg, rl, cl = hip.contrastive_fwd(S)
    assert abs(float(lg) - float(loss)) < 1e-5 * abs(float(loss))
    dS = hip.contrastive_bwd(S, rl, cl, torch.ones(1).cuda())
    dan = hip.linear_fwd(dS, bn.t().contiguous())
    dbn = hip.linear_fwd(dS.t().contiguous(), an.t().contiguous())
    da = hip.l2norm_bwd(dan, an, ia)
    db = hip.l2norm_bwd(dbn, bn, ib)
    assert (da.cpu() - a.grad).abs().max() < 1e-4 * float(a.grad.abs().max())
    assert (db.cpu() - b.grad).abs().max() < 1e-4 * float(b.grad.abs().max())


def test_sgd_clip_step():
    from cddmsl_amd import hip
    from oracle import model as om
    cfg = om.Cfg()
    shapes = [(64, 3, 3, 32), (1000,), (7, 5), (2048, 1, 1, 512)]
    ps = [_rand(s, i) for i, s in enumerate(shapes)]
    gs = [_rand(s, 10 + i, 3.0 if i % 2 else 0.01) for i, s in enumerate(shapes)]
    sd = {str(i): p.clone() for i, p in enumerate(ps)}
    mom = {}
    P = [p.clone().cuda() for p in ps]
    M = [torch.zeros_like(p) for p in P]
    ws = torch.zeros(len(P)).cuda()
    for it in (150, 151):
        lr = om.sgd_step(sd, {str(i): g for i, g in enumerate(gs)}, mom, cfg, it)
        hip.sgd_clip_step(P, [g.cuda() for g in gs], M, ws, lr, cfg.momentum, cfg.weight_decay, cfg.clip_value, it == 150)
    for i in range(len(ps)):
        assert (P[i].cpu() - sd[str(i)]).abs().max() < 1e-6
        assert (M[i].cpu() - mom[str(i)]).abs().max() < 1e-5


def test_layernorm_and_focal_ce():
    from cddmsl_amd import hip, layers, synthetic
    from oracle import model as om
    x = _rand((37, 768), 1).requires_grad_(True)
    g, b = 1.0 + 0.1 * _rand((768,), 2), 0.1 * _rand((768,), 3)
    ref = F.layer_norm(x, (768,), g, b)
    w = _rand((37, 768), 4)
    (ref * w).sum().backward()
    xg = x.detach().cuda().requires_grad_(True)
    y = layers.layer_norm(xg, g.cuda(), b.cuda(), torch.float32)
    assert (y.cpu() - ref.detach()).abs().max() < 1e-5
    (y * w.cuda()).sum().backward()
    assert (xg.grad.cpu() - x.grad).abs().max() < 1e-5 * max(1.0, float(x.grad.abs().max()))
    yb = layers.layer_norm(x.detach().cuda(), g.cuda(), b.cuda(), torch.bfloat16)
    assert yb.dtype == torch.bfloat16 and (yb.float().cpu() - ref.detach()).abs().max() < 3e-2

    cfg = om.Cfg()
    s = (_rand((61, 21), 5) * 30).requires_grad_(True)
    t = torch.randint(0, 21, (61,), generator=torch.Generator().manual_seed(6))
    t[:7] = 20
    ref_l = om.focal_loss(cfg, s, t)
    ref_l.backward()
    sg = s.detach().cuda().requires_grad_(True)
    l = layers.focal_cross_entropy(sg, t.cuda(), cfg.focal_gamma, cfg.num_classes, cfg.bg_cls_loss_weight)
    l.backward()
    assert abs(float(l) - float(ref_l)) < 1e-5 * abs(float(ref_l))
    assert (sg.grad.cpu() - s.grad).abs().max() < 1e-5 * float(s.grad.abs().max()) + 1e-9


@pytest.mark.parametrize("n,t", [(3, 80), (2, 96), (5, 37)])
def test_small_attention_fwd_bwd(n, t):
    """Fused mapper attention (csrc/attn_small.hip, bf16 MFMA + fp32 softmax) vs fp32 torch on the same bf16 inputs
    (clipcap.py:59-83: softmax(q k^T * scale) v per head, no mask)."""
    from cddmsl_amd import layers
    H, dh = 8, 96
    d = H * dh
    g = torch.Generator().manual_seed(7)
    q = (torch.randn(n * t, d, generator=g) * 0.5).bfloat16()
    kv = (torch.randn(n * t, 2 * d, generator=g) * 0.5).bfloat16()
    do = torch.randn(n * t, d, generator=g).bfloat16()
    scale = dh ** -0.5
    qr, kvr = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    qq = qr.view(n, t, H, dh).permute(0, 2, 1, 3)
    kk = kvr[:, :d].reshape(n, t, H, dh).permute(0, 2, 1, 3)
    vv = kvr[:, d:].reshape(n, t, H, dh).permute(0, 2, 1, 3)
    ref = (torch.softmax(qq @ kk.transpose(-1, -2) * scale, dim=-1) @ vv).permute(0, 2, 1, 3).reshape(n * t, d)
    ref.backward(do.float())
    qg, kvg = q.cuda().requires_grad_(True), kv.cuda().requires_grad_(True)
    o = layers.small_attention(qg, kvg, t, H, scale)
    o.backward(do.cuda())
    for got, want, name in ((o, ref, "o"), (qg.grad, qr.grad, "dq"), (kvg.grad, kvr.grad, "dkv")):
        err = (got.float().cpu() - want.detach()).abs().max() / want.detach().abs().max()
        assert err < 2e-2, (name, float(err))
    # the same kernels on ONE fused q|k|v tensor (row stride 3d): bit-identical output and gradients
    qkv = torch.cat([q, kv], dim=1).cuda().requires_grad_(True)
    o2 = layers.small_attention_qkv(qkv, t, H, scale)
    o2.backward(do.cuda())
    assert torch.equal(o2, o)
    assert torch.equal(qkv.grad[:, :d], qg.grad) and torch.equal(qkv.grad[:, d:], kvg.grad)


def test_fast_rcnn_inference_single_image_matches_reference():
    """Detection post-processing on the HIP sort / NMS kernels vs the reference's own fast_rcnn_inference_single_image and
    detector_postprocess (tests/golden/ref_inference.npz): same kept (proposal, class) pairs in the same order."""
    import numpy as np
    from cddmsl_amd.modeling.postprocessing import detector_postprocess
    from cddmsl_amd.modeling.roi_heads import fast_rcnn_inference_single_image
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_inference.npz"))
    inst, kept = fast_rcnn_inference_single_image(torch.from_numpy(fx["boxes"]).cuda(), torch.from_numpy(fx["scores"]).cuda(), (200, 300), 0.05, 0.5, 20)
    assert torch.equal(inst.pred_classes.cpu(), torch.from_numpy(fx["det_classes"]))
    assert torch.equal(kept.cpu(), torch.from_numpy(fx["det_kept"]))
    assert torch.equal(inst.pred_boxes.tensor.cpu(), torch.from_numpy(fx["det_boxes"])) and torch.equal(inst.scores.cpu(), torch.from_numpy(fx["det_scores"]))
    post = detector_postprocess(inst, 333, 480)
    assert post.image_size == (333, 480) and torch.equal(post.pred_classes.cpu(), torch.from_numpy(fx["post_classes"]))
    assert torch.allclose(post.pred_boxes.tensor.cpu(), torch.from_numpy(fx["post_boxes"]), rtol=0, atol=1e-4)


def test_layer_norm_skip_backward_accumulates_in_place():
    """LayerNormSkipFn (pre-norm residual helper): (LN(x), x) forward, dx = d_skip + LN'(d_ln) in one accumulating kernel --
    a two-block residual chain vs torch autograd with F.layer_norm."""
    import torch.nn.functional as F
    from cddmsl_amd import layers
    g = torch.Generator().manual_seed(11)
    R, D = 37, 768
    h0 = torch.randn(R, D, generator=g)
    gam, bet = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) * 0.1
    w1, w2 = torch.randn(D, D, generator=g) * D ** -0.5, torch.randn(D, D, generator=g) * D ** -0.5
    wout = torch.randn(R, D, generator=g)

    def chain(h, ln):
        y, hs = ln(h)
        h = hs + y @ w1.to(h.device)
        y, hs = ln(h)
        return hs + torch.relu(y @ w2.to(h.device))

    hr = h0.clone().requires_grad_(True)
    (chain(hr, lambda h: (F.layer_norm(h, (D,), gam, bet), h)) * wout).sum().backward()
    hg = h0.clone().cuda().requires_grad_(True)
    gc, bc = gam.cuda(), bet.cuda()
    out = chain(hg, lambda h: layers.layer_norm_skip(h, gc, bc, torch.float32))
    (out * wout.cuda()).sum().backward()
    err = (hg.grad.cpu() - hr.grad).abs().max() / hr.grad.abs().max()
    assert err < 1e-4, float(err)


def test_prepared_weights_refresh_in_one_launch_after_update():
    """PreparedWeight: after the optimizer bumps the weight version, the first ``get`` refreshes every registered weight
    (cddmsl_weight_prep_multi) into its persistent buffers -- forward and flipped / BN-scaled dgrad copies follow the masters."""
    from cddmsl_amd import layers
    g = torch.Generator().manual_seed(3)
    ws = [torch.nn.Parameter(torch.randn(16, 8, 3, 3, generator=g).cuda().contiguous(memory_format=torch.channels_last)),
          torch.nn.Parameter(torch.randn(24, 16, generator=g).cuda())]
    scale = (torch.rand(16, generator=g) + 0.5).cuda()
    pws = [layers.PreparedWeight(ws[0], scale), layers.PreparedWeight(ws[1], None)]

    def expect(w, sc):
        o = layers._ohwi(w.detach())
        wd = o.flip(1, 2).permute(3, 1, 2, 0) * (sc if sc is not None else 1.0)
        return o.bfloat16(), wd.contiguous().bfloat16()

    for rnd in range(3):
        outs = [pw.get(torch.bfloat16, True) for pw in pws]
        for (wf, wd), w, sc in zip(outs, ws, (scale, None)):
            ef, ed = expect(w, sc)
            assert torch.equal(wf, ef) and torch.equal(wd, ed), rnd
        ptrs = [(o[0].data_ptr(), o[1].data_ptr()) for o in outs]
        if rnd:
            assert ptrs == last_ptrs                      # persistent buffers, refreshed in place
        last_ptrs = ptrs
        with torch.no_grad():
            for w in ws:
                w.add_(0.25)
        layers.bump_weight_version()


def test_iou_match_batched_equals_per_image():
    """cddmsl_iou_match_batched (all images in one launch pair) vs the per-image entry point: bit-identical matches and
    labels, shared predictions (anchors, low-quality matches on) and concatenated predictions (proposals), incl. an image
    without boxes."""
    from cddmsl_amd import hip
    g = torch.Generator().manual_seed(5)

    def boxes(n):
        c = torch.rand(n, 2, generator=g) * 200
        wh = torch.rand(n, 2, generator=g) * 80 + 4
        return torch.cat([c, c + wh], 1).cuda()

    gts = [boxes(3), boxes(0), boxes(7), boxes(1)]
    anchors = boxes(1500)
    m, l = hip.iou_match_batched(gts, anchors, None, [0.3, 0.7], [0, -1, 1], True)
    for n, gt in enumerate(gts):
        m1, l1 = hip.iou_match(gt.contiguous(), anchors, [0.3, 0.7], [0, -1, 1], True)
        assert torch.equal(m[n], m1) and torch.equal(l[n], l1), n
    counts = [300, 1, 517, 256]
    props = [boxes(c) for c in counts]
    m, l = hip.iou_match_batched(gts, torch.cat(props).contiguous(), counts, [0.5], [0, 1], False)
    off = 0
    for n, (gt, pr) in enumerate(zip(gts, props)):
        m1, l1 = hip.iou_match(gt.contiguous(), pr.contiguous(), [0.5], [0, 1], False)
        assert torch.equal(m[off:off + counts[n]], m1) and torch.equal(l[off:off + counts[n]], l1), n
        off += counts[n]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attn_tokens_bwd_with_positional_gradient(dtype):
    """dx[k][p] = dtok[k][p+1] + dtok[k][0] / P (masked where the pooled map is <= 0) and, in the same pass, the positional
    embedding's gradient sum_k dtok[k][t] accumulated into an f32 buffer (clip_backbone.py:86-88 backward) -- vs torch."""
    from cddmsl_amd import hip
    K, P, TP, C = 300, 49, 56, 64
    g = torch.Generator().manual_seed(3)
    dtok = torch.randn(K, TP, C, generator=g).to(dtype).cuda()
    mask = torch.randn(K, P, C, generator=g).to(dtype).cuda()
    gpos = torch.full((P + 1, C), 0.5, device="cuda")
    dx = hip.attn_tokens_bwd(dtok, P, mask, gpos)
    d = dtok.float()
    Pt = torch.full((1, 1, 1), float(P), device="cuda")        # a tensor divisor: true division (a Python scalar becomes * (1/P))
    want = (d[:, 1:P + 1] + d[:, :1] / Pt) * (mask.float() > 0)
    assert torch.equal(dx, want.to(dtype))
    wpos = 0.5 + d[:, :P + 1].double().sum(0)
    assert float((gpos.double() - wpos).abs().max()) < 1e-3
    # gradient-only and map-only forms
    gpos2 = torch.zeros(P + 1, C, device="cuda")
    assert hip.attn_tokens_bwd(dtok, P, None, gpos2, want_dx=False) is None
    assert float((gpos2.double() - (wpos - 0.5)).abs().max()) < 1e-3
    dx2 = hip.attn_tokens_bwd(dtok, P)
    assert torch.equal(dx2, (d[:, 1:P + 1] + d[:, :1] / Pt).to(dtype))


@pytest.mark.parametrize("D", [200, 512, 768, 1024])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm_kernels_all_widths(D, dtype):
    """Generic kernel (any width) and the register-resident one (whole 256-column segments): forward, backward and the
    accumulating backward vs torch, with the operand dtype of the mapper's bf16 path and of the f32 parity path."""
    from cddmsl_amd import hip
    g = torch.Generator().manual_seed(D)
    R = 131
    x = torch.randn(R, D, generator=g).cuda()
    gam, bet = (torch.rand(D, generator=g) + 0.5).cuda(), (torch.randn(D, generator=g) * 0.1).cuda()
    dy = torch.randn(R, D, generator=g).to(dtype).cuda()
    skip = torch.randn(R, D, generator=g).cuda()
    y, mean, rstd = hip.layernorm_fwd(x, gam, bet, dtype)
    xr = x.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (D,), gam, bet)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert float((y.float() - ref.detach()).abs().max()) < tol
    (ref * dy.float()).sum().backward()
    dx = hip.layernorm_bwd(dy, x, gam, mean, rstd)
    assert float((dx - xr.grad).abs().max()) < 2e-5 * max(1.0, float(xr.grad.abs().max()))
    acc = skip.clone()
    out = hip.layernorm_bwd(dy, x, gam, mean, rstd, accumulate_into=acc)
    assert out.data_ptr() == acc.data_ptr()
    assert float((acc - (skip + xr.grad)).abs().max()) < 2e-5 * max(1.0, float(xr.grad.abs().max()))


@pytest.mark.parametrize("H", [32, 40])          # one wave per region (H <= 32) / one thread per row
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attnpool_softmax_glue(dtype, H):
    """Softmax over the 50 keys of the query-0 attention pool and its backward, in the padded / transposed layouts the
    batched products read (clip_backbone.py:95-105) -- vs torch.softmax and its autograd."""
    from cddmsl_amd import hip
    K, P1, TP = 77, 50, 56
    g = torch.Generator().manual_seed(5)
    S = (torch.randn(K, H, TP, generator=g) * 3).cuda()
    dP = torch.randn(K, H, TP, generator=g).cuda()
    scale = 64 ** -0.5
    p, pT = hip.attnpool_softmax_fwd(S, P1, scale, dtype)
    Sr = S[:, :, :P1].clone().requires_grad_(True)
    ref = torch.softmax(Sr * scale, dim=-1)
    assert float((p - ref.detach()).abs().max()) < 1e-6
    want_pT = torch.zeros(K, TP, H, device="cuda", dtype=dtype)
    want_pT[:, :P1] = p.transpose(1, 2)
    assert torch.equal(pT, want_pT)
    (ref * dP[:, :, :P1]).sum().backward()
    dsT, pds = hip.attnpool_softmax_bwd(p, dP, scale, dtype)
    ds = Sr.grad                                               # d/dS of sum(softmax(S * scale) * dP)
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    assert float((dsT[:, :P1].float() - ds.transpose(1, 2)).abs().max()) < tol * max(1.0, float(ds.abs().max()))
    assert float(dsT[:, P1:].abs().max()) == 0.0
    assert torch.equal(pds[:, :H, :P1], p.to(dtype)) and torch.equal(pds[:, H:, :P1], dsT[:, :P1].transpose(1, 2))
    assert float(pds[:, :, P1:].abs().max()) == 0.0


def _roi_entry_run(dtype, mode, monkeypatch):
    import os
    from cddmsl_amd import synthetic
    from cddmsl_amd.config import get_cfg
    from cddmsl_amd.modeling import build_model
    from cddmsl_amd.structures import Boxes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(root, "configs", "VOC-Experiments", "faster_rcnn_CLIP_R_50_C4.yaml"))
    cfg.merge_from_list(["MODEL.COMPUTE_DTYPE", dtype])
    model = build_model(cfg)
    model.load_state_dict(synthetic.make_state_dict(0), strict=False)
    model.train()
    T = model.compute_dtype
    g = torch.Generator().manual_seed(5)
    N, H, W = 2, 13, 21
    boxes = []
    for n in range(N):
        k = 9 + 4 * n
        x0, y0 = torch.rand(k, generator=g) * W * 16 * 0.8 - 20, torch.rand(k, generator=g) * H * 16 * 0.8 - 20
        b = torch.stack([x0, y0, x0 + 8 + torch.rand(k, generator=g) * 250, y0 + 8 + torch.rand(k, generator=g) * 180], dim=1)
        b[0] = torch.tensor([30.0, 30.0, 30.0, 30.0])            # empty box
        b[1] = torch.tensor([5.0, 5.0, 9.0, 8.0])                # smaller than one bin
        boxes.append(Boxes(b.cuda()))
    monkeypatch.setenv("CDDMSL_ROI_COMMUTE", mode)
    feat = (_rand((N, H, W, 1024), 31).relu()).bfloat16().to(T).cuda().requires_grad_(True)    # (bf16-representable inputs for both dtypes)
    extra = (_rand((3, 14, 14, 1024), 32).relu()).bfloat16().to(T).cuda().requires_grad_(True)
    out = model.roi_heads._pooled_embeddings(feat, boxes, model.backbone.layer4, model.backbone.attnpool, extra)
    assert tuple(out.shape) == (sum(len(b) for b in boxes) + 3, 1024) and out.dtype == torch.float32
    (out * _rand(tuple(out.shape), 33).cuda()).sum().backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().float().cpu().clone() for k, p in model.named_parameters()
             if p.grad is not None and (k.startswith("backbone.layer4.") or k.startswith("backbone.attnpool."))}
    grads["(embeddings)"], grads["(d feature map)"], grads["(d appended maps)"] = out.detach().cpu(), feat.grad.float().cpu(), extra.grad.float().cpu()
    return grads


def test_roi_head_entry_with_conv1_in_front_of_the_pooling(monkeypatch):
    """RoIAlign -> layer4 -> attention pool with layer4.0's conv1 evaluated on the feature map BEFORE the pooling and the
    downsample path's pooled crops written directly (layers.RoIStageFn, no [K,14,14,1024] crop tensor) against the literal order
    of clip_roi_heads.py:113-115,160-165 (pooler, then layer4 on the crops): embeddings, the gradients wrt the feature map and
    the appended maps, and every layer4 / attention-pool weight gradient.  Boxes include out-of-image, tiny and empty ones.
    Exact f32: the two orders agree to 2e-3 of each tensor's max (a handful of ReLU decisions at ~0 flip).  bf16: both orders are
    compared with the f32 result -- the commuted one must be as close as the literal one (it rounds the conv1 map instead of the
    crops, and the gathered gradient instead of the crop gradient)."""
    rel = lambda u, v: float((u - v).abs().max() / max(float(v.abs().max()), 1e-6))
    lit = _roi_entry_run("f32", "0", monkeypatch)
    com = _roi_entry_run("f32", "1", monkeypatch)
    assert set(lit) == set(com) and "backbone.layer4.0.conv1.weight" in lit and "backbone.layer4.0.downsample.0.weight" in lit
    for k in lit:
        if float(lit[k].abs().max()) >= 1e-7:
            assert rel(com[k], lit[k]) < 2e-3, (k, rel(com[k], lit[k]))
    lit16 = _roi_entry_run("bf16", "0", monkeypatch)
    com16 = _roi_entry_run("bf16", "1", monkeypatch)
    worst = (0.0, 0.0, None)
    for k in lit:
        if float(lit[k].abs().max()) < 1e-7:
            continue
        e_lit, e_com = rel(lit16[k], lit[k]), rel(com16[k], lit[k])
        if e_com > worst[1]:
            worst = (e_lit, e_com, k)
        assert e_com < 1.5 * e_lit + 1e-2, (k, e_lit, e_com)      # (as close to f32 as the literal order, tensor by tensor)
    print("bf16 error vs exact f32 (literal order, commuted order, tensor), worst commuted:", worst)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1.5e-2)])
def test_roi_align_affine_and_pooled_entry_points(dtype, tol):
    """cddmsl_roi_align_forward_affine / _backward_pooled against compositions of the plain entry points: relu(s*roi_align(x)+b),
    the pooled-only output = avgpool2 of the crops, and the pooled backward = roi_align_backward(avgpool2_bwd(dy))."""
    from cddmsl_amd import hip
    N, C, H, W, K = 2, 64, 11, 17, 23
    g = torch.Generator().manual_seed(9)
    x = _rand((N, H, W, C), 41).to(dtype).cuda()
    bi = torch.sort(torch.randint(0, N, (K,), generator=g)).values.float()
    x0, y0 = torch.rand(K, generator=g) * W * 16 * 0.9 - 10, torch.rand(K, generator=g) * H * 16 * 0.9 - 10
    rois = torch.stack([bi, x0, y0, x0 + 4 + torch.rand(K, generator=g) * 200, y0 + 4 + torch.rand(K, generator=g) * 150], dim=1).cuda()
    start = torch.tensor([0, int((bi == 0).sum()), K], dtype=torch.int32).cuda()
    sc, bs = (torch.rand(C, generator=g) + 0.5).cuda(), (_rand((C,), 42, 0.3)).cuda()
    plain, pooled = hip.roi_align_forward(x, rois, 14, 14, 1 / 16, 0, True, with_pooled=True)
    y = hip.roi_align_forward_affine(x, rois, 14, 14, 1 / 16, 0, True, sc, bs, relu=True)
    # the affine is applied to the unrounded pooled value; the reference composition rounds the crop first
    ref = torch.relu(plain.float() * sc + bs)
    assert float((y.float() - ref).abs().max()) <= tol * float(ref.abs().max())
    yp = hip.roi_align_forward_affine(x, rois, 14, 14, 1 / 16, 0, True, pooled_only=True)
    if dtype == torch.float32:
        assert torch.equal(yp, pooled)
    else:        # the bf16 pooled-only output averages the four UNROUNDED bins (k_roi_align_fwd_rows); `pooled` the four rounded ones
        assert float((yp.float() - pooled.float()).abs().max()) <= 2.0 ** -7 * float(pooled.float().abs().max())
    dy = _rand((K, 7, 7, C), 43).to(dtype).cuda()
    want = hip.roi_align_backward(hip.avgpool2_bwd(dy, (K, 14, 14, C)), rois, start, (N, H, W, C), 1 / 16, 0, True)
    got = hip.roi_align_backward(dy, rois, start, (N, H, W, C), 1 / 16, 0, True, pooled=True)
    assert float((got.float() - want.float()).abs().max()) <= tol * float(want.float().abs().max())


def test_nms_with_the_torchvision_signature():
    """cddmsl_nms_anyorder: torchvision.ops.nms(boxes, scores, thr) as layers/nms.py:30,35 calls it -- candidates in ANY order,
    kept indices in descending-score order -- against the oracle's NMS (C restatement of the published algorithm): identical
    keep lists incl. duplicate boxes and tied scores (stable: the earlier index wins), and the empty case."""
    from cddmsl_amd import hip
    from oracle import ops as oo
    g = torch.Generator().manual_seed(17)
    for K, thr in ((1, 0.5), (300, 0.7), (5000, 0.5)):
        xy = torch.rand(K, 2, generator=g) * 400
        boxes = torch.cat([xy, xy + 10 + torch.rand(K, 2, generator=g) * 120], dim=1)
        scores = torch.rand(K, generator=g)
        if K > 10:
            boxes[7] = boxes[3]                       # duplicates
            scores[11] = scores[5]                    # a tie
        keep, nkeep = hip.nms_anyorder(boxes.cuda(), scores.cuda(), thr)
        n = int(nkeep)
        want = oo.nms(boxes, scores, thr)
        assert n == len(want) and torch.equal(keep[:n].cpu(), want) and bool((keep[n:] == -1).all())
    keep, nkeep = hip.nms_anyorder(torch.zeros(0, 4).cuda(), torch.zeros(0).cuda(), 0.5)
    assert int(nkeep) == 0 and keep.numel() == 0


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1.5e-2)])
def test_roi_align_with_the_torchvision_signature(dtype, tol):
    """cddmsl_roi_align_nchw_anyorder / _backward_nchw_anyorder: ``torchvision.ops.roi_align(input[N,C,H,W], rois[K,5], ...)`` as
    layers/roi_align.py:58-65 calls it -- NCHW, RoIs in ANY order (here shuffled across images, one naming an image outside the
    batch, one empty box), a channel count that is not a whole 16-byte chunk -- against the oracle, forward and backward; and the
    reference's own 5x5 known-answer table (tests/layers/test_roi_align.py:14-47) with its single channel."""
    from cddmsl_amd import hip
    from oracle import ops as oo
    k = json.load(open(os.path.join(G, "kat.json")))["roi_align_5x5"]
    x5 = torch.arange(25, dtype=torch.float32).reshape(1, 1, 5, 5).cuda().to(dtype)
    r5 = T([[0.0] + [float(v) for v in k["box"]]]).cuda()
    for aligned in (False, True):
        out = hip.roi_align_nchw(x5, r5, (4, 4), 1.0, 0, aligned)
        assert out.shape == (1, 1, 4, 4)
        assert torch.allclose(out[0, 0].float().cpu(), T(k["aligned_true" if aligned else "aligned_false"]), atol=0.1 if dtype == torch.bfloat16 else 1e-5)
    N, C, H, W, K = 3, 37, 13, 21, 41
    g = torch.Generator().manual_seed(11)
    x = _rand((N, C, H, W), 1).to(dtype).float()
    b = torch.randint(0, N, (K,), generator=g)                                   # NOT grouped by image
    x0 = torch.rand(K, generator=g) * W * 16 * 0.9 - 10
    y0 = torch.rand(K, generator=g) * H * 16 * 0.9 - 10
    rois = torch.stack([b.float(), x0, y0, x0 + torch.rand(K, generator=g) * W * 12, y0 + torch.rand(K, generator=g) * H * 12], 1)
    rois[5, 3:] = rois[5, 1:3]                                                   # empty box
    assert not torch.equal(torch.sort(b).values, b)
    xr = x.clone().requires_grad_(True)
    ref = oo.roi_align(xr, rois, 14, 1 / 16, 0, True)
    dy = _rand(tuple(ref.shape), 3).to(dtype).float()
    ref.backward(dy)
    out = hip.roi_align_nchw(x.cuda().to(dtype), rois.cuda(), 14, 1 / 16, 0, True)
    assert out.shape == (K, C, 14, 14)
    assert (out.float().cpu() - ref.detach()).abs().max() < tol * max(1.0, float(ref.abs().max()))
    dx = hip.roi_align_backward_nchw(dy.cuda().to(dtype), rois.cuda(), (N, C, H, W), 1 / 16, 0, True)
    assert dx.shape == (N, C, H, W)
    assert (dx.float().cpu() - xr.grad).abs().max() < tol * max(1.0, float(xr.grad.abs().max()))
    # equal, element for element, to the channels-last entry points on the image-sorted RoIs
    order = torch.sort(b, stable=True).indices
    srt = rois[order].contiguous()
    y_cl = hip.roi_align_forward(_nhwc(torch.nn.functional.pad(x, (0, 0, 0, 0, 0, 3))).cuda().to(dtype), srt.cuda(), 14, 14, 1 / 16, 0, True)
    assert torch.equal(out[order.cuda()], y_cl.permute(0, 3, 1, 2)[:, :C])
    # a RoI naming an image outside the batch: zeros forward, ignored backward; K = 0: shaped empties
    bad = rois.clone()
    bad[7, 0] = 9.0
    ob = hip.roi_align_nchw(x.cuda().to(dtype), bad.cuda(), 14, 1 / 16, 0, True)
    assert (ob[7] == 0).all() and torch.equal(ob[8:], out[8:]) and torch.equal(ob[:7], out[:7])
    dy_z = dy.clone()
    dy_z[7] = 0
    dxb = hip.roi_align_backward_nchw(dy.cuda().to(dtype), bad.cuda(), (N, C, H, W), 1 / 16, 0, True)
    dxz = hip.roi_align_backward_nchw(dy_z.cuda().to(dtype), rois.cuda(), (N, C, H, W), 1 / 16, 0, True)
    assert torch.equal(dxb, dxz)
    assert hip.roi_align_nchw(x.cuda().to(dtype), torch.zeros(0, 5).cuda(), 14, 1 / 16, 0, True).shape == (0, C, 14, 14)
    assert (hip.roi_align_backward_nchw(torch.zeros(0, C, 14, 14, dtype=dtype).cuda(), torch.zeros(0, 5).cuda(), (N, C, H, W), 1 / 16, 0, True) == 0).all()


@pytest.mark.parametrize("n,t,H,dh", [(5, 80, 8, 96), (3, 37, 4, 64), (2, 128, 2, 128)])
def test_last_token_attention_fwd_bwd(n, t, H, dh):
    """cddmsl_attn_last_fwd/bwd (the mapper's last layer for the one token v2l keeps, clipcap.py:59-83,714-719): one query row per
    sequence against fp32 torch attention on the same bf16 operands, forward and the three gradients."""
    from cddmsl_amd import layers
    d = H * dh
    q = (_rand((n, d), 1) * 0.5).bfloat16().cuda().requires_grad_(True)
    kv = (_rand((n * t, 2 * d), 2) * 0.5).bfloat16().cuda().requires_grad_(True)
    scale = dh ** -0.5
    o = layers.last_token_attention(q, kv, t, H, scale)
    do = _rand((n, d), 3).bfloat16().cuda()
    o.backward(do)
    qf = q.detach().float().requires_grad_(True)
    kvf = kv.detach().float().requires_grad_(True)
    k4 = kvf.view(n, t, 2, H, dh)
    k, v = k4[:, :, 0].permute(0, 2, 1, 3), k4[:, :, 1].permute(0, 2, 1, 3)
    att = torch.softmax((qf.view(n, H, 1, dh) * k).sum(-1) * scale, dim=-1)
    ref = (att.unsqueeze(-1) * v).sum(2).reshape(n, d)
    ref.backward(do.float())
    assert o.dtype == torch.bfloat16 and (o.float() - ref).abs().max() <= 1e-2 * ref.abs().max()
    assert (q.grad.float() - qf.grad).abs().max() <= 1.5e-2 * qf.grad.abs().max()
    assert (kv.grad.float() - kvf.grad).abs().max() <= 1.5e-2 * kvf.grad.abs().max()


@pytest.mark.parametrize("sr", [0, 2])
def test_roi_align_row_sliding_kernel_equals_the_tap_kernel(sr, monkeypatch):
    """k_roi_align_fwd_rows (bf16 throughput forward: each feature pixel of a bin row loaded once, separable weights, sliding
    window) against k_roi_align_fwd (tap by tap, CDDMSL_ROI_ROWS=0) and against the f32 kernel on the same bf16 input: small and
    large RoIs, RoIs hanging over every image border, an empty box, a RoI outside the image, one whose sampling grid exceeds the
    kernel's tables (in-kernel fallback), 64 and 256 channel chunks per row, crops / affine + ReLU / pooled-only outputs."""
    from cddmsl_amd import hip
    g = torch.Generator().manual_seed(23)
    N, H, W = 2, 40, 67
    for C in (512, 2048):
        x = torch.randn(N, H, W, C, generator=g).bfloat16().cuda()
        K = 48
        b = torch.rand(K, 4, generator=g)
        x0, y0 = b[:, 0] * W * 16 - 60, b[:, 1] * H * 16 - 60
        rois = torch.stack([torch.randint(0, N, (K,), generator=g).float(), x0, y0, x0 + 6 + b[:, 2] ** 2 * 600, y0 + 6 + b[:, 3] ** 2 * 500], dim=1)
        rois[3, 3:] = rois[3, 1:3]                                               # empty box
        rois[4, 1:] = torch.tensor([-500.0, -400.0, -300.0, -250.0])             # entirely outside
        rois[5, 1:] = torch.tensor([-4000.0, -3000.0, 6000.0, 5000.0])           # 625 x 500 feature pixels: grid beyond the tables
        rois[6, 1:] = torch.tensor([0.0, 0.0, W * 16.0, H * 16.0])               # the whole image
        rois = rois[rois[:, 0].argsort(stable=True)].contiguous().cuda()
        sc, bs = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
        xf = x.float()
        ref_f32 = hip.roi_align_forward(xf, rois, 14, 14, 1 / 16, sr, True)
        got = {}
        junk = [torch.full((64 << 20,), float("nan"), device="cuda") for _ in range(4)]     # (freed below: fresh outputs land on NaNs, not zeros)
        del junk
        for mode in ("1", "0"):
            monkeypatch.setenv("CDDMSL_ROI_ROWS", mode)
            got[mode] = (hip.roi_align_forward(x, rois, 14, 14, 1 / 16, sr, True),
                         hip.roi_align_forward_affine(x, rois, 14, 14, 1 / 16, sr, True, sc, bs, relu=True),
                         hip.roi_align_forward_affine(x, rois, 14, 14, 1 / 16, sr, True, pooled_only=True))
        ulp = 2.0 ** -7
        mx = float(ref_f32.abs().max())
        assert float((got["1"][0].float() - ref_f32).abs().max()) <= ulp * mx                      # one bf16 rounding of the exact value
        ref_aff = torch.relu(ref_f32 * sc + bs)
        assert float((got["1"][1].float() - ref_aff).abs().max()) <= ulp * float(ref_aff.abs().max())
        ref_pool = ref_f32.view(K, 7, 2, 7, 2, C).mean(dim=(2, 4))
        assert float((got["1"][2].float() - ref_pool).abs().max()) <= ulp * mx
        for a, o in zip(got["1"], got["0"]):
            assert a.shape == o.shape and float((a.float() - o.float()).abs().max()) <= 2 * ulp * mx
        # the box outside the image pools to zero; so does the empty one under the adaptive grid (0 x 0 samples; a fixed grid samples its one point)
        zero = ((rois[:, 3] < -100) | ((rois[:, 3] == rois[:, 1]) if sr == 0 else torch.zeros_like(rois[:, 0], dtype=torch.bool))).nonzero().flatten().tolist()
        assert len(zero) == (2 if sr == 0 else 1) and all(bool((got["1"][0][i] == 0).all()) for i in zero)


@pytest.mark.parametrize("premask", [False, True])
def test_attention_pool_input_gradient_fused_epilogue(premask, monkeypatch):
    """cddmsl_attnpool_dx (the token-gradient product with the map's gradient, the ReLU mask as bit words and the positional
    embedding's gradient in its epilogue) against the stored-dtok route (CDDMSL_ATTNPOOL_DX=0: product, ``dtok[:, 0] +=``,
    cddmsl_attn_tokens_bwd): the map's gradient, every parameter gradient of the pool, and -- the kernel alone -- against an f32
    torch evaluation of its definition.  300 regions (run boundaries of the streaming kernel: 8-region blocks + a tail)."""
    from cddmsl_amd import hip, layers
    K, C, H, P, TP = 300, 2048, 32, 49, 56
    g = torch.Generator().manual_seed(5)
    # the kernel against its definition
    pds = (torch.randn(K, 2 * H, TP, generator=g) * 0.3).bfloat16().cuda()
    pds[:, :, P + 1:] = 0
    zu = torch.randn(K, 2 * H, C, generator=g).bfloat16().cuda()
    g0 = torch.randn(K, C, generator=g).cuda()
    xm = torch.randn(K, P, C, generator=g).cuda()
    bits = ((xm > 0).long() << torch.arange(P, device="cuda").view(1, P, 1)).sum(1)
    gpos = torch.full((P + 1, C), 0.25, device="cuda")
    dx = hip.attnpool_dx(pds, zu, g0, bits, P, gpos)
    dtok = torch.einsum("kht,khc->ktc", pds.float(), zu.float())
    dtok[:, 0] += g0
    want = (dtok[:, 1:P + 1] + dtok[:, :1] / P) * (xm > 0)
    assert float((dx.float() - want).abs().max()) <= 2.0 ** -7 * float(want.abs().max())
    wpos = 0.25 + dtok[:, :P + 1].double().sum(0)
    assert float((gpos.double() - wpos).abs().max()) <= 1e-4 * float(wpos.abs().max())
    # the whole pool, both routes
    pos = (torch.randn(P + 1, C, generator=g) * 0.05).cuda().requires_grad_(True)
    mk = lambda o, i, s: (torch.randn(o, i, generator=g) * s).cuda().requires_grad_(True)
    ws = [mk(C, C, C ** -0.5) for _ in range(3)] + [mk(1024, C, C ** -0.5)]
    bs = [(torch.randn(n, generator=g) * 0.1).cuda().requires_grad_(True) for n in (C, C, C, 1024)]
    x0 = torch.relu(torch.randn(K, 7, 7, C, generator=g)).bfloat16().cuda()
    dout = torch.randn(K, 1024, generator=g).cuda()
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("CDDMSL_ATTNPOOL_DX", mode)
        for t in [pos] + ws + bs:
            t.grad = None
        ap = layers.AttnPoolParams(pos, ws[0], bs[0], ws[1], bs[1], ws[2], bs[2], ws[3], bs[3], H)
        x = x0.clone().requires_grad_(True)
        out = layers.AttnPoolFn.apply(x, ws[0], ap, premask)
        out.backward(dout)
        res[mode] = [x.grad.float()] + [t.grad.float().clone() for t in [pos] + ws + bs]
    names = ["dx", "pos", "q_w", "k_w", "v_w", "c_w", "q_b", "k_b", "v_b", "c_b"]
    for n, a, b in zip(names, res["1"], res["0"]):
        tol = 2.0 ** -6 if n in ("dx", "pos") else 1e-3       # (dx / pos: one bf16 rounding less on the fused route; the rest is untouched)
        assert float((a - b).abs().max()) <= tol * max(float(b.abs().max()), 1e-6), n
    if premask:
        assert bool((res["1"][0][x0.float() <= 0] == 0).all())


def test_sampled_loss_kernels_match_torch():
    """cddmsl_rpn_losses / cddmsl_box_l1 (RPN.losses rpn.py:365-429, box_reg_loss fast_rcnn.py:646-689 over the sampled index lists)
    against the torch expressions they replace -- the dense weighted BCE + gathered L1 of round 2 -- values and gradients; an image
    without ground truth, no positives at all, and the class-agnostic form."""
    from cddmsl_amd import layers
    from cddmsl_amd.modeling.rpn import get_deltas
    g = torch.Generator().manual_seed(31)
    N, A = 3, 500
    anchors = torch.rand(A, 4, generator=g) * 200
    anchors[:, 2:] += anchors[:, :2] + 8
    gts = [torch.tensor([[10.0, 20.0, 120.0, 160.0], [50.0, 40.0, 90.0, 200.0]]), torch.zeros(0, 4), torch.tensor([[5.0, 5.0, 60.0, 70.0]])]
    gt_cat = torch.cat(gts).cuda()
    gt_off = torch.tensor([0, 2, 2]).cuda()
    midx = torch.randint(0, 2, (N, A), generator=g)
    midx[1:] = 0
    pos = torch.tensor([3, 17, 402, 2 * A + 9, 2 * A + 77])                         # images 0 and 2 only (image 1 has no boxes)
    neg = torch.cat([torch.randperm(A, generator=g)[:40] + 20, A + torch.randperm(A, generator=g)[:30], 2 * A + 100 + torch.randperm(300, generator=g)[:20]])
    neg = neg[~torch.isin(neg, pos)]
    logits = torch.randn(N, A, generator=g).cuda().requires_grad_(True)
    deltas = torch.randn(N, A, 4, generator=g).cuda().requires_grad_(True)
    w, norm = (1.0, 1.0, 1.0, 1.0), 256.0 * N
    both = layers.rpn_losses(logits.reshape(-1), deltas.reshape(-1, 4), pos.cuda(), neg.cuda(), midx.cuda().view(-1), gt_cat, gt_off, anchors.cuda(), w, 1.0 / norm)
    (both[0] * 1.5 + both[1] * 0.7).backward()
    lg2, dl2 = logits.detach().clone().requires_grad_(True), deltas.detach().clone().requires_grad_(True)
    lab = torch.full((N * A,), -1.0)
    lab[pos], lab[neg] = 1.0, 0.0
    lab = lab.cuda()
    cls = F.binary_cross_entropy_with_logits(lg2.view(-1), lab.clamp(min=0), weight=(lab >= 0).float(), reduction="sum") / norm
    pc = pos.cuda()
    mb = gt_cat[midx.cuda().view(-1)[pc] + gt_off[pc // A]]
    loc = (dl2.view(-1, 4)[pc] - get_deltas(anchors.cuda()[pc % A], mb, w)).abs().sum() / norm
    (cls * 1.5 + loc * 0.7).backward()
    assert torch.allclose(both[0], cls, rtol=1e-5) and torch.allclose(both[1], loc, rtol=1e-5)
    assert torch.allclose(logits.grad, lg2.grad, rtol=1e-5, atol=1e-9) and torch.allclose(deltas.grad, dl2.grad, rtol=1e-5, atol=1e-9)
    empty = torch.zeros(0, dtype=torch.int64).cuda()
    z = layers.rpn_losses(logits.detach().reshape(-1), deltas.detach().reshape(-1, 4), empty, empty, midx.cuda().view(-1), gt_cat, gt_off, anchors.cuda(), w, 1.0)
    assert float(z[0]) == 0.0 and float(z[1]) == 0.0
    # box head: class-specific columns of the foreground rows
    R, Kc = 64, 20
    d = torch.randn(R, 4 * Kc, generator=g).cuda().requires_grad_(True)
    cls_id = torch.randint(0, Kc + 1, (R,), generator=g).cuda()
    fg = torch.nonzero(cls_id < Kc).flatten()
    src = torch.rand(R, 4, generator=g).cuda() * 100
    src[:, 2:] += src[:, :2] + 4
    tgt = torch.rand(R, 4, generator=g).cuda() * 100
    tgt[:, 2:] += tgt[:, :2] + 4
    bw = (10.0, 10.0, 5.0, 5.0)
    l1 = layers.box_l1(d, fg, cls_id, src, tgt, bw, 1.0 / R)
    l1.backward()
    d2 = d.detach().clone().requires_grad_(True)
    ref = (d2.view(R, Kc, 4)[fg, cls_id[fg]] - get_deltas(src[fg], tgt[fg], bw)).abs().sum() / R
    ref.backward()
    assert torch.allclose(l1, ref, rtol=1e-5) and torch.allclose(d.grad, d2.grad, rtol=1e-5, atol=1e-9)
    d3 = torch.randn(R, 4, generator=g).cuda()
    agn = layers.box_l1(d3, fg, None, src, tgt, bw, 1.0)
    assert torch.allclose(agn, (d3[fg] - get_deltas(src[fg], tgt[fg], bw)).abs().sum(), rtol=1e-5)

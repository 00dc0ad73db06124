"""GPU, through the C-ABI: the product's modules against the golden vectors that the REFERENCE's own leaf modules produced
(tests/golden/ref_*.npz, generator tests/golden/make_golden.py) -- no oracle in between.  The inputs are re-derived from the
same seeds the generator used (the goldens only store outputs); the exact-f32 path has to meet BASELINE.json's 1e-3, the
bf16 throughput path a stated bf16 tolerance (its kernels -- k_attn_small_*, the bf16 MFMA instantiations -- never run on
the f32 path, so they meet the reference's numbers here)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
DT = {"f32": torch.float32, "bf16": torch.bfloat16}


def seeded(shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _npz(name):
    return np.load(os.path.join(G, name))


def _model(dtype):
    from cddmsl_amd import synthetic
    from cddmsl_amd.config import get_cfg
    from cddmsl_amd.modeling import build_model
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "VOC-Experiments", "faster_rcnn_CLIP_R_50_C4.yaml"))
    cfg.merge_from_list(["MODEL.COMPUTE_DTYPE", dtype])
    model = build_model(cfg)
    sd = synthetic.make_state_dict(0)
    model.load_state_dict(sd, strict=False)
    model.train()
    return model, sd


def _close(got, want, rel, what):
    got = got.detach().float().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-12)
    assert err <= rel, (what, float(err), rel)
    return float(err)


# f32: 1e-3 of the tensor's max (BASELINE.json); bf16: 8 significant bits through ~50 conv layers
TOL = {"f32": 1e-3, "bf16": 6e-2}


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_backbone_layer4_attnpool_vs_reference(dtype):
    """ModifiedResNet res4 / res5, layer4 on 14x14 crops, AttentionPool2d (clip_backbone.py:57-107,193-219) vs ref_backbone.npz"""
    g = _npz("ref_backbone.npz")
    model, _ = _model(dtype)
    bb = model.backbone
    rel = TOL[dtype]
    with torch.no_grad():
        o = bb(seeded((2, 3, 64, 96), 11).cuda())
        _close(o["res4"], g["res4_64x96"], rel, "res4")
        _close(o["res5"], g["res5_64x96"], rel, "res5")
        o2 = bb(seeded((1, 3, 224, 224), 12).cuda())
        _close(o2["res5"][0, ::16], g["res5_224_slice"], rel, "res5_224")
        _close(bb.attnpool(o2["res5"]), g["attnpool_224"], rel, "attnpool_224")
        _close(bb.attnpool(seeded((4, 2048, 7, 7), 13).to(DT[dtype]).cuda()), g["attnpool_rand4"], rel if dtype == "f32" else 2e-2, "attnpool_rand4")
        l4 = bb.layer4(seeded((3, 1024, 14, 14), 14).to(DT[dtype]).cuda())
        _close(l4[:, ::8], g["layer4_14"], rel, "layer4_14")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_attnpool_gradients_vs_reference(dtype):
    """AttentionPool2d backward (input, q_proj, c_proj, positional embedding) vs ref_attnpool_grad.npz: the whole reassociated
    query-0 op (layers.AttnPoolFn) against the reference's F.multi_head_attention_forward."""
    g = _npz("ref_attnpool_grad.npz")
    model, _ = _model(dtype)
    ap = model.backbone.attnpool
    x = seeded((4, 2048, 7, 7), 13).to(DT[dtype]).cuda().requires_grad_(True)
    y = ap(x)
    assert y.dtype == torch.float32
    (y * seeded(tuple(y.shape), 15).cuda()).sum().backward()
    rel = 1e-3 if dtype == "f32" else 3e-2
    _close(x.grad[:, ::32], g["gx"], rel, "gx")
    _close(ap.q_proj.weight.grad[::16, ::16], g["gq"], rel, "gq")
    _close(ap.c_proj.weight.grad[::16, ::16], g["gc"], rel, "gc")
    _close(ap.positional_embedding.grad[:, ::16], g["gpos"], rel, "gpos")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_mapper_v2l_vs_reference(dtype):
    """TransformerMapper + v2l (clipcap.py:39-163,714-719) forward and input gradient vs ref_mapper.npz.  bf16 = the fused
    k_attn_small_* kernels + bf16 GEMMs + LayerNorm kernels: the benchmarked mapper path against the reference's numbers."""
    from cddmsl_amd import synthetic
    from cddmsl_amd.modeling import TransformerMapper
    from cddmsl_amd.modeling.clipcap import v2l
    g = _npz("ref_mapper.npz")
    mp = TransformerMapper(compute_dtype=DT[dtype])
    mp.load_state_dict(synthetic.make_mapper_state_dict(1))
    mp.cuda().eval()
    x = seeded((4, 1024), 21).cuda().requires_grad_(True)
    e = v2l(x, mp)
    (e * seeded(tuple(e.shape), 22).cuda()).sum().backward()
    rel = 1e-3 if dtype == "f32" else 3e-2
    _close(e, g["v2l"], rel, "v2l")
    _close(x.grad, g["gx"], rel if dtype == "f32" else 6e-2, "gx")
    # the full 40-token output path (last_only=False: every layer on the fused attention kernel) agrees with the last-token form
    with torch.no_grad():
        full = mp(x.detach())[:, -1]
    _close(full, g["v2l"], rel, "v2l via all tokens")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_box_predictor_and_losses_vs_reference(dtype):
    """FastRCNNOutputLayers (fast_rcnn.py:529-689): cosine logits / T, bbox_pred, focal-scaled CE, box L1, gradient wrt the
    features, classification stats vs ref_fastrcnn.npz (the reference's own module on its own Instances)."""
    from cddmsl_amd.structures import Boxes, Instances
    g = _npz("ref_fastrcnn.npz")
    model, _ = _model(dtype)
    bp = model.roi_heads.box_predictor
    with torch.no_grad():
        bp.bbox_pred.bias.copy_(torch.from_numpy(g["bbox_bias"]))
    feats = torch.from_numpy(g["feats"]).cuda().requires_grad_(True)
    scores, deltas = bp(feats)
    rel = 1e-3 if dtype == "f32" else 2e-2
    _close(scores, g["scores"], 1e-3, "scores")               # cosine logits are f32 on both paths
    _close(deltas, g["deltas"], rel, "deltas")
    inst = Instances((200, 200))
    inst.proposal_boxes, inst.gt_boxes = Boxes(torch.from_numpy(g["pbox"]).cuda()), Boxes(torch.from_numpy(g["gbox"]).cuda())
    inst.gt_classes = torch.from_numpy(g["gcls"]).cuda()
    losses = bp.losses((scores, deltas), [inst])
    assert abs(float(losses["loss_cls"]) - float(g["loss_cls"])) <= 1e-3 * abs(float(g["loss_cls"]))
    assert abs(float(losses["loss_box_reg"]) - float(g["loss_box_reg"])) <= rel * abs(float(g["loss_box_reg"]))
    (losses["loss_cls"] + losses["loss_box_reg"]).backward()
    _close(feats.grad, g["gfeats"], rel, "gfeats")
    st = bp.storage
    got = [float(st["fast_rcnn/cls_accuracy"]), float(st["fast_rcnn/fg_cls_accuracy"]), float(st["fast_rcnn/false_negative"])]
    assert np.allclose(got, g["stats"], atol=1e-6)


def test_rpn_losses_and_proposals_vs_reference():
    """RPN.forward (rpn.py:431-533, proposal_utils.py:22-130) on the reference's tiny map: losses, gradient wrt the features and
    the proposals (same NMS keep count, same boxes and logits in the same order) vs ref_rpn.npz.  f32 path (index parity)."""
    from cddmsl_amd.structures import Boxes, Instances
    g = _npz("ref_rpn.npz")
    model, _ = _model("f32")
    rpn = model.proposal_generator
    rpn.sample_generator = torch.Generator().manual_seed(55)
    feat = torch.from_numpy(g["feat"]).cuda().permute(0, 2, 3, 1).contiguous().requires_grad_(True)      # NHWC
    sizes = [(96, 144), (90, 130)]
    gts = []
    for size, b in zip(sizes, (g["gt0"], g["gt1"])):
        gts.append(Instances(size, gt_boxes=Boxes(torch.from_numpy(b).cuda()), gt_classes=torch.zeros(len(b), dtype=torch.int64).cuda()))
    props, losses = rpn.forward_nhwc(sizes, feat, gts)
    assert abs(float(losses["loss_rpn_cls"]) - float(g["loss_rpn_cls"])) <= 1e-3 * abs(float(g["loss_rpn_cls"]))
    assert abs(float(losses["loss_rpn_loc"]) - float(g["loss_rpn_loc"])) <= 1e-3 * abs(float(g["loss_rpn_loc"]))
    (losses["loss_rpn_cls"] + losses["loss_rpn_loc"]).backward()
    _close(feat.grad.permute(0, 3, 1, 2), g["gfeat"], 1e-3, "gfeat")
    for i in range(2):
        b, s = props[i].proposal_boxes.tensor.cpu().numpy(), props[i].objectness_logits.cpu().numpy()
        assert b.shape == g[f"boxes{i}"].shape, (i, b.shape, g[f"boxes{i}"].shape)          # same keep count
        assert np.allclose(b, g[f"boxes{i}"], rtol=1e-4, atol=1e-3) and np.allclose(s, g[f"logits{i}"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_roi_align_vs_golden(dtype):
    """RoIAlign forward / backward on the multi-image case of ref_roialign.npz (adaptive grid, aligned, out-of-image and empty
    boxes; generated by make_golden.py's independent torch implementation -- reference-held numbers exist for the 5x5 KAT only)."""
    from cddmsl_amd import hip
    g = _npz("ref_roialign.npz")
    T = DT[dtype]
    x = torch.from_numpy(g["x"])                                                  # [2, 6, 9, 11] -> channels padded to 8
    xp = torch.zeros(2, 9, 11, 8)
    xp[..., :6] = x.permute(0, 2, 3, 1)
    rois = torch.from_numpy(g["rois"])
    order = torch.argsort(rois[:, 0], stable=True)                               # the C-ABI takes RoIs grouped by image
    rs = rois[order].contiguous().cuda()
    out = hip.roi_align_forward(xp.to(T).cuda(), rs, 4, 4, 1.0 / 16, 0, True)
    inv = torch.argsort(order)
    got = out.float().cpu()[inv][..., :6].permute(0, 3, 1, 2)
    rel = 1e-5 if dtype == "f32" else 1.5e-2
    _close(got, g["out"], rel, "out")
    w = torch.zeros(5, 4, 4, 8)
    w[..., :6] = torch.from_numpy(g["w"]).permute(0, 2, 3, 1)
    counts = torch.bincount(rs[:, 0].long().cpu(), minlength=2)
    start = torch.cat([torch.zeros(1, dtype=torch.int64), counts.cumsum(0)]).to(torch.int32).cuda()
    dx = hip.roi_align_backward(w[order].to(T).contiguous().cuda(), rs, start, (2, 9, 11, 8), 1.0 / 16, 0, True)
    _close(dx.float().cpu()[..., :6].permute(0, 3, 1, 2), g["gx"], rel, "gx")


# ------------------------------------------------------------------------------------------------ the whole step vs the reference
# tests/golden/ref_step_w*_r*.npz: the reference's own GeneralizedRCNN.forward x3 + backward as SimpleTrainer.run_step composes
# them (tests/golden/make_golden_step.py), world size 1 and a real two-rank gloo run.  Here: the HIP step (exact f32) directly
# against those numbers.
STEP = dict(H=96, W=128, per_rank=2, roi_batch=16, pre=200, post=60, seed=77)
GRAD_SLICES = {"backbone.layer2.0.conv1.weight": np.s_[::4, ::8], "backbone.layer3.5.conv2.weight": np.s_[::16, ::16],
               "backbone.layer4.0.downsample.0.weight": np.s_[::64, ::32], "backbone.attnpool.k_proj.weight": np.s_[::64, ::64],
               "backbone.attnpool.positional_embedding": np.s_[:, ::64], "proposal_generator.rpn_head.conv.weight": np.s_[::64, ::64],
               "proposal_generator.rpn_head.anchor_deltas.bias": np.s_[:], "roi_heads.box_predictor.bbox_pred.weight": np.s_[::4, ::32],
               "projector.0.weight": np.s_[::24, ::24], "projector.2.bias": np.s_[:]}


def _hip_reference_step(rank, share):
    """the HIP trainer on rank ``rank``'s batch of the golden run; returns (trainer, losses)"""
    from cddmsl_amd import synthetic
    from cddmsl_amd.config import get_cfg
    from cddmsl_amd.engine import SimpleTrainer
    from cddmsl_amd.modeling import TransformerMapper, build_model
    from cddmsl_amd.solver import build_optimizer
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "VOC-Experiments", "faster_rcnn_CLIP_R_50_C4.yaml"))
    cfg.merge_from_list(["MODEL.COMPUTE_DTYPE", "f32", "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", STEP["roi_batch"], "MODEL.RPN.PRE_NMS_TOPK_TRAIN",
                         STEP["pre"], "MODEL.RPN.POST_NMS_TOPK_TRAIN", STEP["post"], "MODEL.KD_REGULRAZIATION", True])
    model = build_model(cfg)
    model.load_state_dict(synthetic.drift_offline(synthetic.make_state_dict(0)), strict=False)
    mapper = TransformerMapper(compute_dtype=model.compute_dtype)
    mapper.load_state_dict(synthetic.make_mapper_state_dict(1))
    mapper.to(model.device).eval()
    g = torch.Generator().manual_seed(STEP["seed"] + rank)          # the reference's one global stream, seed + rank
    model.proposal_generator.sample_generator = model.roi_heads.sample_generator = model.region_generator = g
    model.train()
    batch = synthetic.make_batch(STEP["per_rank"], STEP["H"], STEP["W"], rank=rank, num_gt=3)
    tr = SimpleTrainer(model, iter([batch]), build_optimizer(cfg, model), cfg, clipcap_model=mapper, metrics_period=0)
    tr.iter = 20000
    tr.share_source_pass = tr.fuse_consistency = share
    tr.buckets.zero()
    ld = tr.compute_losses(batch)
    sum(ld.values()).backward()
    torch.cuda.synchronize()
    return tr, {k: float(v.detach()) for k, v in ld.items()}


def _check_against_step_golden(g, model, losses):
    for k, v in losses.items():
        want = float(g["loss/" + k])
        assert abs(v - want) <= 1e-3 * abs(want) + 1e-6, (k, v, want)
    assert {k[5:] for k in g.files if k.startswith("loss/")} == set(losses)
    params = dict(model.named_parameters())
    worst = 0.0
    for n, nrm, mx in zip([str(n) for n in g["grad_names"]], g["grad_norms"], g["grad_absmax"]):
        gr = params[n].grad.detach().float().cpu()
        if mx < 1e-8:
            assert float(gr.abs().max()) < 1e-6, n
            continue
        e = abs(float(gr.double().norm()) - nrm) / nrm
        worst = max(worst, e)
        assert e <= 5e-3, (n, float(gr.norm()), nrm)
    for n, ix in GRAD_SLICES.items():
        want = g["grad/" + n]
        got = params[n].grad.detach().float().cpu().numpy()[ix]
        assert np.abs(got - want).max() <= 5e-3 * np.abs(want).max() + 1e-9, n
    return worst


@pytest.mark.parametrize("share", [True, False])
def test_run_step_vs_reference_world1(share):
    """HIP step (f32) vs the reference's run_step composition at world size 1: 7 losses within 1e-3, every gradient tensor's norm
    within 5e-3 and slices of ten tensors within 5e-3 of their max -- with the shared source pass / fused consistency pass on
    and off (the reference recomputes everything)."""
    g = _npz("ref_step_w1_r0.npz")
    tr, losses = _hip_reference_step(0, share)
    worst = _check_against_step_golden(g, tr.model, losses)
    print("world 1 vs reference: losses", losses, "worst grad-norm rel err", worst)


def _w2_worker(rank, port, ret):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        g = _npz(f"ref_step_w2_r{rank}.npz")
        tr, losses = _hip_reference_step(rank, True)
        worst = _check_against_step_golden(g, tr.model, losses)
        # DDP gradient mean (engine/defaults.py:74) on the flat buffer: the mean of both ranks' reference gradients
        go = _npz(f"ref_step_w2_r{1 - rank}.npz")
        tr.buckets.all_reduce_mean()
        torch.cuda.synchronize()
        params = dict(tr.model.named_parameters())
        for n, ix in GRAD_SLICES.items():
            want = 0.5 * (g["grad/" + n] + go["grad/" + n])
            got = params[n].grad.detach().float().cpu().numpy()[ix]
            assert np.abs(got - want).max() <= 5e-3 * np.abs(want).max() + 1e-9, n
        ret[rank] = (losses, worst)
    finally:
        dist.destroy_process_group()


def test_run_step_vs_reference_world2():
    """Two ranks (gloo; both on this box's one GPU -- RCCL needs a device per rank) vs the reference's REAL two-rank run: the
    cross-rank contrastive batch through GatherLayer (all_gather forward, own-slice backward), per-rank losses and gradients,
    then the flat-buffer gradient mean against the mean of the two reference gradients."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_w2_worker, args=(port, ret), nprocs=2, join=True)
    assert len(ret) == 2
    assert abs(ret[0][0]["cont_loss"] - ret[1][0]["cont_loss"]) < 1e-5 and abs(ret[0][0]["cont_region_loss"] - ret[1][0]["cont_region_loss"]) < 1e-5
    print("world 2 vs reference:", dict(ret))


def _overlap_worker(rank, port, ret):
    import itertools
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        out = {}
        for overlap in (True, False):
            tr, _ = _hip_reference_step(rank, True)
            tr.buckets.overlap = overlap
            from cddmsl_amd import synthetic
            batch = synthetic.make_batch(STEP["per_rank"], STEP["H"], STEP["W"], rank=rank, num_gt=3)
            tr._data_loader_iter = itertools.cycle([batch])
            g = torch.Generator().manual_seed(STEP["seed"] + rank)
            tr.model.proposal_generator.sample_generator = tr.model.roi_heads.sample_generator = tr.model.region_generator = g
            logs = []
            for _ in range(3):
                tr.run_step()
                logs.append(list(tr.buckets.launch_log))
            torch.cuda.synchronize()
            params = dict(tr.model.named_parameters())
            out[overlap] = ({k: params[k].detach().float().cpu().clone() for k in GRAD_SLICES}, logs)
        (wa, la), (wb, lb) = out[True], out[False]
        for k in wa:
            assert torch.allclose(wa[k], wb[k], rtol=1e-4, atol=1e-6), k
        total = max(a for _, a in la[1])
        assert all(a == max(x for _, x in la[0]) for _, a in la[0])                  # first step of a signature: counted, reduced after backward
        assert any(a < total for _, a in la[1]) and la[1] == la[2], la               # then buckets leave while backward is still writing
        assert lb[1] and all(a == lb[1][0][1] for _, a in lb[1])                     # overlap off: everything after backward
        ret[rank] = la[1]
    finally:
        dist.destroy_process_group()


def test_overlapped_allreduce_gives_the_same_training_two_ranks():
    """engine.GradBuckets with the all-reduce overlapped with backward (side stream, event behind each bucket's last gradient
    write) vs the plain post-backward reduction: the same weights after three optimizer steps on two ranks; buckets holding the
    RoI head / attention pool / RPN gradients go on the wire while the backbone's backward is still running."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_overlap_worker, args=(port, ret), nprocs=2, join=True)
    assert len(ret) == 2
    print("bucket launches (bucket, gradient writes announced before it left):", ret[0])


def test_roi_label_and_sample_vs_reference():
    """CLIPRes5ROIHeads.label_and_sample_proposals (batched HIP IoU + matcher, host-replayed sampling) vs the reference's own
    ROIHeads.label_and_sample_proposals (roi_heads.py:236-319; tests/golden/ref_roi_sampling.npz): the SAME sampled proposals in
    the same order -- boxes, objectness logits, classes, matched ground-truth boxes -- for an image with 3 boxes, one without
    ground truth and one with fewer candidates than the batch.  Index parity, bit for bit."""
    from cddmsl_amd.config import get_cfg
    from cddmsl_amd.modeling import build_model
    from cddmsl_amd.structures import Boxes, Instances
    g = _npz("ref_roi_sampling.npz")
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "VOC-Experiments", "faster_rcnn_CLIP_R_50_C4.yaml"))
    cfg.merge_from_list(["MODEL.COMPUTE_DTYPE", "f32", "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 64])
    heads = build_model(cfg).roi_heads
    heads.sample_generator = torch.Generator().manual_seed(92)
    props, tgts = [], []
    for i in range(3):
        p = Instances((200, 240))
        p.proposal_boxes, p.objectness_logits = Boxes(torch.from_numpy(g[f"boxes{i}"]).cuda()), torch.from_numpy(g[f"logits{i}"]).cuda()
        t = Instances((200, 240), gt_boxes=Boxes(torch.from_numpy(g[f"gt_boxes{i}"]).reshape(-1, 4).cuda()),
                      gt_classes=torch.from_numpy(g[f"gt_classes{i}"]).long().cuda())
        props.append(p)
        tgts.append(t)
    out = heads.label_and_sample_proposals(props, tgts)
    for i, o in enumerate(out):
        assert np.array_equal(o.proposal_boxes.tensor.cpu().numpy(), g[f"s_boxes{i}"]), i
        assert np.array_equal(o.gt_classes.cpu().numpy(), g[f"s_classes{i}"]), i
        assert np.allclose(o.objectness_logits.cpu().numpy(), g[f"s_logits{i}"])
        assert o.has("gt_boxes") == (f"s_gt_boxes{i}" in g.files)
        if o.has("gt_boxes"):
            assert np.array_equal(o.gt_boxes.tensor.cpu().numpy(), g[f"s_gt_boxes{i}"]), i
    assert np.allclose([float(heads.storage["roi_head/num_fg_samples"]), float(heads.storage["roi_head/num_bg_samples"])], g["num_fg_bg"])

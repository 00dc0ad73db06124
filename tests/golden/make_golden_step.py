"""Generates tests/golden/ref_step_w{1,2}_r*.npz, ref_roi_sampling.npz, ref_sgd.npz, ref_stock_resnet.npz by running the REFERENCE's own code above the
leaf modules (read from /root/reference, never copied; runs only in the build container):

  * ``GeneralizedRCNN.forward`` (modeling/meta_arch/rcnn.py:351-623) for the three branches the trainer calls -- supervised,
    ``caption_consistency`` (-> ``v2l_contrastive`` :255-319) and ``caption_consistency_regionLevel`` (:422-470) -- composed
    exactly as ``SimpleTrainer.run_step`` does (engine/train_loop.py:311-383: one batch, three forwards, ``sum(losses)``,
    one backward), on the reference's own ``ModifiedResNet`` x2, ``RPN``, ``CLIPRes5ROIHeads`` + ``FastRCNNOutputLayers``,
    ``TransformerMapper`` and ``GatherLayer`` (gather.py:5-20), inside a REAL ``torch.distributed`` gloo group: world size 1
    and world size 2 (two processes, each with its own half of the batch -- the cross-rank contrastive batch and the
    "backward keeps the own slice" semantics are the reference's, not a simulation);
  * ``ROIHeads.label_and_sample_proposals`` (roi_heads/roi_heads.py:236-319) on seeded proposals;
  * ``maybe_add_gradient_clipping(cfg, torch.optim.SGD)`` (solver/build.py:43-110): per-parameter norm clip + SGD momentum / wd.

Third-party arithmetic the reference calls but does not vendor comes from make_golden.py's independent definitions
(roi_align, nms, smooth_l1) plus, here, torchvision's ``Resize(224, bicubic, antialias=None)`` / ``CenterCrop`` on tensors
(= ``F.interpolate(mode='bicubic', align_corners=False)`` and the rounded centre crop: published behaviour, parity unpinned).

usage:  python tests/golden/make_golden_step.py
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

H, W, PER_RANK = 96, 128, 2
ROI_BATCH, PRE_NMS, POST_NMS = 16, 200, 60
SEED = 77


# ---------------------------------------------------------------- torchvision.transforms on tensors (published behaviour)
class Resize(torch.nn.Module):
    def __init__(self, size, interpolation=None, max_size=None, antialias=None):
        super().__init__()
        self.size = size

    def forward(self, x):
        h, w = x.shape[-2:]
        short, long = (w, h) if w <= h else (h, w)
        new_short, new_long = self.size, int(self.size * long / short)
        nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
        return F.interpolate(x, size=(nh, nw), mode="bicubic", align_corners=False)


class CenterCrop(torch.nn.Module):
    def __init__(self, size):
        super().__init__()
        self.size = size

    def forward(self, x):
        th, tw = self.size
        h, w = x.shape[-2:]
        top, left = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
        return x[..., top:top + th, left:left + tw]


def setup_step():
    mg.setup()
    import numpy.lib
    if not hasattr(numpy.lib, "pad"):          # rcnn.py:5 ``from numpy.lib import pad`` (unused there; NumPy >= 2 dropped the alias)
        numpy.lib.pad = np.pad
    tv = importlib.import_module("torchvision.transforms")
    tv.Resize, tv.CenterCrop = Resize, CenterCrop
    tvf = importlib.import_module("torchvision.transforms.functional")
    tvf.InterpolationMode = types.SimpleNamespace(BICUBIC="bicubic")
    # vendored Normalize (data/transforms/torchvision_transforms/transforms.py:189-221 -> functional.normalize :295-336):
    # the file imports PIL / accimage helpers at module scope; its Normalize is a per-channel (x - mean) / std
    mg._pkg("detectron2.data.transforms", "detectron2/data/transforms")
    mg._pkg("detectron2.data.transforms.torchvision_transforms", "detectron2/data/transforms/torchvision_transforms")
    try:
        tvt = importlib.import_module("detectron2.data.transforms.torchvision_transforms.transforms")
        tvt.Normalize
    except Exception as e:  # noqa: BLE001
        raise RuntimeError(f"vendored torchvision_transforms.transforms did not import: {e!r}")
    du = types.ModuleType("detectron2.data.detection_utils")
    du.convert_image_to_rgb = mg._Anything
    sys.modules["detectron2.data.detection_utils"] = du
    L = sys.modules["detectron2.layers"]
    L.DeformConv = L.ModulatedDeformConv = L.ConvTranspose2d = L.interpolate = L.paste_masks_in_image = mg._Anything
    bbb = types.ModuleType("detectron2.modeling.backbone.build")
    bbb.BACKBONE_REGISTRY = mg.Registry("BACKBONE")
    sys.modules["detectron2.modeling.backbone.build"] = bbb
    mg._pkg("detectron2.solver", "detectron2/solver")
    # packages the meta-arch imports by name
    bbp = sys.modules["detectron2.modeling.backbone"]
    bbp.Backbone = importlib.import_module("detectron2.modeling.backbone.backbone").Backbone
    bbp.build_backbone = mg._Anything
    pp = types.ModuleType("detectron2.modeling.postprocessing")
    pp.detector_postprocess = mg._Anything
    sys.modules["detectron2.modeling.postprocessing"] = pp
    pgb = types.ModuleType("detectron2.modeling.proposal_generator.build")
    pgb.PROPOSAL_GENERATOR_REGISTRY = mg.Registry("PROPOSAL_GENERATOR")
    pgb.build_proposal_generator = mg._Anything
    sys.modules["detectron2.modeling.proposal_generator.build"] = pgb
    sys.modules["detectron2.modeling.proposal_generator"].build_proposal_generator = mg._Anything
    rh = sys.modules["detectron2.modeling.roi_heads"]
    rh.build_roi_heads = mg._Anything
    for n in ("box_head", "keypoint_head", "mask_head"):
        m = types.ModuleType(f"detectron2.modeling.roi_heads.{n}")
        setattr(m, "build_" + n, mg._Anything)
        sys.modules[f"detectron2.modeling.roi_heads.{n}"] = m
    mb = types.ModuleType("detectron2.modeling.meta_arch.build")
    mb.META_ARCH_REGISTRY = mg.Registry("META_ARCH")
    sys.modules["detectron2.modeling.meta_arch.build"] = mb
    ev = sys.modules["detectron2.utils.events"]
    ev.EventStorage = mg._Anything


def build_reference_model(num_classes=20):
    """The reference's own modules, wired with explicit keyword arguments (``@configurable`` split, config/config.py:163-247)."""
    from cddmsl_amd import synthetic
    S = sys.modules["detectron2.structures"]
    L = sys.modules["detectron2.layers"]
    cb = importlib.import_module("detectron2.modeling.backbone.clip_backbone")
    cc = importlib.import_module("detectron2.modeling.backbone.clipcap.clipcap")
    mt = importlib.import_module("detectron2.modeling.matcher")
    br = importlib.import_module("detectron2.modeling.box_regression")
    ag = importlib.import_module("detectron2.modeling.anchor_generator")
    rp = importlib.import_module("detectron2.modeling.proposal_generator.rpn")
    fr = importlib.import_module("detectron2.modeling.roi_heads.fast_rcnn")
    pl = importlib.import_module("detectron2.modeling.poolers")
    crh = importlib.import_module("detectron2.modeling.roi_heads.clip_roi_heads")
    rc = importlib.import_module("detectron2.modeling.meta_arch.rcnn")
    sd = synthetic.drift_offline(synthetic.make_state_dict(0, num_classes=num_classes))     # teacher != student: a live kd_loss

    def backbone(prefix):
        net = cb.ModifiedResNet(layers=[3, 4, 6, 3], output_dim=1024, heads=32, input_resolution=224, width=64,
                                out_features=["res4", "res5"], freeze_at=2, depth=50, pool_vec=False)
        net.load_state_dict({k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}, strict=True)
        return net

    student, offline = backbone("backbone."), backbone("offline_backbone.")
    for p in offline.parameters():                                  # rcnn.py:106-107
        p.requires_grad = False
    offline.eval()
    head = rp.StandardRPNHead(in_channels=1024, num_anchors=15, box_dim=4)
    pfx = "proposal_generator.rpn_head."
    head.load_state_dict({k[len(pfx):]: v for k, v in sd.items() if k.startswith(pfx)})
    gen = ag.DefaultAnchorGenerator(sizes=[[32, 64, 128, 256, 512]], aspect_ratios=[[0.5, 1.0, 2.0]], strides=[16], offset=0.0)
    rpn = rp.RPN(in_features=["res4"], head=head, anchor_generator=gen,
                 anchor_matcher=mt.Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=True),
                 box2box_transform=br.Box2BoxTransform(weights=(1.0, 1.0, 1.0, 1.0)),
                 batch_size_per_image=256, positive_fraction=0.5, pre_nms_topk=(PRE_NMS, PRE_NMS),
                 post_nms_topk=(POST_NMS, POST_NMS), nms_thresh=0.7, min_box_size=0.0, anchor_boundary_thresh=-1.0,
                 loss_weight=1.0, box_reg_loss_type="smooth_l1", smooth_l1_beta=0.0)
    pred = fr.FastRCNNOutputLayers(
        L.ShapeSpec(channels=2048, height=1, width=1), box2box_transform=br.Box2BoxTransform(weights=(10.0, 10.0, 5.0, 5.0)),
        num_classes=num_classes, clip_cls_emb=(True, None, "CLIPRes5ROIHeads", 1024), bg_cls_loss_weight=0.2,
        openset_test=(None, None, 0.01, 0.5), loss_weight={"loss_box_reg": 1.0})
    with torch.no_grad():
        for n in ("cls_score.weight", "bbox_pred.weight", "bbox_pred.bias"):
            dict(pred.named_parameters())[n].copy_(sd["roi_heads.box_predictor." + n])
    heads = crh.CLIPRes5ROIHeads(
        in_features=["res4"], pooler=pl.ROIPooler(output_size=14, scales=(1.0 / 16,), sampling_ratio=0, pooler_type="ROIAlignV2"),
        res5=None, box_predictor=pred, num_classes=num_classes, batch_size_per_image=ROI_BATCH, positive_fraction=0.25,
        proposal_matcher=mt.Matcher([0.5], [0, 1], allow_low_quality_matches=False), proposal_append_gt=True)
    model = rc.GeneralizedRCNN(offline_backbone=offline, backbone=student, proposal_generator=rpn, roi_heads=heads,
                               pixel_mean=(0.48145466, 0.4578275, 0.40821073), pixel_std=(0.26862954, 0.26130258, 0.27577711),
                               input_format="RGB", vis_period=0, use_clip_c4=True, use_clip_attpool=True)
    with torch.no_grad():
        for n in ("0.weight", "0.bias", "2.weight", "2.bias"):
            dict(model.projector.named_parameters())[n].copy_(sd["projector." + n])
    mapper = cc.TransformerMapper(1024, 768, 40, 40, 8)
    mapper.load_state_dict(synthetic.make_mapper_state_dict(1), strict=True)
    mapper.eval()
    for p in mapper.parameters():                                   # train_loop.py:286-288
        p.requires_grad = False
    model.train()
    return model, mapper, sd, S


def reference_names(model):
    """reference parameter name -> state-dict key of this repo (they are the same names; the box predictor sits under roi_heads)"""
    return {n: p for n, p in model.named_parameters() if p.requires_grad}


def run_rank(rank, world, port, kd, out_path):
    torch.set_num_threads(4)
    setup_step()
    from cddmsl_amd import synthetic
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        model, mapper, sd, S = build_reference_model()
        raw = synthetic.make_batch(PER_RANK, H, W, rank=rank, num_gt=3)
        data = []
        for x in raw:
            inst = S.Instances((H, W))
            inst.gt_boxes, inst.gt_classes = S.Boxes(x["instances"]["gt_boxes"]), x["instances"]["gt_classes"]
            data.append({"image": x["image"], "image_trgt": x["image_trgt"], "instances": inst, "height": H, "width": W})
        rec = {}
        rpn, heads = model.proposal_generator, model.roi_heads
        orig_rpn, orig_ls = rpn.forward, heads.label_and_sample_proposals

        def rpn_fwd(images, features, gt_instances=None):
            props, losses = orig_rpn(images, features, gt_instances)
            rec.setdefault("proposals", []).append([(p.proposal_boxes.tensor.clone(), p.objectness_logits.clone()) for p in props])
            return props, losses

        def ls(proposals, targets):
            out = orig_ls(proposals, targets)
            rec["sampled"] = [(p.proposal_boxes.tensor.clone(), p.gt_classes.clone()) for p in out]
            return out

        rpn.forward, heads.label_and_sample_proposals = rpn_fwd, ls
        orig_fgf = heads.forward_get_features

        def fgf(fs, ft, proposals, **kw):
            rec["region_boxes"] = [p.proposal_boxes.tensor.clone() for p in proposals]
            return orig_fgf(fs, ft, proposals, **kw)

        heads.forward_get_features = fgf
        torch.cuda.empty_cache = lambda: None                      # rcnn.py:375 (no GPU here)
        # ---- SimpleTrainer.run_step, iteration > 10000 (train_loop.py:330-370)
        torch.manual_seed(SEED + rank)                             # the reference's global RNG (seed + rank, engine/defaults.py:221-222)
        model.zero_grad()
        loss_dict = model(data)
        loss = {}
        loss.update(model(data, clipcap_model=mapper, branch="caption_consistency", KD_regularization=kd))
        loss["cont_region_loss"] = model(data, clipcap_model=mapper, branch="caption_consistency_regionLevel", KD_regularization=kd)
        loss_dict.update(loss)
        losses = sum(loss_dict.values())
        losses.backward()
        out = {"loss/" + k: np.float64(v.detach()) for k, v in loss_dict.items()}
        params = reference_names(model)
        names = sorted(params)
        out["grad_names"] = np.array(names)
        out["grad_norms"] = np.array([float(params[n].grad.double().norm()) if params[n].grad is not None else -1.0 for n in names])
        out["grad_absmax"] = np.array([float(params[n].grad.abs().max()) if params[n].grad is not None else -1.0 for n in names])
        for n, sl in (("backbone.layer2.0.conv1.weight", (slice(None, None, 4), slice(None, None, 8))),
                      ("backbone.layer3.5.conv2.weight", (slice(None, None, 16), slice(None, None, 16))),
                      ("backbone.layer4.0.downsample.0.weight", (slice(None, None, 64), slice(None, None, 32))),
                      ("backbone.attnpool.k_proj.weight", (slice(None, None, 64), slice(None, None, 64))),
                      ("backbone.attnpool.positional_embedding", (slice(None), slice(None, None, 64))),
                      ("proposal_generator.rpn_head.conv.weight", (slice(None, None, 64), slice(None, None, 64))),
                      ("proposal_generator.rpn_head.anchor_deltas.bias", (slice(None),)),
                      ("roi_heads.box_predictor.bbox_pred.weight", (slice(None, None, 4), slice(None, None, 32))),
                      ("projector.0.weight", (slice(None, None, 24), slice(None, None, 24))),
                      ("projector.2.bias", (slice(None),))):
            out["grad/" + n] = params[n].grad[sl].numpy().copy()
        for i, (b, s) in enumerate(rec["proposals"][0]):
            out[f"prop_boxes{i}"], out[f"prop_logits{i}"] = b.numpy(), s.numpy()
        assert all(torch.equal(a[0], b[0]) for a, b in zip(rec["proposals"][0], rec["proposals"][1])), "the region-level RPN pass sees the same features"
        for i, (b, c) in enumerate(rec["sampled"]):
            out[f"sampled_boxes{i}"], out[f"sampled_classes{i}"] = b.numpy(), c.numpy()
        for i, b in enumerate(rec["region_boxes"]):
            out[f"region_boxes{i}"] = b.numpy()
        out["meta"] = np.array([world, rank, int(kd), SEED + rank, H, W, PER_RANK, ROI_BATCH, PRE_NMS, POST_NMS])
        np.savez_compressed(out_path.format(rank=rank), **out)
        print(f"world {world} rank {rank}:", {k: round(float(v), 6) for k, v in loss_dict.items()}, flush=True)
    finally:
        dist.destroy_process_group()


def roi_sampling_golden():
    """ROIHeads.label_and_sample_proposals (roi_heads.py:236-319, + add_ground_truth_to_proposals proposal_utils.py:133-200)"""
    setup_step()
    S = sys.modules["detectron2.structures"]
    mt = importlib.import_module("detectron2.modeling.matcher")
    rhm = importlib.import_module("detectron2.modeling.roi_heads.roi_heads")
    heads = rhm.ROIHeads(num_classes=20, batch_size_per_image=64, positive_fraction=0.25,
                         proposal_matcher=mt.Matcher([0.5], [0, 1], allow_low_quality_matches=False), proposal_append_gt=True)
    g = torch.Generator().manual_seed(91)
    out = {}
    props, tgts = [], []
    for i, (ngt, n) in enumerate(((3, 300), (0, 120), (1, 40))):     # an image without ground truth, one with fewer candidates than the batch
        gt = torch.rand(ngt, 4, generator=g) * 100
        gt[:, 2:] = gt[:, :2] + 30 + torch.rand(ngt, 2, generator=g) * 60
        pr = torch.rand(n, 4, generator=g) * 140
        pr[:, 2:] = pr[:, :2] + 10 + torch.rand(n, 2, generator=g) * 70
        if ngt:
            pr[:40:4] = gt[torch.arange(10) % ngt] + torch.randn(10, 4, generator=g) * 3.0      # near-GT boxes -> foreground candidates
        p = S.Instances((200, 240))
        p.proposal_boxes, p.objectness_logits = S.Boxes(pr), torch.randn(n, generator=g)
        t = S.Instances((200, 240))
        t.gt_boxes, t.gt_classes = S.Boxes(gt), torch.randint(0, 20, (ngt,), generator=g)
        props.append(p)
        tgts.append(t)
        out[f"gt_boxes{i}"], out[f"gt_classes{i}"] = gt.numpy(), t.gt_classes.numpy()
        out[f"boxes{i}"], out[f"logits{i}"] = pr.numpy(), p.objectness_logits.numpy()
    torch.manual_seed(92)
    res = heads.label_and_sample_proposals(props, tgts)
    for i, r in enumerate(res):
        out[f"s_boxes{i}"], out[f"s_logits{i}"], out[f"s_classes{i}"] = r.proposal_boxes.tensor.numpy(), r.objectness_logits.numpy(), r.gt_classes.numpy()
        if r.has("gt_boxes"):
            out[f"s_gt_boxes{i}"] = r.gt_boxes.tensor.numpy()
    out["num_fg_bg"] = np.array([mg.STORAGE.scalars["roi_head/num_fg_samples"], mg.STORAGE.scalars["roi_head/num_bg_samples"]])
    np.savez_compressed(os.path.join(HERE, "ref_roi_sampling.npz"), **out)
    print("roi sampling ok", [len(r) for r in res], out["num_fg_bg"])


def sgd_golden():
    """solver/build.py:43-110: ``maybe_add_gradient_clipping(cfg, torch.optim.SGD)`` -> per-parameter clip_grad_norm_(p, 5.0) then
    SGD(momentum 0.9, weight decay 1e-4, nesterov False); three steps with changing learning rates (the LR schedule is pinned by
    the scheduler KAT)."""
    setup_step()
    sys.modules["detectron2.config"].CfgNode = dict
    sb = importlib.import_module("detectron2.solver.build")
    ns = types.SimpleNamespace
    cfg = ns(SOLVER=ns(CLIP_GRADIENTS=ns(ENABLED=True, CLIP_TYPE="norm", CLIP_VALUE=5.0, NORM_TYPE=2.0)))
    cls = sb.maybe_add_gradient_clipping(cfg, torch.optim.SGD)
    g = torch.Generator().manual_seed(101)
    shapes = {"a.weight": (16, 8, 3, 3), "b.weight": (32, 16), "b.bias": (32,), "c.weight": (4, 4)}
    params = {k: torch.nn.Parameter(torch.randn(*s, generator=g)) for k, s in shapes.items()}
    opt = cls([{"params": [p], "lr": 0.002, "weight_decay": 1e-4} for p in params.values()], lr=0.002, momentum=0.9, nesterov=False)
    out = {}
    for k, p in params.items():
        out["w0/" + k] = p.detach().numpy().copy()
    lrs = [0.002, 0.002, 0.0002]
    scales = {"a.weight": 3.0, "b.weight": 0.01, "b.bias": 40.0, "c.weight": 1.0}      # norms above and below the clip value
    for step, lr in enumerate(lrs):
        for grp in opt.param_groups:
            grp["lr"] = lr
        for k, p in params.items():
            gr = torch.randn(*shapes[k], generator=g) * scales[k]
            out[f"g{step}/" + k] = gr.numpy().copy()
            p.grad = gr.clone()
        opt.step()
        for k, p in params.items():
            out[f"w{step + 1}/" + k] = p.detach().numpy().copy()
    out["lrs"] = np.array(lrs)
    np.savez_compressed(os.path.join(HERE, "ref_sgd.npz"), **out)
    print("sgd ok")


def stock_resnet_golden():
    """Stock Detectron2 ResNet-50 (backbone/resnet.py:100-210 BottleneckBlock with the stride in the 1x1, :330-359 BasicStem,
    :362-459 ResNet) up to res4, and the RoI head's res5 stage (roi_heads.py:440-463 _build_res5_block) with input / weight
    gradients -- the architecture of BASELINE.json configs[0]."""
    setup_step()
    from cddmsl_amd import synthetic
    rn = importlib.import_module("detectron2.modeling.backbone.resnet")
    sd = synthetic.make_state_dict_r50(0)
    stem = rn.BasicStem(in_channels=3, out_channels=64, norm="FrozenBN")
    stages, cin, cout, bott = [], 64, 256, 64
    for idx, nb in enumerate([3, 4, 6]):
        stages.append(rn.ResNet.make_stage(block_class=rn.BottleneckBlock, num_blocks=nb, stride_per_block=[1 if idx == 0 else 2] + [1] * (nb - 1),
                                           in_channels=cin, out_channels=cout, norm="FrozenBN", bottleneck_channels=bott,
                                           stride_in_1x1=True, dilation=1, num_groups=1))
        cin, cout, bott = cout, cout * 2, bott * 2
    net = rn.ResNet(stem, stages, out_features=["res4"], freeze_at=2)
    net.load_state_dict({k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}, strict=True)
    x = mg.seeded((2, 3, 64, 96), 111, 50.0)
    with torch.no_grad():
        res4 = net(x)["res4"]
    res5 = torch.nn.Sequential(*rn.ResNet.make_stage(rn.BottleneckBlock, 3, stride_per_block=[2, 1, 1], in_channels=1024,
                                                     bottleneck_channels=512, out_channels=2048, num_groups=1, norm="FrozenBN",
                                                     stride_in_1x1=True))
    res5.load_state_dict({k[len("roi_heads.res5."):]: v for k, v in sd.items() if k.startswith("roi_heads.res5.")}, strict=True)
    xr = mg.seeded((3, 1024, 14, 14), 112).requires_grad_(True)
    y = res5(xr)
    feats = y.mean(dim=[2, 3])
    (feats * mg.seeded(tuple(feats.shape), 113)).sum().backward()
    np.savez_compressed(os.path.join(HERE, "ref_stock_resnet.npz"), res4=res4.numpy(), res5_mean=feats.detach().numpy(),
                        gx=xr.grad[:, ::16].numpy(), gw=dict(res5.named_parameters())["0.conv1.weight"].grad[::8, ::16, 0, 0].numpy(),
                        gw3=dict(res5.named_parameters())["2.conv2.weight"].grad[::16, ::16].numpy())
    print("stock resnet ok", tuple(res4.shape), float(res4.std()), tuple(feats.shape))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    which = sys.argv[1:] or ["sampling", "sgd", "stock", "w1", "w2"]
    if "stock" in which:
        stock_resnet_golden()
    if "sampling" in which:
        roi_sampling_golden()
    if "sgd" in which:
        sgd_golden()
    if "w1" in which:
        mp.spawn(run_rank, args=(1, _free_port(), True, os.path.join(HERE, "ref_step_w1_r{rank}.npz")), nprocs=1, join=True)
    if "w2" in which:
        mp.spawn(run_rank, args=(2, _free_port(), True, os.path.join(HERE, "ref_step_w2_r{rank}.npz")), nprocs=2, join=True)


if __name__ == "__main__":
    main()

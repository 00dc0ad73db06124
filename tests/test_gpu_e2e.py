"""GPU end-to-end parity: the HIP training step (f32 parity path) against the CPU oracle on identical inputs.
Tolerance from BASELINE.json north_star: every fp32 loss within 1e-3 relative; RoI/NMS/matcher stages are checked
bit-exactly per stage in test_gpu_ops.py (end-to-end, last-bit conv differences may legally flip a tie)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(dtype="f32", roi_batch=48, pre_nms=600, post_nms=200, kd=False):
    from cddmsl_amd.config import get_cfg
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "VOC-Experiments", "faster_rcnn_CLIP_R_50_C4.yaml"))
    cfg.merge_from_list(["MODEL.COMPUTE_DTYPE", dtype, "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", roi_batch,
                         "MODEL.RPN.PRE_NMS_TOPK_TRAIN", pre_nms, "MODEL.RPN.POST_NMS_TOPK_TRAIN", post_nms,
                         "MODEL.KD_REGULRAZIATION", kd])
    return cfg


def _build(cfg, seed):
    from cddmsl_amd import synthetic
    from cddmsl_amd.modeling import build_model, TransformerMapper
    sd = synthetic.make_state_dict(0)
    msd = synthetic.make_mapper_state_dict(1)
    model = build_model(cfg)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all("cell_anchors" in m or "pixel_" in m for m in missing), (missing, unexpected)
    mapper = TransformerMapper(compute_dtype=model.compute_dtype)
    mapper.load_state_dict(msd)
    mapper.to(model.device).eval()
    g = torch.Generator().manual_seed(seed)
    model.proposal_generator.sample_generator = g
    model.roi_heads.sample_generator = g
    model.region_generator = g
    model.train()
    return model, mapper, sd, msd


def _oracle_cfg(cfg, kd):
    from oracle import model as om
    return om.Cfg(roi_batch_per_image=cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE, rpn_pre_nms_topk=cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN,
                  rpn_post_nms_topk=cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN, kd_regularization=kd)


@pytest.mark.parametrize("kd,share,gemm256", [(False, True, "1"), (True, True, "1"), (True, False, "1"), (False, True, "2")])
def test_step_losses_and_grads_match_oracle(kd, share, gemm256, monkeypatch):
    """share=True: the region-level branch reuses the supervised pass's source-image res4 + RPN proposals (engine.py /
    rcnn.py notes); share=False: every branch recomputes, as the reference does.  Both must match the oracle, which
    always recomputes.  gemm256="2" forces every eligible layer onto the 256x256 ping-pong kernel (its f32 instantiation),
    which the size heuristic would not pick at this test's small shapes."""
    monkeypatch.setenv("CDDMSL_GEMM256", gemm256)
    from cddmsl_amd import synthetic
    from cddmsl_amd.engine import SimpleTrainer
    from cddmsl_amd.solver import build_optimizer
    from oracle import model as om
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    cfg = _cfg("f32", kd=kd)
    model, mapper, sd, msd = _build(cfg, seed=5)
    batch = synthetic.make_batch(2, 160, 224, num_gt=3)
    opt = build_optimizer(cfg, model)
    tr = SimpleTrainer(model, iter([batch]), opt, cfg, clipcap_model=mapper, metrics_period=0)
    tr.iter = 20000   # past burn-in: all three branches live
    tr.share_source_pass = share
    tr.fuse_consistency = share       # (share=True also runs both consistency branches through ONE mapper / projector pass)
    tr.buckets.zero()
    ld = tr.compute_losses(batch)
    sum(ld.values()).backward()
    got = {k: float(v) for k, v in ld.items()}

    ocfg = _oracle_cfg(cfg, kd)
    keys = om.trainable_keys(sd, ocfg)
    for k in keys:
        sd[k].requires_grad_(True)
    ref = om.run_step_losses(sd, msd, ocfg, batch, 20000, torch.Generator().manual_seed(5))
    sum(ref.values()).backward()
    want = {k: float(v) for k, v in ref.items()}
    assert set(got) == set(want)
    for k in want:
        assert abs(got[k] - want[k]) <= 1e-3 * abs(want[k]) + 1e-6, (k, got[k], want[k])
    # parameter gradients (f32): relative to each tensor's max
    params = dict(model.named_parameters())
    worst = 0.0
    for k in keys:
        g, r = params[k].grad.detach().float().cpu(), sd[k].grad
        err = float((g - r).abs().max() / max(float(r.abs().max()), 1e-5))  # k_proj.bias has an exactly-zero true gradient
        worst = max(worst, err)
        assert err < 5e-3, (k, err)
    print("losses", got, "worst grad rel err", worst)


def test_deferred_rpn_losses_give_the_same_step():
    """With one generator per sampler (build_trainer's setup) the RPN's host-side anchor sampling and its losses are
    finished after the box head has been enqueued (rcnn.py, rpn.py forward_nhwc defer_losses).  The draws of each stream
    are the same either way, so losses and gradients must not change (f32; only the f32-atomic summation order differs)."""
    from cddmsl_amd import synthetic
    from cddmsl_amd.engine import SimpleTrainer
    from cddmsl_amd.solver import build_optimizer
    cfg = _cfg("f32")
    batch = synthetic.make_batch(2, 160, 224, num_gt=3)
    res = []
    for defer in (True, False):
        model, mapper, _, _ = _build(cfg, seed=5)
        model.proposal_generator.sample_generator = torch.Generator().manual_seed(11)
        model.roi_heads.sample_generator = torch.Generator().manual_seed(12)
        model.region_generator = torch.Generator().manual_seed(13)
        model.defer_rpn_losses = defer
        tr = SimpleTrainer(model, iter([batch]), build_optimizer(cfg, model), cfg, clipcap_model=mapper, metrics_period=0)
        tr.iter = 20000
        tr.buckets.zero()
        ld = tr.compute_losses(batch)
        sum(ld.values()).backward()
        res.append(({k: float(v) for k, v in ld.items()},
                    {k: p.grad.detach().float().cpu() for k, p in model.named_parameters() if p.grad is not None}))
    (la, ga), (lb, gb) = res
    assert set(la) == set(lb) and set(ga) == set(gb)
    for k in la:
        assert abs(la[k] - lb[k]) <= 1e-6 * max(1.0, abs(lb[k])), (k, la[k], lb[k])
    for k in ga:
        assert float((ga[k] - gb[k]).abs().max()) <= 1e-4 * max(float(gb[k].abs().max()), 1e-5), k


@pytest.mark.parametrize("gemm256", ["1", "2"])
def test_bf16_step_runs_and_is_close(gemm256, monkeypatch):
    """Throughput path (bf16 MFMA, fp32 accumulate): finite losses, loosely near the f32 oracle values (bf16 has
    8 significant bits; index stages may pick different RoIs, so this is a sanity bound, not the parity gate).
    gemm256="2": the whole step on the 256x256 forward and weight-gradient kernels wherever they are legal."""
    monkeypatch.setenv("CDDMSL_GEMM256", gemm256)
    from cddmsl_amd import synthetic
    from cddmsl_amd.engine import SimpleTrainer
    from cddmsl_amd.solver import build_optimizer
    cfg = _cfg("bf16")
    model, mapper, sd, msd = _build(cfg, seed=5)
    batch = synthetic.make_batch(2, 160, 224, num_gt=3)
    tr = SimpleTrainer(model, iter([batch, batch]), build_optimizer(cfg, model), cfg, clipcap_model=mapper, metrics_period=1)
    tr.iter = 20000
    before = model.backbone.layer3[0].conv1.weight.detach().clone()
    ld = tr.run_step()
    vals = {k: float(v) for k, v in ld.items()}
    assert all(v == v and abs(v) < 1e4 for v in vals.values()), vals
    assert not torch.equal(before, model.backbone.layer3[0].conv1.weight.detach())
    tr.run_step()


def test_stock_r50_c4_config1_matches_oracle():
    """BASELINE.json configs[0]: configs/PascalVOC-Detection/faster_rcnn_R_50_C4.yaml, 2 synthetic images, supervised step
    (stock ResNet-50 backbone, Res5ROIHeads mean pool, linear classifier) -- HIP path (f32) vs the CPU oracle."""
    from cddmsl_amd import synthetic
    from cddmsl_amd.config import get_cfg
    from cddmsl_amd.modeling import build_model
    from oracle import model_r50 as r50
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "PascalVOC-Detection", "faster_rcnn_R_50_C4.yaml"))
    cfg.merge_from_list(["MODEL.COMPUTE_DTYPE", "f32", "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 48, "MODEL.RPN.PRE_NMS_TOPK_TRAIN", 600,
                         "MODEL.RPN.POST_NMS_TOPK_TRAIN", 200])
    assert cfg.MODEL.BACKBONE.NAME == "build_resnet_backbone" and cfg.MODEL.ROI_HEADS.NAME == "Res5ROIHeads"
    sd = synthetic.make_state_dict_r50(0)
    model = build_model(cfg)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.startswith(("offline_backbone.", "projector.")) for m in missing), (missing, unexpected)
    g = torch.Generator().manual_seed(9)
    model.proposal_generator.sample_generator = g
    model.roi_heads.sample_generator = g
    model.train()
    batch = synthetic.make_batch(2, 160, 224, num_gt=3)
    ld = model(batch)
    sum(ld.values()).backward()
    got = {k: float(v) for k, v in ld.items()}

    ocfg = r50.cfg_r50()
    ocfg.roi_batch_per_image, ocfg.rpn_pre_nms_topk, ocfg.rpn_post_nms_topk = 48, 600, 200
    keys = r50.trainable_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    ref = r50.forward(sd, ocfg, batch, torch.Generator().manual_seed(9))
    sum(ref.values()).backward()
    for k, v in ref.items():
        assert abs(got[k] - float(v)) <= 1e-3 * abs(float(v)) + 1e-6, (k, got[k], float(v))
    params = dict(model.named_parameters())
    worst = 0.0
    for k in keys:
        gk, rk = params[k].grad.detach().float().cpu(), sd[k].grad
        err = float((gk - rk).abs().max() / max(float(rk.abs().max()), 1e-5))
        worst = max(worst, err)
        assert err < 5e-3, (k, err)
    print("stock R50-C4 losses", got, "worst grad rel err", worst)


def test_inference_and_voc_map_match_oracle(tmp_path):
    """eval-mode forward (RPN test top-k, RoI head without sampling, softmax scores, per-class NMS,
    top-k, rescale to the sample's height/width) on the HIP path (f32) vs the CPU oracle, then VOC07 AP of both against
    the same synthetic ground truth: detections agree and mAP is within 1e-3 relative (BASELINE.json north_star)."""
    from cddmsl_amd import synthetic
    from cddmsl_amd import evaluation as ev
    from cddmsl_amd.structures import Boxes, Instances
    from oracle import model as om
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    cfg = _cfg("f32")
    cfg.merge_from_list(["MODEL.RPN.PRE_NMS_TOPK_TEST", 400, "MODEL.RPN.POST_NMS_TOPK_TEST", 60, "TEST.DETECTIONS_PER_IMAGE", 30])
    model, _, sd, _ = _build(cfg, seed=3)
    model.eval()
    batch = synthetic.make_batch(3, 160, 224, num_gt=3)
    for i, x in enumerate(batch):                      # dataset-dict fields the postprocess / evaluator read
        x["height"], x["width"], x["image_id"] = 320, 448, f"im{i}"
    got = model(batch)
    ocfg = om.Cfg(rpn_pre_nms_topk_test=400, rpn_post_nms_topk_test=60, detections_per_image=30)
    want = om.inference(sd, ocfg, batch)
    assert sum(len(w["scores"]) for w in want) > 10, "degenerate case: nothing detected"
    for g, w in zip(got, want):
        inst = g["instances"]
        assert inst.image_size == (320, 448)
        assert len(inst) == len(w["scores"]), (len(inst), len(w["scores"]))
        assert torch.equal(inst.pred_classes.cpu(), w["classes"])
        assert (inst.scores.cpu() - w["scores"]).abs().max() <= 1e-3 * max(float(w["scores"].abs().max()), 1e-6)
        assert (inst.pred_boxes.tensor.cpu() - w["boxes"]).abs().max() <= 1e-2        # pixels at 448x320
    # mAP parity on a synthetic annotation set: the samples' GT boxes (scaled to the output resolution) as "aeroplane".. classes
    names = list(ev.VOC_CLASS_NAMES)
    os.makedirs(tmp_path / "Annotations"), os.makedirs(tmp_path / "ImageSets" / "Main")
    for x in batch:
        objs = []
        for b, c in zip(x["instances"]["gt_boxes"].tolist(), x["instances"]["gt_classes"].tolist()):
            objs.append("<object><name>%s</name><difficult>0</difficult><bndbox><xmin>%d</xmin><ymin>%d</ymin><xmax>%d</xmax><ymax>%d</ymax></bndbox></object>"
                        % (names[c], int(2 * b[0]) + 1, int(2 * b[1]) + 1, int(2 * b[2]), int(2 * b[3])))
        (tmp_path / "Annotations" / (x["image_id"] + ".xml")).write_text("<annotation>" + "".join(objs) + "</annotation>")
    (tmp_path / "ImageSets" / "Main" / "test.txt").write_text("\n".join(x["image_id"] for x in batch) + "\n")
    res = []
    for outs in (got, [{"instances": Instances((320, 448), pred_boxes=Boxes(w["boxes"]), scores=w["scores"], pred_classes=w["classes"])} for w in want]):
        e = ev.PascalVOCDetectionEvaluator(str(tmp_path), "test", 2007)
        e.process(batch, outs)
        res.append(e.evaluate()["bbox"])
    for k in ("AP", "AP50", "AP75"):
        assert abs(res[0][k] - res[1][k]) <= 1e-3 * max(abs(res[1][k]), 1e-9), (k, res[0][k], res[1][k])


def test_ragged_batch_step_matches_oracle():
    """Images of DIFFERENT sizes in one batch (zero-padded to the batch maximum, image_list.py:72-124; proposals clipped to
    each image's own size; the 224-crop branch pads per pair) -- all three branches, losses vs the oracle on the same inputs."""
    from cddmsl_amd import synthetic
    from cddmsl_amd.engine import SimpleTrainer
    from cddmsl_amd.solver import build_optimizer
    from oracle import model as om
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    cfg = _cfg("f32")
    model, mapper, sd, msd = _build(cfg, seed=8)
    batch = synthetic.make_batch(1, 160, 224, num_gt=3) + synthetic.make_batch(1, 128, 192, num_gt=2, iteration=1)
    tr = SimpleTrainer(model, iter([batch]), build_optimizer(cfg, model), cfg, clipcap_model=mapper, metrics_period=0)
    tr.iter = 20000
    tr.buckets.zero()
    ld = tr.compute_losses(batch)
    sum(ld.values()).backward()
    got = {k: float(v.detach()) for k, v in ld.items()}
    ocfg = _oracle_cfg(cfg, False)
    keys = om.trainable_keys(sd, ocfg)
    for k in keys:
        sd[k].requires_grad_(True)
    ref = om.run_step_losses(sd, msd, ocfg, batch, 20000, torch.Generator().manual_seed(8))
    sum(ref.values()).backward()
    for k, v in ref.items():
        assert abs(got[k] - float(v)) <= 1e-3 * abs(float(v)) + 1e-6, (k, got[k], float(v))
    params = dict(model.named_parameters())
    for k in keys:
        g, r = params[k].grad.detach().float().cpu(), sd[k].grad
        assert float((g - r).abs().max() / max(float(r.abs().max()), 1e-5)) < 5e-3, k


def test_image_without_ground_truth_matches_oracle():
    """One sample carries no ground-truth boxes (matcher.py:78-90: everything is background / ignored; roi_heads.py:262-277: all
    sampled proposals are background, no gt_boxes field): supervised losses vs the oracle."""
    from cddmsl_amd import synthetic
    from oracle import model as om
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    cfg = _cfg("f32")
    model, _, sd, _ = _build(cfg, seed=4)
    batch = synthetic.make_batch(2, 160, 224, num_gt=3)
    batch[1]["instances"] = {"gt_boxes": torch.zeros(0, 4), "gt_classes": torch.zeros(0, dtype=torch.int64), "image_size": (160, 224)}
    ld = model(batch)
    ref = om.forward(sd, _oracle_cfg(cfg, False), batch, gen=torch.Generator().manual_seed(4))
    for k, v in ref.items():
        assert abs(float(ld[k].detach()) - float(v)) <= 1e-3 * abs(float(v)) + 1e-6, (k, float(ld[k].detach()), float(v))

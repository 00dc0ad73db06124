"""CPU: pins the oracle (oracle/) against (a) the known-answer tables the reference's own tests hold
(tests/golden/kat.json) and (b) golden vectors produced by the reference's own leaf modules
(tests/golden/ref_*.npz, generator: tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from cddmsl_amd import synthetic
from oracle import model as om
from oracle import ops as oo

G = os.path.join(os.path.dirname(__file__), "golden")
KAT = json.load(open(os.path.join(G, "kat.json")))
T = torch.tensor


def _npz(name):
    return np.load(os.path.join(G, name))


def seeded(shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.fixture(scope="module")
def sd():
    return synthetic.make_state_dict(0)


# ------------------------------------------------------------------ known-answer tables
@pytest.mark.parametrize("aligned", [False, True])
def test_kat_roi_align(aligned):
    k = KAT["roi_align_5x5"]
    x = torch.arange(25, dtype=torch.float32).reshape(1, 1, 5, 5)
    rois = T([[0.0] + [float(v) for v in k["box"]]])
    out = oo.roi_align(x, rois, tuple(k["output_size"]), 1.0, 0, aligned)
    exp = T(k["aligned_true" if aligned else "aligned_false"])
    assert torch.allclose(out[0, 0], exp)


def test_kat_roi_align_empty():
    # tests/layers/test_roi_align.py:111-128: empty box -> zeros, zero grad; empty batch -> shape (0,C,h,w)
    x = torch.rand(1, 3, 5, 5, requires_grad=True)
    out = oo.roi_align(x, T([[0.0, 3.0, 3.0, 3.0, 3.0]]), 7, 1.0, 0, True)
    assert out.shape == (1, 3, 7, 7) and (out == 0).all()
    out.sum().backward()
    assert (x.grad == 0).all()
    assert oo.roi_align(x, torch.zeros(0, 5), 7, 1.0, 0, True).shape == (0, 3, 7, 7)


def test_kat_matcher():
    k = KAT["matcher"]
    m, l = oo.matcher(T(k["quality"]), k["thresholds"], k["labels"], k["allow_low_quality"])
    assert m.tolist() == k["matches"] and l.tolist() == k["match_labels"] and l.dtype == torch.int8


def test_kat_pairwise_iou():
    k = KAT["pairwise_iou"]
    iou = oo.pairwise_iou(T(k["boxes1"]), T(k["boxes2"]))
    assert torch.allclose(iou, T([k["iou_row"], k["iou_row"]]))


def test_kat_anchors():
    k = KAT["anchors"]
    a = oo.grid_anchors(k["grid"][0], k["grid"][1], k["stride"], k["offset"], k["sizes"], k["ratios"])
    assert torch.allclose(a, T(k["expected"]))


def test_kat_scheduler():
    k = KAT["scheduler"]
    cfg = om.Cfg(base_lr=k["base_lr"], steps=tuple(k["steps"]), gamma=k["gamma"], warmup_factor=k["warmup_factor"],
                 warmup_iters=k["warmup_iters"], max_iter=k["max_iter"])
    lrs = [om.lr_at(cfg, i) for i in range(31)]
    assert np.allclose(lrs[:5], k["lrs_0_5"])
    assert np.allclose(lrs[5:10], k["lr_5_10"]) and np.allclose(lrs[10:15], k["lr_10_15"])
    assert np.allclose(lrs[15:20], k["lr_15_20"]) and np.allclose(lrs[20:], k["lr_20_30"])


# ------------------------------------------------------------------ reference leaf-module goldens
def test_ref_backbone(sd):
    g = _npz("ref_backbone.npz")
    cfg = om.Cfg()
    with torch.no_grad():
        o = om.backbone(sd, cfg, seeded((2, 3, 64, 96), 11))
        assert np.allclose(o["res4"].numpy(), g["res4_64x96"], rtol=1e-4, atol=1e-5)
        assert np.allclose(o["res5"].numpy(), g["res5_64x96"], rtol=1e-4, atol=1e-5)
        o2 = om.backbone(sd, cfg, seeded((1, 3, 224, 224), 12))
        assert np.allclose(o2["res5"][0, ::16].numpy(), g["res5_224_slice"], rtol=1e-4, atol=1e-5)
        assert np.allclose(om.attnpool(sd, cfg, o2["res5"]).numpy(), g["attnpool_224"], rtol=1e-4, atol=1e-5)
        assert np.allclose(om.attnpool(sd, cfg, seeded((4, 2048, 7, 7), 13)).numpy(), g["attnpool_rand4"], rtol=1e-4, atol=1e-5)
        assert np.allclose(om.layer4(sd, cfg, seeded((3, 1024, 14, 14), 14))[:, ::8].numpy(), g["layer4_14"], rtol=1e-4, atol=1e-5)


def test_ref_attnpool_grad(sd):
    g = _npz("ref_attnpool_grad.npz")
    cfg = om.Cfg()
    sd = dict(sd)
    keys = ["backbone.attnpool.q_proj.weight", "backbone.attnpool.c_proj.weight", "backbone.attnpool.positional_embedding"]
    for k in keys:
        sd[k] = sd[k].clone().requires_grad_(True)
    x = seeded((4, 2048, 7, 7), 13).requires_grad_(True)
    y = om.attnpool(sd, cfg, x)
    (y * seeded(tuple(y.shape), 15)).sum().backward()
    assert np.allclose(x.grad[:, ::32].numpy(), g["gx"], rtol=1e-4, atol=1e-6)
    assert np.allclose(sd[keys[0]].grad[::16, ::16].numpy(), g["gq"], rtol=1e-4, atol=1e-6)
    assert np.allclose(sd[keys[1]].grad[::16, ::16].numpy(), g["gc"], rtol=1e-4, atol=1e-6)
    assert np.allclose(sd[keys[2]].grad[:, ::16].numpy(), g["gpos"], rtol=1e-4, atol=1e-6)


def test_ref_mapper():
    g = _npz("ref_mapper.npz")
    msd = synthetic.make_mapper_state_dict(1)
    x = seeded((4, 1024), 21).requires_grad_(True)
    e = om.v2l(msd, om.Cfg(), x)
    (e * seeded(tuple(e.shape), 22)).sum().backward()
    assert np.allclose(e.detach().numpy(), g["v2l"], rtol=1e-4, atol=1e-5)
    assert np.allclose(x.grad.numpy(), g["gx"], rtol=1e-4, atol=1e-6)


def test_ref_boxes():
    g = _npz("ref_boxes.npz")
    gt, pr = T(g["gt"]), T(g["pr"])
    iou = oo.pairwise_iou(gt, pr)
    assert np.array_equal(iou.numpy(), g["iou"])
    a1, l1 = oo.matcher(iou, [0.3, 0.7], [0, -1, 1], True)
    a2, l2 = oo.matcher(iou, [0.5], [0, 1], False)
    assert np.array_equal(a1.numpy(), g["match_rpn"]) and np.array_equal(l1.numpy(), g["label_rpn"])
    assert np.array_equal(a2.numpy(), g["match_roi"]) and np.array_equal(l2.numpy(), g["label_roi"])
    pos, neg = oo.subsample_labels(l1, 64, 0.5, 0, torch.Generator().manual_seed(33))
    assert np.array_equal(pos.numpy(), g["sub_pos"]) and np.array_equal(neg.numpy(), g["sub_neg"])
    w = (10.0, 10.0, 5.0, 5.0)
    assert np.allclose(oo.get_deltas(pr[:50], gt[a2[:50]], w).numpy(), g["deltas"], rtol=1e-6, atol=1e-6)
    assert np.allclose(oo.apply_deltas(T(g["dd"]), pr[:50], w).numpy(), g["applied"], rtol=1e-6, atol=1e-4)
    assert np.array_equal(oo.grid_anchors(3, 5, 16).numpy(), g["anchors_3x5"])


def test_ref_fastrcnn(sd):
    g = _npz("ref_fastrcnn.npz")
    cfg = om.Cfg()
    sd = dict(sd)
    sd["roi_heads.box_predictor.bbox_pred.bias"] = T(g["bbox_bias"])
    feats = T(g["feats"]).requires_grad_(True)
    scores, deltas = om.box_predictor(sd, cfg, feats)
    assert np.allclose(scores.detach().numpy(), g["scores"], rtol=1e-4, atol=1e-4)
    assert np.allclose(deltas.detach().numpy(), g["deltas"], rtol=1e-4, atol=1e-5)
    gcls = T(g["gcls"])
    lc = om.focal_loss(cfg, scores, gcls)
    lb = om.box_reg_loss(cfg, T(g["pbox"]), T(g["gbox"]), deltas, gcls)
    assert np.allclose(float(lc), float(g["loss_cls"]), rtol=1e-5)
    assert np.allclose(float(lb), float(g["loss_box_reg"]), rtol=1e-5)
    (lc + lb).backward()
    assert np.allclose(feats.grad.numpy(), g["gfeats"], rtol=1e-3, atol=1e-6)
    st = om.classification_stats(scores.detach(), gcls)
    assert np.allclose([st["fast_rcnn/cls_accuracy"], st["fast_rcnn/fg_cls_accuracy"], st["fast_rcnn/false_negative"]], g["stats"])


def test_ref_rpn(sd):
    g = _npz("ref_rpn.npz")
    cfg = om.Cfg()
    feat = T(g["feat"]).requires_grad_(True)
    props, losses = om.rpn_forward(sd, cfg, feat, [(96, 144), (90, 130)], [T(g["gt0"]), T(g["gt1"])],
                                   torch.Generator().manual_seed(55), True)
    assert np.allclose(float(losses["loss_rpn_cls"]), float(g["loss_rpn_cls"]), rtol=1e-5)
    assert np.allclose(float(losses["loss_rpn_loc"]), float(g["loss_rpn_loc"]), rtol=1e-5)
    (losses["loss_rpn_cls"] + losses["loss_rpn_loc"]).backward()
    assert np.allclose(feat.grad.numpy(), g["gfeat"], rtol=1e-3, atol=1e-7)
    for i in range(2):
        assert props[i][0].shape == g[f"boxes{i}"].shape  # same NMS keep count (independent python NMS)
        assert np.allclose(props[i][0].numpy(), g[f"boxes{i}"], rtol=1e-5, atol=1e-4)
        assert np.allclose(props[i][1].numpy(), g[f"logits{i}"], rtol=1e-5, atol=1e-6)


def test_ref_roialign():
    g = _npz("ref_roialign.npz")
    x = T(g["x"]).requires_grad_(True)
    out = oo.roi_align(x, T(g["rois"]), 4, 1.0 / 16, 0, True)
    assert np.allclose(out.detach().numpy(), g["out"], rtol=1e-5, atol=1e-6)
    (out * T(g["w"])).sum().backward()
    assert np.allclose(x.grad.numpy(), g["gx"], rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------------ inference / evaluation rows
def _write_voc(root, data, split="test"):
    os.makedirs(os.path.join(root, "Annotations"), exist_ok=True)
    os.makedirs(os.path.join(root, "ImageSets", "Main"), exist_ok=True)
    for name, objs in data.items():
        xml = ["<annotation>"]
        for o in objs:
            xml.append("<object><name>%s</name><pose>Unspecified</pose><truncated>0</truncated><difficult>%d</difficult>"
                       "<bndbox><xmin>%d</xmin><ymin>%d</ymin><xmax>%d</xmax><ymax>%d</ymax></bndbox></object>"
                       % (o["name"], o["difficult"], *o["bbox"]))
        xml.append("</annotation>")
        with open(os.path.join(root, "Annotations", name + ".xml"), "w") as f:
            f.write("".join(xml))
    with open(os.path.join(root, "ImageSets", "Main", split + ".txt"), "w") as f:
        f.write("\n".join(data.keys()) + "\n")


def test_voc_ap_matches_reference_voc_eval(tmp_path):
    """cddmsl_amd.evaluation (the product's CPU evaluator) vs AP values produced by the reference's own voc_eval
    (tests/golden/make_golden_eval.py): VOC07 11-point and area AP, IoU .50 / .75, incl. difficult boxes, duplicates, score ties."""
    import json
    import numpy as np
    from cddmsl_amd import evaluation as ev
    fx = json.load(open(os.path.join(G, "ref_voc_eval.json")))
    _write_voc(str(tmp_path), fx["data"])
    names = list(fx["data"].keys())
    recs = {n: ev.read_voc_objects(os.path.join(str(tmp_path), "Annotations", n + ".xml")) for n in names}
    for c in fx["classes"]:
        gt = {n: (np.array([o["bbox"] for o in recs[n] if o["name"] == c], dtype=float).reshape(-1, 4),
                  np.array([o["difficult"] for o in recs[n] if o["name"] == c], dtype=bool)) for n in names}
        lines = [l.split(" ") for l in fx["dets"][c]]
        ids = [l[0] for l in lines]
        conf = np.array([float(l[1]) for l in lines])
        bbs = np.array([[float(z) for z in l[2:]] for l in lines]).reshape(-1, 4)
        for thr in (0.5, 0.75):
            for m07 in (True, False):
                got = ev.class_ap(ids, conf, bbs, gt, thr, m07)
                assert abs(got - fx["ap"][f"{c}|{thr}|{int(m07)}"]) < 1e-12, (c, thr, m07)
    # the evaluator object end to end (process -> evaluate): AP50 of the 2007 metric = mean over classes of the values above
    from cddmsl_amd.structures import Boxes, Instances
    e = ev.PascalVOCDetectionEvaluator(str(tmp_path), "test", 2007, class_names=fx["classes"])
    for n in names:
        rows = [(ci, l.split(" ")) for ci, c in enumerate(fx["classes"]) for l in fx["dets"][c] if l.split(" ")[0] == n]
        inst = Instances((1, 1))
        # undo the writer's +1 on xmin / ymin so that process() reproduces the stored line
        inst.pred_boxes = Boxes(torch.tensor([[float(l[2]) - 1, float(l[3]) - 1, float(l[4]), float(l[5])] for _, l in rows], dtype=torch.float64).reshape(-1, 4).float())
        inst.scores = torch.tensor([float(l[1]) for _, l in rows])
        inst.pred_classes = torch.tensor([ci for ci, _ in rows], dtype=torch.int64)
        e.process([{"image_id": n}], [{"instances": inst}])
    out = e.evaluate()["bbox"]
    want50 = 100 * np.mean([fx["ap"][f"{c}|0.5|1"] for c in fx["classes"]])
    assert abs(out["AP50"] - want50) < 1e-6 * max(want50, 1.0), (out["AP50"], want50)


def test_oracle_inference_matches_reference():
    """oracle fast_rcnn_inference_single_image / detector_postprocess vs the reference's own functions (ref_inference.npz)."""
    from oracle import model as om
    fx = np.load(os.path.join(G, "ref_inference.npz"))
    b, s, c, kept = om.fast_rcnn_inference_single_image(torch.from_numpy(fx["boxes"]), torch.from_numpy(fx["scores"]), (200, 300), 0.05, 0.5, 20)
    assert torch.equal(c, torch.from_numpy(fx["det_classes"])) and torch.equal(kept, torch.from_numpy(fx["det_kept"]))
    assert torch.equal(b, torch.from_numpy(fx["det_boxes"])) and torch.equal(s, torch.from_numpy(fx["det_scores"]))
    pb, ps, pc = om.detector_postprocess(b, s, c, (200, 300), 333, 480)
    assert torch.allclose(pb, torch.from_numpy(fx["post_boxes"]), rtol=0, atol=1e-4) and torch.equal(pc, torch.from_numpy(fx["post_classes"]))
    assert torch.equal(ps, torch.from_numpy(fx["post_scores"]))


# ------------------------------------------------------------------------------------------------ step-level goldens
# tests/golden/make_golden_step.py: the reference's OWN GeneralizedRCNN.forward (three branches, composed as SimpleTrainer.run_step
# does), label_and_sample_proposals and clipping-SGD wrapper -- SURVEY.md 8(c) items (9), (10).
STEP = dict(H=96, W=128, per_rank=2, roi_batch=16, pre=200, post=60, seed=77)


def _step_cfg():
    return om.Cfg(roi_batch_per_image=STEP["roi_batch"], rpn_pre_nms_topk=STEP["pre"], rpn_post_nms_topk=STEP["post"], kd_regularization=True)


def _step_state():
    sd = synthetic.drift_offline(synthetic.make_state_dict(0))
    return sd, synthetic.make_mapper_state_dict(1)


def _oracle_rank(rank, others=None, grad=True):
    sd, msd = _step_state()
    cfg = _step_cfg()
    keys = om.trainable_keys(sd, cfg)
    if grad:
        for k in keys:
            sd[k].requires_grad_(True)
    batch = synthetic.make_batch(STEP["per_rank"], STEP["H"], STEP["W"], rank=rank, num_gt=3)
    record = {}
    with torch.set_grad_enabled(grad):
        ld = om.run_step_losses(sd, msd, cfg, batch, 20000, torch.Generator().manual_seed(STEP["seed"] + rank), record=record,
                                others=others, rank=rank)
        if grad:
            sum(ld.values()).backward()
    return ld, sd, keys, record


def _check_step(g, ld, sd, keys, record, loss_rtol=2e-4):
    for k, v in ld.items():
        want = float(g["loss/" + k])
        assert abs(float(v) - want) <= loss_rtol * abs(want) + 1e-7, (k, float(v), want)
    assert {k[5:] for k in g.files if k.startswith("loss/")} == set(ld)
    # index stages, bit for bit: proposals kept by NMS, sampled RoI classes, the 16 region picks per image
    for i in range(STEP["per_rank"]):
        b, s = record["proposals"][i]
        assert b.shape == g[f"prop_boxes{i}"].shape, (i, b.shape, g[f"prop_boxes{i}"].shape)
        assert np.allclose(b.numpy(), g[f"prop_boxes{i}"], rtol=1e-5, atol=1e-3) and np.allclose(s.numpy(), g[f"prop_logits{i}"], rtol=1e-5, atol=1e-5)
        assert np.array_equal(record["roi_gt_classes"][i].numpy(), g[f"sampled_classes{i}"])
        assert np.allclose(record["region_boxes"][i].numpy(), g[f"region_boxes{i}"], rtol=1e-5, atol=1e-3)
    # gradients: every trainable tensor's norm and max, and slices of ten of them
    names = [str(n) for n in g["grad_names"]]
    assert sorted(keys) == names
    for n, nrm, mx in zip(names, g["grad_norms"], g["grad_absmax"]):
        gr = sd[n].grad
        if mx < 1e-8:                                   # attnpool.k_proj.bias: exactly zero in exact arithmetic (softmax shift invariance)
            assert float(gr.abs().max()) < 1e-7, n
            continue
        assert abs(float(gr.double().norm()) - nrm) <= 2e-3 * nrm + 1e-9, (n, float(gr.norm()), nrm)
        assert abs(float(gr.abs().max()) - mx) <= 5e-3 * mx + 1e-9, (n, float(gr.abs().max()), mx)
    sl = {"backbone.layer2.0.conv1.weight": np.s_[::4, ::8], "backbone.layer3.5.conv2.weight": np.s_[::16, ::16],
          "backbone.layer4.0.downsample.0.weight": np.s_[::64, ::32], "backbone.attnpool.k_proj.weight": np.s_[::64, ::64],
          "backbone.attnpool.positional_embedding": np.s_[:, ::64], "proposal_generator.rpn_head.conv.weight": np.s_[::64, ::64],
          "proposal_generator.rpn_head.anchor_deltas.bias": np.s_[:], "roi_heads.box_predictor.bbox_pred.weight": np.s_[::4, ::32],
          "projector.0.weight": np.s_[::24, ::24], "projector.2.bias": np.s_[:]}
    for n, ix in sl.items():
        want = g["grad/" + n]
        got = sd[n].grad.numpy()[ix]
        assert np.abs(got - want).max() <= 2e-3 * np.abs(want).max() + 1e-9, n


def test_ref_run_step_world1():
    """the oracle's run_step_losses == the reference's three forwards + backward (world size 1, KD on): 7 losses, the index
    stages, all 48.4 M gradients (norms / maxima of every tensor + slices)"""
    g = _npz("ref_step_w1_r0.npz")
    assert list(g["meta"][:3]) == [1, 0, 1]
    ld, sd, keys, record = _oracle_rank(0)
    assert float(ld["kd_loss"]) > 1e-3
    _check_step(g, ld, sd, keys, record)


@pytest.mark.parametrize("rank", [0, 1])
def test_ref_run_step_world2(rank):
    """world size 2 (two real gloo ranks in the generator): the cross-rank contrastive batch (GatherLayer forward = all ranks'
    embeddings, backward = own slice) against the oracle's simulated ranks -- the other rank's embeddings come from the oracle's
    own no-grad run of that rank's batch."""
    g = _npz(f"ref_step_w2_r{rank}.npz")
    assert list(g["meta"][:3]) == [2, rank, 1]
    _, _, _, rec_o = _oracle_rank(1 - rank, grad=False)
    others = {"img": ([rec_o["img_emb_tgt"]], [rec_o["img_emb_src"]]), "reg": ([rec_o["reg_emb_src"]], [rec_o["reg_emb_tgt"]])}
    ld, sd, keys, record = _oracle_rank(rank, others=others)
    _check_step(g, ld, sd, keys, record)
    # both ranks see the same full-matrix loss
    g_other = _npz(f"ref_step_w2_r{1 - rank}.npz")
    assert abs(float(g["loss/cont_loss"]) - float(g_other["loss/cont_loss"])) < 1e-6


def test_ref_roi_sampling():
    """oracle label_and_sample_proposals vs the reference's ROIHeads.label_and_sample_proposals (roi_heads.py:236-319): sampled
    boxes, logits, classes and matched GT boxes per image, incl. an image without ground truth and one with few candidates"""
    g = _npz("ref_roi_sampling.npz")
    cfg = om.Cfg(roi_batch_per_image=64)
    props = [(T(g[f"boxes{i}"]), T(g[f"logits{i}"])) for i in range(3)]
    gtb = [T(g[f"gt_boxes{i}"]).reshape(-1, 4) for i in range(3)]
    gtc = [T(g[f"gt_classes{i}"]).long() for i in range(3)]
    out = om.label_and_sample_proposals(cfg, props, gtb, gtc, torch.Generator().manual_seed(92))
    nfg = nbg = 0
    for i, o in enumerate(out):
        assert np.array_equal(o["proposal_boxes"].numpy(), g[f"s_boxes{i}"]) and np.array_equal(o["gt_classes"].numpy(), g[f"s_classes{i}"])
        assert np.allclose(o["objectness_logits"].numpy(), g[f"s_logits{i}"])
        assert ("gt_boxes" in o) == (f"s_gt_boxes{i}" in g.files)
        if "gt_boxes" in o:
            assert np.array_equal(o["gt_boxes"].numpy(), g[f"s_gt_boxes{i}"])
        nfg += int((o["gt_classes"] != 20).sum())
        nbg += int((o["gt_classes"] == 20).sum())
    assert np.allclose([nfg / 3, nbg / 3], g["num_fg_bg"])


def test_ref_sgd():
    """oracle sgd_step vs the reference's clipping-SGD class (solver/build.py:43-110 around torch.optim.SGD): three steps"""
    g = _npz("ref_sgd.npz")
    names = ["a.weight", "b.weight", "b.bias", "c.weight"]
    sd = {k: T(g["w0/" + k]) for k in names}
    mom = {}
    for step, lr in enumerate(g["lrs"]):
        cfg = om.Cfg(base_lr=float(lr), warmup_iters=0, steps=(), clip_value=5.0)
        assert abs(om.sgd_step(sd, {k: T(g[f"g{step}/" + k]) for k in names}, mom, cfg, 10) - float(lr)) < 1e-12
        for k in names:
            assert np.allclose(sd[k].numpy(), g[f"w{step + 1}/" + k], rtol=1e-6, atol=1e-7), (step, k)


def test_ref_stock_resnet():
    """oracle/model_r50.py (config #1) vs the reference's own stock ResNet (backbone/resnet.py) and res5 stage: res4 on a seeded
    image, the RoI head's res5 + mean pool with input / weight gradients"""
    from oracle import model_r50 as r50
    g = _npz("ref_stock_resnet.npz")
    sd = synthetic.make_state_dict_r50(0)
    with torch.no_grad():
        res4 = r50.backbone(sd, seeded((2, 3, 64, 96), 111, 50.0))
    assert np.allclose(res4.numpy(), g["res4"], rtol=1e-4, atol=1e-5)
    for k in ("roi_heads.res5.0.conv1.weight", "roi_heads.res5.2.conv2.weight"):
        sd[k] = sd[k].clone().requires_grad_(True)
    x = seeded((3, 1024, 14, 14), 112).requires_grad_(True)
    feats = r50.stage(sd, "roi_heads.res5", x, 3, 2).mean(dim=[2, 3])
    assert np.allclose(feats.detach().numpy(), g["res5_mean"], rtol=1e-4, atol=1e-5)
    (feats * seeded(tuple(feats.shape), 113)).sum().backward()
    assert np.allclose(x.grad[:, ::16].numpy(), g["gx"], rtol=1e-3, atol=1e-7)
    assert np.allclose(sd["roi_heads.res5.0.conv1.weight"].grad[::8, ::16, 0, 0].numpy(), g["gw"], rtol=1e-3, atol=1e-6)
    assert np.allclose(sd["roi_heads.res5.2.conv2.weight"].grad[::16, ::16].numpy(), g["gw3"], rtol=1e-3, atol=1e-6)

"""ORACLE (test infrastructure, CPU only): fp32 ATen/CPU restatement of the CDDMSL training step.

Functional style: every function takes ``sd`` = a state dict with the *reference's* key names
(checkpoint row) and plain tensors; nothing here is imported by the product.
Paths cited are relative to /root/reference/detectron2.

Deliberate, documented deviations from the reference *as executed* (results identical):
  * world_size 1 does not need a DDP wrapper / process group (reference crashes; SURVEY warnings 3);
  * all random draws come from one replayable CPU ``torch.Generator`` (see ``ops.subsample_labels``);
  * ``torch.sort(descending=True)`` ties are broken lower-index-first (stable) -- the reference leaves
    them unspecified (proposal_utils.py:77).
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import ops


@dataclass
class Cfg:
    """Resolved hot-path configuration."""
    pixel_mean: Tuple[float, ...] = (0.48145466, 0.4578275, 0.40821073)
    pixel_std: Tuple[float, ...] = (0.26862954, 0.26130258, 0.27577711)
    layers: Tuple[int, ...] = (3, 4, 6, 3)
    width: int = 64
    heads: int = 32
    embed_dim: int = 1024
    anchor_sizes: Tuple[float, ...] = (32, 64, 128, 256, 512)
    anchor_ratios: Tuple[float, ...] = (0.5, 1.0, 2.0)
    anchor_offset: float = 0.0
    feat_stride: int = 16
    rpn_iou_thresholds: Tuple[float, ...] = (0.3, 0.7)
    rpn_iou_labels: Tuple[int, ...] = (0, -1, 1)
    rpn_batch_per_image: int = 256
    rpn_positive_fraction: float = 0.5
    rpn_bbox_weights: Tuple[float, ...] = (1.0, 1.0, 1.0, 1.0)
    rpn_smooth_l1_beta: float = 0.0
    rpn_pre_nms_topk: int = 12000
    rpn_post_nms_topk: int = 2000
    rpn_pre_nms_topk_test: int = 6000     # config/defaults.py:344-355
    rpn_post_nms_topk_test: int = 1000    # configs/Base-RCNN-C4.yaml:4-5
    test_score_thresh: float = 0.05       # MODEL.ROI_HEADS.SCORE_THRESH_TEST
    test_nms_thresh: float = 0.5          # MODEL.ROI_HEADS.NMS_THRESH_TEST
    detections_per_image: int = 100       # TEST.DETECTIONS_PER_IMAGE
    multiply_rpn_score: bool = False      # MODEL.CLIP.MULTIPLY_RPN_SCORE
    rpn_nms_thresh: float = 0.7
    rpn_min_box_size: float = 0.0
    num_classes: int = 20
    roi_iou_thresholds: Tuple[float, ...] = (0.5,)
    roi_iou_labels: Tuple[int, ...] = (0, 1)
    roi_batch_per_image: int = 512
    roi_positive_fraction: float = 0.25
    roi_append_gt: bool = True
    pooler_resolution: int = 14
    pooler_sampling_ratio: int = 0
    roi_bbox_weights: Tuple[float, ...] = (10.0, 10.0, 5.0, 5.0)
    roi_smooth_l1_beta: float = 0.0
    cls_temp: float = 0.01
    bg_cls_loss_weight: float = 0.2
    focal_gamma: float = 0.5
    regions_per_image: int = 16           # meta_arch/rcnn.py:437
    prefix_length: int = 40               # engine/train_loop.py:281
    mapper_layers: int = 8
    mapper_heads: int = 8
    mapper_dim: int = 768
    kd_regularization: bool = False       # MODEL.KD_REGULRAZIATION (VOC yaml :4)
    burn_in_iters: int = 10000            # engine/train_loop.py:334
    # solver (configs/VOC-Experiments/faster_rcnn_CLIP_R_50_C4.yaml:40-49, config/defaults.py:634-692)
    base_lr: float = 0.002
    momentum: float = 0.9
    weight_decay: float = 1e-4
    weight_decay_norm: float = 0.0
    clip_value: float = 5.0
    warmup_iters: int = 100
    warmup_factor: float = 1e-3
    steps: Tuple[int, ...] = (10000, 18000, 25000, 30000, 35000, 39000, 49000)
    gamma: float = 0.1
    max_iter: int = 90000


# ------------------------------------------------------------------------------------------------
# backbone  (modeling/backbone/clip_backbone.py)
# ------------------------------------------------------------------------------------------------
def frozen_bn(sd, p, x, eps=1e-5):
    """layers/batch_norm.py:45-66 (the grad-path form; numerically the same affine)."""
    scale = sd[p + ".weight"] * (sd[p + ".running_var"] + eps).rsqrt()
    bias = sd[p + ".bias"] - sd[p + ".running_mean"] * scale
    return x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)


def bottleneck(sd, p, x, stride):
    """clip_backbone.py:57-70."""
    out = F.relu(frozen_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"])))
    out = F.relu(frozen_bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], padding=1)))
    if stride > 1:
        out = F.avg_pool2d(out, stride)
    out = frozen_bn(sd, p + ".bn3", F.conv2d(out, sd[p + ".conv3.weight"]))
    if (p + ".downsample.0.weight") in sd:
        idn = F.avg_pool2d(x, stride) if stride > 1 else x
        idn = frozen_bn(sd, p + ".downsample.1", F.conv2d(idn, sd[p + ".downsample.0.weight"]))
    else:
        idn = x
    return F.relu(out + idn)


def res_layer(sd, p, x, nblocks, stride):
    for i in range(nblocks):
        x = bottleneck(sd, f"{p}.{i}", x, stride if i == 0 else 1)
    return x


def stem(sd, p, x):
    """clip_backbone.py:194-198."""
    x = F.relu(frozen_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], stride=2, padding=1)))
    x = F.relu(frozen_bn(sd, p + ".bn2", F.conv2d(x, sd[p + ".conv2.weight"], padding=1)))
    x = F.relu(frozen_bn(sd, p + ".bn3", F.conv2d(x, sd[p + ".conv3.weight"], padding=1)))
    return F.avg_pool2d(x, 2)


def backbone(sd, cfg, x, p="backbone", want_res5=True):
    """ModifiedResNet.forward clip_backbone.py:193-219 -> {'res4','res5'}.
    ``want_res5=False`` skips the full-image layer4 whose output the caller never reads
    (supervised / region-level branches) -- results are unaffected."""
    x = stem(sd, p, x)
    x = res_layer(sd, p + ".layer1", x, cfg.layers[0], 1)
    x = res_layer(sd, p + ".layer2", x, cfg.layers[1], 2)
    res4 = res_layer(sd, p + ".layer3", x, cfg.layers[2], 2)
    out = {"res4": res4}
    if want_res5:
        out["res5"] = res_layer(sd, p + ".layer4", res4, cfg.layers[3], 2)
    return out


def layer4(sd, cfg, x, p="backbone"):
    return res_layer(sd, p + ".layer4", x, cfg.layers[3], 2)


def attnpool(sd, cfg, x, p="backbone.attnpool"):
    """AttentionPool2d.forward clip_backbone.py:83-107."""
    x = x.reshape(x.shape[0], x.shape[1], x.shape[2] * x.shape[3]).permute(2, 0, 1)
    x = torch.cat([x.mean(dim=0, keepdim=True), x], dim=0)
    x = x + sd[p + ".positional_embedding"][:, None, :].to(x.dtype)
    x, _ = F.multi_head_attention_forward(
        query=x, key=x, value=x, embed_dim_to_check=x.shape[-1], num_heads=cfg.heads,
        q_proj_weight=sd[p + ".q_proj.weight"], k_proj_weight=sd[p + ".k_proj.weight"],
        v_proj_weight=sd[p + ".v_proj.weight"], in_proj_weight=None,
        in_proj_bias=torch.cat([sd[p + ".q_proj.bias"], sd[p + ".k_proj.bias"], sd[p + ".v_proj.bias"]]),
        bias_k=None, bias_v=None, add_zero_attn=False, dropout_p=0,
        out_proj_weight=sd[p + ".c_proj.weight"], out_proj_bias=sd[p + ".c_proj.bias"],
        use_separate_proj_weight=True, training=True, need_weights=False)
    return x[0]


# ------------------------------------------------------------------------------------------------
# ClipCap mapper  (modeling/backbone/clipcap/clipcap.py:39-163,714-719)
# ------------------------------------------------------------------------------------------------
def mapper(msd, cfg, x):
    """TransformerMapper.forward clipcap.py:151-155 -> (N, 40, 768).  ``msd`` = clip_project.* state dict
    (keys without the ``clip_project.`` prefix)."""
    n, d, H = x.shape[0], cfg.mapper_dim, cfg.mapper_heads
    L = cfg.prefix_length
    h = F.linear(x, msd["linear.weight"], msd["linear.bias"]).view(n, L, d)
    prefix = msd["prefix_const"].unsqueeze(0).expand(n, *msd["prefix_const"].shape)
    h = torch.cat((h, prefix), dim=1)
    for i in range(cfg.mapper_layers):
        q = f"transformer.layers.{i}"
        # TransformerLayer.forward clipcap.py:97-100 (pre-LN), MultiHeadAttention.forward :69-86
        y = F.layer_norm(h, (d,), msd[q + ".norm1.weight"], msd[q + ".norm1.bias"])
        b, t, c = y.shape
        qs = F.linear(y, msd[q + ".attn.to_queries.weight"]).reshape(b, t, H, c // H)
        kv = F.linear(y, msd[q + ".attn.to_keys_values.weight"]).reshape(b, t, 2, H, c // H)
        k, v = kv[:, :, 0], kv[:, :, 1]
        att = torch.einsum("bnhd,bmhd->bnmh", qs, k) * ((c // H) ** -0.5)
        att = att.softmax(dim=2)
        o = torch.einsum("bnmh,bmhd->bnhd", att, v).reshape(b, t, c)
        h = h + F.linear(o, msd[q + ".attn.project.weight"], msd[q + ".attn.project.bias"])
        y = F.layer_norm(h, (d,), msd[q + ".norm2.weight"], msd[q + ".norm2.bias"])
        y = F.linear(F.relu(F.linear(y, msd[q + ".mlp.fc1.weight"], msd[q + ".mlp.fc1.bias"])),
                     msd[q + ".mlp.fc2.weight"], msd[q + ".mlp.fc2.bias"])
        h = h + y
    return h[:, L:]


def v2l(msd, cfg, prefix):
    """clipcap.py:714-719: keep the LAST of the 40 output tokens."""
    e = mapper(msd, cfg, prefix).reshape(-1, cfg.prefix_length, cfg.mapper_dim)[:, -1, :]
    return e.reshape(e.shape[0], -1)


def projector(sd, x):
    """meta_arch/rcnn.py:95-99."""
    x = F.relu(F.linear(x, sd["projector.0.weight"], sd["projector.0.bias"]))
    return F.linear(x, sd["projector.2.weight"], sd["projector.2.bias"])


# ------------------------------------------------------------------------------------------------
# preprocessing  (meta_arch/rcnn.py:161-207,758-768)
# ------------------------------------------------------------------------------------------------
def _mean_std(cfg):
    return (torch.tensor(cfg.pixel_mean).view(-1, 1, 1), torch.tensor(cfg.pixel_std).view(-1, 1, 1))


def preprocess_image(cfg, batched_inputs, key="image"):
    """rcnn.py:758-768 (div_pixel=True) / :196-207: normalise each image then zero-pad to the batch max."""
    mean, std = _mean_std(cfg)
    imgs = [((x[key].float() / 255.0) - mean) / std for x in batched_inputs]
    return ops.pad_batch(imgs)


def preprocess_image_train(cfg, batched_inputs, key):
    """rcnn.py:161-179: /255 -> pad -> bicubic short-side 224 -> center crop -> normalise."""
    mean, std = _mean_std(cfg)
    imgs = [x[key].float() / 255.0 for x in batched_inputs]
    t, _ = ops.pad_batch(imgs)
    t = ops.center_crop(ops.resize_short_bicubic(t, 224), 224)
    return (t - mean) / std


# ------------------------------------------------------------------------------------------------
# RPN  (modeling/proposal_generator/rpn.py, proposal_utils.py)
# ------------------------------------------------------------------------------------------------
def rpn_head(sd, feat, p="proposal_generator.rpn_head"):
    """StandardRPNHead.forward rpn.py:158-177."""
    t = F.relu(F.conv2d(feat, sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=1))
    logits = F.conv2d(t, sd[p + ".objectness_logits.weight"], sd[p + ".objectness_logits.bias"])
    deltas = F.conv2d(t, sd[p + ".anchor_deltas.weight"], sd[p + ".anchor_deltas.bias"])
    return logits, deltas


def rpn_flatten(logits, deltas):
    """rpn.py:456-467."""
    lg = logits.permute(0, 2, 3, 1).flatten(1)
    dl = deltas.view(deltas.shape[0], -1, 4, deltas.shape[-2], deltas.shape[-1]).permute(0, 3, 4, 1, 2).flatten(1, -2)
    return lg, dl


def rpn_label_and_sample(cfg, anchors, gt_boxes_list, gen, record=None):
    """RPN.label_and_sample_anchors rpn.py:305-363 (boundary thresh -1 => off)."""
    labels, matched = [], []
    for gt in gt_boxes_list:
        q = ops.pairwise_iou(gt, anchors)
        idx, lab = ops.matcher(q, cfg.rpn_iou_thresholds, cfg.rpn_iou_labels, True)
        if record is not None:
            record.setdefault("rpn_match_labels", []).append(lab.clone())
            record.setdefault("rpn_matches", []).append(idx.clone())
        pos, neg = ops.subsample_labels(lab, cfg.rpn_batch_per_image, cfg.rpn_positive_fraction, 0, gen)
        lab = lab.clone()
        lab.fill_(-1)
        lab.scatter_(0, pos, 1)
        lab.scatter_(0, neg, 0)
        labels.append(lab)
        matched.append(torch.zeros_like(anchors) if len(gt) == 0 else gt[idx])
    return labels, matched


def rpn_losses(cfg, anchors, logits, labels, deltas, matched):
    """RPN.losses rpn.py:365-429; _dense_box_regression_loss box_regression.py:229-270."""
    n = len(labels)
    gl = torch.stack(labels)
    pos = gl == 1
    gt_d = torch.stack([ops.get_deltas(anchors, k, cfg.rpn_bbox_weights) for k in matched])
    loc = ops.smooth_l1_loss(deltas[pos], gt_d[pos], cfg.rpn_smooth_l1_beta, "sum")
    valid = gl >= 0
    obj = F.binary_cross_entropy_with_logits(logits[valid], gl[valid].to(torch.float32), reduction="sum")
    norm = cfg.rpn_batch_per_image * n
    return {"loss_rpn_cls": obj / norm, "loss_rpn_loc": loc / norm}


def find_top_rpn_proposals(cfg, proposals, logits, image_sizes, training=True, record=None):
    """proposal_utils.py:22-130, single feature level."""
    out = []
    pre_topk = cfg.rpn_pre_nms_topk if training else cfg.rpn_pre_nms_topk_test
    post_topk = cfg.rpn_post_nms_topk if training else cfg.rpn_post_nms_topk_test
    k = min(logits.shape[1], pre_topk)
    srt = torch.sort(logits, descending=True, dim=1, stable=True)
    top_scores, top_idx = srt.values[:, :k], srt.indices[:, :k]
    for n, size in enumerate(image_sizes):
        boxes = proposals[n][top_idx[n]]
        scores = top_scores[n]
        valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores)
        if not valid.all():
            if training:
                raise FloatingPointError("Predicted boxes or scores contain Inf/NaN. Training has diverged.")
            boxes, scores = boxes[valid], scores[valid]
        boxes = ops.clip_boxes(boxes, size)
        keep = ops.nonempty(boxes, cfg.rpn_min_box_size)
        if keep.sum().item() != len(boxes):
            boxes, scores = boxes[keep], scores[keep]
        keep = ops.batched_nms(boxes, scores, torch.zeros(len(boxes), dtype=torch.int64), cfg.rpn_nms_thresh)
        keep = keep[:post_topk]
        if record is not None:
            record.setdefault("nms_keep", []).append(keep.clone())
        out.append((boxes[keep], scores[keep]))
    return out


def rpn_forward(sd, cfg, res4, image_sizes, gt_boxes_list, gen, training=True, record=None):
    """RPN.forward rpn.py:431-480 -> (proposals [(boxes, logits)], losses)."""
    anchors = ops.grid_anchors(res4.shape[-2], res4.shape[-1], cfg.feat_stride, cfg.anchor_offset,
                               cfg.anchor_sizes, cfg.anchor_ratios)
    logits, deltas = rpn_head(sd, res4)
    lg, dl = rpn_flatten(logits, deltas)
    losses = {}
    if training:
        labels, matched = rpn_label_and_sample(cfg, anchors, gt_boxes_list, gen, record)
        losses = rpn_losses(cfg, anchors, lg, labels, dl, matched)
        if record is not None:
            record["rpn_labels"] = labels
    with torch.no_grad():
        N = dl.shape[0]
        prop = ops.apply_deltas(dl.reshape(-1, 4), anchors.unsqueeze(0).expand(N, -1, -1).reshape(-1, 4),
                                cfg.rpn_bbox_weights).view(N, -1, 4)
        props = find_top_rpn_proposals(cfg, prop, lg.detach(), image_sizes, training, record)
    return props, losses


# ------------------------------------------------------------------------------------------------
# ROI heads  (modeling/roi_heads/roi_heads.py:123-319, clip_roi_heads.py:28-199, fast_rcnn.py)
# ------------------------------------------------------------------------------------------------
GT_LOGIT = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))  # proposal_utils.py:183


def label_and_sample_proposals(cfg, proposals, gt_boxes_list, gt_classes_list, gen, record=None):
    """ROIHeads.label_and_sample_proposals roi_heads.py:236-319 (+ _sample_proposals :184-234)."""
    out = []
    for (boxes, logits), gtb, gtc in zip(proposals, gt_boxes_list, gt_classes_list):
        if cfg.roi_append_gt:  # proposal_utils.py:133-200
            boxes = torch.cat([boxes, gtb])
            logits = torch.cat([logits, GT_LOGIT * torch.ones(len(gtb))])
        q = ops.pairwise_iou(gtb, boxes)
        midx, mlab = ops.matcher(q, cfg.roi_iou_thresholds, cfg.roi_iou_labels, False)
        if gtc.numel() > 0:
            cls = gtc[midx].clone()
            cls[mlab == 0] = cfg.num_classes
            cls[mlab == -1] = -1
        else:
            cls = torch.zeros_like(midx) + cfg.num_classes
        fg, bg = ops.subsample_labels(cls, cfg.roi_batch_per_image, cfg.roi_positive_fraction, cfg.num_classes, gen)
        sidx = torch.cat([fg, bg], dim=0)
        item = {"proposal_boxes": boxes[sidx], "objectness_logits": logits[sidx], "gt_classes": cls[sidx]}
        if gtc.numel() > 0:
            item["gt_boxes"] = gtb[midx[sidx]]
        if record is not None:
            record.setdefault("roi_sampled_idx", []).append(sidx.clone())
            record.setdefault("roi_gt_classes", []).append(cls[sidx].clone())
        out.append(item)
    return out


def boxes_to_rois(box_lists):
    """poolers.py:68-95 convert_boxes_to_pooler_format."""
    parts = [torch.cat([torch.full((len(b), 1), float(i)), b], dim=1) for i, b in enumerate(box_lists)]
    return torch.cat(parts, dim=0)


def roi_pool(cfg, res4, box_lists):
    """ROIPooler.forward poolers.py:190-229, single level, ROIAlignV2 (aligned=True)."""
    rois = boxes_to_rois(box_lists)
    if rois.shape[0] == 0:
        return torch.zeros((0, res4.shape[1], cfg.pooler_resolution, cfg.pooler_resolution))
    return ops.roi_align(res4, rois, cfg.pooler_resolution, 1.0 / cfg.feat_stride, cfg.pooler_sampling_ratio, True)


def box_predictor(sd, cfg, x, p="roi_heads.box_predictor"):
    """FastRCNNOutputLayers.forward fast_rcnn.py:529-572 (text-embedding classifier)."""
    nx = F.normalize(x, p=2.0, dim=1)
    cls = nx @ F.normalize(sd[p + ".cls_score.weight"], p=2.0, dim=1).t()
    bg = F.linear(nx, sd[p + ".cls_bg_score.weight"])
    scores = torch.cat((cls, bg), dim=1) / cfg.cls_temp
    deltas = F.linear(x, sd[p + ".bbox_pred.weight"], sd[p + ".bbox_pred.bias"])
    return scores, deltas


def focal_loss(cfg, scores, targets):
    """FastRCNNOutputLayers.focal_loss fast_rcnn.py:624-644."""
    ce = F.cross_entropy(scores, targets, reduction="none")
    p = F.softmax(scores, dim=-1)
    pt = p[torch.arange(p.size(0)), targets]
    loss = ce * ((1 - pt) ** cfg.focal_gamma)
    w = torch.ones(loss.size(0))
    w[targets == cfg.num_classes] = cfg.bg_cls_loss_weight
    return (loss * w).mean()


def box_reg_loss(cfg, proposal_boxes, gt_boxes, pred_deltas, gt_classes):
    """FastRCNNOutputLayers.box_reg_loss fast_rcnn.py:646-689."""
    fg = torch.nonzero((gt_classes >= 0) & (gt_classes < cfg.num_classes), as_tuple=True)[0]
    fg_pred = pred_deltas.view(-1, cfg.num_classes, 4)[fg, gt_classes[fg]]
    gt_d = ops.get_deltas(proposal_boxes[fg], gt_boxes[fg], cfg.roi_bbox_weights)
    loss = ops.smooth_l1_loss(fg_pred, gt_d, cfg.roi_smooth_l1_beta, "sum")
    return loss / max(gt_classes.numel(), 1.0)


def classification_stats(scores, gt_classes):
    """_log_classification_stats fast_rcnn.py:100-127."""
    n = gt_classes.numel()
    if n == 0:
        return {}
    pred = scores.argmax(dim=1)
    bg = scores.shape[1] - 1
    fg = (gt_classes >= 0) & (gt_classes < bg)
    nfg = int(fg.sum())
    out = {"fast_rcnn/cls_accuracy": float((pred == gt_classes).sum()) / n}
    if nfg > 0:
        out["fast_rcnn/fg_cls_accuracy"] = float((pred[fg] == gt_classes[fg]).sum()) / nfg
        out["fast_rcnn/false_negative"] = float((pred[fg] == bg).sum()) / nfg
    return out


def roi_heads_forward(sd, cfg, res4, proposals, gt_boxes_list, gt_classes_list, gen, record=None):
    """CLIPRes5ROIHeads.forward clip_roi_heads.py:134-175 (training)."""
    with torch.no_grad():
        sampled = label_and_sample_proposals(cfg, proposals, gt_boxes_list, gt_classes_list, gen, record)
    x = roi_pool(cfg, res4, [s["proposal_boxes"] for s in sampled])
    x = layer4(sd, cfg, x)
    feats = attnpool(sd, cfg, x)
    scores, deltas = box_predictor(sd, cfg, feats)
    gt_classes = torch.cat([s["gt_classes"] for s in sampled])
    pboxes = torch.cat([s["proposal_boxes"] for s in sampled])
    gboxes = torch.cat([s.get("gt_boxes", s["proposal_boxes"]) for s in sampled])
    if record is not None:
        record["roi_scores"] = scores.detach()
        record["roi_deltas"] = deltas.detach()
        record["roi_feats"] = feats.detach()
        record["stats"] = classification_stats(scores.detach(), gt_classes)
    return {"loss_cls": focal_loss(cfg, scores, gt_classes),
            "loss_box_reg": box_reg_loss(cfg, pboxes, gboxes, deltas, gt_classes)}


# ------------------------------------------------------------------------------------------------
# inference  (fast_rcnn.py:47-209,691-811; clip_roi_heads.py:171-174; meta_arch/rcnn.py:690-784; postprocessing.py:9-75)
# ------------------------------------------------------------------------------------------------
def fast_rcnn_inference_single_image(boxes, scores, image_shape, score_thresh, nms_thresh, topk_per_image):
    """fast_rcnn.py:130-209 with hard NMS: boxes [R, 4K], scores [R, K+1] -> (boxes [D,4], scores [D], classes [D], proposal idx [D])"""
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
    if not valid.all():
        boxes, scores = boxes[valid], scores[valid]
    scores = scores[:, :-1]
    k = boxes.shape[1] // 4
    boxes = ops.clip_boxes(boxes.reshape(-1, 4), image_shape).view(-1, k, 4)
    mask = scores > score_thresh
    inds = mask.nonzero()
    boxes = boxes[inds[:, 0], 0] if k == 1 else boxes[mask]
    scores = scores[mask]
    keep = ops.batched_nms(boxes, scores, inds[:, 1], nms_thresh)
    if topk_per_image >= 0:
        keep = keep[:topk_per_image]
    return boxes[keep], scores[keep], inds[keep, 1], inds[keep, 0]


def detector_postprocess(boxes, scores, classes, image_size, out_h, out_w):
    """postprocessing.py:9-75 (boxes): rescale to the output resolution, clip, drop empty boxes"""
    sx, sy = out_w / image_size[1], out_h / image_size[0]
    b = boxes.clone()
    b[:, 0::2] *= sx
    b[:, 1::2] *= sy
    b = ops.clip_boxes(b, (out_h, out_w))
    keep = ops.nonempty(b)
    return b[keep], scores[keep], classes[keep]


@torch.no_grad()
def inference(sd, cfg, batched_inputs):
    """GeneralizedRCNN.inference rcnn.py:690-756 -> per image dict(boxes, scores, classes) at the sample's height / width"""
    images, sizes = preprocess_image(cfg, batched_inputs, "image")
    res4 = backbone(sd, cfg, images, want_res5=False)["res4"]
    props, _ = rpn_forward(sd, cfg, res4, sizes, None, None, training=False)
    x = roi_pool(cfg, res4, [b for b, _ in props])
    feats = attnpool(sd, cfg, layer4(sd, cfg, x))
    scores, deltas = box_predictor(sd, cfg, feats)
    pboxes = torch.cat([b for b, _ in props])
    pred = ops.apply_deltas(deltas, pboxes, cfg.roi_bbox_weights)
    probs = F.softmax(scores, dim=-1)
    counts = [len(b) for b, _ in props]
    out = []
    for inp, size, pb, pr, (_, logit) in zip(batched_inputs, sizes, pred.split(counts), probs.split(counts), props):
        if cfg.multiply_rpn_score:
            pr = (pr * logit[:, None]) ** 0.5
        b, s, c, _ = fast_rcnn_inference_single_image(pb, pr, size, cfg.test_score_thresh, cfg.test_nms_thresh, cfg.detections_per_image)
        b, s, c = detector_postprocess(b, s, c, size, inp.get("height", size[0]), inp.get("width", size[1]))
        out.append({"boxes": b, "scores": s, "classes": c})
    return out


# ------------------------------------------------------------------------------------------------
# contrastive  (meta_arch/rcnn.py:255-319,422-470; backbone/clipcap/gather.py)
# ------------------------------------------------------------------------------------------------
def symmetric_ce(a, b):
    """rcnn.py:308-317: rows L2-normalised by plain division, S = a b^T, 0.5*(CE(S)+CE(S^T))."""
    a = a / a.norm(dim=1, keepdim=True)
    b = b / b.norm(dim=1, keepdim=True)
    s = a @ b.t()
    gt = torch.arange(len(s), dtype=torch.long)
    return (F.cross_entropy(s, gt) + F.cross_entropy(s.t(), gt)) / 2


def gather_cat(x, others=None, rank=0):
    """GatherLayer (gather.py:5-20) + torch.cat, simulated: ``others`` = list of the other ranks'
    (detached) tensors in rank order with this rank's slot skipped.  Backward = own slice only."""
    if not others:
        return x
    parts = list(others)
    parts.insert(rank, x)
    return torch.cat(parts, dim=0)


def v2l_contrastive(sd, msd, cfg, img_src, img_tgt, kd, others=None, rank=0, record=None):
    """GeneralizedRCNN.v2l_contrastive rcnn.py:255-319.  ``others`` = (other ranks' target embeddings, other ranks' source
    embeddings), each a list in rank order with this rank's slot skipped (``gather_cat``)."""
    ft = projector(sd, v2l(msd, cfg, attnpool(sd, cfg, backbone(sd, cfg, img_tgt)["res5"])))
    fs = v2l(msd, cfg, attnpool(sd, cfg, backbone(sd, cfg, img_src)["res5"]))
    kd_loss = None
    if kd:
        with torch.no_grad():
            teacher = v2l(msd, cfg, attnpool(sd, cfg, backbone(sd, cfg, img_src, p="offline_backbone")["res5"],
                                             p="offline_backbone.attnpool"))
        kd_loss = F.l1_loss(teacher.detach(), fs)
    fs = projector(sd, fs)
    if record is not None:
        record["img_emb_tgt"], record["img_emb_src"] = ft.detach(), fs.detach()
    o_t, o_s = (others or (None, None))
    ft = gather_cat(ft, o_t, rank)
    fs = gather_cat(fs, o_s, rank)
    return symmetric_ce(ft, fs), kd_loss


# ------------------------------------------------------------------------------------------------
# GeneralizedRCNN.forward  (meta_arch/rcnn.py:351-623)
# ------------------------------------------------------------------------------------------------
def _gt(batched_inputs):
    return ([x["instances"]["gt_boxes"].float() for x in batched_inputs],
            [x["instances"]["gt_classes"].long() for x in batched_inputs])


def forward(sd, cfg, batched_inputs, msd=None, branch="supervised", kd=True, gen=None, region_gen=None,
            record=None, others=None, rank=0):
    """Training-mode GeneralizedRCNN.forward.  ``batched_inputs``: list of dicts with uint8 CHW
    ``image`` / ``image_trgt`` and ``instances`` = {'gt_boxes' [G,4], 'gt_classes' [G]}."""
    gen = gen if gen is not None else torch.Generator().manual_seed(0)
    if branch == "caption_consistency":  # rcnn.py:413-421
        src = preprocess_image_train(cfg, batched_inputs, "image")
        tgt = preprocess_image_train(cfg, batched_inputs, "image_trgt")
        cont, kdl = v2l_contrastive(sd, msd, cfg, src, tgt, kd, others, rank, record)
        return {"cont_loss": cont, "kd_loss": kdl} if kdl is not None else {"cont_loss": cont}
    if branch == "caption_consistency_regionLevel":  # rcnn.py:422-470
        src, sizes = preprocess_image(cfg, batched_inputs, "image")
        tgt, _ = preprocess_image(cfg, batched_inputs, "image_trgt")
        fs = backbone(sd, cfg, src, want_res5=False)["res4"]
        ft = backbone(sd, cfg, tgt, want_res5=False)["res4"]
        gtb, _ = _gt(batched_inputs)
        with torch.no_grad():
            props, _ = rpn_forward(sd, cfg, fs.detach(), sizes, gtb, gen, True, None)
            rg = region_gen if region_gen is not None else gen
            sel = [torch.randperm(len(b), generator=rg)[: cfg.regions_per_image] for b, _ in props]
            boxes = [b[s] for (b, _), s in zip(props, sel)]
            if record is not None:
                record["region_boxes"] = boxes
        # clip_roi_heads.py:117-132 forward_get_features
        rs = attnpool(sd, cfg, layer4(sd, cfg, roi_pool(cfg, fs, boxes)))
        rt = attnpool(sd, cfg, layer4(sd, cfg, roi_pool(cfg, ft, boxes)))
        es = projector(sd, v2l(msd, cfg, rs))
        et = projector(sd, v2l(msd, cfg, rt))
        if record is not None:
            record["reg_emb_src"], record["reg_emb_tgt"] = es.detach(), et.detach()
        o_s, o_t = (others or (None, None))
        return symmetric_ce(gather_cat(es, o_s, rank), gather_cat(et, o_t, rank))
    # supervised: rcnn.py:592-623
    images, sizes = preprocess_image(cfg, batched_inputs, "image")
    gtb, gtc = _gt(batched_inputs)
    res4 = backbone(sd, cfg, images, want_res5=False)["res4"]
    if record is not None:
        record["res4"] = res4.detach()
    props, rpn_l = rpn_forward(sd, cfg, res4, sizes, gtb, gen, True, record)
    if record is not None:
        record["proposals"] = props
    det_l = roi_heads_forward(sd, cfg, res4, props, gtb, gtc, gen, record)
    losses = {}
    losses.update(det_l)
    losses.update(rpn_l)
    return losses


def run_step_losses(sd, msd, cfg, batched_inputs, iteration, gen=None, record=None, others=None, rank=0):
    """SimpleTrainer.run_step engine/train_loop.py:311-383 up to ``losses = sum(...)``.  ``others`` (simulated world size > 1):
    {"img": (other ranks' target embeddings, source embeddings), "reg": (source, target)} as ``record`` of those ranks' own
    runs gives them (``img_emb_*``, ``reg_emb_*``); ``rank`` = this rank's slot in the gathered batch."""
    gen = gen if gen is not None else torch.Generator().manual_seed(0)
    others = others or {}
    loss_dict = forward(sd, cfg, batched_inputs, branch="supervised", gen=gen, record=record)
    loss = {}
    if iteration > cfg.burn_in_iters:
        loss.update(forward(sd, cfg, batched_inputs, msd, "caption_consistency", cfg.kd_regularization, gen, record=record,
                            others=others.get("img"), rank=rank))
        loss["cont_region_loss"] = forward(sd, cfg, batched_inputs, msd, "caption_consistency_regionLevel",
                                           cfg.kd_regularization, gen, record=record, others=others.get("reg"), rank=rank)
    else:
        cc = forward(sd, cfg, batched_inputs, msd, "caption_consistency", False, gen)
        for k in cc:
            loss[k] = cc[k] * 0.0
    loss_dict.update(loss)
    return loss_dict


# ------------------------------------------------------------------------------------------------
# solver  (solver/build.py:43-130,220-262; solver/lr_scheduler.py:17-129)
# ------------------------------------------------------------------------------------------------
def lr_at(cfg, it):
    """WarmupParamScheduler(MultiStepParamScheduler) + LRMultiplier; KAT tests/test_scheduler.py:35-43."""
    steps = [s for s in cfg.steps if s <= cfg.max_iter]
    mult = cfg.gamma ** sum(1 for s in steps if it >= s)
    if it < cfg.warmup_iters:
        end = cfg.gamma ** sum(1 for s in steps if cfg.warmup_iters >= s)
        start = cfg.warmup_factor * 1.0  # warmup_factor * sched(0)
        a = it / cfg.warmup_iters
        mult = start * (1 - a) + end * a
    return cfg.base_lr * mult


def trainable_keys(sd, cfg, freeze_at=2):
    """Parameters with requires_grad in the reference model: everything except the stem + layer1 convs
    (FREEZE_AT 2, clip_backbone.py:250-261), FrozenBN buffers, the text embeddings (fast_rcnn.py:448-463)
    and ``offline_backbone.*`` (rcnn.py:107)."""
    keys = []
    for k in sd:
        if k.startswith("offline_backbone.") or ".bn" in k or "downsample.1" in k:
            continue
        if k.startswith("backbone.conv") or k.startswith("backbone.layer1."):
            continue
        if "cls_score" in k or "cls_bg_score" in k or "cell_anchors" in k:
            continue
        keys.append(k)
    return keys


def sgd_step(sd, grads, mom, cfg, it):
    """Per-parameter L2-norm clip to cfg.clip_value (solver/build.py:59-67,104) then
    torch.optim.SGD(momentum, weight_decay, nesterov=False) semantics; in place on ``sd``/``mom``."""
    lr = lr_at(cfg, it)
    for k, g in grads.items():
        if g is None:
            continue
        g = g.clone()
        nrm = g.norm(2)
        coef = torch.clamp(cfg.clip_value / (nrm + 1e-6), max=1.0)  # torch.nn.utils.clip_grad_norm_
        g = g * coef
        g = g + cfg.weight_decay * sd[k]
        if k not in mom:
            mom[k] = g.clone()
        else:
            mom[k].mul_(cfg.momentum).add_(g)
        sd[k] = sd[k] - lr * mom[k]
    return lr

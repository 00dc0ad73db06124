"""ORACLE (test infrastructure, CPU only): stock Detectron2 R50-C4 Faster R-CNN (BASELINE.json configs[0],
configs/PascalVOC-Detection/faster_rcnn_R_50_C4.yaml) restated on ATen/CPU fp32.

Follows (paths under /root/reference/detectron2): modeling/backbone/resnet.py:100-210 (BottleneckBlock, stride in the
1x1), :330-359 (BasicStem), :362-459,614-695 (ResNet / build_resnet_backbone); modeling/roi_heads/roi_heads.py:358-512
(Res5ROIHeads, mean pool); modeling/roi_heads/fast_rcnn.py:476-479,574-689 (plain linear classifier, CE, box L1);
modeling/meta_arch/rcnn.py:592-623,758-768 (supervised forward, BGR mean/std without /255).
Pinning: backbone (stem, res2-4) and the RoI head's res5 stage + mean pool, with gradients, against the reference's own
``backbone/resnet.py`` classes (tests/golden/ref_stock_resnet.npz, generator tests/golden/make_golden_step.py); RPN / RoIAlign /
matcher / sampling / losses are the pinned functions of oracle/model.py.
"""
import torch
import torch.nn.functional as F

from . import model as om
from . import ops


def cfg_r50():
    return om.Cfg(pixel_mean=(103.530, 116.280, 123.675), pixel_std=(1.0, 1.0, 1.0), focal_gamma=0.0, bg_cls_loss_weight=1.0)


def conv_bn(sd, p, x, stride=1, padding=0):
    """layers/wrappers.py:48-91 Conv2d(norm=FrozenBN): conv then the frozen affine."""
    return om.frozen_bn(sd, p + ".norm", F.conv2d(x, sd[p + ".weight"], stride=stride, padding=padding))


def bottleneck_block(sd, p, x, stride):
    """resnet.py:191-210 with STRIDE_IN_1X1 True (config/defaults.py:608)."""
    out = F.relu(conv_bn(sd, p + ".conv1", x, stride=stride))
    out = F.relu(conv_bn(sd, p + ".conv2", out, padding=1))
    out = conv_bn(sd, p + ".conv3", out)
    sc = conv_bn(sd, p + ".shortcut", x, stride=stride) if (p + ".shortcut.weight") in sd else x
    return F.relu(out + sc)


def stage(sd, p, x, nblocks, first_stride):
    for i in range(nblocks):
        x = bottleneck_block(sd, f"{p}.{i}", x, first_stride if i == 0 else 1)
    return x


def backbone(sd, x):
    """BasicStem (resnet.py:355-358) + res2..res4 -> res4 (stride 16, 1024 ch)."""
    x = F.relu(conv_bn(sd, "backbone.stem.conv1", x, stride=2, padding=3))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    x = stage(sd, "backbone.res2", x, 3, 1)
    x = stage(sd, "backbone.res3", x, 4, 2)
    return stage(sd, "backbone.res4", x, 6, 2)


def preprocess(cfg, batched_inputs):
    """rcnn.py:758-768, div_pixel False: (x - mean) / std on raw 0-255 pixels, zero pad."""
    mean, std = om._mean_std(cfg)
    return ops.pad_batch([(x["image"].float() - mean) / std for x in batched_inputs])


def forward(sd, cfg, batched_inputs, gen, record=None):
    """GeneralizedRCNN.forward, supervised branch, stock heads (rcnn.py:592-623)."""
    images, sizes = preprocess(cfg, batched_inputs)
    gtb, gtc = om._gt(batched_inputs)
    res4 = backbone(sd, images)
    if record is not None:
        record["res4"] = res4.detach()
    props, rpn_l = om.rpn_forward(sd, cfg, res4, sizes, gtb, gen, True, record)
    with torch.no_grad():
        sampled = om.label_and_sample_proposals(cfg, props, gtb, gtc, gen, record)
    x = om.roi_pool(cfg, res4, [s["proposal_boxes"] for s in sampled])
    x = stage(sd, "roi_heads.res5", x, 3, 2)
    feats = x.mean(dim=[2, 3])                                            # roi_heads.py:487
    p = "roi_heads.box_predictor"
    scores = F.linear(feats, sd[p + ".cls_score.weight"], sd[p + ".cls_score.bias"])        # fast_rcnn.py:476-479,567
    deltas = F.linear(feats, sd[p + ".bbox_pred.weight"], sd[p + ".bbox_pred.bias"])
    gt_classes = torch.cat([s["gt_classes"] for s in sampled])
    pboxes = torch.cat([s["proposal_boxes"] for s in sampled])
    gboxes = torch.cat([s.get("gt_boxes", s["proposal_boxes"]) for s in sampled])
    losses = {"loss_cls": F.cross_entropy(scores, gt_classes, reduction="mean"),
              "loss_box_reg": om.box_reg_loss(cfg, pboxes, gboxes, deltas, gt_classes)}
    losses.update(rpn_l)
    return losses


def trainable_keys(sd):
    """FREEZE_AT 2: stem + res2 frozen (resnet.py:417-441); FrozenBN buffers are not parameters."""
    out = []
    for k in sd:
        if ".norm." in k or k.startswith("backbone.stem.") or k.startswith("backbone.res2."):
            continue
        out.append(k)
    return out

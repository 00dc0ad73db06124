"""Seeded synthetic weights and batches (no checkpoints or datasets exist offline).

Weights use the reference's state-dict key names (``backbone.layer3.0.conv1.weight`` ...), so a real
RegionCLIP checkpoint can replace them.  The init is chosen so activations stay O(1) through the 16
bottlenecks (He fan-in convs, small residual-branch BN gain) -- with the reference's own init
(``N(0, 0.01)`` RPN head etc.) random-weight logits would be degenerate and exercise nothing.
Everything is drawn on the CPU generator so the same tensors are reproducible on any box.
"""
import math
from typing import Dict, List

import torch

RN50_LAYERS = (3, 4, 6, 3)


def _conv(g, cout, cin, k, gain=2.0):
    fan_in = cin * k * k
    return torch.randn(cout, cin, k, k, generator=g) * math.sqrt(gain / fan_in)


def _bn(sd, g, p, c, wlo=0.8, whi=1.2):
    sd[p + ".weight"] = torch.empty(c).uniform_(wlo, whi, generator=g)
    sd[p + ".bias"] = torch.randn(c, generator=g) * 0.1
    sd[p + ".running_mean"] = torch.randn(c, generator=g) * 0.1
    sd[p + ".running_var"] = torch.empty(c).uniform_(0.5, 1.5, generator=g)


def _backbone(sd, g, p, layers=RN50_LAYERS, width=64, embed_dim=1024, spacial=7):
    sd[p + ".conv1.weight"] = _conv(g, width // 2, 3, 3)
    _bn(sd, g, p + ".bn1", width // 2)
    sd[p + ".conv2.weight"] = _conv(g, width // 2, width // 2, 3)
    _bn(sd, g, p + ".bn2", width // 2)
    sd[p + ".conv3.weight"] = _conv(g, width, width // 2, 3)
    _bn(sd, g, p + ".bn3", width)
    inpl = width
    for li, nb in enumerate(layers):
        planes = width * (2 ** li)
        stride = 1 if li == 0 else 2
        for b in range(nb):
            q = f"{p}.layer{li + 1}.{b}"
            sd[q + ".conv1.weight"] = _conv(g, planes, inpl, 1)
            _bn(sd, g, q + ".bn1", planes)
            sd[q + ".conv2.weight"] = _conv(g, planes, planes, 3)
            _bn(sd, g, q + ".bn2", planes)
            sd[q + ".conv3.weight"] = _conv(g, planes * 4, planes, 1)
            _bn(sd, g, q + ".bn3", planes * 4, 0.2, 0.4)
            if b == 0 and (stride > 1 or inpl != planes * 4):
                sd[q + ".downsample.0.weight"] = _conv(g, planes * 4, inpl, 1, gain=1.0)
                _bn(sd, g, q + ".downsample.1", planes * 4)
            inpl = planes * 4
    c = width * 32
    a = p + ".attnpool"
    sd[a + ".positional_embedding"] = torch.randn(spacial * spacial + 1, c, generator=g) / c ** 0.5
    for n, o in (("k_proj", c), ("q_proj", c), ("v_proj", c), ("c_proj", embed_dim)):
        sd[f"{a}.{n}.weight"] = torch.randn(o, c, generator=g) * c ** -0.5
        sd[f"{a}.{n}.bias"] = torch.randn(o, generator=g) * 0.02


def make_state_dict(seed=0, num_classes=20, layers=RN50_LAYERS, width=64, embed_dim=1024,
                    num_anchors=15) -> Dict[str, torch.Tensor]:
    """GeneralizedRCNN state dict (student + offline copy + RPN head + box predictor + projector)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    _backbone(sd, g, "backbone", layers, width, embed_dim)
    for k in [k for k in sd if k.startswith("backbone.")]:
        sd["offline_" + k] = sd[k].clone()  # offline teacher = copy of the student at t=0
    c4 = width * 16
    r = "proposal_generator.rpn_head"
    sd[r + ".conv.weight"] = _conv(g, c4, c4, 3)
    sd[r + ".conv.bias"] = torch.randn(c4, generator=g) * 0.02
    sd[r + ".objectness_logits.weight"] = torch.randn(num_anchors, c4, 1, 1, generator=g) * (2.0 / c4) ** 0.5
    sd[r + ".objectness_logits.bias"] = torch.randn(num_anchors, generator=g) * 0.02
    sd[r + ".anchor_deltas.weight"] = torch.randn(num_anchors * 4, c4, 1, 1, generator=g) * 0.25 * (2.0 / c4) ** 0.5
    sd[r + ".anchor_deltas.bias"] = torch.randn(num_anchors * 4, generator=g) * 0.02
    b = "roi_heads.box_predictor"
    t = torch.randn(num_classes, embed_dim, generator=g)
    sd[b + ".cls_score.weight"] = t / t.norm(dim=1, keepdim=True)
    sd[b + ".cls_bg_score.weight"] = torch.zeros(1, embed_dim)
    sd[b + ".bbox_pred.weight"] = torch.randn(num_classes * 4, embed_dim, generator=g) * 0.5 * embed_dim ** -0.5
    sd[b + ".bbox_pred.bias"] = torch.zeros(num_classes * 4)
    sd["projector.0.weight"] = torch.randn(768, 768, generator=g) * 768 ** -0.5
    sd["projector.0.bias"] = torch.randn(768, generator=g) * 0.02
    sd["projector.2.weight"] = torch.randn(256, 768, generator=g) * 768 ** -0.5
    sd["projector.2.bias"] = torch.randn(256, generator=g) * 0.02
    return sd


def drift_offline(sd: Dict[str, torch.Tensor], factor: float = 1.05) -> Dict[str, torch.Tensor]:
    """The synthetic teacher (``offline_backbone.*``) starts as a copy of the student, which makes ``kd_loss`` exactly zero.
    For checks that need a live KD term: scale the teacher's layer3 / layer4 conv weights by ``factor`` and its attention
    pool's output projection by ``1 / factor`` (deterministic, in place; returns ``sd``)."""
    for k, v in sd.items():
        if not k.startswith("offline_backbone."):
            continue
        if (".layer3." in k or ".layer4." in k) and k.endswith(".weight") and v.dim() == 4:
            v.mul_(factor)
        elif k.endswith("attnpool.c_proj.weight"):
            v.div_(factor)
    return sd


def make_state_dict_r50(seed=0, num_classes=20, num_anchors=15) -> Dict[str, torch.Tensor]:
    """Stock Detectron2 R50-C4 Faster R-CNN state dict (reference key names: ``backbone.stem.conv1.norm.weight``,
    ``backbone.res3.0.shortcut.weight``, ``roi_heads.res5.2.conv3.weight``, ``roi_heads.box_predictor.cls_score.bias``)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def conv(p, cout, cin, k, gain=2.0, wlo=0.8, whi=1.2):
        sd[p + ".weight"] = _conv(g, cout, cin, k, gain)
        _bn(sd, g, p + ".norm", cout, wlo, whi)

    conv("backbone.stem.conv1", 64, 3, 7)
    sd["backbone.stem.conv1.weight"] *= 1.0 / 64.0     # inputs are raw 0-255 pixels (mean-subtracted, std 1): keep activations O(1)
    inc = 64
    for name, nb, bott, outc in (("backbone.res2", 3, 64, 256), ("backbone.res3", 4, 128, 512), ("backbone.res4", 6, 256, 1024),
                                 ("roi_heads.res5", 3, 512, 2048)):
        for b in range(nb):
            q = f"{name}.{b}"
            if inc != outc:
                conv(q + ".shortcut", outc, inc, 1, gain=1.0)
            conv(q + ".conv1", bott, inc, 1)
            conv(q + ".conv2", bott, bott, 3)
            conv(q + ".conv3", outc, bott, 1, wlo=0.2, whi=0.4)
            inc = outc
    c4 = 1024
    r = "proposal_generator.rpn_head"
    sd[r + ".conv.weight"] = _conv(g, c4, c4, 3)
    sd[r + ".conv.bias"] = torch.randn(c4, generator=g) * 0.02
    sd[r + ".objectness_logits.weight"] = torch.randn(num_anchors, c4, 1, 1, generator=g) * (2.0 / c4) ** 0.5
    sd[r + ".objectness_logits.bias"] = torch.randn(num_anchors, generator=g) * 0.02
    sd[r + ".anchor_deltas.weight"] = torch.randn(num_anchors * 4, c4, 1, 1, generator=g) * 0.25 * (2.0 / c4) ** 0.5
    sd[r + ".anchor_deltas.bias"] = torch.randn(num_anchors * 4, generator=g) * 0.02
    b = "roi_heads.box_predictor"
    sd[b + ".cls_score.weight"] = torch.randn(num_classes + 1, 2048, generator=g) * 2048 ** -0.5
    sd[b + ".cls_score.bias"] = torch.randn(num_classes + 1, generator=g) * 0.02
    sd[b + ".bbox_pred.weight"] = torch.randn(num_classes * 4, 2048, generator=g) * 0.5 * 2048 ** -0.5
    sd[b + ".bbox_pred.bias"] = torch.zeros(num_classes * 4)
    return sd


def make_mapper_state_dict(seed=1, dim_clip=1024, dim=768, length=40, layers=8, std=0.02) -> Dict[str, torch.Tensor]:
    """``clip_project.*`` (TransformerMapper) state dict, keys without the ``clip_project.`` prefix."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    sd["linear.weight"] = torch.randn(length * dim, dim_clip, generator=g) * std
    sd["linear.bias"] = torch.randn(length * dim, generator=g) * std
    sd["prefix_const"] = torch.randn(length, dim, generator=g)
    for i in range(layers):
        q = f"transformer.layers.{i}"
        sd[q + ".norm1.weight"] = 1.0 + 0.1 * torch.randn(dim, generator=g)
        sd[q + ".norm1.bias"] = 0.1 * torch.randn(dim, generator=g)
        sd[q + ".attn.to_queries.weight"] = torch.randn(dim, dim, generator=g) * std
        sd[q + ".attn.to_keys_values.weight"] = torch.randn(2 * dim, dim, generator=g) * std
        sd[q + ".attn.project.weight"] = torch.randn(dim, dim, generator=g) * std
        sd[q + ".attn.project.bias"] = torch.randn(dim, generator=g) * std
        sd[q + ".norm2.weight"] = 1.0 + 0.1 * torch.randn(dim, generator=g)
        sd[q + ".norm2.bias"] = 0.1 * torch.randn(dim, generator=g)
        sd[q + ".mlp.fc1.weight"] = torch.randn(2 * dim, dim, generator=g) * std
        sd[q + ".mlp.fc1.bias"] = torch.randn(2 * dim, generator=g) * std
        sd[q + ".mlp.fc2.weight"] = torch.randn(dim, 2 * dim, generator=g) * std
        sd[q + ".mlp.fc2.bias"] = torch.randn(dim, generator=g) * std
    return sd


def make_batch(batch_size, height=800, width=1333, rank=0, iteration=0, num_gt=3, num_classes=20,
               seed=1234) -> List[dict]:
    """VOC-shaped paired samples: uint8 CHW ``image``, correlated ``image_trgt``
    (the domain-translated twin), ``instances`` = {'gt_boxes' f32 [G,4] XYXY abs, 'gt_classes' i64 [G]}."""
    g = torch.Generator().manual_seed(seed + 1000 * rank + iteration)
    out = []
    for i in range(batch_size):
        # low-frequency structure + pixel noise, so different samples give different features
        coarse = torch.rand(1, 3, max(height // 32, 2), max(width // 32, 2), generator=g)
        smooth = torch.nn.functional.interpolate(coarse, size=(height, width), mode="bilinear", align_corners=False)[0]
        img = (smooth * 200.0 + torch.rand(3, height, width, generator=g) * 55.0).clamp_(0, 255).to(torch.uint8)
        noise = torch.randn(3, height, width, generator=g) * 20.0
        tgt = (img.float() + noise).clamp_(0, 255).to(torch.uint8)
        x0 = torch.rand(num_gt, generator=g) * 0.6 * width
        y0 = torch.rand(num_gt, generator=g) * 0.6 * height
        mw, mh = min(64.0, 0.2 * width), min(64.0, 0.2 * height)
        w = mw + torch.rand(num_gt, generator=g) * (0.4 * width - mw)
        h = mh + torch.rand(num_gt, generator=g) * (0.4 * height - mh)
        boxes = torch.stack([x0, y0, (x0 + w).clamp(max=width), (y0 + h).clamp(max=height)], dim=1)
        classes = torch.randint(0, num_classes, (num_gt,), generator=g)
        out.append({"image": img, "image_trgt": tgt, "height": height, "width": width,
                    "image_id": rank * 100000 + iteration * 1000 + i,
                    "instances": {"gt_boxes": boxes.float(), "gt_classes": classes.long(),
                                  "image_size": (height, width)}})
    return out

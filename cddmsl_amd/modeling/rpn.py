"""RPN on the HIP box kernels -- detectron2/modeling/proposal_generator/rpn.py:66-533, proposal_utils.py:22-200,
anchor_generator.py:81-228, matcher.py, sampling.py, box_regression.py.

Device pipeline per batch (no G x N IoU matrix is ever materialised, no per-image sort/NMS launches):
  head convs (MFMA implicit GEMM) -> fused IoU+Matcher per image -> host-replayable subsample (CPU generator)
  -> segmented stable radix sort of all images' logits -> decode+clip top-k -> batched bitmask NMS.
Random draws follow sampling.py:47-48 order (positives, then negatives, image by image) on one CPU generator so
the CPU oracle replays them exactly.
"""
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from .. import hip, layers
from .._lib import Readback, to_device_async
from ..registry import ANCHOR_GENERATOR_REGISTRY, PROPOSAL_GENERATOR_REGISTRY, RPN_HEAD_REGISTRY
from ..structures import Boxes, Instances, as_instances
from .backbone import to_nhwc, to_nchw

SCALE_CLAMP = math.log(1000.0 / 16)  # box_regression.py:13


# ------------------------------------------------------------------------------------------------ box math (fp32 torch glue)
def get_deltas(src, tgt, weights):
    """Box2BoxTransform.get_deltas box_regression.py:42-75"""
    sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
    sx, sy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
    tw, th = tgt[:, 2] - tgt[:, 0], tgt[:, 3] - tgt[:, 1]
    tx, ty = tgt[:, 0] + 0.5 * tw, tgt[:, 1] + 0.5 * th
    wx, wy, ww, wh = weights
    return torch.stack((wx * (tx - sx) / sw, wy * (ty - sy) / sh, ww * torch.log(tw / sw), wh * torch.log(th / sh)), dim=1)


def apply_deltas(deltas, boxes, weights, scale_clamp=SCALE_CLAMP):
    """Box2BoxTransform.apply_deltas box_regression.py:77-115: deltas [N, 4k] (k class-specific transforms), boxes [N, 4]."""
    deltas = deltas.float()
    boxes = boxes.to(deltas.dtype)
    w, h = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    cx, cy = boxes[:, 0] + 0.5 * w, boxes[:, 1] + 0.5 * h
    wx, wy, ww, wh = weights
    dx, dy = deltas[:, 0::4] / wx, deltas[:, 1::4] / wy
    dw = torch.clamp(deltas[:, 2::4] / ww, max=scale_clamp)
    dh = torch.clamp(deltas[:, 3::4] / wh, max=scale_clamp)
    pcx, pcy = dx * w[:, None] + cx[:, None], dy * h[:, None] + cy[:, None]
    pw, ph = torch.exp(dw) * w[:, None], torch.exp(dh) * h[:, None]
    out = torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), dim=-1)
    return out.reshape(deltas.shape)


def subsample_labels(labels, num_samples, positive_fraction, bg_label, gen):
    """sampling.py:9-54 with both permutations drawn from the replayable CPU generator ``gen``."""
    positive = torch.nonzero((labels != -1) & (labels != bg_label), as_tuple=True)[0]
    negative = torch.nonzero(labels == bg_label, as_tuple=True)[0]
    num_pos = min(positive.numel(), int(num_samples * positive_fraction))
    num_neg = min(negative.numel(), num_samples - num_pos)
    perm1 = torch.randperm(positive.numel(), generator=gen)[:num_pos].to(labels.device)
    perm2 = torch.randperm(negative.numel(), generator=gen)[:num_neg].to(labels.device)
    return positive[perm1], negative[perm2]


def subsample_begin(label_list, bg_label, lens=None):
    """Device half of the batched ``subsample_labels``: masks, running counts, and the (asynchronous) readback of the
    per-image positive / negative counts.  Independent device work may be enqueued before ``subsample_finish``.
    ``label_list``: per-image label vectors, or (with ``lens``) ONE flat vector holding the images back to back."""
    if lens is None:
        lens = [int(l.numel()) for l in label_list]
        cat = torch.cat(label_list)
    else:
        cat = label_list
    pmask, nmask = (cat != -1) & (cat != bg_label), cat == bg_label
    offs = torch.tensor([0] + lens).cumsum(0)
    # per-image counts: cumulative sums sampled at the image boundaries -> ONE small D2H copy (the stage's only sync)
    cs = torch.stack([pmask.cumsum(0), nmask.cumsum(0)])
    ends = to_device_async((offs[1:] - 1).clamp(min=0), cat.device)
    return {"n": len(lens), "offs": offs, "pmask": pmask, "nmask": nmask, "rb": Readback(cs[:, ends]), "dev": cat.device}


def subsample_finish(st, num_samples, positive_fraction, gen, counts_out=None):
    """Host half: wait for the counts, draw the per-image permutations in the reference's order (image by image: positives,
    negatives -- sampling.py:47-48), ship all index lists back in one non-blocking copy.  The index lists have a known size
    (nonzero_static), so building them does not sync again."""
    offs = st["offs"]
    cnt_end = st["rb"].get() * (offs[1:] > 0)                # (an empty leading image has no last element to sample)
    cnt = torch.cat([torch.zeros(2, 1, dtype=cnt_end.dtype), cnt_end], dim=1)
    pos_all = torch.nonzero_static(st["pmask"], size=int(cnt[0, -1]))[:, 0]
    neg_all = torch.nonzero_static(st["nmask"], size=int(cnt[1, -1]))[:, 0]
    out = []
    for i in range(st["n"]):
        p0, p1 = int(cnt[0, i]), int(cnt[0, i + 1])
        n0, n1 = int(cnt[1, i]), int(cnt[1, i + 1])
        npos, nneg = p1 - p0, n1 - n0
        if counts_out is not None:
            counts_out.append((npos, nneg))
        num_pos = min(npos, int(num_samples * positive_fraction))
        num_neg = min(nneg, num_samples - num_pos)
        perm1 = torch.randperm(npos, generator=gen)[:num_pos]
        perm2 = torch.randperm(nneg, generator=gen)[:num_neg]
        out.append((perm1 + p0, perm2 + n0, int(offs[i])))
    sel = to_device_async(torch.cat([o[0] for o in out] + [o[1] for o in out]), st["dev"])
    npos_sel = sum(len(o[0]) for o in out)
    sel_pos, sel_neg = sel[:npos_sel], sel[npos_sel:]
    pos_idx, neg_idx = pos_all[sel_pos], neg_all[sel_neg]         # indices into the concatenated label vector, image by image
    st["global"] = (pos_idx, neg_idx, [len(o[0]) for o in out], [len(o[1]) for o in out])
    if st.get("global_only"):
        return None
    res, a, b = [], 0, 0
    for (p, n, off) in out:
        res.append((pos_idx[a:a + len(p)] - off, neg_idx[b:b + len(n)] - off))
        a += len(p)
        b += len(n)
    return res


def subsample_labels_batched(label_list, num_samples, positive_fraction, bg_label, gen, counts_out=None):
    """``subsample_labels`` for a list of per-image label vectors with ONE device->host sync in total instead of two per image."""
    return subsample_finish(subsample_begin(label_list, bg_label), num_samples, positive_fraction, gen, counts_out)


# ------------------------------------------------------------------------------------------------ anchors
@ANCHOR_GENERATOR_REGISTRY.register()
class DefaultAnchorGenerator(nn.Module):
    """anchor_generator.py:81-228 (single feature level)."""
    box_dim = 4

    def __init__(self, cfg=None, input_shape=None, *, sizes=None, aspect_ratios=None, strides=None, offset=0.0):
        super().__init__()
        if cfg is not None:
            sizes, aspect_ratios = cfg.MODEL.ANCHOR_GENERATOR.SIZES, cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS
            strides, offset = [s.stride for s in input_shape], cfg.MODEL.ANCHOR_GENERATOR.OFFSET
        assert len(strides) == 1, "C4 models have one feature level"
        self.strides, self.offset = strides, float(offset)
        assert 0.0 <= self.offset < 1.0, self.offset
        cell = []
        for size in sizes[0]:
            area = size ** 2.0
            for r in aspect_ratios[0]:
                w = math.sqrt(area / r)
                h = r * w
                cell.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
        self.register_buffer("cell_anchors_0", torch.tensor(cell), persistent=False)
        self._cache = {}

    @property
    def num_anchors(self):
        return [self.cell_anchors_0.shape[0]]

    def grid(self, hf, wf):
        key = (hf, wf, self.cell_anchors_0.device)
        if key not in self._cache:
            self._cache[key] = hip.anchors(self.cell_anchors_0, hf, wf, float(self.strides[0]), self.offset)
        return self._cache[key]

    def forward(self, features):
        return [Boxes(self.grid(f.shape[-2], f.shape[-1])) for f in features]


def build_anchor_generator(cfg, input_shape):
    return ANCHOR_GENERATOR_REGISTRY.get(cfg.MODEL.ANCHOR_GENERATOR.NAME)(cfg, input_shape)


# ------------------------------------------------------------------------------------------------ head
class _Conv2d(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k).normal_(std=0.01).contiguous(memory_format=torch.channels_last))
        self.bias = nn.Parameter(torch.zeros(cout))
        self._pw = None

    def pw(self):
        if self._pw is None or self._pw.param is not self.weight:
            self._pw = layers.PreparedWeight(self.weight, None, frozen=False)
        return self._pw


@RPN_HEAD_REGISTRY.register()
class StandardRPNHead(nn.Module):
    """rpn.py:66-177: 3x3 conv + ReLU, 1x1 objectness (A), 1x1 anchor deltas (4A)."""

    def __init__(self, cfg=None, input_shape=None, *, in_channels=None, num_anchors=None, box_dim=4):
        super().__init__()
        if cfg is not None:
            in_channels = input_shape[0].channels
            num_anchors = build_anchor_generator(cfg, input_shape).num_anchors[0]
            assert list(cfg.MODEL.RPN.CONV_DIMS) == [-1]
        self.fp8 = cfg is not None and cfg.MODEL.get("COMPUTE_DTYPE", "bf16") == "fp8"
        self.conv = _Conv2d(in_channels, in_channels, 3)
        self.objectness_logits = _Conv2d(in_channels, num_anchors, 1)
        self.anchor_deltas = _Conv2d(in_channels, num_anchors * box_dim, 1)

    def forward_nhwc(self, x):
        t = layers.conv(x, self.conv.pw(), self.conv.bias, 1, 1, relu=True, fp8=self.fp8)
        a = self.objectness_logits.weight.shape[0]
        y = layers.fused_heads(t, [(self.objectness_logits.weight, self.objectness_logits.bias),
                                   (self.anchor_deltas.weight, self.anchor_deltas.bias)])      # one GEMM, N = 5A (padded)
        return y[..., :a], y[..., a:5 * a]     # NHWC f32: [N,H,W,A], [N,H,W,4A]

    def forward(self, features):
        lg, dl = self.forward_nhwc(to_nhwc(features[0]))
        return [to_nchw(lg)], [to_nchw(dl)]


# ------------------------------------------------------------------------------------------------ RPN
@PROPOSAL_GENERATOR_REGISTRY.register()
class RPN(nn.Module):
    def __init__(self, cfg, input_shape: Dict[str, object]):
        super().__init__()
        r = cfg.MODEL.RPN
        self.in_features = r.IN_FEATURES
        shapes = [input_shape[f] for f in self.in_features]
        self.anchor_generator = build_anchor_generator(cfg, shapes)
        self.rpn_head = RPN_HEAD_REGISTRY.get(r.HEAD_NAME)(cfg, shapes)
        self.iou_thresholds, self.iou_labels = list(r.IOU_THRESHOLDS), list(r.IOU_LABELS)
        self.batch_size_per_image, self.positive_fraction = r.BATCH_SIZE_PER_IMAGE, r.POSITIVE_FRACTION
        self.pre_nms_topk = {True: r.PRE_NMS_TOPK_TRAIN, False: r.PRE_NMS_TOPK_TEST}
        self.post_nms_topk = {True: r.POST_NMS_TOPK_TRAIN, False: r.POST_NMS_TOPK_TEST}
        self.nms_thresh, self.min_box_size = r.NMS_THRESH, float(cfg.MODEL.PROPOSAL_GENERATOR.MIN_SIZE)
        self.weights = tuple(r.BBOX_REG_WEIGHTS)
        self.smooth_l1_beta = r.SMOOTH_L1_BETA
        assert r.BBOX_REG_LOSS_TYPE == "smooth_l1" and self.smooth_l1_beta == 0.0 and r.BOUNDARY_THRESH < 0
        lw = r.LOSS_WEIGHT
        self.loss_weight = {"loss_rpn_cls": lw, "loss_rpn_loc": lw * r.BBOX_REG_LOSS_WEIGHT}
        self.sample_generator = torch.Generator()  # replayable CPU stream (seed it per rank: utils/env.py:27-46)
        self.storage = {}

    @torch.no_grad()
    def label_anchors_begin(self, anchors, gt_instances):
        """rpn.py:305-363, device half: IoU + Matcher per image (written into rows of ONE [N, A] tensor pair), counts readback."""
        N, A = len(gt_instances), anchors.shape[0]
        gts = [gi.gt_boxes.tensor.float().contiguous() for gi in gt_instances]
        midx, labels = hip.iou_match_batched(gts, anchors, None, self.iou_thresholds, self.iou_labels, True)     # [N, A] each
        st = subsample_begin(labels.view(-1), 0, lens=[A] * N)
        st["global_only"] = True
        return labels, (midx, gts), st

    @torch.no_grad()
    def label_anchors_finish(self, labels, matched, st):
        """host half: sample 256 anchors per image (<= 128 positive) and write the {-1, 0, 1} labels (three batched fills)"""
        self.last_counts = []
        subsample_finish(st, self.batch_size_per_image, self.positive_fraction, self.sample_generator, self.last_counts)
        pos_g, neg_g, _, _ = st["global"]
        flat = labels.view(-1)
        flat.fill_(-1)
        flat.index_fill_(0, pos_g, 1)               # (``flat[pos_g] = 1`` stages its scalar through a synchronizing host-to-device copy)
        flat.index_fill_(0, neg_g, 0)
        self.last_pos_global, self.last_neg_global = pos_g, neg_g
        return labels, matched

    def label_and_sample_anchors(self, anchors, gt_instances):
        """rpn.py:305-363 -> (labels int8 [N,A], (matched gt index [N,A], per-image gt boxes))."""
        return self.label_anchors_finish(*self.label_anchors_begin(anchors, gt_instances))

    def replay_sampling_draws(self, counts):
        """Advance the sampling generator exactly as one ``label_and_sample_anchors`` call over images with these
        (num_positive, num_negative) anchor counts would (used when a second, gradient-free RPN pass over the same
        features is elided: its proposals are the first pass's, its random draws are still consumed)."""
        for npos, nneg in counts:
            torch.randperm(npos, generator=self.sample_generator)
            torch.randperm(nneg, generator=self.sample_generator)

    def losses(self, anchors, logits, labels, deltas, matched):
        """rpn.py:365-429 (+ _dense_box_regression_loss box_regression.py:229-270, smooth-L1 beta 0 = L1).
        ``labels`` int8 [N, A]; ``matched`` = (matched gt index [N, A], per-image gt boxes)."""
        gl = labels
        n, A = gl.shape
        midx, gts = matched
        pos_g = self.last_pos_global
        gt_off = to_device_async(torch.tensor([0] + [len(g) for g in gts]).cumsum(0)[:-1], gl.device)
        gt_cat = torch.cat(gts) if sum(len(g) for g in gts) else torch.zeros((1, 4), device=gl.device)
        norm = self.batch_size_per_image * n
        if logits.is_cuda and logits.dtype == torch.float32 and getattr(self, "last_neg_global", None) is not None:
            # both losses over the sampled anchors' index lists (known on the host side of the sampling: no mask, no dense pass)
            neg_g = self.last_neg_global
            self.storage["rpn/num_pos_anchors"] = pos_g.numel() / n      # (= (labels == 1).sum() / n: every positive label IS a sampled index)
            self.storage["rpn/num_neg_anchors"] = neg_g.numel() / n
            both = layers.rpn_losses(logits.reshape(-1), deltas.reshape(-1, 4), pos_g, neg_g, midx.view(-1), gt_cat.float().contiguous(), gt_off,
                                     anchors, self.weights, 1.0 / norm)
            out = {"loss_rpn_cls": both[0], "loss_rpn_loc": both[1]}
            return {k: v * self.loss_weight.get(k, 1.0) for k, v in out.items()}
        pos = gl == 1
        self.storage["rpn/num_pos_anchors"] = pos.sum() / n
        self.storage["rpn/num_neg_anchors"] = (gl == 0).sum() / n
        # positives = the sampled foreground picks (known index list, image by image in pick order): no nonzero, no host sync
        img, a = torch.div(pos_g, A, rounding_mode="floor"), pos_g % A
        mbox = gt_cat[midx.view(-1)[pos_g] + gt_off[img]]          # an image without boxes has no positives, so no row of it is read
        gt_d = get_deltas(anchors[a], mbox, self.weights)
        loc = torch.abs(deltas.reshape(-1, 4)[pos_g] - gt_d).sum()
        # (weight = validity instead of ``logits[valid]``: a boolean-mask gather sizes its result on the host -- a device sync in
        # the middle of the step; ignored anchors (-1) contribute exactly 0 to the sum and to the gradient either way)
        valid = gl >= 0
        obj = F.binary_cross_entropy_with_logits(logits, gl.clamp(min=0).to(torch.float32), weight=valid.to(torch.float32), reduction="sum")
        norm = self.batch_size_per_image * n
        out = {"loss_rpn_cls": obj / norm, "loss_rpn_loc": loc / norm}
        return {k: v * self.loss_weight.get(k, 1.0) for k, v in out.items()}

    @torch.no_grad()
    def predict_proposals(self, logits, deltas, image_sizes, hf, wf, defer=False):
        """rpn.py:482-533 + find_top_rpn_proposals proposal_utils.py:22-130 for all images at once.
        ``defer=True`` returns a closure that takes the stage's one host sync and builds the Instances."""
        N, total = logits.shape
        training = self.training
        topk = min(total, self.pre_nms_topk[training])
        keys, order = hip.sort_desc(logits.detach().contiguous())
        img_hw = to_device_async(torch.tensor(image_sizes, dtype=torch.int32), logits.device)
        ag = self.anchor_generator
        boxes, valid = hip.rpn_decode(order, deltas.detach().contiguous(), ag.cell_anchors_0, img_hw, hf, wf, topk,
                                      float(ag.strides[0]), ag.offset, self.weights, SCALE_CLAMP, self.min_box_size)
        post = self.post_nms_topk[training]
        keep, nkeep = hip.nms(boxes, valid, self.nms_thresh, post)
        bad = (valid == 2).any() | ~torch.isfinite(keys[:, :topk]).all()
        rb = Readback(torch.cat([nkeep, bad.to(torch.int32).view(1)]))

        def finish():
            host = rb.get().tolist()                                         # the one host wait of this stage (an event, not a stream sync)
            if host[-1] and training:
                raise FloatingPointError("Predicted boxes or scores contain Inf/NaN. Training has diverged.")  # proposal_utils.py:100-105
            # gather all images' kept boxes / scores in two launches (the counts are on the host now)
            cnt = host[:N]
            total_k = sum(cnt)
            live = torch.arange(keep.shape[1], device=keep.device)[None, :] < nkeep[:, None]
            pos = torch.nonzero_static(live.view(-1), size=total_k)[:, 0]
            src = (keep.long() + torch.arange(N, device=keep.device)[:, None] * boxes.shape[1]).view(-1)[pos]
            kb, ks = boxes.view(-1, 4)[src], keys[:, : boxes.shape[1]].reshape(-1)[src]
            out = []
            for n, (b_, s_) in enumerate(zip(torch.split(kb, cnt), torch.split(ks, cnt))):
                inst = Instances(tuple(image_sizes[n]))
                inst.proposal_boxes = Boxes(b_)
                inst.objectness_logits = s_
                out.append(inst)
            return out

        return finish if defer else finish()

    def forward_nhwc(self, image_sizes, res4, gt_instances=None, defer_losses=False):
        """``defer_losses``: return ``(proposals, losses_fn)``; the caller runs ``losses_fn()`` after it has enqueued the
        box head, so the host half of the anchor sampling (16 x randperm(62 k) ~ 6 ms) runs under that device work instead
        of stalling the queue.  The anchor and proposal samplers own separate generators, so the draw order is unchanged."""
        N, hf, wf, _ = res4.shape
        logits, deltas = self.rpn_head.forward_nhwc(res4)
        lg = logits.reshape(N, -1)                # (N, Hi*Wi*A)   rpn.py:456-460
        dl = deltas.reshape(N, -1, 4)             # (N, Hi*Wi*A, 4) rpn.py:461-467 (NHWC already has (h,w,a,b) order)
        losses = {}
        # Order of enqueue: anchor matching (its counts readback issued) -> proposal stage (sort, decode, NMS; its keep-count
        # readback issued) -> only then the host halves.  Each readback waits on its own event, so while the host draws the
        # sampling permutations (~5 ms for 16 x 62 k anchors) the device works through the proposal stage.  Results are
        # unchanged: the two stages are independent (rpn.py:469-480 runs them in the other order).
        pending = None
        if self.training:
            assert gt_instances is not None, "RPN requires gt_instances in training!"
            anchors = self.anchor_generator.grid(hf, wf)
            pending = self.label_anchors_begin(anchors, gt_instances)
        finish = self.predict_proposals(lg, dl, image_sizes, hf, wf, defer=True)

        def losses_fn():
            if pending is None:
                return {}
            labels, matched = self.label_anchors_finish(*pending)
            return self.losses(anchors, lg, labels, dl, matched)

        if defer_losses:
            return finish(), losses_fn
        losses = losses_fn()
        return finish(), losses

    def forward(self, images, features, gt_instances=None):
        """rpn.py:431-480.  ``images`` needs ``.image_sizes``; features: dict of logical-NCHW maps."""
        f = features[self.in_features[0]]
        gts = None if gt_instances is None else [as_instances(g) for g in gt_instances]
        return self.forward_nhwc(images.image_sizes, to_nhwc(f), gts)


def build_proposal_generator(cfg, input_shape):
    name = cfg.MODEL.PROPOSAL_GENERATOR.NAME
    if name == "PrecomputedProposals":
        return None
    return PROPOSAL_GENERATOR_REGISTRY.get(name)(cfg, input_shape)

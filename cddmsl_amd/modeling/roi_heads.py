"""ROI heads on the HIP kernels -- detectron2/modeling/roi_heads/{roi_heads.py:123-319, clip_roi_heads.py:28-199,
fast_rcnn.py:100-127,368-689}, modeling/poolers.py:61-95,190-229.

``CLIPRes5ROIHeads`` borrows the backbone's ``layer4`` and ``attnpool`` at call time exactly like the reference
(rcnn.py:608-609): RoIAlign 14x14 (HIP) -> layer4 on the K regions as K 14x14 images (MFMA implicit GEMM) ->
query-0 attention pool (HIP) -> cosine-logit text-embedding classifier (HIP, fp32) + bbox_pred.
"""
import math
import os
from typing import Dict, List

import torch
import torch.nn.functional as F
from torch import nn

from .. import hip, layers
from .._lib import to_device_async
from ..registry import ROI_HEADS_REGISTRY
from ..structures import Boxes, Instances, ShapeSpec, as_instances
from .backbone import to_nhwc, to_nchw
from .rpn import apply_deltas, get_deltas, subsample_begin, subsample_finish, subsample_labels_batched

GT_LOGIT = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))  # proposal_utils.py:183


class ROIPooler(nn.Module):
    """poolers.py:98-250, single level, ROIAlignV2 (aligned=True)."""

    def __init__(self, output_size, scales, sampling_ratio, pooler_type="ROIAlignV2"):
        super().__init__()
        assert len(scales) == 1 and pooler_type == "ROIAlignV2"
        self.output_size, self.scale, self.sampling_ratio = output_size, scales[0], sampling_ratio

    def rois_of(self, box_lists: List[Boxes], dev):
        """convert_boxes_to_pooler_format poolers.py:68-95: rois grouped by image, (batch_idx, x0, y0, x1, y1); + per-image offsets"""
        counts = [len(b) for b in box_lists]
        bidx = to_device_async(torch.repeat_interleave(torch.arange(len(counts), dtype=torch.float32), torch.tensor(counts)), dev)
        rois = torch.cat([bidx[:, None], torch.cat([b.tensor.float() for b in box_lists])], dim=1).contiguous()
        start = to_device_async(torch.tensor([0] + list(torch.tensor(counts).cumsum(0).tolist()), dtype=torch.int32), dev)
        return rois, start

    def forward_nhwc(self, feat, box_lists: List[Boxes], with_pooled=False, extra_maps=None):
        """``with_pooled``: the crops carry their 2x2-average-pooled copy for the stride-2 stage that follows (layers.roi_align);
        ``extra_maps``: maps of the crops' geometry appended behind them"""
        rois, start = self.rois_of(box_lists, feat.device)
        return layers.roi_align(feat, rois, start, self.output_size, self.scale, self.sampling_ratio, True, with_pooled=with_pooled,
                                extra=extra_maps)

    def forward(self, x, box_lists):
        return to_nchw(self.forward_nhwc(to_nhwc(x[0]), box_lists))


class _Linear(nn.Module):
    def __init__(self, i, o, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i))
        self.bias = nn.Parameter(torch.zeros(o)) if bias else None
        self._pw = None

    def pw(self):
        if self._pw is None or self._pw.param is not self.weight:
            self._pw = layers.PreparedWeight(self.weight, None, frozen=not self.weight.requires_grad)
        return self._pw


# RoI head entry: evaluate layer4.0's conv1 on the feature map before the pooling (layers.RoIStageFn).  CDDMSL_ROI_COMMUTE=0 keeps
# the literal order (pooler, then layer4 on the crops) -- the two are compared in tests/test_gpu_ops.py.
def commute_roi_conv1():
    return os.environ.get("CDDMSL_ROI_COMMUTE", "1") != "0"

NMS_MAX_CANDIDATES = 12288      # cddmsl_nms handles 192 mask words of 64 boxes per image


def _nms_sorted(boxes, scores, iou_threshold):
    """torchvision.ops.nms on the HIP kernels: kept indices in descending-score order.  More candidates than the kernel's
    mask holds is an error (the C-ABI returns CDDMSL_ERR_ARG for it) -- never a silent truncation."""
    n = boxes.shape[0]
    if n > NMS_MAX_CANDIDATES:
        raise ValueError(f"nms: {n} candidates in one group exceed the kernel's limit of {NMS_MAX_CANDIDATES}")
    keys, order = hip.sort_desc(scores.float().contiguous().view(1, n))
    order = order[0].long()
    sorted_boxes = boxes[order].view(1, n, 4).contiguous()
    keep, nkeep = hip.nms(sorted_boxes, torch.ones((1, n), dtype=torch.uint8, device=boxes.device), float(iou_threshold), n)
    return order[keep[0, : int(nkeep[0])].long()]


def batched_nms(boxes, scores, idxs, iou_threshold):
    """layers/nms.py:19-39 -> torchvision.ops.batched_nms: per-category NMS, on the HIP sort + bit-mask NMS kernels.  Returns
    kept indices in descending-score order.  Up to 12 288 candidates: ONE pass through the coordinate-offset trick
    (torchvision's ``_batched_nms_coordinate_trick``); beyond that -- more than the kernel's mask words hold -- one pass per
    category (torchvision's ``_batched_nms_vanilla``, which it also switches to for large inputs); a single category with more
    than 12 288 candidates raises."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty(0, dtype=torch.int64, device=boxes.device)
    boxes = boxes.float()
    if n <= NMS_MAX_CANDIDATES:
        off = idxs.to(boxes) * (boxes.max() + 1.0)
        return _nms_sorted(boxes + off[:, None], scores, iou_threshold)
    keep_mask = torch.zeros(n, dtype=torch.bool, device=boxes.device)
    for c in torch.unique(idxs).tolist():
        sel = torch.nonzero(idxs == c).squeeze(1)
        keep_mask[sel[_nms_sorted(boxes[sel], scores[sel], iou_threshold)]] = True
    ki = torch.nonzero(keep_mask).squeeze(1)
    return ki[scores[ki].sort(descending=True, stable=True)[1]]


def fast_rcnn_inference_single_image(boxes, scores, image_shape, score_thresh, nms_thresh, topk_per_image):
    """fast_rcnn.py:130-209 (hard NMS): drop non-finite rows, clip, score threshold per (proposal, class), per-class NMS,
    keep the top-k.  boxes [R, 4K], scores [R, K+1] -> (Instances, indices of the kept proposals)."""
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
    if not bool(valid.all()):
        boxes, scores = boxes[valid], scores[valid]
    scores = scores[:, :-1]
    k = boxes.shape[1] // 4
    b = Boxes(boxes.reshape(-1, 4))
    b.clip(image_shape)
    boxes = b.tensor.view(-1, k, 4)
    mask = scores > score_thresh
    inds = mask.nonzero()                            # [R', 2] (proposal, class), row-major order as the reference
    boxes = boxes[inds[:, 0], 0] if k == 1 else boxes[mask]
    scores = scores[mask]
    keep = batched_nms(boxes, scores, inds[:, 1], nms_thresh)
    if topk_per_image >= 0:
        keep = keep[:topk_per_image]
    res = Instances(tuple(image_shape))
    res.pred_boxes, res.scores, res.pred_classes = Boxes(boxes[keep]), scores[keep], inds[keep, 1]
    return res, inds[keep, 0]


class FastRCNNOutputLayers(nn.Module):
    """fast_rcnn.py:368-689, RegionCLIP text-embedding classifier branch."""

    def __init__(self, cfg, input_shape):
        super().__init__()
        c = cfg.MODEL.CLIP
        self.num_classes = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        self.use_clip_cls_emb = bool(c.USE_TEXT_EMB_CLASSIFIER)
        if self.use_clip_cls_emb:       # CLIP text embeddings as the classifier (fast_rcnn.py:440-475)
            d = c.TEXT_EMB_DIM
            self.temperature = c.CLSS_TEMP
            self.cls_score = _Linear(d, self.num_classes, bias=False)
            self.cls_bg_score = _Linear(d, 1, bias=False)
            nn.init.normal_(self.cls_score.weight, std=0.01)
            nn.init.constant_(self.cls_bg_score.weight, 0)
            if c.TEXT_EMB_PATH:
                self.cls_score.weight.data.copy_(torch.load(c.TEXT_EMB_PATH, map_location="cpu", weights_only=True))
            self.cls_score.weight.requires_grad = False       # frozen embeddings fast_rcnn.py:453
            self.cls_bg_score.weight.requires_grad = False    # zero background embedding :458-463
        else:                           # regular linear classifier (fast_rcnn.py:476-479), stock R50-C4
            d = input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
            self.cls_score = _Linear(d, self.num_classes + 1)
            nn.init.normal_(self.cls_score.weight, std=0.01)
        self.bbox_pred = _Linear(d, self.num_classes * 4)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        self.box_weights = tuple(cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS)
        assert cfg.MODEL.ROI_BOX_HEAD.SMOOTH_L1_BETA == 0.0 and not cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG
        self.bg_cls_loss_weight = c.BG_CLS_LOSS_WEIGHT
        self.focal_scaled_loss = c.FOCAL_SCALED_LOSS
        self.loss_weight = {"loss_box_reg": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_WEIGHT}
        # inference (fast_rcnn.py:398-407,432-437)
        self.test_score_thresh = cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST
        self.test_nms_thresh = cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST
        self.test_topk_per_image = cfg.TEST.DETECTIONS_PER_IMAGE
        self.no_box_delta = bool(c.NO_BOX_DELTA)
        self.multiply_rpn_score = bool(c.MULTIPLY_RPN_SCORE)
        assert not cfg.MODEL.ROI_HEADS.get("SOFT_NMS_ENABLED", False), "soft-NMS is off the hot path (defaults.py:399)"
        self.compute_dtype = {"bf16": torch.bfloat16, "f32": torch.float32, "fp8": torch.bfloat16}[cfg.MODEL.get("COMPUTE_DTYPE", "bf16")]
        self.fp8 = cfg.MODEL.get("COMPUTE_DTYPE", "bf16") == "fp8"
        self._wn = None
        self.storage = {}

    def _text_emb(self):
        w = self.cls_score.weight
        if self._wn is None or self._wn[0] != (w.device, w._version):
            assert float(self.cls_bg_score.weight.abs().max()) == 0.0, "background embedding must stay zero (fast_rcnn.py:460)"
            self._wn = ((w.device, w._version), F.normalize(w.detach().float(), p=2.0, dim=1).contiguous())
        return self._wn[1]

    def forward(self, x):
        """x [R, 1024] f32 -> (scores [R, K+1] f32, deltas [R, 4K] f32)   fast_rcnn.py:529-572"""
        xt = x.to(self.compute_dtype)
        if self.use_clip_cls_emb:
            wn = self._text_emb()
            wn8 = None
            if self.fp8:        # e4m3 copy of the unit-norm class embeddings, kept on the module next to the normalised ones (same key)
                if getattr(self, "_wn8", None) is None or self._wn8[0] is not wn:
                    self._wn8 = (wn, layers.fp8_unit_rows(wn))
                wn8 = self._wn8[1]
            scores = layers.cosine_logits(x, wn, self.temperature, fp8=self.fp8, wn8=wn8)
            deltas = layers.linear(xt, self.bbox_pred.pw(), self.bbox_pred.bias, out_f32=True)
            return scores, deltas
        # plain classifier: cls_score (K+1) and bbox_pred (4K) as ONE padded GEMM (row widths must be whole 16-B chunks)
        k1 = self.num_classes + 1
        y = layers.fused_heads(xt.contiguous().view(1, 1, xt.shape[0], xt.shape[1]),
                               [(self.cls_score.weight, self.cls_score.bias), (self.bbox_pred.weight, self.bbox_pred.bias)])
        y = y.view(xt.shape[0], -1)
        return y[:, :k1].contiguous(), y[:, k1:k1 + 4 * self.num_classes].contiguous()

    def losses(self, predictions, proposals):
        """fast_rcnn.py:574-689"""
        scores, deltas = predictions
        gt_classes = torch.cat([p.gt_classes for p in proposals], dim=0)
        pboxes = torch.cat([p.proposal_boxes.tensor for p in proposals], dim=0)
        gboxes = torch.cat([(p.gt_boxes if p.has("gt_boxes") else p.proposal_boxes).tensor for p in proposals], dim=0)
        self._log_stats(scores.detach(), gt_classes)
        loss_cls = layers.focal_cross_entropy(scores, gt_classes, self.focal_scaled_loss, self.num_classes, self.bg_cls_loss_weight)
        if all(getattr(p, "_num_fg", None) is not None for p in proposals):
            # sampled proposals list their foreground picks first (roi_heads.py:211-234): the index list is known on the
            # host, no nonzero / device sync (fast_rcnn.py:653 takes nonzero of the same mask)
            offs, idx = 0, []
            for p in proposals:
                idx.append(torch.arange(offs, offs + p._num_fg))
                offs += len(p)
            fg = to_device_async(torch.cat(idx), gt_classes.device)
        else:
            fg = torch.nonzero((gt_classes >= 0) & (gt_classes < self.num_classes), as_tuple=True)[0]
        if deltas.is_cuda and deltas.dtype == torch.float32 and deltas.shape[1] == 4 * self.num_classes:
            # one kernel per direction over the foreground index list (class-specific columns picked inside)
            loss_box = layers.box_l1(deltas, fg, gt_classes, pboxes.float(), gboxes.float(), self.box_weights, 1.0 / max(gt_classes.numel(), 1.0))
        else:
            fg_pred = deltas.view(-1, self.num_classes, 4)[fg, gt_classes[fg]]
            gt_d = get_deltas(pboxes[fg], gboxes[fg], self.box_weights)
            loss_box = torch.abs(fg_pred - gt_d).sum() / max(gt_classes.numel(), 1.0)
        out = {"loss_cls": loss_cls, "loss_box_reg": loss_box}
        return {k: v * self.loss_weight.get(k, 1.0) for k, v in out.items()}

    # ------------------------------------------------------------------ inference (fast_rcnn.py:47-209,691-811)
    def predict_boxes(self, predictions, proposals):
        """fast_rcnn.py:760-790 -> per image [Ri, 4K] class-specific boxes"""
        if not len(proposals):
            return []
        _, deltas = predictions
        pboxes = torch.cat([p.proposal_boxes.tensor for p in proposals], dim=0)
        boxes = pboxes if self.no_box_delta else apply_deltas(deltas, pboxes, self.box_weights)
        return boxes.split([len(p) for p in proposals])

    def predict_probs(self, predictions, proposals):
        """fast_rcnn.py:792-811: softmax over the K+1 logits"""
        scores, _ = predictions
        return F.softmax(scores.float(), dim=-1).split([len(p) for p in proposals], dim=0)

    def inference(self, predictions, proposals):
        """fast_rcnn.py:691-724 -> (list[Instances(pred_boxes, scores, pred_classes)], list[kept proposal indices])"""
        boxes = self.predict_boxes(predictions, proposals)
        scores = self.predict_probs(predictions, proposals)
        if self.multiply_rpn_score and not self.training:   # geometric mean with the RPN objectness (fast_rcnn.py:708-710)
            scores = [(s * p.objectness_logits[:, None]) ** 0.5 for s, p in zip(scores, proposals)]
        out = [fast_rcnn_inference_single_image(b, s, p.image_size, self.test_score_thresh, self.test_nms_thresh,
                                                self.test_topk_per_image) for b, s, p in zip(boxes, scores, proposals)]
        return [o[0] for o in out], [o[1] for o in out]

    def _log_stats(self, scores, gt_classes):
        """_log_classification_stats fast_rcnn.py:100-127 (kept on device; no sync)."""
        n = gt_classes.numel()
        if n == 0:
            return
        pred = scores.argmax(dim=1)
        bg = scores.shape[1] - 1
        fg = (gt_classes >= 0) & (gt_classes < bg)
        nfg = fg.sum().clamp(min=1)
        self.storage["fast_rcnn/cls_accuracy"] = (pred == gt_classes).sum() / n
        self.storage["fast_rcnn/fg_cls_accuracy"] = ((pred == gt_classes) & fg).sum() / nfg
        self.storage["fast_rcnn/false_negative"] = ((pred == bg) & fg).sum() / nfg


@ROI_HEADS_REGISTRY.register()
class CLIPRes5ROIHeads(nn.Module):
    def __init__(self, cfg, input_shape: Dict[str, ShapeSpec]):
        super().__init__()
        r = cfg.MODEL.ROI_HEADS
        self.num_classes, self.in_features = r.NUM_CLASSES, r.IN_FEATURES
        self.batch_size_per_image, self.positive_fraction = r.BATCH_SIZE_PER_IMAGE, r.POSITIVE_FRACTION
        self.iou_thresholds, self.iou_labels = list(r.IOU_THRESHOLDS), list(r.IOU_LABELS)
        self.proposal_append_gt = r.PROPOSAL_APPEND_GT
        assert len(self.in_features) == 1 and not cfg.MODEL.MASK_ON and not cfg.MODEL.CLIP.ONLY_SAMPLE_FG_PROPOSALS
        b = cfg.MODEL.ROI_BOX_HEAD
        self.pooler = ROIPooler(b.POOLER_RESOLUTION, (1.0 / input_shape[self.in_features[0]].stride,),
                                b.POOLER_SAMPLING_RATIO, b.POOLER_TYPE)
        out_channels = cfg.MODEL.RESNETS.RES2_OUT_CHANNELS * 8
        self.box_predictor = FastRCNNOutputLayers(cfg, ShapeSpec(channels=out_channels, height=1, width=1))   # clip_roi_heads.py:104-107
        self.sample_generator = torch.Generator()
        self.storage = {}

    @torch.no_grad()
    def label_and_sample_proposals(self, proposals: List[Instances], targets: List[Instances]):
        """roi_heads.py:236-319 (+ add_ground_truth_to_proposals proposal_utils.py:133-200, _sample_proposals :184-234), all
        images at once: one concatenated box / logit / class vector, per-image matching written into its slices, one
        sampling pass, one gather per field -- ~40 small launches instead of ~12 per image."""
        dev = proposals[0].proposal_boxes.tensor.device
        gtbs = [t.gt_boxes.tensor.float().contiguous() for t in targets]
        gtcs = [t.gt_classes for t in targets]
        parts_b, parts_l, counts = [], [], []
        for prop, gtb in zip(proposals, gtbs):
            parts_b.append(prop.proposal_boxes.tensor)
            parts_l.append(prop.objectness_logits)
            if self.proposal_append_gt:
                parts_b.append(gtb)
                parts_l.append(torch.full((len(gtb),), GT_LOGIT, device=dev))
            counts.append(len(prop) + (len(gtb) if self.proposal_append_gt else 0))
        boxes_all = torch.cat(parts_b).contiguous()
        logits_all = torch.cat(parts_l)
        total = sum(counts)
        midx_all, mlab_all = hip.iou_match_batched(gtbs, boxes_all, counts, self.iou_thresholds, self.iou_labels, False)
        ngt = [len(g) for g in gtbs]
        has_gt = sum(ngt) > 0
        gt_off_rows = to_device_async(torch.repeat_interleave(torch.tensor([0] + ngt).cumsum(0)[:-1], torch.tensor(counts)), dev)
        gidx_all = midx_all + gt_off_rows                      # row of the matched box in the concatenated ground truth
        if has_gt:
            gtc_cat, gtb_cat = torch.cat(gtcs), torch.cat(gtbs)
            cls_all = gtc_cat[gidx_all.clamp(max=gtc_cat.numel() - 1)]
            cls_all[mlab_all == 0] = self.num_classes
            cls_all[mlab_all == -1] = -1
            if min(ngt) == 0:                                  # images without boxes: every proposal is background (roi_heads.py:208-209)
                nog = to_device_async(torch.repeat_interleave(torch.tensor([g == 0 for g in ngt]), torch.tensor(counts)), dev)
                cls_all[nog] = self.num_classes
        else:
            cls_all = torch.full((total,), self.num_classes, dtype=torch.int64, device=dev)
        st = subsample_begin(cls_all, self.num_classes, lens=counts)
        st["global_only"] = True
        subsample_finish(st, self.batch_size_per_image, self.positive_fraction, self.sample_generator)
        fg_g, bg_g, nfg, nbg = st["global"]
        fgs, bgs = torch.split(fg_g, nfg), torch.split(bg_g, nbg)
        sidx = torch.cat([t for pair in zip(fgs, bgs) for t in pair])          # per image: foreground picks, then background
        b_s, l_s, c_s = boxes_all[sidx], logits_all[sidx], cls_all[sidx]
        g_s = gtb_cat[gidx_all[sidx].clamp(max=gtb_cat.shape[0] - 1)] if has_gt else None
        per = [f + g for f, g in zip(nfg, nbg)]
        out = []
        for i, (prop, pb, pl, pc) in enumerate(zip(proposals, torch.split(b_s, per), torch.split(l_s, per), torch.split(c_s, per))):
            inst = Instances(prop.image_size)
            inst.proposal_boxes, inst.objectness_logits, inst.gt_classes = Boxes(pb), pl, pc
            if ngt[i] > 0:
                inst.gt_boxes = Boxes(torch.split(g_s, per)[i])
            inst._num_fg = int(nfg[i])          # host-side: the first _num_fg rows are the foreground samples
            out.append(inst)
        self.storage["roi_head/num_fg_samples"] = sum(nfg) / max(len(nfg), 1)
        self.storage["roi_head/num_bg_samples"] = sum(nbg) / max(len(nbg), 1)
        return out

    def _pooled_embeddings(self, feat_nhwc, boxes, res5, attnpool, extra_maps=None):
        """RoIAlign -> layer4 -> attention pool (clip_roi_heads.py:160-165); with the package's own modules the last two run
        as one composition (the stage's ReLU backward rides in the pool's backward).  ``extra_maps`` [E, 14, 14, C]: res4 maps
        that go through the same layer4 + attention pool (the image-level consistency branch's 224x224 crops, rcnn.py:255-262):
        they ride behind the RoI crops, the result has their embeddings in rows K.."""
        from .backbone import AttentionPool2d, ResStage
        fused = isinstance(res5, ResStage) and isinstance(attnpool, AttentionPool2d)
        if fused and self.pooler.output_size % 2 == 0 and layers.roi_stage_supported(res5.block_params()) and commute_roi_conv1():
            # pooler + layer4 + attention pool as one composition, the first block's conv1 evaluated on the feature map BEFORE
            # the pooling (layers.RoIStageFn): the [K,14,14,1024] crop tensor is never formed
            rois, start = self.pooler.rois_of(boxes, feat_nhwc.device)
            return layers.roi_stage_attnpool(feat_nhwc, rois, start, res5.block_params(), res5[0].frozen, attnpool._params(),
                                             self.pooler.output_size, self.pooler.scale, self.pooler.sampling_ratio, extra_maps)
        # (CLIP's layer4 is stride 2 with an AvgPool2d on the downsample path: RoIAlign hands it the pooled crops as well)
        x = self.pooler.forward_nhwc(feat_nhwc, boxes, with_pooled=fused and res5[0].stride > 1, extra_maps=extra_maps)
        if fused:
            return res5.forward_nhwc(x, then_attnpool=attnpool)
        return attnpool(to_nchw(res5.forward_nhwc(x)))

    def forward_get_features(self, features_src, features_trgt, proposals, targets=None, res5=None, attnpool=None):
        """clip_roi_heads.py:117-132: the same boxes pooled from the source and the target map."""
        if self.training:
            assert targets
        boxes = [p.proposal_boxes for p in proposals]
        return (self._pooled_embeddings(to_nhwc(features_src[self.in_features[0]]), boxes, res5, attnpool),
                self._pooled_embeddings(to_nhwc(features_trgt[self.in_features[0]]), boxes, res5, attnpool))

    def forward_get_features_paired(self, feat_cat_nhwc, num_images, proposals, res5, attnpool, extra_maps=None):
        """Same result as ``forward_get_features`` when source and target maps are stacked along the batch axis
        (images [0,B) = source, [B,2B) = target): ONE RoIAlign / layer4 / attention-pool pass over 2K regions.
        With ``extra_maps`` a third result: their embeddings (``_pooled_embeddings``)."""
        boxes = [p.proposal_boxes for p in proposals]
        att = self._pooled_embeddings(feat_cat_nhwc, boxes + boxes, res5, attnpool, extra_maps)
        k = sum(len(b) for b in boxes)
        if extra_maps is not None:
            return att[:k], att[k:2 * k], att[2 * k:]
        return att[:k], att[k:]

    def forward(self, images, features, proposals, targets=None, res5=None, attnpool=None):
        """clip_roi_heads.py:134-175"""
        assert attnpool is not None, "CLIPRes5ROIHeads is used with the backbone's attention pool (rcnn.py:606-612)"
        if self.training:
            assert targets
            targets = [as_instances(t) for t in targets]
            proposals = self.label_and_sample_proposals(proposals, targets)
        att = self._pooled_embeddings(to_nhwc(features[self.in_features[0]]), [p.proposal_boxes for p in proposals], res5, attnpool)
        predictions = self.box_predictor(att)
        if self.training:
            return [], self.box_predictor.losses(predictions, proposals)
        pred_instances, _ = self.box_predictor.inference(predictions, proposals)     # clip_roi_heads.py:171-174
        return self.forward_with_given_boxes(features, pred_instances, res5), {}

    def forward_with_given_boxes(self, features, instances, res5=None):
        """clip_roi_heads.py:176-199 with MASK_ON False: nothing to add"""
        assert not self.training
        assert instances[0].has("pred_boxes") and instances[0].has("pred_classes")
        return instances


@ROI_HEADS_REGISTRY.register()
class Res5ROIHeads(CLIPRes5ROIHeads):
    """Stock C4 head (roi_heads.py:358-512): RoIAlign -> own ``res5`` stage (stride 2) -> mean pool -> linear classifier."""

    def __init__(self, cfg, input_shape):
        super().__init__(cfg, input_shape)
        from .resnet import make_stage
        r = cfg.MODEL.RESNETS
        out_channels = r.RES2_OUT_CHANNELS * 8
        self.res5 = make_stage(3, [2, 1, 1], out_channels // 2, r.get("WIDTH_PER_GROUP", 64) * 8, out_channels)   # _build_res5_block :440-463

    def forward(self, images, features, proposals, targets=None, res5=None, attnpool=None):
        if self.training:
            assert targets
            targets = [as_instances(t) for t in targets]
            proposals = self.label_and_sample_proposals(proposals, targets)
        x = self.pooler.forward_nhwc(to_nhwc(features[self.in_features[0]]), [p.proposal_boxes for p in proposals])
        feats = layers.mean_pool(self.res5.forward_nhwc(x))
        predictions = self.box_predictor(feats)
        if self.training:
            return [], self.box_predictor.losses(predictions, proposals)
        pred_instances, _ = self.box_predictor.inference(predictions, proposals)     # roi_heads.py:497-500
        return self.forward_with_given_boxes(features, pred_instances), {}


def build_roi_heads(cfg, input_shape):
    return ROI_HEADS_REGISTRY.get(cfg.MODEL.ROI_HEADS.NAME)(cfg, input_shape)

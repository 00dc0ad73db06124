"""``GeneralizedRCNN`` (CDDMSL variant) on the HIP hot path -- detectron2/modeling/meta_arch/rcnn.py:37-623,758-768,
meta_arch/build.py:16-25, backbone/clipcap/gather.py:5-20.

Same call contract as the reference: ``model(batched_inputs, clipcap_model=None, branch='supervised',
KD_regularization=True)`` returning the loss dict (supervised: loss_cls, loss_box_reg, loss_rpn_cls, loss_rpn_loc;
'caption_consistency': cont_loss[, kd_loss]; 'caption_consistency_regionLevel': a bare tensor).
Intended semantics are kept where the reference as shipped cannot run (SURVEY.md warnings): world_size 1 needs no
process group; the unused full-image layer4 of the supervised / region-level branches is not computed.
"""
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
from torch import nn

from .. import hip, layers
from .._lib import to_device_async
from ..registry import META_ARCH_REGISTRY
from ..structures import Boxes, ImageList, Instances, as_instances
from . import resnet  # noqa: F401  (registers build_resnet_backbone)
from .backbone import build_backbone, to_nchw
from .clipcap import v2l
from .postprocessing import detector_postprocess
from .roi_heads import build_roi_heads
from .rpn import build_proposal_generator


class GatherLayer(torch.autograd.Function):
    """gather.py:5-20: all_gather forward; backward keeps this rank's slice only (no reduction)."""

    @staticmethod
    def forward(ctx, input):
        ctx.save_for_backward(input)
        output = [torch.zeros_like(input) for _ in range(dist.get_world_size())]
        dist.all_gather(output, input.contiguous())
        return tuple(output)

    @staticmethod
    def backward(ctx, *grads):
        (input,) = ctx.saved_tensors
        grad_out = torch.zeros_like(input)
        grad_out[:] = grads[dist.get_rank()]
        return grad_out


def gather_cat(x):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return torch.cat(GatherLayer.apply(x), dim=0)
    return x


class _Linear(nn.Linear):
    def pw(self):
        if getattr(self, "_pw", None) is None or self._pw.param is not self.weight:
            self._pw = layers.PreparedWeight(self.weight, None, frozen=False)
        return self._pw


@META_ARCH_REGISTRY.register()
class GeneralizedRCNN(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.backbone = build_backbone(cfg)
        self.offline_backbone = build_backbone(cfg)
        for p in self.offline_backbone.parameters():
            p.requires_grad = False
        for blk in self.offline_backbone.modules():
            if hasattr(blk, "frozen"):
                blk.frozen = True
        self.offline_backbone.eval()
        self.proposal_generator = build_proposal_generator(cfg, self.backbone.output_shape())
        self.roi_heads = build_roi_heads(cfg, self.backbone.output_shape())
        self.input_format = cfg.INPUT.FORMAT
        self.pixel_mean_list, self.pixel_std_list = list(cfg.MODEL.PIXEL_MEAN), list(cfg.MODEL.PIXEL_STD)
        self.register_buffer("pixel_mean", torch.tensor(self.pixel_mean_list).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(self.pixel_std_list).view(-1, 1, 1), False)
        self.div_pixel = sum(self.pixel_mean_list) < 3.0      # CLIP models take RGB/255 inputs (rcnn.py:87-91)
        assert not self.div_pixel or self.input_format == "RGB"
        self.use_clip_c4 = cfg.MODEL.BACKBONE.NAME == "build_clip_resnet_backbone"
        self.use_clip_attpool = cfg.MODEL.ROI_HEADS.NAME == "CLIPRes5ROIHeads" and cfg.MODEL.CLIP.USE_TEXT_EMB_CLASSIFIER
        self.projector = nn.Sequential(_Linear(768, 768), nn.ReLU(), _Linear(768, 256))   # rcnn.py:95-99
        self.compute_dtype = self.backbone.compute_dtype
        self.regions_per_image = 16                                                        # rcnn.py:437
        self.region_generator = torch.Generator()
        # Within one training step the reference evaluates backbone(res1-4) + RPN on the SAME source images twice, once in
        # the supervised forward and once in the region-level branch (rcnn.py:424-425,434 vs :597-599; identical
        # preprocessing :201 vs :764).  Both evaluations give identical tensors, so the second is elided: the region-level
        # branch reuses the supervised pass's res4 (gradients of both branches then flow through the one graph -- the same sum)
        # and its RPN proposals, and still consumes the second RPN pass's random draws.  Valid only for the same batch
        # object and the same weights (optimizer step counter), and only when all forwards of a step precede its one backward
        # (the reference's run_step) -- so it is off on a bare model and switched on by SimpleTrainer, which guarantees that
        # order (CDDMSL_SHARE_SOURCE_PASS=0 keeps it off there too).
        self.share_source_pass = False
        self._shared = None
        self.defer_rpn_losses = True

    @property
    def device(self):
        return self.pixel_mean.device

    # ------------------------------------------------------------------ preprocessing (fused normalise + pad, NHWC)
    def _images(self, batched_inputs, key):
        return [x[key].to(self.device).contiguous() for x in batched_inputs]

    def preprocess_image(self, batched_inputs, key="image"):
        """rcnn.py:758-768 / :196-207 -> (NHWC padded tensor in the compute dtype, image_sizes)"""
        imgs = self._images(batched_inputs, key)
        sizes = [tuple(i.shape[-2:]) for i in imgs]
        Hp, Wp = max(s[0] for s in sizes), max(s[1] for s in sizes)
        return hip.preprocess(imgs, Hp, Wp, self.pixel_mean_list, self.pixel_std_list, self.compute_dtype, div255=self.div_pixel), sizes

    def preprocess_image_train(self, batched_inputs):
        """rcnn.py:161-179 -> NHWC [2N,224,224,Cp]: rows [0,N) = source images, [N,2N) = target images.
        (The reference pads source and target batches separately; every sample pair shares one size, so the padded
        sizes coincide -- asserted.)"""
        src, tgt = self._images(batched_inputs, "image"), self._images(batched_inputs, "image_trgt")
        Hp, Wp = max(i.shape[-2] for i in src), max(i.shape[-1] for i in src)
        assert (Hp, Wp) == (max(i.shape[-2] for i in tgt), max(i.shape[-1] for i in tgt))
        return hip.preprocess224(src + tgt, Hp, Wp, self.pixel_mean_list, self.pixel_std_list, self.compute_dtype)

    # ------------------------------------------------------------------ pieces
    def project(self, x):
        """projector MLP 768 -> 768 -> ReLU -> 256 (f32 in/out, GEMMs in the compute dtype)"""
        T = self.compute_dtype
        l0, l2 = self.projector[0], self.projector[2]
        h = layers.linear(x.to(T), l0.pw(), l0.bias, relu=True, out_f32=True)
        return layers.linear(h.to(T), l2.pw(), l2.bias, out_f32=True)

    def _encode(self, bb, img_nhwc):
        feats = bb.forward_nhwc(img_nhwc)
        return bb.attnpool(to_nchw(feats["res5"]))

    def v2l_contrastive(self, images_both, clipcap_model, KD_regularization=True):
        """rcnn.py:255-319.  ``images_both`` = [source; target] stacked on the batch axis: the student runs ONCE over 2N
        images (per-sample results are identical to two passes; half the launches)."""
        n = images_both.shape[0] // 2
        f = v2l(self._encode(self.backbone, images_both), clipcap_model)     # [2N, 768]
        fs, ft = f[:n], f[n:]
        kd_loss = None
        if KD_regularization:
            with torch.no_grad():
                teacher = v2l(self._encode(self.offline_backbone, images_both[:n]), clipcap_model)
            kd_loss = torch.nn.functional.l1_loss(teacher.detach(), fs)
        p = self.project(f)
        fs, ft = gather_cat(p[:n].contiguous()), gather_cat(p[n:].contiguous())
        return layers.contrastive_loss(ft, fs), kd_loss

    def _region_level_encode(self, batched_inputs, extra_maps=None):
        """rcnn.py:422-470 up to the pooled region embeddings: (source regions, target regions), each [16 N, 1024]; with
        ``extra_maps`` (res4 maps of the RoI crops' geometry) a third entry: their embeddings from the same layer4 + pool pass"""
        # source and target images stacked on the batch axis: one backbone pass over 2N images, one RoI pass over
        # 2x16N regions (identical per-sample results, half the kernel launches)
        n = len(batched_inputs)
        shared, self._shared = self._shared, None
        if shared is not None and not (shared["inputs"] is batched_inputs and shared["step"] == layers._STEP[0]):
            shared = None
        imgs = ([] if shared else self._images(batched_inputs, "image")) + self._images(batched_inputs, "image_trgt")
        sizes = [tuple(i.shape[-2:]) for i in self._images(batched_inputs, "image")] if shared is None else shared["sizes"]
        assert sizes == [tuple(i.shape[-2:]) for i in imgs[-n:]], "a sample and its domain twin share one geometry"
        Hp, Wp = max(s_[0] for s_ in sizes), max(s_[1] for s_ in sizes)
        x = hip.preprocess(imgs, Hp, Wp, self.pixel_mean_list, self.pixel_std_list, self.compute_dtype)   # always /255: rcnn.py:201
        spec = shared.get("res4_spec") if shared else None      # the supervised pass left room behind its res4 for the target images'
        f = self.backbone.forward_nhwc(x, want_res5=False, res4_spec=spec)["res4"]
        gts = [as_instances(x_["instances"]).to(self.device) for x_ in batched_inputs]
        with torch.no_grad():
            if shared is None:
                props, _ = self.proposal_generator.forward_nhwc(sizes, f[:n].detach(), gts)
            else:
                if spec is not None and spec.used == 2:
                    f = layers.stack_halves(shared["res4"], f, spec.full)      # already adjacent: no copy
                else:
                    f = torch.cat([shared["res4"], f])
                self.proposal_generator.replay_sampling_draws(shared["counts"])
                props = shared["proposals"]
            sel_cpu = [torch.randperm(len(p), generator=self.region_generator)[: self.regions_per_image] for p in props]
            # one pinned, non-blocking H2D of all picks (as rows of the concatenated proposal boxes) and ONE gather
            offs = torch.tensor([0] + [len(p) for p in props]).cumsum(0)
            pick = to_device_async(torch.cat([s_ + int(o) for s_, o in zip(sel_cpu, offs[:-1])]), self.device)
            picked = torch.cat([p.proposal_boxes.tensor for p in props])[pick]
            props = []
            for size, b_ in zip(sizes, torch.split(picked, [len(s_) for s_ in sel_cpu])):
                inst = Instances(tuple(size))
                inst.proposal_boxes = Boxes(b_)
                props.append(inst)
        return self.roi_heads.forward_get_features_paired(f, n, props, self.backbone.layer4, self.backbone.attnpool, extra_maps)

    def forward_consistency(self, batched_inputs, clipcap_model, KD_regularization=True):
        """Both caption-consistency branches (rcnn.py:413-470) with ONE pass through the frozen mapper and ONE through the
        projector: image embeddings [source; target] (2N rows), region embeddings (2 x 16N rows) and the teacher's source
        embeddings (N rows, no gradient) are stacked on the batch axis.  Every mapper / projector operation is per row (or
        per sequence), so each row's result is what its own branch would have computed; the image-level branch's ~150
        launches over 2560-row operands disappear into the region-level ones.  Returns the three loss entries."""
        both = self.preprocess_image_train(batched_inputs)
        n = both.shape[0] // 2
        enc_teacher = None
        if KD_regularization:
            with torch.no_grad():
                enc_teacher = self._encode(self.offline_backbone, both[:n])
        # The student's layer4 + attention pool ARE the RoI head's (rcnn.py:606-612 borrows them): the 2N res4 maps of the
        # 224x224 crops (14x14, the RoI crops' geometry) ride behind the region-level branch's RoI crops through one pass.
        res4_img = self.backbone.forward_nhwc(both, want_res5=False)["res4"]
        pr = self.roi_heads.pooler.output_size if hasattr(self.roi_heads, "pooler") else -1
        if self.use_clip_c4 and tuple(res4_img.shape[1:3]) == (pr, pr):
            rs, rt, enc_img = self._region_level_encode(batched_inputs, extra_maps=res4_img)
        else:
            enc_img = self.backbone.attnpool(to_nchw(self.backbone.layer4.forward_nhwc(res4_img)))
            rs, rt = self._region_level_encode(batched_inputs)
        k = rs.shape[0]
        parts = [enc_img, rs, rt] + ([enc_teacher] if enc_teacher is not None else [])
        f = v2l(torch.cat(parts), clipcap_model)                                      # [2N + 2K (+ N), 768]
        out = {}
        if enc_teacher is not None:
            out["kd_loss"] = torch.nn.functional.l1_loss(f[2 * n + 2 * k:].detach(), f[:n])
        p = self.project(f[:2 * n + 2 * k])
        out["cont_loss"] = layers.contrastive_loss(gather_cat(p[n:2 * n].contiguous()), gather_cat(p[:n].contiguous()))
        e = p[2 * n:]
        out["cont_region_loss"] = layers.contrastive_loss(gather_cat(e[:k].contiguous()), gather_cat(e[k:].contiguous()))
        return out

    # ------------------------------------------------------------------ forward
    # ------------------------------------------------------------------ inference (rcnn.py:690-784)
    @torch.no_grad()
    def inference(self, batched_inputs: List[Dict], detected_instances=None, do_postprocess: bool = True):
        assert not self.training
        images, sizes = self.preprocess_image(batched_inputs, "image")
        res4 = self.backbone.forward_nhwc(images, want_res5=False)["res4"]
        feats = {"res4": to_nchw(res4)}
        if detected_instances is None:
            proposals, _ = self.proposal_generator.forward_nhwc(sizes, res4, None)
            kw = dict(res5=self.backbone.layer4, attnpool=self.backbone.attnpool) if self.use_clip_c4 else {}
            results, _ = self.roi_heads(ImageList(None, sizes), feats, proposals, None, **kw)
        else:
            results = self.roi_heads.forward_with_given_boxes(feats, [x.to(self.device) for x in detected_instances])
        return self._postprocess(results, batched_inputs, sizes) if do_postprocess else results

    @staticmethod
    def _postprocess(instances, batched_inputs, image_sizes):
        """rcnn.py:770-784: rescale to the dataset dict's original ``height`` / ``width``"""
        out = []
        for res, inp, size in zip(instances, batched_inputs, image_sizes):
            out.append({"instances": detector_postprocess(res, inp.get("height", size[0]), inp.get("width", size[1]))})
        return out

    def forward(self, batched_inputs: List[Dict], clipcap_model=None, branch="supervised", KD_regularization=True):
        if not self.training:
            return self.inference(batched_inputs)           # rcnn.py:353-354
        if branch == "caption_consistency":                     # rcnn.py:413-421
            both = self.preprocess_image_train(batched_inputs)
            cont, kd = self.v2l_contrastive(both, clipcap_model, KD_regularization)
            return {"cont_loss": cont, "kd_loss": kd} if kd is not None else {"cont_loss": cont}
        if branch == "caption_consistency_regionLevel":         # rcnn.py:422-470
            rs, rt = self._region_level_encode(batched_inputs)
            e = self.project(v2l(torch.cat([rs, rt]), clipcap_model))
            k = rs.shape[0]
            return layers.contrastive_loss(gather_cat(e[:k].contiguous()), gather_cat(e[k:].contiguous()))
        if branch == "caption_consistency_both":
            return self.forward_consistency(batched_inputs, clipcap_model, KD_regularization)
        # supervised: rcnn.py:592-623
        images, sizes = self.preprocess_image(batched_inputs, "image")
        gts = [as_instances(x["instances"]).to(self.device) for x in batched_inputs]
        share = self.share_source_pass and self.use_clip_c4 and self.div_pixel
        spec = hip.OutSpec() if share else None      # (the region-level branch appends the target images' res4 behind this one)
        res4 = (self.backbone.forward_nhwc(images, want_res5=False, res4_spec=spec) if share else
                self.backbone.forward_nhwc(images, want_res5=False))["res4"]
        # The RPN losses wait for host-side anchor sampling: finish them after the box head is in the queue.  Only when the
        # anchor and proposal samplers draw from separate generators (build_trainer seeds one each); with ONE shared stream
        # -- the reference's global RNG, which the oracle parity tests mirror -- the draws keep the reference's order.
        defer = self.defer_rpn_losses and self.proposal_generator.sample_generator is not self.roi_heads.sample_generator
        proposals, rpn_losses = self.proposal_generator.forward_nhwc(sizes, res4, gts, defer_losses=True)
        if not defer:
            done = rpn_losses()
            rpn_losses = lambda: done
        if self.use_clip_c4:    # C4 + CLIP weights: the head borrows the backbone's layer4 / attnpool (rcnn.py:606-612)
            _, detector_losses = self.roi_heads(ImageList(None, sizes), {"res4": to_nchw(res4)}, proposals, gts,
                                                res5=self.backbone.layer4, attnpool=self.backbone.attnpool)
        else:                   # default setting (rcnn.py:613-614)
            _, detector_losses = self.roi_heads(ImageList(None, sizes), {"res4": to_nchw(res4)}, proposals, gts)
        proposal_losses = rpn_losses()
        self._shared = None
        if share:
            self._shared = {"inputs": batched_inputs, "step": layers._STEP[0], "res4": res4, "sizes": sizes, "res4_spec": spec,
                            "proposals": proposals, "counts": list(self.proposal_generator.last_counts)}
        losses = {}
        losses.update(detector_losses)
        losses.update(proposal_losses)
        return losses


def build_model(cfg):
    """meta_arch/build.py:16-25"""
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model

"""Checkpoint compatibility -- fvcore ``Checkpointer`` / ``DetectionCheckpointer`` for native ``.pth``
files (detectron2/checkpoint/detection_checkpoint.py:62-132) and the three side loads of the CDDMSL trainer:

* ``MODEL.WEIGHTS``: ``{"model": state_dict[, "optimizer", "iteration"]}`` (or a bare state dict), parameter names as in the
  reference (this package keeps them: ``backbone.layer3.0.conv1.weight`` ..., ``roi_heads.box_predictor.*``); tensors whose shape
  does not match the model are skipped and reported, ``pixel_mean`` / ``pixel_std`` may be absent (:122-131);
* ``MODEL.PRE_TRAINED_RCLIP_PATH`` -> ``offline_backbone`` (engine/train_loop.py:150-161): every ``backbone.*`` tensor of that
  file that is not an ``offline_backbone`` / ``teacher_backbone`` one, prefix stripped;
* ``MODEL.VISION_TO_LANG_PATH`` -> the mapper (``clip_project.*`` of the ClipCap file, train_loop.py:281-288) and
  ``MODEL.CLIP.TEXT_EMB_PATH`` -> ``cls_score.weight`` (fast_rcnn.py:448-453) are read where those modules are built.

* the published RegionCLIP / OpenAI-CLIP files: a file whose name contains ``OAI_CLIP`` holds the CLIP tower under
  ``visual.*`` (plus the text tower); its names are converted with the rules of checkpoint/clip_model_loading.py:10-186 and
  matched to the model's by longest dotted suffix (:190-343) -- ``convert_clip_state``; a second checkpoint for the offline
  modules (``bb_rpn_weights``, :221-231) maps ``backbone`` -> ``offline_backbone``.

Files are read with ``torch.load(..., weights_only=True)`` only: nothing in a checkpoint is executed.  Pickled model-zoo
formats (``.pkl`` Caffe2 / ``.pyth`` pycls) are not supported.
"""
import os
import re
from typing import Dict, List, NamedTuple

import torch


class Incompatible(NamedTuple):
    missing_keys: List[str]
    unexpected_keys: List[str]
    incorrect_shapes: List[tuple]


def read_state(path) -> Dict:
    data = torch.load(path, map_location="cpu", weights_only=True)
    return data if isinstance(data, dict) and "model" in data else {"model": data}


def load_model_state(model, state: Dict[str, torch.Tensor]) -> Incompatible:
    """fvcore Checkpointer._load_model: drop shape-mismatched tensors, then a non-strict load."""
    own = model.state_dict()
    state = dict(state)
    bad = []
    for k in list(state.keys()):
        if k in own and tuple(own[k].shape) != tuple(state[k].shape):
            bad.append((k, tuple(state[k].shape), tuple(own[k].shape)))
            state.pop(k)
    res = model.load_state_dict(state, strict=False)
    missing = [k for k in res.missing_keys if k not in ("pixel_mean", "pixel_std")]
    return Incompatible(missing, list(res.unexpected_keys), bad)


# Name rules for CLIP-style checkpoints, in application order (clip_model_loading.py:28-44,68-115).  Plain entries are
# substring replacements, ``^`` entries are anchored regular expressions.  Only the rules a C4 box detector can meet are
# listed (FPN / mask / keypoint blob names have no counterpart in this package's models).
CLIP_NAME_RULES = (
    ("conv.rpn", "proposal_generator.rpn_head.conv"),
    ("rpn.bbox.pred", "proposal_generator.rpn_head.anchor_deltas"),
    ("rpn.cls.logits", "proposal_generator.rpn_head.objectness_logits"),
    (r"^bbox\.pred", "bbox_pred"),
    (r"^cls\.score", "cls_score"),
    (r"^fc6\.", "box_head.fc1."),
    (r"^fc7\.", "box_head.fc2."),
)


def _rename_clip_key(key: str, visual_to: str) -> str:
    if "visual.transformer" not in key:
        key = key.replace("visual.", visual_to)
    for pat, rep in CLIP_NAME_RULES:
        key = re.sub(pat, rep, key) if pat.startswith("^") else key.replace(pat, rep)
    return key


def convert_clip_state(model_state: Dict[str, torch.Tensor], ckpt: Dict[str, torch.Tensor], bb_rpn_weights=False):
    """checkpoint/clip_model_loading.py:190-343 -> (state dict with the MODEL's names, {model key: checkpoint key}).

    1. ``bb_rpn_weights`` (the second, offline-module checkpoint): keep ``backbone`` / ``proposal_generator`` tensors only,
       renamed to ``offline_backbone`` / ``offline_proposal_generator``.
    2. otherwise rename ``visual.`` -> ``backbone.`` when the model holds an ``offline_backbone`` as well (so the CLIP tower
       cannot be matched to both), -> ``clip_backbone.visual.`` for a whole-CLIP model, else strip it; then the blob-name rules.
       A ViT checkpoint (``visual.transformer``) keeps its names.
    3. every model key takes the checkpoint key that equals it or is its longest dotted suffix; a shape mismatch skips the
       pair; one checkpoint tensor matched by two model keys is an error.  ``bbox_pred`` drops the 4 background rows and
       ``cls_score`` moves the background row from first to last (Caffe2 layout, :167-183).
    Unmatched checkpoint tensors pass through under their converted names (reported as unexpected by the loader)."""
    if bb_rpn_weights:
        kept = {}
        for k, v in ckpt.items():
            if "backbone" in k:
                kept[k.replace("backbone", "offline_backbone")] = v
            if "proposal_generator" in k:
                kept[k.replace("proposal_generator", "offline_proposal_generator")] = v
        ckpt = kept
        visual_to = ""
    else:
        keys = list(model_state.keys())
        if any("clip_backbone" in k for k in keys):
            visual_to = "clip_backbone.visual."
        elif any("offline_backbone" in k for k in keys):
            visual_to = "backbone.bottom_up." if any("fpn" in k for k in keys) else "backbone."
        else:
            visual_to = ""
    vit = any("visual.transformer" in k for k in ckpt)
    renamed, origin = {}, {}
    for k in sorted(ckpt.keys()):
        nk = k if vit and k.startswith("visual.") else _rename_clip_key(k, visual_to)
        assert nk not in renamed, f"two checkpoint tensors convert to the same name {nk}"
        v = ckpt[k]
        if nk.startswith("bbox_pred."):
            v = v[4:]
        elif nk.startswith("cls_score."):
            v = torch.cat([v[1:], v[:1]])
        renamed[nk], origin[nk] = v, k
    # longest dotted suffix: walk each model key's suffixes from the longest down; the first one the checkpoint holds wins
    out, taken, pairs = {}, {}, {}
    for mk in sorted(model_state.keys()):
        parts = mk.split(".")
        for i in range(len(parts)):
            cand = ".".join(parts[i:])
            if cand in renamed:
                if tuple(renamed[cand].shape) != tuple(model_state[mk].shape):
                    break                                    # the reference warns and leaves this tensor out
                if cand in taken:
                    raise ValueError(f"Cannot match one checkpoint key to multiple keys in the model: {origin[cand]} -> {taken[cand]}, {mk}")
                taken[cand] = mk
                out[mk] = renamed[cand]
                pairs[mk] = origin[cand]
                break
    for nk, v in renamed.items():
        if nk not in taken:
            out.setdefault(nk, v)
    return out, pairs


def offline_backbone_state(all_params: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """train_loop.py:154-159: 'backbone' tensors of a RegionCLIP checkpoint, without the 'backbone.' prefix"""
    out = {}
    for k, v in all_params.items():
        if "backbone" in k and "offline_backbone" not in k and "teacher_backbone" not in k:
            out[k[9:]] = v
    return out


class DetectionCheckpointer:
    def __init__(self, model, save_dir="", optimizer=None, trainer=None):
        self.model, self.save_dir, self.optimizer, self.trainer = model, save_dir, optimizer, trainer

    def load(self, path, bb_rpn_weights=False) -> Incompatible:
        """detection_checkpoint.py:62-132: ``OAI_CLIP`` in the file name (or ``bb_rpn_weights``) -> name conversion first"""
        if not path:
            return Incompatible([], [], [])
        data = read_state(path)
        if "OAI_CLIP" in os.path.basename(path) or bb_rpn_weights:
            data["model"], self.last_matches = convert_clip_state(self.model.state_dict(), data["model"], bb_rpn_weights)
        inc = load_model_state(self.model, data["model"])
        if self.optimizer is not None and "optimizer" in data:
            self.optimizer.load_state_dict(data["optimizer"])
        if self.trainer is not None and "iteration" in data:
            self.trainer.iter = int(data["iteration"]) + 1            # resume at the next iteration (defaults.py:411-413)
        return inc

    def load_offline_backbone(self, path):
        """train_loop.py:150-161"""
        self.model.offline_backbone.load_state_dict(offline_backbone_state(read_state(path)["model"]))

    def save(self, name, iteration=None):
        os.makedirs(self.save_dir or ".", exist_ok=True)
        data = {"model": {k: v.detach().cpu().contiguous() for k, v in self.model.state_dict().items()}}
        if self.optimizer is not None:
            data["optimizer"] = self.optimizer.state_dict()
        if iteration is not None:
            data["iteration"] = int(iteration)
        path = os.path.join(self.save_dir, name + ".pth")
        torch.save(data, path)
        with open(os.path.join(self.save_dir, "last_checkpoint"), "w") as f:
            f.write(os.path.basename(path))
        return path

    def resume_or_load(self, path, resume=True) -> Incompatible:
        """fvcore Checkpointer.resume_or_load: the file named by ``last_checkpoint`` in save_dir if resuming and present"""
        last = os.path.join(self.save_dir, "last_checkpoint")
        if resume and self.save_dir and os.path.exists(last):
            with open(last) as f:
                path = os.path.join(self.save_dir, f.read().strip())
        return self.load(path)

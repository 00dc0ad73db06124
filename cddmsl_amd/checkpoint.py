"""Checkpoint compatibility -- fvcore ``Checkpointer`` / ``DetectionCheckpointer`` for native ``.pth``
files (detectron2/checkpoint/detection_checkpoint.py:62-132) and the three side loads of the CDDMSL trainer:

* ``MODEL.WEIGHTS``: ``{"model": state_dict[, "optimizer", "iteration"]}`` (or a bare state dict), parameter names as in the
  reference (this package keeps them: ``backbone.layer3.0.conv1.weight`` ..., ``roi_heads.box_predictor.*``); tensors whose shape
  does not match the model are skipped and reported, ``pixel_mean`` / ``pixel_std`` may be absent (:122-131);
* ``MODEL.PRE_TRAINED_RCLIP_PATH`` -> ``offline_backbone`` (engine/train_loop.py:150-161): every ``backbone.*`` tensor of that
  file that is not an ``offline_backbone`` / ``teacher_backbone`` one, prefix stripped;
* ``MODEL.VISION_TO_LANG_PATH`` -> the mapper (``clip_project.*`` of the ClipCap file, train_loop.py:281-288) and
  ``MODEL.CLIP.TEXT_EMB_PATH`` -> ``cls_score.weight`` (fast_rcnn.py:448-453) are read where those modules are built.

Files are read with ``torch.load(..., weights_only=True)`` only: nothing in a checkpoint is executed.  Pickled model-zoo
formats (``.pkl`` Caffe2 / ``.pyth`` pycls, with name-matching heuristics) are not supported.
"""
import os
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

    def load(self, path) -> Incompatible:
        if not path:
            return Incompatible([], [], [])
        data = read_state(path)
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

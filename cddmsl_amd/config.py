"""Dependency-free config with the reference's key names (yacs/fvcore are not installed here).

``get_cfg()`` returns the defaults of the hot-path keys (detectron2/config/defaults.py values, file:line in
SURVEY.md Appendix A); ``merge_from_file`` reads the reference's YAMLs incl. ``_BASE_`` inheritance
(configs/Base-RCNN-C4.yaml, configs/VOC-Experiments/faster_rcnn_CLIP_R_50_C4.yaml, ...); ``merge_from_list``
takes the CLI ``KEY VALUE`` pairs (engine/defaults.py:133-140).
"""
import ast
import copy
import os

import yaml


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def freeze(self):
        return self

    def merge_from_other(self, other, allow_new=True):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge_from_other(v, allow_new)
            else:
                self[k] = _to_node(v)

    def merge_from_file(self, path):
        with open(path) as f:
            d = yaml.safe_load(f) or {}
        base = d.pop("_BASE_", None)
        d.pop("BASE_", None)  # typo'd key in configs/AdverseWeather-Experiments/faster_rcnn_CLIP_R_50_C4.yaml:1
        if base:
            self.merge_from_file(os.path.join(os.path.dirname(path), base))
        self.merge_from_other(_to_node(_eval_strings(d)))

    def merge_from_list(self, lst):
        assert len(lst) % 2 == 0, "Override list has odd length"
        for k, v in zip(lst[0::2], lst[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                node = node.setdefault(p, CfgNode())
            if isinstance(v, str):
                try:
                    v = ast.literal_eval(v)
                except (ValueError, SyntaxError):
                    pass
            node[parts[-1]] = _to_node(v)


def auto_scale_workers(cfg, num_workers):
    """``DefaultTrainer.auto_scale_workers`` (engine/defaults.py:633-701; called from ``DefaultTrainer.__init__`` :374): a config
    written for ``SOLVER.REFERENCE_WORLD_SIZE`` workers, run on ``num_workers``, keeps its PER-GPU batch: total batch and base LR
    scale with the worker count, iteration counts (MAX_ITER, WARMUP_ITERS, STEPS, TEST.EVAL_PERIOD, CHECKPOINT_PERIOD) inversely.
    Returns the config itself when ``REFERENCE_WORLD_SIZE`` is 0 or already equals ``num_workers``, a scaled clone otherwise."""
    old = cfg.SOLVER.get("REFERENCE_WORLD_SIZE", 0)
    if old == 0 or old == num_workers:
        return cfg
    cfg = cfg.clone()
    assert cfg.SOLVER.IMS_PER_BATCH % old == 0, "Invalid REFERENCE_WORLD_SIZE in config!"
    scale = num_workers / old
    cfg.SOLVER.IMS_PER_BATCH = int(round(cfg.SOLVER.IMS_PER_BATCH * scale))
    cfg.SOLVER.BASE_LR = cfg.SOLVER.BASE_LR * scale
    cfg.SOLVER.MAX_ITER = int(round(cfg.SOLVER.MAX_ITER / scale))
    cfg.SOLVER.WARMUP_ITERS = int(round(cfg.SOLVER.WARMUP_ITERS / scale))
    cfg.SOLVER.STEPS = tuple(int(round(s / scale)) for s in cfg.SOLVER.STEPS)
    cfg.TEST.EVAL_PERIOD = int(round(cfg.TEST.EVAL_PERIOD / scale))
    cfg.SOLVER.CHECKPOINT_PERIOD = int(round(cfg.SOLVER.CHECKPOINT_PERIOD / scale))
    cfg.SOLVER.REFERENCE_WORLD_SIZE = num_workers      # maintain the invariant
    return cfg


def _eval_strings(d):
    """YAML leaves like "(480, 512)" are python tuples in the reference's configs."""
    if isinstance(d, dict):
        return {k: _eval_strings(v) for k, v in d.items()}
    if isinstance(d, str) and d[:1] in "([":
        try:
            return ast.literal_eval(d)
        except (ValueError, SyntaxError):
            return d
    return d


def _to_node(v):
    if isinstance(v, dict) and not isinstance(v, CfgNode):
        n = CfgNode()
        for k, x in v.items():
            n[k] = _to_node(x)
        return n
    return v


def get_cfg():
    C = _to_node({
        "VERSION": 2,
        "SEED": 1,
        "VIS_PERIOD": 0,
        "MODEL": {
            "DEVICE": "cuda", "META_ARCHITECTURE": "GeneralizedRCNN", "WEIGHTS": "", "MASK_ON": False, "KEYPOINT_ON": False,
            "KD_REGULRAZIATION": True, "PRE_TRAINED_RCLIP_PATH": "", "VISION_TO_LANG_PATH": "",
            "PIXEL_MEAN": [103.530, 116.280, 123.675], "PIXEL_STD": [1.0, 1.0, 1.0],
            "COMPUTE_DTYPE": "bf16",  # build-specific: "bf16" throughput path / "f32" parity path
            "BACKBONE": {"NAME": "build_resnet_backbone", "FREEZE_AT": 2},
            "RESNETS": {"DEPTH": 50, "OUT_FEATURES": ["res4"], "NORM": "FrozenBN", "RES2_OUT_CHANNELS": 256,
                        "STEM_OUT_CHANNELS": 64},
            "ANCHOR_GENERATOR": {"NAME": "DefaultAnchorGenerator", "SIZES": [[32, 64, 128, 256, 512]],
                                 "ASPECT_RATIOS": [[0.5, 1.0, 2.0]], "OFFSET": 0.0},
            "PROPOSAL_GENERATOR": {"NAME": "RPN", "MIN_SIZE": 0},
            "RPN": {"HEAD_NAME": "StandardRPNHead", "IN_FEATURES": ["res4"], "BOUNDARY_THRESH": -1,
                    "IOU_THRESHOLDS": [0.3, 0.7], "IOU_LABELS": [0, -1, 1], "BATCH_SIZE_PER_IMAGE": 256,
                    "POSITIVE_FRACTION": 0.5, "BBOX_REG_LOSS_TYPE": "smooth_l1", "BBOX_REG_LOSS_WEIGHT": 1.0,
                    "BBOX_REG_WEIGHTS": (1.0, 1.0, 1.0, 1.0), "SMOOTH_L1_BETA": 0.0, "LOSS_WEIGHT": 1.0,
                    "PRE_NMS_TOPK_TRAIN": 12000, "PRE_NMS_TOPK_TEST": 6000, "POST_NMS_TOPK_TRAIN": 2000,
                    "POST_NMS_TOPK_TEST": 1000, "NMS_THRESH": 0.7, "CONV_DIMS": [-1]},
            "ROI_HEADS": {"NAME": "Res5ROIHeads", "NUM_CLASSES": 80, "IN_FEATURES": ["res4"], "IOU_THRESHOLDS": [0.5],
                          "IOU_LABELS": [0, 1], "BATCH_SIZE_PER_IMAGE": 512, "POSITIVE_FRACTION": 0.25,
                          "SCORE_THRESH_TEST": 0.05, "NMS_THRESH_TEST": 0.5, "PROPOSAL_APPEND_GT": True},
            "ROI_BOX_HEAD": {"BBOX_REG_LOSS_TYPE": "smooth_l1", "BBOX_REG_LOSS_WEIGHT": 1.0,
                             "BBOX_REG_WEIGHTS": (10.0, 10.0, 5.0, 5.0), "SMOOTH_L1_BETA": 0.0, "POOLER_RESOLUTION": 14,
                             "POOLER_SAMPLING_RATIO": 0, "POOLER_TYPE": "ROIAlignV2", "CLS_AGNOSTIC_BBOX_REG": False},
            "CLIP": {"CROP_REGION_TYPE": "", "USE_TEXT_EMB_CLASSIFIER": False, "TEXT_EMB_PATH": None, "TEXT_EMB_DIM": 1024,
                     "NO_BOX_DELTA": False, "BG_CLS_LOSS_WEIGHT": None, "ONLY_SAMPLE_FG_PROPOSALS": False,
                     "CLSS_TEMP": 0.01, "FOCAL_SCALED_LOSS": None, "MULTIPLY_RPN_SCORE": False},
        },
        "INPUT": {"MIN_SIZE_TRAIN": (800,), "MAX_SIZE_TRAIN": 1333, "MIN_SIZE_TEST": 800, "MAX_SIZE_TEST": 1333, "FORMAT": "BGR",
                  "MIN_SIZE_TRAIN_SAMPLING": "choice", "RANDOM_FLIP": "horizontal"},
        "DATASETS": {"TRAIN": (), "TEST": ()},
        "DATALOADER": {"NUM_WORKERS": 4, "ASPECT_RATIO_GROUPING": True},
        "SOLVER": {"IMS_PER_BATCH": 16, "BASE_LR": 0.001, "MOMENTUM": 0.9, "NESTEROV": False, "WEIGHT_DECAY": 0.0001,
                   "WEIGHT_DECAY_NORM": 0.0, "GAMMA": 0.1, "STEPS": (30000,), "MAX_ITER": 40000, "WARMUP_FACTOR": 1.0 / 1000,
                   "WARMUP_ITERS": 1000, "WARMUP_METHOD": "linear", "CHECKPOINT_PERIOD": 5000, "BIAS_LR_FACTOR": 1.0,
                   "WEIGHT_DECAY_BIAS": 0.0001, "REFERENCE_WORLD_SIZE": 0,
                   "CLIP_GRADIENTS": {"ENABLED": False, "CLIP_TYPE": "value", "CLIP_VALUE": 1.0, "NORM_TYPE": 2.0},
                   "AMP": {"ENABLED": False}},
        "TEST": {"EVAL_PERIOD": 0, "DETECTIONS_PER_IMAGE": 100},
        "OUTPUT_DIR": "./output",
    })
    return C

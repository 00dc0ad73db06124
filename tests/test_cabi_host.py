"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/cddmsl_hip.h declares
(no compute without a GPU); host logic (config loader, registries, LR schedule, structures); product fails loudly
off-GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from cddmsl_amd import _lib
    return _lib.lib()


def test_exports_match_header(lib):
    hdr = open(os.path.join(ROOT, "include", "cddmsl_hip.h")).read()
    names = sorted(set(re.findall(r"\bint\s+(cddmsl_\w+)\s*\(", hdr)))
    assert len(names) >= 29
    for n in names:
        assert getattr(lib, n) is not None, n
    # and nothing exported that the header does not declare
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "cddmsl_amd", "libcddmsl_hip.so")], text=True)
    exported = sorted(set(re.findall(r"\bT (cddmsl_\w+)", out)))
    assert exported == names, set(exported) ^ set(names)
    assert lib.cddmsl_abi_version() == 3


def test_header_compiles_as_c():
    import subprocess
    subprocess.check_call(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "cddmsl_hip.h")])


def test_no_cpu_fallback():
    from cddmsl_amd import hip
    from cddmsl_amd._lib import HipLibraryError
    x = torch.zeros(1, 4, 4, 8)
    w = torch.zeros(8, 1, 1, 8)
    with pytest.raises(HipLibraryError):
        hip.conv_fwd(x, w)


def test_config_and_registry():
    from cddmsl_amd.config import get_cfg
    from cddmsl_amd.registry import BACKBONE_REGISTRY, META_ARCH_REGISTRY, PROPOSAL_GENERATOR_REGISTRY, ROI_HEADS_REGISTRY, RPN_HEAD_REGISTRY, ANCHOR_GENERATOR_REGISTRY
    import cddmsl_amd.modeling  # noqa: F401  (registers)
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "VOC-Experiments", "faster_rcnn_CLIP_R_50_C4.yaml"))
    cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", "16", "MODEL.CLIP.TEXT_EMB_PATH", "x.pth"])
    assert cfg.MODEL.META_ARCHITECTURE == "GeneralizedRCNN" and cfg.MODEL.ROI_HEADS.NAME == "CLIPRes5ROIHeads"
    assert cfg.MODEL.RPN.POST_NMS_TOPK_TEST == 1000 and cfg.SOLVER.STEPS[0] == 10000 and cfg.SOLVER.IMS_PER_BATCH == 16
    assert cfg.MODEL.CLIP.BG_CLS_LOSS_WEIGHT == 0.2 and cfg.INPUT.FORMAT == "RGB" and cfg.SOLVER.CLIP_GRADIENTS.CLIP_VALUE == 5.0
    assert "GeneralizedRCNN" in META_ARCH_REGISTRY and "build_clip_resnet_backbone" in BACKBONE_REGISTRY
    assert "RPN" in PROPOSAL_GENERATOR_REGISTRY and "StandardRPNHead" in RPN_HEAD_REGISTRY
    assert "DefaultAnchorGenerator" in ANCHOR_GENERATOR_REGISTRY and "CLIPRes5ROIHeads" in ROI_HEADS_REGISTRY
    cw = get_cfg()
    cw.merge_from_file(os.path.join(ROOT, "configs", "AdverseWeather-Experiments", "faster_rcnn_CLIP_R_50_C4.yaml"))
    assert cw.MODEL.KD_REGULRAZIATION is True and cw.MODEL.ROI_HEADS.NUM_CLASSES == 8


def test_lr_schedule_kat():
    """tests/test_scheduler.py:14-43 of the reference, through the product's solver."""
    import json
    from cddmsl_amd.config import get_cfg
    from cddmsl_amd.solver import lr_at
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "kat.json")))["scheduler"]
    cfg = get_cfg()
    cfg.merge_from_list(["SOLVER.BASE_LR", k["base_lr"], "SOLVER.STEPS", tuple(k["steps"]), "SOLVER.GAMMA", k["gamma"],
                         "SOLVER.WARMUP_FACTOR", k["warmup_factor"], "SOLVER.WARMUP_ITERS", k["warmup_iters"], "SOLVER.MAX_ITER", k["max_iter"]])
    lrs = [lr_at(cfg, i) for i in range(31)]
    assert all(abs(a - b) < 1e-9 for a, b in zip(lrs[:5], k["lrs_0_5"]))
    assert all(abs(v - k["lr_5_10"]) < 1e-9 for v in lrs[5:10]) and all(abs(v - k["lr_20_30"]) < 1e-9 for v in lrs[20:])


def test_auto_scale_workers_kat():
    """The worked example in the reference's own docstring (engine/defaults.py:649-668: a config written for 8 workers run on
    16), REFERENCE_WORLD_SIZE 0 / equal = untouched (:676-678), and the inverse direction; the input config is not modified."""
    from cddmsl_amd.config import auto_scale_workers, get_cfg
    cfg = get_cfg()
    cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", 16, "SOLVER.BASE_LR", 0.1, "SOLVER.REFERENCE_WORLD_SIZE", 8, "SOLVER.MAX_ITER", 5000,
                         "SOLVER.STEPS", (4000,), "SOLVER.CHECKPOINT_PERIOD", 1000, "SOLVER.WARMUP_ITERS", 100, "TEST.EVAL_PERIOD", 500])
    new = auto_scale_workers(cfg, 16)
    s = new.SOLVER
    assert (s.IMS_PER_BATCH, s.REFERENCE_WORLD_SIZE, s.MAX_ITER, s.STEPS, s.CHECKPOINT_PERIOD) == (32, 16, 2500, (2000,), 500)
    assert abs(s.BASE_LR - 0.2) < 1e-12 and s.WARMUP_ITERS == 50 and new.TEST.EVAL_PERIOD == 250
    assert cfg.SOLVER.IMS_PER_BATCH == 16 and cfg.SOLVER.MAX_ITER == 5000          # a clone was scaled
    assert auto_scale_workers(cfg, 8) is cfg
    half = auto_scale_workers(cfg, 4).SOLVER
    assert (half.IMS_PER_BATCH, half.MAX_ITER, half.STEPS) == (8, 10000, (8000,)) and abs(half.BASE_LR - 0.05) < 1e-12
    plain = get_cfg()
    assert plain.SOLVER.REFERENCE_WORLD_SIZE == 0 and auto_scale_workers(plain, 8) is plain
    # per-GPU batch is what stays fixed (data/build.py:287)
    assert new.SOLVER.IMS_PER_BATCH // 16 == cfg.SOLVER.IMS_PER_BATCH // 8


def test_state_dict_keys_match_reference_names():
    """SURVEY.md section 5 (checkpoint row): reference key names load into the product model (CPU construction only)."""
    from cddmsl_amd import synthetic
    from cddmsl_amd.config import get_cfg
    from cddmsl_amd.modeling.rcnn import GeneralizedRCNN
    from cddmsl_amd.modeling.clipcap import TransformerMapper
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "VOC-Experiments", "faster_rcnn_CLIP_R_50_C4.yaml"))
    m = GeneralizedRCNN(cfg)
    sd = synthetic.make_state_dict(0)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 48406683   # 48.4 M trainable
    assert m.backbone.layer3[0].conv2.weight.permute(0, 2, 3, 1).is_contiguous()
    mp = TransformerMapper()
    mp.load_state_dict(synthetic.make_mapper_state_dict(1))
    assert sum(p.numel() for p in mp.parameters()) == 69316608                      # 69.32 M frozen mapper params


def test_stock_r50_state_dict_keys():
    """configs[0]: stock Detectron2 R50-C4 names (stem.conv1.norm.*, res3.0.shortcut.*, roi_heads.res5.*) load into the product."""
    from cddmsl_amd import synthetic
    from cddmsl_amd.config import get_cfg
    from cddmsl_amd.modeling.rcnn import GeneralizedRCNN
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "PascalVOC-Detection", "faster_rcnn_R_50_C4.yaml"))
    m = GeneralizedRCNN(cfg)
    missing, unexpected = m.load_state_dict(synthetic.make_state_dict_r50(0), strict=False)
    assert not unexpected and all(k.startswith(("offline_backbone.", "projector.")) for k in missing)
    frozen = [n for n, p in m.backbone.named_parameters() if not p.requires_grad]
    assert any(n.startswith("stem.") for n in frozen) and any(n.startswith("res2.") for n in frozen)
    assert all(p.requires_grad for n, p in m.backbone.named_parameters() if n.startswith(("res3.", "res4.")))

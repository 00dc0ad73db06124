"""``__graft_entry__.smoke()``: one tiny full training step (all three branches, forward + backward) of the hot path
on cuda:0 through the HIP library (f32 parity path), checked against the CPU oracle -- every loss within 1e-3 rel."""
import os

import torch


def run():
    from . import _lib, synthetic
    from .config import get_cfg
    from .engine import SimpleTrainer
    from .modeling import TransformerMapper, build_model
    from .solver import build_optimizer
    _lib.lib()   # raises if the HIP library is missing: no fallback
    assert torch.cuda.is_available(), "smoke() needs the MI355X"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(root, "configs", "VOC-Experiments", "faster_rcnn_CLIP_R_50_C4.yaml"))
    cfg.merge_from_list(["MODEL.COMPUTE_DTYPE", "f32", "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 16, "MODEL.RPN.PRE_NMS_TOPK_TRAIN", 300,
                         "MODEL.RPN.POST_NMS_TOPK_TRAIN", 100, "MODEL.DEVICE", "cuda:0"])
    sd, msd = synthetic.make_state_dict(0), synthetic.make_mapper_state_dict(1)
    model = build_model(cfg)
    model.load_state_dict(sd, strict=False)
    mapper = TransformerMapper(compute_dtype=model.compute_dtype)
    mapper.load_state_dict(msd)
    mapper.to(model.device)
    g = torch.Generator().manual_seed(3)
    model.proposal_generator.sample_generator = model.roi_heads.sample_generator = model.region_generator = g
    batch = synthetic.make_batch(1, 128, 160, num_gt=2)
    tr = SimpleTrainer(model, iter([batch]), build_optimizer(cfg, model), cfg, clipcap_model=mapper, metrics_period=0)
    tr.iter = 20000
    losses = tr.run_step()
    got = {k: float(v.detach()) for k, v in losses.items()}

    from oracle import model as om   # the checker (test infrastructure)
    ocfg = om.Cfg(roi_batch_per_image=16, rpn_pre_nms_topk=300, rpn_post_nms_topk=100)
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    with torch.no_grad():
        ref = om.run_step_losses(sd, msd, ocfg, batch, 20000, torch.Generator().manual_seed(3))
    for k, v in ref.items():
        assert abs(got[k] - float(v)) <= 1e-3 * abs(float(v)) + 1e-6, (k, got[k], float(v))
    print("smoke ok:", got)

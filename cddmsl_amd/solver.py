"""Optimizer / LR schedule of the hot path -- detectron2/solver/build.py:43-130,220-262, solver/lr_scheduler.py:17-129.

``ClippedSGD`` = SGD(momentum 0.9, wd 1e-4) with *per-parameter* L2-norm clipping to 5.0 (solver/build.py:59-67,104),
run as ONE fused multi-tensor HIP launch pair (norms, update) instead of ~120 tensors x (norm + clip + update).
"""
import torch

from . import hip, layers


def lr_at(cfg, it):
    """WarmupParamScheduler(MultiStepParamScheduler) x LRMultiplier; KAT tests/test_scheduler.py:35-43."""
    s = cfg.SOLVER
    steps = [x for x in s.STEPS if x <= s.MAX_ITER]
    mult = s.GAMMA ** sum(1 for x in steps if it >= x)
    if it < s.WARMUP_ITERS:
        end = s.GAMMA ** sum(1 for x in steps if s.WARMUP_ITERS >= x)
        a = it / s.WARMUP_ITERS
        mult = s.WARMUP_FACTOR * (1 - a) + end * a
    return s.BASE_LR * mult


class ClippedSGD:
    def __init__(self, params, cfg):
        self.params = [p for p in params if p.requires_grad]
        s = cfg.SOLVER
        self.cfg = cfg
        self.momentum, self.wd = s.MOMENTUM, s.WEIGHT_DECAY
        cg = s.CLIP_GRADIENTS
        if cg.ENABLED:
            assert cg.CLIP_TYPE == "norm" and cg.NORM_TYPE == 2.0, "hot path = per-parameter L2-norm clipping (solver/build.py:59-67)"
            self.clip = cg.CLIP_VALUE
        else:       # solver/build.py:113-130 builds plain torch.optim.SGD (stock config #1): an infinite clip value = factor 1
            self.clip = float("inf")
        self.moms = None
        self.norm_ws = None
        self.steps_done = 0
        self.iteration = 0

    def zero_grad(self):
        for p in self.params:
            if p.grad is not None:
                p.grad.zero_()

    def state_dict(self):
        """momentum buffers in parameter order (torch.optim.SGD's 'state' keyed by parameter index) + step counters"""
        moms = None if self.moms is None else [self.moms[id(p)].detach().cpu() for p in self.params]
        return {"momentum_buffers": moms, "steps_done": self.steps_done, "iteration": self.iteration}

    def load_state_dict(self, sd):
        self.steps_done, self.iteration = int(sd["steps_done"]), int(sd["iteration"])
        if sd["momentum_buffers"] is not None:
            assert len(sd["momentum_buffers"]) == len(self.params)
            self.moms = {id(p): m.to(p.device).clone(memory_format=torch.preserve_format) for p, m in zip(self.params, sd["momentum_buffers"])}
            self.norm_ws = torch.zeros(len(self.params), device=self.params[0].device, dtype=torch.float32)

    def step(self):
        touched = layers.take_touched()     # torch.optim.SGD skips parameters whose grad is None (e.g. the projector in stock configs)
        ps = [p for p in self.params if p.grad is not None and id(p) in touched]
        if self.moms is None:
            self.moms = {id(p): torch.zeros_like(p, memory_format=torch.preserve_format) for p in self.params}
            self.norm_ws = torch.zeros(len(self.params), device=self.params[0].device, dtype=torch.float32)
        lr = lr_at(self.cfg, self.iteration)
        with torch.no_grad():
            hip.sgd_clip_step([p.data for p in ps], [p.grad for p in ps], [self.moms[id(p)] for p in ps], self.norm_ws,
                              lr, self.momentum, self.wd, self.clip, self.steps_done == 0)
        self.steps_done += 1
        self.iteration += 1
        layers.bump_weight_version()   # prepared (cast / transposed) weights are stale now
        layers.FP8_SCALES.roll()       # fp8 configuration: this step's recorded activation maxima become the next step's scales
        return lr


def build_optimizer(cfg, model):
    return ClippedSGD(model.parameters(), cfg)

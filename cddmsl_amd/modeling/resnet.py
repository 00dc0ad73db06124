"""Stock Detectron2 ResNet-50-C4 on the HIP conv kernels (BASELINE.json configs[0]).

Mirrors detectron2/modeling/backbone/resnet.py: ``BasicStem`` :330-359 (7x7 s2 conv + FrozenBN + ReLU + 3x3 s2 max-pool),
``BottleneckBlock`` :100-210 (stride in the first 1x1, projection shortcut), ``ResNet`` :362-459, ``build_resnet_backbone``
:614-695; same parameter names (``stem.conv1.norm.weight``, ``res3.0.shortcut.weight`` ...).  A stage is one autograd node;
the input gradient of the stride-2 1x1 convs is a dense 1x1 dgrad at low resolution followed by a zero-interleaving
scatter kernel (``cddmsl_upsample_zero2``).
"""
import torch
from torch import nn

from .. import hip, layers
from ..layers import _grad_buf, _ohwi
from ..registry import BACKBONE_REGISTRY
from ..structures import ShapeSpec
from .backbone import FrozenBatchNorm2d, to_nchw, to_nhwc


class ConvNorm(nn.Module):
    """``Conv2d(bias=False, norm=FrozenBN)`` holder with Detectron2's names: ``.weight`` and ``.norm.*``."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k).contiguous(memory_format=torch.channels_last))
        self.norm = FrozenBatchNorm2d(cout)


class BottleneckBlock(nn.Module):
    def __init__(self, in_channels, out_channels, bottleneck_channels, stride=1):
        super().__init__()
        self.stride, self.frozen = stride, False
        self.shortcut = ConvNorm(in_channels, out_channels, 1) if in_channels != out_channels else None
        self.conv1 = ConvNorm(in_channels, bottleneck_channels, 1)     # STRIDE_IN_1X1: the stride lives here
        self.conv2 = ConvNorm(bottleneck_channels, bottleneck_channels, 3)
        self.conv3 = ConvNorm(bottleneck_channels, out_channels, 1)
        self._pw = None

    def prepared(self):
        dev = (self.conv1.weight.device, layers.load_generation())
        if self._pw is None or self._pw[0] != (dev, self.frozen):
            mods = (self.conv1, self.conv2, self.conv3, self.shortcut)
            aff = tuple(None if m is None else m.norm.affine() for m in mods)
            pw = tuple(None if m is None else layers.PreparedWeight(m.weight, a[0], self.frozen) for m, a in zip(mods, aff))
            self._pw = ((dev, self.frozen), pw, aff)
        return self._pw[1], self._pw[2]


def _blk_forward(x, blk, save):
    T = x.dtype
    pw, aff = blk.prepared()
    s = blk.stride
    o1 = hip.conv_fwd(x, pw[0].get(T, False)[0], aff[0][0], aff[0][1], relu=True, stride=s)
    o2 = hip.conv_fwd(o1, pw[1].get(T, False)[0], aff[1][0], aff[1][1], relu=True, pad=1)
    sc = hip.conv_fwd(x, pw[3].get(T, False)[0], aff[3][0], aff[3][1], stride=s) if pw[3] is not None else x
    out = hip.conv_fwd(o2, pw[2].get(T, False)[0], aff[2][0], aff[2][1], residual=sc, relu=True)
    return out, ((o1, o2) if save else None)


def _blk_backward(gs, x, o1, o2, blk, need_dx, mask_x):
    T = x.dtype
    pw, aff = blk.prepared()
    s = blk.stride
    w1, w2, w3 = blk.conv1.weight, blk.conv2.weight, blk.conv3.weight
    hip.conv_wgrad(o2, gs, _ohwi(w3).shape, aff[2][0], out=_ohwi(_grad_buf(w3)))
    dpre2 = hip.conv_fwd(gs, pw[2].get(T, True)[1], relu_mask=o2)
    hip.conv_wgrad(o1, dpre2, _ohwi(w2).shape, aff[1][0], pad=1, out=_ohwi(_grad_buf(w2)))
    dpre1 = hip.conv_fwd(dpre2, pw[1].get(T, True)[1], pad=1, relu_mask=o1)
    hip.conv_wgrad(x, dpre1, _ohwi(w1).shape, aff[0][0], stride=s, out=_ohwi(_grad_buf(w1)))
    if blk.shortcut is not None:
        ws = blk.shortcut.weight
        hip.conv_wgrad(x, gs, _ohwi(ws).shape, aff[3][0], stride=s, out=_ohwi(_grad_buf(ws)))
    if not need_dx:
        return None
    mask = x if mask_x else None
    if s == 1:
        dsc = hip.conv_fwd(gs, pw[3].get(T, True)[1]) if blk.shortcut is not None else gs
        return hip.conv_fwd(dpre1, pw[0].get(T, True)[1], residual=dsc, relu_mask=mask)
    # stride 2: both 1x1 input gradients live on the even pixels; sum them at low resolution, then scatter
    dsc = hip.conv_fwd(gs, pw[3].get(T, True)[1])
    low = hip.conv_fwd(dpre1, pw[0].get(T, True)[1], residual=dsc)
    return hip.upsample_zero2(low, tuple(x.shape), mask=mask)


class StockStageFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, blocks):
        saved, cur = [x], x
        for b in blocks:
            cur, mids = _blk_forward(cur, b, True)
            saved += [mids[0], mids[1], cur]
        ctx.blocks = blocks
        ctx.save_for_backward(*saved)
        return cur

    @staticmethod
    def backward(ctx, g):
        saved, blocks = ctx.saved_tensors, ctx.blocks
        need_dx = ctx.needs_input_grad[0]
        gs = hip.relu_bwd(g.contiguous(), saved[-1])
        for i in range(len(blocks) - 1, -1, -1):
            x, o1, o2 = saved[3 * i], saved[3 * i + 1], saved[3 * i + 2]
            gs = _blk_backward(gs, x, o1, o2, blocks[i], need_dx or i > 0, mask_x=i > 0)
        return gs, None, None


class Stage(nn.Sequential):
    def forward_nhwc(self, x):
        blocks = list(self)
        if blocks[0].frozen or not torch.is_grad_enabled():
            for b in blocks:
                x, _ = _blk_forward(x, b, False)
            return x
        return StockStageFn.apply(x, blocks[0].conv1.weight, blocks)

    def forward(self, x):
        return to_nchw(self.forward_nhwc(to_nhwc(x)))


def make_stage(num_blocks, stride_per_block, in_channels, bottleneck_channels, out_channels):
    """ResNet.make_stage resnet.py:462-520"""
    blocks = []
    for i in range(num_blocks):
        blocks.append(BottleneckBlock(in_channels, out_channels, bottleneck_channels, stride_per_block[i]))
        in_channels = out_channels
    return Stage(*blocks)


class BasicStem(nn.Module):
    def __init__(self, in_channels=3, out_channels=64):
        super().__init__()
        self.conv1 = ConvNorm(in_channels, out_channels, 7)
        self._w = None

    def forward_nhwc(self, x):
        T, cp = x.dtype, x.shape[-1]
        key = (T, cp, self.conv1.weight.device, layers.load_generation(), self.conv1.weight._version)
        if self._w is None or self._w[0] != key:
            w = torch.zeros(self.conv1.weight.shape[0], 7, 7, cp, device=x.device)
            w[..., :3] = self.conv1.weight.detach().permute(0, 2, 3, 1)
            self._w = (key, hip.weight_prep(w, None, T, True, False)[0])
        s, b = self.conv1.norm.affine()
        x = hip.conv_fwd(x, self._w[1], s, b, relu=True, stride=2, pad=3)
        return hip.maxpool3s2_fwd(x)


class ResNet(nn.Module):
    def __init__(self, stem, stages, out_features, freeze_at, compute_dtype):
        super().__init__()
        self.stem, self.compute_dtype = stem, compute_dtype
        self.stage_names = []
        ch, stride = {"stem": 64}, {"stem": 4}
        cur = 4
        for i, st in enumerate(stages):
            name = f"res{i + 2}"
            self.add_module(name, st)
            self.stage_names.append(name)
            cur *= max(b.stride for b in st)
            ch[name], stride[name] = st[-1].conv3.weight.shape[0], cur
        self._out_features = list(out_features)
        self._ch, self._st = ch, stride
        self.freeze(freeze_at)

    @property
    def size_divisibility(self):
        return 0

    def output_shape(self):
        return {n: ShapeSpec(channels=self._ch[n], stride=self._st[n]) for n in self._out_features}

    def freeze(self, freeze_at=0):
        """resnet.py:417-441.  The stem always runs frozen on the HIP path (every shipped config has FREEZE_AT >= 1)."""
        assert freeze_at >= 1
        for p in self.stem.parameters():
            p.requires_grad = False
        for idx, name in enumerate(self.stage_names, start=2):
            if freeze_at >= idx:
                for blk in getattr(self, name):
                    blk.frozen = True
                    for p in blk.parameters():
                        p.requires_grad = False
        return self

    def forward_nhwc(self, x, want_res5=None):
        with torch.no_grad():
            x = self.stem.forward_nhwc(x)
        out = {}
        for name in self.stage_names:
            x = getattr(self, name).forward_nhwc(x)
            if name in self._out_features:
                out[name] = x
        return out

    def forward(self, x):
        T = self.compute_dtype
        cp = 8 if T == torch.bfloat16 else 4
        xin = torch.zeros(x.shape[0], x.shape[2], x.shape[3], cp, device=x.device, dtype=T)
        xin[..., :3] = x.permute(0, 2, 3, 1)
        return {k: to_nchw(v) for k, v in self.forward_nhwc(xin).items()}


@BACKBONE_REGISTRY.register()
def build_resnet_backbone(cfg, input_shape=None):
    """resnet.py:614-695 (depth 50/101/152 bottleneck nets, FrozenBN, no deformable / dilated variants)."""
    r = cfg.MODEL.RESNETS
    assert r.NORM == "FrozenBN" and r.get("STRIDE_IN_1X1", True) and r.get("NUM_GROUPS", 1) == 1 and r.get("RES5_DILATION", 1) == 1
    nblocks = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}[r.DEPTH]
    out_features = r.OUT_FEATURES
    max_stage = max({"res2": 2, "res3": 3, "res4": 4, "res5": 5}[f] for f in out_features)
    inc, outc, bott = r.STEM_OUT_CHANNELS, r.RES2_OUT_CHANNELS, r.get("WIDTH_PER_GROUP", 64)
    stages = []
    for idx, stage_idx in enumerate(range(2, max_stage + 1)):
        first = 1 if idx == 0 else 2
        stages.append(make_stage(nblocks[idx], [first] + [1] * (nblocks[idx] - 1), inc, bott, outc))
        inc, outc, bott = outc, outc * 2, bott * 2
    dt = {"bf16": torch.bfloat16, "f32": torch.float32, "fp8": torch.bfloat16}[cfg.MODEL.get("COMPUTE_DTYPE", "bf16")]   # (stock ResNet: fp8 = bf16)
    return ResNet(BasicStem(3, r.STEM_OUT_CHANNELS), stages, out_features, cfg.MODEL.BACKBONE.FREEZE_AT, dt)

"""CLIP ModifiedResNet backbone on the HIP conv kernels.

Mirrors detectron2/modeling/backbone/clip_backbone.py: ``Bottleneck`` :14-70, ``AttentionPool2d`` :73-107,
``ModifiedResNet`` :110-270, ``build_clip_resnet_backbone`` :664-729 -- same module/parameter names (state-dict
compatible), same ``Backbone`` ABI (``forward(x NCHW) -> dict``, ``output_shape()``, ``size_divisibility``) and the
``.layer4`` / ``.attnpool`` callables the ROI head borrows (rcnn.py:608-609).

Tensors crossing the module boundary are logically NCHW in ``torch.channels_last`` memory (= NHWC for the kernels,
zero-copy).  FrozenBN is never a pass of its own: its affine is folded into the conv epilogues.
"""
from collections import OrderedDict

import torch
from torch import nn

from .. import hip, layers
from ..registry import BACKBONE_REGISTRY
from ..structures import ShapeSpec


def to_nhwc(x):
    """logical NCHW (channels_last memory) -> contiguous NHWC view"""
    v = x.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def to_nchw(x):
    return x.permute(0, 3, 1, 2)


class Conv2dW(nn.Module):
    """Weight holder named like nn.Conv2d (bias-free): ``.weight`` [Cout,Cin,KH,KW] f32, channels_last."""

    def __init__(self, cin, cout, k):
        super().__init__()
        w = torch.empty(cout, cin, k, k).contiguous(memory_format=torch.channels_last)
        self.weight = nn.Parameter(w)


class FrozenBatchNorm2d(nn.Module):
    """detectron2/layers/batch_norm.py:14-66 -- buffers only; ``affine()`` gives the folded (scale, bias)."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)
        self._aff = None

    def affine(self):
        if self._aff is None or self._aff[0].device != self.weight.device:
            scale = self.weight * (self.running_var + self.eps).rsqrt()
            bias = self.bias - self.running_mean * scale
            self._aff = (scale.float().contiguous(), bias.float().contiguous())
        return self._aff

    def _load_from_state_dict(self, *a, **k):
        self._aff = None
        layers.note_weights_loaded()          # derived caches above this module (block params, prepared weights) are stale
        return super()._load_from_state_dict(*a, **k)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1, self.bn1 = Conv2dW(inplanes, planes, 1), FrozenBatchNorm2d(planes)
        self.conv2, self.bn2 = Conv2dW(planes, planes, 3), FrozenBatchNorm2d(planes)
        self.conv3, self.bn3 = Conv2dW(planes, planes * 4, 1), FrozenBatchNorm2d(planes * 4)
        self.stride = stride
        self.downsample = None
        if stride > 1 or inplanes != planes * 4:
            self.downsample = nn.Sequential(OrderedDict([("0", Conv2dW(inplanes, planes * 4, 1)),
                                                         ("1", FrozenBatchNorm2d(planes * 4))]))
        self.frozen = False

    def params(self):
        ds = self.downsample
        return layers.BlockParams(self.conv1.weight, self.conv2.weight, self.conv3.weight,
                                  None if ds is None else ds[0].weight,
                                  self.bn1.affine(), self.bn2.affine(), self.bn3.affine(),
                                  None if ds is None else ds[1].affine(), self.stride, self.frozen, getattr(self, "fp8", False))


class ResStage(nn.Sequential):
    """One residual stage; callable on logical-NCHW tensors (the ROI head calls ``backbone.layer4(x)``)."""

    def __init__(self, *blocks):
        super().__init__(*blocks)
        self._bp = None

    def block_params(self):
        dev = (self[0].conv1.weight.device, layers.load_generation())
        if self._bp is None or self._bp[0] != dev:
            self._bp = (dev, [b.params() for b in self])
        return self._bp[1]

    def forward_nhwc(self, x, then_attnpool=None, out_spec=None):
        """``then_attnpool`` (an AttentionPool2d): returns the pooled embeddings of the stage output instead of the map --
        the RoI head's layer4 -> attnpool composition with the ReLU backward fused (layers.res_stage_attnpool)."""
        frozen = self[0].frozen
        if then_attnpool is not None:
            return layers.res_stage_attnpool(x, self.block_params(), frozen, then_attnpool._params())
        return layers.res_stage(x, self.block_params(), frozen, out_spec=out_spec)

    def forward(self, x):
        return to_nchw(self.forward_nhwc(to_nhwc(x)))


class AttentionPool2d(nn.Module):
    def __init__(self, spacial_dim, embed_dim, num_heads, output_dim=None):
        super().__init__()
        self.positional_embedding = nn.Parameter(torch.randn(spacial_dim ** 2 + 1, embed_dim) / embed_dim ** 0.5)
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.c_proj = nn.Linear(embed_dim, output_dim or embed_dim)
        self.num_heads = num_heads
        self._ap = None

    def _params(self):
        frozen = not self.q_proj.weight.requires_grad
        if self._ap is None or self._ap.pos.device != self.positional_embedding.device or self._ap.frozen != frozen:
            self._ap = layers.AttnPoolParams(self.positional_embedding, self.q_proj.weight, self.q_proj.bias,
                                             self.k_proj.weight, self.k_proj.bias, self.v_proj.weight, self.v_proj.bias,
                                             self.c_proj.weight, self.c_proj.bias, self.num_heads, frozen)
        return self._ap

    def forward(self, x):
        """x logical NCHW [K,C,7,7] -> [K, output_dim] f32 (token 0 of the MHA output, clip_backbone.py:107)"""
        return layers.attnpool(to_nhwc(x), self._params())


class ModifiedResNet(nn.Module):
    def __init__(self, layers_, output_dim, heads, input_resolution=224, width=64, out_features=None, freeze_at=0,
                 depth=None, pool_vec=False, create_att_pool=True, compute_dtype=torch.bfloat16):
        super().__init__()
        self.output_dim, self.input_resolution, self.compute_dtype = output_dim, input_resolution, compute_dtype
        self.conv1, self.bn1 = Conv2dW(3, width // 2, 3), FrozenBatchNorm2d(width // 2)
        self.conv2, self.bn2 = Conv2dW(width // 2, width // 2, 3), FrozenBatchNorm2d(width // 2)
        self.conv3, self.bn3 = Conv2dW(width // 2, width, 3), FrozenBatchNorm2d(width)
        self._inplanes = width
        self.layer1 = self._make_layer(width, layers_[0])
        self.layer2 = self._make_layer(width * 2, layers_[1], stride=2)
        self.layer3 = self._make_layer(width * 4, layers_[2], stride=2)
        self.layer4 = self._make_layer(width * 8, layers_[3], stride=2)
        self.attnpool = AttentionPool2d(input_resolution // 32, width * 32, heads, output_dim)
        self._out_features = list(out_features) if out_features else []
        ch = {"stem": width, "res2": width * 4, "res3": width * 8, "res4": width * 16, "res5": width * 32}
        st = {"stem": 4, "res2": 4, "res3": 8, "res4": 16, "res5": 32}
        self._out_feature_channels = {k: ch[k] for k in ch if k != "res5" or "res5" in self._out_features}
        self._out_feature_strides = {k: st[k] for k in self._out_feature_channels}
        self._stem_w = None
        self.freeze(freeze_at)

    def _make_layer(self, planes, blocks, stride=1):
        ls = [Bottleneck(self._inplanes, planes, stride)]
        self._inplanes = planes * 4
        for _ in range(1, blocks):
            ls.append(Bottleneck(self._inplanes, planes))
        return ResStage(*ls)

    @property
    def size_divisibility(self):
        return 0  # modeling/backbone/backbone.py:31-40 (not overridden by ModifiedResNet)

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n])
                for n in self._out_features}

    def freeze(self, freeze_at=0):
        """clip_backbone.py:221-261.  The stem is always run frozen here (every shipped config has FREEZE_AT >= 1)."""
        assert freeze_at >= 1, "the HIP stem path has no backward: MODEL.BACKBONE.FREEZE_AT must be >= 1"
        self.freeze_at = freeze_at
        for m in (self.conv1, self.conv2, self.conv3):
            m.weight.requires_grad = False
        for idx, stage in enumerate([self.layer1, self.layer2, self.layer3, self.layer4], start=2):
            if freeze_at >= idx:
                for blk in stage:
                    blk.frozen = True
                    for p in blk.parameters():
                        p.requires_grad = False
        return self

    # ------------------------------------------------------------------ forward
    def _stem_weights(self, T, cp):
        dev = self.conv1.weight.device
        key = (T, cp, dev, layers.load_generation(), self.conv1.weight._version, self.conv2.weight._version, self.conv3.weight._version)
        if self._stem_w is None or self._stem_w[0] != key:
            w1 = torch.zeros(self.conv1.weight.shape[0], 3, 3, cp, device=dev)
            w1[..., :3] = self.conv1.weight.detach().permute(0, 2, 3, 1)
            ws = [hip.weight_prep(w1, None, T, True, False)[0],
                  hip.weight_prep(layers._ohwi(self.conv2.weight.detach()), None, T, True, False)[0],
                  hip.weight_prep(layers._ohwi(self.conv3.weight.detach()), None, T, True, False)[0]]
            self._stem_w = (key, ws)
        return self._stem_w[1]

    def stem_nhwc(self, x):
        """x NHWC [N,H,W,Cp] (Cp = 3 padded to a 16-byte pixel) -> [N,H/4,W/4,width]   clip_backbone.py:194-198"""
        w1, w2, w3 = self._stem_weights(x.dtype, x.shape[-1])
        (s1, b1), (s2, b2), (s3, b3) = self.bn1.affine(), self.bn2.affine(), self.bn3.affine()
        x = hip.conv_fwd(x, w1, s1, b1, relu=True, stride=2, pad=1)
        x = hip.conv_fwd(x, w2, s2, b2, relu=True, pad=1)
        x = hip.conv_fwd(x, w3, s3, b3, relu=True, pad=1)
        return hip.avgpool2_fwd(x)

    def forward_nhwc(self, x, want_res5=None, res4_spec=None):
        """x: preprocessed NHWC input in the compute dtype.  Returns NHWC feature maps.
        ``want_res5=False`` skips the full-image layer4 the caller will not read.  ``res4_spec`` (hip.OutSpec):
        res4 is written into one half of a buffer shared with a second pass over other images of the same size."""
        with torch.no_grad():
            x = self.stem_nhwc(x)
        x = self.layer1.forward_nhwc(x)
        x = self.layer2.forward_nhwc(x)
        res4 = self.layer3.forward_nhwc(x, out_spec=res4_spec)
        out = {"res4": res4}
        if "res5" in self._out_features and want_res5 is not False:
            out["res5"] = self.layer4.forward_nhwc(res4)
        return out

    def forward(self, x, want_res5=None):
        """x logical NCHW float [N,3,H,W] (already normalised) -> {'res4','res5'} logical NCHW (channels_last)."""
        assert x.dim() == 4, f"ResNet takes an input of shape (N, C, H, W). Got {x.shape} instead!"
        T = self.compute_dtype
        cp = 8 if T == torch.bfloat16 else 4
        xin = torch.zeros(x.shape[0], x.shape[2], x.shape[3], cp, device=x.device, dtype=T)
        xin[..., :3] = x.permute(0, 2, 3, 1)
        return {k: to_nchw(v) for k, v in self.forward_nhwc(xin, want_res5).items()}


@BACKBONE_REGISTRY.register()
def build_clip_resnet_backbone(cfg, input_shape=None):
    """clip_backbone.py:664-729"""
    depth = cfg.MODEL.RESNETS.DEPTH
    blocks = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 200: [4, 6, 10, 6]}[depth]
    width = {50: 64, 101: 64, 200: 80}[depth]
    embed_dim = {50: 1024, 101: 512, 200: 640}[depth]
    res = {50: 224, 101: 224, 200: 288}[depth]
    mode = cfg.MODEL.get("COMPUTE_DTYPE", "bf16")
    dt = {"bf16": torch.bfloat16, "f32": torch.float32, "fp8": torch.bfloat16}[mode]
    net = ModifiedResNet(blocks, embed_dim, width * 32 // 64, res, width, cfg.MODEL.RESNETS.OUT_FEATURES,
                         cfg.MODEL.BACKBONE.FREEZE_AT, depth, False, True, dt)
    if mode == "fp8":      # BASELINE.json configs[4]: e4m3 forward GEMMs where they are MFMA-bound (layers.conv_fwd_auto), rest bf16
        for m in net.modules():
            if isinstance(m, Bottleneck):
                m.fp8 = True
    return net


def build_backbone(cfg, input_shape=None):
    """detectron2/modeling/backbone/build.py:20-33"""
    if input_shape is None:
        input_shape = ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))
    return BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, input_shape)

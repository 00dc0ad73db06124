"""Generates tests/golden/ref_*.npz by running the REFERENCE's own leaf modules (read from
/root/reference, never copied) on seeded inputs.  Runs only in the build container; the fixtures
(inputs + expected outputs, plain arrays) are what travels.

The reference package cannot be imported as a package (fvcore/torchvision/yacs/clip
are not installed, detectron2._C cannot be built, meta_arch/__init__ imports a missing file), so the
leaf files are imported with their package ``__init__``s bypassed and third-party roots stubbed.
Third-party arithmetic the reference calls but does not vendor (torchvision roi_align / nms /
Resize, fvcore smooth_l1) is supplied here by small independent pure-PyTorch/Python definitions
following SURVEY.md Appendix C -- independent of oracle/ so the comparison is not circular.

usage:  python tests/golden/make_golden.py
"""
import importlib
import importlib.abc
import importlib.machinery
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


# ---------------------------------------------------------------- stubs
class _Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Anything()


class _StubModule(types.ModuleType):
    __path__ = []

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Anything


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ("fvcore", "torchvision", "clip", "diffdist", "pycocotools", "cv2", "omegaconf", "yacs",
             "termcolor", "iopath", "lvis", "timm", "ftfy", "tensorboard", "skimage", "transformers")

    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in self.ROOTS and name not in sys.modules:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        pass


def _pkg(name, rel):
    m = types.ModuleType(name)
    m.__path__ = [os.path.join(REF, rel)]
    m.__package__ = name
    sys.modules[name] = m
    return m


class Registry:
    """Minimal fvcore.common.registry.Registry."""

    def __init__(self, name):
        self._name, self._obj_map = name, {}

    def register(self, obj=None):
        if obj is None:
            def deco(f):
                self._obj_map[f.__name__] = f
                return f
            return deco
        self._obj_map[obj.__name__] = obj

    def get(self, name):
        return self._obj_map[name]


def smooth_l1_loss(inp, tgt, beta, reduction="none"):  # fvcore.nn.smooth_l1_loss (published definition)
    if beta < 1e-5:
        loss = torch.abs(inp - tgt)
    else:
        n = torch.abs(inp - tgt)
        loss = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
    return loss.sum() if reduction == "sum" else loss.mean() if reduction == "mean" else loss


def py_nms(boxes, scores, thr):
    """torchvision.ops.nms published algorithm, O(n^2) python (independent of oracle/native.c)."""
    order = sorted(range(len(scores)), key=lambda i: (-float(scores[i]), i))
    b = boxes.tolist()
    keep, dead = [], set()
    for ai, i in enumerate(order):
        if i in dead:
            continue
        keep.append(i)
        xi0, yi0, xi1, yi1 = [np.float32(v) for v in b[i]]
        area_i = (xi1 - xi0) * (yi1 - yi0)
        for j in order[ai + 1:]:
            if j in dead:
                continue
            xj0, yj0, xj1, yj1 = [np.float32(v) for v in b[j]]
            w = max(np.float32(0), min(xi1, xj1) - max(xi0, xj0))
            h = max(np.float32(0), min(yi1, yj1) - max(yi0, yj0))
            inter = np.float32(w * h)
            area_j = (xj1 - xj0) * (yj1 - yj0)
            if inter / (area_i + area_j - inter) > np.float32(thr):
                dead.add(j)
    return torch.tensor(keep, dtype=torch.int64)


def py_batched_nms(boxes, scores, idxs, thr):
    if boxes.numel() == 0:
        return torch.zeros(0, dtype=torch.int64)
    off = idxs.to(boxes) * (boxes.max() + 1)
    return py_nms(boxes + off[:, None], scores, thr)


def torch_roi_align(x, rois, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    """torchvision.ops.roi_align published algorithm as differentiable PyTorch ops (slow, tiny cases)."""
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    N, C, H, W = x.shape
    outs = []
    for r in rois:
        b = int(r[0])
        off = 0.5 if aligned else 0.0
        x0, y0, x1, y1 = [float(v) * spatial_scale - off for v in r[1:]]
        x0, y0, x1, y1 = [np.float32(v) for v in (x0, y0, x1, y1)]
        rw, rh = x1 - x0, y1 - y0
        if not aligned:
            rw, rh = max(rw, np.float32(1)), max(rh, np.float32(1))
        bh, bw = np.float32(rh / ph), np.float32(rw / pw)
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / ph))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / pw))
        count = max(gh * gw, 1)
        out = x.new_zeros(C, ph, pw)
        for i in range(ph):
            for j in range(pw):
                acc = x.new_zeros(C)
                for iy in range(gh):
                    y = np.float32(y0 + np.float32(i) * bh + np.float32(iy + 0.5) * bh / np.float32(gh))
                    for ix in range(gw):
                        xx = np.float32(x0 + np.float32(j) * bw + np.float32(ix + 0.5) * bw / np.float32(gw))
                        if y < -1.0 or y > H or xx < -1.0 or xx > W:
                            continue
                        yy, xc = max(y, np.float32(0)), max(xx, np.float32(0))
                        yl, xl = int(yy), int(xc)
                        if yl >= H - 1:
                            yh = yl = H - 1
                            yy = np.float32(yl)
                        else:
                            yh = yl + 1
                        if xl >= W - 1:
                            xh = xl = W - 1
                            xc = np.float32(xl)
                        else:
                            xh = xl + 1
                        ly, lx = np.float32(yy - yl), np.float32(xc - xl)
                        hy, hx = np.float32(1) - ly, np.float32(1) - lx
                        acc = acc + float(hy * hx) * x[b, :, yl, xl] + float(hy * lx) * x[b, :, yl, xh] \
                            + float(ly * hx) * x[b, :, yh, xl] + float(ly * lx) * x[b, :, yh, xh]
                out[:, i, j] = acc / count
        outs.append(out)
    return torch.stack(outs) if outs else x.new_zeros(0, C, ph, pw)


class _Storage:
    def __init__(self):
        self.scalars = {}
        self.iter = 0

    def put_scalar(self, k, v, **kw):
        self.scalars[k] = float(v)

    def put_scalars(self, **kw):
        for k, v in kw.items():
            self.put_scalar(k, v)


STORAGE = _Storage()


def configurable(init_func=None, *, from_config=None):
    """Stand-in for detectron2.config.configurable: explicit-kwarg construction only."""
    if init_func is not None:
        return init_func
    return lambda f: f


def setup():
    sys.meta_path.insert(0, _StubFinder())
    # clipcap.py:6 imports GPT-2 classes and AdamW from transformers at module scope; only
    # TransformerMapper is used here, so the whole package is stubbed (GPT-2 is off the hot path).
    _pkg("detectron2", "detectron2")
    for sub in ("utils", "layers", "structures", "config", "modeling", "data", "engine"):
        _pkg(f"detectron2.{sub}", f"detectron2/{sub}")
    for sub in ("backbone", "proposal_generator", "roi_heads", "meta_arch"):
        _pkg(f"detectron2.modeling.{sub}", f"detectron2/modeling/{sub}")
    _pkg("detectron2.modeling.backbone.clipcap", "detectron2/modeling/backbone/clipcap")
    # fvcore bits with real behaviour
    fv = importlib.import_module("fvcore.nn")
    fv.smooth_l1_loss = smooth_l1_loss
    importlib.import_module("fvcore.common.registry").Registry = Registry
    # detectron2.utils.*
    ev = types.ModuleType("detectron2.utils.events")
    ev.get_event_storage = lambda: STORAGE
    sys.modules["detectron2.utils.events"] = ev
    for n in ("comm", "logger", "file_io"):
        sys.modules[f"detectron2.utils.{n}"] = _StubModule(f"detectron2.utils.{n}")
    envm = types.ModuleType("detectron2.utils.env")
    envm.TORCH_VERSION = tuple(int(x) for x in torch.__version__.split(".")[:2])
    sys.modules["detectron2.utils.env"] = envm
    mem = types.ModuleType("detectron2.utils.memory")
    mem.retry_if_cuda_oom = lambda f: f
    sys.modules["detectron2.utils.memory"] = mem
    sys.modules["detectron2.utils"].comm = sys.modules["detectron2.utils.comm"]
    sys.modules["detectron2.utils"].env = envm
    importlib.import_module("detectron2.utils.registry")
    cfgm = sys.modules["detectron2.config"]
    cfgm.configurable = configurable
    # layers
    L = sys.modules["detectron2.layers"]
    ss = importlib.import_module("detectron2.layers.shape_spec")
    wr = importlib.import_module("detectron2.layers.wrappers")
    bn = importlib.import_module("detectron2.layers.batch_norm")
    L.ShapeSpec, L.cat, L.nonzero_tuple, L.cross_entropy, L.Conv2d = ss.ShapeSpec, wr.cat, wr.nonzero_tuple, wr.cross_entropy, wr.Conv2d
    L.FrozenBatchNorm2d, L.get_norm, L.NaiveSyncBatchNorm = bn.FrozenBatchNorm2d, bn.get_norm, bn.NaiveSyncBatchNorm
    L.CNNBlockBase = importlib.import_module("detectron2.layers.blocks").CNNBlockBase
    L.batched_nms = py_batched_nms
    L.nms = py_nms

    class ROIAlign(torch.nn.Module):  # layers/roi_align.py:7-65 semantics over the independent roi_align
        def __init__(self, output_size, spatial_scale, sampling_ratio, aligned=True):
            super().__init__()
            self.a = (output_size, spatial_scale, sampling_ratio, aligned)

        def forward(self, x, rois):
            assert rois.dim() == 2 and rois.size(1) == 5
            return torch_roi_align(x, rois.to(x.dtype), *self.a)

    L.ROIAlign = ROIAlign
    L.RoIPool = L.ROIAlignRotated = L.batched_nms_rotated = _Anything
    sn = types.ModuleType("detectron2.layers.soft_nms")
    sn.batched_soft_nms = _Anything
    sys.modules["detectron2.layers.soft_nms"] = sn
    # structures
    S = sys.modules["detectron2.structures"]
    bx = importlib.import_module("detectron2.structures.boxes")
    S.Boxes, S.BoxMode, S.pairwise_iou, S.pairwise_ioa = bx.Boxes, bx.BoxMode, bx.pairwise_iou, bx.pairwise_ioa
    S.RotatedBoxes = type("RotatedBoxes", (), {})
    S.pairwise_iou_rotated = _Anything
    S.Instances = importlib.import_module("detectron2.structures.instances").Instances
    S.ImageList = importlib.import_module("detectron2.structures.image_list").ImageList
    S.BitMasks = S.PolygonMasks = S.Keypoints = S.heatmaps_to_keypoints = S.ROIMasks = _Anything


def seeded(shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def main():
    setup()
    from cddmsl_amd import synthetic
    out = {}

    # ---- 1. ModifiedResNet / AttentionPool2d  (clip_backbone.py:110-270, 73-107)
    cb = importlib.import_module("detectron2.modeling.backbone.clip_backbone")
    sd = synthetic.make_state_dict(0)
    net = cb.ModifiedResNet(layers=[3, 4, 6, 3], output_dim=1024, heads=32, input_resolution=224, width=64,
                            out_features=["res4", "res5"], freeze_at=2, depth=50, pool_vec=False)
    missing = net.load_state_dict({k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}, strict=True)
    net.train()
    x = seeded((2, 3, 64, 96), 11)
    with torch.no_grad():
        o = net(x)
        x2 = seeded((1, 3, 224, 224), 12)
        o2 = net(x2)
        ap = net.attnpool(o2["res5"])
        xr = seeded((4, 2048, 7, 7), 13)
        ap4 = net.attnpool(xr)
        l4 = net.layer4(seeded((3, 1024, 14, 14), 14))
    np.savez_compressed(os.path.join(HERE, "ref_backbone.npz"),
                        res4_64x96=o["res4"].numpy(), res5_64x96=o["res5"].numpy(),
                        res5_224_slice=o2["res5"][0, ::16].numpy(), res5_224_sum=o2["res5"].double().sum().numpy(),
                        attnpool_224=ap.numpy(), attnpool_rand4=ap4.numpy(), layer4_14=l4[:, ::8].numpy())
    # attnpool gradient wrt input + a parameter
    xr.requires_grad_(True)
    y = net.attnpool(xr)
    w = seeded(tuple(y.shape), 15)
    (y * w).sum().backward()
    np.savez_compressed(os.path.join(HERE, "ref_attnpool_grad.npz"), gx=xr.grad[:, ::32].numpy(),
                        gq=net.attnpool.q_proj.weight.grad[::16, ::16].numpy(),
                        gc=net.attnpool.c_proj.weight.grad[::16, ::16].numpy(),
                        gpos=net.attnpool.positional_embedding.grad[:, ::16].numpy())
    print("backbone ok", o["res4"].shape, o["res5"].shape, float(o["res4"].std()))

    # ---- 2. TransformerMapper + v2l  (clipcap.py:149-163, 714-719)
    cc = importlib.import_module("detectron2.modeling.backbone.clipcap.clipcap")
    msd = synthetic.make_mapper_state_dict(1)
    mp = cc.TransformerMapper(1024, 768, 40, 40, 8)
    mp.load_state_dict(msd, strict=True)
    mp.eval()
    px = seeded((4, 1024), 21)
    px.requires_grad_(True)
    e = cc.v2l(px, mp)
    (e * seeded(tuple(e.shape), 22)).sum().backward()
    np.savez_compressed(os.path.join(HERE, "ref_mapper.npz"), v2l=e.detach().numpy(), gx=px.grad.numpy())
    print("mapper ok", e.shape, sum(p.numel() for p in mp.parameters()) / 1e6)

    # ---- 3. Matcher / pairwise_iou / subsample / Box2Box / anchors
    S = sys.modules["detectron2.structures"]
    mt = importlib.import_module("detectron2.modeling.matcher")
    sm = importlib.import_module("detectron2.modeling.sampling")
    br = importlib.import_module("detectron2.modeling.box_regression")
    ag = importlib.import_module("detectron2.modeling.anchor_generator")
    g = torch.Generator().manual_seed(31)
    gt = torch.rand(5, 4, generator=g) * 100
    gt[:, 2:] = gt[:, :2] + 20 + torch.rand(5, 2, generator=g) * 80
    pr = torch.rand(400, 4, generator=g) * 150
    pr[:, 2:] = pr[:, :2] + 5 + torch.rand(400, 2, generator=g) * 90
    pr[:5] = gt  # exact matches
    pr[7] = pr[6]  # duplicate -> ties
    iou = S.pairwise_iou(S.Boxes(gt), S.Boxes(pr))
    m1 = mt.Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=True)
    a1, l1 = m1(iou)
    m2 = mt.Matcher([0.5], [0, 1], allow_low_quality_matches=False)
    a2, l2 = m2(iou)
    torch.manual_seed(33)
    pos, neg = sm.subsample_labels(l1.clone(), 64, 0.5, 0)
    b2b = br.Box2BoxTransform(weights=(10.0, 10.0, 5.0, 5.0))
    deltas = b2b.get_deltas(pr[:50], gt[a2[:50]])
    dd = seeded((50, 8), 34, 0.5)
    dd[0, 2] = 40.0  # exercises the clamp
    applied = b2b.apply_deltas(dd, pr[:50])
    gen = ag.DefaultAnchorGenerator(sizes=[[32, 64, 128, 256, 512]], aspect_ratios=[[0.5, 1.0, 2.0]], strides=[16], offset=0.0)
    anc = gen([torch.zeros(1, 1, 3, 5)])[0].tensor
    np.savez_compressed(os.path.join(HERE, "ref_boxes.npz"), gt=gt.numpy(), pr=pr.numpy(), iou=iou.numpy(),
                        match_rpn=a1.numpy(), label_rpn=l1.numpy(), match_roi=a2.numpy(), label_roi=l2.numpy(),
                        sub_pos=pos.numpy(), sub_neg=neg.numpy(), deltas=deltas.numpy(), dd=dd.numpy(),
                        applied=applied.numpy(), anchors_3x5=anc.numpy())
    print("boxes ok")

    # ---- 4. FastRCNNOutputLayers  (fast_rcnn.py:368-689)
    fr = importlib.import_module("detectron2.modeling.roi_heads.fast_rcnn")
    L = sys.modules["detectron2.layers"]
    pred = fr.FastRCNNOutputLayers(
        L.ShapeSpec(channels=1024, height=1, width=1), box2box_transform=br.Box2BoxTransform(weights=(10.0, 10.0, 5.0, 5.0)),
        num_classes=20, clip_cls_emb=(True, None, "CLIPRes5ROIHeads", 1024), bg_cls_loss_weight=0.2,
        openset_test=(None, None, 0.01, 0.5), loss_weight={"loss_box_reg": 1.0})
    with torch.no_grad():
        pred.cls_score.weight.copy_(sd["roi_heads.box_predictor.cls_score.weight"])
        pred.bbox_pred.weight.copy_(sd["roi_heads.box_predictor.bbox_pred.weight"])
        pred.bbox_pred.bias.copy_(seeded((80,), 41, 0.1))
    pred.train()
    feats = seeded((48, 1024), 42)
    feats.requires_grad_(True)
    scores, pdeltas = pred(feats)
    gcls = torch.randint(0, 21, (48,), generator=torch.Generator().manual_seed(43))
    gcls[:6] = 20
    pbox = pr[:48].clone()
    gbox = gt[torch.arange(48) % 5]
    inst = S.Instances((200, 200))
    inst.proposal_boxes, inst.gt_boxes, inst.gt_classes = S.Boxes(pbox), S.Boxes(gbox), gcls
    losses = pred.losses((scores, pdeltas), [inst])
    (losses["loss_cls"] + losses["loss_box_reg"]).backward()
    np.savez_compressed(os.path.join(HERE, "ref_fastrcnn.npz"), feats=feats.detach().numpy(), scores=scores.detach().numpy(),
                        deltas=pdeltas.detach().numpy(), gcls=gcls.numpy(), pbox=pbox.numpy(), gbox=gbox.numpy(),
                        bbox_bias=pred.bbox_pred.bias.detach().numpy(),
                        loss_cls=losses["loss_cls"].detach().numpy(), loss_box_reg=losses["loss_box_reg"].detach().numpy(),
                        gfeats=feats.grad.numpy(), stats=np.array([STORAGE.scalars.get("fast_rcnn/cls_accuracy", -1),
                                                                    STORAGE.scalars.get("fast_rcnn/fg_cls_accuracy", -1),
                                                                    STORAGE.scalars.get("fast_rcnn/false_negative", -1)]))
    print("fastrcnn ok", float(losses["loss_cls"]), float(losses["loss_box_reg"]))

    # ---- 5. RPN forward (losses + proposals) on a tiny map  (rpn.py:180-533, proposal_utils.py:22-130)
    pgb = types.ModuleType("detectron2.modeling.proposal_generator.build")  # build.py imports rrpn (rotated, off-path)
    pgb.PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")
    sys.modules["detectron2.modeling.proposal_generator.build"] = pgb
    rp = importlib.import_module("detectron2.modeling.proposal_generator.rpn")
    head = rp.StandardRPNHead(in_channels=1024, num_anchors=15, box_dim=4)
    pfx = "proposal_generator.rpn_head."
    head.load_state_dict({k[len(pfx):]: v for k, v in sd.items() if k.startswith(pfx)})
    rpn = rp.RPN(in_features=["res4"], head=head, anchor_generator=gen,
                 anchor_matcher=mt.Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=True),
                 box2box_transform=br.Box2BoxTransform(weights=(1.0, 1.0, 1.0, 1.0)),
                 batch_size_per_image=256, positive_fraction=0.5, pre_nms_topk=(12000, 6000),
                 post_nms_topk=(2000, 1000), nms_thresh=0.7, min_box_size=0.0, anchor_boundary_thresh=-1.0,
                 loss_weight=1.0, box_reg_loss_type="smooth_l1", smooth_l1_beta=0.0)
    rpn.train()
    feat = seeded((2, 1024, 6, 9), 51)
    feat.requires_grad_(True)
    images = S.ImageList(torch.zeros(2, 3, 96, 144), [(96, 144), (90, 130)])
    gts = []
    gtb = [torch.tensor([[10.0, 12.0, 80.0, 70.0], [60.0, 20.0, 140.0, 90.0]]), torch.tensor([[5.0, 5.0, 50.0, 60.0]])]
    for (h, w), b in zip(images.image_sizes, gtb):
        i = S.Instances((h, w))
        i.gt_boxes, i.gt_classes = S.Boxes(b), torch.zeros(len(b), dtype=torch.int64)
        gts.append(i)
    torch.manual_seed(55)
    props, rl = rpn(images, {"res4": feat}, gts)
    (rl["loss_rpn_cls"] + rl["loss_rpn_loc"]).backward()
    np.savez_compressed(os.path.join(HERE, "ref_rpn.npz"), feat=feat.detach().numpy(),
                        loss_rpn_cls=rl["loss_rpn_cls"].detach().numpy(), loss_rpn_loc=rl["loss_rpn_loc"].detach().numpy(),
                        gfeat=feat.grad.numpy(),
                        boxes0=props[0].proposal_boxes.tensor.numpy(), logits0=props[0].objectness_logits.numpy(),
                        boxes1=props[1].proposal_boxes.tensor.numpy(), logits1=props[1].objectness_logits.numpy(),
                        gt0=gtb[0].numpy(), gt1=gtb[1].numpy())
    print("rpn ok", float(rl["loss_rpn_cls"]), float(rl["loss_rpn_loc"]), len(props[0]), len(props[1]))

    # ---- 6. RoIAlign KATs are literal tables from tests/layers/test_roi_align.py:24-41 (kat_roi_align.json);
    #         here: the independent roi_align on a random multi-image case incl. gradient.
    xin = seeded((2, 6, 9, 11), 61)
    xin.requires_grad_(True)
    rois = torch.tensor([[0, 1.3, 2.1, 60.7, 50.2], [1, 20.0, 10.0, 170.0, 140.0], [0, -8.0, -4.0, 30.0, 200.0],
                         [1, 40.0, 40.0, 41.0, 41.5], [0, 0.0, 0.0, 0.0, 0.0]])
    ro = torch_roi_align(xin, rois, 4, 1.0 / 16, 0, True)
    wgt = seeded(tuple(ro.shape), 62)
    (ro * wgt).sum().backward()
    np.savez_compressed(os.path.join(HERE, "ref_roialign.npz"), x=xin.detach().numpy(), rois=rois.numpy(),
                        out=ro.detach().numpy(), w=wgt.numpy(), gx=xin.grad.numpy())
    print("roialign ok")


if __name__ == "__main__":
    main()

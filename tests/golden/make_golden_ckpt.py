"""Golden vectors for the checkpoint row, produced by the reference's OWN
``align_and_update_state_dicts_for_CLIP`` (checkpoint/clip_model_loading.py:190-343, incl. ``convert_clip_names`` :47-186
and ``convert_basic_clip_names`` :10-44).  That file is pure torch + tabulate (both installed here), so it is loaded
directly from /root/reference by path -- nothing stubbed, nothing copied.

  ref_ckpt_align.json   per case: the model state dict (names + shapes), the checkpoint (names + shapes; every tensor is
                        ``start + arange(numel)`` with a start unique to the tensor, so a value identifies its source
                        element), ``bb_rpn_weights``, and what the reference returned: result name -> flattened values, or the
                        exception type it raised.

The model NAMES are the product's real state-dict keys (CLIP RN50-C4 GeneralizedRCNN incl. the offline backbone); shapes are
shrunk to a few elements (the aligner only compares shapes), so the fixture stays small.

Run here (needs /root/reference):  python tests/golden/make_golden_ckpt.py
"""
import importlib.util
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_FILE = "/root/reference/detectron2/checkpoint/clip_model_loading.py"


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_clip_model_loading", REF_FILE)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def tiny_shape(name, real_shape):
    """a small stand-in shape that still tells tensors of different rank / role apart"""
    r = len(real_shape)
    if r == 0:
        return ()
    base = 2 + (sum(map(ord, name)) % 3)
    return tuple([base] + [1 + (i % 2) for i in range(r - 1)])


def model_names():
    from cddmsl_amd.config import get_cfg
    from cddmsl_amd.modeling import build_model
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "VOC-Experiments", "faster_rcnn_CLIP_R_50_C4.yaml"))
    cfg.MODEL.DEVICE = "cpu"
    m = build_model(cfg)
    return {k: tiny_shape(k, tuple(v.shape)) for k, v in m.state_dict().items()}


def materialise(spec):
    """{name: [shape, start]} -> {name: tensor}"""
    out = {}
    for k, (shape, start) in spec.items():
        n = 1
        for s in shape:
            n *= s
        out[k] = (torch.arange(n, dtype=torch.float32) + float(start)).view(*shape) if shape else torch.tensor(float(start))
    return out


def cases():
    full = model_names()                                     # backbone.*, offline_backbone.*, proposal_generator.*, roi_heads.*, ...
    no_offline = {k: v for k, v in full.items() if not k.startswith("offline_")}
    bb = {k[len("backbone."):]: v for k, v in full.items() if k.startswith("backbone.")}
    out = []
    # 1. OpenAI-CLIP style file (visual tower + text tower) into the model WITH an offline backbone: 'visual.' -> 'backbone.'
    ck, s = {}, 1000
    for k, shp in sorted(bb.items()):
        ck["visual." + k] = [list(shp), s]
        s += 1000
    for k, shp in (("transformer.resblocks.0.attn.in_proj_weight", (3, 2)), ("token_embedding.weight", (5, 2)), ("logit_scale", ()),
                   ("positional_embedding", (4, 2)), ("text_projection", (2, 2)), ("ln_final.weight", (2,))):
        ck[k] = [list(shp), s]
        s += 1000
    out.append({"name": "oai_clip_into_model_with_offline_backbone", "model": {k: list(v) for k, v in full.items()}, "ckpt": ck, "bb_rpn_weights": False})
    # 2. the same file into a model without offline modules: 'visual.' is stripped, longest dotted suffix does the rest
    out.append({"name": "oai_clip_into_model_without_offline_backbone", "model": {k: list(v) for k, v in no_offline.items()}, "ckpt": ck, "bb_rpn_weights": False})
    # 3. second checkpoint (offline modules): a full detector state dict, only backbone / proposal_generator survive, renamed
    ck2, s = {}, 500
    for k, shp in sorted(no_offline.items()):
        ck2[k] = [list(shp), s]
        s += 1000
    out.append({"name": "second_checkpoint_bb_rpn_weights", "model": {k: list(v) for k, v in full.items()}, "ckpt": ck2, "bb_rpn_weights": True})
    # 4. Caffe2-layout blob names: heads re-ordered (bbox_pred drops 4 rows, cls_score moves row 0 last), RPN blobs renamed, one
    #    shape mismatch (left out of the matches, passed through under its converted name)
    tgt = {"roi_heads.box_predictor.bbox_pred.weight": [80, 2], "roi_heads.box_predictor.bbox_pred.bias": [80],
           "roi_heads.box_predictor.cls_score.weight": [21, 3], "roi_heads.box_predictor.cls_score.bias": [21],
           "proposal_generator.rpn_head.conv.weight": [4, 2], "proposal_generator.rpn_head.conv.bias": [4],
           "proposal_generator.rpn_head.anchor_deltas.weight": [60, 1], "proposal_generator.rpn_head.objectness_logits.weight": [15, 1],
           "roi_heads.box_head.fc1.weight": [3, 2], "roi_heads.box_head.fc2.weight": [3, 3], "backbone.stem.conv1.weight": [2, 2]}
    ck4 = {"bbox.pred.weight": [[84, 2], 10000], "bbox.pred.bias": [[84], 20000], "cls.score.weight": [[21, 3], 30000], "cls.score.bias": [[21], 40000],
           "conv.rpn.weight": [[4, 2], 50000], "conv.rpn.bias": [[5], 60000], "rpn.bbox.pred.weight": [[60, 1], 70000],
           "rpn.cls.logits.weight": [[15, 1], 80000], "fc6.weight": [[3, 2], 90000], "fc7.weight": [[3, 3], 100000],
           "visual.stem.conv1.weight": [[2, 2], 110000], "unrelated.blob": [[2], 120000]}
    out.append({"name": "caffe2_layout_heads_and_rpn_blobs", "model": tgt, "ckpt": ck4, "bb_rpn_weights": False})
    # 5. one checkpoint tensor claimed by two model keys
    out.append({"name": "ambiguous_suffix", "model": {"a.conv1.weight": [1], "b.conv1.weight": [1]}, "ckpt": {"conv1.weight": [[1], 7]}, "bb_rpn_weights": False})
    # 6. nothing matches: the converted checkpoint comes back as it is
    out.append({"name": "no_match", "model": {"backbone.layer1.0.conv1.weight": [2, 2]}, "ckpt": {"visual.other.weight": [[2, 2], 9], "fc6.bias": [[3], 30]}, "bb_rpn_weights": False})
    # 7. longest suffix wins over a shorter one
    out.append({"name": "longest_suffix_wins", "model": {"backbone.res2.conv1.weight": [2], "backbone.conv1.weight": [2]},
                "ckpt": {"conv1.weight": [[2], 100], "res2.conv1.weight": [[2], 200]}, "bb_rpn_weights": False})
    return out


def main():
    ref = load_reference()
    rec = []
    for c in cases():
        model = {k: torch.zeros(*shp) if shp else torch.zeros(()) for k, shp in c["model"].items()}
        ckpt = materialise({k: (tuple(v[0]), v[1]) for k, v in c["ckpt"].items()})
        try:
            res = ref.align_and_update_state_dicts_for_CLIP(model, dict(ckpt), bb_rpn_weights=c["bb_rpn_weights"])
            c["expect"] = {k: [list(v.shape), [float(x) for x in v.flatten().tolist()]] for k, v in res.items()}
            c["raises"] = None
        except Exception as e:           # noqa: BLE001 -- the exception type is the recorded behaviour
            c["expect"], c["raises"] = None, type(e).__name__
        rec.append(c)
        print(c["name"], "->", c["raises"] or f"{len(c['expect'])} tensors")
    with open(os.path.join(HERE, "ref_ckpt_align.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden_ckpt.py", "reference": "detectron2/checkpoint/clip_model_loading.py:190-343", "cases": rec}, f)


if __name__ == "__main__":
    main()

"""Golden vectors for the inference / evaluation rows, produced by the reference's OWN leaf modules
imported from /root/reference with the package __init__s bypassed (same harness as make_golden.py):

  ref_inference.npz   fast_rcnn_inference_single_image (fast_rcnn.py:130-209; torchvision batched_nms replaced by the
                      harness' independent python NMS) and detector_postprocess (postprocessing.py:9-75) on seeded inputs
  ref_voc_eval.json   voc_eval / voc_ap (pascal_voc_evaluation.py:166-313) on a seeded synthetic annotation set, VOC07
                      11-point and area metrics, IoU thresholds .50 and .75 -- the annotation set and the detections are
                      stored in the fixture, the XML files are regenerated from it by the test

Run here (needs /root/reference): python tests/golden/make_golden_eval.py
The reference's numpy calls ``np.bool`` (removed in numpy 1.24); the harness restores that alias, nothing else.
"""
import importlib
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

CLASSES = ["aeroplane", "bicycle", "bird", "boat"]


def synth_dataset(seed=5, nimg=12):
    g = np.random.RandomState(seed)
    data = {}
    for i in range(nimg):
        objs = []
        for _ in range(g.randint(0, 5)):
            x0, y0 = g.randint(1, 300), g.randint(1, 200)
            objs.append({"name": CLASSES[g.randint(0, len(CLASSES))], "difficult": int(g.rand() < 0.2),
                         "bbox": [int(x0), int(y0), int(x0 + g.randint(8, 180)), int(y0 + g.randint(8, 160))]})
        data[f"img{i:03d}"] = objs
    return data


def write_voc(root, data, split="test"):
    os.makedirs(os.path.join(root, "Annotations"), exist_ok=True)
    os.makedirs(os.path.join(root, "ImageSets", "Main"), exist_ok=True)
    for name, objs in data.items():
        xml = ["<annotation>"]
        for o in objs:
            xml.append("<object><name>%s</name><pose>Unspecified</pose><truncated>0</truncated><difficult>%d</difficult>"
                       "<bndbox><xmin>%d</xmin><ymin>%d</ymin><xmax>%d</xmax><ymax>%d</ymax></bndbox></object>"
                       % (o["name"], o["difficult"], *o["bbox"]))
        xml.append("</annotation>")
        with open(os.path.join(root, "Annotations", name + ".xml"), "w") as f:
            f.write("".join(xml))
    with open(os.path.join(root, "ImageSets", "Main", split + ".txt"), "w") as f:
        f.write("\n".join(data.keys()) + "\n")


def synth_detections(data, seed=6):
    """per class: lines 'id score x0 y0 x1 y1' -- jittered ground truth (incl. duplicates) + random false positives"""
    g = np.random.RandomState(seed)
    lines = {c: [] for c in CLASSES}
    for name, objs in data.items():
        for o in objs:
            for _ in range(g.randint(0, 3)):
                j = g.randn(4) * 6
                b = [o["bbox"][k] + j[k] for k in range(4)]
                lines[o["name"]].append(f"{name} {g.rand():.3f} {b[0]:.1f} {b[1]:.1f} {b[2]:.1f} {b[3]:.1f}")
        for _ in range(g.randint(0, 3)):
            x0, y0 = g.rand() * 300, g.rand() * 200
            c = CLASSES[g.randint(0, len(CLASSES))]
            lines[c].append(f"{name} {round(g.rand(), 1):.3f} {x0:.1f} {y0:.1f} {x0 + 50:.1f} {y0 + 40:.1f}")   # coarse scores: ties
    return lines


def main():
    mg.setup()
    np.bool = bool                                                   # numpy >= 1.24 (see module docstring)
    fio = types.ModuleType("detectron2.utils.file_io")
    fio.PathManager = type("PathManager", (), {"open": staticmethod(open), "get_local_path": staticmethod(lambda p: p)})
    sys.modules["detectron2.utils.file_io"] = fio
    dm = sys.modules["detectron2.data"]
    dm.MetadataCatalog = mg._Anything
    evm = types.ModuleType("detectron2.evaluation.evaluator")
    evm.DatasetEvaluator = type("DatasetEvaluator", (), {})
    mg._pkg("detectron2.evaluation", "detectron2/evaluation")
    sys.modules["detectron2.evaluation.evaluator"] = evm
    pv = importlib.import_module("detectron2.evaluation.pascal_voc_evaluation")

    # ---- voc_eval
    data = synth_dataset()
    dets = synth_detections(data)
    res = {}
    with tempfile.TemporaryDirectory() as root:
        write_voc(root, data)
        for c in CLASSES:
            with open(os.path.join(root, c + ".txt"), "w") as f:
                f.write("\n".join(dets[c]))
        for c in CLASSES:
            for thr in (0.5, 0.75):
                for m07 in (True, False):
                    pv.parse_rec.cache_clear()
                    _, _, ap = pv.voc_eval(os.path.join(root, "{}.txt"), os.path.join(root, "Annotations", "{}.xml"),
                                           os.path.join(root, "ImageSets", "Main", "test.txt"), c, ovthresh=thr, use_07_metric=m07)
                    res[f"{c}|{thr}|{int(m07)}"] = float(ap)
    json.dump({"classes": CLASSES, "data": data, "dets": dets, "ap": res}, open(os.path.join(HERE, "ref_voc_eval.json"), "w"), indent=0)
    print("voc_eval ok", {k: round(v, 4) for k, v in list(res.items())[:4]})

    # ---- fast_rcnn_inference_single_image + detector_postprocess
    fr = importlib.import_module("detectron2.modeling.roi_heads.fast_rcnn")
    pp = importlib.import_module("detectron2.modeling.postprocessing")
    S = sys.modules["detectron2.structures"]
    g = torch.Generator().manual_seed(71)
    R, K = 60, 5
    ctr = torch.rand(R, 2, generator=g) * torch.tensor([300.0, 200.0])
    wh = torch.rand(R, 2, generator=g) * 120 + 10
    base = torch.cat([ctr - wh / 2, ctr + wh / 2], dim=1)
    boxes = (base[:, None, :] + torch.randn(R, K, 4, generator=g) * 4).reshape(R, K * 4)
    boxes[:10] = boxes[10:20] + torch.randn(10, K * 4, generator=g)             # near-duplicates for the NMS to remove
    scores = torch.softmax(torch.randn(R, K + 1, generator=g) * 2.5, dim=1)
    boxes[3, 2] = float("inf")                                                  # a non-finite row (fast_rcnn.py:158-162)
    inst, kept = fr.fast_rcnn_inference_single_image(boxes.clone(), scores.clone(), (200, 300), 0.05, 0.5, False, "gaussian", 0.5,
                                                     0.001, 20, scores.clone(), False)
    post = pp.detector_postprocess(S.Instances((200, 300), pred_boxes=S.Boxes(inst.pred_boxes.tensor.clone()), scores=inst.scores.clone(),
                                               pred_classes=inst.pred_classes.clone()), 333, 480)
    np.savez_compressed(os.path.join(HERE, "ref_inference.npz"), boxes=boxes.numpy(), scores=scores.numpy(),
                        det_boxes=inst.pred_boxes.tensor.numpy(), det_scores=inst.scores.numpy(), det_classes=inst.pred_classes.numpy(),
                        det_kept=kept.numpy(), post_boxes=post.pred_boxes.tensor.numpy(), post_scores=post.scores.numpy(),
                        post_classes=post.pred_classes.numpy())
    print("inference ok", len(inst), len(post))


if __name__ == "__main__":
    main()

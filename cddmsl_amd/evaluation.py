"""Pascal VOC detection evaluation (CPU / numpy) -- detectron2/evaluation/pascal_voc_evaluation.py:20-313.

``PascalVOCDetectionEvaluator`` keeps the reference's protocol: ``reset()``, ``process(inputs, outputs)`` per batch,
``evaluate()`` on every rank (predictions are gathered to rank 0) returning
``{"bbox": {"AP", "AP50", "AP75", "AP50-<class>"...}}`` with AP in percent: VOC07 11-point AP for year 2007, area under
the monotone precision envelope otherwise, averaged over IoU thresholds 0.50:0.05:0.95 for "AP".

Details that decide the numbers and are therefore kept exactly:
* a detection is serialised the way the reference writes its result files -- score to 3 decimals, box to 1 decimal, with
  +1 on xmin/ymin (the inverse of the loader's -1, datasets/pascal_voc.py) -- and evaluated from those rounded values;
* detections are ranked with ``np.argsort(-confidence)`` on the per-class list in processing order (ties between equal
  rounded scores fall where that call puts them);
* overlaps use the devkit's inclusive pixel convention (+1 on widths/heights), a match needs IoU strictly above the threshold,
  "difficult" ground truth neither counts as positive nor as false positive, a second match of one box is a false positive.
"""
import os
import xml.etree.ElementTree as ET
from collections import OrderedDict, defaultdict

import numpy as np
import torch

VOC_CLASS_NAMES = ("aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable", "dog",
                   "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor")   # datasets/pascal_voc.py:19-24


def read_voc_objects(xml_path):
    """One annotation file -> list of {name, difficult, bbox [xmin, ymin, xmax, ymax] (1-based ints as stored)}"""
    out = []
    for obj in ET.parse(xml_path).getroot().findall("object"):
        bb = obj.find("bndbox")
        out.append({"name": obj.find("name").text, "difficult": int(obj.find("difficult").text),
                    "bbox": [int(bb.find(k).text) for k in ("xmin", "ymin", "xmax", "ymax")]})
    return out


def average_precision(rec, prec, use_07_metric):
    """VOC AP from a recall / precision curve (pascal_voc_evaluation.py:166-196)"""
    if use_07_metric:
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            sel = rec >= t
            ap += (np.max(prec[sel]) if sel.any() else 0.0) / 11.0
        return ap
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]            # precision envelope
    step = np.where(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[step + 1] - mrec[step]) * mpre[step + 1]))


def class_ap(image_ids, confidence, boxes, gt_by_image, ovthresh, use_07_metric):
    """AP of one class at one IoU threshold.  ``gt_by_image[id] = (bbox [G,4] float, difficult [G] bool)`` for that class."""
    npos = sum(int((~d).sum()) for _, d in gt_by_image.values())
    claimed = {k: np.zeros(len(d), dtype=bool) for k, (_, d) in gt_by_image.items()}
    order = np.argsort(-confidence)
    tp, fp = np.zeros(len(order)), np.zeros(len(order))
    for rank, d in enumerate(order):
        gtb, diff = gt_by_image[image_ids[d]]
        best, arg = -np.inf, -1
        if gtb.size:
            bb = boxes[d]
            iw = np.maximum(np.minimum(gtb[:, 2], bb[2]) - np.maximum(gtb[:, 0], bb[0]) + 1.0, 0.0)
            ih = np.maximum(np.minimum(gtb[:, 3], bb[3]) - np.maximum(gtb[:, 1], bb[1]) + 1.0, 0.0)
            inter = iw * ih
            union = (bb[2] - bb[0] + 1.0) * (bb[3] - bb[1] + 1.0) + (gtb[:, 2] - gtb[:, 0] + 1.0) * (gtb[:, 3] - gtb[:, 1] + 1.0) - inter
            ov = inter / union
            arg = int(np.argmax(ov))
            best = ov[arg]
        if best > ovthresh:
            if not diff[arg]:
                if not claimed[image_ids[d]][arg]:
                    tp[rank] = 1.0
                    claimed[image_ids[d]][arg] = True
                else:
                    fp[rank] = 1.0
        else:
            fp[rank] = 1.0
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(npos) if npos else tp * np.nan          # no positives: the reference divides by zero as well
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return average_precision(rec, prec, use_07_metric)


class PascalVOCDetectionEvaluator:
    def __init__(self, dirname, split, year, class_names=VOC_CLASS_NAMES, target_classnames=None):
        assert year in (2007, 2012), year
        self.anno = os.path.join(dirname, "Annotations", "{}.xml")
        self.image_set = os.path.join(dirname, "ImageSets", "Main", split + ".txt")
        self.class_names = list(class_names)
        self.target_classnames = list(target_classnames) if target_classnames is not None else list(class_names)
        self.is_2007 = year == 2007
        self.reset()

    def reset(self):
        self._predictions = defaultdict(list)        # class id -> [(image_id, score, xmin, ymin, xmax, ymax)] as written

    def process(self, inputs, outputs):
        for inp, out in zip(inputs, outputs):
            inst = out["instances"]
            boxes = inst.pred_boxes.tensor.detach().cpu().numpy()
            scores = inst.scores.detach().cpu().tolist()
            classes = inst.pred_classes.detach().cpu().tolist()
            for (x0, y0, x1, y1), s, c in zip(boxes, scores, classes):
                # what the reference's result line "{id} {score:.3f} {xmin:.1f} {ymin:.1f} {xmax:.1f} {ymax:.1f}" reads back as
                self._predictions[c].append((str(inp["image_id"]), float(f"{s:.3f}"), float(f"{x0 + 1:.1f}"), float(f"{y0 + 1:.1f}"),
                                             float(f"{x1:.1f}"), float(f"{y1:.1f}")))

    def _gathered(self):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return self._predictions
        parts = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
        dist.gather_object(dict(self._predictions), parts, dst=0)
        if dist.get_rank() != 0:
            return None
        merged = defaultdict(list)
        for part in parts:
            for c, lines in part.items():
                merged[c].extend(lines)
        return merged

    def evaluate(self):
        preds = self._gathered()
        if preds is None:
            return None
        with open(self.image_set) as f:
            image_names = [l.strip() for l in f.readlines()]
        recs = {n: read_voc_objects(self.anno.format(n)) for n in image_names}
        aps = defaultdict(list)
        for cid, cname in enumerate(self.class_names):
            if cname not in self.target_classnames:
                continue
            gt = {}
            for n in image_names:
                objs = [o for o in recs[n] if o["name"] == cname]
                gt[n] = (np.array([o["bbox"] for o in objs], dtype=float).reshape(-1, 4), np.array([o["difficult"] for o in objs], dtype=bool))
            lines = preds.get(cid, [])
            ids = [l[0] for l in lines]
            conf = np.array([l[1] for l in lines], dtype=float)
            bbs = np.array([l[2:] for l in lines], dtype=float).reshape(-1, 4)
            for thresh in range(50, 100, 5):
                aps[thresh].append(class_ap(ids, conf, bbs, gt, thresh / 100.0, self.is_2007) * 100)
        mean = {t: float(np.mean(v)) for t, v in aps.items()}
        ret = OrderedDict()
        ret["bbox"] = {"AP": float(np.mean(list(mean.values()))), "AP50": mean[50], "AP75": mean[75]}
        for i, name in enumerate(self.target_classnames):
            ret["bbox"]["AP50-" + name] = aps[50][i]
        return ret


@torch.no_grad()
def inference_on_dataset(model, data_loader, evaluator):
    """evaluation/evaluator.py:85-181 without the timing log: eval mode, one ``process`` per batch, then ``evaluate``."""
    was_training = model.training
    model.eval()
    evaluator.reset()
    for inputs in data_loader:
        evaluator.process(inputs, model(inputs))
    model.train(was_training)
    return evaluator.evaluate()


def run_eval_only(model, cfg, args, rank=0, world=1, return_results=False):
    """tools/train_caption_consistency.py:143-152 (``--eval-only``): VOC-style test set under ``args.voc_root`` -> AP dict
    printed by rank 0.  Every rank evaluates its shard of the test set (``world`` has to be the process group's size: the
    evaluator merges all ranks' detections, so unsharded ranks would count every detection ``world`` times).  Returns the
    process exit code (or the result dict: rank 0, ``return_results``)."""
    import torch.distributed as dist
    ws = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    assert world == ws, f"run_eval_only: world={world} but the process group has {ws} ranks"
    from .data import build_detection_test_loader, load_voc_instances
    dicts = load_voc_instances(args.voc_root, args.voc_split, VOC_CLASS_NAMES[: cfg.MODEL.ROI_HEADS.NUM_CLASSES])
    loader = build_detection_test_loader(cfg, dicts, batch_size=1, rank=rank, world=world, device=cfg.MODEL.DEVICE)
    ev = PascalVOCDetectionEvaluator(args.voc_root, args.voc_split, args.voc_year, VOC_CLASS_NAMES[: cfg.MODEL.ROI_HEADS.NUM_CLASSES])
    res = inference_on_dataset(model, loader, ev)
    if rank == 0:
        print({k: round(v, 4) for k, v in res["bbox"].items()})
    return res if return_results else 0

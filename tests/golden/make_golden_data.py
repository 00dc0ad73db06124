"""Generates tests/golden/ref_data_transforms.npz by running the REFERENCE's own data-side code (read from /root/reference, never
copied; runs only in the build container) -- SURVEY.md 8(f)2, the paired VOC pipeline's geometry:

  * ``ResizeShortestEdge.get_transform`` (data/transforms/augmentation_impl.py:149-199): the (new_h, new_w) rule,
  * ``ResizeTransform.apply_image / apply_coords`` (data/transforms/transform.py:94-152): PIL bilinear on uint8, box scaling,
  * ``RandomFlip.get_transform`` (:95-127),
  * ``transform_instance_annotations`` / ``annotations_to_instances`` / ``filter_empty_instances`` (data/detection_utils.py:253-330,
    378-445,476-503): box clip to the image, Instances construction, empty-box filter,
  applied the way ``DatasetMapper.__call__`` applies them (data/dataset_mapper.py:126-217): ONE sampled transform list for the
  image, its domain twin and the boxes.

Un-vendored third-party code is replaced by small stand-ins with the published behaviour (parity unpinned for THOSE lines only):
fvcore's ``Transform`` base class (``_set_attributes``, ``apply_box`` = apply_coords on the four corners, then min / max),
``HFlipTransform`` (image[:, ::-1]; x -> width - x), ``NoOpTransform``, ``TransformList``.

usage:  python tests/golden/make_golden_data.py
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


# ---------------------------------------------------------------- fvcore.transforms.transform stand-ins (published behaviour)
class Transform:
    @classmethod
    def register_type(cls, data_type, func):          # (fvcore: per-data-type handlers; the rotated-box ones registered by transform.py:347-350 are unused here)
        setattr(cls, "apply_" + data_type, func)

    def _set_attributes(self, params=None):
        if params:
            for k, v in params.items():
                if k != "self" and not k.startswith("_"):
                    setattr(self, k, v)

    def apply_box(self, box):
        idxs = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
        coords = np.asarray(box).reshape(-1, 4)[:, idxs].reshape(-1, 2)
        coords = self.apply_coords(coords).reshape((-1, 4, 2))
        return np.concatenate((coords.min(axis=1), coords.max(axis=1)), axis=1)


class NoOpTransform(Transform):
    def apply_image(self, img):
        return img

    def apply_coords(self, coords):
        return coords


class HFlipTransform(Transform):
    def __init__(self, width):
        super().__init__()
        self._set_attributes(locals())

    def apply_image(self, img):
        return np.flip(img, axis=1)

    def apply_coords(self, coords):
        coords[:, 0] = self.width - coords[:, 0]
        return coords


class TransformList(Transform):
    def __init__(self, transforms):
        super().__init__()
        self.transforms = list(transforms)

    def apply_image(self, img):
        for t in self.transforms:
            img = t.apply_image(img)
        return img

    def apply_coords(self, coords):
        for t in self.transforms:
            coords = t.apply_coords(coords)
        return coords

    def apply_box(self, box):
        for t in self.transforms:
            box = t.apply_box(box)
        return box


def setup_data():
    mg.setup()
    from PIL import Image
    if not hasattr(Image, "LINEAR"):           # transform.py:46 default argument (unused here; Pillow >= 10 dropped the alias of BILINEAR)
        Image.LINEAR = Image.BILINEAR
    ft = importlib.import_module("fvcore.transforms.transform")
    for c in (Transform, NoOpTransform, HFlipTransform, TransformList):
        setattr(ft, c.__name__, c)
    mg._pkg("detectron2.data.transforms", "detectron2/data/transforms")
    tr = importlib.import_module("detectron2.data.transforms.transform")
    ai = importlib.import_module("detectron2.data.transforms.augmentation_impl")
    T = sys.modules["detectron2.data.transforms"]
    T.TransformList, T.ResizeTransform = TransformList, tr.ResizeTransform
    cat = types.ModuleType("detectron2.data.catalog")
    cat.MetadataCatalog = mg._Anything
    sys.modules["detectron2.data.catalog"] = cat
    S = sys.modules["detectron2.structures"]
    S.polygons_to_bitmask = mg._Anything
    du = importlib.import_module("detectron2.data.detection_utils")
    return tr, ai, du, S


def main():
    tr, ai, du, S = setup_data()
    g = np.random.RandomState(5)
    out = {}
    cases = [(90, 120, 96, 128), (130, 80, 64, 128), (333, 500, 800, 1333), (500, 375, 640, 1333), (600, 1800, 800, 1333), (48, 48, 32, 1000)]
    sizes = []
    for i, (h, w, size, max_size) in enumerate(cases):
        aug = ai.ResizeShortestEdge([size], max_size, "choice")
        img = g.randint(0, 256, (h, w, 3), dtype=np.uint8)
        twin = (255 - img).astype(np.uint8)
        t_resize = aug.get_transform(img)
        sizes.append((h, w, size, max_size, t_resize.new_h, t_resize.new_w))
        if h * w <= 200 * 200:                                  # pixels for the small cases only (fixture size)
            out[f"img{i}"], out[f"resized{i}"] = img, t_resize.apply_image(img)
            out[f"twin_resized{i}"] = t_resize.apply_image(twin)
        # the mapper's composition: resize, then (for odd cases) a horizontal flip; boxes through transform_instance_annotations
        flip = HFlipTransform(t_resize.new_w) if i % 2 else NoOpTransform()
        tl = TransformList([t_resize, flip])
        boxes = np.stack([g.uniform(-5, w * 0.7, 6), g.uniform(-5, h * 0.7, 6), np.zeros(6), np.zeros(6)], axis=1)
        boxes[:, 2] = boxes[:, 0] + g.uniform(0.5, w * 0.6, 6)
        boxes[:, 3] = boxes[:, 1] + g.uniform(0.5, h * 0.6, 6)
        boxes[5] = [w + 3.0, 2.0, w + 9.0, 8.0]                 # entirely outside the image: clipped to an empty box and filtered
        annos = [{"bbox": b.tolist(), "bbox_mode": S.BoxMode.XYXY_ABS, "category_id": int(c)} for b, c in zip(boxes, g.randint(0, 20, 6))]
        image_size = (t_resize.new_h, t_resize.new_w)
        annos_t = [du.transform_instance_annotations(dict(a), tl, image_size) for a in annos]
        inst = du.filter_empty_instances(du.annotations_to_instances(annos_t, image_size))
        out[f"boxes_in{i}"], out[f"classes_in{i}"] = boxes, np.array([a["category_id"] for a in annos])
        out[f"boxes_out{i}"], out[f"classes_out{i}"] = inst.gt_boxes.tensor.numpy(), inst.gt_classes.numpy()
        out[f"flip{i}"] = np.array(int(i % 2))
        if f"img{i}" in out:
            out[f"final{i}"] = np.ascontiguousarray(tl.apply_image(img))
    out["sizes"] = np.array(sizes)
    # RandomFlip's draw: do = np.random.uniform() < prob  (augmentation.py _rand_range)
    np.random.seed(123)
    rf = ai.RandomFlip(prob=0.5, horizontal=True, vertical=False)
    dummy = np.zeros((4, 6, 3), np.uint8)
    out["flip_draws"] = np.array([isinstance(rf.get_transform(dummy), HFlipTransform) for _ in range(32)])
    # the draw ORDER of one sample (dataset_mapper.py:149-153 via detection_utils.build_augmentation :590-614): short-edge choice, then flip
    np.random.seed(321)
    rs = ai.ResizeShortestEdge(list(range(480, 801, 32)), 1333, "choice")
    seq = []
    im = np.zeros((375, 500, 3), np.uint8)
    for _ in range(16):
        t1 = rs.get_transform(im)
        t2 = rf.get_transform(im)
        seq.append((t1.new_h, t1.new_w, int(isinstance(t2, HFlipTransform))))
    out["draw_sequence"] = np.array(seq)
    np.savez_compressed(os.path.join(HERE, "ref_data_transforms.npz"), **out)
    print("data transforms ok", out["sizes"].tolist(), out["flip_draws"].astype(int).tolist())


MINI_VOC = {   # file id -> (width, height, [(class, difficult, xmin, ymin, xmax, ymax)])
    "000005": (120, 90, [("chair", 0, 5, 3, 40, 50), ("person", 1, 12, 8, 49, 54)]),
    "000007": (80, 130, [("car", 0, 1, 1, 80, 130)]),
    "000012": (100, 100, []),
    "2008_000003": (64, 48, [("train", 0, 10, 5, 60, 40), ("tvmonitor", 0, 2, 2, 9, 9), ("person", 0, 30, 1, 44, 48)]),
}


def write_mini_voc(root):
    """the XML / split files both sides read (no images: the loaders only build paths to them)"""
    for year in ("VOC2007", "VOC2012"):
        base = os.path.join(root, "VOCdevkit", year)
        os.makedirs(os.path.join(base, "Annotations"), exist_ok=True)
        os.makedirs(os.path.join(base, "ImageSets", "Main"), exist_ok=True)
        for fid, (w, h, objs) in MINI_VOC.items():
            body = "".join("<object><name>%s</name><pose>Unspecified</pose><truncated>0</truncated><difficult>%d</difficult>"
                           "<bndbox><xmin>%d</xmin><ymin>%d</ymin><xmax>%d</xmax><ymax>%d</ymax></bndbox></object>" % o for o in objs)
            with open(os.path.join(base, "Annotations", fid + ".xml"), "w") as f:
                f.write("<annotation><folder>%s</folder><size><width>%d</width><height>%d</height><depth>3</depth></size>%s</annotation>" % (year, w, h, body))
        for split in ("trainval", "test"):
            with open(os.path.join(base, "ImageSets", "Main", split + ".txt"), "w") as f:
                f.write("\n".join(MINI_VOC) + "\n")


def voc_dicts_golden():
    """``load_voc_DG_instances`` (data/datasets/pascal_voc.py:98-172): dataset dicts of the paired VOC + domain-twin loader"""
    import json
    import tempfile
    setup_data()
    if not hasattr(np, "str"):                    # pascal_voc.py:36,113 ``dtype=np.str`` (NumPy >= 1.24 dropped the alias of str)
        np.str = str
    fio = sys.modules["detectron2.utils.file_io"]
    fio.PathManager = types.SimpleNamespace(open=open, get_local_path=lambda p: p)
    dd = sys.modules["detectron2.data"]
    dd.DatasetCatalog = dd.MetadataCatalog = mg._Anything
    mg._pkg("detectron2.data.datasets", "detectron2/data/datasets")
    pv = importlib.import_module("detectron2.data.datasets.pascal_voc")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        write_mini_voc(tmp)
        cwd = os.getcwd()
        os.chdir(tmp)                             # relative paths in the fixture
        try:
            for year in ("VOC2007", "VOC2012"):
                for split, dt in (("trainval", "clipart"), ("trainval", None), ("test", "clipart")):
                    dicts = pv.load_voc_DG_instances(os.path.join("VOCdevkit", year), split, pv.CLASS_NAMES, dt)
                    for d in dicts:
                        for a in d["annotations"]:
                            a["bbox_mode"] = int(a["bbox_mode"])
                    out["%s|%s|%s" % (year, split, dt)] = dicts
        finally:
            os.chdir(cwd)
    json.dump({"mini_voc": {k: [v[0], v[1], [list(o) for o in v[2]]] for k, v in MINI_VOC.items()}, "dicts": out},
              open(os.path.join(HERE, "ref_voc_dicts.json"), "w"), indent=0)
    print("voc dicts ok", {k: len(v) for k, v in out.items()})


def samplers_golden():
    """``TrainingSampler`` (data/samplers/distributed_sampler.py:12-54: one seeded permutation stream shared by all ranks, rank r
    takes indices[r::W]) and ``AspectRatioGroupedDataset`` (data/common.py:152-186: two buckets, a batch leaves when its bucket is full)"""
    import itertools
    import json
    setup_data()
    comm = sys.modules["detectron2.utils.comm"]
    mg._pkg("detectron2.data.samplers", "detectron2/data/samplers")
    ser = types.ModuleType("detectron2.utils.serialize")
    ser.PicklableWrapper = mg._Anything
    sys.modules["detectron2.utils.serialize"] = ser
    ds = importlib.import_module("detectron2.data.samplers.distributed_sampler")
    cm = importlib.import_module("detectron2.data.common")
    out = {"sampler": {}}
    for world in (1, 2, 3):
        for rank in range(world):
            comm.get_rank, comm.get_world_size = (lambda r=rank: r), (lambda w=world: w)
            for size, seed, shuffle in ((11, 7, True), (5, 0, True), (4, 3, False)):
                sm = ds.TrainingSampler(size, shuffle=shuffle, seed=seed)
                out["sampler"]["%d|%d|%d|%d|%d" % (world, rank, size, seed, int(shuffle))] = list(itertools.islice(iter(sm), 40))
    g = np.random.RandomState(9)
    items = [{"id": i, "width": int(w), "height": int(h)} for i, (w, h) in enumerate(zip(g.randint(50, 200, 37), g.randint(50, 200, 37)))]
    items[3]["width"] = items[3]["height"]                       # a square image goes with the "w <= h" bucket
    out["items"] = items
    out["groups"] = {str(b): [[d["id"] for d in batch] for batch in cm.AspectRatioGroupedDataset(items, b)] for b in (1, 2, 4)}
    json.dump(out, open(os.path.join(HERE, "ref_samplers.json"), "w"))
    print("samplers ok", len(out["sampler"]), {k: len(v) for k, v in out["groups"].items()})


if __name__ == "__main__":
    if "samplers" in sys.argv[1:]:
        samplers_golden()
    elif "voc" in sys.argv[1:]:
        voc_dicts_golden()
    else:
        main()

"""Paired (labeled image + domain-translated twin) VOC data pipeline -> device batches.

Follows the reference's input side:
  * dataset dicts: ``load_voc_DG_instances`` data/datasets/pascal_voc.py:98-171 (boxes XYXY with xmin/ymin - 1, "difficult"
    objects kept, twin at ``<dirname>/../<dt_data>/<VOC2007|VOC2012>/JPEGImages/<id>.jpg`` for the training splits);
  * mapper: ``DatasetMapper.__call__`` data/dataset_mapper.py:126-217 -- read both images (RGB/BGR, EXIF orientation), ONE
    sampled transform list (``ResizeShortestEdge`` + ``RandomFlip``, detection_utils.py:590-614, augmentation_impl.py:149-199)
    applied to both images and to the boxes, boxes clipped to the image, empty boxes dropped (detection_utils.py:257-317,393-430);
  * sampler / batching: ``TrainingSampler`` samplers/distributed_sampler.py:12-58 (one seeded infinite permutation stream, rank r
    takes every world-th index), ``AspectRatioGroupedDataset`` data/common.py:152-186 (a batch holds only landscape or only
    portrait images), per-rank batch = IMS_PER_BATCH / world (data/build.py:280-287);
  * batch item: ``{"image": u8 [3,H,W], "image_trgt": u8 [3,H,W], "instances": Instances(gt_boxes, gt_classes), "height",
    "width", "file_name", "image_id"}`` -- the contract ``GeneralizedRCNN.forward`` reads.

Host -> device: images leave the loader as pinned uint8 tensors and are copied with ``non_blocking=True``; normalisation and
padding happen on the GPU (``cddmsl_preprocess``).  Resizing is PIL bilinear on uint8, exactly the call the reference makes
(``Image.resize((w, h), BILINEAR)``, transforms/transform.py ResizeTransform); the reference's transform classes derive from
fvcore, which is not installed here, so their numerics are pinned by that shared PIL call, not by a fixture (parity unpinned).
"""
import os
import xml.etree.ElementTree as ET
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .evaluation import VOC_CLASS_NAMES
from .structures import Boxes, Instances


# ------------------------------------------------------------------------------------------------ dataset dicts
def load_voc_instances(dirname: str, split: str, class_names: Sequence[str] = VOC_CLASS_NAMES, dt_data: Optional[str] = None) -> List[Dict]:
    with open(os.path.join(dirname, "ImageSets", "Main", split + ".txt")) as f:
        fileids = [l.strip() for l in f if l.strip()]
    paired = split in ("train", "trainval") and dt_data is not None
    voc_dir = "VOC2007" if "VOC2007" in os.path.join(dirname, "JPEGImages") else "VOC2012"
    dicts = []
    for fid in fileids:
        root = ET.parse(os.path.join(dirname, "Annotations", fid + ".xml")).getroot()
        r = {"file_name": os.path.join(dirname, "JPEGImages", fid + ".jpg"), "image_id": fid,
             "height": int(root.findall("./size/height")[0].text), "width": int(root.findall("./size/width")[0].text)}
        if paired:
            r["data_dt_file_name"] = os.path.join(dirname, "..", dt_data, voc_dir, "JPEGImages", fid + ".jpg")
        anns = []
        for obj in root.findall("object"):
            bb = obj.find("bndbox")
            box = [float(bb.find(k).text) for k in ("xmin", "ymin", "xmax", "ymax")]
            box[0] -= 1.0                       # 1-based inclusive pixel indices -> continuous coordinates
            box[1] -= 1.0
            anns.append({"category_id": list(class_names).index(obj.find("name").text), "bbox": box})
        r["annotations"] = anns
        dicts.append(r)
    return dicts


# ------------------------------------------------------------------------------------------------ image I/O + transforms
def read_image(path, fmt="RGB"):
    """detection_utils.py:171-190: PIL decode, EXIF orientation, RGB or BGR uint8 HWC"""
    from PIL import Image, ImageOps
    with open(path, "rb") as f:
        img = Image.open(f)
        img = ImageOps.exif_transpose(img)
        arr = np.asarray(img.convert("RGB"))
    return arr[:, :, ::-1] if fmt == "BGR" else arr


def shortest_edge_size(h, w, size, max_size):
    """augmentation_impl.py:188-198 -> (new_h, new_w)"""
    scale = size * 1.0 / min(h, w)
    newh, neww = (size, scale * w) if h < w else (scale * h, size)
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)


def resize_image(img, newh, neww):
    """ResizeTransform.apply_image for uint8: PIL bilinear"""
    from PIL import Image
    if img.shape[:2] == (newh, neww):
        return img
    return np.asarray(Image.fromarray(img).resize((neww, newh), Image.BILINEAR))


class DatasetMapper:
    """Dataset dict -> model input dict; the random draws (short-edge choice, flip) come from ``rng``."""

    def __init__(self, cfg, is_train=True, rng: Optional[np.random.RandomState] = None):
        i = cfg.INPUT
        self.is_train, self.fmt = is_train, i.FORMAT
        self.min_sizes = list(i.MIN_SIZE_TRAIN) if is_train else [i.MIN_SIZE_TEST]
        self.max_size = i.MAX_SIZE_TRAIN if is_train else i.MAX_SIZE_TEST
        self.flip = is_train and i.get("RANDOM_FLIP", "horizontal") == "horizontal"
        self.rng = rng or np.random.RandomState()

    def __call__(self, d: Dict) -> Dict:
        d = dict(d)
        img = read_image(d["file_name"], self.fmt)
        assert img.shape[:2] == (d["height"], d["width"]), f"{d['file_name']}: image size differs from its annotation"
        twin = read_image(d["data_dt_file_name"], self.fmt) if "data_dt_file_name" in d else None
        h, w = img.shape[:2]
        size = int(self.rng.choice(self.min_sizes))
        newh, neww = shortest_edge_size(h, w, size, self.max_size) if size else (h, w)
        do_flip = bool(self.flip and self.rng.uniform() < 0.5)

        def tf(a):
            a = resize_image(a, newh, neww)
            return a[:, ::-1] if do_flip else a

        d["image"] = torch.from_numpy(np.ascontiguousarray(tf(img).transpose(2, 0, 1)))
        if twin is not None:
            assert twin.shape[:2] == (h, w), f"{d['data_dt_file_name']}: twin size differs"
            d["image_trgt"] = torch.from_numpy(np.ascontiguousarray(tf(twin).transpose(2, 0, 1)))
        anns = d.pop("annotations", None)
        if not self.is_train or anns is None:
            return d
        boxes = np.asarray([a["bbox"] for a in anns], dtype=np.float64).reshape(-1, 4)
        boxes = boxes * np.array([neww / w, newh / h, neww / w, newh / h])            # ResizeTransform.apply_coords
        if do_flip:
            boxes = np.stack([neww - boxes[:, 2], boxes[:, 1], neww - boxes[:, 0], boxes[:, 3]], axis=1)   # HFlip + re-sort corners
        boxes = np.minimum(boxes, np.array([neww, newh, neww, newh], dtype=np.float64)).clip(min=0)       # transform_instance_annotations
        gt = Boxes(torch.from_numpy(boxes.astype(np.float32)))
        cls = torch.tensor([a["category_id"] for a in anns], dtype=torch.int64)
        keep = gt.nonempty(threshold=1e-5)                                                                  # filter_empty_instances
        d["instances"] = Instances((newh, neww), gt_boxes=gt[keep], gt_classes=cls[keep])
        return d


# ------------------------------------------------------------------------------------------------ sampling, batching, device
class TrainingSampler:
    """Infinite stream of indices: permutations drawn from one seeded generator shared by all ranks, sharded rank::world."""

    def __init__(self, size, shuffle=True, seed=0, rank=0, world=1):
        self.size, self.shuffle, self.seed, self.rank, self.world = size, shuffle, int(seed), rank, world

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        pos = 0
        while True:
            order = torch.randperm(self.size, generator=g).tolist() if self.shuffle else list(range(self.size))
            for i in order:
                if pos % self.world == self.rank:
                    yield i
                pos += 1


class _Mapped(torch.utils.data.IterableDataset):
    def __init__(self, dicts, mapper, sampler):
        self.dicts, self.mapper, self.sampler = dicts, mapper, sampler

    def __iter__(self):
        info = torch.utils.data.get_worker_info()
        wid, nw = (info.id, info.num_workers) if info is not None else (0, 1)
        if info is not None:                                      # decorrelate the augmentation streams of the workers
            self.mapper.rng = np.random.RandomState((self.sampler.seed * 1009 + self.sampler.rank * 101 + wid) % (2 ** 31))
        for n, idx in enumerate(self.sampler):
            if n % nw == wid:
                yield self.mapper(self.dicts[idx])


def _worker_init(_):
    torch.set_num_threads(1)          # a worker decodes / resizes with PIL: no intra-op pool per worker process


def aspect_ratio_batches(stream, batch_size):
    """data/common.py:152-186: two buckets (w > h, w <= h); a bucket is emitted when it holds batch_size samples"""
    buckets = ([], [])
    for d in stream:
        b = buckets[0 if d["image"].shape[2] > d["image"].shape[1] else 1]
        b.append(d)
        if len(b) == batch_size:
            yield b[:]
            del b[:]


class DeviceBatches:
    """Iterator of per-rank batches resident on ``device``.

    CUDA: a background thread takes the host batches as they come, packs every image of a batch (and its twin) into ONE pinned
    staging buffer from a small pool, sends it with one non-blocking copy on a side stream (boxes and classes likewise, one copy
    each) and hands the device views over together with an event; ``__next__`` makes the caller's stream wait for that event --
    the host never blocks on a copy, and batch i+1 is staged while the GPU runs step i.  (Round 3 measurement,
    tools/loader_bench.py: ``tensor.pin_memory()`` per image -- a pinned allocation each -- capped the loader at ~139 samples/s
    whatever the worker count, below the 152 samples/s the step consumes.)"""

    def __init__(self, batches, device, prefetch=2, max_batch_bytes=0):
        self.batches, self.device = iter(batches), torch.device(device)
        self._thread = None
        if self.device.type == "cuda":
            import atexit
            import queue
            import threading
            import weakref
            self._q = queue.Queue(maxsize=prefetch)
            self._stop = False
            self._stream = torch.cuda.Stream(self.device)
            self._free, self._busy = [], []                # pinned staging buffers: reusable / (event, buffer) still being read
            self._cap = int(max_batch_bytes)               # a pinned allocation costs tens of ms: buffers are sized for the largest batch
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
            ref = weakref.ref(self)
            atexit.register(lambda: ref() is not None and ref().close())   # the thread must be gone before the runtime is torn down

    def __iter__(self):
        return self

    def close(self):
        """drop the batch stream (shuts the DataLoader's worker processes down with it)"""
        if self._thread is not None:
            self._stop = True
            try:
                while True:
                    self._q.get_nowait()
            except Exception:
                pass
            self._thread.join(timeout=5.0)
            self._thread = None
        self.batches = iter(())
        import gc
        gc.collect()

    # ---------------------------------------------------------------- staging thread
    def _pinned(self, nbytes):
        still = []
        for ev, b in self._busy:                           # buffers whose copy has completed go back to the pool
            if ev.query():
                self._free.append(b)
            else:
                still.append((ev, b))
        self._busy = still
        for i, b in enumerate(self._free):
            if b.numel() >= nbytes:
                return self._free.pop(i)
        self._cap = max(self._cap, int(nbytes * 1.25) + 4096)
        self._free = []                                    # (smaller ones would never be picked again)
        return torch.empty(self._cap, dtype=torch.uint8, pin_memory=True)

    def _stage(self, batch):
        """host batch -> (device batch, event, device buffer): every image, twin, box and class tensor of the batch packed into one
        pinned buffer (256-byte slots) and sent with ONE copy"""
        items = []                                         # (sample index, field, tensor)
        for i, d in enumerate(batch):
            for k in ("image", "image_trgt"):
                if k in d:
                    items.append((i, k, d[k]))
            if "instances" in d:
                items.append((i, "gt_boxes", d["instances"].gt_boxes.tensor.contiguous()))
                items.append((i, "gt_classes", d["instances"].gt_classes.contiguous()))
        sizes = [(t.numel() * t.element_size() + 255) // 256 * 256 for _, _, t in items]
        buf = self._pinned(sum(sizes))
        off, slots = 0, []
        for (i, k, t), n in zip(items, sizes):
            nb = t.numel() * t.element_size()
            if nb:
                buf[off:off + nb].view(t.dtype).view(t.shape).copy_(t)
            slots.append((i, k, off, nb, t.dtype, tuple(t.shape)))
            off += n
        with torch.cuda.stream(self._stream):
            dev = buf[:max(off, 1)].to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        self._busy.append((ev, buf))
        out = [dict(d) for d in batch]
        fields = {}
        for i, k, o, nb, dt, shape in slots:
            v = dev[o:o + nb].view(dt).view(shape)
            if k in ("image", "image_trgt"):
                out[i][k] = v
            else:
                fields.setdefault(i, {})[k] = v
        for i, f in fields.items():
            out[i]["instances"] = Instances(batch[i]["instances"].image_size, gt_boxes=Boxes(f["gt_boxes"]), gt_classes=f["gt_classes"])
        return out, ev, [dev]

    def _run(self):
        import time
        st = self.stats = {"batches": 0, "pull_s": 0.0, "stage_s": 0.0, "handover_wait_s": 0.0}   # where the thread's time goes
        try:
            torch.cuda.set_device(self.device)
            t0 = time.perf_counter()
            for batch in self.batches:
                if self._stop:
                    return
                t1 = time.perf_counter()
                item = self._stage(batch)
                t2 = time.perf_counter()
                self._q.put(item)
                t3 = time.perf_counter()
                st["batches"] += 1; st["pull_s"] += t1 - t0; st["stage_s"] += t2 - t1; st["handover_wait_s"] += t3 - t2
                t0 = t3
                if self._stop:
                    return
            self._q.put(StopIteration())
        except BaseException as e:       # noqa: BLE001 -- handed to the consumer
            self._q.put(e)

    def __next__(self):
        if self._thread is None:
            if self.device.type == "cuda":
                raise StopIteration
            out = []
            for d in next(self.batches):
                d = dict(d)
                for k in ("image", "image_trgt"):
                    if k in d:
                        d[k] = d[k].to(self.device)
                if "instances" in d:
                    d["instances"] = d["instances"].to(self.device)
                out.append(d)
            return out
        item = self._q.get()
        if isinstance(item, BaseException):
            self._thread = None
            if isinstance(item, StopIteration):
                raise StopIteration
            raise item
        out, ev, owned = item
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        for t in owned:
            t.record_stream(cur)         # allocated on the staging stream, consumed on the caller's
        return out


def build_detection_train_loader(cfg, dicts, per_rank_batch, rank=0, world=1, device="cuda", num_workers=None):
    """data/build.py:262-308: mapped infinite stream -> aspect-ratio batches -> device"""
    mapper = DatasetMapper(cfg, True, np.random.RandomState(cfg.SEED + rank if cfg.SEED >= 0 else None))
    sampler = TrainingSampler(len(dicts), True, max(cfg.SEED, 0), rank, world)
    nw = cfg.DATALOADER.NUM_WORKERS if num_workers is None else num_workers
    ds = _Mapped(dicts, mapper, sampler)
    # workers are SPAWNED, not forked: a forked child of a process that holds a HIP context inherits it (and counts as a GPU process)
    stream = torch.utils.data.DataLoader(ds, batch_size=None, num_workers=nw, prefetch_factor=4, multiprocessing_context="spawn", worker_init_fn=_worker_init) if nw else ds
    # upper bound of a batch's bytes: image + twin, short edge <= max(MIN_SIZE_TRAIN), long edge <= MAX_SIZE_TRAIN
    cap = per_rank_batch * 2 * 3 * (max(cfg.INPUT.MIN_SIZE_TRAIN) or cfg.INPUT.MAX_SIZE_TRAIN) * cfg.INPUT.MAX_SIZE_TRAIN + (per_rank_batch << 16)
    return DeviceBatches(aspect_ratio_batches(stream, per_rank_batch), device, max_batch_bytes=cap)


def build_detection_test_loader(cfg, dicts, batch_size=1, rank=0, world=1, device="cuda"):
    """data/build.py:311-357: in order, sharded over ranks (InferenceSampler), annotations dropped by the mapper"""
    mapper = DatasetMapper(cfg, False)
    n = len(dicts)
    shard = (n + world - 1) // world
    mine = list(range(rank * shard, min((rank + 1) * shard, n)))
    batches = ([mapper(dicts[i]) for i in mine[s:s + batch_size]] for s in range(0, len(mine), batch_size))
    return DeviceBatches(batches, device)

#!/usr/bin/env python
"""Does the REAL input pipeline keep up with the step?  (data/build.py:262-308, data/dataset_mapper.py:126-217)

bench.py times the step on synthetic batches already resident in HBM; this tool measures the other side:

  1. writes a VOC-shaped tree of JPEGs (500x375 / 375x500 like VOC2007, 1-5 boxes each) plus the domain-translated twins on the
     box's local disk (PIL; no dataset ships with the repo),
  2. loader alone: samples/s of ``build_detection_train_loader`` (JPEG decode of image + twin, ONE ResizeShortestEdge draw from
     INPUT.MIN_SIZE_TRAIN = 480..800 + flip applied to both, aspect-ratio batches, pinned uint8 -> device) for several
     DATALOADER.NUM_WORKERS,
  3. the training step fed by that loader (multi-scale shapes through the kernel dispatch, the caching allocator and the gradient
     buckets), against the same trainer on synthetic 800x1333 batches: ms/step, samples/s, allocator state after iteration 10
     and at the end.

usage (GPU box): python tools/loader_bench.py --iters 50 > profiles/r03_loader.txt
"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def make_voc_tree(root, n, seed=0, year="VOC2007", twin="clipart"):
    """n JPEG pairs + annotations; smooth synthetic pictures (JPEG decode cost depends on content: noise would overstate it)"""
    from PIL import Image
    from cddmsl_amd.evaluation import VOC_CLASS_NAMES
    g = np.random.RandomState(seed)
    base = os.path.join(root, "VOC", year)
    tw = os.path.normpath(os.path.join(base, "..", twin, year))
    for d in (os.path.join(base, "Annotations"), os.path.join(base, "ImageSets", "Main"), os.path.join(base, "JPEGImages"), os.path.join(tw, "JPEGImages")):
        os.makedirs(d, exist_ok=True)
    ids = []
    for i in range(n):
        w, h = (500, 375) if g.rand() < 0.75 else (375, 500)
        fid = f"{i:06d}"
        ids.append(fid)
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        img = np.stack([127 + 100 * np.sin(xx / g.uniform(8, 60) + g.uniform(0, 6)) * np.cos(yy / g.uniform(8, 60)) for _ in range(3)], axis=2)
        img = np.clip(img + g.normal(0, 12, img.shape), 0, 255).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(base, "JPEGImages", fid + ".jpg"), quality=90)
        Image.fromarray(np.clip(img.astype(np.int16) // 32 * 32 + 16, 0, 255).astype(np.uint8)).save(os.path.join(tw, "JPEGImages", fid + ".jpg"), quality=90)
        objs = ""
        for _ in range(g.randint(1, 6)):
            x0, y0 = g.randint(1, w - 80), g.randint(1, h - 80)
            x1, y1 = min(w, x0 + g.randint(40, 300)), min(h, y0 + g.randint(40, 300))
            objs += ("<object><name>%s</name><difficult>0</difficult><bndbox><xmin>%d</xmin><ymin>%d</ymin><xmax>%d</xmax><ymax>%d</ymax></bndbox></object>"
                     % (VOC_CLASS_NAMES[g.randint(0, 20)], x0, y0, x1, y1))
        with open(os.path.join(base, "Annotations", fid + ".xml"), "w") as f:
            f.write(f"<annotation><size><width>{w}</width><height>{h}</height></size>{objs}</annotation>")
    for split in ("trainval", "test"):
        with open(os.path.join(base, "ImageSets", "Main", split + ".txt"), "w") as f:
            f.write("\n".join(ids) + "\n")
    return base


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=256)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--workers", default="0,4,8,12")
    ap.add_argument("--loader-batches", type=int, default=12)
    args = ap.parse_args()
    import bench
    from cddmsl_amd import data, engine, synthetic
    from cddmsl_amd.evaluation import VOC_CLASS_NAMES
    cfg = bench.make_cfg("bf16")
    cfg.MODEL.DEVICE = "cuda:0"
    engine.limit_host_threads()       # as build_trainer does: the box grants ~16 CPUs, torch's default intra-op pool has one thread per logical CPU
    root = tempfile.mkdtemp(prefix="cddmsl_voc_")
    t0 = time.perf_counter()
    base = make_voc_tree(root, args.images)
    print(f"# synthetic VOC tree: {args.images} JPEG pairs (500x375 / 375x500) under {root}, written in {time.perf_counter() - t0:.1f} s")
    dicts = data.load_voc_instances(base, "trainval", VOC_CLASS_NAMES, dt_data="clipart")
    print(f"# INPUT.MIN_SIZE_TRAIN {tuple(cfg.INPUT.MIN_SIZE_TRAIN)} max {cfg.INPUT.MAX_SIZE_TRAIN}, flip {cfg.INPUT.RANDOM_FLIP}; host: {os.cpu_count()} logical CPUs ({bench.cpu_model_name()})")
    print("# 1. loader alone (decode 2 JPEGs + resize both + flip + batch + pin + H2D), samples/s after 2 warm-up batches")
    rates = {}
    for nw in [int(x) for x in args.workers.split(",")]:
        ld = data.build_detection_train_loader(cfg, dicts, args.batch, 0, 1, cfg.MODEL.DEVICE, num_workers=nw)
        for _ in range(2):
            next(ld)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.loader_batches):
            b = next(ld)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rates[nw] = args.batch * args.loader_batches / dt
        shapes = sorted({tuple(x["image"].shape[1:]) for x in b})
        st = getattr(ld, "stats", None) or {"batches": 1, "pull_s": 0, "stage_s": 0, "handover_wait_s": 0}
        nb = max(st["batches"], 1)
        print(f"workers {nw:3d}   {rates[nw]:8.1f} samples/s   ({dt / args.loader_batches * 1e3:7.1f} ms per batch of {args.batch}; staging thread per batch: "
              f"{st['pull_s'] / nb * 1e3:.1f} ms pulling samples, {st['stage_s'] / nb * 1e3:.1f} ms pinned copy + H2D enqueue, {st['handover_wait_s'] / nb * 1e3:.1f} ms waiting for the consumer; "
              f"last batch shapes {shapes[:2]}...)")
        ld.close()
        del ld
    best = max(rates, key=rates.get)

    def run(loader, tag):
        tr = engine.build_trainer(cfg, args.batch, 800, 1333)
        tr.model.load_state_dict(synthetic.make_state_dict(0), strict=False)
        tr.clipcap_model.load_state_dict(synthetic.make_mapper_state_dict(1))
        tr.iter, tr.metrics_period = 20000, 0
        if loader is not None:
            tr.data_loader, tr._data_loader_iter = loader, iter(loader)
        times = []
        for it in range(11):                                   # iterations 0-10: synchronised one by one (start-up behaviour)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tr.run_step()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        mem10 = (torch.cuda.memory_allocated(), torch.cuda.memory_reserved())
        n = max(args.iters - 11, 1)
        t0 = time.perf_counter()
        for it in range(n):                                    # the rest as the trainer runs them: no host synchronisation,
            tr.run_step()                                      # the loader's next() overlaps the previous step on the GPU
        torch.cuda.synchronize()
        med = (time.perf_counter() - t0) / n
        end = (torch.cuda.memory_allocated(), torch.cuda.memory_reserved())
        print(f"{tag:34s} {med * 1e3:7.1f} ms/step = {args.batch / med:6.1f} samples/s over {n} un-synchronised iterations   (iterations 0-10, synchronised: "
              f"{', '.join('%.0f' % (t * 1e3) for t in times)} ms)   reserved after iteration 10: {mem10[1] / 2**30:.1f} GiB, at the end: {end[1] / 2**30:.1f} GiB "
              f"(allocated {mem10[0] / 2**30:.1f} -> {end[0] / 2**30:.1f} GiB)")
        if loader is not None:
            loader.close()
        del tr
        torch.cuda.empty_cache()
        return med

    print(f"# 2. the training step, {args.iters} iterations each")
    s = run(None, "synthetic 800x1333 batches in HBM")
    r = run(data.build_detection_train_loader(cfg, dicts, args.batch, 0, 1, cfg.MODEL.DEVICE, num_workers=best), f"real loader, {best} workers, multi-scale")
    print(f"# the step consumes {args.batch / s:.0f} samples/s on synthetic 800x1333 inputs; the loader delivers {rates[best]:.0f} samples/s with {best} workers "
          f"({rates[best] / (args.batch / s):.1f}x); real-loader step / synthetic step = {r / s:.2f} (multi-scale images are smaller on average: 480-800 short edge)")


if __name__ == "__main__":
    main()

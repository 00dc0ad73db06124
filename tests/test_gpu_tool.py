"""GPU: the drop-in entry point end to end on a generated mini VOC + twin set -- real paired loader -> two training
iterations (HIP step) -> checkpoint -> ``--eval-only`` (HIP inference + VOC AP) from that checkpoint."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_checkpoint_eval_on_mini_voc(tmp_path, capsys):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import train_caption_consistency as tool
    from test_data_pipeline import _make_voc
    base = _make_voc(str(tmp_path), n=6)
    out = str(tmp_path / "out")
    common = ["--config-file", os.path.join(ROOT, "configs", "VOC-Experiments", "faster_rcnn_CLIP_R_50_C4.yaml"), "--voc-root", base]
    opts = ["SOLVER.IMS_PER_BATCH", "2", "INPUT.MIN_SIZE_TRAIN", "(160,)", "INPUT.MAX_SIZE_TRAIN", "224", "INPUT.MIN_SIZE_TEST", "160",
            "INPUT.MAX_SIZE_TEST", "224", "SOLVER.CHECKPOINT_PERIOD", "2", "OUTPUT_DIR", out, "DATALOADER.NUM_WORKERS", "0",
            "MODEL.RPN.PRE_NMS_TOPK_TRAIN", "300", "MODEL.RPN.POST_NMS_TOPK_TRAIN", "100", "MODEL.RPN.PRE_NMS_TOPK_TEST", "300",
            "MODEL.RPN.POST_NMS_TOPK_TEST", "50", "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", "32", "TEST.DETECTIONS_PER_IMAGE", "20"]
    tool.main(tool.default_argument_parser().parse_args(common + ["--dt-data", "clipart", "--max-iter", "2"] + opts))
    ck = os.path.join(out, "model_final.pth")
    assert os.path.exists(ck) and open(os.path.join(out, "last_checkpoint")).read() == "model_final.pth"
    data = torch.load(ck, map_location="cpu", weights_only=True)
    assert data["iteration"] == 1 and "backbone.layer3.0.conv1.weight" in data["model"] and data["optimizer"]["steps_done"] == 2
    with pytest.raises(SystemExit) as e:
        tool.main(tool.default_argument_parser().parse_args(common + ["--eval-only"] + opts + ["MODEL.WEIGHTS", ck]))
    assert e.value.code == 0
    printed = capsys.readouterr().out
    assert "'AP50'" in printed and "shape-skipped" in printed


def test_device_batches_stage_through_one_pinned_copy(tmp_path):
    """cddmsl_amd/data.py::DeviceBatches on CUDA: the staging thread's device batches equal the host batches it was fed (images,
    twins, boxes, classes, sizes), in order, for ragged multi-scale shapes; a finite stream ends with StopIteration; an exception in
    the producer reaches the consumer."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from cddmsl_amd import data
    from cddmsl_amd.structures import Boxes, Instances
    g = torch.Generator().manual_seed(3)
    host = []
    for b in range(5):
        batch = []
        for i in range(3):
            h, w = int(torch.randint(40, 90, (1,), generator=g)), int(torch.randint(40, 90, (1,), generator=g))
            n = int(torch.randint(0, 4, (1,), generator=g))
            batch.append({"image": torch.randint(0, 256, (3, h, w), dtype=torch.uint8, generator=g),
                          "image_trgt": torch.randint(0, 256, (3, h, w), dtype=torch.uint8, generator=g),
                          "instances": Instances((h, w), gt_boxes=Boxes(torch.rand(n, 4, generator=g) * 40), gt_classes=torch.randint(0, 20, (n,), generator=g)),
                          "image_id": f"{b}_{i}"})
        host.append(batch)
    got = list(data.DeviceBatches(iter(host), "cuda:0"))
    assert len(got) == len(host)
    for hb, db in zip(host, got):
        for hd, dd in zip(hb, db):
            assert dd["image"].is_cuda and torch.equal(dd["image"].cpu(), hd["image"]) and torch.equal(dd["image_trgt"].cpu(), hd["image_trgt"])
            assert dd["instances"].image_size == hd["instances"].image_size and dd["image_id"] == hd["image_id"]
            assert torch.equal(dd["instances"].gt_boxes.tensor.cpu(), hd["instances"].gt_boxes.tensor)
            assert torch.equal(dd["instances"].gt_classes.cpu(), hd["instances"].gt_classes)

    def bad():
        yield host[0]
        raise ValueError("decoder failed")

    it = data.DeviceBatches(bad(), "cuda:0")
    next(it)
    with pytest.raises(ValueError):
        next(it)


@pytest.mark.parametrize("workers,dtype", [(0, "bf16"), (2, "bf16"), (0, "fp8")])
def test_multi_scale_training_on_the_real_loader_keeps_the_allocator_flat(tmp_path, monkeypatch, workers, dtype):
    """SURVEY.md 8(f)2 / data/build.py:262-308: the paired VOC loader with INPUT.MIN_SIZE_TRAIN multi-scale sampling feeds the
    full step (all three branches) for 24 iterations -- every batch another shape through the kernel dispatch, the workspace
    registry and the gradient buckets -- without an error, with finite losses, and without the caching allocator growing after
    iteration 10 by more than the largest batch's share (shapes seen late may still be larger than any before).  The fp8 case forces
    every e4m3 kernel on at this size (forward, input- and weight-gradient GEMMs, delayed-scaling slots, the copies kept for backward):
    their shapes change every iteration too, and ``_write_metrics`` reads the quantisers' non-finite flag each step."""
    if dtype == "fp8":
        monkeypatch.setenv("CDDMSL_FP8_MIN_TILES", "1")
        monkeypatch.setenv("CDDMSL_FP8_WGRAD_MIN_M", "1")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench
    import loader_bench
    from cddmsl_amd import data, engine, synthetic
    from cddmsl_amd.evaluation import VOC_CLASS_NAMES
    base = loader_bench.make_voc_tree(str(tmp_path), 24)
    cfg = bench.make_cfg(dtype)
    cfg.merge_from_list(["MODEL.DEVICE", "cuda:0", "INPUT.MIN_SIZE_TRAIN", (224, 256, 288, 320), "INPUT.MAX_SIZE_TRAIN", 448,
                         "MODEL.RPN.PRE_NMS_TOPK_TRAIN", 2000, "MODEL.RPN.POST_NMS_TOPK_TRAIN", 500, "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 128])
    dicts = data.load_voc_instances(base, "trainval", VOC_CLASS_NAMES, dt_data="clipart")
    tr = engine.build_trainer(cfg, 4, 320, 448)
    tr.model.load_state_dict(synthetic.make_state_dict(0), strict=False)
    tr.clipcap_model.load_state_dict(synthetic.make_mapper_state_dict(1))
    tr.iter, tr.metrics_period = 20000, 1
    tr.data_loader = data.build_detection_train_loader(cfg, dicts, 4, 0, 1, "cuda:0", num_workers=workers)   # (2: spawned worker processes)
    tr._data_loader_iter = iter(tr.data_loader)
    shapes, reserved = set(), []
    for it in range(24):
        tr.run_step()
        assert all(v == v and abs(v) < 1e4 for v in tr.storage.values()), (it, tr.storage)
        reserved.append(torch.cuda.memory_reserved())
    tr.data_loader.close()
    assert reserved[-1] <= reserved[10] * 1.25 + (1 << 28), [r >> 20 for r in reserved]
    print("reserved MiB per iteration", [r >> 20 for r in reserved])

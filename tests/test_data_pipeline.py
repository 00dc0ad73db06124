"""Host-side rows next to the hot path (SURVEY.md 8(f)2,4): paired VOC data pipeline and checkpoint compatibility.
CPU only.  Geometry and resize numerics are checked against hand-computed values and the PIL call both sides make, and -- round 2 --
against a fixture produced by the reference's own transform / annotation code with fvcore's base classes stubbed
(tests/golden/make_golden_data.py -> ref_data_transforms.npz: test_mapper_geometry_matches_reference_transforms)."""
import os

import numpy as np
import pytest
import torch

from cddmsl_amd import data
from cddmsl_amd.config import get_cfg

CLASSES = data.VOC_CLASS_NAMES


def _make_voc(root, n=7, year="VOC2007", twin="clipart"):
    from PIL import Image
    g = np.random.RandomState(0)
    base = os.path.join(root, "VOC", year)
    tw = os.path.join(root, "VOC", "..", twin, year)      # <dirname>/../<dt_data>/<VOC2007>
    tw = os.path.normpath(os.path.join(base, "..", twin, year))
    for d in (os.path.join(base, "Annotations"), os.path.join(base, "ImageSets", "Main"), os.path.join(base, "JPEGImages"), os.path.join(tw, "JPEGImages")):
        os.makedirs(d, exist_ok=True)
    ids = []
    for i in range(n):
        w, h = (120, 90) if i % 3 else (80, 130)
        fid = f"{i:06d}"
        ids.append(fid)
        img = g.randint(0, 256, (h, w, 3), dtype=np.uint8)
        Image.fromarray(img).save(os.path.join(base, "JPEGImages", fid + ".jpg"), quality=95)
        Image.fromarray(255 - img).save(os.path.join(tw, "JPEGImages", fid + ".jpg"), quality=95)
        objs = "".join("<object><name>%s</name><difficult>%d</difficult><bndbox><xmin>%d</xmin><ymin>%d</ymin><xmax>%d</xmax><ymax>%d</ymax></bndbox></object>"
                       % (CLASSES[(i + k) % 20], k % 2, 5 + 7 * k, 3 + 5 * k, 40 + 9 * k, 50 + 4 * k) for k in range(1 + i % 3))
        open(os.path.join(base, "Annotations", fid + ".xml"), "w").write(
            f"<annotation><size><width>{w}</width><height>{h}</height></size>{objs}</annotation>")
    for split in ("trainval", "test"):
        open(os.path.join(base, "ImageSets", "Main", split + ".txt"), "w").write("\n".join(ids) + "\n")
    return base


def test_voc_dicts_and_mapper(tmp_path):
    base = _make_voc(str(tmp_path))
    dicts = data.load_voc_instances(base, "trainval", dt_data="clipart")
    assert len(dicts) == 7 and dicts[1]["annotations"][0]["bbox"] == [4.0, 2.0, 40.0, 50.0]          # xmin-1, ymin-1
    assert dicts[0]["data_dt_file_name"].endswith(os.path.join("clipart", "VOC2007", "JPEGImages", "000000.jpg"))
    assert "data_dt_file_name" not in data.load_voc_instances(base, "test", dt_data="clipart")[0]       # twins only for training splits
    cfg = get_cfg()
    cfg.merge_from_list(["INPUT.MIN_SIZE_TRAIN", (64, 96), "INPUT.MAX_SIZE_TRAIN", 128, "INPUT.FORMAT", "RGB"])

    class FixedRng:                                  # scripted draws: short edge 96, flip
        def choice(self, a):
            return a[1]

        def uniform(self):
            return 0.1

    m = data.DatasetMapper(cfg, True, FixedRng())
    out = m(dicts[1])                                # 120 x 90 (w x h): short edge 96 -> 128 x 96, capped at max 128 exactly
    assert tuple(out["image"].shape) == (3, 96, 128) and out["image"].dtype == torch.uint8
    assert tuple(out["image_trgt"].shape) == (3, 96, 128)
    from PIL import Image
    src = data.read_image(dicts[1]["file_name"], "RGB")
    want = np.asarray(Image.fromarray(src).resize((128, 96), Image.BILINEAR))[:, ::-1]
    assert np.array_equal(out["image"].numpy().transpose(1, 2, 0), want)
    twin = data.read_image(dicts[1]["data_dt_file_name"], "RGB")
    assert np.array_equal(out["image_trgt"].numpy().transpose(1, 2, 0), np.asarray(Image.fromarray(twin).resize((128, 96), Image.BILINEAR))[:, ::-1])
    sx, sy = 128 / 120, 96 / 90
    b = out["instances"].gt_boxes.tensor[0].tolist()           # [4, 2, 40, 50] scaled, then flipped in x
    assert np.allclose(b, [128 - 40 * sx, 2 * sy, 128 - 4 * sx, 50 * sy], atol=1e-4)
    assert out["instances"].image_size == (96, 128) and out["instances"].gt_classes.dtype == torch.int64
    # test-time mapper: MIN_SIZE_TEST, no flip, annotations dropped
    mt = data.DatasetMapper(cfg, False)
    o = mt(dicts[0])                                 # 80 x 130 (w x h): short edge 80 -> 800, long edge 1300 <= MAX_SIZE_TEST
    assert "instances" not in o and tuple(o["image"].shape) == (3, 1300, 800) and o["height"] == 130


def test_sampler_grouping_and_loader(tmp_path):
    base = _make_voc(str(tmp_path), n=9)
    dicts = data.load_voc_instances(base, "trainval", dt_data="clipart")
    # two ranks see disjoint, interleaved slices of ONE permutation stream
    a = data.TrainingSampler(9, True, 5, 0, 2)
    b = data.TrainingSampler(9, True, 5, 1, 2)
    one = data.TrainingSampler(9, True, 5, 0, 1)
    ia, ib, io = iter(a), iter(b), iter(one)
    merged = [next(io) for _ in range(18)]
    assert [next(ia) for _ in range(9)] == merged[0::2] and [next(ib) for _ in range(9)] == merged[1::2]
    assert sorted(merged[:9]) == list(range(9))
    cfg = get_cfg()
    cfg.merge_from_list(["INPUT.MIN_SIZE_TRAIN", (64,), "INPUT.MAX_SIZE_TRAIN", 128, "INPUT.FORMAT", "RGB", "SEED", 3])
    loader = data.build_detection_train_loader(cfg, dicts, per_rank_batch=2, device="cpu", num_workers=0)
    for _ in range(4):
        batch = next(loader)
        assert len(batch) == 2
        land = [x["image"].shape[2] > x["image"].shape[1] for x in batch]
        assert land[0] == land[1]                    # aspect-ratio grouping
        for x in batch:
            assert x["image"].shape == x["image_trgt"].shape and len(x["instances"]) >= 1
    # worker processes deliver the same kind of batches
    loader = data.build_detection_train_loader(cfg, dicts, per_rank_batch=2, device="cpu", num_workers=2)
    assert len(next(loader)) == 2
    test = list(data.build_detection_test_loader(cfg, data.load_voc_instances(base, "test"), batch_size=2, rank=1, world=2, device="cpu"))
    assert [x["image_id"] for bt in test for x in bt] == ["000005", "000006", "000007", "000008"]   # InferenceSampler shard of rank 1


def test_checkpoint_roundtrip_and_side_loads(tmp_path):
    from cddmsl_amd import synthetic
    from cddmsl_amd.checkpoint import DetectionCheckpointer, offline_backbone_state
    from cddmsl_amd.modeling import build_model
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "VOC-Experiments", "faster_rcnn_CLIP_R_50_C4.yaml"))
    cfg.MODEL.DEVICE = "cpu"
    model = build_model(cfg)
    sd = synthetic.make_state_dict(0)
    # a reference-format file: {"model": {...}} with the reference's parameter names, one wrong-shaped tensor, one stranger
    bad = dict(sd)
    bad["roi_heads.box_predictor.bbox_pred.bias"] = torch.zeros(7)
    bad["lang_encoder.token_embedding.weight"] = torch.zeros(3, 3)
    path = str(tmp_path / "regionclip_like.pth")
    torch.save({"model": bad, "iteration": 41}, path)
    ck = DetectionCheckpointer(model, str(tmp_path / "out"))
    inc = ck.load(path)
    assert inc.incorrect_shapes == [("roi_heads.box_predictor.bbox_pred.bias", (7,), (80,))]
    assert inc.unexpected_keys == ["lang_encoder.token_embedding.weight"]
    assert all(k.startswith(("offline_backbone.", "projector.")) or "bbox_pred.bias" in k or "cell_anchors" in k for k in inc.missing_keys), inc.missing_keys
    got = model.state_dict()
    for k in ("backbone.layer3.2.conv2.weight", "backbone.attnpool.k_proj.weight", "proposal_generator.rpn_head.conv.weight"):
        assert torch.equal(got[k], sd[k])
    # PRE_TRAINED_RCLIP_PATH -> offline_backbone (train_loop.py:150-161)
    assert set(offline_backbone_state(sd)) == {k[9:] for k in sd if k.startswith("backbone.")}
    ck.load_offline_backbone(path)
    assert torch.equal(model.offline_backbone.state_dict()["layer2.1.conv3.weight"], sd["backbone.layer2.1.conv3.weight"])
    # save -> resume
    p2 = ck.save("model_0000041", iteration=41)
    model2 = build_model(cfg)

    class T:
        iter = 0
    t = T()
    inc2 = DetectionCheckpointer(model2, str(tmp_path / "out"), trainer=t).resume_or_load("", resume=True)
    assert t.iter == 42 and not inc2.unexpected_keys and not inc2.incorrect_shapes and os.path.basename(p2) == "model_0000041.pth"
    for k, v in model.state_dict().items():
        assert torch.equal(v, model2.state_dict()[k]), k


def test_clip_checkpoint_name_conversion(tmp_path):
    """checkpoint/clip_model_loading.py:10-343 on a synthetic state dict: an OpenAI-CLIP-style file (``visual.*`` tower + text
    tower) converts to the model's ``backbone.*`` names (never ``offline_backbone.*``), the offline-module checkpoint
    (bb_rpn_weights) to ``offline_backbone.*``; Caffe2-layout box-head rows are re-ordered; round trip through a file."""
    from cddmsl_amd import synthetic
    from cddmsl_amd.checkpoint import DetectionCheckpointer, convert_clip_state
    from cddmsl_amd.modeling import build_model
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "VOC-Experiments", "faster_rcnn_CLIP_R_50_C4.yaml"))
    cfg.MODEL.DEVICE = "cpu"
    model = build_model(cfg)
    sd = synthetic.make_state_dict(0)
    bb = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}
    oai = {"visual." + k: v for k, v in bb.items()}
    oai.update({"transformer.resblocks.0.attn.in_proj_weight": torch.zeros(4, 4), "token_embedding.weight": torch.zeros(5, 4),
                "logit_scale": torch.tensor(4.6)})
    own = model.state_dict()
    conv, pairs = convert_clip_state(own, oai)
    assert all(("backbone." + k) in conv and torch.equal(conv["backbone." + k], v) for k, v in bb.items())
    assert not any(k.startswith("offline_backbone.") for k in conv)
    assert pairs["backbone.layer3.0.downsample.0.weight"] == "visual.layer3.0.downsample.0.weight"
    assert {"transformer.resblocks.0.attn.in_proj_weight", "token_embedding.weight", "logit_scale"} <= set(conv)   # passed through
    # through a file named like the published one
    path = str(tmp_path / "OAI_CLIP_RN50.pth")
    torch.save(oai, path)
    model.backbone.layer2[0].conv1.weight.data.zero_()
    inc = DetectionCheckpointer(model, str(tmp_path)).load(path)
    assert sorted(inc.unexpected_keys) == ["logit_scale", "token_embedding.weight", "transformer.resblocks.0.attn.in_proj_weight"]
    assert not [k for k in inc.missing_keys if k.startswith("backbone.")] and not inc.incorrect_shapes
    assert torch.equal(model.state_dict()["backbone.layer2.0.conv1.weight"], sd["backbone.layer2.0.conv1.weight"])
    # second checkpoint -> offline modules only
    second = {"backbone.layer1.0.conv1.weight": sd["backbone.layer1.0.conv1.weight"] + 1.0,
              "proposal_generator.rpn_head.conv.weight": sd["proposal_generator.rpn_head.conv.weight"], "roi_heads.x": torch.zeros(1)}
    conv2, _ = convert_clip_state(own, second, bb_rpn_weights=True)
    assert set(conv2) == {"offline_backbone.layer1.0.conv1.weight", "offline_proposal_generator.rpn_head.conv.weight"}
    # Caffe2-layout heads: bbox_pred drops the 4 background rows, cls_score moves the background row last; shape mismatch skips
    heads = {"bbox.pred.w": torch.arange(84.0 * 2).view(84, 2), "cls.score.w": torch.arange(21.0).view(21, 1), "conv.rpn.w": torch.zeros(3)}
    tgt = {"roi_heads.box_predictor.bbox_pred.w": torch.zeros(80, 2), "roi_heads.box_predictor.cls_score.w": torch.zeros(21, 1),
           "proposal_generator.rpn_head.conv.w": torch.zeros(4)}
    conv3, pairs3 = convert_clip_state(tgt, heads)
    assert torch.equal(conv3["roi_heads.box_predictor.bbox_pred.w"], heads["bbox.pred.w"][4:])
    assert conv3["roi_heads.box_predictor.cls_score.w"].flatten().tolist() == list(range(1, 21)) + [0]
    assert "proposal_generator.rpn_head.conv.w" in conv3 and "proposal_generator.rpn_head.conv.w" not in pairs3   # shape mismatch: unmatched
    # one checkpoint tensor claimed by two model keys is an error
    with pytest.raises(ValueError):
        convert_clip_state({"a.conv1.weight": torch.zeros(1), "b.conv1.weight": torch.zeros(1)}, {"conv1.weight": torch.zeros(1)})


def test_mapper_geometry_matches_reference_transforms(tmp_path):
    """cddmsl_amd/data.py against numbers produced by the reference's own ResizeShortestEdge / ResizeTransform / RandomFlip /
    transform_instance_annotations / annotations_to_instances / filter_empty_instances (tests/golden/ref_data_transforms.npz,
    generator tests/golden/make_golden_data.py): new sizes, PIL-resized pixels of image and twin (bit-equal), flipped result,
    transformed + clipped + filtered boxes, and the ORDER of the random draws of a sample (short edge, then flip)."""
    from PIL import Image
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_data_transforms.npz"))
    for h, w, size, max_size, nh, nw in g["sizes"].tolist():
        assert data.shortest_edge_size(h, w, size, max_size) == (nh, nw)
    n_img = 0
    for i in range(len(g["sizes"])):
        h, w, size, max_size, nh, nw = g["sizes"][i].tolist()
        flip = int(g[f"flip{i}"])

        class Scripted:                                   # the mapper's two draws, scripted: this short edge, this flip decision
            def choice(self, a):
                return size

            def uniform(self):
                return 0.25 if flip else 0.75

        cfg = get_cfg()
        cfg.merge_from_list(["INPUT.MIN_SIZE_TRAIN", (size,), "INPUT.MAX_SIZE_TRAIN", max_size, "INPUT.FORMAT", "RGB"])
        m = data.DatasetMapper(cfg, True, Scripted())
        if f"img{i}" in g.files:
            img = g[f"img{i}"]
            assert np.array_equal(data.resize_image(img, nh, nw), g[f"resized{i}"])
            p, pt = str(tmp_path / f"a{i}.png"), str(tmp_path / f"t{i}.png")
            Image.fromarray(img).save(p)
            Image.fromarray(255 - img).save(pt)
            n_img += 1
        else:                                             # geometry only: a blank image of the right size
            p = pt = str(tmp_path / f"b{i}.png")
            Image.fromarray(np.zeros((h, w, 3), np.uint8)).save(p)
        d = {"file_name": p, "data_dt_file_name": pt, "height": h, "width": w, "image_id": str(i),
             "annotations": [{"bbox": b.tolist(), "category_id": int(c)} for b, c in zip(g[f"boxes_in{i}"], g[f"classes_in{i}"])]}
        out = m(d)
        assert tuple(out["image"].shape) == (3, nh, nw)
        if f"img{i}" in g.files:
            assert np.array_equal(out["image"].permute(1, 2, 0).numpy(), g[f"final{i}"])
            want_twin = g[f"twin_resized{i}"][:, ::-1] if flip else g[f"twin_resized{i}"]
            assert np.array_equal(out["image_trgt"].permute(1, 2, 0).numpy(), want_twin)
        inst = out["instances"]
        assert np.allclose(inst.gt_boxes.tensor.numpy(), g[f"boxes_out{i}"], rtol=0, atol=1e-4), i
        assert np.array_equal(inst.gt_classes.numpy(), g[f"classes_out{i}"]) and len(inst) < len(g[f"boxes_in{i}"])   # the outside box is gone
    assert n_img >= 3
    # draw order of a sample with the shipped augmentation settings
    cfg = get_cfg()
    cfg.merge_from_list(["INPUT.MIN_SIZE_TRAIN", tuple(range(480, 801, 32)), "INPUT.MAX_SIZE_TRAIN", 1333, "INPUT.FORMAT", "RGB"])
    m = data.DatasetMapper(cfg, True, np.random.RandomState(321))
    p = str(tmp_path / "z.png")
    Image.fromarray(np.zeros((375, 500, 3), np.uint8)).save(p)
    seen = []
    for _ in range(16):
        out = m({"file_name": p, "height": 375, "width": 500, "image_id": "z", "annotations": [{"bbox": [10.0, 10.0, 60.0, 50.0], "category_id": 1}]})
        nh, nw = out["image"].shape[1:]
        x0 = float(out["instances"].gt_boxes.tensor[0, 0])
        seen.append((nh, nw, int(abs(x0 - (nw - 60.0 * nw / 500)) < 1e-3)))          # flipped: x0' = new_w - x1 * scale
    assert seen == [tuple(r) for r in g["draw_sequence"].tolist()]


def test_voc_dataset_dicts_match_reference_loader(tmp_path, monkeypatch):
    """data.load_voc_instances against the dataset dicts the reference's own ``load_voc_DG_instances`` built from the same
    annotation files (tests/golden/ref_voc_dicts.json, generator make_golden_data.py voc): file and twin paths, ids, sizes,
    category ids, the xmin / ymin - 1 convention -- VOC2007 and VOC2012, training splits with and without a twin directory, test split."""
    import json
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_voc_dicts.json")))
    for year in ("VOC2007", "VOC2012"):
        base = tmp_path / "VOCdevkit" / year
        (base / "Annotations").mkdir(parents=True)
        (base / "ImageSets" / "Main").mkdir(parents=True)
        for fid, (w, h, objs) in fx["mini_voc"].items():
            body = "".join("<object><name>%s</name><difficult>%d</difficult><bndbox><xmin>%d</xmin><ymin>%d</ymin><xmax>%d</xmax><ymax>%d</ymax></bndbox></object>"
                           % tuple(o) for o in objs)
            (base / "Annotations" / (fid + ".xml")).write_text("<annotation><size><width>%d</width><height>%d</height></size>%s</annotation>" % (w, h, body))
        for split in ("trainval", "test"):
            (base / "ImageSets" / "Main" / (split + ".txt")).write_text("\n".join(fx["mini_voc"]) + "\n")
    monkeypatch.chdir(tmp_path)
    for key, want in fx["dicts"].items():
        year, split, dt = key.split("|")
        got = data.load_voc_instances(os.path.join("VOCdevkit", year), split, CLASSES, dt_data=None if dt == "None" else dt)
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert set(a) == set(b), (key, set(a) ^ set(b))
            for k in ("file_name", "image_id", "height", "width"):
                assert a[k] == b[k], (key, k)
            assert a.get("data_dt_file_name") == b.get("data_dt_file_name")
            assert [(x["category_id"], x["bbox"]) for x in a["annotations"]] == [(x["category_id"], x["bbox"]) for x in b["annotations"]]


def test_sampler_streams_and_grouping_match_reference():
    """``ref_samplers.json`` (tests/golden/make_golden_data.py samplers): index streams of the reference's ``TrainingSampler``
    (data/samplers/distributed_sampler.py:12-54) for worlds 1-3, shuffled and not, and the batches its
    ``AspectRatioGroupedDataset`` (data/common.py:152-186) forms from 37 images of mixed orientation."""
    import itertools
    import json
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_samplers.json")))
    for key, want in ref["sampler"].items():
        world, rank, size, seed, shuffle = (int(v) for v in key.split("|"))
        got = list(itertools.islice(iter(data.TrainingSampler(size, bool(shuffle), seed, rank, world)), len(want)))
        assert got == want, key
    items = [{"id": d["id"], "image": torch.empty(3, d["height"], d["width"], dtype=torch.uint8)} for d in ref["items"]]
    for b, want in ref["groups"].items():
        got = [[d["id"] for d in batch] for batch in data.aspect_ratio_batches(iter(items), int(b))]
        assert got == want, b


def test_clip_checkpoint_alignment_matches_reference():
    """``convert_clip_state`` against what the reference's OWN ``align_and_update_state_dicts_for_CLIP`` returned
    (checkpoint/clip_model_loading.py:190-343; fixture tests/golden/ref_ckpt_align.json, generator make_golden_ckpt.py): the same
    result names, the same tensor behind each name (every checkpoint element carries a unique value), the same errors.  Cases:
    an OpenAI-CLIP file into the model with / without an offline backbone, the second (``bb_rpn_weights``) checkpoint,
    Caffe2-layout heads + RPN blob names + a shape mismatch, an ambiguous suffix, no match at all, longest-suffix priority."""
    import json
    from cddmsl_amd.checkpoint import convert_clip_state
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_ckpt_align.json")))
    assert len(fx["cases"]) >= 7
    for c in fx["cases"]:
        model = {k: torch.zeros(*shp) if shp else torch.zeros(()) for k, shp in c["model"].items()}
        ckpt = {}
        for k, (shp, start) in c["ckpt"].items():
            n = int(np.prod(shp)) if shp else 1
            ckpt[k] = (torch.arange(n, dtype=torch.float32) + float(start)).view(*shp) if shp else torch.tensor(float(start))
        if c["raises"]:
            with pytest.raises(ValueError):
                convert_clip_state(model, ckpt, c["bb_rpn_weights"])
            assert c["raises"] == "ValueError"
            continue
        got, pairs = convert_clip_state(model, ckpt, c["bb_rpn_weights"])
        want = dict(c["expect"])
        want.pop("ignore_others", None)        # (the reference's marker tensor for its FrozenBN loader, clip_model_loading.py:230: not a weight)
        assert set(got) == set(want), (c["name"], sorted(set(got) ^ set(want))[:8])
        for k, (shp, vals) in want.items():
            assert list(got[k].shape) == shp and got[k].flatten().tolist() == vals, (c["name"], k)
        assert all(k in got for k in pairs)

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

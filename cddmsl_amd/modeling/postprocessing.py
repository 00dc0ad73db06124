"""``detector_postprocess`` -- detectron2/modeling/postprocessing.py:9-75 (boxes only: masks / keypoints are off the path)."""
from ..structures import Boxes, Instances


def detector_postprocess(results: Instances, output_height: int, output_width: int):
    """Rescale the detector's boxes from the resolution it saw (``results.image_size``) to the requested output resolution,
    clip them to it and drop the boxes that end up empty."""
    scale_x, scale_y = output_width / results.image_size[1], output_height / results.image_size[0]
    fields = dict(results.get_fields())
    key = "pred_boxes" if "pred_boxes" in fields else "proposal_boxes"
    assert key in fields, "Predictions must contain boxes!"
    boxes = Boxes(fields[key].tensor.clone())
    boxes.scale(scale_x, scale_y)
    boxes.clip((output_height, output_width))
    fields[key] = boxes
    out = Instances((output_height, output_width), **fields)
    return out[boxes.nonempty()]

"""Minimal ``Boxes`` / ``Instances`` / ``ImageList`` / ``ShapeSpec`` with the reference's field names
(detectron2/structures/{boxes,instances,image_list}.py, layers/shape_spec.py) -- the batch-dict contract of
SURVEY.md 8(b): ``{"image", "image_trgt", "instances": Instances(gt_boxes=Boxes, gt_classes)}``."""
from collections import namedtuple
from typing import Any, Dict, List, Tuple

import torch

ShapeSpec = namedtuple("ShapeSpec", ["channels", "height", "width", "stride"], defaults=(None, None, None, None))


class Boxes:
    def __init__(self, tensor: torch.Tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        tensor = tensor.to(torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4))
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def to(self, device):
        return Boxes(self.tensor.to(device))

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        return Boxes(self.tensor[item])

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def clip(self, box_size):
        """boxes.py:192-206: clamp x to [0, width], y to [0, height] (box_size = (height, width)); in place."""
        h, w = box_size
        t = self.tensor
        self.tensor = torch.stack((t[:, 0].clamp(min=0, max=w), t[:, 1].clamp(min=0, max=h),
                                   t[:, 2].clamp(min=0, max=w), t[:, 3].clamp(min=0, max=h)), dim=-1)

    def nonempty(self, threshold: float = 0.0):
        """boxes.py:208-222"""
        t = self.tensor
        return ((t[:, 2] - t[:, 0]) > threshold) & ((t[:, 3] - t[:, 1]) > threshold)

    def scale(self, scale_x, scale_y):
        """boxes.py:280-285 (in place)"""
        self.tensor[:, 0::2] *= scale_x
        self.tensor[:, 1::2] *= scale_y

    @property
    def device(self):
        return self.tensor.device

    @staticmethod
    def cat(boxes_list):
        return Boxes(torch.cat([b.tensor for b in boxes_list], dim=0)) if boxes_list else Boxes(torch.empty(0, 4))


class Instances:
    def __init__(self, image_size: Tuple[int, int], **kwargs: Any):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_fields", {})
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            object.__setattr__(self, name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
        return self._fields[name]

    def set(self, name, value):
        n = len(value)
        if len(self._fields):
            assert len(self) == n, f"Adding a field of length {n} to a Instances of length {len(self)}"
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def get(self, name):
        return self._fields[name]

    def get_fields(self) -> Dict[str, Any]:
        return self._fields

    def to(self, device):
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v.to(device) if hasattr(v, "to") else v)
        return ret

    def __getitem__(self, item):
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        raise NotImplementedError("Empty Instances does not support __len__!")


class ImageList:
    def __init__(self, tensor: torch.Tensor, image_sizes: List[Tuple[int, int]]):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self):
        return len(self.image_sizes)


def as_instances(item):
    """Accept either an ``Instances`` or the plain dict form {'gt_boxes': Tensor, 'gt_classes': Tensor}."""
    if isinstance(item, Instances):
        return item
    size = item.get("image_size", (0, 0))
    return Instances(size, gt_boxes=Boxes(item["gt_boxes"]), gt_classes=item["gt_classes"])

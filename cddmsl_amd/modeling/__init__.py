from .backbone import build_backbone, build_clip_resnet_backbone, ModifiedResNet, AttentionPool2d, Bottleneck, FrozenBatchNorm2d  # noqa
from .clipcap import TransformerMapper, v2l  # noqa
from .rpn import RPN, StandardRPNHead, DefaultAnchorGenerator, build_proposal_generator  # noqa
from .resnet import build_resnet_backbone, ResNet, BasicStem, BottleneckBlock  # noqa
from .roi_heads import Res5ROIHeads, CLIPRes5ROIHeads, FastRCNNOutputLayers, ROIPooler, build_roi_heads  # noqa
from .rcnn import GeneralizedRCNN, GatherLayer, build_model  # noqa

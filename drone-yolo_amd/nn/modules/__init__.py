"""Modules of the Drone-YOLO path; same names as ultralytics.nn.modules exports for them."""
from .block import DFL, SPPF, Bottleneck, C2f, RepVGGBlock, SEBlock, conv_bn
from .conv import Concat, Conv, DWConv, PlainConv2d, Upsample, autopad
from .head import Detect

__all__ = ("Conv", "DWConv", "Concat", "Upsample", "PlainConv2d", "autopad", "DFL", "SPPF", "C2f", "Bottleneck",
           "RepVGGBlock", "SEBlock", "conv_bn", "Detect")

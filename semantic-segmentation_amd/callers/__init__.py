"""Callers of the DDCN operator: the DDCN-ResNet backbone, the PointNet depth branch, the
DANet head and segmentor, and the data-parallel training step -- enough of `detseg/models` and `detseg/apis` to run
BASELINE configs[1]-[3] on synthetic data.  Plain PyTorch modules (MIOpen convolutions) around the HIP operator;
parameter names follow the reference so that its checkpoints' keys line up.

PARITY UNPINNED: `detseg.models` cannot be imported (segmentors/single_stage.py:166 is a SyntaxError, mmcv is
absent) and the reference has no test for it; tests/test_callers.py checks structure, shapes and the invariants
the reference's code implies instead.
"""
from .resnet import Bottleneck, ResNet, make_res_layer
from .pointnet import PointNet
from .danet import CAM_Module, DANet, DANetHead, PAM_Module
from .train import OverlappedGradReducer, build_optimizer, poly_lr, train_step

__all__ = ["Bottleneck", "ResNet", "make_res_layer", "PointNet", "CAM_Module", "DANet", "DANetHead", "PAM_Module",
           "OverlappedGradReducer", "build_optimizer", "poly_lr", "train_step"]

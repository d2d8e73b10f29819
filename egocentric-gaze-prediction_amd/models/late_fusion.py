"""Mirror of the reference's ``models/late_fusion.py``: the LF head
``cat(f, g) -> 3 x [Conv3x3, BN, ReLU] (32, 32, 8) -> Conv1x1 (8 -> 1) -> Sigmoid`` at full 224x224
(models/late_fusion.py:6-23).  Same attributes (upsample -- declared but unused, like the reference --,
fusion, final), state-dict keys ``fusion.{0,1,3,4,6,7,9}.*`` and init; channel 0 = f, channel 1 = g, so the
argument order of the caller matters (LF.py:90 passes (feat, im); run_spatialstream.py:138 passes (out, weighted)).
"""
import torch
import torch.nn as nn

from ..functions import Cat2Planes
from ..utils import FusedSequential, init_like_reference


class late_fusion(nn.Module):
    def __init__(self):
        super(late_fusion, self).__init__()
        self.upsample = nn.Upsample(scale_factor=16)
        layers, cin = [], 2
        for cout in (32, 32, 8):
            layers += [nn.Conv2d(cin, cout, kernel_size=3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True)]
            cin = cout
        layers.append(nn.Conv2d(cin, 1, kernel_size=1, padding=0))
        self.fusion = FusedSequential(*layers)
        self.final = nn.Sigmoid()
        self._initialize_weights()

    def forward(self, f, g):
        if not (f.is_cuda and g.is_cuda):
            raise RuntimeError("late_fusion: expected HIP ('cuda') tensors -- this package has no CPU path")
        # (B,2,H,W) NCHW in one kernel (functions.Cat2Planes: any alignment, gradients flow to f and g as views): read directly
        # by the first conv kernel
        fused = Cat2Planes.apply(f, g)
        return self.fusion(fused, fuse_sigmoid=True)     # fusion stack + self.final fused

    def _initialize_weights(self):
        init_like_reference(self)

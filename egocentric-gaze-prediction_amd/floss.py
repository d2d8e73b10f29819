"""Mirror of the reference's ``floss.py``: gaze-distance-weighted binary cross-entropy (floss.py:5-41),
computed entirely on the device (per-sample arg-max-set centroid, weight map, clamped BCE, mean) by
``egz_floss_fwd`` / ``egz_floss_bwd``."""
import torch.nn as nn

from .functions import FlossLoss


class floss(nn.Module):
    def __init__(self):
        super(floss, self).__init__()

    def forward(self, input, target):
        return FlossLoss.apply(input, target, True)


class BCELoss(nn.Module):
    """torch.nn.BCELoss() replacement for ``--loss_function != 'f'`` (SP.py:103-106, LF.py:73-76)."""

    def forward(self, input, target):
        return FlossLoss.apply(input, target, False)

"""Mirror of the reference's root `utils.py` (the module its training scripts import as `from utils import *`,
stem/trainSTEM.py:20, stem_roi/train_stem_roi.py): criteria, optimiser factory, and the two small host helpers.

    EMLoss, RateDistortionLoss, PixelwiseRateDistortionLoss, quality2lambda   -> .losses   (utils.py:8-74, 97-101)
    configure_optimizers                                                        -> .optim    (utils.py:104-135)
    MovingAverage, save_checkpoint                                              here        (utils.py:77-94, 138-139)
"""
from collections import deque

import torch

from .losses import EMLoss, PixelwiseRateDistortionLoss, RateDistortionLoss, quality2lambda  # noqa: F401
from .optim import configure_optimizers  # noqa: F401

__all__ = ["EMLoss", "RateDistortionLoss", "PixelwiseRateDistortionLoss", "MovingAverage", "quality2lambda",
           "configure_optimizers", "save_checkpoint"]


class MovingAverage:
    """Mean of the last `size` values fed to `next()` (the variable-rate trainer's loss gate uses it to spot
    diverging iterations, stem_roi/train_stem_roi.py).  Same attribute names as upstream: `.queue`, `.Max_size`."""

    def __init__(self, size):
        self.Max_size = size
        self.queue = deque()

    def next(self, val):
        q = self.queue
        q.append(val)
        while len(q) > self.Max_size:
            q.popleft()
        return float(sum(q)) / len(q)


def save_checkpoint(state, filename="checkpoint.pth.tar"):
    """torch.save of the training state dictionary ({"epoch", "state_dict", "loss", "optimizer", "aux_optimizer",
    "lr_scheduler"} in stem/trainSTEM.py:286-297).  Optimiser entries produced by FusedClipAdam.state_dict() are in
    torch.optim.Adam's layout, so checkpoints interchange with the reference in both directions."""
    torch.save(state, filename)

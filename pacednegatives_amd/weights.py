"""Difficulty weights of the paced-negatives schedules (mirror of ``pacednegatives/weights.py``).

``EtaWeights`` (reference ``:10-45``) and ``LCEWeights`` (``:47-78``) threshold a per-example loss at a learnable
``eta``; ``Weights`` (``:80-94``) keeps one free weight per training example.  The reference loops over the batch on
the host and lets autograd find the slope in ``eta``; here one kernel writes the weights and their slope
(``paced_eta_weights`` in include/paced_hip.h), and ``eta`` is a 1-element fp32 device tensor.

(The live training step uses its own variant, ``pacednegatives_amd.lceT5.LCEWeights``, reference ``lceT5.py:82-99``.)
"""
from __future__ import annotations

import numpy as np
import torch

from . import kernels as K

F32 = torch.float32


class _ThresholdWeights:
    FORM = 1

    def __init__(self, eta: float, device=None):
        self.device = torch.device(device) if device else torch.device("cuda")
        self.eta = torch.tensor([float(eta)], dtype=F32, device=self.device)
        self.mask = 0.0
        self.slope = None            # d w_i / d eta of the last forward()

    def parameters(self):
        return [self.eta]

    def set_mask(self, mask: float):
        self.mask = float(mask)

    def _run(self, loss, eta, want_slope):
        loss = loss.to(self.device, F32).contiguous()
        w = torch.empty_like(loss)
        dw = torch.empty_like(loss) if want_slope else None
        K.eta_weights(loss, eta, w, dw, mask=self.mask, form=self.FORM)
        return w, dw

    def no_grad(self, loss, eta):
        """Weights at a given threshold (a float or a 1-element tensor), nothing recorded."""
        if not torch.is_tensor(eta):
            eta = torch.tensor([float(eta)], dtype=F32, device=self.device)
        return self._run(loss, eta.to(self.device, F32).reshape(-1)[:1].contiguous(), False)[0]

    def forward(self, loss=None):
        """Weights at the current ``eta``; ``self.slope`` holds d w / d eta for the caller's meta gradient."""
        w, self.slope = self._run(loss, self.eta, True)
        return w

    __call__ = forward


class EtaWeights(_ThresholdWeights):
    """w = 1 - loss/eta where loss <= eta, else loss * mask * eta; ``clamp()`` keeps eta in [min, max]."""
    FORM = 1

    def __init__(self, eta: float, device=None, min=-np.log(0.5) * 0.5, max=10):
        super().__init__(eta, device)
        self.min, self.max = float(min), float(max)
        self.weighting = lambda x, y: (-x / y) + 1

    def clamp(self):
        self.eta.clamp_(self.min, self.max)


class LCEWeights(_ThresholdWeights):
    """w = loss/eta where loss <= eta, else loss * mask * eta."""
    FORM = 0

    def __init__(self, eta: float, device=None):
        super().__init__(eta, device)
        self.weighting = lambda x, y: (x / y)


class Weights:
    """One weight per (step, position in batch), initially 1; ``set_weight`` stores sigmoid(val)."""

    def __init__(self, shape, device=None):
        self.shape = shape
        self.device = torch.device(device) if device else torch.device("cuda")
        self.weight = torch.ones(shape, dtype=F32, device=self.device)

    def set_weight(self, idx, val):
        self.weight[idx] = torch.sigmoid(val.to(self.device, F32))

    def __getitem__(self, idx):
        return self.weight[idx].clone()

    def forward(self, loss=None, idx=0):
        return self.weight[idx].clone()

    __call__ = forward

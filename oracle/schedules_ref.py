"""Restatement of the threshold weights of ``pacednegatives/weights.py`` and of the pace schedules under ``old/``.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Pinned by ``tests/golden/schedules.npz``, generated from the
reference's own ``EtaWeights.no_grad`` / ``LCEWeights.no_grad`` / ``Weights`` (``pacednegatives/weights.py``),
``interpolate_scalar`` (``old/interp.py:10-18``) and ``LevelWrapper.check_success`` (``old/levels.py:35-39``) by
``tools/make_golden_schedules.py``.  The reference's differentiable ``forward`` raises on CPU (in-place write into a
view of a leaf), so the slope in eta is pinned by autograd through the functional form below: parity unpinned
against the reference for that one quantity.
"""
from __future__ import annotations

import torch


def eta_weights(loss: torch.Tensor, eta: torch.Tensor, mask: float, form: int) -> torch.Tensor:
    """``weights.py:36-45`` (form 1, EtaWeights: weighting = 1 - x/eta) and ``:69-78`` (form 0, LCEWeights: x/eta);
    above the threshold both give loss * mask * eta.  Differentiable in eta."""
    easy = loss / eta if form == 0 else 1.0 - loss / eta
    return torch.where(loss > eta, loss * mask * eta, easy)


def eta_weights_loop(loss: torch.Tensor, eta: float, mask: float, form: int) -> torch.Tensor:
    """The reference's per-element loop (``weights.py:26-34`` / ``:59-67``), literally."""
    out = torch.zeros(loss.shape)
    for i in range(len(loss)):
        if loss[i] > eta:
            out[i] = loss[i] * mask * eta
        else:
            out[i] = (loss[i] / eta) if form == 0 else (-loss[i] / eta) + 1
    return out


def eta_bound_objective(pce: torch.Tensor, nce: torch.Tensor, eta: torch.Tensor, mask: float):
    """``old/eta_bound.py:73-78``: returns (J, v) with ce = (pce + nce)/2, v = EtaWeights(ce),
    J = mean(pce v) + mean(nce v) - sum(v)."""
    ce = torch.div(pce + nce, 2)
    v = eta_weights(ce, eta, mask, 1)
    return torch.mean(pce * v) + torch.mean(nce * v) - torch.sum(v), v


def interpolate_scalar(start_value, end_value, num_steps):
    """``old/interp.py:10-18``."""
    step_size = (end_value - start_value) / num_steps

    def value(step):
        if step > num_steps:
            return end_value
        return start_value + step * step_size
    return value


def success_rate(pos_logits: torch.Tensor, neg_logits: torch.Tensor, rel: int, nrel: int) -> float:
    """``old/levels.py:35-39``: share of pairs whose positive has the higher P("true") at the first position."""
    pos = pos_logits[:, 0, (rel, nrel)].softmax(dim=-1)[:, 0]
    neg = neg_logits[:, 0, (rel, nrel)].softmax(dim=-1)[:, 0]
    return torch.mean((pos > neg).float()).item()


def level_update(difficulty: float, rates, threshold: float, n_neg: int) -> float:
    """``old/levels.py:74-78``: one list position harder when the window's mean success rate beats the threshold."""
    import numpy as np
    if np.mean(rates) > threshold:
        return min(1.0, difficulty + 1 / n_neg)
    return difficulty


def sample_weight_update(ce_p: torch.Tensor, ce_n: torch.Tensor, v: torch.Tensor, lr: float) -> torch.Tensor:
    """``pairwrapper.py:219-239`` (NewWrapper.meta_loop): J = sum(ce_p v)/len + sum(ce_n v)/len - sum(v);
    the stored weight becomes sigmoid(v - lr dJ/dv) (``weights.py:87-88``)."""
    v = v.clone().requires_grad_(True)
    j = torch.sum(ce_p * v) / len(ce_p) + torch.sum(ce_n * v) / len(ce_n) - torch.sum(v)
    (g,) = torch.autograd.grad(j, (v,))
    return torch.sigmoid(v.detach() - lr * g)


def adam_step(p: float, g: float, m: float, v: float, t: int, lr: float, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam on one scalar (``old/eta_bound.py:49``), returns (p, m, v)."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    mhat, vhat = m / (1 - b1 ** t), v / (1 - b2 ** t)
    return p - lr * mhat / (vhat ** 0.5 + eps), m, v

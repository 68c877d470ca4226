"""Restatement of the reference's LCE loss, difficulty weights, optimiser and training step.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

from .t5_ref import T5Dims, t5_forward

TRUE_ID, FALSE_ID, EOS_ID, PAD_ID = 1176, 6136, 1, 0   # T5 vocabulary ids of "true", "false", </s>, <pad>


def pair_loss(plogits: torch.Tensor, nlogits: torch.Tensor, op: torch.Tensor, on: torch.Tensor,
              n: int, use_mean: bool = False, ignore_index: int = -100) -> torch.Tensor:
    """``pacednegatives/lceT5.py:116-123`` together with the caller's reshape at ``:134,151``.

    plogits [B,T,V]; nlogits [B*n,T,V] (the caller views it as [-1, n, V] first, which does not
    move data); labels op [B,T], on [B*n,T].  Result row r = pce_flat[r] + sum (or mean) of
    nce_flat[r*n:(r+1)*n] where nce_flat is indexed by (doc*T + t): a [B*T] vector."""
    V = plogits.shape[-1]
    pce = F.cross_entropy(plogits.reshape(-1, V), op.reshape(-1), ignore_index=ignore_index, reduction="none")
    nce = F.cross_entropy(nlogits.reshape(-1, V), on.reshape(-1), ignore_index=ignore_index, reduction="none")
    nce = nce.view(-1, n)
    nce = nce.mean(dim=1) if use_mean else nce.sum(dim=1)
    return pce + nce


def lce_weights(loss: torch.Tensor, eta: torch.Tensor) -> torch.Tensor:
    """``pacednegatives/lceT5.py:91-99`` vectorised: w_i = loss_i/eta where loss_i <= eta, else
    loss_i*0*eta (a zero that still carries a zero gradient to eta)."""
    return torch.where(loss > eta, loss * 0.0 * eta, loss / eta)


def lce_weights_loop(loss: torch.Tensor, eta: torch.Tensor) -> torch.Tensor:
    """The same, written as the reference's element loop (used to check the vectorised form)."""
    w = torch.zeros(loss.size())
    out = []
    for i in range(len(loss)):
        if loss[i] > eta:
            out.append(loss[i] * torch.zeros(1) * eta)
        else:
            out.append((loss[i] / eta).reshape(1))
    return torch.cat(out) if out else w


def linear_schedule(step: int, warmup_steps: int, total_steps: int) -> float:
    """``get_linear_schedule_with_warmup`` multiplier; same formula as ``pacednegatives/pairwrapper.py:16-25``."""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(total_steps - step) / float(max(1, total_steps - warmup_steps)))


class HFAdamW:
    """``transformers.AdamW`` as the reference constructs it (``lceT5.py:171-172``): betas (0.9, 0.999),
    eps 1e-6, weight_decay 0, correct_bias True.  Update rule (published HF source, class removed in
    transformers 5.x, hence parity unpinned): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
    denom = sqrt(v) + eps; p -= lr * sqrt(1-b2^t)/(1-b1^t) * m/denom."""

    def __init__(self, params: Dict[str, torch.Tensor], lr: float, betas=(0.9, 0.999), eps: float = 1e-6,
                 variant: str = "hf"):
        # variant "torch" applies eps after the second-moment bias correction, as torch.optim.AdamW
        # does: denom = sqrt(v)/sqrt(1-b2^t) + eps.  It exists only so that the step orchestration can
        # be pinned against a fixture that had to be recorded with torch.optim.AdamW.
        assert variant in ("hf", "torch")
        self.variant = variant
        self.params, self.lr, self.betas, self.eps = params, lr, betas, eps
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = 0

    def step(self, grads: Dict[str, torch.Tensor], lr_mult: float = 1.0):
        b1, b2 = self.betas
        self.t += 1
        step_size = self.lr * lr_mult * math.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)
        for k, p in self.params.items():
            g = grads[k]
            self.m[k].mul_(b1).add_(g, alpha=1.0 - b1)
            self.v[k].mul_(b2).addcmul_(g, g, value=1.0 - b2)
            if self.variant == "hf":
                denom = self.v[k].sqrt().add_(self.eps)
                p.addcdiv_(self.m[k], denom, value=-step_size)
            else:
                denom = (self.v[k].sqrt() / math.sqrt(1.0 - b2 ** self.t)).add_(self.eps)
                p.addcdiv_(self.m[k], denom, value=-self.lr * lr_mult / (1.0 - b1 ** self.t))


def training_step(params: Dict[str, torch.Tensor], dims: T5Dims, eta: torch.Tensor,
                  opt: HFAdamW, meta_opt: HFAdamW, step_idx: int,
                  pos_ids, pos_labels, neg_ids, neg_labels,
                  n: int, use_mean: bool, warmup_steps: int, total_steps: int,
                  emulate_bf16: bool = False) -> Dict[str, torch.Tensor]:
    """One ``LCEModel.training_step`` (``pacednegatives/lceT5.py:125-168``) with dropout 0.

    ``params`` (leaf tensors) and ``eta`` ([1]) are updated in place.  ``step_idx`` is the number of
    scheduler steps already taken (LambdaLR evaluates the multiplier at that index for this update)."""
    mult = linear_schedule(step_idx, warmup_steps, total_steps)
    with torch.no_grad():
        pl = t5_forward(params, dims, pos_ids, pos_labels, emulate_bf16)
        nl = t5_forward(params, dims, neg_ids, neg_labels, emulate_bf16)
    loss = pair_loss(pl, nl, pos_labels, neg_labels, n, use_mean)                      # :131-135

    eta_leaf = eta.detach().clone().requires_grad_(True)
    weights = lce_weights(loss, eta_leaf)                                              # :137
    meta_loss = weights * loss
    meta_loss.mean().backward()                                                        # :140-141
    eta_grad = eta_leaf.grad.detach().clone()
    meta_opt.step({"eta": eta_grad}, mult)                                             # :142-143

    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    pl = t5_forward(leaves, dims, pos_ids, pos_labels, emulate_bf16)                   # :149-151
    nl = t5_forward(leaves, dims, neg_ids, neg_labels, emulate_bf16)
    main_loss = pair_loss(pl, nl, pos_labels, neg_labels, n, use_mean)
    main_loss.mean().backward()                                                        # :154-155
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    opt.step(grads, mult)                                                              # :156-157
    return {
        "loss": loss.detach(), "weights": weights.detach(), "meta_loss": meta_loss.detach().mean(),
        "eta_grad": eta_grad, "eta": eta.detach().clone(), "main_loss": main_loss.detach(),
        "grads": grads, "plogits": pl.detach(), "nlogits": nl.detach(),
    }

"""Pair-loss training loops with a paced difficulty (mirror of ``pacednegatives/pairwrapper.py`` and of the pace
schedules built on it under ``old/``).

All of them train the same model on (positive, one negative) text pairs drawn by ``TripletLoader.get_batch(j, w)`` at
difficulty ``w`` with the loss mean(CE_pos) + mean(CE_neg) on the single target word "true"/"false"; they differ in
how the difficulty moves:

* ``InterpWrapper``  -- linear in the number of examples seen (``old/interp.py:10-18,84``);
* ``LevelWrapper``   -- one list position harder whenever the windowed success rate P(true|pos) > P(true|neg) beats a
  threshold (``old/levels.py:35-39,74-78``);
* ``EtaWrapper``     -- difficulty = eta of ``EtaWeights``, moved by Adam on the meta objective
  mean(pce v) + mean(nce v) - sum(v) and clamped to (0, 1] (``old/eta_bound.py:60-93,164``);
* ``NewWrapper``     -- one free weight per example, updated by a gradient step on the same objective and squashed by
  a sigmoid (``pairwrapper.py:208-284``).

The T5 forward/backward, the cross-entropy, the weights and both optimisers run in libpaced_hip.so.  The model's
decoder kernels are built for the two target positions of the live lceT5 step; the one-word targets used here are
run as [word, ignored]: causal decoding makes the first position identical, and the ignored row has no gradient.
"""
from __future__ import annotations

import math
import time
from collections import defaultdict

import numpy as np
import torch

from . import kernels as K
from .lceT5 import FusedAdamW, LinearWarmupSchedule
from .weights import EtaWeights, Weights

RND = 42
F32, BF16 = torch.float32, torch.bfloat16


def linear_warmup_multiplier(step, num_warmup_steps, num_training_steps):
    """The linear warm-up / linear decay factor both schedules of this package use (``lceT5.LinearWarmupSchedule``)."""
    sched = LinearWarmupSchedule.__new__(LinearWarmupSchedule)
    sched.warm, sched.total = num_warmup_steps, num_training_steps
    return sched.multiplier(step)


def adjust_lr(optimizer, init_lr, num_warmup_steps, num_training_steps):
    """``update(step)`` sets every parameter group's lr to init_lr x the linear warm-up factor (same contract as the
    reference's ``pairwrapper.adjust_lr``, ``pairwrapper.py:16-25``)."""
    def update(step):
        lr = init_lr * linear_warmup_multiplier(step, num_warmup_steps, num_training_steps)
        for group in optimizer.param_groups:
            group["lr"] = lr
    return update


def interpolate_scalar(start_value, end_value, num_steps):
    """Pace that moves linearly from ``start_value`` to ``end_value`` over ``num_steps`` steps and stays there
    (contract of the reference's ``old/interp.py:10-18``)."""
    slope = (end_value - start_value) / num_steps
    return lambda step: end_value if step > num_steps else start_value + step * slope


class PacedWrapper:
    """Reference ``pairwrapper.py:27-89``: model, tokenizer, AdamW, per-token cross-entropy and ``prep_batch``."""

    def __init__(self, dataset, model_name, batch_size, model_init, tokenizer, lr, ignore_index) -> None:
        self.logs = {"dataset": dataset, "model_name": model_name, "batch_size": batch_size, "lr": lr,
                     "loss": defaultdict(list), "avg_weight": defaultdict(list)}
        self.model = model_init()
        self.device = self.model.params.device
        self.tokenizer = tokenizer
        self.ignore_index = int(ignore_index)
        P = self.model.params
        self.optimizer = FusedAdamW(P.p32, P.g32, P.m, P.v, P.p16, lr=lr)
        self.batch_size = batch_size
        self.dropout = float(self.model.config.dropout_rate)
        self._call = 0

    def prep_batch(self, batch):
        """Token ids of the two text lists and the FIRST token of each target word as a [B, 1] column (``:76-89``)."""
        px, nx, o_p, o_n = batch
        tok = self.tokenizer
        px = tok(px, padding=True, return_tensors="pt")["input_ids"].to(self.device)
        nx = tok(nx, padding=True, return_tensors="pt")["input_ids"].to(self.device)
        o_p = tok(o_p, padding=True, return_tensors="pt")["input_ids"][:, 0].view(-1, 1).to(self.device)
        o_n = tok(o_n, padding=True, return_tensors="pt")["input_ids"][:, 0].view(-1, 1).to(self.device)
        return px, nx, o_p, o_n

    # ---- one forward of the model on a [B, 1] target column
    def _forward(self, ids, target, save):
        labels = torch.cat([target, torch.full_like(target, self.ignore_index)], 1)
        self._call += 1
        p = self.dropout if self.model.training else 0.0
        return self.model.engine.forward(ids, labels, save=save, drop_p=p, seed=(RND * 0x9E3779B1 + self._call) & 0xFFFFFFFF)

    def _ce(self, st, target, want_grad, gscale=1.0):
        """Per-example cross-entropy of the first target position (and bf16 d logits scaled by ``gscale``)."""
        N = target.shape[0]
        labels = torch.cat([target, torch.full_like(target, self.ignore_index)], 1).view(-1).contiguous()
        loss = torch.empty(2 * N, dtype=F32, device=self.device)
        dl = torch.empty(st.logits.shape, dtype=BF16, device=self.device) if want_grad else None
        K.cross_entropy(st.logits, labels, loss, dl, ignore_index=self.ignore_index, gscale=gscale)
        return loss.view(N, 2)[:, 0].contiguous(), dl

    def pair_losses(self, px, nx, o_p, o_n):
        """No-grad CE of positives and negatives, plus the logits of the first position [B, V]."""
        sp, sn = self._forward(px, o_p, False), self._forward(nx, o_n, False)
        pce, _ = self._ce(sp, o_p, False)
        nce, _ = self._ce(sn, o_n, False)
        return pce, nce, sp.logits.view(-1, 2, sp.logits.shape[-1])[:, 0], sn.logits.view(-1, 2, sn.logits.shape[-1])[:, 0]

    def pair_step(self, px, nx, o_p, o_n):
        """loss = mean(CE_pos) + mean(CE_neg); backward; AdamW; schedule.  Returns (loss, pce, nce, first-position logits)."""
        eng = self.model.engine
        sp = self._forward(px, o_p, True)
        pce, dlp = self._ce(sp, o_p, True, gscale=1.0 / px.shape[0])
        plog = sp.logits.view(-1, 2, sp.logits.shape[-1])[:, 0].clone()
        sp.logits = None
        eng.backward(sp, dlp)
        sn = self._forward(nx, o_n, True)
        nce, dln = self._ce(sn, o_n, True, gscale=1.0 / nx.shape[0])
        nlog = sn.logits.view(-1, 2, sn.logits.shape[-1])[:, 0].clone()
        sn.logits = None
        eng.backward(sn, dln)
        self.optimizer.step()
        self.scheduler.step()
        self.optimizer.zero_grad()
        return float((pce.mean() + nce.mean()).item()), pce, nce, plog, nlog

    def _schedules(self, total_steps, warmup_steps, per_batch):
        warm = (warmup_steps // self.train_loader.batch_size if per_batch else warmup_steps) if warmup_steps else total_steps // 100
        return LinearWarmupSchedule(self.optimizer, warm, total_steps)

    def success(self, plog, nlog):
        """P("true") of the positive above that of the negative, over ("true", "false") at the first position."""
        pos = plog[:, (self.REL, self.NREL)].float().softmax(dim=-1)[:, 0]
        neg = nlog[:, (self.REL, self.NREL)].float().softmax(dim=-1)[:, 0]
        return torch.mean((pos > neg).float()).item()

    def _word_ids(self):
        self.REL = int(self.tokenizer(["true"], padding=True, return_tensors="pt")["input_ids"][0, 0])
        self.NREL = int(self.tokenizer(["false"], padding=True, return_tensors="pt")["input_ids"][0, 0])


class StdWrapper(PacedWrapper):
    """Reference ``pairwrapper.py:91-206``.  Its meta loop differentiates through a parameter update of a second
    model (``create_graph=True``) that the reference never creates (``self.meta_model``, ``:51`` is commented out), so
    the reference class fails on its first step; it is not reproduced."""

    def train(self, train_loader, epochs, warmup_steps):
        raise NotImplementedError("StdWrapper needs the second-order meta model the reference itself no longer builds "
                                  "(pairwrapper.py:51); use NewWrapper or one of the pace schedules")


class NewWrapper(PacedWrapper):
    """Reference ``pairwrapper.py:208-284``: free per-example weights."""

    def meta_loop(self, j):
        px, nx, o_p, o_n = self.prep_batch(self.train_loader.get_batch(j, self.weights[j]))
        pce, nce, _, _ = self.pair_losses(px, nx, o_p, o_n)
        v = self.weights.forward(idx=j)
        # J = sum(ce_p v)/len + sum(ce_n v)/len - sum(v)  =>  dJ/dv = ce_p/len + ce_n/len - 1
        g = pce / pce.numel() + nce / nce.numel() - 1.0
        self.weights.set_weight(idx=j, val=v - self.scheduler.get_last_lr()[0] * g)

    def main_loop(self, j):
        px, nx, o_p, o_n = self.prep_batch(self.train_loader.get_batch(j, self.weights[j]))
        return self.pair_step(px, nx, o_p, o_n)[0]

    def train(self, train_loader, epochs, warmup_steps):
        torch.manual_seed(RND)
        steps = len(train_loader) // self.batch_size
        self.weights = Weights((steps, self.batch_size), device=self.device)
        self.train_loader = train_loader
        self.epochs = epochs
        self.scheduler = LinearWarmupSchedule(self.optimizer, warmup_steps if warmup_steps else len(train_loader) // 4,
                                              epochs * len(train_loader))
        start = time.time()
        for epoch in range(epochs):
            for j in range(steps):
                self.meta_loop(j)
                loss = self.main_loop(j)
                self.logs["loss"]["train"].append(loss)
                self.logs["avg_weight"][epoch].append(self.weights[j].mean().item())
        self.logs["time"] = time.time() - start
        return self.logs


class InterpWrapper(PacedWrapper):
    """Reference ``old/interp.py:20-98``."""

    def __init__(self, dataset, model_name, batch_size, model_init, tokenizer, lr, ignore_index,
                 start_difficulty, max_difficulty, interpolate_steps) -> None:
        super().__init__(dataset, model_name, batch_size, model_init, tokenizer, lr, ignore_index)
        self.difficulty = self.start_difficulty = start_difficulty
        self.max_difficulty, self.interpolate_steps = max_difficulty, interpolate_steps
        self.interpolate_difficulty = interpolate_scalar(start_difficulty, max_difficulty, interpolate_steps)
        self._word_ids()
        self.logs["difficulty"] = {"train": []}
        self.logs["lr"] = []

    def main_loop(self, j):
        return self.pair_step(*self.prep_batch(self.train_loader.get_batch(j, self.difficulty)))[0]

    def train(self, train_loader, total_steps, warmup_steps):
        torch.manual_seed(RND)
        self.train_loader, self.total_steps = train_loader, total_steps
        self.scheduler = self._schedules(total_steps, warmup_steps, per_batch=True)
        start = time.time()
        for i in range(total_steps // train_loader.batch_size):
            loss = self.main_loop(i)
            self.difficulty = self.interpolate_difficulty(i * train_loader.batch_size)
            self.logs["loss"]["train"].append(loss)
            self.logs["lr"].append(self.scheduler.get_last_lr()[0])
            self.logs["difficulty"]["train"].append(self.difficulty)
        self.logs["time"] = time.time() - start
        return self.logs


class LevelWrapper(PacedWrapper):
    """Reference ``old/levels.py:11-98``."""

    def __init__(self, dataset, model_name, batch_size, model_init, tokenizer, lr, ignore_index,
                 success_threshold, heuristic_step_check) -> None:
        super().__init__(dataset, model_name, batch_size, model_init, tokenizer, lr, ignore_index)
        self.difficulty = 0.0
        self.success_threshold, self.heuristic_step_check = success_threshold, heuristic_step_check
        self.success_rate = []
        self._word_ids()
        self.logs["difficulty"] = {"train": []}
        self.logs["lr"] = []

    def check_success(self, plog, nlog):
        self.success_rate.append(self.success(plog, nlog))

    def main_loop(self, j):
        loss, _pce, _nce, plog, nlog = self.pair_step(*self.prep_batch(self.train_loader.get_batch(j, self.difficulty)))
        self.check_success(plog, nlog)
        return loss

    def train(self, train_loader, total_steps, warmup_steps):
        torch.manual_seed(RND)
        self.train_loader, self.total_steps = train_loader, total_steps
        self.scheduler = self._schedules(total_steps, warmup_steps, per_batch=False)
        start = time.time()
        for i in range(total_steps // train_loader.batch_size):
            if i % self.heuristic_step_check == 0 and len(self.success_rate) > 0:
                if np.mean(self.success_rate) > self.success_threshold:
                    self.difficulty = min(1.0, self.difficulty + 1 / train_loader.dataset.n_neg)
                self.success_rate = []
            loss = self.main_loop(i)
            self.logs["loss"]["train"].append(loss)
            self.logs["success_rate"] = self.success_rate[-1]
            self.logs["lr"].append(self.scheduler.get_last_lr()[0])
            self.logs["difficulty"]["train"].append(self.difficulty)
        self.logs["time"] = time.time() - start
        return self.logs


class EtaWrapper(PacedWrapper):
    """Reference ``old/eta_bound.py:22-174``: difficulty = eta, eta learned by Adam on the meta objective."""

    def __init__(self, eta, dataset, model_name, batch_size, model_init, tokenizer, lr, meta_lr, ignore_index) -> None:
        super().__init__(dataset, model_name, batch_size, model_init, tokenizer, lr, ignore_index)
        self.weights = EtaWeights(eta, device=self.device, min=0. + 1e-10, max=1.)
        self.logs["eta"] = []
        self.logs["loss"] = {"main": [], "meta": []}
        self.logs["difficulty"] = []
        self.logs["lr"] = {"main": [], "meta": []}
        self.logs["success_rate"] = []
        self.logs["probs"] = []
        self.running_rate = []
        self._word_ids()
        self.meta_lr = meta_lr
        # torch.optim.Adam(eta, lr=meta_lr) (:49) through the fused kernel: Adam divides by sqrt(v_hat) + eps, the kernel
        # by sqrt(v) + eps', so eps' = eps * sqrt(1 - beta2^t) is set before every step
        self._eta_g = torch.zeros(1, dtype=F32, device=self.device)
        self._eta_m, self._eta_v = torch.zeros_like(self._eta_g), torch.zeros_like(self._eta_g)
        self.meta_optimizer = FusedAdamW(self.weights.eta, self._eta_g, self._eta_m, self._eta_v, None, lr=meta_lr, eps=1e-8)
        self._meta_out = torch.zeros(4, dtype=F32, device=self.device)

    def meta_loop(self, j):
        px, nx, o_p, o_n = self.prep_batch(self.train_loader.get_batch(j, self.difficulty))
        pce, nce, _, _ = self.pair_losses(px, nx, o_p, o_n)
        v = torch.empty_like(pce)
        K.eta_bound_meta(pce, nce, self.weights.eta, v, self._meta_out, mask=self.weights.mask)
        self._eta_g.copy_(self._meta_out[1:2])
        opt = self.meta_optimizer
        opt.eps = 1e-8 * math.sqrt(1.0 - opt.betas[1] ** (opt.t + 1))
        opt.step()
        opt.zero_grad()
        self.weights.clamp()
        self.meta_scheduler.step()
        meta_loss = float(self._meta_out[0].item())
        self.logs["loss"]["meta"].append(meta_loss)
        return meta_loss

    def main_loop(self, j):
        loss, pce, nce, plog, nlog = self.pair_step(*self.prep_batch(self.train_loader.get_batch(j, self.difficulty)))
        self.logs["probs"].append(self.success(plog, nlog))
        ce = (pce + nce) / 2
        self.running_rate.append(torch.mean((ce < self.weights.eta).float()).item())
        return loss

    def train(self, train_loader, total_steps, warmup_steps):
        torch.manual_seed(RND)
        self.train_loader, self.total_steps = train_loader, total_steps
        self.difficulty = 0.0
        self.scheduler = self._schedules(total_steps, warmup_steps, per_batch=True)
        warm = warmup_steps // train_loader.batch_size if warmup_steps else total_steps // 100
        self.meta_scheduler = LinearWarmupSchedule(self.meta_optimizer, warm, total_steps)
        start = time.time()
        for i in range(total_steps // train_loader.batch_size):
            self.meta_loop(i)
            loss = self.main_loop(i)
            self.logs["loss"]["main"].append(loss)
            self.logs["lr"]["main"].append(self.scheduler.get_last_lr()[0])
            self.logs["lr"]["meta"].append(self.meta_scheduler.get_last_lr()[0])
            self.logs["difficulty"].append(self.difficulty)
            self.logs["eta"].append(self.weights.eta.item())
            self.difficulty = self.weights.eta.item()
        self.logs["time"] = time.time() - start
        return self.logs

"""A small stand-in for the part of ``lightning.Trainer`` that ``train/train_lce.py`` uses
(reference ``train/train_lce.py:76-100``): devices, callbacks with ``on_train_batch_start``,
``max_epochs``, ``max_steps``, ``default_root_dir``, a logger, and ``fit(model, datamodule)``.

One process per GPU: under ``torchrun`` / ``torch.distributed.run`` (RANK, LOCAL_RANK, WORLD_SIZE in
the environment) ``fit`` joins the RCCL process group (backend "nccl"; "gloo" when there is no GPU,
which the CPU tests use) and gives rank r the dataset items r, r+W, ...
"""
from __future__ import annotations

import json
import os
import time

import torch


class JSONLLogger:
    """Writes the step's logged scalars (avg_weight, meta_loss, eta, main_loss, lr-*) as JSON lines."""

    def __init__(self, save_dir: str, name: str = "metrics.jsonl", project=None):
        os.makedirs(save_dir, exist_ok=True)
        self.path = os.path.join(save_dir, name)
        self.project = project
        self._f = open(self.path, "a")

    def log_metrics(self, metrics: dict, step: int):
        rec = {"step": step}
        rec.update({k: (float(v) if hasattr(v, "__float__") else v) for k, v in metrics.items()})
        self._f.write(json.dumps(rec) + "\n")
        self._f.flush()

    def finalize(self):
        self._f.close()


def init_distributed():
    """Join the process group described by the torchrun environment; returns (rank, world, device index).

    With fewer visible GPUs than local ranks (a rehearsal of the multi-rank path on a one-GPU machine) the ranks
    share devices round-robin; RCCL refuses two ranks on one device, so such a run needs PACED_DIST_BACKEND=gloo."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        local %= max(1, torch.cuda.device_count())
    rehearsal = os.environ.get("PACED_FORCE_EXCHANGE") == "1"        # one rank that still exchanges: see parallel.exchange_forced
    if (world > 1 or rehearsal) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("PACED_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class Trainer:
    def __init__(self, devices=1, callbacks=None, logger=None, max_epochs=1, default_root_dir=".", profiler=None,
                 max_steps=None, accelerator=None, strategy=None, log_every_n_steps=1, **_ignored):
        self.devices = devices
        self.callbacks = list(callbacks or [])
        self.logger = logger
        self.max_epochs = max_epochs
        self.max_steps = max_steps
        self.default_root_dir = default_root_dir
        self.profiler = profiler
        self.strategy = strategy
        self.log_every_n_steps = log_every_n_steps
        self.train_dataloader = None
        self.global_step = 0
        self.timings = {}

    def fit(self, model, datamodule):
        rank, world, _local = init_distributed()
        devices = self.devices if isinstance(self.devices, int) else len(self.devices) if isinstance(self.devices, (list, tuple)) else None
        if devices is not None and devices != world:
            # Lightning starts `devices` processes by itself; here they come from torchrun or pacednegatives_amd.launcher
            raise RuntimeError(f"Trainer(devices={self.devices}) but {world} rank process(es) are running: launch one process per "
                               "device (train/train_lce.py --num_gpus N does; or python -m torch.distributed.run --nproc-per-node N)")
        if getattr(datamodule, "dataset", None) is None:
            datamodule.setup()
        self.train_dataloader = datamodule.train_dataloader(rank=rank, world_size=world)
        t_fit = time.time()
        step_time = 0.0
        done = False
        for _epoch in range(self.max_epochs):
            for batch_idx, batch in enumerate(self.train_dataloader):
                for cb in self.callbacks:
                    hook = getattr(cb, "on_train_batch_start", None)
                    if hook:
                        hook(self, model, batch, batch_idx)
                t0 = time.time()
                model.training_step(batch, batch_idx)
                step_time += time.time() - t0
                self.global_step += 1
                if self.logger is not None and rank == 0 and self.global_step % self.log_every_n_steps == 0:
                    self.logger.log_metrics(dict(model.logged), self.global_step)
                if self.max_steps and self.global_step >= self.max_steps:
                    done = True
                    break
            if done:
                break
        self.timings = {"fit_s": time.time() - t_fit, "training_step_s": step_time, "steps": self.global_step}
        if self.profiler and rank == 0:
            print(f"[profiler] fit {self.timings['fit_s']:.2f}s, training_step {step_time:.2f}s over {self.global_step} steps")
        if self.logger is not None and rank == 0:
            self.logger.finalize()

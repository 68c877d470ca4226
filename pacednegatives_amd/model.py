"""Hugging Face-shaped wrapper around the T5 engine.

The reference uses exactly this much of ``T5ForConditionalGeneration`` (pacednegatives/lceT5.py:106,
132-133,149-150; train/train_lce.py:103): construction from a name or directory, a call with
``input_ids=``/``labels=`` whose result exposes ``.logits``, ``.parameters()`` and
``.save_pretrained(dir)``.  Checkpoints are read and written in Hugging Face layout (config.json +
model.safetensors with the usual T5 tensor names) so that ``MonoT5ReRanker(model=dir)`` in the
reference's ``eval.py:23`` can load what this writes.
"""
from __future__ import annotations

import os
import warnings
from types import SimpleNamespace

import torch

from .config import NAMED, T5Dims, resolve
from .params import ParamStore
from .t5_engine import T5Engine


class T5ForConditionalGenerationHIP:
    def __init__(self, dims: T5Dims, device="cuda", seed: int = 0, init: bool = True):
        self.config = dims
        self.params = ParamStore(dims, device=device)
        self.engine = T5Engine(dims, self.params)
        self.training = True
        self._call_counter = 0
        if init:
            self.params.init_random(seed)

    # ---- construction / IO
    @classmethod
    def from_pretrained(cls, name_or_dir: str, device="cuda", seed: int = 0) -> "T5ForConditionalGenerationHIP":
        dims = resolve(name_or_dir)
        if os.path.isdir(name_or_dir):
            m = cls(dims, device=device, init=False)
            st = os.path.join(name_or_dir, "model.safetensors")
            pt = os.path.join(name_or_dir, "pytorch_model.bin")
            if os.path.isfile(st):
                from safetensors.torch import load_file
                state = load_file(st)
            elif os.path.isfile(pt):
                state = torch.load(pt, map_location="cpu", weights_only=True)
            else:
                raise FileNotFoundError(f"{name_or_dir} holds neither model.safetensors nor pytorch_model.bin")
            m.params.load_hf(state)
            return m
        warnings.warn(f"{name_or_dir!r} is a size name, not a local checkpoint directory: there is no network, so the "
                      f"weights are a seeded random initialisation of that architecture (seed {seed})")
        return cls(dims, device=device, seed=seed)

    def save_pretrained(self, directory: str) -> None:
        from safetensors.torch import save_file
        os.makedirs(directory, exist_ok=True)
        self.config.save(directory)
        state = {k: v.to("cpu").contiguous() for k, v in self.params.to_hf("master").items()}
        save_file(state, os.path.join(directory, "model.safetensors"), metadata={"format": "pt"})

    def state_dict(self):
        return self.params.to_hf("master")

    def load_state_dict(self, state, strict: bool = True):
        self.params.load_hf(state)

    # ---- torch.nn.Module-like surface
    def parameters(self):
        for name in self.params.names():
            yield self.params.master(name)

    def named_parameters(self):
        for k, v in self.params.to_hf("master").items():
            yield k, v

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def to(self, *_a, **_k):
        return self

    def __call__(self, input_ids=None, labels=None, attention_mask=None, **_unused):
        """Forward only (no state kept for backward).  ``attention_mask`` is accepted and ignored: the
        reference never passes one (pacednegatives/lceT5.py:53), so padding positions are attended."""
        if input_ids is None or labels is None:
            raise ValueError("input_ids and labels are both required")
        self._call_counter += 1
        p = self.config.dropout_rate if self.training else 0.0
        st = self.engine.forward(input_ids, labels, save=False, drop_p=p, seed=0x5EED0000 ^ self._call_counter)
        N = input_ids.shape[0]
        return SimpleNamespace(logits=st.logits.view(N, 2, self.config.vocab_size))


__all__ = ["T5ForConditionalGenerationHIP", "NAMED"]

#!/usr/bin/env python3
"""Experiment: one no-grad encoder+decoder forward of 544 sequences on one stream against the same sequences as two halves
on two streams (does running a half's attention / norm kernels beside the other half's products pay?)."""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pacednegatives_amd.lceT5 import LCEModel
from pacednegatives_amd.synthetic import token_batch

hp = {"model_name": "t5-base", "total_steps": 1000, "eta": 0.3466, "batch_size": 32, "lr": 1e-3, "meta_lr": 1e-3, "n": 16,
      "warmup_steps": 100, "use_mean": False, "ignore_index": -100, "dropout": 0.1, "seed": 0}
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    model = LCEModel(hp)
eng = model.model.engine
p, n = token_batch(32, 16, 256, model.model.config.vocab_size, seed=1, device="cuda")
ids = torch.cat([p["input_ids"], n["input_ids"]], 0)
lab = torch.cat([p["labels"], n["labels"]], 0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
half = ids.shape[0] // 2


def full():
    eng.forward(ids, lab, save=False, drop_p=0.1, seed=1, want_logits=False)


def split():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        eng.forward(ids[:half], lab[:half], save=False, drop_p=0.1, seed=1, want_logits=False)
    with torch.cuda.stream(s2):
        eng.forward(ids[half:], lab[half:], save=False, drop_p=0.1, seed=2, want_logits=False)
    cur.wait_stream(s1); cur.wait_stream(s2)


for name, fn in (("full", full), ("split", split), ("full", full), ("split", split)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    print(name, round((time.perf_counter() - t0) / 5 * 1e3, 2), "ms", flush=True)

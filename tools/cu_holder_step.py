#!/usr/bin/env python3
"""What a communication kernel beside the backward pass would cost the step: tools/cu_holder.hip holds C compute units for
T ms from the moment the decoder's gradients are final (where a bucketed gradient all-reduce would start), on a stream
of its own, and the benchmark step is timed with and without it.

  hipcc --offload-arch=gfx950 -O2 -fPIC -shared tools/cu_holder.hip -o tools/libcu_holder.so     (here, before the GPU job)
  python tools/cu_holder_step.py [--steps 6] [--ms 5]
"""
import argparse, ctypes, os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pacednegatives_amd.config import resolve
from pacednegatives_amd.lceT5 import LCEModel
from pacednegatives_amd.synthetic import token_batch

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--ms", type=float, default=5.0)
args = ap.parse_args()
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libcu_holder.so"))
lib.hold_cus.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
dev = torch.device("cuda:0")
dims = resolve("t5-base")
B, nneg, S = 32, 16, 256
hp = {"model_name": "t5-base", "total_steps": 100000 // B, "eta": 0.3466, "batch_size": B, "lr": 1e-3, "meta_lr": 1e-3,
      "n": nneg, "warmup_steps": 10000 // B, "use_mean": False, "ignore_index": -100, "dropout": 0.1, "seed": 0}
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    model = LCEModel(hp, device=dev)
batch = token_batch(B, nneg, S, dims.vocab_size, seed=1234, device=dev)
comm = torch.cuda.Stream(device=dev)
cfg = {"wgs": 0, "lds": 0}
eng = model.model.engine
plain_backward = eng.backward


def backward(st, dl, on_segments_done=None):
    def hook(first, last):
        if on_segments_done is not None:
            on_segments_done(first, last)
        if first.startswith("dec.") and cfg["wgs"]:
            comm.wait_stream(torch.cuda.current_stream())
            rc = lib.hold_cus(cfg["wgs"], cfg["lds"], int(args.ms * 1000), ctypes.c_void_p(comm.cuda_stream))
            assert rc == 0, rc
    plain_backward(st, dl, on_segments_done=hook)
    torch.cuda.current_stream().wait_stream(comm)


eng.backward = backward
for i in range(3):
    model.training_step(batch, i)
torch.cuda.synchronize()
it = 3
for rnd in range(2):
    for wgs, lds in ((0, 0), (16, 96 * 1024), (32, 96 * 1024), (64, 96 * 1024), (64, 0), (256, 0)):
        cfg["wgs"], cfg["lds"] = wgs, lds
        model.training_step(batch, it); it += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            model.training_step(batch, it); it += 1
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        print(f"round {rnd}: {wgs:3d} workgroups, {lds // 1024:3d} KB LDS each, held {args.ms} ms per step: {ms:7.2f} ms/step", flush=True)

"""GEMM time of one in-order training step by (M, N, K, flags): where the launch-average TFLOP/s comes from."""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pacednegatives_amd import _lib, kernels as K
from pacednegatives_amd.lceT5 import LCEModel
from pacednegatives_amd.synthetic import token_batch

B, nneg, S = 32, 16, 256
hp = {"model_name": "t5-base", "total_steps": 3000, "eta": 0.3466, "batch_size": B, "lr": 1e-3, "meta_lr": 1e-3, "n": nneg,
      "warmup_steps": 300, "use_mean": False, "ignore_index": -100, "dropout": 0.1, "seed": 0}
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    model = LCEModel(hp, device=torch.device("cuda"))
model.overlap_forwards = False
batch = token_batch(B, nneg, S, model.model.config.vocab_size, seed=1234, device="cuda")
for i in range(2):
    model.training_step(batch, i)
torch.cuda.synchronize()

lib = _lib.load()
records = []


def wrap(name, shape_of):
    real = getattr(lib, name)

    def call(*args):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = real(*args)
        e1.record()
        records.append((shape_of(args), e0, e1))
        return rc
    setattr(lib, name, call)


# paced_gemm_bf16(a, b, c, aux, M, N, K, lda, ldb, ldc, ldaux, trans_a, trans_b, epilogue, ...)
wrap("paced_gemm_bf16", lambda a: (a[4], a[5], a[6], 1, a[11], a[12], a[13]))
# paced_gemm_bf16_batched(a, b, c, M, N, K, lda, ldb, ldc, trans_a, trans_b, epilogue, alpha, splits, batch, ...)
wrap("paced_gemm_bf16_batched", lambda a: (a[3], a[4], a[5], a[14], a[9], a[10], a[11]))
model.training_step(batch, 2)
torch.cuda.synchronize()
agg = {}
for key, e0, e1 in records:
    t = agg.setdefault(key, [0, 0.0])
    t[0] += 1
    t[1] += e0.elapsed_time(e1)
tot_t = sum(v[1] for v in agg.values())
tot_f = sum(2.0 * k[0] * k[1] * k[2] * k[3] * v[0] for k, v in agg.items())
print(f"{len(records)} launches, {tot_t:.2f} ms, {tot_f / tot_t / 1e9:.0f} TFLOP/s average")
print(f"{'M':>7} {'N':>6} {'K':>7} {'batch':>5} ta tb epi {'calls':>5} {'ms':>8} {'share':>6} {'us/call':>8} {'TFLOP/s':>8}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    fl = 2.0 * k[0] * k[1] * k[2] * k[3]
    print(f"{k[0]:7d} {k[1]:6d} {k[2]:7d} {k[3]:5d} {k[4]:2d} {k[5]:2d} {k[6]:3d} {v[0]:5d} {v[1]:8.3f} {v[1] / tot_t:6.1%} {v[1] / v[0] * 1e3:8.1f} {fl * v[0] / v[1] / 1e9:8.0f}")

"""Host time to enqueue one training step (up to the step's single read-back) against the whole step."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pacednegatives_amd.lceT5 import LCEModel
from pacednegatives_amd.synthetic import token_batch
B, nneg, S = 32, 16, 256
hp = {"model_name": "t5-base", "total_steps": 3000, "eta": 0.3466, "batch_size": B, "lr": 1e-3, "meta_lr": 1e-3, "n": nneg,
      "warmup_steps": 300, "use_mean": False, "ignore_index": -100, "dropout": 0.1, "seed": 0}
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    model = LCEModel(hp, device=torch.device("cuda"))
batch = token_batch(B, nneg, S, model.model.config.vocab_size, seed=1234, device="cuda")
marks = []
orig = torch.Tensor.tolist
def tolist(self):
    marks.append(time.perf_counter())
    return orig(self)
torch.Tensor.tolist = tolist
for i in range(3):
    model.training_step(batch, i)
torch.cuda.synchronize()
for i in range(3, 9):
    marks.clear()
    t0 = time.perf_counter()
    model.training_step(batch, i)
    t1 = time.perf_counter()
    print(f"step {i}: host enqueue {1e3 * (marks[0] - t0):6.1f} ms, whole step {1e3 * (t1 - t0):6.1f} ms", flush=True)

"""The N > 1 host path on CPU: two gloo ranks exercise rendezvous, dataset sharding through the Trainer
loop, and the sum-then-scale gradient averaging rule."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["PACED_ROOT"])
import torch
from pacednegatives_amd import parallel
from pacednegatives_amd.lceT5 import ChangeDifficulty, LCEDataModule
from pacednegatives_amd.synthetic import write_synthetic_dataset
from pacednegatives_amd.tokenizer import SyntheticTokenizer
from pacednegatives_amd.trainer import Trainer, init_distributed

rank, world, local = init_distributed()
assert world == 2 and parallel.world_size() == 2 and parallel.rank() == rank

# gradient averaging: sum over ranks, then the optimiser's grad_scale
g = torch.full((5,), float(rank + 1))
parallel.allreduce_sum_(g)
mean = g * parallel.grad_scale()
assert torch.allclose(mean, torch.full((5,), 1.5))

out_dir = os.environ["PACED_OUT"]
path, corpus = write_synthetic_dataset(os.path.join(out_dir, f"r{rank}"), n_records=11, n_neg=16, seed=0) if os.makedirs(os.path.join(out_dir, f"r{rank}"), exist_ok=True) is None else None
dm = LCEDataModule(path, corpus, SyntheticTokenizer(32128), batch_size=2, n=3, init_weight=0.3, num_workers=0)
dm.setup()


class FakeModel:
    """Stands in for LCEModel: records what each rank is fed and moves eta like a real step would."""
    def __init__(self):
        self.difficulty, self.seen, self.paces, self.logged = 0.3, [], [], {}
    def training_step(self, batch, idx):
        p, n = batch
        assert n["input_ids"].shape[0] == 3 * p["input_ids"].shape[0]
        self.seen.append(p["input_ids"].shape[0])
        self.paces.append(trainer.train_dataloader.paced_dataset.weight)
        eta = torch.tensor([self.difficulty + 0.1 * (rank + 1)])
        parallel.allreduce_sum_(eta)
        self.difficulty = float(eta) * parallel.grad_scale()
        self.logged = {"eta": self.difficulty}

model = FakeModel()
trainer = Trainer(devices=2, callbacks=[ChangeDifficulty()], max_epochs=1, default_root_dir=out_dir, strategy="ddp")
trainer.fit(model, dm)
items = parallel.shard_indices(len(dm.dataset), rank, world)
json.dump({"rank": rank, "items": items, "batches": model.seen, "paces": model.paces, "eta": model.difficulty},
          open(os.path.join(out_dir, f"result{rank}.json"), "w"))
torch.distributed.barrier()
torch.distributed.destroy_process_group()
'''


def test_two_rank_gloo_training_loop(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    port = 29500 + os.getpid() % 2000
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PACED_ROOT=ROOT, PACED_OUT=str(tmp_path), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    res = [json.load(open(tmp_path / f"result{r}.json")) for r in range(2)]
    # 11 records over 2 ranks: padded with item 0 to 12 (torch's DistributedSampler rule), 6 each, rank r holds r, r+2, ...
    assert res[0]["items"] == [0, 2, 4, 6, 8, 10] and res[1]["items"] == [1, 3, 5, 7, 9, 0]
    assert res[0]["batches"] == res[1]["batches"] == [2, 2, 2]
    # every rank ends on the same eta (mean of the two local updates each step), and the callback fed it to the sampler
    assert abs(res[0]["eta"] - res[1]["eta"]) < 1e-12
    assert abs(res[0]["eta"] - (0.3 + 3 * 0.15)) < 1e-6
    assert res[0]["paces"] == res[1]["paces"] and abs(res[0]["paces"][1] - 0.45) < 1e-6


@pytest.mark.parametrize("n,world", [(11, 2), (23, 2), (16, 8), (17, 8), (3, 8), (1, 4), (100, 3), (0, 2)])
def test_shard_indices_equal_torchs_distributed_sampler(n, world):
    """The rule Lightning's ddp strategy applies to the reference's sequential loader: DistributedSampler(shuffle=False)."""
    from torch.utils.data import DistributedSampler
    from pacednegatives_amd import parallel
    for r in range(world):
        want = list(DistributedSampler(range(n), num_replicas=world, rank=r, shuffle=False)) if n else []
        assert parallel.shard_indices(n, r, world) == want, (n, world, r)


def test_trainer_refuses_a_device_count_that_is_not_the_world_size():
    from pacednegatives_amd.trainer import Trainer
    t = Trainer(devices=2, max_epochs=1)
    with pytest.raises(RuntimeError, match="rank process"):
        t.fit(object(), object())


LAUNCHED = r"""
import os, sys
r = int(os.environ["RANK"])
assert os.environ["WORLD_SIZE"] == "3" and os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1"
print("{" + f'"rank": {r}, "port": {os.environ["MASTER_PORT"]}' + "}")
print("noise from rank", r)
mode = sys.argv[1]
if mode == "fail" and r == 1:
    sys.exit(7)
if mode == "hang" or (mode == "fail" and r != 1):
    import time; time.sleep(60)
"""


def test_launcher_starts_ranks_relays_rank0_and_reports_failures(tmp_path, capfd):
    """pacednegatives_amd.launcher (what train/train_lce.py --num_gpus N and bench.py --gpus N use when no torchrun
    environment is present): ranks get the torchrun variables, rank 0's stdout is relayed, a failing rank stops the
    others with a non-zero code, and a hung run is stopped at the time limit with code 124."""
    import time
    from pacednegatives_amd import launcher
    script = tmp_path / "ranked.py"
    script.write_text(LAUNCHED)
    assert launcher.launch_ranks(3, [sys.executable, str(script), "ok"], json_only_stdout=True) == 0
    out, err = capfd.readouterr()
    assert out.strip().startswith('{"rank": 0') and "noise" not in out and "noise from rank 0" in err
    t0 = time.time()
    assert launcher.launch_ranks(3, [sys.executable, str(script), "fail"]) == 1
    assert launcher.launch_ranks(3, [sys.executable, str(script), "hang"], timeout_s=2.0) == 124
    assert time.time() - t0 < 40

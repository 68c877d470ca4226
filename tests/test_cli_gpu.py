"""train/train_lce.py on the device (reference train/train_lce.py:20-103): flags -> hparams -> LCEModel ->
LCEDataModule -> Trainer.fit with ChangeDifficulty -> save_pretrained(out_dir/finalmodel)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "train", "train_lce.py")
KEYS = {"avg_weight", "meta_loss", "eta", "main_loss", "lr-AdamW", "lr-AdamW-meta"}


def _metrics(out_dir):
    with open(os.path.join(out_dir, "metrics.jsonl")) as f:
        return [json.loads(line) for line in f]


def test_cli_in_a_child_process(tmp_path):
    """The command line as grid_var.py:28 spells it, plus --synthetic / --model_name (nothing can be downloaded here)."""
    out = str(tmp_path / "run")
    cmd = [sys.executable, CLI, "--data", "unused", "--dataset_name", "unused", "--out_dir", out, "--batch_size", "4", "--lr", "0.001",
           "--eta", "30.0", "--var", "0.05", "--n", "2", "--num_gpus", "1", "--num_workers", "2", "--max_steps", "4",
           "--total_steps", "400", "--warmup_steps", "40", "--synthetic", "64", "--model_name", "t5-tiny", "--dropout", "0.1"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-3000:]
    recs = _metrics(out)
    assert [m["step"] for m in recs] == [1, 2, 3, 4]
    for m in recs:
        assert KEYS <= set(m) and all(isinstance(m[k], float) and m[k] == m[k] for k in KEYS), m
    # total_steps // batch_size = 100, warmup_steps // batch_size = 10 (train_lce.py:55,61): linear warm-up, logged after the step
    assert [round(m["lr-AdamW"] / 0.001 * 10) for m in recs] == [1, 2, 3, 4]
    from pacednegatives_amd.model import T5ForConditionalGenerationHIP
    m = T5ForConditionalGenerationHIP.from_pretrained(os.path.join(out, "finalmodel"))
    assert m.config.d_model == 128 and all(torch.isfinite(p).all() for p in m.parameters())
    assert os.path.isfile(os.path.join(out, "finalmodel", "config.json"))


def test_cli_main_in_process(tmp_path):
    """Same path called as a function: the saved directory reloads to the trained weights bit for bit, training moved
    them, and the pace ChangeDifficulty wrote into the dataset is the eta logged by the step before."""
    sys.path.insert(0, os.path.join(ROOT, "train"))
    import importlib
    cli = importlib.import_module("train_lce")
    from pacednegatives_amd.model import T5ForConditionalGenerationHIP
    out = str(tmp_path / "run")
    model, trainer = cli.main(data="", dataset_name="", out_dir=out, total_steps=400, eta=0.3466, batch_size=4, lr=1e-3, n=2,
                              warmup_steps=40, num_workers=0, max_steps=3, model_name="t5-tiny", synthetic=32, dropout=0.0, seed=3)
    recs = _metrics(out)
    assert len(recs) == 3 and trainer.global_step == 3
    ds = trainer.train_dataloader.paced_dataset
    assert ds.weight == min(1 - 1e-10, recs[1]["eta"])          # written before batch 3 from the eta after step 2
    assert abs(model.difficulty - recs[2]["eta"]) < 1e-12
    re = T5ForConditionalGenerationHIP.from_pretrained(os.path.join(out, "finalmodel"))
    fresh = T5ForConditionalGenerationHIP.from_pretrained("t5-tiny", seed=3)
    moved = 0.0
    for (k, a), (k2, b), (_k3, c) in zip(model.model.state_dict().items(), re.state_dict().items(), fresh.state_dict().items()):
        assert k == k2 and torch.equal(a, b), k
        moved = max(moved, (a - c).abs().max().item())
    assert moved > 1e-5


def _final_weights(out):
    from safetensors.torch import load_file
    return load_file(os.path.join(out, "finalmodel", "model.safetensors"))


def test_num_gpus_flag_starts_that_many_ranks(tmp_path):
    """``--num_gpus 2`` with no torchrun environment (how the reference's grid_var.py:28 calls the script; Lightning then
    starts one process per device, train_lce.py:89-94): the script launches two rank processes of itself.  They share
    this machine's one GPU, so the exchange goes over gloo (RCCL refuses two ranks on a device).  24 records in batches
    of 4 are 6 steps for one process and 3 for two ranks; the result equals a torch.distributed.run launch of the same
    command, and a world size that contradicts the flag is refused."""
    args = ["--data", "unused", "--dataset_name", "unused", "--batch_size", "4", "--lr", "0.001", "--eta", "30.0", "--n", "2",
            "--num_workers", "0", "--total_steps", "400", "--warmup_steps", "40", "--synthetic", "24", "--model_name", "t5-tiny",
            "--dropout", "0.0", "--seed", "2"]
    env = dict(os.environ, PACED_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", PACED_LAUNCH_TIMEOUT="240")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    out_self = str(tmp_path / "self")
    r = subprocess.run([sys.executable, CLI, "--out_dir", out_self, "--num_gpus", "2"] + args, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-3000:]
    recs = _metrics(out_self)
    assert [m["step"] for m in recs] == [1, 2, 3]
    out_run = str(tmp_path / "torchrun")
    # (torch.distributed.run parses the script's arguments too and takes "--n" for an abbreviation of its own options)
    args = ["--n_neg" if a == "--n" else a for a in args]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29617", CLI, "--out_dir", out_run, "--num_gpus", "2"] + args, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-3000:]
    recs2 = _metrics(out_run)
    assert len(recs2) == 3
    for a, b in zip(recs, recs2):
        for k in KEYS:
            assert abs(a[k] - b[k]) <= 1e-4 * max(1.0, abs(a[k])), (k, a[k], b[k])
    wa, wb = _final_weights(out_self), _final_weights(out_run)
    assert wa.keys() == wb.keys()
    for k in wa:           # fp32 atomics sum in another order from run to run: entries whose tiny gradient changes sign move by up to 2 lr per step
        d = (wa[k].float() - wb[k].float()).abs()
        assert d.max().item() <= 6.3e-3 and (d > 1e-4).float().mean().item() < 0.03, k
    # one process, --num_gpus 1: twice as many steps over the same records
    out_one = str(tmp_path / "one")
    r = subprocess.run([sys.executable, CLI, "--out_dir", out_one, "--num_gpus", "1"] + args, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0 and len(_metrics(out_one)) == 6
    # a torchrun world that contradicts the flag is an error, not a silent single-GPU run
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29618", CLI, "--out_dir", str(tmp_path / "bad"), "--num_gpus", "4"] + args, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode != 0 and b"--num_gpus 4" in r.stdout

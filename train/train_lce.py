#!/usr/bin/env python3
"""Drop-in counterpart of the reference's ``train/train_lce.py``: same flag names, defaults and
hyper-parameter arithmetic (``total_steps//batch_size``, ``warmup_steps//batch_size``,
``meta_lr = lr`` when unset, model 't5-base', ignore_index -100; reference ``train_lce.py:20-64``),
same sequence (model, data module, trainer with ChangeDifficulty, fit, save to out_dir/finalmodel;
``:66-103``).  Additions: ``--model_name`` (a size name or a local Hugging Face checkpoint directory;
names cannot be downloaded here) and ``--synthetic N`` (build an N-record synthetic corpus and
negatives file in out_dir instead of reading ``--data`` / ``--dataset_name``).

Multi-GPU: ``--num_gpus N`` alone is enough, as for the reference (whose Lightning trainer starts one process per
device, ``train_lce.py:89-94``): run without a torchrun environment, the script starts N rank processes of itself
before it touches the GPU, relays rank 0's output and exits with a non-zero code if a rank fails.  Under
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train/train_lce.py ... --num_gpus 8
it joins the ranks it is given; a WORLD_SIZE that differs from --num_gpus is an error.  (torch.distributed.run rejects
the reference's ``--n`` flag as an ambiguous abbreviation of its own options: spell it ``--n_neg`` there.)
"""
import argparse
import logging
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _bool(v):
    return str(v).lower() in ("1", "true", "yes", "y")


def main(data: str, dataset_name: str, out_dir: str, total_steps: int = 100000, eta: float = 0.0, batch_size: int = 16,
         lr: float = 0.001, meta_lr=None, var: float = 0.01, n: int = 2, use_max: bool = False, warmup_steps: int = 10000,
         sample: bool = False, use_mean: bool = False, num_gpus: int = 1, wandb_project: str = None, num_workers: int = 8,
         max_steps: int = None, model_name: str = "t5-base", synthetic: int = 0, dropout: float = None, seed: int = 0):
    from pacednegatives_amd.lceT5 import ChangeDifficulty, LCEDataModule, LCEModel
    from pacednegatives_amd.trainer import JSONLLogger, Trainer, init_distributed
    from pacednegatives_amd import synthetic as syn

    os.makedirs(out_dir, exist_ok=True)
    _rank, world, _local = init_distributed()
    if world != max(1, int(num_gpus)):
        raise RuntimeError(f"--num_gpus {num_gpus} but this run has {world} rank process(es): start the script without a "
                           "torchrun environment (it launches its own ranks) or with --nproc-per-node equal to --num_gpus")
    args = {
        "model_name": model_name,
        "total_steps": total_steps // batch_size,
        "eta": eta,
        "batch_size": batch_size,
        "lr": lr,
        "meta_lr": lr if not meta_lr else meta_lr,
        "n": n,
        "warmup_steps": warmup_steps // batch_size,
        "use_mean": use_mean,
        "ignore_index": -100,
        "dropout": dropout,
        "seed": seed,
    }
    model = LCEModel(args)
    corpus = dataset_name
    if synthetic:
        data, corpus = syn.write_synthetic_dataset(out_dir, n_records=synthetic, n_neg=max(64, 4 * n), seed=seed)
    data_module = LCEDataModule(data, corpus, model.tokenizer, batch_size, sample, use_max, var=var, n=n,
                                init_weight=model.weights.eta.item(), num_workers=num_workers)
    data_module.setup()
    logger = JSONLLogger(out_dir, project=wandb_project)
    trainer_args = {
        "devices": num_gpus,
        "callbacks": [ChangeDifficulty()],
        "logger": logger,
        "max_epochs": 1,
        "default_root_dir": out_dir,
        "profiler": "simple",
    }
    if max_steps:
        trainer_args["max_steps"] = max_steps
    if num_gpus > 1:
        trainer_args["accelerator"] = "gpu"
        trainer_args["strategy"] = "ddp"
    trainer = Trainer(**trainer_args)
    trainer.fit(model, data_module)
    if int(os.environ.get("RANK", "0")) == 0:
        model.model.save_pretrained(os.path.join(out_dir, "finalmodel"))
    return model, trainer


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO)
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--data", default="")
    ap.add_argument("--dataset_name", default="")
    ap.add_argument("--out_dir", required=True)
    ap.add_argument("--total_steps", type=int, default=100000)
    ap.add_argument("--eta", type=float, default=0.0)
    ap.add_argument("--batch_size", type=int, default=16)
    ap.add_argument("--lr", type=float, default=0.001)
    ap.add_argument("--meta_lr", type=float, default=None)
    ap.add_argument("--var", type=float, default=0.01)
    # (--n_neg: the same flag under a name torch.distributed.run's own parser accepts -- it reads every argument of the
    # command line, the script's included, and rejects "--n" as an ambiguous abbreviation of --nnodes / --nproc-per-node / ...)
    ap.add_argument("--n", "--n_neg", dest="n", type=int, default=2)
    ap.add_argument("--use_max", type=_bool, nargs="?", const=True, default=False)
    ap.add_argument("--warmup_steps", type=int, default=10000)
    ap.add_argument("--sample", type=_bool, nargs="?", const=True, default=False)
    ap.add_argument("--use_mean", type=_bool, nargs="?", const=True, default=False)
    ap.add_argument("--num_gpus", type=int, default=1)
    ap.add_argument("--wandb_project", default=None)
    ap.add_argument("--num_workers", type=int, default=8)
    ap.add_argument("--max_steps", type=int, default=None)
    ap.add_argument("--model_name", default="t5-base")
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--dropout", type=float, default=None)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    if not a.synthetic and not (a.data and a.dataset_name):
        ap.error("--data and --dataset_name are required unless --synthetic N is given")
    from pacednegatives_amd import launcher
    if a.num_gpus > 1 and not launcher.under_launcher():
        # one process per device, as Lightning's ddp strategy gives the reference; this process makes no GPU call
        raise SystemExit(launcher.launch_ranks(a.num_gpus))
    main(**vars(a))

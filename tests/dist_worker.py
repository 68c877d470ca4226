"""One rank of the multi-process GPU tests (started by tests/test_distributed_gpu.py, never collected by pytest).

  python tests/dist_worker.py OUT_DIR        with RANK / WORLD_SIZE / MASTER_* in the environment

Runs two lceT5 training steps on this rank's contiguous share of one fixed batch and a sharded negative-pool
selection; writes what the parent compares to OUT_DIR/rank{r}.npz.  Several ranks may share one GPU (backend gloo).
"""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

B, NNEG, S = 4, 3, 40          # global batch


def main(out_dir):
    from pacednegatives_amd import mining, parallel
    from pacednegatives_amd.config import NAMED
    from pacednegatives_amd.lceT5 import LCEModel
    from pacednegatives_amd.synthetic import token_batch
    from pacednegatives_amd.trainer import init_distributed
    rank, world, local = init_distributed()
    torch.cuda.set_device(local)
    dims = NAMED["t5-tiny"]
    hp = {"model_name": "t5-tiny", "total_steps": 10, "eta": 35.0, "batch_size": B // world, "lr": 1e-3, "meta_lr": 1e-2,
          "n": NNEG, "warmup_steps": 0, "use_mean": False, "ignore_index": -100, "dropout": 0.0, "seed": 0}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = LCEModel(hp)
    grads = []
    # PACED_TEST_EARLY=1: no look at the summed gradients, so that from the second step on each segment is exchanged
    # and updated as soon as its gradients are final, under the rest of the backward pass
    early = os.environ.get("PACED_TEST_EARLY") == "1"
    if not early:
        model.on_gradients_ready = lambda m: grads.append((m.model.params.g32 / world).cpu().numpy().copy())
    out = {}
    b = B // world
    for it in range(3 if early else 2):
        pos, neg = token_batch(B, NNEG, S, dims.vocab_size, seed=50 + it)
        sl, nsl = slice(rank * b, (rank + 1) * b), slice(rank * b * NNEG, (rank + 1) * b * NNEG)
        batch = ({k: v[sl] for k, v in pos.items()}, {k: v[nsl] for k, v in neg.items()})
        res = model.training_step(batch, it)
        out[f"rows{it}"] = res["loss_rows"].cpu().numpy()
        out[f"main_rows{it}"] = res["main_loss_rows"].cpu().numpy()
        out[f"weights{it}"] = res["weights"].cpu().numpy()
        out[f"eta{it}"] = np.array([model.difficulty])
        if not early:
            out[f"grad{it}"] = grads[-1]
    assert model.model.params.g32.abs().max().item() == 0, "gradients not cleared after the update"
    out["p32"] = model.model.params.p32.cpu().numpy()

    # sharded pool selection: integer-valued embeddings (exact scores, many ties), rows split over the ranks
    g = torch.Generator().manual_seed(7)
    Np, D, Q, k = 6000, 128, 24, 100
    pool = torch.randint(-8, 9, (Np, D), generator=g).to(torch.bfloat16).cuda()
    qry = torch.randint(-8, 9, (Q, D), generator=g).to(torch.bfloat16).cuda()
    per = (Np + world - 1) // world
    lo, hi = rank * per, min(Np, (rank + 1) * per)
    s, i = mining.pool_topk_sharded(qry, pool[lo:hi].contiguous(), k, lo)
    s1, i1 = mining.pool_topk(qry, pool, k)
    out["topk_s"], out["topk_i"] = s.cpu().numpy(), i.cpu().numpy()
    out["full_s"], out["full_i"] = s1.cpu().numpy(), i1.cpu().numpy()
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])

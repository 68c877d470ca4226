#!/usr/bin/env python3
"""Generate tests/golden/schedules.npz from the reference itself (build container only: needs /root/reference).

Executed from the reference: ``EtaWeights.no_grad``, ``LCEWeights.no_grad``, ``Weights.set_weight/__getitem__``
(pacednegatives/weights.py), ``interpolate_scalar`` (old/interp.py) and ``LevelWrapper.check_success`` (old/levels.py),
the old/ modules imported with in-process stand-ins for the absent packages ``ir_datasets`` and ``wandb`` and the
removed ``transformers.AdamW`` name (none takes part in the recorded arithmetic)."""
import importlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, "/root/reference/old")
OUT = os.path.join(ROOT, "tests", "golden", "schedules.npz")


def main():
    from pacednegatives.weights import EtaWeights, LCEWeights, Weights
    cpu = torch.device("cpu")
    out = {}
    g = torch.Generator().manual_seed(11)
    losses = torch.rand(24, generator=g) * 1.4
    losses[3] = 0.4                                   # a loss exactly on a threshold
    out["losses"] = losses.numpy()
    etas, masks = [0.1, 0.4, 1.0], [0.0, 0.5]
    out["etas"], out["masks"] = np.array(etas), np.array(masks)
    for form, cls in ((1, EtaWeights), (0, LCEWeights)):
        for ei, eta in enumerate(etas):
            for mi, mask in enumerate(masks):
                w = cls(eta, device=cpu)
                w.set_mask(mask)
                out[f"w_form{form}_eta{ei}_mask{mi}"] = w.no_grad(losses, torch.tensor([eta])).numpy()
    W = Weights((3, 5), device=cpu)
    val = torch.tensor([0.0, 1.5, -2.0, 0.25, 8.0])
    W.set_weight(1, val)
    out["sample_val"], out["sample_weight_row1"] = val.numpy(), W[1].detach().numpy()
    out["sample_weight_row0"] = W[0].detach().numpy()

    for name in ("ir_datasets", "wandb"):
        sys.modules.setdefault(name, types.ModuleType(name))
    import transformers
    if not hasattr(transformers, "AdamW"):
        transformers.AdamW = torch.optim.AdamW
    interp = importlib.import_module("interp")
    f = interp.interpolate_scalar(0.1, 0.9, 1000)
    steps = np.array([0, 1, 32, 500, 999, 1000, 1001, 50000])
    out["interp_steps"], out["interp_values"] = steps, np.array([f(int(s)) for s in steps])
    levels = importlib.import_module("levels")
    V, REL, NREL = 6400, 1176, 6136
    pos = torch.randn(12, 1, V, generator=g)
    neg = torch.randn(12, 1, V, generator=g)
    neg[5, 0, REL], neg[5, 0, NREL] = pos[5, 0, REL], pos[5, 0, NREL]     # a tie: not a success
    ns = types.SimpleNamespace(REL=REL, NREL=NREL, success_rate=[])
    levels.LevelWrapper.check_success(ns, pos, neg)
    out["success_pos"], out["success_neg"] = pos[:, 0, (REL, NREL)].numpy(), neg[:, 0, (REL, NREL)].numpy()
    out["success_rate"] = np.array(ns.success_rate)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()

"""Small end-to-end check used by ``__graft_entry__.smoke()``: one lceT5 training step of a tiny T5 on
cuda:0 through the HIP path, compared with the CPU oracle on the same parameters and token batch."""
from __future__ import annotations

import torch


def smoke_step(verbose: bool = True):
    # the oracle is test infrastructure: imported here only as the checker
    from oracle import lce_ref
    from oracle.t5_ref import T5Dims as RefDims, init_params

    from .config import NAMED
    from .lceT5 import LCEModel
    from .synthetic import token_batch

    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs a GPU")
    torch.cuda.set_device(0)
    dims = NAMED["t5-tiny"]
    B, n, S = 2, 2, 16
    hp = {"model_name": "t5-tiny", "total_steps": 10, "eta": 30.0, "batch_size": B, "lr": 1e-3, "meta_lr": 1e-3, "n": n,
          "warmup_steps": 0, "use_mean": False, "ignore_index": -100, "dropout": 0.0}
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = LCEModel(hp)
    rd = RefDims(dims.vocab_size, dims.d_model, dims.d_kv, dims.num_heads, dims.d_ff, dims.num_layers, dims.num_decoder_layers)
    P = init_params(rd, 0)
    model.model.params.load_hf(P)
    batch = token_batch(B, n, S, dims.vocab_size, seed=7)
    out = model.training_step(batch, 0)
    torch.cuda.synchronize()

    eta = torch.tensor([30.0])
    opt = lce_ref.HFAdamW(P, 1e-3)
    meta = lce_ref.HFAdamW({"eta": eta}, 1e-3)
    ref = lce_ref.training_step(P, rd, eta, opt, meta, 0, batch[0]["input_ids"], batch[0]["labels"], batch[1]["input_ids"],
                                batch[1]["labels"], n, False, 0, 10, emulate_bf16=True)
    got = out["loss_rows"].cpu()
    err = (got - ref["loss"]).abs().max().item() / ref["loss"].abs().max().item()
    eta_err = abs(model.difficulty - float(eta))
    if verbose:
        print(f"[smoke] LCE rows HIP {got.tolist()} oracle {ref['loss'].tolist()} rel err {err:.2e}; eta HIP {model.difficulty:.6f} oracle {float(eta):.6f}")
    # bf16 rounding at the same points on both sides, different accumulation order: measured 4.4e-4 of the loss scale
    assert err < 2e-3, f"LCE loss rows differ from the oracle by {err}"
    assert eta_err < 1e-4, f"eta after the meta step differs by {eta_err}"
    w = model.model.params.master("enc.0.wi").cpu()
    w_ref = P["encoder.block.0.layer.1.DenseReluDense.wi.weight"]
    assert (w - w_ref).abs().max().item() < 2.5e-3, "parameter update differs from the oracle"
    return err

#!/usr/bin/env python3
"""Which bf16 storage point carries how much of the logits / loss error of the HIP path's arithmetic.

CPU only.  The fp32 oracle (oracle/t5_ref.py) is the yardstick; its bf16-emulating mode rounds at the storage points
the HIP path rounds at (``ROUNDING_POINTS``).  Each row of the table un-rounds some of them and reports, against the fp32
forward of the same parameters and inputs: the largest logit error relative to the largest |logit|, the largest error
of the per-token cross-entropy at the label positions (relative to the largest loss), and of the LCE rows.
usage: rounding_points.py [model=t5-base] [seq_len=256] [n_neg=3] [out.txt]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import lce_ref  # noqa: E402
from oracle.t5_ref import ROUNDING_POINTS, T5Dims, init_params, t5_forward  # noqa: E402
from pacednegatives_amd.config import NAMED  # noqa: E402
from pacednegatives_amd.synthetic import token_batch  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "t5-base"
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    out_path = sys.argv[4] if len(sys.argv) > 4 else None
    dm = NAMED[name]
    rd = T5Dims(dm.vocab_size, dm.d_model, dm.d_kv, dm.num_heads, dm.d_ff, dm.num_layers, dm.num_decoder_layers)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    P = init_params(rd, 0)
    pos, neg = token_batch(1, n, S, dm.vocab_size, seed=1234)
    ids = torch.cat([pos["input_ids"], neg["input_ids"]], 0)
    labels = torch.cat([pos["labels"], neg["labels"]], 0)
    V = dm.vocab_size

    def measures(logits):
        ce = torch.nn.functional.cross_entropy(logits.reshape(-1, V), labels.view(-1), reduction="none")
        rows = lce_ref.pair_loss(logits[:1], logits[1:].reshape(-1, n, V), labels[:1], labels[1:], n, False)
        return logits, ce, rows

    with torch.no_grad():
        ref = measures(t5_forward(P, rd, ids, labels))
        dec = tuple(p for p in ROUNDING_POINTS if p.startswith("dec_")) + ("head_in",)
        enc = tuple(p for p in ROUNDING_POINTS if p.startswith("enc_"))
        variants = [("every point rounded (the HIP path)", ())]
        variants += [(f"fp32 at {p}", (p,)) for p in ROUNDING_POINTS]
        variants += [("fp32 decoder rows + LM-head input (1 + n sequences x 2 rows: free)", dec),
                     ("fp32 everywhere but the weights", enc + dec),
                     ("fp32 everywhere but the encoder stream", tuple(p for p in ROUNDING_POINTS if p != "enc_stream")),
                     ("fp32 weights and decoder, bf16 encoder activations", ("params",) + dec)]
        lines = [f"# {name} random-init (seed 0), 1 positive + {n} negatives x {S} tokens, fp32 oracle as yardstick",
                 f"# largest |logit| {ref[0].abs().max():.3f}, largest per-token loss {ref[1].max():.3f}, LCE rows {ref[2].tolist()}",
                 f"{'variant':72s} {'logits/max|logit|':>18s} {'token loss (rel.)':>18s} {'LCE rows (rel.)':>16s}"]
        print("\n".join(lines), flush=True)
        for label, keep in variants:
            t0 = time.time()
            got = measures(t5_forward(P, rd, ids, labels, emulate_bf16=True, keep_fp32=keep))
            e_log = ((got[0] - ref[0]).abs().max() / ref[0].abs().max()).item()
            e_ce = ((got[1] - ref[1]).abs().max() / ref[1].abs().max()).item()
            e_rows = ((got[2] - ref[2]).abs().max() / ref[2].abs().max()).item()
            lines.append(f"{label:72s} {e_log:18.2e} {e_ce:18.2e} {e_rows:16.2e}")
            print(lines[-1], f"   ({time.time() - t0:.0f} s)", flush=True)
    if out_path:
        with open(out_path, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()

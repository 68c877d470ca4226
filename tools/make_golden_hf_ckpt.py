#!/usr/bin/env python3
"""Writes tests/golden/hf_ckpt/ : a checkpoint directory produced by Hugging Face's own
``T5ForConditionalGeneration.save_pretrained`` (seeded random init of a very small T5) and, next to it,
``tests/golden/hf_ckpt_logits.npz`` with that model's fp32 logits on a fixed batch.  Run in the build container
(transformers is part of the image); the GPU tests load the directory through this package's ``from_pretrained``.
"""
import os
import sys

import numpy as np
import torch
from transformers import T5Config, T5ForConditionalGeneration

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "hf_ckpt")


def main():
    torch.manual_seed(11)
    cfg = T5Config(vocab_size=6144, d_model=64, d_kv=64, d_ff=128, num_layers=2, num_decoder_layers=2, num_heads=2,
                   dropout_rate=0.1, decoder_start_token_id=0, pad_token_id=0, eos_token_id=1, feed_forward_proj="relu",
                   tie_word_embeddings=True)
    model = T5ForConditionalGeneration(cfg).eval()
    # default init leaves the tied embedding at N(0, 1): logits of order d_model^0.5 * |h|; keep it, it is what HF does
    model.save_pretrained(OUT, safe_serialization=True)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(2, 6144, (4, 21), generator=g)
    labels = torch.tensor([[1176, 1], [6136, 1], [6136, 1], [1176, 1]])
    with torch.no_grad():
        logits = model(input_ids=ids, labels=labels).logits
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "hf_ckpt_logits.npz"), ids=ids.numpy(), labels=labels.numpy(),
                        logits=logits.numpy())
    print("wrote", OUT, sorted(os.listdir(OUT)), "logits", tuple(logits.shape), float(logits.abs().max()))


if __name__ == "__main__":
    sys.exit(main())

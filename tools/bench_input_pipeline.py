#!/usr/bin/env python3
"""Sequences per second of the host input pipeline at the benchmark batch shape (32 tuples x (1 + 16) sequences,
easy->hard lists of 1000), one process, no GPU:

  reference-style : per item scipy binom.pmf + np.random.choice (pacednegatives/dataloader.py:134-151), string
                    formatting and tokenisation of every text in the collate function (lceT5.py:38-54)
  mirror          : this package's LCEDataset + collate (pace probabilities cached per batch)
  fast            : TokenStore + one Gumbel top-n per batch (pacednegatives_amd/fastdata.py)

The tokenizer is the package's stand-in (regex word hashing); SentencePiece costs more per text, which widens the
gap.  The GPU consumes 32 x 17 sequences per step: the last column is batches per second."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import sampler_ref
from pacednegatives_amd import synthetic
from pacednegatives_amd.fastdata import TokenStore
from pacednegatives_amd.lceT5 import LCEDataModule
from pacednegatives_amd.tokenizer import SyntheticTokenizer

B, N, NNEG, BATCHES = 32, 16, 1000, 12


def main():
    tmp = tempfile.mkdtemp()
    path, corpus = synthetic.write_synthetic_dataset(tmp, n_records=B * BATCHES, n_neg=NNEG, seed=0)
    tok = SyntheticTokenizer()
    t0 = time.perf_counter()
    TokenStore.build(corpus, tok, os.path.join(tmp, "tokens"))
    build_s = time.perf_counter() - t0

    def run(name, loader):
        t0 = time.perf_counter()
        n = 0
        for pos, neg in loader:
            n += pos["input_ids"].shape[0] + neg["input_ids"].shape[0]
        dt = time.perf_counter() - t0
        print(f"{name:16s} {n / dt:10.0f} sequences/s   {n / dt / (B * (1 + N)):7.2f} batches/s", flush=True)

    slow = LCEDataModule(path, corpus, tok, batch_size=B, n=N, var=0.01, init_weight=0.3466, num_workers=0)
    slow.setup()
    ds = slow.dataset
    collate = slow.collate()

    def reference_style():
        for lo in range(0, len(ds), B):
            batch = []
            for i in range(lo, min(lo + B, len(ds))):
                picks = sampler_ref.sample(ds.weight, ds.n_neg, ds.n, ds.var)
                (q, p), negs = ds.get((i, picks))
                batch.append({"pos": ds.format(q, p), "neg": [ds.format(q, d) for d in negs]})
            yield collate(batch)
    run("reference-style", reference_style())
    run("mirror", slow.train_dataloader())
    fast = LCEDataModule(path, corpus, tok, batch_size=B, n=N, var=0.01, init_weight=0.3466, num_workers=0,
                         token_store=os.path.join(tmp, "tokens"), propagate_pace=True)
    fast.setup()
    loader = fast.train_dataloader()
    loader.prefetch = 0
    run("fast", loader)
    print(f"(token store of {len(list(corpus.docs_iter()))} documents built in {build_s:.1f} s)")


if __name__ == "__main__":
    main()

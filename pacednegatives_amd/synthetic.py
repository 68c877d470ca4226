"""Synthetic MS-MARCO-shaped inputs: there is no network for datasets, so benchmarks and the CPU
plumbing config use generated text / token ids of the same shapes as the reference's inputs."""
from __future__ import annotations

import json
import os
import random

import torch

from .tokenizer import EOS_ID, FALSE_ID, TRUE_ID


class ListCorpus:
    """Duck-type of an ir_datasets corpus: ``docs_iter()`` / ``queries_iter()`` over dict records."""

    def __init__(self, docs, queries):
        self._docs, self._queries = docs, queries

    def docs_iter(self):
        return iter(self._docs)

    def queries_iter(self):
        return iter(self._queries)


_WORDS = ("the of and to in is for on with as by at from that this it are was be or an which can has have not but their "
          "more one all also into other new some time when system data model search query passage document answer "
          "relevant rank score index term weight neural language learn train large small fast slow high low").split()


def _sentence(rng, lo, hi):
    return " ".join(rng.choice(_WORDS) for _ in range(rng.randint(lo, hi)))


def write_synthetic_dataset(out_dir: str, n_records: int, n_neg: int, seed: int = 0):
    """Writes ``synthetic.json`` in the reference's record format ({query_id, doc_id_a, doc_id_b: easy->hard
    list}; ``pacednegatives/utilities/group_results.py:35-38``) and returns (path, corpus)."""
    rng = random.Random(seed)
    n_docs = n_neg * 4 + n_records
    docs = [{"doc_id": f"d{i}", "text": _sentence(rng, 20, 60)} for i in range(n_docs)]
    queries = [{"query_id": f"q{i}", "text": _sentence(rng, 4, 9)} for i in range(n_records)]
    recs = []
    for i in range(n_records):
        negs = rng.sample(range(n_neg * 4), n_neg)
        recs.append({"query_id": f"q{i}", "doc_id_a": f"d{n_neg * 4 + i}", "doc_id_b": [f"d{j}" for j in negs]})
    path = os.path.join(out_dir, "synthetic.json")
    # written under a private name and moved into place: several rank processes may write the (identical) file at once,
    # and none must ever read a half-written one
    tmp = f"{path}.{os.getpid()}.tmp"
    with open(tmp, "w") as f:
        json.dump(recs, f)
    os.replace(tmp, path)
    return path, ListCorpus(docs, queries)


def token_batch(batch: int, n_neg: int, seq_len: int, vocab_size: int = 32128, seed: int = 1234, device="cpu"):
    """Token-id batch in the collate layout (reference ``lceT5.py:53``): ids uniform in [2, 32000),
    last token </s>; labels [[true, </s>]] for positives and [[false, </s>]] for negatives."""
    g = torch.Generator().manual_seed(seed)
    hi = min(32000, vocab_size)
    pos = torch.randint(2, hi, (batch, seq_len), generator=g)
    neg = torch.randint(2, hi, (batch * n_neg, seq_len), generator=g)
    pos[:, -1] = EOS_ID
    neg[:, -1] = EOS_ID
    plab = torch.tensor([[TRUE_ID, EOS_ID]] * batch)
    nlab = torch.tensor([[FALSE_ID, EOS_ID]] * (batch * n_neg))
    mk = lambda x: x.to(device)   # noqa: E731
    return ({"input_ids": mk(pos), "labels": mk(plab)}, {"input_ids": mk(neg), "labels": mk(nlab)})

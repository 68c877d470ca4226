"""Host input pipeline that keeps up with the GPU step (same batches, in distribution, as the reference path).

The reference builds every batch in DataLoader workers: per item one ``scipy.stats.binom.pmf`` over the whole
easy->hard list and one ``np.random.choice(..., replace=False, p=p)`` (``pacednegatives/dataloader.py:134-151``),
string formatting of 1 + n texts (``:153-162``) and SentencePiece over all of them in the collate function
(``pacednegatives/lceT5.py:38-54``).  At 32 tuples x 17 sequences per 0.14 s step and GPU that is ~4000 sequences
a second per GPU.  This module produces the same batches from

* ``gumbel_top_n``: all items of a batch share one pace, hence one probability vector; n draws without replacement
  for the whole batch are one Gumbel-perturbed top-n (``np.argpartition``) instead of B calls of ``choice``.
  numpy's ``choice(replace=False, p=p)`` keeps the first occurrences of an i.i.d. stream from p, i.e. successive
  sampling (Plackett-Luce); so does the Gumbel top-n, ordered by perturbed key.  The two differ in the random
  stream they consume, not in the distribution of the index tuples.
* ``TokenStore``: the corpus tokenised once into flat int32 arrays opened with ``np.memmap``; a sequence is the
  concatenation ``[Query:] q [Document:] d [Relevant:] </s>`` of stored pieces.  For a tokenizer that never merges
  across whitespace (SentencePiece's default ``split_by_whitespace``; the stand-in tokenizer of this package) this
  equals tokenising the formatted string.
* ``PacedBatchLoader``: batches built in the training process (no worker copies of the dataset, so the pace written
  by ``ChangeDifficulty`` is seen at once -- in the reference, worker copies never see it, ``lceT5.py:77-80``), with a
  one-batch-ahead background thread.
"""
from __future__ import annotations

import json
import os
import queue
import threading
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

from . import parallel
from .tokenizer import EOS_ID, FALSE_ID, PAD_ID, TRUE_ID


def gumbel_top_n(p: np.ndarray, n: int, size: int, rng: np.random.Generator) -> np.ndarray:
    """``size`` independent draws of ``n`` distinct indices, each distributed as
    ``np.random.choice(len(p), size=n, replace=False, p=p)`` (order included).  int64 [size, n].

    Raises ValueError when fewer than ``n`` entries of p are non-zero, the condition under which numpy's
    ``choice`` raises and the reference prints "Failed to sample" and exits (``dataloader.py:146-150``)."""
    p = np.asarray(p, dtype=np.float64)
    support = np.flatnonzero(p > 0)
    if support.size < n:
        raise ValueError(f"Fewer non-zero entries in p ({support.size}) than size ({n})")
    # perturb only the support: entries of probability zero can never be drawn
    logp = np.log(p[support])
    u = rng.random((size, support.size))
    keys = logp - np.log(-np.log(u))                      # log p + Gumbel(0, 1)
    if n < support.size:
        part = np.argpartition(-keys, n - 1, axis=1)[:, :n]
    else:
        part = np.broadcast_to(np.arange(support.size), (size, support.size)).copy()
    order = np.argsort(-np.take_along_axis(keys, part, axis=1), axis=1)
    return support[np.take_along_axis(part, order, axis=1)].astype(np.int64)


# ------------------------------------------------------------------------------------------ token store
def _field(rec, name):
    return rec[name] if isinstance(rec, dict) else getattr(rec, name)


def _encode_piece(tokenizer, text: str) -> List[int]:
    """Token ids of ``text`` without the end-of-sequence id."""
    ids = tokenizer([text], padding=True, return_tensors="pt")["input_ids"][0].tolist()
    while ids and ids[-1] == PAD_ID:
        ids.pop()
    if ids and ids[-1] == EOS_ID:
        ids.pop()
    return ids


class _Table:
    """id -> row of a ragged int32 token array stored as (flat tokens, offsets)."""

    def __init__(self, tokens: np.ndarray, offsets: np.ndarray, ids: Sequence[str]):
        self.tokens, self.offsets = tokens, offsets
        self.row = {k: i for i, k in enumerate(ids)}

    def get(self, key: str) -> np.ndarray:
        r = self.row[key]
        return self.tokens[self.offsets[r]:self.offsets[r + 1]]

    def __len__(self):
        return len(self.row)


class TokenStore:
    """Pre-tokenised corpus on disk.  ``build`` once, ``open`` (memory-mapped, read-only) in every process."""

    FILES = ("doc_tokens.i32", "doc_offsets.i64", "doc_ids.json", "query_tokens.i32", "query_offsets.i64",
             "query_ids.json", "template.json")

    def __init__(self, docs: _Table, queries: _Table, template: Dict[str, List[int]]):
        self.docs, self.queries = docs, queries
        self.t_query = np.asarray(template["query"], dtype=np.int32)
        self.t_document = np.asarray(template["document"], dtype=np.int32)
        self.t_relevant = np.asarray(template["relevant"], dtype=np.int32)

    @staticmethod
    def _write(directory: str, name: str, records: Iterable, id_field: str, tokenizer, batch: int = 1024):
        ids, offsets, total = [], [0], 0
        with open(os.path.join(directory, f"{name}_tokens.i32"), "wb") as f:
            buf_ids, buf_txt = [], []

            def flush():
                nonlocal total
                if not buf_txt:
                    return
                enc = tokenizer(buf_txt, padding=True, return_tensors="pt")["input_ids"].numpy()
                for row in enc:
                    row = row[row != PAD_ID]
                    if row.size and row[-1] == EOS_ID:
                        row = row[:-1]
                    f.write(row.astype(np.int32).tobytes())
                    total += row.size
                    offsets.append(total)
                ids.extend(buf_ids)
                buf_ids.clear()
                buf_txt.clear()
            for rec in records:
                buf_ids.append(str(_field(rec, id_field)))
                buf_txt.append(_field(rec, "text"))
                if len(buf_txt) == batch:
                    flush()
            flush()
        np.asarray(offsets, dtype=np.int64).tofile(os.path.join(directory, f"{name}_offsets.i64"))
        with open(os.path.join(directory, f"{name}_ids.json"), "w") as f:
            json.dump(ids, f)

    @classmethod
    def build(cls, corpus, tokenizer, directory: str) -> "TokenStore":
        """Tokenise ``corpus.docs_iter()`` / ``corpus.queries_iter()`` (the duck type ``dataloader.py:113-114`` reads)
        and the three fixed pieces of ``"Query: " + q + " Document: " + d + " Relevant:"`` (``dataloader.py:153-155``)."""
        os.makedirs(directory, exist_ok=True)
        cls._write(directory, "doc", corpus.docs_iter(), "doc_id", tokenizer)
        cls._write(directory, "query", corpus.queries_iter(), "query_id", tokenizer)
        template = {"query": _encode_piece(tokenizer, "Query:"), "document": _encode_piece(tokenizer, "Document:"),
                    "relevant": _encode_piece(tokenizer, "Relevant:")}
        with open(os.path.join(directory, "template.json"), "w") as f:
            json.dump(template, f)
        return cls.open(directory)

    @classmethod
    def open(cls, directory: str) -> "TokenStore":
        missing = [n for n in cls.FILES if not os.path.isfile(os.path.join(directory, n))]
        if missing:
            raise FileNotFoundError(f"{directory} is not a token store (missing {missing}); run TokenStore.build first")

        def table(name):
            offsets = np.fromfile(os.path.join(directory, f"{name}_offsets.i64"), dtype=np.int64)
            path = os.path.join(directory, f"{name}_tokens.i32")
            tokens = np.memmap(path, dtype=np.int32, mode="r") if os.path.getsize(path) else np.zeros(0, dtype=np.int32)
            with open(os.path.join(directory, f"{name}_ids.json")) as f:
                ids = json.load(f)
            if offsets.size != len(ids) + 1 or (tokens.size != offsets[-1]):
                raise ValueError(f"{directory}: {name} table is inconsistent")
            return _Table(tokens, offsets, ids)
        with open(os.path.join(directory, "template.json")) as f:
            template = json.load(f)
        return cls(table("doc"), table("query"), template)

    def sequence(self, query_id: str, doc_id: str) -> np.ndarray:
        """Token ids of ``format(q, d)`` followed by </s>."""
        return np.concatenate([self.t_query, self.queries.get(query_id), self.t_document, self.docs.get(doc_id),
                               self.t_relevant, np.asarray([EOS_ID], dtype=np.int32)])


def _pad(rows: List[np.ndarray]) -> torch.Tensor:
    width = max(r.size for r in rows)
    out = np.full((len(rows), width), PAD_ID, dtype=np.int64)
    for i, r in enumerate(rows):
        out[i, :r.size] = r
    return torch.from_numpy(out)


# ---------------------------------------------------------------------------------------- batch loader
class PacedBatchLoader:
    """Iterates the batches ``DataLoader(LCEDataset, batch_size, collate_fn=LCEDataModule.collate())`` would give --
    ``({'input_ids': [B, Sp], 'labels': [B, 2]}, {'input_ids': [B n, Sn], 'labels': [B n, 2]})`` with positives and
    negatives each padded to their own longest sequence and labels "true </s>" / "false </s>" (``lceT5.py:38-54``) --
    from a ``TokenStore`` and one ``gumbel_top_n`` call per batch.

    ``dataset`` is the ``LCEDataset`` (its ``pace_probabilities`` is the reference's arithmetic, cached per pace; its
    ``weight`` is the pace ``ChangeDifficulty`` writes).  Items are visited in order; rank r of W visits r, r+W, ...
    The last batch may be short, as with ``drop_last=False``.
    """

    def __init__(self, dataset, pairs: Sequence[Sequence[str]], neg_ids: Sequence[Sequence[str]], store: TokenStore,
                 batch_size: int, rank: int = 0, world_size: int = 1, seed: int = 0, prefetch: int = 1,
                 pin_memory: Optional[bool] = None, propagate_pace: bool = True):
        if len(pairs) != len(neg_ids):
            raise ValueError("pairs and neg_ids differ in length")
        self.paced_dataset = self.dataset = dataset
        self.pairs, self.store, self.batch_size = pairs, store, int(batch_size)
        # negatives as rows of the doc table: [items, n_neg] int64 (one lookup per id here, none per batch)
        self.neg_rows = np.asarray([[store.docs.row[d] for d in row] for row in neg_ids], dtype=np.int64)
        self.items = list(parallel.shard_indices(len(pairs), rank, world_size)) if world_size > 1 else list(range(len(pairs)))
        self.rng = np.random.default_rng([seed, rank])
        self.prefetch = int(prefetch)
        self.pin_memory = torch.cuda.is_available() if pin_memory is None else bool(pin_memory)
        # False reproduces what the reference's worker processes do: sample at the pace the dataset had when the
        # iterator was created, whatever ChangeDifficulty writes afterwards
        self.propagate_pace = bool(propagate_pace)
        self._frozen_pace = None

    def __len__(self):
        return (len(self.items) + self.batch_size - 1) // self.batch_size

    def build_batch(self, items: Sequence[int]):
        ds, st = self.dataset, self.store
        pace = ds.weight if self.propagate_pace or self._frozen_pace is None else self._frozen_pace
        p = ds.pace_probabilities(pace)
        try:
            picks = gumbel_top_n(p, ds.n, len(items), self.rng)
        except ValueError as e:
            raise ValueError(f"Failed to sample with mean {float(np.clip(pace, ds.min, ds.max))} and var {ds.var}") from e
        tok, off = st.docs.tokens, st.docs.offsets
        pos, neg = [], []
        for b, item in enumerate(items):
            qid, did = self.pairs[item]
            q = st.queries.get(qid)
            head = np.concatenate([st.t_query, q, st.t_document])
            tail = np.concatenate([st.t_relevant, np.asarray([EOS_ID], dtype=np.int32)])
            pos.append(np.concatenate([head, st.docs.get(did), tail]))
            for r in self.neg_rows[item][picks[b]]:
                neg.append(np.concatenate([head, tok[off[r]:off[r + 1]], tail]))
        pos_ids, neg_ids = _pad(pos), _pad(neg)
        pos_l = torch.tensor([[TRUE_ID, EOS_ID]] * len(pos), dtype=torch.int64)
        neg_l = torch.tensor([[FALSE_ID, EOS_ID]] * len(neg), dtype=torch.int64)
        if self.pin_memory:
            pos_ids, neg_ids, pos_l, neg_l = (t.pin_memory() for t in (pos_ids, neg_ids, pos_l, neg_l))
        return ({"input_ids": pos_ids, "labels": pos_l}, {"input_ids": neg_ids, "labels": neg_l})

    def _chunks(self):
        for lo in range(0, len(self.items), self.batch_size):
            yield self.items[lo:lo + self.batch_size]

    def __iter__(self):
        self._frozen_pace = self.dataset.weight
        if self.prefetch <= 0:
            for chunk in self._chunks():
                yield self.build_batch(chunk)
            return
        # One batch ahead on a thread: numpy releases the GIL in the heavy calls and the training process
        # spends its time waiting on the GPU.  The batch in flight was sampled at the pace of the step before.
        q: "queue.Queue" = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()

        def work():
            try:
                for chunk in self._chunks():
                    if stop.is_set():
                        return
                    q.put(("batch", self.build_batch(chunk)))
                q.put(("done", None))
            except BaseException as e:       # surfaced in the consumer
                q.put(("error", e))
        t = threading.Thread(target=work, daemon=True)
        t.start()
        try:
            while True:
                kind, val = q.get()
                if kind == "done":
                    return
                if kind == "error":
                    raise val
                yield val
        finally:
            stop.set()
            while not q.empty():
                q.get_nowait()

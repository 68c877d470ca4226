"""Host-side datasets of the paced-negatives path (mirror of ``pacednegatives/dataloader.py``).

``LCEDataset`` is the live one (reference ``dataloader.py:97-162``): item ``idx`` is the (query,
positive) pair ``idx`` plus ``n`` negatives drawn without replacement from that query's easy->hard
list with binomial pace probabilities centred on ``weight``.  The pair/level/triplet loaders feed only
legacy entry points of the reference and are kept for API parity.
"""
from __future__ import annotations

from functools import partial
from math import ceil, floor

import numpy as np
import torch
from scipy.stats import binom
from torch.autograd import Variable
from torch.utils.data import Dataset

gen_var = lambda x, y: Variable(x, requires_grad=y)   # noqa: E731  (reference dataloader.py:12)

OUTPUTS = ["true", "false"]


def _text_tables(corpus):
    """doc_id -> text and query_id -> text from an ir_datasets-like corpus (``docs_iter`` /
    ``queries_iter`` yielding records with doc_id/query_id and text; mappings or attribute records)."""
    def field(rec, name):
        return rec[name] if isinstance(rec, dict) else getattr(rec, name)
    docs = {field(r, "doc_id"): field(r, "text") for r in corpus.docs_iter()}
    queries = {field(r, "query_id"): field(r, "text") for r in corpus.queries_iter()}
    return docs, queries


def format_pair(q: str, d: str) -> str:
    return "Query: " + q + " Document: " + d + " Relevant:"


class TripletDataset:
    """Reference ``dataloader.py:16-33``: one negative picked at index round(weight * n_neg)."""

    def __init__(self, pairs, neg_idx, corpus, max=False):
        self.neg_idx = neg_idx
        self.n_neg = len(neg_idx[0]) - 1
        self.docs, self.queries = _text_tables(corpus)
        self.round = ceil if max else floor
        self.data = [(self.queries[q], self.docs[p]) for q, p in pairs]

    def __len__(self):
        return len(self.data)

    def get_items(self, idx, weight=None):
        assert weight is not None, "Weights not set"
        q, p = self.data[idx]
        n = self.neg_idx[idx][self.round(weight * self.n_neg)]
        return q, p, self.docs[n]


class _BatchLoader:
    def __init__(self, dataset, batch_size) -> None:
        self.dataset = dataset
        self.batch_size = batch_size

    def __len__(self):
        return len(self.dataset)

    def format(self, q, d):
        return format_pair(q, d)


class PairLoader(_BatchLoader):
    """Reference ``dataloader.py:35-56``: per-item weights choose each item's negative."""

    def get_batch(self, idx, weights=None):
        px, nx, o_p, o_n = [], [], [], []
        if weights is None:
            weights = torch.ones(self.batch_size)
        for i, j in enumerate(range(idx * self.batch_size, (idx + 1) * self.batch_size)):
            q, p, n = self.dataset.get_items(j, float(weights[i]))
            px.append(self.format(q, p))
            nx.append(self.format(q, n))
            o_p.append(OUTPUTS[0])
            o_n.append(OUTPUTS[1])
        return px, nx, o_p, o_n


class LevelLoader(_BatchLoader):
    """Reference ``dataloader.py:58-79``: one pace level for the whole batch."""

    def get_batch(self, idx, weight=None):
        px, nx, o_p, o_n = [], [], [], []
        if weight is None:
            weight = 0.0
        for j in range(idx * self.batch_size, (idx + 1) * self.batch_size):
            q, p, n = self.dataset.get_items(j, weight)
            px.append(self.format(q, p))
            nx.append(self.format(q, n))
            o_p.append(OUTPUTS[0])
            o_n.append(OUTPUTS[1])
        return px, nx, o_p, o_n


class TripletLoader:
    """Reference ``dataloader.py:164-182`` (raw texts; note its range stops one item short)."""

    def __init__(self, dataset, batch_size) -> None:
        self.dataset = dataset
        self.num_items = len(dataset)
        self.batch_size = batch_size

    def __len__(self):
        return self.num_items

    def get_batch(self, idx, weights=None):
        q, p, n = [], [], []
        if weights is None:
            weights = torch.ones((idx + 1) * self.batch_size)
        for i in range(idx * self.batch_size, (idx + 1) * self.batch_size - 1):
            _q, _p, _n = self.dataset.get_items(i, float(weights[i]))
            q.append(_q)
            p.append(_p)
            n.append(_n)
        return q, p, n


class LCEDataset(Dataset):
    def __init__(self, pairs, neg_idx, corpus, tokenizer, init_weight, batch_size: int, var: float, n: int,
                 min: float, max: float, use_max=False) -> None:
        self.neg_idx = [np.array(x) for x in neg_idx]
        self.n_neg = len(neg_idx[0])
        self.docs, self.queries = _text_tables(corpus)
        self.tokenizer = tokenizer
        self.data = [(self.queries[q], self.docs[p]) for q, p in pairs]
        self.batch_size = batch_size
        self.var = var
        self.n = n
        self.round = torch.floor
        self.min = min
        self.max = max
        self._shared_pace = None
        self.weight = init_weight
        self._pace_cache = (None, None)    # (clipped pace, probabilities): items of one batch share a pace

    # The pace is a plain attribute in the reference, so DataLoader workers sample from their own stale copy of it
    # (``lceT5.py:77-80`` writes the parent's).  After ``share_pace()`` it lives in shared memory instead and a write
    # in the training process reaches every worker.
    @property
    def weight(self):
        return float(self._shared_pace[0]) if self._shared_pace is not None else self._weight

    @weight.setter
    def weight(self, value):
        if self._shared_pace is not None:
            self._shared_pace[0] = float(value)
        else:
            self._weight = value

    def share_pace(self):
        if self._shared_pace is None:
            cell = torch.zeros(1, dtype=torch.float64).share_memory_()
            cell[0] = float(self._weight)
            self._shared_pace = cell
        return self

    def get(self, idx):
        return self.data[idx[0]], [self.docs[x] for x in self.neg_idx[idx[0]][idx[1]].tolist()]

    def __len__(self):
        return len(self.data)

    def pace_probabilities(self, mean) -> np.ndarray:
        """Reference ``dataloader.py:135-145``.  The variance rescale multiplies by a constant and
        renormalises; it is kept so that the float64 values equal the reference's bit for bit."""
        mean = float(np.clip(mean, self.min, self.max))
        if self._pace_cache[0] == mean:
            return self._pace_cache[1]
        idx = np.arange(self.n_neg)
        probabilities = binom.pmf(idx, self.n_neg - 1, mean)
        adjusted = probabilities / probabilities.sum()
        adjusted = adjusted * np.sqrt(self.var / np.var(adjusted))
        adjusted = adjusted / adjusted.sum()
        self._pace_cache = (mean, adjusted)
        return adjusted

    def sample(self, mean):
        """``n`` distinct indices into the easy->hard list (reference ``dataloader.py:134-151``; the
        reference prints and exits the process when fewer than ``n`` probabilities are non-zero, this
        raises instead)."""
        p = self.pace_probabilities(mean)
        try:
            return np.random.choice(np.arange(self.n_neg), size=(self.n,), replace=False, p=p)
        except ValueError as e:
            raise ValueError(f"Failed to sample with mean {float(np.clip(mean, self.min, self.max))} and var {self.var}") from e

    def format(self, q, d):
        return format_pair(q, d)

    def __getitem__(self, idx):
        _idx = self.sample(self.weight)
        qp, n = self.get((idx, _idx))
        q, p = qp
        return {"pos": self.format(q, p), "neg": list(map(partial(self.format, q), n))}

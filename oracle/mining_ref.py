"""Restatement of the negative-pool selection contract with a dense dot-product scorer.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).

The reference builds its pools with third-party lexical engines; what it pins is the contract
around the scorer (``pacednegatives/utilities/compute_all_bm25.py:37-44``): per query keep the
top-``cutoff`` by score, drop queries with fewer than ``cutoff`` hits, and reverse the list so that
index 0 is the lowest-scoring (easiest) and index cutoff-1 the highest-scoring (hardest).  The
scorer here is ``scores = Q . Pool^T`` in fp32; ties are broken by (score descending, pool index
ascending), which a stable sort of the negated scores gives.
"""
from __future__ import annotations

import numpy as np


def pool_topk(queries: np.ndarray, pool: np.ndarray, k: int, chunk: int = 65536):
    """Exact top-k of ``queries @ pool.T`` per query.  Inputs are float32 arrays holding values that
    are exactly representable in bf16.  Returns (scores f32 [Q,k], idx i64 [Q,k]) ordered by
    (score desc, idx asc).  The pool is walked in chunks with a running merge so that the
    [Q, Np] score matrix is never held."""
    Q = queries.shape[0]
    best_s = np.full((Q, 0), 0, np.float32)
    best_i = np.full((Q, 0), 0, np.int64)
    for lo in range(0, pool.shape[0], chunk):
        s = (queries.astype(np.float32) @ pool[lo:lo + chunk].astype(np.float32).T).astype(np.float32)
        i = np.broadcast_to(np.arange(lo, lo + s.shape[1], dtype=np.int64), s.shape)
        s = np.concatenate([best_s, s], axis=1)
        i = np.concatenate([best_i, i], axis=1)
        # stable sort by -score keeps ascending index among equal scores (indices are ascending
        # inside [best | new] only per part, so sort by index first)
        order = np.argsort(i, axis=1, kind="stable")
        s = np.take_along_axis(s, order, 1)
        i = np.take_along_axis(i, order, 1)
        order = np.argsort(-s, axis=1, kind="stable")[:, :k]
        best_s = np.take_along_axis(s, order, 1)
        best_i = np.take_along_axis(i, order, 1)
    return best_s, best_i


def easy_to_hard(idx_desc: np.ndarray) -> np.ndarray:
    """``compute_all_bm25.py:43-44``: take the top-``cutoff`` list (best first) and reverse it."""
    return idx_desc[:, ::-1].copy()

"""Restatement of the paced binomial sampler and dataset item formatting.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).
"""
from __future__ import annotations

import numpy as np
from scipy.stats import binom


def pace_probabilities(mean: float, n_neg: int, var: float, lo: float = 1e-10, hi: float = 1.0 - 1e-10) -> np.ndarray:
    """``pacednegatives/dataloader.py:135-145``: binomial pmf over the easy->hard index range at
    pace ``mean``; the ``var`` rescale multiplies by a constant and renormalises (a no-op up to
    rounding) but is kept so that the float64 values match bit for bit."""
    mean = np.clip(mean, lo, hi)
    idx = np.arange(n_neg)
    p = binom.pmf(idx, n_neg - 1, mean)
    adj = p / p.sum()
    adj_var = np.var(adj)
    adj = adj * np.sqrt(var / adj_var)
    adj = adj / adj.sum()
    return adj


def sample(mean: float, n_neg: int, n: int, var: float, rng=np.random) -> np.ndarray:
    """``pacednegatives/dataloader.py:134-151``: ``n`` distinct indices drawn without replacement
    with the pace probabilities, using numpy's legacy global generator (or a ``RandomState``)."""
    p = pace_probabilities(mean, n_neg, var)
    return rng.choice(np.arange(n_neg), size=(n,), replace=False, p=p)


def format_pair(q: str, d: str) -> str:
    """``pacednegatives/dataloader.py:153-155``."""
    return "Query: " + q + " Document: " + d + " Relevant:"

"""Paced negative-pool construction with a dense scorer (counterpart of the reference's
``pacednegatives/utilities/compute_all_bm25.py`` / ``compute_all_splade.py`` / ``group_results.py``).

The reference obtains, per query, the top-``cutoff`` documents of a lexical retriever, drops queries
with fewer than ``cutoff`` hits, REVERSES the list so that index 0 is the lowest-scoring (easiest) and
index cutoff-1 the best-scoring (hardest) document (``compute_all_bm25.py:37-44``), and joins the lists
with (query, positive) pairs into records ``{query_id, doc_id_a, doc_id_b}``
(``group_results.py:34-38``) -- the file ``train_lce.py --data`` reads.  Here the retriever is a
dot-product scorer over a bf16 document-embedding pool, computed by ``paced_pool_topk`` (exact top-k,
ties broken by lower pool row), optionally sharded over the GPUs of a node by pool rows.
"""
from __future__ import annotations

import json
from typing import Iterable, List, Optional, Sequence

import torch

from . import kernels as K
from . import parallel


def pool_topk(query_emb: torch.Tensor, pool_emb: torch.Tensor, k: int, index_base: int = 0, workspace=None):
    """(scores f32 [Q,k], rows i64 [Q,k]) best first; entries beyond the pool size are (-inf, -1)."""
    q = query_emb.to(torch.bfloat16).contiguous()
    p = pool_emb.to(torch.bfloat16).contiguous()
    return K.pool_topk(q, p, k, index_base=index_base, workspace=workspace)


def pool_topk_sharded(query_emb: torch.Tensor, pool_shard: torch.Tensor, k: int, shard_row_offset: int, workspace=None):
    """Every rank holds a contiguous slice of pool rows starting at ``shard_row_offset`` and the full
    query set; returns the global top-k on every rank (all_gather of per-shard results + merge)."""
    s, i = pool_topk(query_emb, pool_shard, k, index_base=shard_row_offset, workspace=workspace)
    world = parallel.world_size()
    if world == 1:
        return s, i
    Q = s.shape[0]
    gs = parallel.all_gather_stack(s).permute(1, 0, 2).reshape(Q, world * k).contiguous()      # [Q, W*k]
    gi = parallel.all_gather_stack(i).permute(1, 0, 2).reshape(Q, world * k).contiguous()
    return K.topk_merge(gs, gi, k)


def easy_to_hard(rows_best_first: torch.Tensor) -> torch.Tensor:
    """``compute_all_bm25.py:43-44``: keep the top-``cutoff`` list and reverse it."""
    return torch.flip(rows_best_first, dims=[1])


def mine_negatives(query_ids: Sequence, query_emb: torch.Tensor, pool_emb: torch.Tensor, cutoff: int = 1000,
                   doc_ids: Optional[Sequence] = None) -> List[dict]:
    """Records ``{"qid", "doc_id_b"}`` (easy -> hard) for every query with at least ``cutoff`` hits."""
    _scores, rows = pool_topk(query_emb, pool_emb, cutoff)
    rows = easy_to_hard(rows).cpu()
    out = []
    for qid, r in zip(query_ids, rows.tolist()):
        if r and min(r) < 0:                  # fewer than cutoff documents exist: the reference drops the query
            continue
        out.append({"qid": qid, "doc_id_b": [doc_ids[j] if doc_ids is not None else str(j) for j in r]})
    return out


def group_results(results: Iterable[dict], docpairs: Iterable[dict]) -> List[dict]:
    """``group_results.py:25-36``: attach each query's easy->hard list to its (query_id, doc_id_a) pairs."""
    lookup = {str(r["qid"]): r["doc_id_b"] for r in results}
    return [{"query_id": str(p["query_id"]), "doc_id_a": str(p["doc_id_a"]), "doc_id_b": lookup[str(p["query_id"])]}
            for p in docpairs if str(p["query_id"]) in lookup]


def write_records(path: str, records: List[dict]) -> None:
    with open(path, "w") as f:
        json.dump(records, f)

"""Tokenizer access for the lceT5 path.

The reference calls ``T5Tokenizer.from_pretrained(model_name)`` (pacednegatives/lceT5.py:107), which
needs the SentencePiece model of the checkpoint.  When ``model_name`` is a local directory that holds
one, Hugging Face's tokenizer is used unchanged.  Otherwise (size names such as "t5-base" cannot be
fetched without a network) a deterministic stand-in is returned that keeps the tokenizer CALL
contract the collate function relies on -- ``tok(list_of_str, padding=True, return_tensors="pt")``
returning ``input_ids`` padded with 0 and ending in </s>=1, and "true"/"false" mapping to the T5
vocabulary ids 1176/6136 -- so that synthetic-data runs exercise the same plumbing.
"""
from __future__ import annotations

import os
import re
import zlib

import torch

TRUE_ID, FALSE_ID, EOS_ID, PAD_ID = 1176, 6136, 1, 0


class _Encoding(dict):
    __getattr__ = dict.get


class SyntheticTokenizer:
    """Whitespace/punctuation word hashing into [2, vocab) (not a language tokenizer)."""

    def __init__(self, vocab_size: int = 32128):
        self.vocab_size = vocab_size
        self.pad_token_id, self.eos_token_id = PAD_ID, EOS_ID

    def _word(self, w: str) -> int:
        if w == "true":
            return TRUE_ID
        if w == "false":
            return FALSE_ID
        return 2 + zlib.crc32(w.encode("utf-8")) % (self.vocab_size - 2)

    def encode(self, text: str):
        return [self._word(w) for w in re.findall(r"\w+|[^\w\s]", text)] + [EOS_ID]

    def __call__(self, texts, padding=True, return_tensors="pt", **_kw):
        if isinstance(texts, str):
            texts = [texts]
        rows = [self.encode(t) for t in texts]
        width = max(len(r) for r in rows)
        ids = torch.full((len(rows), width), PAD_ID, dtype=torch.int64)
        mask = torch.zeros((len(rows), width), dtype=torch.int64)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = torch.tensor(r, dtype=torch.int64)
            mask[i, :len(r)] = 1
        return _Encoding(input_ids=ids, attention_mask=mask)

    def save_pretrained(self, directory: str):
        os.makedirs(directory, exist_ok=True)


def load_tokenizer(name_or_dir: str, vocab_size: int):
    if os.path.isdir(name_or_dir) and any(os.path.isfile(os.path.join(name_or_dir, f)) for f in ("spiece.model", "tokenizer.json")):
        from transformers import AutoTokenizer
        return AutoTokenizer.from_pretrained(name_or_dir)
    return SyntheticTokenizer(vocab_size)

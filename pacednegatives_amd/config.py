"""T5 architecture description (the fields of Hugging Face ``T5Config`` the hot path depends on)."""
from __future__ import annotations

import json
import os
from dataclasses import asdict, dataclass


@dataclass(frozen=True)
class T5Dims:
    vocab_size: int = 32128
    d_model: int = 768
    d_kv: int = 64
    num_heads: int = 12
    d_ff: int = 3072
    num_layers: int = 12
    num_decoder_layers: int = 12
    num_buckets: int = 32
    max_distance: int = 128
    eps: float = 1e-6
    dropout_rate: float = 0.1

    @property
    def inner(self) -> int:
        return self.num_heads * self.d_kv

    def validate(self) -> None:
        if self.d_kv != 64:
            raise ValueError(f"d_kv={self.d_kv}: the attention kernels are built for head size 64 only")
        for name in ("d_model", "d_ff", "inner"):
            v = getattr(self, name)
            if v % 64:
                raise ValueError(f"{name}={v} must be a multiple of 64 (K step of the MFMA contraction)")
        if self.vocab_size % 64:
            raise ValueError(f"vocab_size={self.vocab_size} must be a multiple of 64")
        if self.d_model > 2048:
            raise ValueError("d_model above 2048 is not built (RMSNorm backward register budget)")

    # ---- Hugging Face config.json round trip (reference: T5ForConditionalGeneration.from_pretrained,
    # pacednegatives/lceT5.py:106, and save_pretrained, train/train_lce.py:103)
    def to_hf_config(self) -> dict:
        return {
            "architectures": ["T5ForConditionalGeneration"], "model_type": "t5",
            "vocab_size": self.vocab_size, "d_model": self.d_model, "d_kv": self.d_kv, "d_ff": self.d_ff,
            "num_layers": self.num_layers, "num_decoder_layers": self.num_decoder_layers, "num_heads": self.num_heads,
            "relative_attention_num_buckets": self.num_buckets, "relative_attention_max_distance": self.max_distance,
            "dropout_rate": self.dropout_rate, "layer_norm_epsilon": self.eps, "initializer_factor": 1.0,
            "feed_forward_proj": "relu", "is_encoder_decoder": True, "use_cache": True, "tie_word_embeddings": True,
            "pad_token_id": 0, "eos_token_id": 1, "decoder_start_token_id": 0,
        }

    @staticmethod
    def from_hf_config(cfg: dict) -> "T5Dims":
        if cfg.get("feed_forward_proj", "relu") != "relu":
            raise ValueError("only the ReLU feed-forward of the original T5 checkpoints is built")
        if not cfg.get("tie_word_embeddings", True):
            raise ValueError("untied LM heads are not built")
        return T5Dims(
            vocab_size=cfg["vocab_size"], d_model=cfg["d_model"], d_kv=cfg["d_kv"], num_heads=cfg["num_heads"],
            d_ff=cfg["d_ff"], num_layers=cfg["num_layers"],
            num_decoder_layers=cfg.get("num_decoder_layers") or cfg["num_layers"],
            num_buckets=cfg.get("relative_attention_num_buckets", 32),
            max_distance=cfg.get("relative_attention_max_distance", 128),
            eps=cfg.get("layer_norm_epsilon", 1e-6), dropout_rate=cfg.get("dropout_rate", 0.1))

    def save(self, directory: str) -> None:
        with open(os.path.join(directory, "config.json"), "w") as f:
            json.dump(self.to_hf_config(), f, indent=2)

    def asdict(self) -> dict:
        return asdict(self)


NAMED = {
    "t5-small": T5Dims(32128, 512, 64, 8, 2048, 6, 6),
    "t5-base": T5Dims(32128, 768, 64, 12, 3072, 12, 12),
    "t5-large": T5Dims(32128, 1024, 64, 16, 4096, 24, 24),
    # test-sized model with real label ids in range (true=1176, false=6136)
    "t5-tiny": T5Dims(6400, 128, 64, 2, 256, 2, 2),
}


def resolve(name_or_dir: str) -> T5Dims:
    """A named size, or a local Hugging Face checkpoint directory containing config.json."""
    if name_or_dir in NAMED:
        return NAMED[name_or_dir]
    cfg_path = os.path.join(name_or_dir, "config.json")
    if os.path.isfile(cfg_path):
        with open(cfg_path) as f:
            return T5Dims.from_hf_config(json.load(f))
    raise ValueError(f"unknown model {name_or_dir!r}: pass one of {sorted(NAMED)} or a local checkpoint directory")

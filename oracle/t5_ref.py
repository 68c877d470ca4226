"""Plain-torch fp32 restatement of the T5 encoder-decoder forward the reference calls.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).

Reference call sites: ``pacednegatives/lceT5.py:132,133,149,150`` call
``T5ForConditionalGeneration(input_ids=, labels=).logits`` with no attention
mask (``lceT5.py:53`` drops it), so padding tokens are attended.  The arithmetic
is Hugging Face ``modeling_t5.py`` (transformers 5.15.0), cited as ``HF:line``.

Parameters are held in a plain ``dict`` keyed by the Hugging Face state-dict
names so that a ``T5ForConditionalGeneration.state_dict()`` can be passed in
unchanged.  Backward is torch autograd over this forward.

``emulate_bf16=True`` rounds weights and every tensor the HIP path stores in
HBM to bfloat16 at the same points (GEMM outputs, normalised activations,
attention probabilities, residual sums), keeping fp32 accumulation, so that
the HIP path can be compared at a tight tolerance.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch


@dataclass(frozen=True)
class T5Dims:
    """Shape of a T5 checkpoint (HF ``T5Config`` fields of the same meaning)."""
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

    @property
    def inner(self) -> int:
        return self.num_heads * self.d_kv

    @staticmethod
    def named(name: str) -> "T5Dims":
        table = {
            "t5-small": T5Dims(32128, 512, 64, 8, 2048, 6, 6),
            "t5-base": T5Dims(32128, 768, 64, 12, 3072, 12, 12),
            "t5-large": T5Dims(32128, 1024, 64, 16, 4096, 24, 24),
            "t5-tiny": T5Dims(512, 128, 64, 2, 256, 2, 2),
        }
        return table[name]


def param_names(dims: T5Dims):
    """HF state-dict names and shapes, in HF module order."""
    d, i, f = dims.d_model, dims.inner, dims.d_ff
    out = [("shared.weight", (dims.vocab_size, d))]
    for l in range(dims.num_layers):
        p = f"encoder.block.{l}.layer."
        out += [(p + "0.SelfAttention.q.weight", (i, d)), (p + "0.SelfAttention.k.weight", (i, d)),
                (p + "0.SelfAttention.v.weight", (i, d)), (p + "0.SelfAttention.o.weight", (d, i))]
        if l == 0:
            out.append((p + "0.SelfAttention.relative_attention_bias.weight", (dims.num_buckets, dims.num_heads)))
        out += [(p + "0.layer_norm.weight", (d,)),
                (p + "1.DenseReluDense.wi.weight", (f, d)), (p + "1.DenseReluDense.wo.weight", (d, f)),
                (p + "1.layer_norm.weight", (d,))]
    out.append(("encoder.final_layer_norm.weight", (d,)))
    for l in range(dims.num_decoder_layers):
        p = f"decoder.block.{l}.layer."
        out += [(p + "0.SelfAttention.q.weight", (i, d)), (p + "0.SelfAttention.k.weight", (i, d)),
                (p + "0.SelfAttention.v.weight", (i, d)), (p + "0.SelfAttention.o.weight", (d, i))]
        if l == 0:
            out.append((p + "0.SelfAttention.relative_attention_bias.weight", (dims.num_buckets, dims.num_heads)))
        out += [(p + "0.layer_norm.weight", (d,)),
                (p + "1.EncDecAttention.q.weight", (i, d)), (p + "1.EncDecAttention.k.weight", (i, d)),
                (p + "1.EncDecAttention.v.weight", (i, d)), (p + "1.EncDecAttention.o.weight", (d, i)),
                (p + "1.layer_norm.weight", (d,)),
                (p + "2.DenseReluDense.wi.weight", (f, d)), (p + "2.DenseReluDense.wo.weight", (d, f)),
                (p + "2.layer_norm.weight", (d,))]
    out.append(("decoder.final_layer_norm.weight", (d,)))
    return out


def init_params(dims: T5Dims, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Seeded random initialisation with the standard deviations of HF ``_init_weights``
    (factor 1.0): shared N(0,1); q N(0,(d*dkv)^-.5); k,v,wi N(0,d^-.5); o N(0,inner^-.5);
    wo N(0,ff^-.5); relative bias N(0,d^-.5); norms 1.  The draw order is this
    function's own (it is not bit-identical to HF's initialiser)."""
    g = torch.Generator().manual_seed(seed)
    d, i, f = dims.d_model, dims.inner, dims.d_ff
    params = {}
    for name, shape in param_names(dims):
        if name.endswith("layer_norm.weight"):
            t = torch.ones(shape)
        else:
            if name == "shared.weight":
                std = 1.0
            elif name.endswith(".q.weight"):
                std = (d * dims.d_kv) ** -0.5
            elif name.endswith(".k.weight") or name.endswith(".v.weight") or name.endswith("wi.weight"):
                std = d ** -0.5
            elif name.endswith(".o.weight"):
                std = i ** -0.5
            elif name.endswith("wo.weight"):
                std = f ** -0.5
            else:  # relative_attention_bias
                std = d ** -0.5
            t = torch.randn(shape, generator=g) * std
        params[name] = t.to(dtype)
    return params


def relative_position_bucket(rel: torch.Tensor, bidirectional: bool, num_buckets: int, max_distance: int) -> torch.Tensor:
    """HF:217-262.  ``rel`` = key position - query position (int64)."""
    buckets = torch.zeros_like(rel)
    if bidirectional:
        num_buckets //= 2
        buckets = buckets + (rel > 0).to(torch.long) * num_buckets
        rel = rel.abs()
    else:
        rel = -torch.minimum(rel, torch.zeros_like(rel))
    max_exact = num_buckets // 2
    is_small = rel < max_exact
    large = max_exact + (
        torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)
    ).to(torch.long)
    large = torch.minimum(large, torch.full_like(large, num_buckets - 1))
    return buckets + torch.where(is_small, rel, large)


def position_bias(table: torch.Tensor, q_len: int, k_len: int, bidirectional: bool, dims: T5Dims) -> torch.Tensor:
    """HF:264-279: [1, H, q_len, k_len] additive bias from the [num_buckets, H] table."""
    ctx = torch.arange(q_len)[:, None]
    mem = torch.arange(k_len)[None, :]
    b = relative_position_bucket(mem - ctx, bidirectional, dims.num_buckets, dims.max_distance)
    return table[b].permute(2, 0, 1).unsqueeze(0)


# Storage points at which the HIP path holds a bf16 value (names accepted by ``t5_forward(keep_fp32=...)``):
#   params      every weight (the bf16 shadow the kernels read)
#   enc_stream  the encoder's residual stream (embedding output and both residual sums of every layer)
#   enc_norm    the encoder's normalised rows, the final norm included (= the encoder output)
#   enc_qkv     q, k, v of the encoder
#   enc_probs   un-normalised attention probabilities fed to the PV product
#   enc_ctx     attention context
#   enc_act     feed-forward activations
#   dec_stream, dec_norm, dec_qkv, dec_probs, dec_ctx, dec_act   the same for the decoder's two-token rows (q/k/v and
#               context of its cross-attention included)
#   head_in     the rescaled final decoder norm that feeds the LM head
ROUNDING_POINTS = ("params", "enc_stream", "enc_norm", "enc_qkv", "enc_probs", "enc_ctx", "enc_act",
                   "dec_stream", "dec_norm", "dec_qkv", "dec_probs", "dec_ctx", "dec_act", "head_in")


class _R:
    """Rounding policy: identity in fp32 mode, round-trip through bf16 when emulating, except at the storage points
    named in ``keep`` (for experiments that ask what each point contributes to the error)."""

    def __init__(self, emulate: bool, keep=()):
        self.emulate = emulate
        self.keep = frozenset(keep)
        unknown = self.keep - set(ROUNDING_POINTS)
        if unknown:
            raise ValueError(f"unknown storage points {sorted(unknown)}")

    def __call__(self, x: torch.Tensor, point: str = None) -> torch.Tensor:
        if not self.emulate or (point is not None and point in self.keep):
            return x
        return x.to(torch.bfloat16).to(torch.float32)

    def at(self, point: str):
        """The policy bound to one storage point (what ``attention`` receives for its probabilities)."""
        return lambda x: self(x, point)


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """HF:59-72: fp32 variance, no mean subtraction, weight only."""
    var = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


def _heads(x: torch.Tensor, h: int, dk: int) -> torch.Tensor:
    return x.view(x.shape[0], x.shape[1], h, dk).transpose(1, 2)


def attention(q, k, v, bias, causal: bool, r: _R, prob_drop: Optional[torch.Tensor] = None) -> torch.Tensor:
    """HF:144-173 with ``scaling = 1.0`` (HF:196-197).  q,k,v: [N,H,T,dk]; bias [1,H,T,S] or None.
    Returns [N, T, H*dk].  Softmax in fp32; when emulating, probabilities are rounded to bf16
    un-normalised (exp(s - rowmax)) before the PV product and divided by the fp32 row sum after it,
    as the flash-style HIP kernel does."""
    s = torch.matmul(q, k.transpose(2, 3))
    if bias is not None:
        s = s + bias
    if causal:
        t, sl = s.shape[-2], s.shape[-1]
        mask = torch.ones(t, sl, dtype=torch.bool).tril(diagonal=sl - t)
        s = s.masked_fill(~mask, float("-inf"))
    m = s.max(dim=-1, keepdim=True).values
    p = torch.exp(s - m)
    l = p.sum(dim=-1, keepdim=True)
    if prob_drop is not None:          # prob_drop already holds mask/(1-p)
        p = p * prob_drop
    o = torch.matmul(r(p), v) / l
    o = o.transpose(1, 2).contiguous()
    return o.view(o.shape[0], o.shape[1], -1)


def _lin(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    return torch.matmul(x, w.t())


def t5_forward(params: Dict[str, torch.Tensor], dims: T5Dims, input_ids: torch.Tensor, labels: torch.Tensor,
               emulate_bf16: bool = False, return_internals: bool = False, drop=None, keep_fp32=(), relu_mask=None):
    """Logits [N, T, V] (fp32) of ``T5ForConditionalGeneration(input_ids, labels)`` with dropout 0, or with
    caller-supplied dropout masks: ``drop(site, layer, dec, shape)`` returns the factor tensor (0 or 1/(1-p)) of
    that shape for the dropout sites of HF modeling_t5.py - "emb" (:725), "final" (:745), "att" (:168, on the
    probabilities, shape [N,H,T,S]), "att_out" (:400), "ff_act" (:86), "ff_out" (:140), "catt" / "catt_out"
    (:168 / :431 of the cross-attention) - so that a test can replay the masks another implementation used.
    ``relu_mask(layer, dec, shape)`` (optional) returns a 0/1 tensor, or None, that REPLACES the ReLU's own decision
    for that feed-forward layer (a = pre-activation * mask, gradient passes where mask = 1): a unit whose
    pre-activation is a rounding error away from zero can fall on either side in two implementations, and replaying
    the other implementation's decisions takes that discontinuity out of a gradient comparison.
    ``keep_fp32``: storage points (``ROUNDING_POINTS``) NOT rounded to bf16 under ``emulate_bf16``.

    Structure: HF:663-750 (stack), HF:392-400/304-367 (self-attention sub-layer), HF:138-140/84-93
    (feed-forward sub-layer), HF:459-498 (decoder block), HF:618-637 (shift right), HF:1044-1047
    (tied-head rescale + lm_head).  No attention mask on the encoder (reference drops it), causal
    mask on decoder self-attention, zero bias on cross-attention (HF:337-342)."""
    r = _R(emulate_bf16, keep_fp32)
    P = {k: r(v, "params") for k, v in params.items()} if emulate_bf16 else params

    def relu(layer, dec, x):
        m = None if relu_mask is None else relu_mask(layer, dec, tuple(x.shape))
        return torch.relu(x) if m is None else x * m
    H, dk, eps = dims.num_heads, dims.d_kv, dims.eps
    internals = {}

    def dm(site, layer, dec, x):            # dropout on a hidden-state tensor
        return x if drop is None else x * drop(site, layer, dec, tuple(x.shape))

    def pm(site, layer, dec, n, t, sl):     # dropout factors for attention probabilities
        return None if drop is None else drop(site, layer, dec, (n, H, t, sl))
    N = input_ids.shape[0]

    # ---- encoder
    h = r(dm("emb", 0, False, P["shared.weight"][input_ids]), "enc_stream")  # [N,S,d]
    S = input_ids.shape[1]
    enc_bias = position_bias(P["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"], S, S, True, dims)
    for l in range(dims.num_layers):
        p = f"encoder.block.{l}.layer."
        xn = r(rms_norm(h, P[p + "0.layer_norm.weight"], eps), "enc_norm")
        q = r(_lin(xn, P[p + "0.SelfAttention.q.weight"]), "enc_qkv")
        k = r(_lin(xn, P[p + "0.SelfAttention.k.weight"]), "enc_qkv")
        v = r(_lin(xn, P[p + "0.SelfAttention.v.weight"]), "enc_qkv")
        ctx = r(attention(_heads(q, H, dk), _heads(k, H, dk), _heads(v, H, dk), enc_bias, False, r.at("enc_probs"),
                          pm("att", l, False, N, S, S)), "enc_ctx")
        h = r(h + dm("att_out", l, False, _lin(ctx, P[p + "0.SelfAttention.o.weight"])), "enc_stream")
        xn = r(rms_norm(h, P[p + "1.layer_norm.weight"], eps), "enc_norm")
        a = r(dm("ff_act", l, False, relu(l, False, _lin(xn, P[p + "1.DenseReluDense.wi.weight"]))), "enc_act")
        h = r(h + dm("ff_out", l, False, _lin(a, P[p + "1.DenseReluDense.wo.weight"])), "enc_stream")
        if return_internals:
            internals[f"enc{l}"] = h
    enc = r(dm("final", 0, False, rms_norm(h, P["encoder.final_layer_norm.weight"], eps)), "enc_norm")
    if return_internals:
        internals["enc_out"] = enc

    # ---- decoder
    dec_ids = torch.zeros_like(labels)
    dec_ids[:, 1:] = labels[:, :-1]
    dec_ids = dec_ids.masked_fill(dec_ids == -100, 0)
    T = dec_ids.shape[1]
    h = r(dm("emb", 0, True, P["shared.weight"][dec_ids]), "dec_stream")
    dec_bias = position_bias(P["decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"], T, T, False, dims)
    for l in range(dims.num_decoder_layers):
        p = f"decoder.block.{l}.layer."
        xn = r(rms_norm(h, P[p + "0.layer_norm.weight"], eps), "dec_norm")
        q = r(_lin(xn, P[p + "0.SelfAttention.q.weight"]), "dec_qkv")
        k = r(_lin(xn, P[p + "0.SelfAttention.k.weight"]), "dec_qkv")
        v = r(_lin(xn, P[p + "0.SelfAttention.v.weight"]), "dec_qkv")
        ctx = r(attention(_heads(q, H, dk), _heads(k, H, dk), _heads(v, H, dk), dec_bias, True, r.at("dec_probs"),
                          pm("att", l, True, N, T, T)), "dec_ctx")
        h = r(h + dm("att_out", l, True, _lin(ctx, P[p + "0.SelfAttention.o.weight"])), "dec_stream")
        xn = r(rms_norm(h, P[p + "1.layer_norm.weight"], eps), "dec_norm")
        q = r(_lin(xn, P[p + "1.EncDecAttention.q.weight"]), "dec_qkv")
        k = r(_lin(enc, P[p + "1.EncDecAttention.k.weight"]), "dec_qkv")
        v = r(_lin(enc, P[p + "1.EncDecAttention.v.weight"]), "dec_qkv")
        ctx = r(attention(_heads(q, H, dk), _heads(k, H, dk), _heads(v, H, dk), None, False, r.at("dec_probs"),
                          pm("catt", l, True, N, T, S)), "dec_ctx")
        h = r(h + dm("catt_out", l, True, _lin(ctx, P[p + "1.EncDecAttention.o.weight"])), "dec_stream")
        xn = r(rms_norm(h, P[p + "2.layer_norm.weight"], eps), "dec_norm")
        a = r(dm("ff_act", l, True, relu(l, True, _lin(xn, P[p + "2.DenseReluDense.wi.weight"]))), "dec_act")
        h = r(h + dm("ff_out", l, True, _lin(a, P[p + "2.DenseReluDense.wo.weight"])), "dec_stream")
        if return_internals:
            internals[f"dec{l}"] = h
    y = dm("final", 0, True, rms_norm(h, P["decoder.final_layer_norm.weight"], eps))
    y = r(y * (dims.d_model ** -0.5), "head_in")               # HF:1044-1045 (tied embeddings)
    logits = _lin(y, P["shared.weight"])                       # HF:1047
    if return_internals:
        return logits, internals
    return logits

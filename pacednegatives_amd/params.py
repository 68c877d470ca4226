"""Flat parameter storage for the T5 engine.

All parameters live in ONE fp32 master buffer with a bf16 shadow (what the kernels read), an fp32
gradient buffer and two fp32 AdamW moment buffers of the same layout, so that the optimiser and the
data-parallel all-reduce are single passes over contiguous HBM.  Attention projections are stored
fused ([q;k;v] for self-attention, [k;v] for cross-attention) because that is how the GEMMs consume
them; ``to_hf`` / ``from_hf`` translate to and from Hugging Face state-dict names.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Iterator, Tuple

import torch

from .config import T5Dims

ALIGN = 64   # elements; keeps every segment 256-byte (fp32) / 128-byte (bf16) aligned


def layout(dims: T5Dims) -> "OrderedDict[str, Tuple[int, Tuple[int, ...]]]":
    d, i, f, H = dims.d_model, dims.inner, dims.d_ff, dims.num_heads
    segs = [("shared", (dims.vocab_size, d))]
    for l in range(dims.num_layers):
        p = f"enc.{l}."
        segs += [(p + "ln1", (d,)), (p + "qkv", (3 * i, d)), (p + "o", (d, i)),
                 (p + "ln2", (d,)), (p + "wi", (f, d)), (p + "wo", (d, f))]
    segs += [("enc.rel_bias", (dims.num_buckets, H)), ("enc.final_ln", (d,))]
    for l in range(dims.num_decoder_layers):
        p = f"dec.{l}."
        segs += [(p + "ln1", (d,)), (p + "qkv", (3 * i, d)), (p + "o", (d, i)),
                 (p + "ln2", (d,)), (p + "cq", (i, d)), (p + "ckv", (2 * i, d)), (p + "co", (d, i)),
                 (p + "ln3", (d,)), (p + "wi", (f, d)), (p + "wo", (d, f))]
    segs += [("dec.rel_bias", (dims.num_buckets, H)), ("dec.final_ln", (d,))]
    out, off = OrderedDict(), 0
    for name, shape in segs:
        n = 1
        for s in shape:
            n *= s
        out[name] = (off, shape)
        off += (n + ALIGN - 1) // ALIGN * ALIGN
    out["__total__"] = (off, ())
    return out


def hf_name_map(dims: T5Dims):
    """Yield (hf_name, segment, row_start, row_count): which rows of which fused segment a HF tensor is."""
    i = dims.inner
    yield "shared.weight", "shared", 0, dims.vocab_size
    for stack, nl in (("encoder", dims.num_layers), ("decoder", dims.num_decoder_layers)):
        s = "enc" if stack == "encoder" else "dec"
        for l in range(nl):
            hp, sp = f"{stack}.block.{l}.layer.", f"{s}.{l}."
            for j, nm in enumerate("qkv"):
                yield hp + f"0.SelfAttention.{nm}.weight", sp + "qkv", j * i, i
            yield hp + "0.SelfAttention.o.weight", sp + "o", 0, dims.d_model
            yield hp + "0.layer_norm.weight", sp + "ln1", 0, dims.d_model
            if l == 0:
                yield hp + "0.SelfAttention.relative_attention_bias.weight", s + ".rel_bias", 0, dims.num_buckets
            if stack == "encoder":
                ff, ln = "1", "ln2"
            else:
                yield hp + "1.EncDecAttention.q.weight", sp + "cq", 0, i
                yield hp + "1.EncDecAttention.k.weight", sp + "ckv", 0, i
                yield hp + "1.EncDecAttention.v.weight", sp + "ckv", i, i
                yield hp + "1.EncDecAttention.o.weight", sp + "co", 0, dims.d_model
                yield hp + "1.layer_norm.weight", sp + "ln2", 0, dims.d_model
                ff, ln = "2", "ln3"
            yield hp + f"{ff}.DenseReluDense.wi.weight", sp + "wi", 0, dims.d_ff
            yield hp + f"{ff}.DenseReluDense.wo.weight", sp + "wo", 0, dims.d_model
            yield hp + f"{ff}.layer_norm.weight", sp + ln, 0, dims.d_model
        yield f"{stack}.final_layer_norm.weight", s + ".final_ln", 0, dims.d_model


class ParamStore:
    def __init__(self, dims: T5Dims, device="cuda", with_optimizer_state: bool = True):
        dims.validate()
        self.dims = dims
        self.layout = layout(dims)
        self.total = self.layout["__total__"][0]
        self.device = torch.device(device)
        self.p32 = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.p16 = torch.zeros(self.total, dtype=torch.bfloat16, device=self.device)
        self.g32 = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.m = self.v = None
        if with_optimizer_state:
            self.m = torch.zeros(self.total, dtype=torch.float32, device=self.device)
            self.v = torch.zeros(self.total, dtype=torch.float32, device=self.device)

    # ---- views
    def _view(self, buf, name):
        off, shape = self.layout[name]
        n = 1
        for s in shape:
            n *= s
        return buf[off:off + n].view(shape)

    def w(self, name):       # bf16 weights the kernels read
        return self._view(self.p16, name)

    def master(self, name):
        return self._view(self.p32, name)

    def grad(self, name):
        return self._view(self.g32, name)

    def span(self, first: str, last: str) -> Tuple[int, int]:
        """[lo, hi) of the flat buffers covering segments ``first`` .. ``last`` (layout order), padding included."""
        keys = list(self.layout)
        lo = self.layout[first][0]
        hi = self.layout[keys[keys.index(last) + 1]][0]
        if hi <= lo:
            raise ValueError(f"{first!r} does not precede {last!r} in the layout")
        return lo, hi

    def names(self) -> Iterator[str]:
        return (k for k in self.layout if k != "__total__")

    def num_parameters(self) -> int:
        n = 0
        for k in self.names():
            c = 1
            for s in self.layout[k][1]:
                c *= s
            n += c
        return n

    # ---- initialisation / IO
    def refresh_shadow(self):
        from . import kernels as K
        K.cast_f32_bf16(self.p32, self.p16)

    def load_hf(self, state: Dict[str, torch.Tensor]):
        """Fill the master copy from Hugging Face-named tensors (any float dtype, any device)."""
        seen = set()
        for hf, seg, r0, rc in hf_name_map(self.dims):
            if hf not in state:
                raise KeyError(f"checkpoint is missing {hf}")
            t = state[hf].to(device=self.device, dtype=torch.float32)
            dst = self.master(seg)
            if dst.dim() == 1:
                dst.copy_(t)
            else:
                if tuple(t.shape) != (rc, dst.shape[1]):
                    raise ValueError(f"{hf}: shape {tuple(t.shape)} does not match ({rc}, {dst.shape[1]})")
                dst[r0:r0 + rc].copy_(t)
            seen.add(hf)
        self.refresh_shadow()
        return seen

    def to_hf(self, source: str = "master") -> "OrderedDict[str, torch.Tensor]":
        get = {"master": self.master, "grad": self.grad, "bf16": self.w}[source]
        out = OrderedDict()
        for hf, seg, r0, rc in hf_name_map(self.dims):
            t = get(seg)
            out[hf] = (t if t.dim() == 1 else t[r0:r0 + rc]).detach().clone()
        return out

    def init_random(self, seed: int = 0):
        """Seeded random initialisation with Hugging Face's T5 standard deviations (factor 1.0):
        shared N(0,1); q N(0,(d*dkv)^-.5); k, v, wi, relative bias N(0,d^-.5); o N(0,inner^-.5);
        wo N(0,ff^-.5); norms 1.  Drawn on the CPU generator so every rank gets identical values."""
        dims = self.dims
        g = torch.Generator().manual_seed(seed)
        state = {}
        for hf, seg, r0, rc in hf_name_map(dims):
            shape = (rc,) if self.layout[seg][1].__len__() == 1 else (rc, self.layout[seg][1][1])
            if hf.endswith("layer_norm.weight"):
                t = torch.ones(shape)
            else:
                if hf == "shared.weight":
                    std = 1.0
                elif hf.endswith(".q.weight"):
                    std = (dims.d_model * dims.d_kv) ** -0.5
                elif hf.endswith(".o.weight"):
                    std = dims.inner ** -0.5
                elif hf.endswith("wo.weight"):
                    std = dims.d_ff ** -0.5
                else:
                    std = dims.d_model ** -0.5
                t = torch.randn(shape, generator=g) * std
            state[hf] = t
        self.load_hf(state)

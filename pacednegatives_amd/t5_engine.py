"""T5 encoder-decoder forward and hand-written backward, as an explicit sequence of HIP kernel calls.

What the reference reaches through ``self.model(input_ids=, labels=).logits`` and
``manual_backward`` (pacednegatives/lceT5.py:132-133,149-155) is executed here as launches of the
kernels in libpaced_hip.so: no autograd graph, activations held in bf16, residual stream in bf16,
statistics (rstd, log-sum-exp) and gradients of parameters in fp32.

Token layout: sequences are packed as rows (n*S + s) of [N*S, width] matrices; the decoder's two
target positions as rows (n*2 + t).  Dropout sites follow Hugging Face T5 (HF:725,745 embeddings and
final norm; HF:168 attention probabilities; HF:400/431 attention outputs; HF:86,140 feed-forward);
every site gets its own seed derived from (step seed, call id, site id) and masks are recomputed in
the backward pass from the same seeds.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from . import kernels as K
from .config import T5Dims
from .params import ParamStore

BF16, F32 = torch.bfloat16, torch.float32


def relative_position_bucket(rel: torch.Tensor, bidirectional: bool, num_buckets: int, max_distance: int) -> torch.Tensor:
    """HF:217-262, evaluated with the same torch fp32 operations so that bucket edges agree exactly."""
    buckets = torch.zeros_like(rel)
    if bidirectional:
        num_buckets //= 2
        buckets = buckets + (rel > 0).to(torch.long) * num_buckets
        rel = rel.abs()
    else:
        rel = -torch.minimum(rel, torch.zeros_like(rel))
    max_exact = num_buckets // 2
    is_small = rel < max_exact
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact)
                         * (num_buckets - max_exact)).to(torch.long)
    large = torch.minimum(large, torch.full_like(large, num_buckets - 1))
    return buckets + torch.where(is_small, rel, large)


def _mix(*xs: int) -> int:
    h = 0x9E3779B9
    for x in xs:
        h ^= (x + 0x7F4A7C15 + ((h << 6) & 0xFFFFFFFF) + (h >> 2)) & 0xFFFFFFFF
        h = (h * 0x85EBCA6B) & 0xFFFFFFFF
        h ^= h >> 13
    return h & 0xFFFFFFFF


@dataclass
class _EncLayerSaved:
    h_in: torch.Tensor = None
    xn: torch.Tensor = None
    rstd1: torch.Tensor = None
    qkv: torch.Tensor = None
    ctx: torch.Tensor = None
    lse: torch.Tensor = None
    h_mid: torch.Tensor = None
    actbits: torch.Tensor = None
    xn2: torch.Tensor = None
    rstd2: torch.Tensor = None
    act: torch.Tensor = None


@dataclass
class _DecLayerSaved:
    g_in: torch.Tensor = None
    xn: torch.Tensor = None
    rstd1: torch.Tensor = None
    qkv: torch.Tensor = None
    ctx: torch.Tensor = None
    g1: torch.Tensor = None
    xn2: torch.Tensor = None
    rstd2: torch.Tensor = None
    qc: torch.Tensor = None
    qk: torch.Tensor = None
    z: torch.Tensor = None
    kv: torch.Tensor = None
    cctx: torch.Tensor = None
    clse: torch.Tensor = None
    g2: torch.Tensor = None
    xn3: torch.Tensor = None
    rstd3: torch.Tensor = None
    act: torch.Tensor = None


@dataclass
class ForwardState:
    """Everything the backward pass of one forward call needs."""
    N: int = 0
    S: int = 0
    T: int = 2
    p: float = 0.0
    seed: int = 0
    cross2: bool = False
    ids: torch.Tensor = None
    dec_ids: torch.Tensor = None
    enc_layers: List[_EncLayerSaved] = field(default_factory=list)
    dec_layers: List[_DecLayerSaved] = field(default_factory=list)
    h_last: torch.Tensor = None
    rstd_ef: torch.Tensor = None
    enc_out: torch.Tensor = None
    g_last: torch.Tensor = None
    rstd_df: torch.Tensor = None
    y: torch.Tensor = None
    bias_tbl: torch.Tensor = None
    bias2: torch.Tensor = None
    bucket_enc: torch.Tensor = None
    bucket_dec: torch.Tensor = None
    logits: torch.Tensor = None


# dropout site ids (per layer sites add 16*layer)
_SITE_EMB, _SITE_ENC_FINAL, _SITE_DEC_EMB, _SITE_DEC_FINAL = 1, 2, 3, 4
_SITE_ATT, _SITE_ATT_OUT, _SITE_FF_ACT, _SITE_FF_OUT, _SITE_CATT, _SITE_CATT_OUT = 5, 6, 7, 8, 9, 10


class T5Engine:
    """Runs T5 forward/backward for packed token batches on one GPU."""

    def __init__(self, dims: T5Dims, params: ParamStore):
        dims.validate()
        self.dims, self.params = dims, params
        self.device = params.device
        self._bucket_cache: Dict[int, torch.Tensor] = {}
        self.force_kv_projection = False      # tests: take the explicit K/V projection path of the cross-attention
        self.dw_stream = None                  # when set: weight-gradient products of backward() run on this stream
        self.ffn_bit_mask = os.environ.get("PACED_FFN_BITMASK", "1") != "0"   # encoder FFN backward masks by saved sign bits
        self.head_major = os.environ.get("PACED_HEAD_MAJOR", "1") != "0"   # encoder attention operands head-major (see forward)
        self._bucket_dec = torch.tensor([0, 1], dtype=torch.int32, device=self.device)   # offsets 0 and -1 (HF:244-249)
        dec_b = relative_position_bucket(torch.tensor([0, -1]), False, dims.num_buckets, dims.max_distance)
        assert dec_b.tolist() == [0, 1]

    # ------------------------------------------------------------------ helpers
    def _buckets(self, S: int) -> torch.Tensor:
        b = self._bucket_cache.get(S)
        if b is None:
            rel = torch.arange(-(S - 1), S, dtype=torch.long)            # key - query
            b = relative_position_bucket(rel, True, self.dims.num_buckets, self.dims.max_distance).to(torch.int32).to(self.device)
            self._bucket_cache[S] = b
        return b

    def _seed(self, st: ForwardState, site: int, layer: int = 0, dec: bool = False) -> int:
        return _mix(st.seed, site + 16 * layer + (4096 if dec else 0))

    def _cross2_ok(self, S: int) -> bool:
        """The projection-free cross-attention kernels need 2H <= 32 queries per sequence and d_model a multiple of 64 up
        to 1024; any sequence length (batches are padded to their longest sequence); other shapes take the K/V path."""
        dm = self.dims
        return (2 * dm.num_heads <= 32 and dm.d_model % 64 == 0 and dm.d_model <= 1024
                and dm.inner == dm.num_heads * 64 and not self.force_kv_projection)

    def _empty(self, rows, cols, dtype=BF16):
        return torch.empty(rows, cols, dtype=dtype, device=self.device)

    def _heads(self, nheads, rows):
        """Head-major bf16 matrix [nheads, rows, 64].  The weight-gradient products read such a matrix transposed, 64 rows
        per step: rows are allocated up to a multiple of 64 and the padding is zeroed."""
        alloc = (rows + 63) // 64 * 64
        buf = torch.empty(nheads, alloc, 64, dtype=BF16, device=self.device)
        if alloc != rows:
            buf[:, rows:].zero_()
        return buf[:, :rows]

    @staticmethod
    def _split_qkv(t, H):
        """q, k, v parts of a fused projection output: [M, 3*H*64] rows, or head-major [3H, M, 64]."""
        if t.dim() == 3:
            return t[:H], t[H:2 * H], t[2 * H:]
        I = H * 64
        return t[:, :I], t[:, I:2 * I], t[:, 2 * I:]

    def _dweight(self, dy, x, name, rows: Optional[slice] = None):
        """grad[name] (+)= dy^T x  -- dy [M, out], x [M, in] -> [out, in], fp32 atomics."""
        g = self.params.grad(name)
        if rows is not None:
            g = g[rows]
        side = self.dw_stream
        if side is None:
            K.gemm(dy, x, g, trans_a=True, trans_b=True, epilogue=K.EPI_ATOMIC_F32, splits=0)   # 0: library picks the split
            return
        # Nothing in the backward chain reads a weight gradient, so the product runs on a second stream and its
        # workgroups fill the CUs that the chain's kernel tails leave idle.  Both operands are complete once the
        # work enqueued on the calling stream so far has run; they are never written again, and record_stream keeps
        # their memory from being handed out again before the product has read it.
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            K.gemm(dy, x, g, trans_a=True, trans_b=True, epilogue=K.EPI_ATOMIC_F32, splits=0)
        dy.record_stream(side)
        x.record_stream(side)

    # ------------------------------------------------------------------ forward
    def forward(self, input_ids: torch.Tensor, labels: torch.Tensor, *, save: bool, drop_p: float = 0.0, seed: int = 0,
                want_logits: bool = True, after_encoder=None, gemm_grid: int = 0, gemm_grid_layers: int = 1000) -> ForwardState:
        """``after_encoder`` (optional callable) runs once the encoder's kernels are enqueued, before the decoder's
        small kernels: the place to start independent work on another stream.  ``gemm_grid`` > 0: the 256x256 products of
        this call (of its encoder layers below ``gemm_grid_layers``, its decoder and its LM head) are launched with at
        most that many workgroups, leaving the other compute units to kernels of other streams (a per-call option of the
        library: nothing else launched meanwhile is affected)."""
        dims, P = self.dims, self.params
        if input_ids.dim() != 2 or labels.dim() != 2 or labels.shape[1] != 2 or labels.shape[0] != input_ids.shape[0]:
            raise ValueError(f"expected input_ids [N,S] and labels [N,2], got {tuple(input_ids.shape)} and {tuple(labels.shape)}")
        N, S = input_ids.shape
        if S > 2048:
            raise ValueError(f"sequence length {S} exceeds the 2048 the attention kernels support")
        if not input_ids.is_cuda and input_ids.numel() and (int(input_ids.min()) < 0 or int(input_ids.max()) >= dims.vocab_size):
            raise ValueError("input_ids out of range for the vocabulary")   # device-resident ids are clamped by the kernel
        ids = input_ids.to(self.device, torch.int64).contiguous().view(-1)
        lab = labels.to(self.device, torch.int64)
        dec_ids = torch.zeros_like(lab)
        dec_ids[:, 1:] = lab[:, :-1]                                   # shift right, start token 0 (HF:618-637)
        dec_ids = dec_ids.masked_fill(dec_ids == -100, 0).contiguous().view(-1)
        d, I, ff, H = dims.d_model, dims.inner, dims.d_ff, dims.num_heads
        M, Md = N * S, N * 2
        dk = I // H
        st = ForwardState(N=N, S=S, p=float(drop_p), seed=int(seed), ids=ids, dec_ids=dec_ids)
        p = st.p
        cross2 = st.cross2 = self._cross2_ok(S)

        # ---- encoder
        st.bucket_enc = self._buckets(S)
        st.bias_tbl = torch.empty(H, 2 * S - 1, dtype=F32, device=self.device)
        K.relbias_table_fwd(P.w("enc.rel_bias"), st.bucket_enc, st.bias_tbl)
        h = self._empty(M, d)
        K.embedding_fwd(ids, P.w("shared"), h, drop_p=p, drop_seed=self._seed(st, _SITE_EMB))
        for l in range(dims.num_layers):
            pre = f"enc.{l}."
            sv = _EncLayerSaved()
            grid = gemm_grid if (gemm_grid > 0 and l < gemm_grid_layers) else None
            xn, rstd1 = self._empty(M, d), torch.empty(M, dtype=F32, device=self.device)
            K.rmsnorm_fwd(h, P.w(pre + "ln1"), xn, rstd1, eps=dims.eps)
            # The attention's operands are head-major ([heads, M, 64]; the products write and read that layout): a
            # (sequence, head) is one contiguous block for the attention kernels instead of 128-byte pieces at a row stride.
            if self.head_major:
                qkv, ctx = self._heads(3 * H, M), self._heads(H, M)
            else:
                qkv, ctx = self._empty(M, 3 * I), self._empty(M, I)
            K.gemm(xn, P.w(pre + "qkv"), qkv, grid_limit=grid)
            q_, k_, v_ = self._split_qkv(qkv, H)
            lse = torch.empty(N * H * S, dtype=F32, device=self.device)
            K.enc_attn_fwd(q_, k_, v_, ctx, lse, st.bias_tbl, N, S, H,
                           drop_p=p, drop_seed=self._seed(st, _SITE_ATT, l))
            h_mid = self._empty(M, d)
            K.gemm(ctx, P.w(pre + "o"), h_mid, epilogue=K.EPI_RES_DROP, aux=h, drop_p=p, drop_seed=self._seed(st, _SITE_ATT_OUT, l), grid_limit=grid)
            xn2, rstd2 = self._empty(M, d), torch.empty(M, dtype=F32, device=self.device)
            K.rmsnorm_fwd(h_mid, P.w(pre + "ln2"), xn2, rstd2, eps=dims.eps)
            act = self._empty(M, ff)
            # the backward only needs to know which activations survived relu and dropout: one bit each
            actbits = K.sign_bits_buffer(M, ff, self.device) if save and self.ffn_bit_mask and ff % 8 == 0 else None
            K.gemm(xn2, P.w(pre + "wi"), act, epilogue=K.EPI_RELU_DROP, aux=actbits, drop_p=p, drop_seed=self._seed(st, _SITE_FF_ACT, l), grid_limit=grid)
            h_out = self._empty(M, d)
            K.gemm(act, P.w(pre + "wo"), h_out, epilogue=K.EPI_RES_DROP, aux=h_mid, drop_p=p, drop_seed=self._seed(st, _SITE_FF_OUT, l), grid_limit=grid)
            if save:
                sv.h_in, sv.xn, sv.rstd1, sv.qkv, sv.ctx, sv.lse = h, xn, rstd1, qkv, ctx, lse
                sv.h_mid, sv.xn2, sv.rstd2, sv.act, sv.actbits = h_mid, xn2, rstd2, act, actbits
                st.enc_layers.append(sv)
            h = h_out
        enc = self._empty(M, d)
        rstd_ef = torch.empty(M, dtype=F32, device=self.device)
        K.rmsnorm_fwd(h, P.w("enc.final_ln"), enc, rstd_ef, eps=dims.eps, drop_p=p, drop_seed=self._seed(st, _SITE_ENC_FINAL))
        st.enc_out = enc
        if after_encoder is not None:
            after_encoder()
        if save:
            st.h_last, st.rstd_ef = h, rstd_ef

        # ---- decoder (2 target positions)
        st.bias2 = torch.empty(H, 2, dtype=F32, device=self.device)
        K.relbias_table_fwd(P.w("dec.rel_bias"), self._bucket_dec, st.bias2)
        g = self._empty(Md, d)
        K.embedding_fwd(dec_ids, P.w("shared"), g, drop_p=p, drop_seed=self._seed(st, _SITE_DEC_EMB))
        for l in range(dims.num_decoder_layers):
            pre = f"dec.{l}."
            sv = _DecLayerSaved()
            xn, rstd1 = self._empty(Md, d), torch.empty(Md, dtype=F32, device=self.device)
            K.rmsnorm_fwd(g, P.w(pre + "ln1"), xn, rstd1, eps=dims.eps)
            qkv = self._empty(Md, 3 * I)
            K.gemm(xn, P.w(pre + "qkv"), qkv)
            ctx = self._empty(Md, I)
            K.dec_self_attn_fwd(qkv, ctx, st.bias2, N, H, drop_p=p, drop_seed=self._seed(st, _SITE_ATT, l, True))
            g1 = self._empty(Md, d)
            K.gemm(ctx, P.w(pre + "o"), g1, epilogue=K.EPI_RES_DROP, aux=g, drop_p=p, drop_seed=self._seed(st, _SITE_ATT_OUT, l, True))
            xn2, rstd2 = self._empty(Md, d), torch.empty(Md, dtype=F32, device=self.device)
            K.rmsnorm_fwd(g1, P.w(pre + "ln2"), xn2, rstd2, eps=dims.eps)
            qc = self._empty(Md, I)
            K.gemm(xn2, P.w(pre + "cq"), qc)
            cctx = self._empty(Md, I)
            clse = torch.empty(N * H * 2, dtype=F32, device=self.device)
            kv = qk = z = None
            if cross2:
                # scores = (q_h Wk_h) . enc and ctx_h = (p . enc) Wv_h^T: the encoder states are never projected
                Wckv = P.w(pre + "ckv")
                qk, z = self._empty(Md, H * d), self._empty(Md, H * d)
                K.gemm_batched(qc, Wckv[:I], qk, M=Md, N=d, K=dk, lda=I, ldb=d, ldc=H * d, batch=H,
                               stride_a=dk, stride_b=dk * d, stride_c=d, trans_b=True)
                K.cross2_fwd(enc, qk, z, clse, N, S, H, drop_p=p, drop_seed=self._seed(st, _SITE_CATT, l, True))
                K.gemm_batched(z, Wckv[I:], cctx, M=Md, N=dk, K=d, lda=H * d, ldb=d, ldc=I, batch=H,
                               stride_a=d, stride_b=dk * d, stride_c=dk)
            else:
                kv = self._empty(M, 2 * I)
                K.gemm(enc, P.w(pre + "ckv"), kv)
                K.cross_attn_fwd(qc, kv, cctx, clse, N, S, H, drop_p=p, drop_seed=self._seed(st, _SITE_CATT, l, True))
            g2 = self._empty(Md, d)
            K.gemm(cctx, P.w(pre + "co"), g2, epilogue=K.EPI_RES_DROP, aux=g1, drop_p=p, drop_seed=self._seed(st, _SITE_CATT_OUT, l, True))
            xn3, rstd3 = self._empty(Md, d), torch.empty(Md, dtype=F32, device=self.device)
            K.rmsnorm_fwd(g2, P.w(pre + "ln3"), xn3, rstd3, eps=dims.eps)
            act = self._empty(Md, ff)
            K.gemm(xn3, P.w(pre + "wi"), act, epilogue=K.EPI_RELU_DROP, drop_p=p, drop_seed=self._seed(st, _SITE_FF_ACT, l, True))
            g3 = self._empty(Md, d)
            K.gemm(act, P.w(pre + "wo"), g3, epilogue=K.EPI_RES_DROP, aux=g2, drop_p=p, drop_seed=self._seed(st, _SITE_FF_OUT, l, True))
            if save:
                sv.g_in, sv.xn, sv.rstd1, sv.qkv, sv.ctx, sv.g1 = g, xn, rstd1, qkv, ctx, g1
                sv.xn2, sv.rstd2, sv.qc, sv.kv, sv.cctx, sv.clse, sv.g2 = xn2, rstd2, qc, kv, cctx, clse, g2
                sv.qk, sv.z = qk, z
                sv.xn3, sv.rstd3, sv.act = xn3, rstd3, act
                st.dec_layers.append(sv)
            g = g3
        # final norm, dropout, tied-head rescale d^-0.5 (HF:1044-1045) folded into the norm output
        y = self._empty(Md, d)
        rstd_df = torch.empty(Md, dtype=F32, device=self.device)
        K.rmsnorm_fwd(g, P.w("dec.final_ln"), y, rstd_df, eps=dims.eps, out_scale=d ** -0.5,
                      drop_p=p, drop_seed=self._seed(st, _SITE_DEC_FINAL))
        st.y = y
        if save:
            st.g_last, st.rstd_df = g, rstd_df
        if want_logits:
            st.logits = torch.empty(Md, dims.vocab_size, dtype=F32, device=self.device)
            K.gemm(y, P.w("shared"), st.logits, epilogue=K.EPI_F32, grid_limit=gemm_grid if gemm_grid > 0 else None)
        if not save:
            st.enc_out = None
        return st

    # ------------------------------------------------------------------ backward
    def backward(self, st: ForwardState, dlogits: torch.Tensor, on_segments_done=None) -> None:
        """Accumulate parameter gradients of one forward call into ``params.g32`` given d loss / d logits
        (bf16 [N*2, V]).  The forward must have been run with ``save=True``.

        ``on_segments_done(first, last)``, if given, is called when every kernel that writes the gradients of the
        layout segments first .. last, or reads their weights, has been enqueued (on the calling stream and on
        ``dw_stream``): once after the decoder stack (all of its segments) and once after each encoder layer.  The
        tied embedding, the encoder's relative bias and its final norm are complete only when backward() returns."""
        dims, P = self.dims, self.params
        if not st.enc_layers or st.g_last is None:
            raise RuntimeError("backward needs a forward state produced with save=True")
        N, S, p = st.N, st.S, st.p
        d, I, ff, H = dims.d_model, dims.inner, dims.d_ff, dims.num_heads
        M, Md = N * S, N * 2
        scale = 1.0 / (1.0 - p) if p > 0 else 1.0
        if tuple(dlogits.shape) != (Md, dims.vocab_size):
            raise ValueError("dlogits has the wrong shape")

        # ---- LM head (tied with the embedding)
        self._dweight(dlogits, st.y, "shared")
        dy = self._empty(Md, d)
        K.gemm(dlogits, P.w("shared"), dy, trans_b=True)

        # ---- decoder
        nd = dims.num_decoder_layers

        def masked_copy_needed():
            return p > 0

        dg = self._empty(Md, d)
        dz = self._empty(Md, d) if masked_copy_needed() else None
        K.rmsnorm_bwd(dy, st.g_last, P.w("dec.final_ln"), st.rstd_df, dg, dx2=dz, dw=P.grad("dec.final_ln"),
                      out_scale=d ** -0.5, drop_p=p, drop_seed=self._seed(st, _SITE_DEC_FINAL),
                      drop2_p=p, drop2_seed=self._seed(st, _SITE_FF_OUT, nd - 1, True))
        d_enc = None
        xW = xU = None
        dk = I // H
        for l in reversed(range(nd)):
            pre = f"dec.{l}."
            sv = st.dec_layers[l]
            # feed-forward sub-layer
            z = dz if dz is not None else dg
            self._dweight(z, sv.act, pre + "wo")
            dact = self._empty(Md, ff)
            K.gemm(z, P.w(pre + "wo"), dact, trans_b=True, epilogue=K.EPI_MASK, aux=sv.act, mask_scale=scale)
            self._dweight(dact, sv.xn3, pre + "wi")
            dxn = self._empty(Md, d)
            K.gemm(dact, P.w(pre + "wi"), dxn, trans_b=True)
            dg2 = self._empty(Md, d)
            dz = self._empty(Md, d) if masked_copy_needed() else None
            K.rmsnorm_bwd(dxn, sv.g2, P.w(pre + "ln3"), sv.rstd3, dg2, dres=dg, dx2=dz, dw=P.grad(pre + "ln3"),
                          drop2_p=p, drop2_seed=self._seed(st, _SITE_CATT_OUT, l, True))
            # cross-attention sub-layer
            z = dz if dz is not None else dg2
            self._dweight(z, sv.cctx, pre + "co")
            dcctx = self._empty(Md, I)
            K.gemm(z, P.w(pre + "co"), dcctx, trans_b=True)
            dqc = self._empty(Md, I)
            if st.cross2:
                Wckv, gckv = P.w(pre + "ckv"), P.grad(pre + "ckv")
                C2 = 2 * H
                if xW is None:      # factors of d enc over all decoder layers: [N][nd*2C][S8] and [N][nd*2C][d]
                    S8 = (S + 7) // 8 * 8       # row stride the token-contraction product below can read; S columns are used
                    xW = torch.empty(N, nd * 2 * C2, S8, dtype=BF16, device=self.device)
                    xU = torch.empty(N, nd * 2 * C2, d, dtype=BF16, device=self.device)
                # dWv_h += dcctx_h^T z_h ;  dz_h = dcctx_h Wv_h
                K.gemm_batched(dcctx, sv.z, gckv[I:], M=dk, N=d, K=Md, lda=I, ldb=H * d, ldc=d, batch=H,
                               stride_a=dk, stride_b=d, stride_c=dk * d, trans_a=True, trans_b=True, epilogue=K.EPI_ATOMIC_F32, splits=0)
                dzz, dqk = self._empty(Md, H * d), self._empty(Md, H * d)
                K.gemm_batched(dcctx, Wckv[I:], dzz, M=Md, N=d, K=dk, lda=I, ldb=d, ldc=H * d, batch=H,
                               stride_a=dk, stride_b=dk * d, stride_c=d, trans_b=True)
                K.cross2_bwd(st.enc_out, sv.qk, dzz, sv.z, sv.clse, dqk, xW[0, l * 2 * C2:], nd * 2 * C2 * S8,
                             xU[0, l * 2 * C2:], nd * 2 * C2 * d, N, S, H, wld=S8, drop_p=p, drop_seed=self._seed(st, _SITE_CATT, l, True))
                # dq_h = dqk_h Wk_h^T ;  dWk_h += q_h^T dqk_h
                K.gemm_batched(dqk, Wckv[:I], dqc, M=Md, N=dk, K=d, lda=H * d, ldb=d, ldc=I, batch=H,
                               stride_a=d, stride_b=dk * d, stride_c=dk)
                K.gemm_batched(sv.qc, dqk, gckv[:I], M=dk, N=d, K=Md, lda=I, ldb=H * d, ldc=d, batch=H,
                               stride_a=dk, stride_b=d, stride_c=dk * d, trans_a=True, trans_b=True, epilogue=K.EPI_ATOMIC_F32, splits=0)
            else:
                dkv = self._empty(M, 2 * I)
                K.cross_attn_bwd(sv.qc, sv.kv, sv.cctx, dcctx, sv.clse, dqc, dkv, N, S, H,
                                 drop_p=p, drop_seed=self._seed(st, _SITE_CATT, l, True))
                self._dweight(dkv, st.enc_out, pre + "ckv")
                if d_enc is None:
                    d_enc = self._empty(M, d)
                    K.gemm(dkv, P.w(pre + "ckv"), d_enc, trans_b=True)
                else:
                    K.gemm(dkv, P.w(pre + "ckv"), d_enc, trans_b=True, epilogue=K.EPI_RES_DROP, aux=d_enc)
            self._dweight(dqc, sv.xn2, pre + "cq")
            K.gemm(dqc, P.w(pre + "cq"), dxn, trans_b=True)
            dg1 = self._empty(Md, d)
            dz = self._empty(Md, d) if masked_copy_needed() else None
            K.rmsnorm_bwd(dxn, sv.g1, P.w(pre + "ln2"), sv.rstd2, dg1, dres=dg2, dx2=dz, dw=P.grad(pre + "ln2"),
                          drop2_p=p, drop2_seed=self._seed(st, _SITE_ATT_OUT, l, True))
            # self-attention sub-layer
            z = dz if dz is not None else dg1
            self._dweight(z, sv.ctx, pre + "o")
            dctx = self._empty(Md, I)
            K.gemm(z, P.w(pre + "o"), dctx, trans_b=True)
            dqkv = self._empty(Md, 3 * I)
            dbias2 = torch.zeros(H, 2, dtype=F32, device=self.device)
            K.dec_self_attn_bwd(sv.qkv, dctx, dqkv, st.bias2, dbias2, N, H, drop_p=p, drop_seed=self._seed(st, _SITE_ATT, l, True))
            K.relbias_table_bwd(dbias2, self._bucket_dec, P.grad("dec.rel_bias"))
            self._dweight(dqkv, sv.xn, pre + "qkv")
            K.gemm(dqkv, P.w(pre + "qkv"), dxn, trans_b=True)
            dg0 = self._empty(Md, d)
            need2 = masked_copy_needed() and l > 0
            dz = self._empty(Md, d) if need2 else None
            K.rmsnorm_bwd(dxn, sv.g_in, P.w(pre + "ln1"), sv.rstd1, dg0, dres=dg1, dx2=dz, dw=P.grad(pre + "ln1"),
                          drop2_p=p if need2 else 0.0, drop2_seed=self._seed(st, _SITE_FF_OUT, l - 1, True) if need2 else 0)
            dg = dg0
            sv.__dict__.clear()            # release this layer's activations
        K.embedding_bwd(st.dec_ids, dg, P.grad("shared"), drop_p=p, drop_seed=self._seed(st, _SITE_DEC_EMB))
        if on_segments_done is not None:
            on_segments_done("dec.0.ln1", "dec.final_ln")
        if st.cross2:
            # d enc[n] = sum over layers and projected queries of ds (x) qk + dropout(p) (x) dz: one token-contraction product
            Kc = nd * 4 * H
            d_enc = self._empty(M, d)
            K.gemm_batched(xW, xU, d_enc, M=S, N=d, K=Kc, lda=S8, ldb=d, ldc=d, batch=N,
                           stride_a=Kc * S8, stride_b=Kc * d, stride_c=S * d, trans_a=True, trans_b=True)
            del xW, xU

        # ---- encoder
        ne = dims.num_layers
        dbias_tbl = torch.zeros(64, H, 2 * S - 1, dtype=F32, device=self.device)   # 64 replicas, summed at the end
        dh = self._empty(M, d)
        dz = self._empty(M, d) if masked_copy_needed() else None
        K.rmsnorm_bwd(d_enc, st.h_last, P.w("enc.final_ln"), st.rstd_ef, dh, dx2=dz, dw=P.grad("enc.final_ln"),
                      drop_p=p, drop_seed=self._seed(st, _SITE_ENC_FINAL),
                      drop2_p=p, drop2_seed=self._seed(st, _SITE_FF_OUT, ne - 1))
        for l in reversed(range(ne)):
            pre = f"enc.{l}."
            sv = st.enc_layers[l]
            z = dz if dz is not None else dh
            self._dweight(z, sv.act, pre + "wo")
            dact = self._empty(M, ff)
            if sv.actbits is not None:
                K.gemm(z, P.w(pre + "wo"), dact, trans_b=True, epilogue=K.EPI_MASK_BITS, aux=sv.actbits, mask_scale=scale)
            else:
                K.gemm(z, P.w(pre + "wo"), dact, trans_b=True, epilogue=K.EPI_MASK, aux=sv.act, mask_scale=scale)
            self._dweight(dact, sv.xn2, pre + "wi")
            dxn = self._empty(M, d)
            K.gemm(dact, P.w(pre + "wi"), dxn, trans_b=True)
            del dact
            dh_mid = self._empty(M, d)
            dz = self._empty(M, d) if masked_copy_needed() else None
            K.rmsnorm_bwd(dxn, sv.h_mid, P.w(pre + "ln2"), sv.rstd2, dh_mid, dres=dh, dx2=dz, dw=P.grad(pre + "ln2"),
                          drop2_p=p, drop2_seed=self._seed(st, _SITE_ATT_OUT, l))
            z = dz if dz is not None else dh_mid
            self._dweight(z, sv.ctx, pre + "o")
            if sv.ctx.dim() == 3:       # head-major, like the saved context and q, k, v
                dctx, dqkv = self._heads(H, M), self._heads(3 * H, M)
            else:
                dctx, dqkv = self._empty(M, I), self._empty(M, 3 * I)
            K.gemm(z, P.w(pre + "o"), dctx, trans_b=True)
            q_, k_, v_ = self._split_qkv(sv.qkv, H)
            dq_, dk_, dv_ = self._split_qkv(dqkv, H)
            K.enc_attn_bwd(q_, k_, v_, sv.ctx, dctx, sv.lse, st.bias_tbl, dq_, dk_, dv_, dbias_tbl, N, S, H,
                           drop_p=p, drop_seed=self._seed(st, _SITE_ATT, l))
            self._dweight(dqkv, sv.xn, pre + "qkv")
            K.gemm(dqkv, P.w(pre + "qkv"), dxn, trans_b=True)
            del dqkv, dctx
            dh_in = self._empty(M, d)
            need2 = masked_copy_needed() and l > 0
            dz = self._empty(M, d) if need2 else None
            K.rmsnorm_bwd(dxn, sv.h_in, P.w(pre + "ln1"), sv.rstd1, dh_in, dres=dh_mid, dx2=dz, dw=P.grad(pre + "ln1"),
                          drop2_p=p if need2 else 0.0, drop2_seed=self._seed(st, _SITE_FF_OUT, l - 1) if need2 else 0)
            dh = dh_in
            sv.__dict__.clear()
            if on_segments_done is not None:
                on_segments_done(pre + "ln1", pre + "wo")
        K.embedding_bwd(st.ids, dh, P.grad("shared"), drop_p=p, drop_seed=self._seed(st, _SITE_EMB))
        K.relbias_table_bwd(dbias_tbl, st.bucket_enc, P.grad("enc.rel_bias"))
        if self.dw_stream is not None:
            torch.cuda.current_stream().wait_stream(self.dw_stream)
        st.enc_layers.clear()
        st.dec_layers.clear()

"""Thin torch-tensor wrappers over the C ABI (one function per entry point).

torch supplies device memory and the current HIP stream; every computation happens in
libpaced_hip.so.  Shapes are validated here so that a kernel is never launched on operands that do
not match what its grid assumes.
"""
from __future__ import annotations

import os

import torch

from . import _lib

EPI_BF16, EPI_RES_DROP, EPI_RELU_DROP, EPI_MASK, EPI_F32, EPI_ATOMIC_F32, EPI_MASK_BITS = range(7)
BF16, F32, I64, I32, U8 = torch.bfloat16, torch.float32, torch.int64, torch.int32, torch.uint8

# When set to a list, every gemm() launch is bracketed by HIP events on the launch stream and
# ((trans_a, trans_b, epilogue, kernel), algorithmic FLOPs, start, end) is appended (bench.py's roofline leg);
# kernel is paced_gemm_last_kernel(): 256 = phased kernel, 128 / 64 = small tiles, +1 = their ring variant.
PROFILE = None


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _chk(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise ValueError(f"{name} must be a device tensor")
    if t.dtype != dtype:
        raise ValueError(f"{name} must be {dtype}, got {t.dtype}")


def _rows2d(t, name):
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name} must be 2-D with unit inner stride, got shape {tuple(t.shape)} strides {t.stride()}")
    return t.stride(0)


def _operand(t, name):
    """(rows, features, leading dimension, head stride) of a GEMM operand: a matrix [rows, features], or head-major
    [features/64, rows, 64] (contiguous [rows, 64] members; head stride 0 means "not head-major")."""
    if t.dim() == 3:
        if t.shape[2] != 64 or t.stride(2) != 1 or t.stride(1) != 64:
            raise ValueError(f"{name}: a head-major operand is [features/64, rows, 64] with contiguous [rows, 64] members, got {tuple(t.shape)} strides {t.stride()}")
        return t.shape[1], t.shape[0] * 64, 64, t.stride(0)
    return t.shape[0], t.shape[1], _rows2d(t, name), 0


def _gemm_opts(grid_limit, tile, ring):
    """paced_gemm_opts for one call, or None when every option is left to the process default."""
    if grid_limit is None and tile is None and ring is None and PROFILE is None:
        return None
    return _lib.GemmOpts(-1 if grid_limit is None else int(grid_limit), -1 if tile is None else int(tile),
                         -1 if ring is None else int(bool(ring)), 0)


def gemm(a, b, c, *, trans_a=False, trans_b=False, epilogue=EPI_BF16, aux=None, alpha=1.0, mask_scale=1.0,
         drop_p=0.0, drop_seed=0, splits=1, grid_limit=None, tile=None, ring=None):
    """c = alpha * op(a) op(b) with a fused epilogue (see include/paced_hip.h).

    aux is the bf16 residual / mask operand of EPI_RES_DROP / EPI_MASK, or a uint8 [M, N/8] bit matrix
    (bit n%8 of byte n//8 of row m <-> element (m, n)): written by EPI_RELU_DROP with the sign of its
    output, read by EPI_MASK_BITS; a 1-D uint8 aux is a tile-ordered bit matrix from sign_bits_buffer.
    a, c and a transposed b may be head-major 3-D tensors (see _operand).
    grid_limit / tile / ring: launch options of THIS call (``paced_gemm_opts``; None = the process default).
    """
    lib = _lib.load()
    bits = epilogue in (EPI_RELU_DROP, EPI_MASK_BITS)
    _chk(a, BF16, "a"); _chk(b, BF16, "b"); _chk(aux, U8 if bits else BF16, "aux")
    _chk(c, F32 if epilogue in (EPI_F32, EPI_ATOMIC_F32) else BF16, "c")
    ar, af, lda, ah = _operand(a, "a")
    br, bf_, ldb, bh = _operand(b, "b")
    M, N, ldc, ch = _operand(c, "c")
    K = ar if trans_a else af
    if (af if trans_a else ar) != M:
        raise ValueError(f"a {tuple(a.shape)} does not give M={M} (trans_a={trans_a})")
    if (bf_ if trans_b else br) != N or (br if trans_b else bf_) != K:
        raise ValueError(f"b {tuple(b.shape)} does not give N={N}, K={K} (trans_b={trans_b})")
    ldaux = 0
    if aux is not None and bits and aux.dim() == 1:
        need = lib.paced_gemm_tile_bits_bytes(M, N)
        if need == 0 or aux.numel() < need or not aux.is_contiguous():
            raise ValueError(f"a tile-ordered bit matrix for [{M}, {N}] needs {need} contiguous bytes (0 = shape not eligible), got {aux.numel()}")
        ldaux = -1
    elif aux is not None:
        if bits and (N % 8 or tuple(aux.shape) != (M, N // 8)):
            raise ValueError(f"the bit matrix must be [M, N/8] = [{M}, {N}/8], got {tuple(aux.shape)}")
        if not bits and tuple(aux.shape) != (M, N):
            raise ValueError("aux must have the shape of c")
        ldaux = _rows2d(aux, "aux")
    prof = PROFILE
    opts = _gemm_opts(grid_limit, tile, ring)
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(lib.paced_gemm_bf16_ex(_p(a), _p(b), _p(c), _p(aux), M, N, K, lda, ldb, ldc, ldaux,
                                      int(trans_a), int(trans_b), epilogue, alpha, mask_scale, drop_p, drop_seed, splits,
                                      ah, bh, ch, 1, 0, 0, 0, opts, _stream()), "paced_gemm_bf16_ex")
    if prof is not None:
        e1.record()
        prof.append(((int(trans_a), int(trans_b), int(epilogue), opts.kernel), 2.0 * M * N * K, e0, e1))
    return c


def sign_bits_buffer(M, N, device):
    """Buffer for the sign bits EPI_RELU_DROP writes and EPI_MASK_BITS reads for an [M, N] output: 1-D (tile order) when
    the product takes the 256x256 kernel, else the row-major [M, N/8] matrix (PACED_TILE_BITS=0: always row-major)."""
    need = _lib.load().paced_gemm_tile_bits_bytes(M, N)
    if need and os.environ.get("PACED_TILE_BITS", "1") != "0":
        return torch.empty(need, dtype=torch.uint8, device=device)
    return torch.empty(M, N // 8, dtype=torch.uint8, device=device)


def gemm_batched(a, b, c, *, M, N, K, lda, ldb, ldc, batch, stride_a, stride_b, stride_c, trans_a=False, trans_b=False,
                 epilogue=EPI_BF16, alpha=1.0, splits=1, grid_limit=None, tile=None, ring=None):
    """``batch`` products of one shape in one launch; member i reads a.data + i*stride_a, b.data + i*stride_b and writes
    c.data + i*stride_c (strides and leading dimensions in elements).  The tensors only provide base pointers and dtypes."""
    lib = _lib.load()
    _chk(a, BF16, "a"); _chk(b, BF16, "b")
    _chk(c, F32 if epilogue in (EPI_F32, EPI_ATOMIC_F32) else BF16, "c")
    prof = PROFILE
    opts = _gemm_opts(grid_limit, tile, ring)
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(lib.paced_gemm_bf16_ex(_p(a), _p(b), _p(c), 0, M, N, K, lda, ldb, ldc, 0, int(trans_a), int(trans_b), epilogue,
                                      alpha, 1.0, 0.0, 0, splits, 0, 0, 0, batch, stride_a, stride_b, stride_c, opts, _stream()),
               "paced_gemm_bf16_ex (batched)")
    if prof is not None:
        e1.record()
        prof.append(((int(trans_a), int(trans_b), int(epilogue), opts.kernel), 2.0 * M * N * K * batch, e0, e1))
    return c


def gemm_set_ring(on: bool):
    """Ring variant of the small-tile kernels on (default) or off (benchmarks and tests)."""
    _lib.check(_lib.load().paced_gemm_set_ring(int(bool(on))), "paced_gemm_set_ring")


def gemm_set_grid(workgroups: int):
    """Process default: at most that many workgroups per 256x256 phased launch from now on (0 = one per compute unit).
    For benchmarks and tests; product code passes ``grid_limit=`` per call."""
    _lib.check(_lib.load().paced_gemm_set_grid(int(workgroups)), "paced_gemm_set_grid")


def gemm_get_grid() -> int:
    return int(_lib.load().paced_gemm_get_grid())


def gemm_set_tile(tile: int):
    """0 = heuristic, 128 or 256 = force that tile configuration (benchmarks and tests)."""
    _lib.check(_lib.load().paced_gemm_set_tile(tile), "paced_gemm_set_tile")


def rmsnorm_fwd(x, w, y, rstd=None, *, eps=1e-6, out_scale=1.0, drop_p=0.0, drop_seed=0):
    lib = _lib.load()
    _chk(x, BF16, "x"); _chk(w, BF16, "w"); _chk(y, BF16, "y"); _chk(rstd, F32, "rstd")
    M, d = x.shape
    assert x.is_contiguous() and y.is_contiguous() and y.shape == x.shape and w.numel() == d
    assert rstd is None or rstd.numel() >= M
    _lib.check(lib.paced_rmsnorm_fwd(_p(x), _p(w), _p(y), _p(rstd), M, d, eps, out_scale, drop_p, drop_seed, _stream()), "paced_rmsnorm_fwd")
    return y


def rmsnorm_bwd(dy, x, w, rstd, dx, *, dres=None, dx2=None, dw=None, out_scale=1.0, drop_p=0.0, drop_seed=0,
                drop2_p=0.0, drop2_seed=0):
    lib = _lib.load()
    for t, n in ((dy, "dy"), (x, "x"), (w, "w"), (dres, "dres"), (dx, "dx"), (dx2, "dx2")):
        _chk(t, BF16, n)
    _chk(rstd, F32, "rstd"); _chk(dw, F32, "dw")
    M, d = x.shape
    for t in (dy, x, dx, dres, dx2):
        assert t is None or (t.is_contiguous() and tuple(t.shape) == (M, d))
    assert dw is None or dw.numel() == d
    _lib.check(lib.paced_rmsnorm_bwd(_p(dy), _p(x), _p(w), _p(rstd), _p(dres), _p(dx), _p(dx2), _p(dw), M, d, out_scale,
                                     drop_p, drop_seed, drop2_p, drop2_seed, _stream()), "paced_rmsnorm_bwd")
    return dx


def embedding_fwd(ids, table, out, *, drop_p=0.0, drop_seed=0):
    lib = _lib.load()
    _chk(ids, I64, "ids"); _chk(table, BF16, "table"); _chk(out, BF16, "out")
    M = ids.numel()
    V, d = table.shape
    assert ids.is_contiguous() and table.is_contiguous() and out.is_contiguous() and tuple(out.shape) == (M, d)
    _lib.check(lib.paced_embedding_fwd(_p(ids), _p(table), _p(out), M, d, V, drop_p, drop_seed, _stream()), "paced_embedding_fwd")
    return out


def embedding_bwd(ids, dout, dtable, *, drop_p=0.0, drop_seed=0):
    lib = _lib.load()
    _chk(ids, I64, "ids"); _chk(dout, BF16, "dout"); _chk(dtable, F32, "dtable")
    M = ids.numel()
    V, d = dtable.shape
    assert ids.is_contiguous() and dout.is_contiguous() and dtable.is_contiguous() and tuple(dout.shape) == (M, d)
    _lib.check(lib.paced_embedding_bwd(_p(ids), _p(dout), _p(dtable), M, d, V, drop_p, drop_seed, _stream()), "paced_embedding_bwd")


def dropout_apply(x, y, *, drop_p, drop_seed):
    lib = _lib.load()
    _chk(x, BF16, "x"); _chk(y, BF16, "y")
    assert x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    _lib.check(lib.paced_dropout_apply(_p(x), _p(y), x.numel(), drop_p, drop_seed, _stream()), "paced_dropout_apply")
    return y


def dropout_mask(n, *, drop_p, drop_seed, device="cuda"):
    lib = _lib.load()
    out = torch.empty(n, dtype=F32, device=device)
    _lib.check(lib.paced_dropout_mask(_p(out), n, drop_p, drop_seed, _stream()), "paced_dropout_mask")
    return out


def cast_f32_bf16(x, y):
    lib = _lib.load()
    _chk(x, F32, "x"); _chk(y, BF16, "y")
    assert x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    _lib.check(lib.paced_cast_f32_bf16(_p(x), _p(y), x.numel(), _stream()), "paced_cast_f32_bf16")
    return y


def relbias_table_fwd(rel_bias, bucket, tbl):
    lib = _lib.load()
    _chk(rel_bias, BF16, "rel_bias"); _chk(bucket, I32, "bucket"); _chk(tbl, F32, "tbl")
    H, L = tbl.shape
    assert rel_bias.is_contiguous() and rel_bias.shape[1] == H and bucket.numel() == L and tbl.is_contiguous()
    _lib.check(lib.paced_relbias_table_fwd(_p(rel_bias), _p(bucket), _p(tbl), H, L, _stream()), "paced_relbias_table_fwd")
    return tbl


def relbias_table_bwd(dtbl, bucket, drel_bias):
    """dtbl is [H, L] or [replicas, H, L]; the replicas are summed."""
    lib = _lib.load()
    _chk(dtbl, F32, "dtbl"); _chk(bucket, I32, "bucket"); _chk(drel_bias, F32, "drel_bias")
    reps = 1 if dtbl.dim() == 2 else dtbl.shape[0]
    H, L = dtbl.shape[-2:]
    assert drel_bias.is_contiguous() and drel_bias.shape[1] == H and bucket.numel() == L and dtbl.is_contiguous()
    _lib.check(lib.paced_relbias_table_bwd(_p(dtbl), reps, _p(bucket), _p(drel_bias), H, L, _stream()), "paced_relbias_table_bwd")


def _head_layout(ts, names, N, S, H):
    """(row stride, head stride) shared by the given matrices: [N*S, H*64] (or column slices of one) with the heads side
    by side, or head-major [H, N*S, 64] (what gemm writes into / reads from 3-D operands)."""
    t0 = ts[0]
    for t, n in zip(ts, names):
        _chk(t, BF16, n)
    if t0.dim() == 3:
        for t in ts:
            assert tuple(t.shape) == (H, N * S, 64) and t.stride(2) == 1 and t.stride(1) == 64 and t.stride(0) == t0.stride(0)
        return 64, t0.stride(0)
    ld = _rows2d(t0, names[0])
    for t, n in zip(ts, names):
        assert tuple(t.shape) == (N * S, H * 64) and _rows2d(t, n) == ld
    return ld, 64


def enc_attn_fwd(q, k, v, ctx, lse, bias_tbl, N, S, H, *, drop_p=0.0, drop_seed=0):
    lib = _lib.load()
    _chk(lse, F32, "lse"); _chk(bias_tbl, F32, "bias_tbl")
    ld, hs = _head_layout((q, k, v), ("q", "k", "v"), N, S, H)
    ldo, hso = _head_layout((ctx,), ("ctx",), N, S, H)
    assert lse.numel() == N * H * S and lse.is_contiguous()
    assert bias_tbl is None or (tuple(bias_tbl.shape) == (H, 2 * S - 1) and bias_tbl.is_contiguous())
    _lib.check(lib.paced_enc_attn_fwd(_p(q), _p(k), _p(v), ld, hs, _p(ctx), ldo, hso, _p(lse), _p(bias_tbl),
                                      N, S, H, 64, drop_p, drop_seed, _stream()), "paced_enc_attn_fwd")
    return ctx


def enc_attn_bwd(q, k, v, ctx, dctx, lse, bias_tbl, dq, dk, dv, dbias_tbl, N, S, H, *, drop_p=0.0, drop_seed=0):
    lib = _lib.load()
    _chk(lse, F32, "lse"); _chk(bias_tbl, F32, "bias_tbl"); _chk(dbias_tbl, F32, "dbias_tbl")
    ld, hs = _head_layout((q, k, v), ("q", "k", "v"), N, S, H)
    ldo, hso = _head_layout((ctx,), ("ctx",), N, S, H)
    lddo, hsdo = _head_layout((dctx,), ("dctx",), N, S, H)
    ldd, hsd = _head_layout((dq, dk, dv), ("dq", "dk", "dv"), N, S, H)
    assert lse.numel() == N * H * S and lse.is_contiguous()
    assert bias_tbl is None or (tuple(bias_tbl.shape) == (H, 2 * S - 1) and bias_tbl.is_contiguous())
    reps = 1
    if dbias_tbl is not None:
        assert dbias_tbl.is_contiguous() and tuple(dbias_tbl.shape[-2:]) == (H, 2 * S - 1) and dbias_tbl.dim() in (2, 3)
        reps = 1 if dbias_tbl.dim() == 2 else dbias_tbl.shape[0]
    _lib.check(lib.paced_enc_attn_bwd(_p(q), _p(k), _p(v), ld, hs, _p(ctx), ldo, hso, _p(dctx), lddo, hsdo,
                                      _p(lse), _p(bias_tbl), _p(dq), _p(dk), _p(dv), ldd, hsd, _p(dbias_tbl), reps,
                                      N, S, H, 64, drop_p, drop_seed, _stream()), "paced_enc_attn_bwd")


def dec_self_attn_fwd(qkv, ctx, bias2, N, H, *, drop_p=0.0, drop_seed=0):
    lib = _lib.load()
    _chk(qkv, BF16, "qkv"); _chk(ctx, BF16, "ctx"); _chk(bias2, F32, "bias2")
    assert tuple(qkv.shape) == (N * 2, 3 * H * 64) and tuple(ctx.shape) == (N * 2, H * 64) and bias2.numel() == 2 * H
    _lib.check(lib.paced_dec_self_attn_fwd(_p(qkv), _rows2d(qkv, "qkv"), _p(ctx), _rows2d(ctx, "ctx"), _p(bias2), N, H, 64,
                                           drop_p, drop_seed, _stream()), "paced_dec_self_attn_fwd")
    return ctx


def dec_self_attn_bwd(qkv, dctx, dqkv, bias2, dbias2, N, H, *, drop_p=0.0, drop_seed=0):
    lib = _lib.load()
    _chk(qkv, BF16, "qkv"); _chk(dctx, BF16, "dctx"); _chk(dqkv, BF16, "dqkv"); _chk(bias2, F32, "bias2"); _chk(dbias2, F32, "dbias2")
    assert tuple(qkv.shape) == (N * 2, 3 * H * 64) and tuple(dqkv.shape) == tuple(qkv.shape) and tuple(dctx.shape) == (N * 2, H * 64)
    assert bias2.numel() == 2 * H and (dbias2 is None or dbias2.numel() == 2 * H)
    _lib.check(lib.paced_dec_self_attn_bwd(_p(qkv), _rows2d(qkv, "qkv"), _p(dctx), _rows2d(dctx, "dctx"), _p(dqkv), _rows2d(dqkv, "dqkv"),
                                           _p(bias2), _p(dbias2), N, H, 64, drop_p, drop_seed, _stream()), "paced_dec_self_attn_bwd")


def cross_attn_fwd(qc, kv, ctx, lse, N, S, H, *, drop_p=0.0, drop_seed=0):
    lib = _lib.load()
    _chk(qc, BF16, "qc"); _chk(kv, BF16, "kv"); _chk(ctx, BF16, "ctx"); _chk(lse, F32, "lse")
    I = H * 64
    assert tuple(qc.shape) == (N * 2, I) and tuple(kv.shape) == (N * S, 2 * I) and tuple(ctx.shape) == (N * 2, I)
    assert lse.numel() == N * H * 2 and lse.is_contiguous()
    _lib.check(lib.paced_cross_attn_fwd(_p(qc), _rows2d(qc, "qc"), _p(kv), _rows2d(kv, "kv"), I, _p(ctx), _rows2d(ctx, "ctx"), _p(lse),
                                        N, S, H, 64, drop_p, drop_seed, _stream()), "paced_cross_attn_fwd")
    return ctx


def cross_attn_bwd(qc, kv, ctx, dctx, lse, dqc, dkv, N, S, H, *, drop_p=0.0, drop_seed=0):
    lib = _lib.load()
    for t, n in ((qc, "qc"), (kv, "kv"), (ctx, "ctx"), (dctx, "dctx"), (dqc, "dqc"), (dkv, "dkv")):
        _chk(t, BF16, n)
    _chk(lse, F32, "lse")
    I = H * 64
    assert tuple(qc.shape) == (N * 2, I) and tuple(kv.shape) == (N * S, 2 * I) and tuple(ctx.shape) == (N * 2, I)
    assert tuple(dctx.shape) == (N * 2, I) and tuple(dqc.shape) == (N * 2, I) and tuple(dkv.shape) == (N * S, 2 * I)
    _lib.check(lib.paced_cross_attn_bwd(_p(qc), _rows2d(qc, "qc"), _p(kv), _rows2d(kv, "kv"), I, _p(ctx), _rows2d(ctx, "ctx"),
                                        _p(dctx), _rows2d(dctx, "dctx"), _p(lse), _p(dqc), _rows2d(dqc, "dqc"), _p(dkv), _rows2d(dkv, "dkv"),
                                        N, S, H, 64, drop_p, drop_seed, _stream()), "paced_cross_attn_bwd")


def cross2_fwd(enc, qk, z, lse, N, S, H, *, drop_p=0.0, drop_seed=0):
    """Projection-free decoder cross-attention: enc [N*S, d], qk/z [N*2, H*d] (row n*2+t, head h at column h*d), lse [N*2H]."""
    lib = _lib.load()
    _chk(enc, BF16, "enc"); _chk(qk, BF16, "qk"); _chk(z, BF16, "z"); _chk(lse, F32, "lse")
    d = enc.shape[1]
    assert enc.shape[0] == N * S and tuple(qk.shape) == (N * 2, H * d) and tuple(z.shape) == (N * 2, H * d)
    assert qk.is_contiguous() and z.is_contiguous() and lse.is_contiguous() and lse.numel() == N * 2 * H
    _lib.check(lib.paced_cross2_fwd(_p(enc), _rows2d(enc, "enc"), _p(qk), _p(z), _p(lse), N, S, H, d, drop_p, drop_seed, _stream()),
               "paced_cross2_fwd")
    return z


def cross2_bwd(enc, qk, dz, z, lse, dqk, wslot, wstride, uslot, ustride, N, S, H, *, wld=None, drop_p=0.0, drop_seed=0):
    """Backward of cross2_fwd; wslot/uslot are views whose first element is this layer's slot of the factor buffers;
    ``wld`` is the row stride of the [2C][wld] score factors (default S; S rounded up to a multiple of 8 lets the
    factors feed the token-contraction product for any S)."""
    wld = S if wld is None else wld
    lib = _lib.load()
    for t, n in ((enc, "enc"), (qk, "qk"), (dz, "dz"), (z, "z"), (dqk, "dqk"), (wslot, "wslot"), (uslot, "uslot")):
        _chk(t, BF16, n)
    _chk(lse, F32, "lse")
    d = enc.shape[1]
    for t in (qk, dz, z, dqk):
        assert tuple(t.shape) == (N * 2, H * d) and t.is_contiguous()
    _lib.check(lib.paced_cross2_bwd(_p(enc), _rows2d(enc, "enc"), _p(qk), _p(dz), _p(z), _p(lse), _p(dqk), _p(wslot), wstride, wld,
                                    _p(uslot), ustride, N, S, H, d, drop_p, drop_seed, _stream()), "paced_cross2_bwd")


def cross_entropy(logits, labels, loss, dlogits=None, *, ignore_index=-100, gscale=1.0):
    lib = _lib.load()
    _chk(logits, F32, "logits"); _chk(labels, I64, "labels"); _chk(loss, F32, "loss"); _chk(dlogits, BF16, "dlogits")
    R, V = logits.shape
    ldl = _rows2d(logits, "logits")
    assert labels.numel() == R and labels.is_contiguous() and loss.numel() == R
    assert dlogits is None or (tuple(dlogits.shape) == (R, V) and _rows2d(dlogits, "dlogits") == ldl)
    _lib.check(lib.paced_cross_entropy(_p(logits), _p(labels), _p(loss), _p(dlogits), R, V, ldl, ignore_index, gscale, _stream()), "paced_cross_entropy")
    return loss


def lce_group(pce, nce, ce, n, use_mean):
    lib = _lib.load()
    _chk(pce, F32, "pce"); _chk(nce, F32, "nce"); _chk(ce, F32, "ce")
    R = pce.numel()
    assert nce.numel() == R * n and ce.numel() == R
    _lib.check(lib.paced_lce_group(_p(pce), _p(nce), _p(ce), R, n, int(bool(use_mean)), _stream()), "paced_lce_group")
    return ce


def lce_weights(loss, eta, w, stats):
    lib = _lib.load()
    _chk(loss, F32, "loss"); _chk(eta, F32, "eta"); _chk(w, F32, "w"); _chk(stats, F32, "stats")
    R = loss.numel()
    assert w.numel() == R and stats.numel() >= 4 and eta.numel() == 1
    _lib.check(lib.paced_lce_weights(_p(loss), _p(eta), _p(w), _p(stats), R, _stream()), "paced_lce_weights")
    return w


def eta_weights(loss, eta, w, dw=None, *, mask=0.0, form=1):
    """Threshold weights of pacednegatives/weights.py (form 0 = LCEWeights, 1 = EtaWeights) and their slope in eta."""
    lib = _lib.load()
    _chk(loss, F32, "loss"); _chk(eta, F32, "eta"); _chk(w, F32, "w"); _chk(dw, F32, "dw")
    R = loss.numel()
    if w.numel() != R or (dw is not None and dw.numel() != R) or eta.numel() < 1:
        raise ValueError("eta_weights: w and dw must have the length of loss")
    _lib.check(lib.paced_eta_weights(_p(loss), _p(eta), float(mask), int(form), _p(w), _p(dw), R, _stream()), "paced_eta_weights")


def eta_bound_meta(pce, nce, eta, v, out, *, mask=0.0):
    """Meta objective of the eta-bound schedule; out = [J, dJ/d eta, mean v, success rate]."""
    lib = _lib.load()
    for t, n in ((pce, "pce"), (nce, "nce"), (eta, "eta"), (v, "v"), (out, "out")):
        _chk(t, F32, n)
    R = pce.numel()
    if nce.numel() != R or v.numel() != R or out.numel() < 4:
        raise ValueError("eta_bound_meta: pce, nce and v must have one length and out four slots")
    _lib.check(lib.paced_eta_bound_meta(_p(pce), _p(nce), _p(eta), float(mask), _p(v), _p(out), R, _stream()), "paced_eta_bound_meta")


def adamw(p, g, m, v, p_bf16, *, beta1=0.9, beta2=0.999, eps=1e-6, step_size, decay=0.0, grad_scale=1.0, clear_grad=False):
    lib = _lib.load()
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, F32, n)
        assert t.is_contiguous() and t.numel() == p.numel()
    _chk(p_bf16, BF16, "p_bf16")
    assert p_bf16 is None or (p_bf16.is_contiguous() and p_bf16.numel() == p.numel())
    _lib.check(lib.paced_adamw(_p(p), _p(g), _p(m), _p(v), _p(p_bf16), p.numel(), beta1, beta2, eps, step_size, decay, grad_scale,
                               int(bool(clear_grad)), _stream()), "paced_adamw")


def pool_topk(queries, pool, k, *, index_base=0, workspace=None):
    """Exact top-k of queries @ pool.T per query: (scores f32 [Q,k], idx i64 [Q,k]), best first."""
    lib = _lib.load()
    _chk(queries, BF16, "queries"); _chk(pool, BF16, "pool")
    Q, D = queries.shape
    Np = pool.shape[0]
    assert pool.shape[1] == D and queries.is_contiguous() and pool.is_contiguous()
    need = lib.paced_pool_topk_workspace_bytes(Q, k)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=queries.device)
    scores = torch.empty(Q, k, dtype=F32, device=queries.device)
    idx = torch.empty(Q, k, dtype=I64, device=queries.device)
    _lib.check(lib.paced_pool_topk(_p(queries), _p(pool), Q, Np, D, k, index_base, _p(scores), _p(idx), _p(workspace), workspace.numel(),
                                   _stream()), "paced_pool_topk")
    return scores, idx


def pool_topk_static(queries, pool, k, out_scores, out_idx, *, index_base=0):
    """``pool_topk`` into caller-provided outputs with the library's built-in candidate array instead of a workspace:
    the call allocates nothing (limits: ``paced_pool_topk_static_max_queries()`` queries, ``..._max_k()`` results)."""
    lib = _lib.load()
    _chk(queries, BF16, "queries"); _chk(pool, BF16, "pool"); _chk(out_scores, F32, "out_scores"); _chk(out_idx, I64, "out_idx")
    Q, D = queries.shape
    if pool.dim() != 2 or pool.shape[1] != D or not (queries.is_contiguous() and pool.is_contiguous()):
        raise ValueError("pool must be a contiguous [Np, D] matrix with the queries' D")
    if tuple(out_scores.shape) != (Q, k) or tuple(out_idx.shape) != (Q, k) or not (out_scores.is_contiguous() and out_idx.is_contiguous()):
        raise ValueError(f"outputs must be contiguous [{Q}, {k}]")
    _lib.check(lib.paced_pool_topk_static(_p(queries), _p(pool), Q, pool.shape[0], D, k, index_base, _p(out_scores), _p(out_idx),
                                          _stream()), "paced_pool_topk_static")
    return out_scores, out_idx


def topk_merge(scores, idx, k):
    lib = _lib.load()
    _chk(scores, F32, "scores"); _chk(idx, I64, "idx")
    Q, total = scores.shape
    assert idx.shape == scores.shape and scores.is_contiguous() and idx.is_contiguous()
    out_s = torch.empty(Q, k, dtype=F32, device=scores.device)
    out_i = torch.empty(Q, k, dtype=I64, device=scores.device)
    _lib.check(lib.paced_topk_merge(_p(scores), _p(idx), Q, total, k, _p(out_s), _p(out_i), _stream()), "paced_topk_merge")
    return out_s, out_i

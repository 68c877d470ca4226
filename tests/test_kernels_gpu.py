"""Per-kernel parity tests through the C ABI (ctypes) against plain fp32 torch math on the same inputs.

Integer-valued inputs are used wherever an exact answer exists (every product and partial sum is
exactly representable, so the result is independent of summation order): those checks are bit-exact.
Floating-point checks state their tolerance next to the comparison.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from pacednegatives_amd import kernels as K  # noqa: E402

DEV = "cuda"


def ints(shape, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).float()


def bf(x):
    return x.to(DEV).to(torch.bfloat16).contiguous()


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


# ------------------------------------------------------------------------------------------- GEMM
@pytest.fixture(params=[64, 128, 256])
def tile(request):
    K.gemm_set_tile(request.param)
    yield request.param
    K.gemm_set_tile(0)


@pytest.mark.parametrize("M,N,K_", [(128, 128, 64), (300, 136, 128), (1, 8, 64), (257, 520, 192), (1000, 768, 256)])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True)])
def test_gemm_exact_integers(M, N, K_, ta, tb, tile):
    if ta and M % 8:
        M = (M + 7) // 8 * 8
    Kc = K_ if not ta else K_ + 24          # ragged token contraction is allowed for transposed operands
    if not ta and tb:
        Kc = K_
    a = ints((M, Kc), -3, 3, 1)
    b = ints((N, Kc), -3, 3, 2)
    want = (a @ b.t())                       # |sum| <= 9*280: exact in fp32
    A = bf(a.t() if ta else a)
    B = bf(b.t() if tb else b)
    c = torch.full((M, N), float("nan"), device=DEV)
    K.gemm(A, B, c, trans_a=ta, trans_b=tb, epilogue=K.EPI_F32)
    torch.cuda.synchronize()
    assert torch.equal(c.cpu(), want)


@pytest.mark.parametrize("M,N,K_", [(101, 136, 96), (250, 768, 200), (3, 8, 40), (515, 256, 129)])
def test_gemm_transposed_a_with_ragged_rows_exact(M, N, K_, tile):
    """A stored [K][M] with M not a multiple of 8: rows padded to a multiple of 8 elements.  The padding is read with the
    last 16-byte piece of every contraction row but only feeds output rows >= M, which are never stored: NaN padding
    must leave the result exact.  (The token-contraction product of the decoder cross-attention's backward has M = the
    sequence length, which is arbitrary.)"""
    M8 = (M + 7) // 8 * 8
    a, b = ints((K_, M), -3, 3, 5), ints((K_, N), -3, 3, 6)
    A = torch.full((K_, M8), float("nan"), device=DEV, dtype=torch.bfloat16)
    A[:, :M] = a.to(DEV).to(torch.bfloat16)
    c = torch.full((M, N), float("nan"), device=DEV)
    K.gemm(A[:, :M], bf(b), c, trans_a=True, trans_b=True, epilogue=K.EPI_F32)
    assert torch.equal(c.cpu(), a.t() @ b)
    c16 = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
    K.gemm(A[:, :M], bf(b), c16, trans_a=True, trans_b=True)
    assert torch.equal(c16.float().cpu(), (a.t() @ b).to(torch.bfloat16).float())


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True)])
def test_gemm_many_tiles_per_workgroup_exact(ta, tb, tile):
    """More output tiles than compute units: every persistent workgroup walks several work items, the
    last ones ragged in both directions."""
    M, N, Kc = 5000, 4104, 192
    if ta:
        Kc += 40
    a = ints((M, Kc), -2, 2, 21)
    b = ints((N, Kc), -2, 2, 22)
    A = bf(a.t() if ta else a)
    B = bf(b.t() if tb else b)
    want = A.float().t() @ B.float() if (ta and tb) else (A.float() @ (B.float() if tb else B.float().t()))
    c = torch.full((M, N), float("nan"), device=DEV)
    K.gemm(A, B, c, trans_a=ta, trans_b=tb, epilogue=K.EPI_F32)
    torch.cuda.synchronize()
    assert torch.equal(c, want)
    if ta:
        c2 = torch.zeros(M, N, device=DEV)
        K.gemm(A, B, c2, trans_a=ta, trans_b=tb, epilogue=K.EPI_ATOMIC_F32, splits=2)
        torch.cuda.synchronize()
        assert torch.equal(c2, want)


@pytest.mark.parametrize("limit", [1, 100, 224])
def test_gemm_grid_limit_gives_the_same_product(limit):
    """paced_gemm_set_grid: a phased launch with fewer workgroups than compute units walks more items per workgroup
    and leaves the other units free; the product is the same, ragged tiles and split contraction included."""
    M, N, Kc = 5000, 4104, 192
    a, b = ints((M, Kc), -2, 2, 31), ints((N, Kc), -2, 2, 32)
    A, B = bf(a), bf(b)
    want = A.float() @ B.float().t()
    K.gemm_set_tile(256)
    try:
        K.gemm_set_grid(limit)
        c = torch.full((M, N), float("nan"), device=DEV)
        K.gemm(A, B, c, epilogue=K.EPI_F32)
        assert K._lib.load().paced_gemm_last_kernel() == 256
        g = torch.zeros(Kc, Kc, device=DEV)                      # weight-gradient shape: the split follows the limit
        K.gemm(A[:, :Kc], A[:, :Kc], g, trans_a=True, trans_b=True, epilogue=K.EPI_ATOMIC_F32, splits=0)
        torch.cuda.synchronize()
    finally:
        K.gemm_set_grid(0)
        K.gemm_set_tile(0)
    assert torch.equal(c, want)
    assert torch.equal(g, A.float().t() @ A.float())
    with pytest.raises(RuntimeError):
        K.gemm_set_grid(-1)
    # the same limit and tile as options of the call: same bits, and the process defaults are left alone
    assert K.gemm_get_grid() == 0
    c2 = torch.full((M, N), float("nan"), device=DEV)
    K.gemm(A, B, c2, epilogue=K.EPI_F32, grid_limit=limit, tile=256)
    g2 = torch.zeros(Kc, Kc, device=DEV)
    K.gemm(A[:, :Kc], A[:, :Kc], g2, trans_a=True, trans_b=True, epilogue=K.EPI_ATOMIC_F32, splits=0, grid_limit=limit, tile=256)
    torch.cuda.synchronize()
    assert torch.equal(c2, want) and torch.equal(g2, g)
    assert K.gemm_get_grid() == 0
    K.gemm_set_grid(7)
    try:
        assert K.gemm_get_grid() == 7
        K.gemm(A, B, c2, epilogue=K.EPI_F32, grid_limit=0, tile=128)       # an explicit 0 overrides the default of 7
        torch.cuda.synchronize()
        assert torch.equal(c2, want)
        assert K.gemm_get_grid() == 7                                       # a call's options never touch the default
    finally:
        K.gemm_set_grid(0)


def test_gemm_entry_points_without_options_equal_the_one_with():
    """paced_gemm_bf16 / _heads / _batched (no per-call options) against paced_gemm_bf16_ex, which the wrappers use."""
    lib = K._lib.load()
    st = torch.cuda.current_stream().cuda_stream
    M, N, Kc = 300, 256, 192
    A, B = bf(rnd((M, Kc), 70)), bf(rnd((N, Kc), 71))
    res = bf(rnd((M, N), 72))
    want = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    K.gemm(A, B, want, epilogue=K.EPI_RES_DROP, aux=res, drop_p=0.2, drop_seed=9)
    got = torch.empty_like(want)
    K._lib.check(lib.paced_gemm_bf16(A.data_ptr(), B.data_ptr(), got.data_ptr(), res.data_ptr(), M, N, Kc, Kc, Kc, N, N, 0, 0,
                                     K.EPI_RES_DROP, 1.0, 1.0, 0.2, 9, 1, st))
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    # head-major output
    N2 = 256
    wanth = torch.zeros(N2 // 64, M, 64, device=DEV, dtype=torch.bfloat16)
    K.gemm(A, B, wanth)
    goth = torch.zeros_like(wanth)
    K._lib.check(lib.paced_gemm_bf16_heads(A.data_ptr(), B.data_ptr(), goth.data_ptr(), 0, M, N2, Kc, Kc, Kc, 64, 0, 0, 0,
                                           K.EPI_BF16, 1.0, 1.0, 0.0, 0, 1, 0, 0, wanth.stride(0), st))
    torch.cuda.synchronize()
    assert torch.equal(goth, wanth)
    # batched: 3 members of [64, 64] x [64, 64]^T
    a3, b3 = bf(rnd((3 * 64, 64), 73)), bf(rnd((3 * 64, 64), 74))
    want3 = torch.empty(3 * 64, 64, device=DEV, dtype=torch.bfloat16)
    K.gemm_batched(a3, b3, want3, M=64, N=64, K=64, lda=64, ldb=64, ldc=64, batch=3, stride_a=4096, stride_b=4096, stride_c=4096)
    got3 = torch.empty_like(want3)
    K._lib.check(lib.paced_gemm_bf16_batched(a3.data_ptr(), b3.data_ptr(), got3.data_ptr(), 64, 64, 64, 64, 64, 64, 0, 0, K.EPI_BF16, 1.0, 1,
                                             3, 4096, 4096, 4096, st))
    torch.cuda.synchronize()
    assert torch.equal(got3, want3)
    opts = K._lib.GemmOpts(-1, 999, -1, 0)
    assert lib.paced_gemm_bf16_ex(A.data_ptr(), B.data_ptr(), got.data_ptr(), 0, M, N, Kc, Kc, Kc, N, 0, 0, 0, K.EPI_BF16, 1.0, 1.0, 0.0, 0, 1,
                                  0, 0, 0, 1, 0, 0, 0, opts, st) != 0 and b"tile" in lib.paced_last_error()


def test_gemm_batched_per_head_products(tile):
    """The three batched shapes of the decoder cross-attention: per-head slices of a projection matrix."""
    R, H, dk, d = 72, 3, 64, 192                       # rows (sequences x 2), heads, head size, model width
    q = ints((R, H * dk), -2, 2, 31)                   # per-head queries side by side
    W = ints((H * dk, d), -2, 2, 32)                   # projection weight [out = H*dk, in = d]
    Q, Wd = bf(q), bf(W)
    # (i) out[r, h, :] = q[r, h*dk:(h+1)*dk] @ W[h*dk:(h+1)*dk, :]
    out = torch.full((R, H * d), float("nan"), device=DEV)
    K.gemm_batched(Q, Wd, out, M=R, N=d, K=dk, lda=H * dk, ldb=d, ldc=H * d, batch=H, stride_a=dk, stride_b=dk * d, stride_c=d,
                   trans_b=True, epilogue=K.EPI_F32)
    want = torch.einsum("rhk,hkd->rhd", q.view(R, H, dk), W.view(H, dk, d)).reshape(R, H * d)
    assert torch.equal(out.cpu(), want)
    # (ii) ctx[r, h*dk:(h+1)*dk] = z[r, h, :] @ W[h*dk:(h+1)*dk, :]^T
    z = ints((R, H * d), -2, 2, 33)
    ctx = torch.full((R, H * dk), float("nan"), device=DEV)
    K.gemm_batched(bf(z), Wd, ctx, M=R, N=dk, K=d, lda=H * d, ldb=d, ldc=H * dk, batch=H, stride_a=d, stride_b=dk * d, stride_c=dk,
                   epilogue=K.EPI_F32)
    want = torch.einsum("rhd,hkd->rhk", z.view(R, H, d), W.view(H, dk, d)).reshape(R, H * dk)
    assert torch.equal(ctx.cpu(), want)
    # (iii) dW[h*dk:(h+1)*dk, :] += q[:, h]^T @ z[:, h, :]   (contraction over rows, atomic accumulation)
    dW = torch.ones(H * dk, d, device=DEV)
    K.gemm_batched(Q, bf(z), dW, M=dk, N=d, K=R, lda=H * dk, ldb=H * d, ldc=d, batch=H, stride_a=dk, stride_b=d, stride_c=dk * d,
                   trans_a=True, trans_b=True, epilogue=K.EPI_ATOMIC_F32, splits=0)
    want = 1.0 + torch.einsum("rhk,rhd->hkd", q.view(R, H, dk), z.view(R, H, d)).reshape(H * dk, d)
    assert torch.equal(dW.cpu(), want)


def test_gemm_batched_large_tiles():
    """Batched token-contraction product with bf16 output through the 256x256 kernel (one batch member per sequence)."""
    Nb, S, Kc, d = 40, 256, 128, 768
    w = ints((Nb, Kc, S), -2, 2, 34)
    u = ints((Nb, Kc, d), -2, 2, 35)
    out = torch.empty(Nb * S, d, device=DEV, dtype=torch.bfloat16)
    K.gemm_batched(bf(w.view(Nb * Kc, S)), bf(u.view(Nb * Kc, d)), out, M=S, N=d, K=Kc, lda=S, ldb=d, ldc=d, batch=Nb,
                   stride_a=Kc * S, stride_b=Kc * d, stride_c=S * d, trans_a=True, trans_b=True)
    want = torch.einsum("nks,nkd->nsd", w, u).reshape(Nb * S, d)
    assert torch.equal(out.float().cpu(), want.to(torch.bfloat16).float())


def test_gemm_strided_views_and_alpha(tile):
    a = ints((200, 128), -2, 2, 3)
    b = ints((72, 128), -2, 2, 4)
    Abig = bf(torch.cat([a, a + 1], dim=1))           # lda = 256, use the first 128 columns
    B = bf(b)
    cbig = torch.zeros(200, 80, device=DEV)
    K.gemm(Abig[:, :128], B, cbig[:, :72], epilogue=K.EPI_F32, alpha=0.5)
    torch.cuda.synchronize()
    assert torch.equal(cbig[:, :72].cpu(), 0.5 * (a @ b.t()))
    assert (cbig[:, 72:] == 0).all()


def test_gemm_epilogues(tile):
    M, N, Kc = 260, 256, 128
    a, b = ints((M, Kc), -2, 2, 5), ints((N, Kc), -2, 2, 6)
    res = ints((M, N), -20, 20, 7)
    A, B, R = bf(a), bf(b), bf(res)
    acc = a @ b.t()
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    K.gemm(A, B, out, epilogue=K.EPI_BF16)
    assert torch.equal(out.float().cpu(), acc.to(torch.bfloat16).float())
    # residual form: a tile configuration may round the product to bf16 before the add (what a bf16 linear
    # layer followed by a bf16 add does) or add in fp32; both are within one bf16 ulp of the exact sum
    want = acc + res
    K.gemm(A, B, out, epilogue=K.EPI_RES_DROP, aux=R)
    assert ((out.float().cpu() - want).abs() <= want.abs() * 2 ** -7).all()
    inplace = R.clone()
    K.gemm(A, B, inplace, epilogue=K.EPI_RES_DROP, aux=inplace)
    assert torch.equal(inplace, out)
    K.gemm(A, B, out, epilogue=K.EPI_RELU_DROP)
    assert torch.equal(out.float().cpu(), acc.clamp_min(0).to(torch.bfloat16).float())
    K.gemm(A, B, out, epilogue=K.EPI_MASK, aux=R, mask_scale=2.0)
    assert torch.equal(out.float().cpu(), torch.where(res > 0, acc * 2.0, torch.zeros_like(acc)).to(torch.bfloat16).float())


def unpack_bits(bits, N):
    """[M, N/8] uint8 -> [M, N] bool, bit n%8 of byte n//8 <-> column n."""
    sh = torch.arange(8, device=bits.device, dtype=torch.uint8)
    return ((bits.unsqueeze(-1) >> sh) & 1).bool().reshape(bits.shape[0], -1)[:, :N]


@pytest.mark.parametrize("shape", [(260, 256, 128), (515, 776, 64)])
@pytest.mark.parametrize("p", [0.0, 0.3])
def test_gemm_sign_bit_matrix(shape, p, tile):
    """relu+dropout writes one bit per output (output > 0); the bit-masked product equals the one masked by the
    saved output itself, on ragged row counts and column counts that end inside a tile."""
    M, N, Kc = shape
    a, b = ints((M, Kc), -2, 2, 40), ints((N, Kc), -2, 2, 41)
    act = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    plain = torch.empty_like(act)
    bits = torch.full((M + 1, N // 8 + 1), 0xA5, device=DEV, dtype=torch.uint8)      # guard row and column
    K.gemm(bf(a), bf(b), act, epilogue=K.EPI_RELU_DROP, aux=bits[:M, :N // 8], drop_p=p, drop_seed=77)
    K.gemm(bf(a), bf(b), plain, epilogue=K.EPI_RELU_DROP, drop_p=p, drop_seed=77)
    assert torch.equal(act, plain)
    assert torch.equal(unpack_bits(bits[:M, :N // 8], N), act > 0)
    assert (bits[M] == 0xA5).all() and (bits[:, N // 8] == 0xA5).all()
    Kb = 64
    g, w = ints((M, Kb), -2, 2, 42), ints((N, Kb), -2, 2, 43)
    by_bits, by_act = torch.empty_like(act), torch.empty_like(act)
    K.gemm(bf(g), bf(w), by_bits, epilogue=K.EPI_MASK_BITS, aux=bits[:M, :N // 8], mask_scale=1.5)
    K.gemm(bf(g), bf(w), by_act, epilogue=K.EPI_MASK, aux=act, mask_scale=1.5)
    assert torch.equal(by_bits, by_act)
    want = torch.where(act.cpu() > 0, (g @ w.t()) * 1.5, torch.zeros(M, N))
    assert torch.equal(by_bits.float().cpu(), want.to(torch.bfloat16).float())


@pytest.mark.parametrize("shape", [(512, 512, 128), (700, 776, 192), (4352, 3072, 128)])
@pytest.mark.parametrize("p", [0.0, 0.3])
def test_gemm_tile_ordered_sign_bits(shape, p):
    """The 256x256 kernel's own order for the sign bits (a 1-D buffer, ldaux = -1): the writer's outputs and the
    reader's masked product equal those of the row-major bit matrix and of the mask by the saved output, on whole and
    ragged tiles, with two and more K steps on the reader's side; shapes that do not take that kernel are refused."""
    M, N, Kc = shape
    lib = K._lib.load()
    assert lib.paced_gemm_tile_bits_bytes(128, 3072) == 0 and lib.paced_gemm_tile_bits_bytes(M, 260) == 0
    a, b = ints((M, Kc), -2, 2, 50), ints((N, Kc), -2, 2, 51)
    g, w = ints((M, Kc), -2, 2, 52), ints((N, Kc), -2, 2, 53)
    act, plain = torch.empty(M, N, device=DEV, dtype=torch.bfloat16), torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    by_tiles, by_rows, by_act = torch.empty_like(act), torch.empty_like(act), torch.empty_like(act)
    rows = torch.empty(M, N // 8, device=DEV, dtype=torch.uint8)
    K.gemm_set_tile(256)
    try:
        need = lib.paced_gemm_tile_bits_bytes(M, N)
        assert need == ((M + 255) // 256) * ((N + 255) // 256) * 8192
        tiles = torch.full((need + 16,), 0xA5, device=DEV, dtype=torch.uint8)
        K.gemm(bf(a), bf(b), act, epilogue=K.EPI_RELU_DROP, aux=tiles[:need], drop_p=p, drop_seed=77)
        K.gemm(bf(a), bf(b), plain, epilogue=K.EPI_RELU_DROP, aux=rows, drop_p=p, drop_seed=77)
        assert lib.paced_gemm_last_kernel() == 256
        assert torch.equal(act, plain)
        assert (tiles[need:] == 0xA5).all()
        assert int(tiles[:need].view(torch.int32).ne(0).sum()) > 0
        K.gemm(bf(g), bf(w), by_tiles, epilogue=K.EPI_MASK_BITS, aux=tiles[:need], mask_scale=1.5)
        K.gemm(bf(g), bf(w), by_rows, epilogue=K.EPI_MASK_BITS, aux=rows, mask_scale=1.5)
        K.gemm(bf(g), bf(w), by_act, epilogue=K.EPI_MASK, aux=act, mask_scale=1.5)
        assert torch.equal(by_tiles, by_rows) and torch.equal(by_tiles, by_act)
        # the reader needs two K steps to cover its early load
        with pytest.raises(RuntimeError):
            K.gemm(bf(g[:, :64].contiguous()), bf(w[:, :64].contiguous()), by_tiles, epilogue=K.EPI_MASK_BITS, aux=tiles[:need])
        with pytest.raises(ValueError):
            K.gemm(bf(g), bf(w), by_tiles, epilogue=K.EPI_MASK_BITS, aux=tiles[:need - 16])
    finally:
        K.gemm_set_tile(0)
    if lib.paced_gemm_tile_bits_bytes(M, N) == 0:          # not a shape the heuristic gives the 256x256 kernel
        assert ((M + 255) // 256) * ((N + 255) // 256) < 200
        with pytest.raises(ValueError):
            K.gemm(bf(g), bf(w), by_tiles, epilogue=K.EPI_MASK_BITS, aux=tiles[:need])
    else:
        assert tuple(K.sign_bits_buffer(M, N, DEV).shape) == (need,)


@pytest.mark.parametrize("splits", [1, 3, 8])
def test_gemm_weight_gradient_atomic_split(splits, tile):
    Mtok, No, Ko = 1000, 136, 192                     # dW[No,Ko] = dY[Mtok,No]^T X[Mtok,Ko]
    dy, x = ints((Mtok, No), -2, 2, 8), ints((Mtok, Ko), -2, 2, 9)
    c = torch.ones(No, Ko, device=DEV)
    K.gemm(bf(dy), bf(x), c, trans_a=True, trans_b=True, epilogue=K.EPI_ATOMIC_F32, splits=splits)
    torch.cuda.synchronize()
    assert torch.equal(c.cpu(), 1.0 + dy.t() @ x)


def test_gemm_float_tolerance(tile):
    M, N, Kc = 512, 768, 3072
    a, b = rnd((M, Kc), 10), rnd((N, Kc), 11, Kc ** -0.5)
    A, B = bf(a), bf(b)
    c = torch.empty(M, N, device=DEV)
    K.gemm(A, B, c, epilogue=K.EPI_F32)
    want = A.float() @ B.float().t()
    # same bf16 inputs, fp32 accumulation in a different order: 1e-5 relative to the row scale
    assert (c - want).abs().max().item() < 1e-4 * want.abs().max().item()


def test_gemm_dropout_matches_mask_function(tile):
    M, N, Kc = 192, 128, 64
    a, b = ints((M, Kc), -2, 2, 12), ints((N, Kc), -2, 2, 13)
    acc = (a @ b.t()).to(DEV)
    p, seed = 0.25, 1234
    mask = K.dropout_mask(M * N, drop_p=p, drop_seed=seed).view(M, N)
    keep = (mask > 0).float().mean().item()
    assert abs(keep - (1 - p)) < 0.02
    assert torch.allclose(mask[mask > 0], torch.tensor(1 / (1 - p), device=DEV))
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    K.gemm(bf(a), bf(b), out, epilogue=K.EPI_RELU_DROP, drop_p=p, drop_seed=seed)
    assert torch.equal(out.float(), (acc.clamp_min(0) * mask).to(torch.bfloat16).float())
    res = bf(ints((M, N), -9, 9, 14))
    K.gemm(bf(a), bf(b), out, epilogue=K.EPI_RES_DROP, aux=res, drop_p=p, drop_seed=seed)
    # the kernel may contract res + acc*factor into one fma (one rounding instead of two): allow 1 bf16 ulp
    # (or round acc*factor to bf16 first and then add: one more half-ulp of that product)
    want = res.float() + acc * mask
    assert ((out.float() - want).abs() <= want.abs() * 2 ** -7 + (acc * mask).abs() * 2 ** -8 + 1e-6).all()
    assert torch.equal((out.float() == res.float()), (mask == 0) | (acc == 0))
    y = torch.empty(M * N, device=DEV, dtype=torch.bfloat16)
    x = bf(acc.cpu().reshape(-1))
    K.dropout_apply(x, y, drop_p=p, drop_seed=seed)
    assert torch.equal(y.float(), (x.float() * mask.view(-1)).to(torch.bfloat16).float())


def test_gemm_rejects_bad_shapes():
    from pacednegatives_amd._lib import PacedHipError
    A, B = bf(torch.zeros(64, 72)), bf(torch.zeros(64, 72))
    with pytest.raises(PacedHipError, match="multiple of 64"):
        K.gemm(A, B, torch.zeros(64, 64, device=DEV), epilogue=K.EPI_F32)


@pytest.mark.parametrize("M,N,K_", [(1088, 768, 768), (1088, 768, 3072), (1088, 2304, 768), (1088, 3072, 768), (200, 264, 1088)])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True)])
@pytest.mark.parametrize("epi", ["bf16", "res_drop", "relu_drop", "mask", "f32"])
def test_gemm_ring_kernel_matches_the_single_buffer_kernel(M, N, K_, ta, tb, epi):
    """Shapes with a few tiles per CU (the decoder's) take the ring kernel; it multiplies the same K steps in the same
    order, so every non-atomic epilogue must give the same bits as the kernel it replaces."""
    if ta and (M % 8 or epi != "bf16" and (M, N, K_) != (1088, 768, 768)):
        pytest.skip("transposed A: one epilogue per shape is enough")
    a = bf(rnd((K_, M) if ta else (M, K_), 70, 0.3))
    b = bf(rnd((K_, N) if tb else (N, K_), 71, 0.3))
    aux = bf(rnd((M, N), 72))
    kw = {"bf16": dict(epilogue=K.EPI_BF16), "res_drop": dict(epilogue=K.EPI_RES_DROP, aux=aux, drop_p=0.1, drop_seed=9),
          "relu_drop": dict(epilogue=K.EPI_RELU_DROP, drop_p=0.1, drop_seed=9), "mask": dict(epilogue=K.EPI_MASK, aux=aux),
          "f32": dict(epilogue=K.EPI_F32)}[epi]
    outs = []
    try:
        for ring in (True, False):
            K.gemm_set_ring(ring)
            c = torch.zeros(M, N, device=DEV, dtype=torch.float32 if epi == "f32" else torch.bfloat16)
            K.gemm(a, b, c, trans_a=ta, trans_b=tb, **kw)
            outs.append(c)
    finally:
        K.gemm_set_ring(True)
    assert torch.equal(outs[0], outs[1])
    if epi in ("bf16", "f32"):
        want = (a.float().t() if ta else a.float()) @ (b.float() if tb else b.float().t())
        assert (outs[0].float() - want).abs().max() <= 2e-2 * want.abs().max()


@pytest.mark.parametrize("M,N,K_", [(768, 768, 1088), (3072, 768, 1088), (64, 768, 1088), (768, 776, 1000)])
def test_gemm_ring_kernel_atomic_accumulate(M, N, K_):
    """Weight-gradient form (both operands K-strided, fp32 atomics into an existing buffer) with and without the ring."""
    a, b = bf(rnd((K_, M), 73, 0.3)), bf(rnd((K_, N), 74, 0.3))
    base = rnd((M, N), 75).to(DEV)
    want = base + a.float().t() @ b.float()
    try:
        for ring in (True, False):
            K.gemm_set_ring(ring)
            c = base.clone()
            K.gemm(a, b, c, trans_a=True, trans_b=True, epilogue=K.EPI_ATOMIC_F32, splits=0)
            assert (c - want).abs().max() <= 1e-3 * want.abs().max()
    finally:
        K.gemm_set_ring(True)


# --------------------------------------------------------------------------------- norm / embedding
@pytest.mark.parametrize("M,d", [(5, 128), (333, 768), (64, 1024)])
def test_rmsnorm_fwd_bwd(M, d):
    x = bf(rnd((M, d), 20, 3.0))
    w = bf(1.0 + 0.1 * rnd((d,), 21))
    dy = bf(rnd((M, d), 22))
    dres = bf(rnd((M, d), 23))
    y = torch.empty_like(x)
    rstd = torch.empty(M, device=DEV)
    K.rmsnorm_fwd(x, w, y, rstd, out_scale=0.5)
    xf = x.float().requires_grad_(True)
    wf = w.float().requires_grad_(True)
    yr = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * wf * 0.5
    assert torch.equal(y.float(), yr.detach().to(torch.bfloat16).float()) or (y.float() - yr.detach()).abs().max() < 2e-2
    # bf16 output: at most 1 ulp (2^-8 relative) from the fp32 value
    assert ((y.float() - yr.detach()).abs() <= yr.detach().abs() * 2 ** -7 + 1e-6).all()
    np.testing.assert_allclose(rstd.cpu().numpy(), torch.rsqrt(x.float().pow(2).mean(-1) + 1e-6).cpu().numpy(), rtol=1e-5)
    yr.backward(dy.float())
    dx = torch.empty_like(x)
    dw = torch.zeros(d, device=DEV)
    K.rmsnorm_bwd(dy, x, w, rstd, dx, dres=dres, dw=dw, out_scale=0.5)
    want_dx = xf.grad + dres.float()
    assert ((dx.float() - want_dx).abs() <= want_dx.abs() * 2 ** -7 + 1e-3 * want_dx.abs().max()).all()
    np.testing.assert_allclose(dw.cpu().numpy(), wf.grad.cpu().numpy(), rtol=2e-3, atol=2e-3 * wf.grad.abs().max().item())


def test_rmsnorm_bwd_second_output_is_masked_copy():
    M, d = 40, 256
    x, w, dy = bf(rnd((M, d), 24)), bf(torch.ones(d)), bf(rnd((M, d), 25))
    rstd = torch.empty(M, device=DEV)
    y = torch.empty_like(x)
    K.rmsnorm_fwd(x, w, y, rstd)
    dx, dx2 = torch.empty_like(x), torch.empty_like(x)
    K.rmsnorm_bwd(dy, x, w, rstd, dx, dx2=dx2, drop2_p=0.1, drop2_seed=77)
    mask = K.dropout_mask(M * d, drop_p=0.1, drop_seed=77).view(M, d)
    assert torch.equal(dx2.float(), (dx.float() * mask).to(torch.bfloat16).float())


@pytest.mark.parametrize("M,d", [(70, 512), (131, 768), (37, 1024)])
def test_rmsnorm_bwd_with_output_dropout_and_second_output(M, d):
    """dy masked by the (seeded) dropout of the norm output, dx2 = dx masked by a second dropout; the row-group
    kernel (d = 512, 768) and the one-row-per-wave kernel (other d) against autograd in fp32."""
    x, w, dy = bf(rnd((M, d), 26, 2.0)), bf(1.0 + 0.1 * rnd((d,), 27)), bf(rnd((M, d), 28))
    rstd = torch.empty(M, device=DEV)
    K.rmsnorm_fwd(x, w, torch.empty_like(x), rstd)
    dx, dx2 = torch.empty_like(x), torch.empty_like(x)
    dw = torch.zeros(d, device=DEV)
    K.rmsnorm_bwd(dy, x, w, rstd, dx, dx2=dx2, dw=dw, drop_p=0.1, drop_seed=5, drop2_p=0.1, drop2_seed=6)
    m1 = K.dropout_mask(M * d, drop_p=0.1, drop_seed=5).view(M, d)
    m2 = K.dropout_mask(M * d, drop_p=0.1, drop_seed=6).view(M, d)
    xf, wf = x.float().requires_grad_(True), w.float().requires_grad_(True)
    (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * wf).backward(dy.float() * m1)
    assert ((dx.float() - xf.grad).abs() <= xf.grad.abs() * 2 ** -7 + 1e-3 * xf.grad.abs().max()).all()
    assert torch.equal(dx2.float(), (dx.float() * m2).to(torch.bfloat16).float())
    np.testing.assert_allclose(dw.cpu().numpy(), wf.grad.cpu().numpy(), rtol=2e-3, atol=2e-3 * wf.grad.abs().max().item())


@pytest.mark.parametrize("p,seed", [(0.1, 5), (0.5, 0xdeadbeef), (0.013, 123456789)])
def test_dropout_mask_equals_the_numpy_restatement(p, seed):
    """The device's keep decisions against the numpy restatement of the same counter-based function (the one whose
    statistics tools/dropout_hash_stats.py reports), element for element, and the realised rate against p."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from dropout_hash_stats import drop_values
    n = 1 << 20
    got = K.dropout_mask(n, drop_p=p, drop_seed=seed).cpu().numpy()
    thresh = int(p * 65536 + 0.5)
    keep = (drop_values(seed, np.arange(n, dtype=np.uint64)) >> np.uint64(16)) >= np.uint64(thresh)
    scale = np.float32(1.0 / (1.0 - thresh / 65536.0))
    assert np.array_equal(got, np.where(keep, scale, np.float32(0)))
    assert abs(keep.mean() - (1 - p)) < 4 * math.sqrt(p * (1 - p) / n)


def test_dropout_statistics_over_the_sites_and_steps_of_a_run():
    """The masks the training step actually draws: seeds are derived per (step, call, site, layer) as T5Engine does, and a
    mask is a pure function of (seed, element index).  On 2^22 elements per mask at p = 0.1: the keep rate of every site,
    of every residue of the index modulo 4 and modulo 768 blocks; the correlation of the decisions of two sites at the
    same indices, of two consecutive steps at one site, and of neighbouring elements of one mask.  ~200 z scores are
    taken; a function without structure keeps them below ~4."""
    from pacednegatives_amd import t5_engine as E
    n, p = 1 << 22, 0.1
    thresh = int(p * 65536 + 0.5)
    keep_p = 1.0 - thresh / 65536.0
    sig_rate = math.sqrt(keep_p * (1 - keep_p) / n)
    sites = (E._SITE_EMB, E._SITE_ATT, E._SITE_ATT_OUT, E._SITE_FF_ACT, E._SITE_FF_OUT, E._SITE_CATT, E._SITE_CATT_OUT)

    def mask(step_seed, site, layer, dec=False):
        seed = E._mix(step_seed, site + 16 * layer + (4096 if dec else 0))
        return (K.dropout_mask(n, drop_p=p, drop_seed=seed) > 0).float()

    worst = {"rate": 0.0, "residue": 0.0, "cross-site": 0.0, "cross-step": 0.0, "neighbours": 0.0}
    base = 0x1234ABCD
    masks = {}
    for site in sites:
        for layer in (0, 5, 11):
            m = mask(base, site, layer)
            masks[(site, layer)] = m
            worst["rate"] = max(worst["rate"], abs(m.mean().item() - keep_p) / sig_rate)
            r4 = m.view(-1, 4).mean(0)
            worst["residue"] = max(worst["residue"], ((r4 - keep_p).abs().max().item()) / (sig_rate * 2))
    var = keep_p * (1 - keep_p)
    sig_corr = 1.0 / math.sqrt(n)

    def corr(a, b):
        return (((a - keep_p) * (b - keep_p)).mean().item() / var) / sig_corr

    keys = list(masks)
    for i in range(0, len(keys) - 1):
        worst["cross-site"] = max(worst["cross-site"], abs(corr(masks[keys[i]], masks[keys[i + 1]])))
    for site in (E._SITE_ATT, E._SITE_FF_ACT):
        a = masks[(site, 0)]
        for step in (1, 2, 8):             # training_step adds 8 per step to the step seed, the call id below that
            worst["cross-step"] = max(worst["cross-step"], abs(corr(a, mask(base + step, site, 0))))
        for d in (1, 2, 3, 4, 5, 64, 768, 3072):
            worst["neighbours"] = max(worst["neighbours"], abs(corr(a[:-d], a[d:])))
    print("MEASURED dropout statistics, largest |z| per family:", {k: round(v, 2) for k, v in worst.items()})
    for k, v in worst.items():
        assert v < 5.0, (k, v)


def test_embedding_fwd_bwd():
    V, d, M = 1000, 128, 777
    table = bf(rnd((V, d), 30))
    g = torch.Generator().manual_seed(31)
    ids = torch.randint(0, V, (M,), generator=g).to(DEV)
    out = torch.empty(M, d, device=DEV, dtype=torch.bfloat16)
    K.embedding_fwd(ids, table, out)
    assert torch.equal(out, table[ids])
    dout = bf(ints((M, d), -3, 3, 32))
    dtable = torch.zeros(V, d, device=DEV)
    K.embedding_bwd(ids, dout, dtable)
    want = torch.zeros(V, d, device=DEV).index_add_(0, ids, dout.float())
    assert torch.equal(dtable, want)


# -------------------------------------------------------------------------------------- attention
def _bias_table(H, S, seed):
    return (rnd((H, 2 * S - 1), seed, 0.5)).to(DEV).contiguous()


def _dense_bias(tbl, S):
    idx = torch.arange(S, device=DEV)[None, :] - torch.arange(S, device=DEV)[:, None] + S - 1   # [q, k]
    return tbl[:, idx]                                                                       # [H, q, k]


def _ref_attention(q, k, v, bias, N, S, H, mask=None):
    qh = q.float().view(N, S, H, 64).transpose(1, 2)
    kh = k.float().view(N, S, H, 64).transpose(1, 2)
    vh = v.float().view(N, S, H, 64).transpose(1, 2)
    s = qh @ kh.transpose(2, 3) + bias[None]
    p = torch.softmax(s, dim=-1)
    lse = torch.logsumexp(s, dim=-1)
    if mask is not None:
        p = p * mask
    o = (p @ vh).transpose(1, 2).reshape(N * S, H * 64)
    return o, lse


# (S = 384, 512: the dK/dV kernel asks for more than 64 KB of LDS -- 39168 + 72 S bytes from S = 367 -- and needs the
# opt-in attribute; S = 600: beyond 512 tokens the bias-gradient sums share one LDS array through atomics)
@pytest.mark.parametrize("N,S,H", [(2, 16, 2), (1, 100, 3), (3, 128, 2), (2, 256, 12), (1, 333, 1), (1, 384, 2), (2, 512, 2), (1, 600, 1)])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_enc_attention_fwd_bwd(N, S, H, p):
    I = H * 64
    qkv = bf(rnd((N * S, 3 * I), 40, 0.35))
    q, k, v = qkv[:, :I], qkv[:, I:2 * I], qkv[:, 2 * I:]
    tbl = _bias_table(H, S, 41)
    ctx = torch.empty(N * S, I, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(N * H * S, device=DEV)
    seed = 99
    K.enc_attn_fwd(q, k, v, ctx, lse, tbl, N, S, H, drop_p=p, drop_seed=seed)
    mask = None
    if p > 0:
        S4 = (S + 3) // 4 * 4          # the kernels count keys in rows padded to a multiple of 4
        mask = K.dropout_mask(N * H * S * S4, drop_p=p, drop_seed=seed).view(N, H, S, S4)[..., :S]
    ql, kl, vl = (t.float().detach().clone().requires_grad_(True) for t in (q, k, v))
    tl = tbl.clone().requires_grad_(True)
    o_ref, lse_ref = _ref_attention(ql, kl, vl, _dense_bias(tl, S), N, S, H, mask)
    scale = o_ref.abs().max().item()
    # probabilities and the output are rounded to bf16 inside the kernel: 2^-8 relative steps
    assert (ctx.float() - o_ref).abs().max().item() < 1.5e-2 * scale
    np.testing.assert_allclose(lse.view(N, H, S).cpu().numpy(), lse_ref.detach().cpu().numpy(), rtol=1e-4, atol=2e-4)

    dctx = bf(rnd((N * S, I), 42))
    o_ref.backward(dctx.float())
    dqkv = torch.full((N * S, 3 * I), float("nan"), device=DEV, dtype=torch.bfloat16)
    dtbl3 = torch.zeros(5, H, 2 * S - 1, device=DEV)        # replicated accumulator
    K.enc_attn_bwd(q, k, v, ctx, dctx, lse, tbl, dqkv[:, :I], dqkv[:, I:2 * I], dqkv[:, 2 * I:], dtbl3, N, S, H,
                   drop_p=p, drop_seed=seed)
    dtbl = dtbl3.sum(0)
    for got, want, name in ((dqkv[:, :I], ql.grad, "dq"), (dqkv[:, I:2 * I], kl.grad, "dk"), (dqkv[:, 2 * I:], vl.grad, "dv")):
        err = (got.float() - want).abs().max().item()
        assert err < 2.5e-2 * want.abs().max().item(), (name, err, want.abs().max().item())
    err = (dtbl - tl.grad).abs().max().item()
    assert err < 2e-2 * tl.grad.abs().max().item() + 1e-3, ("dbias", err)


def _head_major(t, pad=0, fill=float("nan")):
    """[rows, features] -> head-major view [features/64, rows, 64] of a buffer with `pad` unused rows after every head."""
    R, F = t.shape
    buf = torch.full((F // 64, R + pad, 64), fill, device=t.device, dtype=t.dtype)
    buf[:, :R] = t.view(R, F // 64, 64).permute(1, 0, 2)
    return buf[:, :R], buf


@pytest.mark.parametrize("M,N,Kc,trans_b", [(40, 192, 64, False), (300, 384, 128, False), (1000, 2304, 768, False),
                                            (1000, 768, 256, True), (4099, 1536, 192, False)])
def test_gemm_head_major_output(M, N, Kc, trans_b):
    """Every tile configuration writes the same bits as the row-major product, as [N/64] contiguous [M, 64] matrices."""
    a = bf(rnd((M, Kc), 60))
    b = bf(rnd((Kc, N) if trans_b else (N, Kc), 61))
    want = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    K.gemm(a, b, want, trans_b=trans_b)
    got, buf = _head_major(torch.zeros_like(want), pad=24)
    buf.fill_(float("nan"))
    K.gemm(a, b, got, trans_b=trans_b)
    assert torch.equal(got.permute(1, 0, 2).reshape(M, N), want)
    assert torch.isnan(buf[:, M:].float()).all()        # nothing written between the heads
    with pytest.raises(ValueError):
        K.gemm(a, b, torch.empty(N // 64, M, 32, device=DEV, dtype=torch.bfloat16), trans_b=trans_b)


@pytest.mark.parametrize("M,N,Kc", [(40, 64, 128), (300, 256, 192), (1000, 768, 768), (4099, 768, 2304), (2500, 3072, 768)])
@pytest.mark.parametrize("epi", ["plain", "res_drop", "dx"])
def test_gemm_head_major_a(M, N, Kc, epi):
    """y = x W^T (+ epilogue) and dx = dy W with the activation operand head-major: the bits of the row-major call."""
    a = bf(rnd((M, Kc), 62))
    ah, _ = _head_major(a, pad=8)
    tb = epi == "dx"
    b = bf(rnd((Kc, N) if tb else (N, Kc), 63, 0.05))
    kw = dict(trans_b=tb)
    if epi == "res_drop":
        kw.update(epilogue=K.EPI_RES_DROP, aux=bf(rnd((M, N), 64)), drop_p=0.1, drop_seed=7)
    want = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    got = torch.empty_like(want)
    K.gemm(a, b, want, **kw)
    K.gemm(ah, b, got, **kw)
    assert torch.equal(got, want)


@pytest.mark.parametrize("T,O,Iin", [(40, 64, 128), (333, 256, 192), (1000, 2304, 768), (4099, 768, 768), (70000, 768, 768)])
@pytest.mark.parametrize("which", ["a", "b"])
def test_gemm_weight_gradient_with_a_head_major_operand(T, O, Iin, which):
    """dW = dy^T x (fp32 atomics) with dy (operand a) or x (operand b) head-major; ragged token counts included."""
    dy, x = bf(rnd((T, O), 65)), bf(rnd((T, Iin), 66))
    want = torch.zeros(O, Iin, device=DEV)
    K.gemm(dy, x, want, trans_a=True, trans_b=True, epilogue=K.EPI_ATOMIC_F32, splits=1)
    got = torch.zeros_like(want)
    # (a ragged last step of the contraction reads up to 63 rows past T: they must hold finite values -- here zeros and
    # the next head's rows -- and meet the other operand's zeros)
    dyh, _ = _head_major(dy, pad=16, fill=0.0)
    xh, _ = _head_major(x, pad=16, fill=0.0)
    K.gemm(dyh if which == "a" else dy, xh if which == "b" else x, got, trans_a=True, trans_b=True, epilogue=K.EPI_ATOMIC_F32, splits=1)
    assert torch.equal(got, want)                          # one split: no atomic reordering, so the same bits
    got.zero_()
    K.gemm(dyh if which == "a" else dy, xh if which == "b" else x, got, trans_a=True, trans_b=True, epilogue=K.EPI_ATOMIC_F32, splits=0)
    assert (got - want).abs().max().item() <= 2e-3 * want.abs().max().item()


@pytest.mark.parametrize("N,S,H", [(2, 16, 2), (1, 100, 3), (2, 256, 12), (1, 333, 1), (1, 384, 2), (2, 512, 2), (1, 600, 1)])
@pytest.mark.parametrize("p", [0.0, 0.1])
@pytest.mark.parametrize("what", ["qkv", "all"])
def test_enc_attention_head_major_operands_give_the_same_bits(N, S, H, p, what):
    I, M = H * 64, N * S
    qkv = bf(rnd((M, 3 * I), 46, 0.35))
    tbl = _bias_table(H, S, 47)
    dctx = bf(rnd((M, I), 48))
    ctx0 = torch.empty(M, I, device=DEV, dtype=torch.bfloat16)
    lse0 = torch.empty(N * H * S, device=DEV)
    K.enc_attn_fwd(qkv[:, :I], qkv[:, I:2 * I], qkv[:, 2 * I:], ctx0, lse0, tbl, N, S, H, drop_p=p, drop_seed=5)
    dqkv0 = torch.full((M, 3 * I), float("nan"), device=DEV, dtype=torch.bfloat16)
    dtbl0 = torch.zeros(1, H, 2 * S - 1, device=DEV)
    K.enc_attn_bwd(qkv[:, :I], qkv[:, I:2 * I], qkv[:, 2 * I:], ctx0, dctx, lse0, tbl, dqkv0[:, :I], dqkv0[:, I:2 * I], dqkv0[:, 2 * I:],
                   dtbl0, N, S, H, drop_p=p, drop_seed=5)

    heads, _ = _head_major(qkv, pad=8)                                   # [3H, M, 64]
    q, k, v = heads[:H], heads[H:2 * H], heads[2 * H:]
    lse1 = torch.empty_like(lse0)
    dtbl1 = torch.zeros_like(dtbl0)
    if what == "qkv":
        ctx1 = torch.empty_like(ctx0)
        K.enc_attn_fwd(q, k, v, ctx1, lse1, tbl, N, S, H, drop_p=p, drop_seed=5)
        dqkv1 = torch.full_like(dqkv0, float("nan"))
        K.enc_attn_bwd(q, k, v, ctx1, dctx, lse1, tbl, dqkv1[:, :I], dqkv1[:, I:2 * I], dqkv1[:, 2 * I:], dtbl1, N, S, H, drop_p=p, drop_seed=5)
        got_ctx, got_dqkv = ctx1, dqkv1
    else:
        ctx1, _ = _head_major(torch.zeros_like(ctx0), pad=8)
        K.enc_attn_fwd(q, k, v, ctx1, lse1, tbl, N, S, H, drop_p=p, drop_seed=5)
        dctx1, _ = _head_major(dctx, pad=8)
        dqkv1, buf = _head_major(torch.zeros_like(dqkv0), pad=8)
        buf.fill_(float("nan"))
        K.enc_attn_bwd(q, k, v, ctx1, dctx1, lse1, tbl, dqkv1[:H], dqkv1[H:2 * H], dqkv1[2 * H:], dtbl1, N, S, H, drop_p=p, drop_seed=5)
        got_ctx, got_dqkv = ctx1.permute(1, 0, 2).reshape(M, I), dqkv1.permute(1, 0, 2).reshape(M, 3 * I)
    assert torch.equal(got_ctx, ctx0) and torch.equal(lse1, lse0) and torch.equal(got_dqkv, dqkv0)
    assert (dtbl0 - dtbl1).abs().max().item() <= 1e-5 * dtbl0.abs().max().item()      # order of the atomic adds


def test_enc_attention_64_bit_dropout_numbering_gives_the_same_masks():
    """The encoder attention numbers its dropout quads with 32 bits up to 2^34 probabilities and with 64 bits beyond;
    PACED_ATTN_WIDE_DROPOUT=1 (read once per process) forces the 64-bit kernels, so a child process runs the dropout
    cases of test_enc_attention_fwd_bwd, which replay the masks through the library's mask function, under it."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, PACED_ATTN_WIDE_DROPOUT="1")
    cmd = [sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider",
           "-k", "test_enc_attention_fwd_bwd or test_enc_attention_head_major"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    tail = r.stdout.decode()[-1500:]
    assert r.returncode == 0 and " passed" in tail and "failed" not in tail, tail


def test_enc_attention_rescale_branch_forced():
    """One key far above the rest in a late tile forces the online-softmax running-max update."""
    N, S, H, I = 1, 192, 1, 64
    q = rnd((S, I), 50, 0.2)
    k = rnd((S, I), 51, 0.2)
    v = rnd((S, I), 52)
    k[150] = q[7] * 40.0                      # query 7 . key 150 is huge; other rows see it as ordinary noise * 40
    qkv = bf(torch.cat([q, k, v], dim=1))
    ctx = torch.empty(S, I, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(S, device=DEV)
    tbl = torch.zeros(1, 2 * S - 1, device=DEV)
    K.enc_attn_fwd(qkv[:, :I], qkv[:, I:2 * I], qkv[:, 2 * I:], ctx, lse, tbl, N, S, H)
    o_ref, lse_ref = _ref_attention(qkv[:, :I], qkv[:, I:2 * I], qkv[:, 2 * I:], _dense_bias(tbl, S), N, S, H)
    assert (ctx.float() - o_ref).abs().max().item() < 2e-2 * o_ref.abs().max().item()
    np.testing.assert_allclose(lse.cpu().numpy(), lse_ref.view(-1).cpu().numpy(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_decoder_self_attention(p):
    N, H = 37, 3
    I = H * 64
    qkv = bf(rnd((N * 2, 3 * I), 60, 0.3))
    bias2 = rnd((H, 2), 61, 0.5).to(DEV).contiguous()       # [:,0] offset 0, [:,1] offset -1
    ctx = torch.empty(N * 2, I, device=DEV, dtype=torch.bfloat16)
    K.dec_self_attn_fwd(qkv, ctx, bias2, N, H, drop_p=p, drop_seed=5)
    mask = K.dropout_mask(N * H * 4, drop_p=p, drop_seed=5).view(N, H, 2, 2) if p > 0 else torch.ones(N, H, 2, 2, device=DEV)
    x = qkv.float().detach().clone().requires_grad_(True)
    b2 = bias2.clone().requires_grad_(True)
    qh, kh, vh = (x[:, i * I:(i + 1) * I].view(N, 2, H, 64).transpose(1, 2) for i in range(3))
    bias = torch.stack([torch.stack([b2[:, 0], b2[:, 0]], -1), torch.stack([b2[:, 1], b2[:, 0]], -1)], -2)   # [H, q, k]
    s = qh @ kh.transpose(2, 3) + bias[None]
    s = s.masked_fill(torch.tensor([[False, True], [False, False]], device=DEV), float("-inf"))
    o = ((torch.softmax(s, -1) * mask) @ vh).transpose(1, 2).reshape(N * 2, I)
    assert (ctx.float() - o).abs().max().item() < 1e-2 * o.abs().max().item()
    dctx = bf(rnd((N * 2, I), 62))
    o.backward(dctx.float())
    dqkv = torch.empty_like(qkv)
    db2 = torch.zeros(H, 2, device=DEV)
    K.dec_self_attn_bwd(qkv, dctx, dqkv, bias2, db2, N, H, drop_p=p, drop_seed=5)
    assert (dqkv.float() - x.grad).abs().max().item() < 1.5e-2 * x.grad.abs().max().item()
    np.testing.assert_allclose(db2.cpu().numpy(), b2.grad.cpu().numpy(), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("N,S,H", [(3, 16, 2), (2, 101, 3), (5, 256, 12)])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_cross_attention(N, S, H, p):
    I = H * 64
    qc = bf(rnd((N * 2, I), 70, 0.3))
    kv = bf(rnd((N * S, 2 * I), 71, 0.4))
    ctx = torch.empty(N * 2, I, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(N * H * 2, device=DEV)
    K.cross_attn_fwd(qc, kv, ctx, lse, N, S, H, drop_p=p, drop_seed=8)
    mask = K.dropout_mask(N * H * 2 * S, drop_p=p, drop_seed=8).view(N, H, 2, S) if p > 0 else 1.0
    ql = qc.float().detach().clone().requires_grad_(True)
    kvl = kv.float().detach().clone().requires_grad_(True)
    qh = ql.view(N, 2, H, 64).transpose(1, 2)
    kh = kvl[:, :I].view(N, S, H, 64).transpose(1, 2)
    vh = kvl[:, I:].view(N, S, H, 64).transpose(1, 2)
    s = qh @ kh.transpose(2, 3)
    o = ((torch.softmax(s, -1) * mask) @ vh).transpose(1, 2).reshape(N * 2, I)
    assert (ctx.float() - o).abs().max().item() < 1e-2 * o.abs().max().item()
    np.testing.assert_allclose(lse.view(N, H, 2).cpu().numpy(), torch.logsumexp(s, -1).detach().cpu().numpy(), rtol=1e-4, atol=1e-4)
    dctx = bf(rnd((N * 2, I), 72))
    o.backward(dctx.float())
    dqc = torch.empty_like(qc)
    dkv = torch.full_like(kv, float("nan"))
    K.cross_attn_bwd(qc, kv, ctx, dctx, lse, dqc, dkv, N, S, H, drop_p=p, drop_seed=8)
    assert (dqc.float() - ql.grad).abs().max().item() < 1.5e-2 * ql.grad.abs().max().item()
    assert (dkv.float() - kvl.grad).abs().max().item() < 1.5e-2 * kvl.grad.abs().max().item()


@pytest.mark.parametrize("N,S,H,d", [(3, 16, 2, 128), (2, 101, 3, 192), (5, 256, 12, 768), (2, 64, 16, 1024)])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_cross_attention_without_kv_projection(N, S, H, d, p):
    """s = qk . e, z = dropout(softmax(s)) e over the raw encoder states, against fp32 autograd on the same bf16 inputs."""
    C = 2 * H
    enc = bf(rnd((N * S, d), 80, 0.5))
    qk = bf(rnd((N * 2, H * d), 81, 0.12))
    z = torch.empty(N * 2, H * d, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(N * C, device=DEV)
    K.cross2_fwd(enc, qk, z, lse, N, S, H, drop_p=p, drop_seed=9)
    S4 = (S + 3) // 4 * 4
    mask = K.dropout_mask(N * C * S4, drop_p=p, drop_seed=9).view(N, C, S4)[..., :S] if p > 0 else 1.0
    el = enc.float().detach().clone().requires_grad_(True)
    ql = qk.float().detach().clone().requires_grad_(True)
    e3, q3 = el.view(N, S, d), ql.view(N, C, d)
    s = q3 @ e3.transpose(1, 2)                                   # [N, C, S]
    zr = (torch.softmax(s, -1) * mask) @ e3                       # [N, C, d]
    assert (z.float().view(N, C, d) - zr).abs().max().item() < 1e-2 * zr.abs().max().item()
    np.testing.assert_allclose(lse.view(N, C).cpu().numpy(), torch.logsumexp(s, -1).detach().cpu().numpy(), rtol=1e-4, atol=1e-4)
    dz = bf(rnd((N * 2, H * d), 82))
    zr.backward(dz.float().view(N, C, d))
    dqk = torch.empty_like(qk)
    L, l = 2, 1                                                    # this call fills layer slot 1 of 2
    S8 = (S + 7) // 8 * 8                                          # row stride of the score factors (any S)
    W = torch.full((N, L * 2 * C, S8), float("nan"), device=DEV, dtype=torch.bfloat16)
    U = torch.full((N, L * 2 * C, d), float("nan"), device=DEV, dtype=torch.bfloat16)
    K.cross2_bwd(enc, qk, dz, z, lse, dqk, W[0, l * 2 * C:], L * 2 * C * S8, U[0, l * 2 * C:], L * 2 * C * d, N, S, H,
                 wld=S8, drop_p=p, drop_seed=9)
    assert (dqk.float() - ql.grad).abs().max().item() < 1.5e-2 * ql.grad.abs().max().item()
    assert torch.isnan(W[:, :2 * C].float()).all() and torch.isnan(U[:, :2 * C].float()).all()      # other slots untouched
    assert torch.isnan(W[:, 2 * C:, S:].float()).all()             # columns past S are left alone
    Wl, Ul = W[:, 2 * C:, :S].float(), U[:, 2 * C:].float()
    assert torch.equal(Ul[:, :C], qk.float().view(N, C, d)) and torch.equal(Ul[:, C:], dz.float().view(N, C, d))
    de = torch.einsum("nks,nkd->nsd", Wl, Ul).reshape(N * S, d)    # the product the engine takes over all layers
    assert (de - el.grad).abs().max().item() < 2e-2 * el.grad.abs().max().item()


# ------------------------------------------------------------------------------------ loss / optimiser
def test_cross_entropy_and_lce():
    R, V = 24, 32128
    logits = (rnd((R, V), 80, 2.0)).to(DEV).contiguous()
    labels = torch.tensor([1176, 1] * (R // 2)).to(DEV)
    labels[5] = -100
    loss = torch.empty(R, device=DEV)
    dl = torch.empty(R, V, device=DEV, dtype=torch.bfloat16)
    K.cross_entropy(logits, labels, loss, dl, gscale=0.25)
    ll = logits.clone().requires_grad_(True)
    want = torch.nn.functional.cross_entropy(ll, labels, ignore_index=-100, reduction="none")
    np.testing.assert_allclose(loss.cpu().numpy(), want.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
    (want.sum() * 0.25).backward()
    assert (dl.float() - ll.grad).abs().max().item() < 2 ** -8 * ll.grad.abs().max().item() + 1e-7
    assert (dl[5] == 0).all()
    # grouping: rows r of pce with n consecutive entries of nce
    B, n = 6, 3
    pce, nce = rnd((B * 2,), 81).abs().to(DEV), rnd((B * n * 2,), 82).abs().to(DEV)
    ce = torch.empty(B * 2, device=DEV)
    K.lce_group(pce, nce, ce, n, False)
    assert torch.allclose(ce, pce + nce.view(-1, n).sum(1))
    K.lce_group(pce, nce, ce, n, True)
    assert torch.allclose(ce, pce + nce.view(-1, n).mean(1))


def test_cross_entropy_narrow_rows():
    """Fewer columns than threads: partial (max, sum) pairs of idle threads must not poison the reduction."""
    logits = rnd((5, 8), 55, 3.0).to(DEV)
    labels = torch.tensor([0, 3, 1, 2, 0], device=DEV)
    loss = torch.empty(5, device=DEV)
    K.cross_entropy(logits[:, :4], labels, loss)
    want = torch.nn.functional.cross_entropy(logits[:, :4], labels, reduction="none")
    np.testing.assert_allclose(loss.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_lce_weights_known_answer():
    loss = torch.tensor([0.1, 0.5, 2.0, 0.3], device=DEV)
    eta = torch.tensor([0.4], device=DEV)
    w = torch.empty(4, device=DEV)
    stats = torch.empty(4, device=DEV)
    K.lce_weights(loss, eta, w, stats)
    np.testing.assert_allclose(w.cpu().numpy(), [0.25, 0.0, 0.0, 0.75], rtol=1e-6)
    np.testing.assert_allclose(stats.cpu().numpy(), [0.25, (0.025 + 0.225) / 4, -0.15625, 0.725], rtol=1e-5)


def test_adamw_matches_hf_rule():
    n = 1003
    p0, g = rnd((n,), 90), rnd((n,), 91, 1e-3)
    p = p0.to(DEV).clone()
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pb = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    pr, mr, vr = p0.double().clone(), torch.zeros(n).double(), torch.zeros(n).double()
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-6
    for t in range(1, 4):
        step = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        gd = g.to(DEV)
        K.adamw(p, gd, m, v, pb, step_size=step, clear_grad=(t == 2))
        assert torch.equal(gd, torch.zeros_like(gd) if t == 2 else g.to(DEV))      # cleared only when asked, after it was read
        mr = b1 * mr + (1 - b1) * g.double()
        vr = b2 * vr + (1 - b2) * g.double() ** 2
        pr = pr - step * mr / (vr.sqrt() + eps)
    np.testing.assert_allclose(p.cpu().numpy(), pr.float().numpy(), rtol=1e-5, atol=1e-6)
    assert torch.equal(pb, p.to(torch.bfloat16))


def test_relbias_tables_replicas():
    H, S = 2, 9
    g = torch.Generator().manual_seed(98)
    bucket = torch.randint(0, 32, (2 * S - 1,), generator=g).int().to(DEV)
    dt = ints((3, H, 2 * S - 1), -2, 2, 99).to(DEV)
    drel = torch.zeros(32, H, device=DEV)
    K.relbias_table_bwd(dt, bucket, drel)
    want = torch.zeros(32, H, device=DEV).index_add_(0, bucket.long(), dt.sum(0).t().contiguous())
    assert torch.equal(drel, want)


def test_relbias_tables():
    H, S = 4, 50
    rel = bf(rnd((32, H), 95))
    g = torch.Generator().manual_seed(96)
    bucket = torch.randint(0, 32, (2 * S - 1,), generator=g).int().to(DEV)
    tbl = torch.empty(H, 2 * S - 1, device=DEV)
    K.relbias_table_fwd(rel, bucket, tbl)
    assert torch.equal(tbl, rel.float()[bucket.long()].t().contiguous())
    dt = ints((H, 2 * S - 1), -2, 2, 97).to(DEV)
    drel = torch.zeros(32, H, device=DEV)
    K.relbias_table_bwd(dt, bucket, drel)
    want = torch.zeros(32, H, device=DEV).index_add_(0, bucket.long(), dt.t().contiguous())
    assert torch.equal(drel, want)


# ------------------------------------------------------------------------------------------ mining
def test_pool_topk_golden(golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "mining.npz"))
    k = int(g["k"][0])
    s, i = K.pool_topk(bf(torch.from_numpy(g["q"])), bf(torch.from_numpy(g["p"])), k)
    assert np.array_equal(i.cpu().numpy(), g["idx"])
    assert np.array_equal(s.cpu().numpy(), g["scores"])


@pytest.mark.parametrize("Q,Np,D,k", [(40, 50000, 128, 100), (130, 20001, 64, 1000), (7, 300, 64, 1000)])
def test_pool_topk_exact_vs_oracle(Q, Np, D, k):
    from oracle import mining_ref
    rng = np.random.RandomState(3)
    q = rng.randint(-8, 9, size=(Q, D)).astype(np.float32)
    p = rng.randint(-8, 9, size=(Np, D)).astype(np.float32)
    p[Np // 2:] = np.abs(p[Np // 2:])           # later rows score differently: thresholds keep moving
    s, i = K.pool_topk(bf(torch.from_numpy(q)), bf(torch.from_numpy(p)), k, index_base=0)
    ws, wi = mining_ref.pool_topk(q, p, min(k, Np))
    kk = min(k, Np)
    assert np.array_equal(i.cpu().numpy()[:, :kk], wi)
    assert np.array_equal(s.cpu().numpy()[:, :kk], ws)
    if kk < k:
        assert (i.cpu().numpy()[:, kk:] == -1).all()


def test_pool_topk_when_every_row_beats_the_threshold():
    """Scores that grow with the row index: every row of every chunk passes the filter (the most slot reservations the
    epilogue can see, lists filled to capacity before each trim), and ties between neighbouring rows exercise the
    (score desc, row asc) order."""
    from oracle import mining_ref
    Q, Np, D, k = 300, 150000, 64, 257
    q = np.zeros((Q, D), dtype=np.float32)
    q[:, 0] = 1.0
    q[::3, 1] = 1.0                                            # a third of the queries sees a different ranking
    p = np.zeros((Np, D), dtype=np.float32)
    p[:, 0] = np.arange(Np) // 1024                            # bf16-exact (< 256), constant over runs of 1024 rows: ties
    p[:, 1] = (np.arange(Np) % 7).astype(np.float32)
    s, i = K.pool_topk(bf(torch.from_numpy(q)), bf(torch.from_numpy(p)), k)
    ws, wi = mining_ref.pool_topk(q, p, k)
    assert np.array_equal(i.cpu().numpy(), wi)
    assert np.array_equal(s.cpu().numpy(), ws)


@pytest.mark.parametrize("Q,D", [(64, 768), (33, 128), (1, 64), (100, 256), (128, 768)])
@pytest.mark.parametrize("order", ["ascending", "random"])
def test_pool_topk_few_queries_streaming_kernel(Q, D, order):
    """Up to 128 queries the sweep takes the HBM-streaming kernel (queries in registers, pool rows through an LDS ring, survivors
    collected in an LDS list and flushed in bulk; 4 waves up to 64 queries, 8 beyond).  Pools long enough for several growing
    chunks and many row blocks per workgroup, a ragged last block, and -- "ascending" -- scores that grow with the row index,
    so that every row of every chunk passes and the list overflows into flush after flush.  Bit-exact against the oracle
    and against the tiled kernel."""
    from oracle import mining_ref
    Np, k = 70001, 300
    rng = np.random.RandomState(Q)
    q = rng.randint(-4, 5, size=(Q, D)).astype(np.float32)
    p = rng.randint(-4, 5, size=(Np, D)).astype(np.float32)
    if order == "ascending":
        q[:, 0] = 1.0
        p[:, 0] = np.arange(Np) // 512                      # bf16-exact (< 256); ties inside runs of 512 rows
    qd, pd = bf(torch.from_numpy(q)), bf(torch.from_numpy(p))
    s, i = K.pool_topk(qd, pd, k, index_base=5)
    ws, wi = mining_ref.pool_topk(q, p, k)
    assert np.array_equal(i.cpu().numpy() - 5, wi)
    assert np.array_equal(s.cpu().numpy(), ws)


def test_topk_merge_of_shards():
    from oracle import mining_ref
    rng = np.random.RandomState(4)
    Q, Np, D, k, W = 9, 4000, 64, 64, 4
    q = rng.randint(-8, 9, size=(Q, D)).astype(np.float32)
    p = rng.randint(-8, 9, size=(Np, D)).astype(np.float32)
    parts_s, parts_i = [], []
    for r in range(W):
        lo, hi = r * Np // W, (r + 1) * Np // W
        s, i = K.pool_topk(bf(torch.from_numpy(q)), bf(torch.from_numpy(p[lo:hi])), k, index_base=lo)
        parts_s.append(s); parts_i.append(i)
    ms, mi = K.topk_merge(torch.cat(parts_s, 1).contiguous(), torch.cat(parts_i, 1).contiguous(), k)
    ws, wi = mining_ref.pool_topk(q, p, k)
    assert np.array_equal(mi.cpu().numpy(), wi) and np.array_equal(ms.cpu().numpy(), ws)

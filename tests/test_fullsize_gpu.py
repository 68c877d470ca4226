"""Property checks at the sizes the benchmark runs (the oracle does not finish there in seconds): exact identities that
hold for any size -- selection products, checksums, sortedness, idempotence -- through the C ABI on the GPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
ROWS = 32 * 17 * 256          # the benchmark's packed token rows: 32 tuples x (1 + 16) sequences x 256 tokens


@pytest.fixture(scope="module")
def K():
    from pacednegatives_amd import kernels
    return kernels


def _ints(shape, lo, hi, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, device=DEV, generator=g).to(torch.bfloat16)


@pytest.mark.parametrize("N,Kc,epi", [(768, 768, "bf16"), (2304, 768, "bf16"), (768, 3072, "res")])
def test_projection_of_selector_rows_is_a_gather(K, N, Kc, epi):
    """A = one signed unit entry per row: C[m, :] must be +-B[:, col(m)] exactly, for every one of the 139264 rows and
    every tile of the persistent grid (bf16 output, with and without the residual epilogue)."""
    g = torch.Generator(device=DEV).manual_seed(1)
    col = torch.randint(0, Kc, (ROWS,), device=DEV, generator=g)
    sign = torch.randint(0, 2, (ROWS,), device=DEV, generator=g).float() * 2 - 1
    a = torch.zeros(ROWS, Kc, device=DEV, dtype=torch.bfloat16)
    a[torch.arange(ROWS, device=DEV), col] = sign.to(torch.bfloat16)
    b = _ints((N, Kc), -100, 100, 2)
    c = torch.empty(ROWS, N, device=DEV, dtype=torch.bfloat16)
    want = b.t()[col].float() * sign[:, None]
    if epi == "res":
        res = _ints((ROWS, N), -100, 100, 3)
        K.gemm(a, b, c, epilogue=K.EPI_RES_DROP, aux=res)
        want = want + res.float()
    else:
        K.gemm(a, b, c)
    assert torch.equal(c.float(), want)


def test_projection_checksums_in_fp32(K):
    """Integer operands, fp32 output: every row sum and every column sum of C equals the product with the summed
    operand (exact in fp32: all partial sums stay below 2^24)."""
    N, Kc = 768, 768
    a, b = _ints((ROWS, Kc), -2, 2, 4), _ints((N, Kc), -2, 2, 5)
    c = torch.empty(ROWS, N, device=DEV)
    K.gemm(a, b, c, epilogue=K.EPI_F32)
    row_want = a.double() @ b.double().sum(0)
    col_want = a.double().sum(0) @ b.double().t()
    assert torch.equal(c.double().sum(1), row_want)
    assert torch.equal(c.double().sum(0), col_want)


def test_weight_gradient_over_all_rows_is_linear_in_the_row_blocks(K):
    """dW = dY^T X over the 139264 token rows (fp32 atomics, K split over workgroups) equals the sum of the products
    over two halves of the rows, exactly, for integer operands."""
    dy, x = _ints((ROWS, 768), -2, 2, 6), _ints((ROWS, 768), -2, 2, 7)
    whole = torch.zeros(768, 768, device=DEV)
    K.gemm(dy, x, whole, trans_a=True, trans_b=True, epilogue=K.EPI_ATOMIC_F32, splits=0)
    parts = torch.zeros(768, 768, device=DEV)
    h = ROWS // 2
    K.gemm(dy[:h], x[:h], parts, trans_a=True, trans_b=True, epilogue=K.EPI_ATOMIC_F32, splits=0)
    K.gemm(dy[h:], x[h:], parts, trans_a=True, trans_b=True, epilogue=K.EPI_ATOMIC_F32, splits=0)
    assert torch.equal(whole, parts)
    assert torch.equal(whole.double(), dy.double().t() @ x.double())


def test_rmsnorm_rows_are_independent_of_the_batch(K):
    """Row m of the normalised output depends on row m only: the 139264-row launch and a launch over a slice agree bit
    for bit, forward and backward (input gradient)."""
    d = 768
    g = torch.Generator(device=DEV).manual_seed(8)
    x = torch.randn(ROWS, d, device=DEV, generator=g).to(torch.bfloat16)
    dy = torch.randn(ROWS, d, device=DEV, generator=g).to(torch.bfloat16)
    w = (1 + 0.1 * torch.randn(d, device=DEV, generator=g)).to(torch.bfloat16)
    y, rstd, dx = torch.empty_like(x), torch.empty(ROWS, device=DEV), torch.empty_like(x)
    K.rmsnorm_fwd(x, w, y, rstd)
    K.rmsnorm_bwd(dy, x, w, rstd, dx)
    lo, hi = 70001, 70001 + 333
    ys, rs, dxs = torch.empty(hi - lo, d, device=DEV, dtype=torch.bfloat16), torch.empty(hi - lo, device=DEV), torch.empty(hi - lo, d, device=DEV, dtype=torch.bfloat16)
    K.rmsnorm_fwd(x[lo:hi].contiguous(), w, ys, rs)
    K.rmsnorm_bwd(dy[lo:hi].contiguous(), x[lo:hi].contiguous(), w, rs, dxs)
    assert torch.equal(y[lo:hi], ys) and torch.equal(rstd[lo:hi], rs) and torch.equal(dx[lo:hi], dxs)


def test_encoder_attention_is_independent_of_the_other_sequences(K):
    """544 sequences x 12 heads x 256 tokens in one launch: sequence n's context equals the one-sequence launch."""
    N, S, H = 544, 256, 12
    I = H * 64
    g = torch.Generator(device=DEV).manual_seed(9)
    qkv = (torch.randn(N * S, 3 * I, device=DEV, generator=g) * 0.3).to(torch.bfloat16)
    tbl = torch.randn(H, 2 * S - 1, device=DEV, generator=g) * 0.3
    ctx, lse = torch.empty(N * S, I, device=DEV, dtype=torch.bfloat16), torch.empty(N * H * S, device=DEV)
    K.enc_attn_fwd(qkv[:, :I], qkv[:, I:2 * I], qkv[:, 2 * I:], ctx, lse, tbl, N, S, H)
    for n in (0, 271, 543):
        one = qkv[n * S:(n + 1) * S].contiguous()
        c1, l1 = torch.empty(S, I, device=DEV, dtype=torch.bfloat16), torch.empty(H * S, device=DEV)
        K.enc_attn_fwd(one[:, :I], one[:, I:2 * I], one[:, 2 * I:], c1, l1, tbl, 1, S, H)
        assert torch.equal(ctx[n * S:(n + 1) * S], c1)
        assert torch.equal(lse.view(N, H * S)[n], l1)


def test_pool_topk_at_ten_million_rows(K):
    """10 M x 768 pool, 4096 queries, k = 1000 (the mining configuration): results sorted by (score desc, row asc),
    rows distinct and in range, scores equal to the exact products of the returned rows (integer-valued embeddings),
    a sample of the pool never beats a query's k-th score, and selecting again among the returned rows returns them."""
    Q, Np, D, k = 4096, 10_000_000, 768, 1000
    pool = torch.empty(Np, D, dtype=torch.bfloat16, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(10)
    step = 1 << 20
    for lo in range(0, Np, step):
        hi = min(Np, lo + step)
        pool[lo:hi] = torch.randint(-3, 4, (hi - lo, D), device=DEV, generator=g).to(torch.bfloat16)
    queries = torch.randint(-3, 4, (Q, D), device=DEV, generator=g).to(torch.bfloat16)
    s, i = K.pool_topk(queries, pool, k)
    assert s.shape == (Q, k) and i.shape == (Q, k)
    assert int(i.min()) >= 0 and int(i.max()) < Np
    ds = s[:, 1:] - s[:, :-1]
    assert bool((ds <= 0).all())
    assert bool(((ds < 0) | (i[:, 1:] > i[:, :-1])).all())                      # ties broken by ascending row; rows distinct
    for q0 in range(0, Q, 512):                                                 # exact scores of the returned rows
        rows = pool[i[q0:q0 + 512].reshape(-1).long()].view(512, k, D).float()
        exact = torch.einsum("qkd,qd->qk", rows, queries[q0:q0 + 512].float())
        assert torch.equal(exact, s[q0:q0 + 512])
    sample = torch.randint(0, Np, (20000,), device=DEV, generator=g)
    sc = queries.float() @ pool[sample].float().t()                             # [Q, 20000]
    kth, kth_row = s[:, -1:], i[:, -1:]
    beats = (sc > kth) | ((sc == kth) & (sample[None, :] < kth_row))
    # a sampled row that beats the k-th entry must be among the returned rows
    qq, jj = beats.nonzero(as_tuple=True)
    if qq.numel():
        hit = (i[qq] == sample[jj][:, None]).any(1)
        assert bool(hit.all())
    # idempotence: the top-k of the returned rows alone is the same list
    for q0 in (0, 1777, 4095):
        sub = pool[i[q0].long()].contiguous()
        s2, i2 = K.pool_topk(queries[q0:q0 + 1].contiguous(), sub, k)
        assert torch.equal(s2[0], s[q0])
        assert torch.equal(i[q0][i2[0].long()], i[q0])

"""paced_pool_topk_static (the selection entry point that allocates nothing) against paced_pool_topk, through the C ABI:
bit-exact scores and indices, two calls in a row on the same GPU, the second with a smaller input (the built-in
candidate array still holds the first call's lists)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(Q, Np, D, k, seed, integer):
    g = torch.Generator().manual_seed(seed)
    if integer:         # exact scores with many ties: the (score desc, row asc) order is fully exercised
        pool = torch.randint(-8, 9, (Np, D), generator=g).to(torch.bfloat16)
        qry = torch.randint(-8, 9, (Q, D), generator=g).to(torch.bfloat16)
    else:
        pool = (torch.randn(Np, D, generator=g) * 0.05).to(torch.bfloat16)
        qry = (torch.randn(Q, D, generator=g) * 0.05).to(torch.bfloat16)
    return qry.cuda(), pool.cuda()


def test_static_selection_equals_the_workspace_one_twice_in_a_row():
    from pacednegatives_amd import _lib, kernels as K
    lib = _lib.load()
    assert lib.paced_pool_topk_static_max_queries() == 64 and lib.paced_pool_topk_static_max_k() == 1024
    assert 16 << 20 < lib.paced_pool_topk_static_bytes() < 32 << 20
    # first call: the limits themselves, enough rows for several chunks of the static sweep (32768 rows each);
    # second call: fewer queries, smaller k, a pool shorter than the first chunk of the first call's lists
    # warm-up: module load and the per-device symbol lookup happen on the first call; after it, a call must leave the
    # device's free memory (hipMemGetInfo: sees any hipMalloc, torch's or the library's) where it found it
    wq, wp = _case(2, 100, 64, 4, 3, True)
    K.pool_topk_static(wq, wp, 4, torch.empty(2, 4, device="cuda"), torch.empty(2, 4, dtype=torch.int64, device="cuda"))
    torch.cuda.synchronize()
    for (Q, Np, D, k, integer, base) in ((64, 150_000, 128, 1024, True, 0), (5, 3000, 64, 37, True, 1000), (17, 70_000, 256, 300, False, 7)):
        qry, pool = _case(Q, Np, D, k, 11 + Q, integer)
        want_s, want_i = K.pool_topk(qry, pool, k, index_base=base)
        got_s = torch.full((Q, k), float("nan"), device="cuda")
        got_i = torch.full((Q, k), -7, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        before, free_before = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
        K.pool_topk_static(qry, pool, k, got_s, got_i, index_base=base)
        assert torch.cuda.memory_allocated() == before
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before, "the device's free memory changed across the call"
        assert torch.equal(got_i, want_i), (Q, Np, k)
        assert torch.equal(got_s, want_s), (Q, Np, k)
        assert int(got_i.min()) >= base


def test_static_selection_pads_short_pools_and_rejects_sizes_beyond_its_array():
    from pacednegatives_amd import _lib, kernels as K
    qry, pool = _case(3, 20, 64, 32, 5, True)
    s, i = torch.empty(3, 32, device="cuda"), torch.empty(3, 32, dtype=torch.int64, device="cuda")
    K.pool_topk_static(qry, pool, 32, s, i)
    ws, wi = K.pool_topk(qry, pool, 32)
    assert torch.equal(i, wi) and torch.equal(s, ws) and int((i < 0).sum()) == 3 * 12
    qry, pool = _case(65, 500, 64, 8, 6, True)
    with pytest.raises(_lib.PacedHipError, match="at most 64 queries"):
        K.pool_topk_static(qry, pool, 8, torch.empty(65, 8, device="cuda"), torch.empty(65, 8, dtype=torch.int64, device="cuda"))
    qry, pool = _case(2, 5000, 64, 1025, 7, True)
    with pytest.raises(_lib.PacedHipError, match=r"k must be in \[1, 1024\]"):
        K.pool_topk_static(qry, pool, 1025, torch.empty(2, 1025, device="cuda"), torch.empty(2, 1025, dtype=torch.int64, device="cuda"))

"""Multi-process runs of the HIP path on one GPU: W ranks share cuda:0 and exchange over gloo (RCCL refuses two
ranks on one device; the collectives' call sites and the sum-then-scale rule are the same under either backend).

What Lightning's DDP strategy promises the reference (train/train_lce.py:89-91): the averaged gradients of W ranks on
shares of a batch equal the gradients of one process on the whole batch.  Every tuple's loss rows depend on that
tuple alone (pacednegatives/lceT5.py:116-123), so rows, weights, eta and the updated parameters must agree too.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_world(world, out_dir, **extra_env):
    os.makedirs(out_dir, exist_ok=True)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PACED_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, "-X", "faulthandler", WORKER, out_dir], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError(f"world {world}: a rank did not finish in 240 s")
        outs.append(o.decode(errors="replace"))
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} of {world} failed:\n{o[-3000:]}"
    return [np.load(os.path.join(out_dir, f"rank{r}.npz")) for r in range(world)]


@pytest.fixture(scope="module")
def runs(tmp_path_factory):
    base = tmp_path_factory.mktemp("dist")
    return _run_world(1, str(base / "w1"))[0], _run_world(2, str(base / "w2"))


def test_two_rank_step_equals_one_rank_on_the_whole_batch(runs):
    one, two = runs
    for it in range(2):
        # per-tuple quantities: rank r holds the rows of its contiguous share
        for key in ("rows", "main_rows", "weights"):
            got = np.concatenate([two[r][f"{key}{it}"] for r in range(2)])
            np.testing.assert_allclose(got, one[f"{key}{it}"], rtol=2e-3, atol=1e-6, err_msg=f"{key}{it}")
        # replicated quantities: identical on both ranks, equal to the single process
        assert two[0][f"eta{it}"] == two[1][f"eta{it}"]
        np.testing.assert_allclose(two[0][f"eta{it}"], one[f"eta{it}"], rtol=1e-5)
        ga, gb, g1 = two[0][f"grad{it}"], two[1][f"grad{it}"], one[f"grad{it}"]
        assert np.array_equal(ga, gb), "all-reduced gradient buffers differ between the ranks"
        assert np.linalg.norm(g1) > 0
        l2 = np.linalg.norm(ga - g1) / np.linalg.norm(g1)
        print(f"step {it}: |g(W=2) - g(W=1)| / |g(W=1)| = {l2:.3e}")
        assert l2 < 1e-6            # measured 4e-8: fp32 atomics in another order
    assert np.array_equal(two[0]["p32"], two[1]["p32"]), "ranks hold different parameters after two updates"
    # AdamW's first steps move every weight by ~lr whatever the gradient's size, so entries whose tiny gradient
    # flips sign between the two summation orders differ by up to 2 lr per step; they must be rare
    diff = np.abs(two[0]["p32"] - one["p32"])
    assert diff.max() <= 4.2e-3 and (diff > 1e-4).mean() < 0.02, (diff.max(), (diff > 1e-4).mean())


def test_segment_wise_exchange_and_update_equals_one_rank(tmp_path):
    """From the second step on, the decoder's segments and each encoder layer are all-reduced and updated as soon as
    their gradients are final, on a side stream under the rest of the backward pass, and only the remainder (tied
    embedding, encoder bias table and final norm) after it.  Three steps of two ranks on halves of the batch against one
    rank on the whole batch (which takes the same early updates without an exchange), and against the schedule that
    exchanges and updates everything at the end of the step."""
    one = _run_world(1, str(tmp_path / "w1"), PACED_TEST_EARLY="1")[0]
    two = _run_world(2, str(tmp_path / "w2"), PACED_TEST_EARLY="1")
    late = _run_world(2, str(tmp_path / "w2late"), PACED_TEST_EARLY="1", PACED_EARLY_UPDATE="0")
    for it in range(3):
        for key in ("rows", "main_rows", "weights"):
            got = np.concatenate([two[r][f"{key}{it}"] for r in range(2)])
            np.testing.assert_allclose(got, one[f"{key}{it}"], rtol=2e-3, atol=1e-6, err_msg=f"{key}{it}")
            np.testing.assert_allclose(got, np.concatenate([late[r][f"{key}{it}"] for r in range(2)]), rtol=2e-3, atol=1e-6)
        assert two[0][f"eta{it}"] == two[1][f"eta{it}"]
        np.testing.assert_allclose(two[0][f"eta{it}"], one[f"eta{it}"], rtol=1e-5)
    assert np.array_equal(two[0]["p32"], two[1]["p32"]), "ranks hold different parameters after three updates"
    # the same call sites over RCCL: a process group of one rank that still issues every all-reduce (ranges of the
    # gradient buffer as operands, on the stream the several-rank schedule uses)
    rccl = _run_world(1, str(tmp_path / "w1rccl"), PACED_TEST_EARLY="1", PACED_FORCE_EXCHANGE="1", PACED_DIST_BACKEND="nccl")[0]
    for other in (one["p32"], late[0]["p32"], rccl["p32"]):
        diff = np.abs(two[0]["p32"] - other)           # see the note on AdamW's first steps above
        assert diff.max() <= 6.3e-3 and (diff > 1e-4).mean() < 0.03, (diff.max(), (diff > 1e-4).mean())


def test_sharded_pool_selection_equals_unsharded_bit_for_bit(runs):
    one, two = runs
    for r in range(2):
        assert np.array_equal(two[r]["topk_i"], two[r]["full_i"]) and np.array_equal(two[r]["topk_s"], two[r]["full_s"])
        assert np.array_equal(two[r]["topk_i"], one["full_i"]) and np.array_equal(two[r]["topk_s"], one["full_s"])
    assert np.array_equal(one["topk_i"], one["full_i"])

"""Threshold weights, the eta-bound meta objective and the paced pair-loss loops on the GPU against the oracle
(``oracle/schedules_ref.py``, pinned by ``tests/golden/schedules.npz``) and the T5 oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import schedules_ref as R  # noqa: E402
from oracle.t5_ref import T5Dims as RefDims, init_params, t5_forward  # noqa: E402
from pacednegatives_amd import kernels as K  # noqa: E402
from pacednegatives_amd.config import NAMED  # noqa: E402

DEV = "cuda"
TINY = NAMED["t5-tiny"]
RTINY = RefDims(TINY.vocab_size, TINY.d_model, TINY.d_kv, TINY.num_heads, TINY.d_ff, TINY.num_layers, TINY.num_decoder_layers)


def test_threshold_weights_kernel_matches_reference_fixture(golden_dir):
    from pacednegatives_amd.weights import EtaWeights, LCEWeights, Weights
    g = np.load(os.path.join(golden_dir, "schedules.npz"))
    loss = torch.from_numpy(g["losses"])
    for form, cls in ((1, EtaWeights), (0, LCEWeights)):
        for ei, eta in enumerate(g["etas"]):
            for mi, mask in enumerate(g["masks"]):
                w = cls(float(eta), device=DEV)
                w.set_mask(float(mask))
                want = torch.from_numpy(g[f"w_form{form}_eta{ei}_mask{mi}"])
                assert torch.equal(w.no_grad(loss, float(eta)).cpu(), want)              # IEEE division: bit-exact
                assert torch.equal(w.forward(loss).cpu(), want)
                e = torch.tensor([float(eta)], requires_grad=True)
                ref = R.eta_weights(loss, e, float(mask), form)
                slope = torch.stack([torch.autograd.grad(ref[i], (e,), retain_graph=True)[0][0] for i in range(len(loss))])
                np.testing.assert_allclose(w.slope.cpu().numpy(), slope.numpy(), rtol=1e-6, atol=1e-7)
    ew = EtaWeights(5.0, device=DEV, min=1e-10, max=1.0)
    ew.clamp()
    assert ew.eta.item() == 1.0
    W = Weights((3, 5), device=DEV)
    W.set_weight(1, torch.from_numpy(g["sample_val"]))
    np.testing.assert_allclose(W[1].cpu().numpy(), g["sample_weight_row1"], rtol=1e-6)
    assert (W.forward(idx=0) == 1).all()


@pytest.mark.parametrize("R_", [7, 64, 1000])
def test_eta_bound_meta_objective_and_gradient(R_):
    gen = torch.Generator().manual_seed(R_)
    pce, nce = torch.rand(R_, generator=gen) * 1.2, torch.rand(R_, generator=gen) * 1.2
    for eta, mask in ((0.4, 0.0), (0.8, 0.3)):
        e = torch.tensor([eta], dtype=torch.float64, requires_grad=True)
        j, v = R.eta_bound_objective(pce.double(), nce.double(), e, mask)
        (gj,) = torch.autograd.grad(j, (e,))
        out = torch.zeros(4, device=DEV)
        vd = torch.empty(R_, device=DEV)
        K.eta_bound_meta(pce.to(DEV), nce.to(DEV), torch.tensor([eta], device=DEV), vd, out, mask=mask)
        o = out.cpu().double()
        np.testing.assert_allclose(vd.cpu().numpy(), v.detach().float().numpy(), rtol=1e-6, atol=1e-7)
        assert abs(o[0] - j.item()) < 1e-4 * max(1.0, abs(j.item()))
        assert abs(o[1] - gj.item()) < 1e-4 * max(1.0, abs(gj.item()))
        assert abs(o[2] - v.mean().item()) < 1e-5
        assert abs(o[3] - ((pce + nce) / 2 < eta).float().mean().item()) < 1e-6


class _Corpus:
    def __init__(self, n_docs, n_queries, seed=0):
        import random
        rng = random.Random(seed)
        words = "alpha beta gamma delta epsilon zeta eta theta iota kappa lambda mu nu xi".split()
        text = lambda lo, hi: " ".join(rng.choice(words) for _ in range(rng.randint(lo, hi)))  # noqa: E731
        self.docs = [{"doc_id": f"d{i}", "text": text(8, 20)} for i in range(n_docs)]
        self.queries = [{"query_id": f"q{i}", "text": text(3, 6)} for i in range(n_queries)]

    def docs_iter(self):
        return iter(self.docs)

    def queries_iter(self):
        return iter(self.queries)


def loaders(batch_size=4, n_items=16, n_neg=10):
    from pacednegatives_amd.dataloader import LevelLoader, PairLoader, TripletDataset
    corpus = _Corpus(60, n_items)
    pairs = [(f"q{i}", f"d{50 + i % 10}") for i in range(n_items)]
    neg_idx = [[f"d{(3 * i + j) % 50}" for j in range(n_neg + 1)] for i in range(n_items)]
    ds = TripletDataset(pairs, neg_idx, corpus)
    return LevelLoader(ds, batch_size), PairLoader(ds, batch_size)


def model_init(seed=0, dropout=0.0):
    from pacednegatives_amd.model import T5ForConditionalGenerationHIP

    def init():
        import dataclasses
        dims = dataclasses.replace(TINY, dropout_rate=dropout)
        m = T5ForConditionalGenerationHIP(dims, device=DEV, init=False)
        m.params.load_hf(init_params(RTINY, seed))
        return m
    return init


def tokenizer():
    from pacednegatives_amd.tokenizer import SyntheticTokenizer
    return SyntheticTokenizer(TINY.vocab_size)


def test_pair_step_loss_matches_the_oracle_forward():
    """mean(CE_pos) + mean(CE_neg) on one-word targets (``old/interp.py:47-56``) through the two-position engine."""
    from pacednegatives_amd.pairwrapper import InterpWrapper
    level, _ = loaders()
    w = InterpWrapper("synthetic", "t5-tiny", 4, model_init(), tokenizer(), 1e-3, -100, 0.0, 1.0, 64)
    w.train_loader = level
    w.scheduler = w._schedules(64, 0, per_batch=True)
    px, nx, o_p, o_n = w.prep_batch(level.get_batch(0, 0.3))
    assert o_p.tolist() == [[1176]] * 4 and o_n.tolist() == [[6136]] * 4
    P = init_params(RTINY, 0)
    want = 0.0
    for ids, tgt in ((px, o_p), (nx, o_n)):
        lab = torch.cat([tgt.cpu(), torch.full_like(tgt.cpu(), 1)], 1)
        logits = t5_forward(P, RTINY, ids.cpu(), lab, emulate_bf16=True)[:, 0]
        want += torch.nn.functional.cross_entropy(logits.float(), tgt.cpu().view(-1)).item()
    before = w.model.params.p32.clone()
    loss, pce, nce, plog, nlog = w.pair_step(px, nx, o_p, o_n)
    assert abs(loss - want) < 3e-3 * abs(want)
    assert plog.shape == (4, TINY.vocab_size) and pce.shape == (4,)
    assert not torch.equal(before, w.model.params.p32), "the update must have been applied"
    assert float(w.model.params.g32.abs().max()) == 0.0, "gradients are cleared after the step"


def test_interp_and_level_schedules_move_the_difficulty_as_the_reference_rules_say():
    from pacednegatives_amd.pairwrapper import InterpWrapper, LevelWrapper
    level, _ = loaders()
    w = InterpWrapper("synthetic", "t5-tiny", 4, model_init(), tokenizer(), 1e-3, -100, 0.1, 0.9, 8)
    logs = w.train(level, total_steps=16, warmup_steps=0)
    f = R.interpolate_scalar(0.1, 0.9, 8)
    assert logs["difficulty"]["train"] == [f(i * 4) for i in range(4)]
    assert len(logs["loss"]["train"]) == 4 and all(np.isfinite(logs["loss"]["train"]))
    lw = LevelWrapper("synthetic", "t5-tiny", 4, model_init(), tokenizer(), 1e-3, -100, success_threshold=-1.0, heuristic_step_check=2)
    logs = lw.train(level, total_steps=16, warmup_steps=0)
    # threshold below any rate: one level (1/n_neg) harder at every check after the first window
    assert np.allclose(logs["difficulty"]["train"], [0.0, 0.0, 0.1, 0.1])
    lw = LevelWrapper("synthetic", "t5-tiny", 4, model_init(), tokenizer(), 1e-3, -100, success_threshold=2.0, heuristic_step_check=1)
    assert lw.train(level, total_steps=16, warmup_steps=0)["difficulty"]["train"] == [0.0] * 4
    # the success rate itself against the oracle's restatement of old/levels.py:35-39
    g = torch.Generator().manual_seed(3)
    pos, neg = torch.randn(9, 1, TINY.vocab_size, generator=g), torch.randn(9, 1, TINY.vocab_size, generator=g)
    assert lw.success(pos[:, 0].to(DEV), neg[:, 0].to(DEV)) == pytest.approx(R.success_rate(pos, neg, 1176, 6136))


def test_eta_schedule_follows_adam_on_the_meta_objective():
    from pacednegatives_amd.pairwrapper import EtaWrapper
    level, _ = loaders()
    w = EtaWrapper(0.5, "synthetic", "t5-tiny", 4, model_init(), tokenizer(), 1e-3, 5e-2, -100)
    w.train_loader = level
    w.difficulty = 0.0
    from pacednegatives_amd.lceT5 import LinearWarmupSchedule
    w.scheduler = w._schedules(64, 0, per_batch=True)
    w.meta_scheduler = LinearWarmupSchedule(w.meta_optimizer, 0, 64)
    eta, m, v = 0.5, 0.0, 0.0
    for t in range(1, 4):
        px, nx, o_p, o_n = w.prep_batch(level.get_batch(t - 1, w.difficulty))
        pce, nce, _, _ = w.pair_losses(px, nx, o_p, o_n)
        e = torch.tensor([eta], dtype=torch.float64, requires_grad=True)
        j, _ = R.eta_bound_objective(pce.cpu().double(), nce.cpu().double(), e, 0.0)
        (gj,) = torch.autograd.grad(j, (e,))
        lr = w.meta_scheduler.get_last_lr()[0]
        eta, m, v = R.adam_step(eta, gj.item(), m, v, t, lr)
        eta = min(1.0, max(1e-10, eta))
        meta_loss = w.meta_loop(t - 1)
        assert abs(meta_loss - j.item()) < 1e-3 * max(1.0, abs(j.item()))
        assert abs(w.weights.eta.item() - eta) < 2e-4, (t, w.weights.eta.item(), eta)
    logs = w.train(level, total_steps=12, warmup_steps=0)
    assert len(logs["eta"]) == 3 and logs["difficulty"][0] == 0.0 and logs["difficulty"][1] == pytest.approx(logs["eta"][0])
    assert all(0.0 < e <= 1.0 for e in logs["eta"])


def test_free_weights_loop_and_the_unbuildable_wrapper():
    from pacednegatives_amd.pairwrapper import NewWrapper, StdWrapper
    _, pair = loaders()
    w = NewWrapper("synthetic", "t5-tiny", 4, model_init(), tokenizer(), 1e-3, -100)
    w.train_loader = pair
    from pacednegatives_amd.weights import Weights
    from pacednegatives_amd.lceT5 import LinearWarmupSchedule
    w.weights = Weights((4, 4), device=DEV)
    w.scheduler = LinearWarmupSchedule(w.optimizer, 0, 16)
    px, nx, o_p, o_n = w.prep_batch(pair.get_batch(1, w.weights[1]))
    pce, nce, _, _ = w.pair_losses(px, nx, o_p, o_n)
    want = R.sample_weight_update(pce.cpu(), nce.cpu(), torch.ones(4), w.scheduler.get_last_lr()[0])
    w.meta_loop(1)
    np.testing.assert_allclose(w.weights[1].cpu().numpy(), want.numpy(), rtol=1e-5)
    logs = NewWrapper("synthetic", "t5-tiny", 4, model_init(), tokenizer(), 1e-3, -100).train(pair, epochs=1, warmup_steps=1)
    assert len(logs["loss"]["train"]) == 4 and len(logs["avg_weight"][0]) == 4
    with pytest.raises(NotImplementedError):
        StdWrapper("synthetic", "t5-tiny", 4, model_init(), tokenizer(), 1e-3, -100).train(pair, 1, 0)

"""Oracle of the threshold weights and pace schedules against the fixture recorded from the reference
(``tools/make_golden_schedules.py``: EtaWeights/LCEWeights.no_grad, Weights, interpolate_scalar, check_success)."""
import os

import numpy as np
import torch

from oracle import schedules_ref as R
from pacednegatives_amd import pairwrapper


def load(golden_dir):
    return np.load(os.path.join(golden_dir, "schedules.npz"))


def test_threshold_weights_match_reference(golden_dir):
    g = load(golden_dir)
    loss = torch.from_numpy(g["losses"])
    for form in (0, 1):
        for ei, eta in enumerate(g["etas"]):
            for mi, mask in enumerate(g["masks"]):
                want = torch.from_numpy(g[f"w_form{form}_eta{ei}_mask{mi}"])
                eta_t = torch.tensor([eta], dtype=torch.float32)
                assert torch.equal(R.eta_weights(loss, eta_t, float(mask), form), want)
                assert torch.equal(R.eta_weights_loop(loss, float(eta_t), float(mask), form), want)


def test_slope_in_eta_is_the_derivative_of_the_weights(golden_dir):
    loss = torch.from_numpy(load(golden_dir)["losses"]).double()
    for form in (0, 1):
        for eta in (0.3, 0.9):
            e = torch.tensor([eta], dtype=torch.float64, requires_grad=True)
            w = R.eta_weights(loss, e, 0.5, form)
            (g,) = torch.autograd.grad(w.sum(), (e,))
            h = 1e-6
            fd = (R.eta_weights(loss, e.detach() + h, 0.5, form).sum() - R.eta_weights(loss, e.detach() - h, 0.5, form).sum()) / (2 * h)
            assert abs(g.item() - fd.item()) < 1e-5 * max(1.0, abs(fd.item()))


def test_sample_weights_interpolation_and_success_rate(golden_dir):
    g = load(golden_dir)
    assert np.allclose(torch.sigmoid(torch.from_numpy(g["sample_val"])).numpy(), g["sample_weight_row1"], rtol=0, atol=1e-7)
    assert (g["sample_weight_row0"] == 1).all()
    for f in (R.interpolate_scalar(0.1, 0.9, 1000), pairwrapper.interpolate_scalar(0.1, 0.9, 1000)):
        assert [f(int(s)) for s in g["interp_steps"]] == g["interp_values"].tolist()
    V = 6400
    pos, neg = torch.zeros(12, 1, V), torch.zeros(12, 1, V)
    pos[:, 0, (1176, 6136)] = torch.from_numpy(g["success_pos"])
    neg[:, 0, (1176, 6136)] = torch.from_numpy(g["success_neg"])
    assert R.success_rate(pos, neg, 1176, 6136) == g["success_rate"][0]
    assert R.level_update(0.2, [0.9, 0.7], 0.75, 100) == 0.2 + 1 / 100
    assert R.level_update(0.2, [0.9, 0.5], 0.75, 100) == 0.2
    assert R.level_update(0.999, [1.0], 0.5, 100) == 1.0


def test_adjust_lr_is_the_linear_warmup_schedule():
    class Opt:
        param_groups = [{"lr": 0.0}]
    up = pairwrapper.adjust_lr(Opt, 1e-3, 10, 110)
    got = []
    for s in (0, 5, 10, 60, 110, 200):
        up(s)
        got.append(Opt.param_groups[0]["lr"])
    assert np.allclose(got, [0, 5e-4, 1e-3, 5e-4, 0, 0])


def test_adam_restatement_matches_torch():
    p = torch.tensor([0.4], requires_grad=True)
    opt = torch.optim.Adam([p], lr=1e-2)
    q, m, v = 0.4, 0.0, 0.0
    for t, g in enumerate([0.3, -1.2, 0.05, 2.0], 1):
        p.grad = torch.tensor([g])
        opt.step()
        q, m, v = R.adam_step(q, g, m, v, t, 1e-2)
        assert abs(p.item() - q) < 1e-6

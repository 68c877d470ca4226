"""End-to-end parity of the HIP T5 engine and the lceT5 step against the CPU oracle and the committed
fixtures recorded from Hugging Face T5 / the reference.  Tolerances: the HIP path stores activations
in bf16 (8 significant bits); against the oracle run with bf16 rounding at the same points the
difference is accumulation order only (1e-3-class); against the fp32 fixtures it is the bf16 storage
error itself (5e-3 of the logit scale at these depths)."""
import math
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import lce_ref  # noqa: E402
from oracle.t5_ref import T5Dims as RefDims, init_params, t5_forward  # noqa: E402
from pacednegatives_amd.config import NAMED  # noqa: E402

# Bounds, each about twice the largest value measured on an MI355X (``pytest -s`` prints the measurements).
# "emu" = against the oracle with bf16 rounding at the engine's storage points (what differs is summation order),
# "hf" = against fixtures recorded in fp32 from Hugging Face T5 / the reference (what differs is bf16 storage itself).
B_LOGITS_EMU, B_LOGITS_HF = 6e-3, 1e-2          # logits, relative to the largest logit      (measured 4.9e-3, 4.6e-3)
B_LOSS_EMU, B_LOSS_HF = 2e-3, 3e-3              # per-token losses, relative                 (measured 9.2e-4, 1.6e-3)
B_ROWS_EMU, B_ROWS_HF = 2.5e-3, 5e-3            # LCE rows of training steps 0-2, relative    (measured 1.4e-3, 2.3e-3; step 0: 3e-4)
# per-tensor relative L2 error of the gradients with the engine's ReLU decisions replayed in the oracle (see relu_replay)
# (measured 1.16e-2 .. 1.36e-2 over the tiny, t5-small/base/large-width cases; with the oracle taking its own decisions 2.9e-2 .. 4.5e-2
#  without dropout and up to 9.1e-2 with it.  Dropout on, three seeds: worst tensor 1.56e-2 .. 1.84e-2, median 8.0e-3 .. 8.7e-3.)
B_GRAD_REPLAY = 2.5e-2
B_GRAD_REPLAY_DROP, B_GRAD_REPLAY_DROP_MEDIAN = 3.5e-2, 1.7e-2
# What the stated tolerance "logits/loss within 1e-3 bf16" holds for: the loss the step consumes, i.e. the LCE rows of a first
# step (same parameters on both sides) against the oracle that rounds where the kernels round (measured 3e-4) and against the
# fp32 fixture recorded from the reference's own pair_loss (6e-4).  The per-token losses sit AT that figure (9.2e-4 and 1.02e-3
# on the two fixture batches; bound B_LOSS_EMU).  It does NOT hold for the logits: rounding the weights to bf16 alone moves them
# by 3.5e-3 of the largest logit at t5-base depth (tools/rounding_points.py), before any activation is stored.
B_STATED = 1e-3
TINY = NAMED["t5-tiny"]
RTINY = RefDims(TINY.vocab_size, TINY.d_model, TINY.d_kv, TINY.num_heads, TINY.d_ff, TINY.num_layers, TINY.num_decoder_layers)


def within(name, value, bound):
    """Assert value < bound and print both (``pytest -s``): the bounds below are set to about twice the largest value
    measured on an MI355X, so a regression of the arithmetic shows up as a failure, not as slack being used up."""
    print(f"MEASURED {name}: {value:.3e} (bound {bound:.1e})")
    assert value < bound, (name, value, bound)


def relerr(got, want):
    """max |got - want| / max |want|"""
    got, want = torch.as_tensor(got).double(), torch.as_tensor(want).double()
    return ((got - want).abs().max() / want.abs().max().clamp_min(1e-30)).item()


def rel_rows(got, want):
    """max over elements of |got - want| / |want| (per-row losses are far from zero)"""
    got, want = torch.as_tensor(got).double(), torch.as_tensor(want).double()
    return ((got - want).abs() / want.abs().clamp_min(1e-12)).max().item()


def relu_replay(st):
    """``relu_mask`` hook for ``oracle.t5_ref.t5_forward`` that replays the engine's ReLU decisions of the forward state
    ``st`` (call before ``engine.backward``, which releases the saved activations): a unit is on where the engine's
    stored feed-forward activation is positive.  (With dropout on, a dropped unit reads as off; its output and its
    gradient are zero either way.)  A unit whose pre-activation lies within rounding of zero can fall on either side
    in the two implementations and then switches a whole row of a weight gradient on or off; replaying the decisions
    takes that discontinuity out of a gradient comparison."""
    masks = {}
    for dec, layers in ((False, st.enc_layers), (True, st.dec_layers)):
        for l, sv in enumerate(layers):
            masks[(l, dec)] = (sv.act > 0).float().cpu()

    def relu_mask(layer, dec, shape):
        return masks[(layer, dec)].view(shape)
    return relu_mask


def make_model(seed=0):
    from pacednegatives_amd.model import T5ForConditionalGenerationHIP
    m = T5ForConditionalGenerationHIP(TINY, device="cuda", init=False)
    P = init_params(RTINY, seed)
    m.params.load_hf(P)
    return m, P


@pytest.mark.parametrize("ci", [0, 1])
def test_forward_logits_vs_hf_fixture_and_oracle(golden_dir, ci):
    g = np.load(os.path.join(golden_dir, "t5_forward.npz"))
    ids, labels = torch.from_numpy(g[f"t5_ids{ci}"]), torch.from_numpy(g[f"t5_labels{ci}"])
    m, P = make_model()
    m.eval()
    got = m(input_ids=ids, labels=labels).logits.cpu()
    hf = torch.from_numpy(g[f"t5_logits{ci}"])
    emu = t5_forward(P, RTINY, ids, labels, emulate_bf16=True)
    within("logits vs oracle with bf16 rounding at the same points (rel. to max |logit|)", relerr(got, emu), B_LOGITS_EMU)
    within("logits vs Hugging Face fp32 (bf16 storage error)", relerr(got, hf), B_LOGITS_HF)
    # the loss the step consumes
    ce = lambda lg: torch.nn.functional.cross_entropy(lg.view(-1, lg.shape[-1]), labels.view(-1), reduction="none")  # noqa: E731
    within("token loss vs bf16-emulating oracle (rel.)", rel_rows(ce(got), ce(emu)), B_LOSS_EMU)
    within("token loss vs Hugging Face fp32 (rel.)", rel_rows(ce(got), ce(hf)), B_LOSS_HF)


@pytest.mark.parametrize("ci", [0, 1])
def test_monot5_score_vs_hf_fixture_and_oracle(golden_dir, ci):
    """log_softmax(logits[:, 0, (true, false)])[:, 0] (reference distill/wrapper.py:34-38) from the two-row head."""
    from pacednegatives_amd.scorer import MonoT5Model
    g = np.load(os.path.join(golden_dir, "t5_forward.npz"))
    ids, labels = torch.from_numpy(g[f"t5_ids{ci}"]), torch.from_numpy(g[f"t5_labels{ci}"])
    m, P = make_model()
    m.eval()
    got = MonoT5Model(m)({"input_ids": ids})
    assert got.shape == (ids.shape[0],) and got.dtype == torch.float32 and not got.is_cuda
    ref = lambda lg: torch.log_softmax(lg[:, 0, [1176, 6136]].float(), dim=1)[:, 0]  # noqa: E731
    hf = ref(torch.from_numpy(g[f"t5_logits{ci}"]))
    emu = ref(t5_forward(P, RTINY, ids, labels, emulate_bf16=True))
    # a difference of two logits: twice the logit bound of the forward test (6e-3 x scale, scale ~ 1) against the
    # oracle with bf16 rounding at the same points, 3e-2 against the Hugging Face fp32 logits (bf16 storage error)
    within("monoT5 score vs bf16-emulating oracle (abs.)", (got - emu).abs().max().item(), 1.2e-2)
    within("monoT5 score vs Hugging Face fp32 (abs.)", (got - hf).abs().max().item(), 1.5e-2)
    # the generated labels do not matter at position 0: same scores as the full-head path with other labels
    full = m(input_ids=ids, labels=torch.tensor([[6136, 1]] * ids.shape[0])).logits.cpu()
    within("monoT5 two-row head vs full head (abs.)", (got - ref(full)).abs().max().item(), 2e-6)


def test_forward_is_batch_invariant():
    m, _ = make_model()
    m.eval()
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(2, TINY.vocab_size, (5, 40), generator=g)
    labels = torch.tensor([[1176, 1]] * 5)
    full = m(input_ids=ids, labels=labels).logits
    one = m(input_ids=ids[3:4], labels=labels[3:4]).logits
    assert torch.equal(full[3:4], one)


@pytest.mark.parametrize("S", [24, 70])
def test_backward_gradients_vs_oracle(S):
    m, P = make_model()
    B = 3
    g = torch.Generator().manual_seed(9)
    ids = torch.randint(2, TINY.vocab_size, (B, S), generator=g)
    labels = torch.tensor([[6136, 1]] * B)
    st = m.engine.forward(ids, labels, save=True, drop_p=0.0, seed=1)
    from pacednegatives_amd import kernels as K
    replay = relu_replay(st)
    loss = torch.empty(B * 2, device="cuda")
    dl = torch.empty(B * 2, TINY.vocab_size, device="cuda", dtype=torch.bfloat16)
    K.cross_entropy(st.logits, labels.cuda().view(-1).contiguous(), loss, dl, gscale=1.0 / (B * 2))
    m.params.g32.zero_()
    m.engine.backward(st, dl)
    torch.cuda.synchronize()
    got = {k: v.cpu() for k, v in m.params.to_hf("grad").items()}

    # what the un-replayed comparison measures (the oracle takes its own ReLU decisions): printed to show the size of the
    # flip effect, bounded as before
    free = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    lf = t5_forward(free, RTINY, ids, labels, emulate_bf16=True)
    torch.nn.functional.cross_entropy(lf.view(-1, lf.shape[-1]), labels.view(-1), reduction="mean").backward()
    worst_free = max(((got[k] - (v.grad if v.grad is not None else torch.zeros_like(v))).norm() /
                      ((v.grad if v.grad is not None else torch.zeros_like(v)).norm() + 1e-12)).item() for k, v in free.items())
    within(f"S={S} worst per-tensor gradient L2 error, oracle with its OWN ReLU decisions", worst_free, 8e-2)

    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    lg = t5_forward(leaves, RTINY, ids, labels, emulate_bf16=True, relu_mask=replay)
    torch.nn.functional.cross_entropy(lg.view(-1, lg.shape[-1]), labels.view(-1), reduction="mean").backward()
    worst = worst_q = worst_far = 0.0
    for k, v in leaves.items():
        want = v.grad if v.grad is not None else torch.zeros_like(v)
        denom = want.abs().max().item() + 1e-8
        err = (got[k] - want).abs().max().item() / denom
        l2 = ((got[k] - want).norm() / (want.norm() + 1e-12)).item()
        worst = max(worst, l2)
        # the HIP backward also stores every activation gradient in bf16 (the oracle's autograd is
        # fp32 over a bf16-rounded forward whose rounding points follow the reference's K/V-projection form of the
        # cross-attention, not the re-associated form the engine takes when it can): 5e-2 in relative L2 per tensor.  Single elements can be
        # further off where a ReLU input rounds to the other side of zero in one of the two paths
        # (that unit's gradient switches on/off), so the element-wise bound is on the 99.9th percentile.
        # (a tensor with a handful of live entries - the decoder's relative bias sees offsets 0 and -1 only - has no
        # averaging over elements: 8e-2 there)
        assert l2 < (8e-2 if int((want != 0).sum()) < 16 else 5e-2), (k, l2)
        q = torch.quantile((got[k] - want).abs().flatten()[:4_000_000].float(), 0.999).item() / denom
        worst_q = max(worst_q, q)
        assert q < 0.1, (k, q, err)
        # entries off by more than a tenth of the tensor's largest gradient: ReLU units that flip; at most a few per tensor
        far = int(((got[k] - want).abs() > 0.1 * denom).sum())
        worst_far = max(worst_far, far / want.numel())
        assert far <= max(2, 1e-3 * want.numel()), (k, far, want.numel())
    within(f"S={S} worst per-tensor gradient L2 error (ReLU decisions replayed)", worst, B_GRAD_REPLAY)
    within(f"S={S} worst 99.9th-percentile element error", worst_q, 0.1)
    print(f"MEASURED S={S} worst fraction of far-off entries: {worst_far:.2e}")
    assert worst > 0


@pytest.mark.parametrize("d,H,ff", [(512, 8, 1024), (768, 12, 1536)])
def test_forward_backward_at_real_widths(d, H, ff):
    """The t5-small / t5-base widths (2 layers, reduced vocabulary and d_ff to keep the CPU oracle quick): exercises the
    kernel instantiations the benchmark uses (12 heads, 3 feature slices per wave in the cross-attention, ...)."""
    from pacednegatives_amd import kernels as K
    from pacednegatives_amd.config import T5Dims
    from pacednegatives_amd.model import T5ForConditionalGenerationHIP
    dims = T5Dims(6400, d, 64, H, ff, 2, 2)
    rd = RefDims(6400, d, 64, H, ff, 2, 2)
    m = T5ForConditionalGenerationHIP(dims, device="cuda", init=False)
    P = init_params(rd, 3)
    m.params.load_hf(P)
    B, S = 3, 48
    g = torch.Generator().manual_seed(13)
    ids = torch.randint(2, 6400, (B, S), generator=g)
    labels = torch.tensor([[1176, 1], [6136, 1], [6136, 1]])
    st = m.engine.forward(ids, labels, save=True, drop_p=0.0, seed=1)
    assert st.cross2
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    lg = t5_forward(leaves, rd, ids, labels, emulate_bf16=True, relu_mask=relu_replay(st))
    within(f"d={d} logits vs bf16-emulating oracle", relerr(st.logits.cpu().view_as(lg), lg.detach()), B_LOGITS_EMU)
    loss = torch.empty(B * 2, device="cuda")
    dl = torch.empty(B * 2, 6400, device="cuda", dtype=torch.bfloat16)
    K.cross_entropy(st.logits, labels.cuda().view(-1).contiguous(), loss, dl, gscale=1.0 / (B * 2))
    m.params.g32.zero_()
    m.engine.backward(st, dl)
    torch.cuda.synchronize()
    got = {k: v.cpu() for k, v in m.params.to_hf("grad").items()}
    torch.nn.functional.cross_entropy(lg.view(-1, lg.shape[-1]), labels.view(-1), reduction="mean").backward()
    worst = 0.0
    for k, v in leaves.items():
        want = v.grad if v.grad is not None else torch.zeros_like(v)
        l2 = ((got[k] - want).norm() / (want.norm() + 1e-12)).item()
        worst = max(worst, l2)
        assert l2 < (8e-2 if int((want != 0).sum()) < 64 else B_GRAD_REPLAY), (k, l2)
    within(f"d={d} worst per-tensor gradient L2 error (ReLU decisions replayed)", worst, B_GRAD_REPLAY)


def test_tile_ordered_sign_bits_give_the_same_gradients(monkeypatch):
    """At token counts that give the FFN products the 256x256 kernel, the ReLU sign bits travel from the forward to the
    backward pass in that kernel's own order (PACED_TILE_BITS, default on); the row-major bit matrix must give the same
    logits and, up to the order of the fp32 atomics of the weight gradients, the same gradients (dropout on)."""
    from pacednegatives_amd import kernels as K
    from pacednegatives_amd.config import T5Dims
    from pacednegatives_amd.model import T5ForConditionalGenerationHIP
    d, H, ff = 768, 12, 3072
    dims = T5Dims(6400, d, 64, H, ff, 1, 1)
    rd = RefDims(6400, d, 64, H, ff, 1, 1)
    m = T5ForConditionalGenerationHIP(dims, device="cuda", init=False)
    m.params.load_hf(init_params(rd, 5))
    B, S = 20, 256                                   # 5120 tokens x 3072: 240 tiles
    g = torch.Generator().manual_seed(17)
    ids = torch.randint(2, 6400, (B, S), generator=g)
    labels = torch.tensor([[1176, 1], [6136, 1]] * (B // 2))
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PACED_TILE_BITS", mode)
        assert K.sign_bits_buffer(B * S, ff, "cuda").dim() == (1 if mode == "1" else 2)
        st = m.engine.forward(ids, labels, save=True, drop_p=0.1, seed=9)
        loss = torch.empty(B * 2, device="cuda")
        dl = torch.empty(B * 2, 6400, device="cuda", dtype=torch.bfloat16)
        K.cross_entropy(st.logits, labels.cuda().view(-1).contiguous(), loss, dl, gscale=1.0 / (B * 2))
        m.params.g32.zero_()
        m.engine.backward(st, dl)
        torch.cuda.synchronize()
        runs[mode] = (st.logits.clone(), m.params.g32.clone())
    assert torch.equal(runs["1"][0], runs["0"][0])
    ga, gb = runs["1"][1], runs["0"][1]
    assert float(ga.norm()) > 0
    within("gradient L2 difference between the two bit orders", float((ga - gb).norm() / gb.norm()), 1e-6)


@pytest.mark.parametrize("seed", [77, 83, 86])
def test_dropout_masks_replayed_in_the_oracle(seed):
    """Dropout ON: the engine never stores a mask, forward and backward kernels recompute it from (seed, site, index).
    Rebuild every site's mask from the same seeds with the library's mask function, replay them in the oracle and
    compare logits and all gradients: pins the seed wiring and index layout of all ten dropout sites of both stacks.

    Gradient bounds.  About one ReLU input in a thousand lies within bf16 rounding of zero, and a unit whose input rounds
    to the other side of zero in one of the two paths switches a whole row of its weight gradient on or off (measured:
    0.1156 of a tensor's 0.1161 error norm in one row); with six decoder rows per batch one such unit is several per
    cent of a tensor.  Over the 20 step seeds 70..89 the largest per-tensor relative L2 error was 1.0e-2 .. 9.1e-2 (seed
    83: a decoder unit whose change reaches every tensor below it) and the median over tensors 0.9e-2 .. 4.7e-2; the
    seeds here are the former fixed one, the worst and the best of those.  A site wired to the wrong seed or index
    layout puts 2p(1-p) = 18 % of its elements under the wrong factor: tens of per cent in every tensor below it."""
    from pacednegatives_amd import kernels as K
    from pacednegatives_amd import t5_engine as E
    m, P = make_model()
    B, S, p = 3, 32, 0.1
    H = TINY.num_heads
    g = torch.Generator().manual_seed(21)
    ids = torch.randint(2, TINY.vocab_size, (B, S), generator=g)
    labels = torch.tensor([[6136, 1], [1176, 1], [6136, 1]])
    st = m.engine.forward(ids, labels, save=True, drop_p=p, seed=seed)
    assert st.cross2
    site_id = {"emb": (E._SITE_EMB, E._SITE_DEC_EMB), "final": (E._SITE_ENC_FINAL, E._SITE_DEC_FINAL), "att": E._SITE_ATT,
               "att_out": E._SITE_ATT_OUT, "ff_act": E._SITE_FF_ACT, "ff_out": E._SITE_FF_OUT, "catt": E._SITE_CATT,
               "catt_out": E._SITE_CATT_OUT}

    def drop(site, layer, dec, shape):
        sid = site_id[site]
        if isinstance(sid, tuple):             # stack-level sites have their own ids per stack and no decoder offset
            sid, dec = sid[1 if dec else 0], False
        site_seed = m.engine._seed(st, sid, layer, dec)
        mk = lambda n: K.dropout_mask(n, drop_p=p, drop_seed=site_seed).cpu()  # noqa: E731
        if site == "att" and not dec:              # [N,H,S,S4] with key rows padded to a multiple of 4
            n, h, t, sl = shape
            s4 = (sl + 3) // 4 * 4
            return mk(n * h * t * s4).view(n, h, t, s4)[..., :sl]
        if site == "att":                          # decoder self-attention: [N,H,2,2]
            return mk(math.prod(shape)).view(shape)
        if site == "catt":                         # projected queries c = t*H + h: [N, 2, H, S4] -> [N,H,2,S]
            n, h, t, sl = shape
            s4 = (sl + 3) // 4 * 4
            return mk(n * t * h * s4).view(n, t, h, s4)[..., :sl].permute(0, 2, 1, 3)
        return mk(math.prod(shape)).view(shape)    # hidden states: token row x feature

    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    lg = t5_forward(leaves, RTINY, ids, labels, emulate_bf16=True, drop=drop, relu_mask=relu_replay(st))
    scale = lg.abs().max().item()
    with torch.no_grad():          # the logits are compared with the oracle taking its own ReLU decisions, as everywhere else
        lg_own = t5_forward(P, RTINY, ids, labels, emulate_bf16=True, drop=drop)
    within("dropout on: logits vs oracle replaying the masks", relerr(st.logits.cpu().view_as(lg_own), lg_own), B_LOGITS_EMU)
    # with the masks of another seed the logits differ far more than that: the comparison is not vacuous
    st2 = m.engine.forward(ids, labels, save=False, drop_p=p, seed=seed + 1)
    assert (st2.logits.cpu().view_as(lg) - lg.detach()).abs().max().item() > 5e-2 * scale
    loss = torch.empty(B * 2, device="cuda")
    dl = torch.empty(B * 2, TINY.vocab_size, device="cuda", dtype=torch.bfloat16)
    K.cross_entropy(st.logits, labels.cuda().view(-1).contiguous(), loss, dl, gscale=1.0 / (B * 2))
    m.params.g32.zero_()
    m.engine.backward(st, dl)
    torch.cuda.synchronize()
    got = {k: v.cpu() for k, v in m.params.to_hf("grad").items()}
    torch.nn.functional.cross_entropy(lg.view(-1, lg.shape[-1]), labels.view(-1), reduction="mean").backward()
    errs = {}
    for k, v in leaves.items():
        want = v.grad if v.grad is not None else torch.zeros_like(v)
        errs[k] = ((got[k] - want).norm() / (want.norm() + 1e-12)).item()
    worst = max(errs, key=errs.get)
    within(f"dropout on, seed {seed}: worst per-tensor gradient L2 error ({worst}), ReLU decisions replayed", errs[worst], B_GRAD_REPLAY_DROP)
    within(f"dropout on, seed {seed}: median per-tensor gradient L2 error", sorted(errs.values())[len(errs) // 2], B_GRAD_REPLAY_DROP_MEDIAN)


def test_cross_attention_paths_agree():
    """The projection-free decoder cross-attention ((q Wk) . e and (p . e) Wv^T) and the explicit K/V projection path are
    the same function: logits and every gradient agree to bf16 storage error."""
    from pacednegatives_amd import kernels as K
    m, _ = make_model()
    B, S = 4, 64
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(2, TINY.vocab_size, (B, S), generator=g)
    labels = torch.tensor([[1176, 1], [6136, 1]] * 2)
    out = {}
    for force in (False, True):
        m.engine.force_kv_projection = force
        st = m.engine.forward(ids, labels, save=True, drop_p=0.0, seed=1)
        assert st.cross2 == (not force)
        loss = torch.empty(B * 2, device="cuda")
        dl = torch.empty(B * 2, TINY.vocab_size, device="cuda", dtype=torch.bfloat16)
        K.cross_entropy(st.logits, labels.cuda().view(-1).contiguous(), loss, dl, gscale=1.0 / (B * 2))
        m.params.g32.zero_()
        logits = st.logits.clone()
        m.engine.backward(st, dl)
        torch.cuda.synchronize()
        out[force] = (logits.cpu(), {k: v.cpu().clone() for k, v in m.params.to_hf("grad").items()})
    m.engine.force_kv_projection = False
    la, lb = out[False][0], out[True][0]
    assert (la - lb).abs().max().item() < 1e-2 * lb.abs().max().item()
    for k, want in out[True][1].items():
        got = out[False][1][k]
        l2 = ((got - want).norm() / (want.norm() + 1e-12)).item()
        assert l2 < 6e-2, (k, l2)          # two bf16 paths, each within 5e-2 of the fp32 gradient


@pytest.mark.parametrize("use_mean", [False, True])
def test_training_steps_vs_oracle_and_reference_fixture(golden_dir, use_mean):
    from pacednegatives_amd.lceT5 import LCEModel
    g = np.load(os.path.join(golden_dir, "train_step.npz"))
    B, n, S, warm, total = [int(x) for x in g["hparams"]]
    eta0, lr = [float(x) for x in g["eta0_lr"]]
    pos, plab, neg, nlab = (torch.from_numpy(g[k]) for k in ("pos", "plab", "neg", "nlab"))
    hp = {"model_name": "t5-tiny", "total_steps": total, "eta": eta0, "batch_size": B, "lr": lr, "meta_lr": lr, "n": n,
          "warmup_steps": warm, "use_mean": use_mean, "ignore_index": -100, "dropout": 0.0}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = LCEModel(hp)
    P = init_params(RTINY, 0)
    model.model.params.load_hf(P)
    eta = torch.tensor([eta0])
    opt, meta = lce_ref.HFAdamW(P, lr), lce_ref.HFAdamW({"eta": eta}, lr)
    tag = f"step_um{int(use_mean)}_"
    batch = ({"input_ids": pos, "labels": plab}, {"input_ids": neg, "labels": nlab})
    for it in range(3):
        out = model.training_step(batch, it)
        ref = lce_ref.training_step(P, RTINY, eta, opt, meta, it, pos, plab, neg, nlab, n, use_mean, warm, total, emulate_bf16=True)
        rows = out["loss_rows"].cpu().numpy()
        # vs the oracle with bf16 rounding at the same points
        within(f"step {it} LCE rows vs bf16-emulating oracle (rel.)", rel_rows(rows, ref["loss"]), B_ROWS_EMU)
        within(f"step {it} main LCE rows vs bf16-emulating oracle (rel.)", rel_rows(out["main_loss_rows"].cpu(), ref["main_loss"]), B_ROWS_EMU)
        np.testing.assert_allclose(out["weights"].cpu().numpy(), ref["weights"].numpy(), rtol=B_ROWS_EMU, atol=1e-6)
        within(f"step {it} eta vs oracle (abs.)", abs(model.difficulty - float(eta)), 2e-5)
        # vs the fixture recorded from the reference's own pair_loss / LCEWeights on HF T5 in fp32
        within(f"step {it} LCE rows vs reference fixture, fp32 (rel.)", rel_rows(rows, g[tag + f"loss{it}"]), B_ROWS_HF)
        if it == 0:
            np.testing.assert_allclose(model.difficulty, float(g[tag + "eta0"][0]), rtol=1e-5)
            within("step 0 LCE rows vs bf16-emulating oracle (rel.): the stated 1e-3", rel_rows(rows, ref["loss"]), B_STATED)
            within("step 0 LCE rows vs reference fixture, fp32 (rel.): the stated 1e-3", rel_rows(rows, g[tag + "loss0"]), B_STATED)
    w = model.model.params.master("enc.0.qkv")[:TINY.inner].cpu()
    w_ref = P["encoder.block.0.layer.0.SelfAttention.q.weight"]
    # updates 1 and 2 have size ~0.5*lr and ~lr (warm-up); a gradient entry whose sign differs between
    # the two paths moves the weight in opposite directions, so the worst case is ~2*1.5*lr, and rare
    assert (w - w_ref).abs().max().item() <= 3.6 * lr
    assert ((w - w_ref).abs() > 0.5 * lr).float().mean().item() < 0.05


def test_unequal_lengths_take_the_two_call_path():
    from pacednegatives_amd.lceT5 import LCEModel
    hp = {"model_name": "t5-tiny", "total_steps": 10, "eta": 20.0, "batch_size": 2, "lr": 1e-3, "meta_lr": 1e-3, "n": 3,
          "warmup_steps": 0, "use_mean": False, "ignore_index": -100, "dropout": 0.0}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = LCEModel(hp)
    P = init_params(RTINY, 0)
    model.model.params.load_hf(P)
    g = torch.Generator().manual_seed(3)
    pos = torch.randint(2, TINY.vocab_size, (2, 19), generator=g)
    neg = torch.randint(2, TINY.vocab_size, (6, 33), generator=g)
    batch = ({"input_ids": pos, "labels": torch.tensor([[1176, 1]] * 2)}, {"input_ids": neg, "labels": torch.tensor([[6136, 1]] * 6)})
    out = model.training_step(batch, 0)
    eta = torch.tensor([20.0])
    ref = lce_ref.training_step(P, RTINY, eta, lce_ref.HFAdamW(P, 1e-3), lce_ref.HFAdamW({"eta": eta}, 1e-3), 0,
                                pos, batch[0]["labels"], neg, batch[1]["labels"], 3, False, 0, 10, emulate_bf16=True)
    within("two-call path LCE rows vs oracle (rel.)", rel_rows(out["loss_rows"].cpu(), ref["loss"]), B_ROWS_EMU)
    within("two-call path main LCE rows vs oracle (rel.)", rel_rows(out["main_loss_rows"].cpu(), ref["main_loss"]), B_ROWS_EMU)


def test_dropout_step_runs_and_is_seed_deterministic():
    from pacednegatives_amd.lceT5 import LCEModel
    from pacednegatives_amd.synthetic import token_batch
    rows = []
    for _ in range(2):
        hp = {"model_name": "t5-tiny", "total_steps": 10, "eta": 20.0, "batch_size": 2, "lr": 1e-3, "meta_lr": 1e-3, "n": 2,
              "warmup_steps": 0, "use_mean": False, "ignore_index": -100, "dropout": 0.1, "dropout_seed": 77}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = LCEModel(hp)
        out = model.training_step(token_batch(2, 2, 32, TINY.vocab_size), 0)
        rows.append(out["main_loss_rows"].cpu())
        assert torch.isfinite(rows[-1]).all()
    assert torch.equal(rows[0], rows[1])


@pytest.mark.parametrize("neg_len", [48, 61])
def test_side_stream_forwards_give_the_same_steps(neg_len):
    """The no-grad forwards of a step run on a second stream beside the forwards that keep state (from the second
    step on); every logged quantity must match the in-order schedule.  Weight gradients are accumulated with fp32
    atomics, so later steps agree to rounding, not bit for bit."""
    from pacednegatives_amd.lceT5 import LCEModel
    from pacednegatives_amd.synthetic import token_batch
    runs, moved = [], []
    for overlap in (False, True):
        hp = {"model_name": "t5-tiny", "total_steps": 10, "eta": 0.9, "batch_size": 4, "lr": 1e-3, "meta_lr": 1e-2, "n": 3,
              "warmup_steps": 0, "use_mean": False, "ignore_index": -100, "dropout": 0.1, "dropout_seed": 5,
              "overlap_forwards": overlap}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = LCEModel(hp)
        model.model.params.load_hf(init_params(RTINY, 0))
        start = model.model.params.p32.clone().cpu()
        assert model.overlap_forwards is overlap
        steps = []
        for it in range(4):
            pos, neg = token_batch(4, 3, 48, TINY.vocab_size, seed=100 + it)
            if neg_len != 48:           # negatives of another length: positives and negatives go through two engine calls
                neg = {"input_ids": token_batch(4, 3, neg_len, TINY.vocab_size, seed=200 + it)[1]["input_ids"], "labels": neg["labels"]}
            out = model.training_step((pos, neg), it)
            steps.append((out["loss_rows"].cpu(), out["main_loss_rows"].cpu(), out["weights"].cpu(), model.difficulty))
        assert (model._side_stream is not None) is overlap
        assert model.early_update and model.model.params.g32.abs().max().item() == 0      # every range cleared, however it was updated
        runs.append(steps)
        moved.append((model.model.params.p32 - start.to("cuda")).cpu())
    for it, (a, b) in enumerate(zip(*runs)):
        if it == 0:
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        for x, y in zip(a[:3], b[:3]):
            np.testing.assert_allclose(x.numpy(), y.numpy(), rtol=2e-3, atol=1e-5)
        assert abs(a[3] - b[3]) < 1e-5
    # the overlapped schedule updates the decoder and each encoder layer as soon as its gradients are final, on the
    # weight-gradient stream; the four steps must move every parameter as the in-order schedule's single update does
    within("parameter change over 4 steps, overlapped vs in-order schedule (relative L2)",
           ((moved[0] - moved[1]).norm() / moved[0].norm()).item(), 2e-2)
    assert moved[0].norm().item() > 0


def test_share_forward_gives_the_steps_of_the_two_pass_schedule():
    """hparam share_forward (dropout 0 only): one forward serves the no-grad pass and the pass that keeps state."""
    from pacednegatives_amd.lceT5 import LCEModel
    from pacednegatives_amd.synthetic import token_batch
    runs = []
    for share in (False, True):
        hp = {"model_name": "t5-tiny", "total_steps": 10, "eta": 30.0, "batch_size": 4, "lr": 1e-3, "meta_lr": 1e-2, "n": 3,
              "warmup_steps": 0, "use_mean": False, "ignore_index": -100, "dropout": 0.0, "share_forward": share, "overlap_forwards": False}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = LCEModel(hp)
        model.model.params.load_hf(init_params(RTINY, 0))
        steps = []
        for it in range(3):
            out = model.training_step(token_batch(4, 3, 40, TINY.vocab_size, seed=300 + it), it)
            steps.append((out["loss_rows"].cpu(), out["main_loss_rows"].cpu(), out["weights"].cpu(), model.difficulty))
        runs.append(steps)
    for it, (a, b) in enumerate(zip(*runs)):
        if it == 0:
            assert all(torch.equal(x, y) for x, y in zip(a[:3], b[:3])) and a[3] == b[3]
        for x, y in zip(a[:3], b[:3]):            # later steps: weight gradients are summed with atomics
            np.testing.assert_allclose(x.numpy(), y.numpy(), rtol=1e-3, atol=1e-6)
        assert abs(a[3] - b[3]) < 1e-5
    hp["dropout"] = 0.1
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = LCEModel(hp)
    with pytest.raises(ValueError, match="share_forward needs dropout 0"):
        model.training_step(token_batch(4, 3, 40, TINY.vocab_size, seed=1), 0)


def test_save_pretrained_round_trip(tmp_path):
    from pacednegatives_amd.model import T5ForConditionalGenerationHIP
    m, P = make_model(3)
    m.save_pretrained(str(tmp_path / "ckpt"))
    m2 = T5ForConditionalGenerationHIP.from_pretrained(str(tmp_path / "ckpt"))
    for (k, a), (k2, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k == k2 and torch.equal(a, b)
    assert set(m.state_dict()) == set(P)


def test_smoke_entry():
    from pacednegatives_amd.selftest import smoke_step
    within("smoke step LCE rows vs oracle (rel.)", smoke_step(verbose=False), B_ROWS_EMU)

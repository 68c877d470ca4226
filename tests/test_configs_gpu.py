"""The reference's own fixtures through the kernels, and the configurations of the headline benchmarks that no other
test file runs: t5-small at full depth with K=2, B=4 (the CPU plumbing configuration), t5-large widths, the
K=8 / seq_len 128 / batch 64 shape at t5-base widths, and sequence lengths that are not multiples of 8 at real widths."""
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import lce_ref  # noqa: E402
from oracle.t5_ref import T5Dims as RefDims, init_params, t5_forward  # noqa: E402
from test_engine_gpu import B_GRAD_REPLAY, B_LOGITS_EMU, B_ROWS_EMU, rel_rows, relerr, relu_replay, within  # noqa: E402


# t5-large at full depth (24 + 24 layers): bounds of the full-size property test below, ~2x the values measured on an MI355X
LARGE_OTHER_ORDER = 3e-2      # measured 1.53e-2 (alone: small tiles) and 1.51e-2 (oracle)


def _model(hp_extra=None, **hp):
    from pacednegatives_amd.lceT5 import LCEModel
    base = {"model_name": "t5-tiny", "total_steps": 10, "eta": 1.0, "batch_size": 4, "lr": 1e-3, "meta_lr": 1e-3, "n": 2,
            "warmup_steps": 0, "use_mean": False, "ignore_index": -100, "dropout": 0.0}
    base.update(hp)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return LCEModel(base)


# ---- fixtures recorded from the reference's LCEWeights / pair_loss (tools/make_golden.py) -----------------------
@pytest.mark.parametrize("ei", [0, 1, 2])
def test_lce_weights_kernel_vs_reference_fixture(golden_dir, ei):
    """pacednegatives/lceT5.py:82-99 on the device: weights bit for bit, d mean(w*loss)/d eta to 1e-6."""
    from pacednegatives_amd import kernels as K
    from pacednegatives_amd.lceT5 import LCEWeights
    g = np.load(os.path.join(golden_dir, "weights_loss.npz"))
    loss = torch.from_numpy(g[f"w_loss{ei}"]).cuda()
    eta = torch.from_numpy(g[f"w_eta{ei}"]).cuda()
    w, stats = torch.empty_like(loss), torch.zeros(4, device="cuda")
    K.lce_weights(loss, eta, w, stats)
    assert np.array_equal(w.cpu().numpy(), g[f"w_out{ei}"])
    np.testing.assert_allclose(stats[2].item(), g[f"w_grad{ei}"][0], rtol=1e-6)
    np.testing.assert_allclose(stats[0].item(), g[f"w_out{ei}"].mean(), rtol=1e-6)
    np.testing.assert_allclose(stats[1].item(), (g[f"w_out{ei}"] * g[f"w_loss{ei}"]).mean(), rtol=1e-6)
    lw = LCEWeights({"eta": float(g[f"w_eta{ei}"][0])})
    assert np.array_equal(lw(loss).cpu().numpy(), g[f"w_out{ei}"])
    np.testing.assert_allclose(lw.eta_grad[0].item(), g[f"w_grad{ei}"][0], rtol=1e-6)


@pytest.mark.parametrize("ci,n", [(0, 2), (1, 4), (2, 16), (3, 32)])
@pytest.mark.parametrize("use_mean", [False, True])
def test_pair_loss_kernels_vs_reference_fixture(golden_dir, ci, n, use_mean):
    """pacednegatives/lceT5.py:116-123 with the caller's view (:134): fp32 logits recorded with the rows the reference's
    pair_loss produced from them, including ignore_index labels, K=16 and the mean variant."""
    g = np.load(os.path.join(golden_dir, "weights_loss.npz"))
    pl, nl = torch.from_numpy(g[f"l_pl{ci}"]), torch.from_numpy(g[f"l_nl{ci}"])
    op, on = torch.from_numpy(g[f"l_op{ci}"]), torch.from_numpy(g[f"l_on{ci}"])
    model = _model(n=n, use_mean=use_mean)
    V = pl.shape[-1]
    got = model.pair_loss(pl, nl.view(-1, n, V), op, on).cpu()
    want = torch.from_numpy(g[f"l_ce{ci}_{int(use_mean)}"])
    assert got.shape == want.shape
    within(f"pair_loss (B,n)=({pl.shape[0]},{n}) use_mean={use_mean} vs reference fixture (rel.)", rel_rows(got, want), 2e-6)
    np.testing.assert_allclose(got.mean().item(), float(g[f"l_alt{ci}_{int(use_mean)}"]), rtol=1e-5)
    # the un-viewed [B*n, T, V] negatives give the same rows (the kernel groups the flat token losses)
    got2 = model.pair_loss(pl, nl, op, on).cpu()
    assert torch.equal(got, got2)


# ---- configurations --------------------------------------------------------------------------------------------
def _engine(dims_args, seed):
    from pacednegatives_amd.config import T5Dims
    from pacednegatives_amd.model import T5ForConditionalGenerationHIP
    dims, rd = T5Dims(*dims_args), RefDims(*dims_args)
    m = T5ForConditionalGenerationHIP(dims, device="cuda", init=False)
    P = init_params(rd, seed)
    m.params.load_hf(P)
    return m, P, rd


def _fwd_bwd_vs_oracle(m, P, rd, ids, labels, tag, grad_bound=B_GRAD_REPLAY):
    from pacednegatives_amd import kernels as K
    N, V = ids.shape[0], rd.vocab_size
    st = m.engine.forward(ids, labels, save=True, drop_p=0.0, seed=1)
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    lg = t5_forward(leaves, rd, ids, labels, emulate_bf16=True, relu_mask=relu_replay(st))
    within(f"{tag} logits vs bf16-emulating oracle", relerr(st.logits.cpu().view_as(lg), lg.detach()), B_LOGITS_EMU)
    loss = torch.empty(N * 2, device="cuda")
    dl = torch.empty(N * 2, V, device="cuda", dtype=torch.bfloat16)
    K.cross_entropy(st.logits, labels.cuda().view(-1).contiguous(), loss, dl, gscale=1.0 / (N * 2))
    m.params.g32.zero_()
    m.engine.backward(st, dl)
    torch.cuda.synchronize()
    got = {k: v.cpu() for k, v in m.params.to_hf("grad").items()}
    torch.nn.functional.cross_entropy(lg.view(-1, V), labels.view(-1), reduction="mean").backward()
    worst = 0.0
    for k, v in leaves.items():
        want = v.grad if v.grad is not None else torch.zeros_like(v)
        l2 = ((got[k] - want).norm() / (want.norm() + 1e-12)).item()
        worst = max(worst, l2)
        assert l2 < (8e-2 if int((want != 0).sum()) < 64 else grad_bound), (k, l2)
    within(f"{tag} worst per-tensor gradient L2 error (ReLU decisions replayed)", worst, grad_bound)
    return st


def test_engine_at_t5_large_widths():
    """d_model 1024, 16 heads (32 projected queries per sequence in the cross-attention: its limit), d_ff 4096; 2 layers."""
    m, P, rd = _engine((6400, 1024, 64, 16, 4096, 2, 2), 4)
    g = torch.Generator().manual_seed(17)
    ids = torch.randint(2, 6400, (3, 40), generator=g)
    labels = torch.tensor([[1176, 1], [6136, 1], [6136, 1]])
    st = _fwd_bwd_vs_oracle(m, P, rd, ids, labels, "t5-large widths")
    assert st.cross2


def test_t5_small_full_depth_step_vs_oracle():
    """configs[0]: t5-small (6 + 6 layers, d_ff 2048, full vocabulary), K=2 negatives, batch 4 -- one whole training step."""
    from pacednegatives_amd.config import NAMED
    from pacednegatives_amd.synthetic import token_batch
    dm = NAMED["t5-small"]
    rd = RefDims(dm.vocab_size, dm.d_model, dm.d_kv, dm.num_heads, dm.d_ff, dm.num_layers, dm.num_decoder_layers)
    model = _model(model_name="t5-small", eta=40.0, n=2, batch_size=4)
    P = init_params(rd, 0)
    model.model.params.load_hf(P)
    pos, neg = token_batch(4, 2, 48, dm.vocab_size, seed=31)
    out = model.training_step((pos, neg), 0)
    eta = torch.tensor([40.0])
    ref = lce_ref.training_step(P, rd, eta, lce_ref.HFAdamW(P, 1e-3), lce_ref.HFAdamW({"eta": eta}, 1e-3), 0,
                                pos["input_ids"], pos["labels"], neg["input_ids"], neg["labels"], 2, False, 0, 10, emulate_bf16=True)
    within("t5-small step LCE rows vs oracle (rel.)", rel_rows(out["loss_rows"].cpu(), ref["loss"]), B_ROWS_EMU)
    within("t5-small step main LCE rows vs oracle (rel.)", rel_rows(out["main_loss_rows"].cpu(), ref["main_loss"]), B_ROWS_EMU)
    np.testing.assert_allclose(out["weights"].cpu().numpy(), ref["weights"].numpy(), rtol=B_ROWS_EMU, atol=1e-6)
    within("t5-small step eta vs oracle (abs.)", abs(model.difficulty - float(eta)), 2e-5)
    w = model.model.params.master("enc.5.wo").cpu()
    w_ref = P["encoder.block.5.layer.1.DenseReluDense.wo.weight"]
    assert (w - w_ref).abs().max().item() <= 2.1e-3          # one AdamW step of size lr in opposite directions at worst
    assert ((w - w_ref).abs() > 2e-4).float().mean().item() < 0.05


@pytest.mark.parametrize("S", [250, 131])
def test_ragged_sequence_length_at_t5_base_widths(S):
    """Batches are padded to their longest sequence, so S is arbitrary.  d=768, 12 heads, N*S large enough for the
    256x256 phased contraction kernel; S not a multiple of 8 (and odd): such batches take the projection-free
    cross-attention like any other, must equal the oracle, and the explicit K/V projection path (whose backward
    accumulates the encoder-state gradient in place, layer after layer) must agree with it."""
    from pacednegatives_amd import kernels as K
    m, P, rd = _engine((6400, 768, 64, 12, 3072, 2, 2), 5)
    N = 96
    g = torch.Generator().manual_seed(S)
    ids = torch.randint(2, 6400, (N, S), generator=g)
    labels = torch.tensor([[1176, 1], [6136, 1]] * (N // 2))
    # oracle on the first 4 sequences only (a sequence's result does not depend on the rest of the batch)
    st = m.engine.forward(ids, labels, save=True, drop_p=0.0, seed=1)
    lg = t5_forward(P, rd, ids[:4], labels[:4], emulate_bf16=True)
    within(f"S={S} logits vs bf16-emulating oracle", relerr(st.logits[:8].cpu().view_as(lg), lg), B_LOGITS_EMU)
    out = {}
    for force in (False, True):
        m.engine.force_kv_projection = force
        st = m.engine.forward(ids, labels, save=True, drop_p=0.0, seed=1)
        loss = torch.empty(N * 2, device="cuda")
        dl = torch.empty(N * 2, 6400, device="cuda", dtype=torch.bfloat16)
        K.cross_entropy(st.logits, labels.cuda().view(-1).contiguous(), loss, dl, gscale=1.0 / (N * 2))
        logits = st.logits.cpu()
        m.params.g32.zero_()
        m.engine.backward(st, dl)
        torch.cuda.synchronize()
        out[force] = (logits, {k: v.cpu().clone() for k, v in m.params.to_hf("grad").items()}, st.cross2)
    m.engine.force_kv_projection = False
    assert out[True][2] is False and out[False][2] is True
    within(f"S={S} logits, both cross-attention paths", relerr(out[False][0], out[True][0]), 1e-2)
    worst = 0.0
    for k, want in out[True][1].items():
        l2 = ((out[False][1][k] - want).norm() / (want.norm() + 1e-12)).item()
        worst = max(worst, l2)
    within(f"S={S} worst per-tensor gradient L2 difference between the two paths", worst, 6e-2)


def test_k8_s128_b64_shape_properties():
    """configs[1]: t5-base widths, K=8, seq_len 128, batch 64 (576 sequences, 73728 token rows; 2 layers of the 12 to
    keep the test short).  Size-independent properties: every sequence's logits equal those of the same sequence run
    alone (to rounding: the 73728-row products take the 256x256 kernel, a lone sequence the small tiles, and the two sum
    in different orders); the LCE rows equal the grouping of the per-token losses; a second identical step input gives the same rows."""
    from pacednegatives_amd.synthetic import token_batch
    m, P, rd = _engine((32128, 768, 64, 12, 3072, 2, 2), 6)
    pos, neg = token_batch(64, 8, 128, 32128, seed=77)
    ids = torch.cat([pos["input_ids"], neg["input_ids"]], 0)
    labels = torch.cat([pos["labels"], neg["labels"]], 0)
    st = m.engine.forward(ids, labels, save=False, drop_p=0.0, seed=1)
    full = st.logits.view(576, 2, -1)
    worst = 0.0
    for i in (0, 63, 64, 300, 575):
        one = m.engine.forward(ids[i:i + 1], labels[i:i + 1], save=False, drop_p=0.0, seed=1).logits.view(1, 2, -1)
        worst = max(worst, relerr(full[i:i + 1].cpu(), one.cpu()))
    within("K=8,S=128,B=64 a sequence in the batch vs alone (rel. to max |logit|)", worst, B_LOGITS_EMU)
    lg = t5_forward(P, rd, ids[62:66], labels[62:66], emulate_bf16=True)
    within("K=8,S=128,B=64 logits vs bf16-emulating oracle (4 sequences)", relerr(full[62:66].cpu(), lg), B_LOGITS_EMU)
    ce = torch.nn.functional.cross_entropy(full.view(-1, 32128).float(), labels.cuda().view(-1), reduction="none")
    want = ce[:128] + ce[128:].view(-1, 8).sum(1)
    model = _model(n=8)
    rows = model.pair_loss(full[:64], full[64:].view(-1, 8, 32128), pos["labels"], neg["labels"])
    within("K=8,S=128,B=64 LCE rows vs torch grouping of the same logits (rel.)", rel_rows(rows.cpu(), want.cpu()), 2e-6)


def test_t5_large_k32_shard_properties_and_the_run_ahead_rule():
    """configs[4]'s per-GPU shard at full size: t5-large (24 + 24 layers, d 1024, 16 heads, d_ff 4096), K = 32 negatives,
    seq_len 256, 32 tuples = 1056 sequences = 270336 token rows per pass, the eta curriculum's meta step on.  The oracle
    cannot run this size, so size-independent properties: a sequence's logits in the batch equal those of the same
    sequence alone (to rounding: big and small tiles sum in different orders); the LCE rows (the view by n = 32 of
    lceT5.py:134,151) equal torch's grouping of the per-token losses of the same logits; with the learning rates at zero
    and dropout off, step 1 (three streams, no-grad products on 7/8 of the units) reproduces step 0 (one stream) bit for
    bit; and two steps' activations do not fit the GPU beside each other, so from the step after the one that measures a
    step's peak the host must wait for the whole step instead of running ahead."""
    from pacednegatives_amd.synthetic import token_batch
    model = _model(model_name="t5-large", eta=0.3466, n=32, batch_size=32, lr=0.0, meta_lr=0.0, total_steps=100, warmup_steps=0)
    V = model.model.config.vocab_size
    pos, neg = token_batch(32, 32, 256, V, seed=91, device="cuda")
    outs = [model.training_step((pos, neg), i) for i in range(4)]
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o["loss_rows"], outs[0]["loss_rows"]) and torch.equal(o["main_loss_rows"], outs[0]["main_loss_rows"])
        assert torch.equal(o["weights"], outs[0]["weights"])
    assert torch.equal(outs[0]["loss_rows"], outs[0]["main_loss_rows"])        # dropout off: both passes see the same weights
    assert outs[0]["loss_rows"].shape == (64,) and torch.isfinite(outs[0]["loss_rows"]).all()
    # step 0 ran in order, step 1 measured one step's peak (and waited), steps 2 and 3 applied the rule
    (key, peak), = model._step_peak.items()
    total = torch.cuda.get_device_properties(0).total_memory
    print(f"t5-large K=32 shard: one step adds {peak / 2**30:.1f} GiB at its peak; device {total / 2**30:.0f} GiB")
    assert key == (32, 32, 256, 256) and 2 * peak > total, "this shard was chosen because two steps do not fit"
    assert model.ran_ahead is False
    eng = model.model.engine
    ids = torch.cat([pos["input_ids"], neg["input_ids"]], 0)
    labels = torch.cat([pos["labels"], neg["labels"]], 0)
    full = eng.forward(ids, labels, save=False, drop_p=0.0, seed=1).logits.view(1056, 2, V)
    # Three yardsticks for the same logits, in order of strictness.  (1) The sequences 480..575 as a batch of their own
    # take the same kernels as the full batch (24576 token rows: the 256x256 products), so their logits must agree far
    # bit for bit.  (2) A sequence alone takes the small tiles,
    # which sum in another order; 24 + 24 layers of a bf16 residual stream carry that difference to ~1.5e-2 of the
    # largest logit (1.8e-3 after 2 + 2 layers, K=8 test above).  (3) The bf16-emulating fp32 oracle on one sequence
    # at full depth (a third summation order).
    sub = eng.forward(ids[480:576], labels[480:576], save=False, drop_p=0.0, seed=1).logits.view(96, 2, V)
    assert torch.equal(full[480:576], sub), "same kernels, same summation order: the logits must agree bit for bit"
    worst = 0.0
    for i in (0, 31, 32, 500, 1055):
        one = eng.forward(ids[i:i + 1], labels[i:i + 1], save=False, drop_p=0.0, seed=1).logits.view(1, 2, V)
        worst = max(worst, relerr(full[i:i + 1].cpu(), one.cpu()))
    within("t5-large K=32 S=256: a sequence in the batch vs alone (rel. to max |logit|)", worst, LARGE_OTHER_ORDER)
    dm = model.model.config
    rd = RefDims(dm.vocab_size, dm.d_model, dm.d_kv, dm.num_heads, dm.d_ff, dm.num_layers, dm.num_decoder_layers)
    P = {k: v.float().cpu() for k, v in model.model.state_dict().items()}
    lg = t5_forward(P, rd, ids[500:501].cpu(), labels[500:501].cpu(), emulate_bf16=True)
    within("t5-large K=32 S=256: a sequence in the batch vs the bf16-emulating oracle at full depth (rel. to max |logit|)",
           relerr(full[500:501].cpu().view_as(lg), lg), LARGE_OTHER_ORDER)
    del P
    ce = torch.nn.functional.cross_entropy(full.view(-1, V).float(), labels.cuda().view(-1), reduction="none")
    want = ce[:64] + ce[64:].view(-1, 32).sum(1)
    rows = model.pair_loss(full[:32], full[32:].view(-1, 32, V), pos["labels"], neg["labels"])
    within("t5-large K=32 LCE rows vs torch grouping of the same logits (rel.)", rel_rows(rows.cpu(), want.cpu()), 2e-6)
    within("t5-large K=32 step rows vs rows of a plain forward (rel.)", rel_rows(outs[0]["loss_rows"].cpu(), want.cpu()), 2e-6)


@pytest.mark.parametrize("limit,expect", [(1 << 50, True), (1 << 20, False)])
def test_run_ahead_rule_takes_both_branches(limit, expect):
    """The host returns from a step before its backward pass is over only if two steps' activations fit the memory the
    process can count on (``memory_limit_bytes`` stands in for it here).  Either way the steps are the same."""
    from pacednegatives_amd.synthetic import token_batch
    model = _model(eta=5.0, n=2, batch_size=4, memory_limit_bytes=limit)
    ref = _model(eta=5.0, n=2, batch_size=4, run_ahead=False)
    batch = token_batch(4, 2, 40, model.model.config.vocab_size, seed=5, device="cuda")
    flags = []
    for i in range(4):
        a, b = model.training_step(batch, i), ref.training_step(batch, i)
        flags.append(model.ran_ahead)
        assert ref.ran_ahead is False
        assert torch.equal(a["loss_rows"], b["loss_rows"]) and torch.equal(a["main_loss_rows"], b["main_loss_rows"])
    torch.cuda.synchronize()
    assert flags == [False, False, expect, expect]        # in order; measuring step; then the rule
    assert len(model._step_peak) == 1 and next(iter(model._step_peak.values())) > 0
    # another batch shape is measured on its own
    batch2 = token_batch(4, 2, 56, model.model.config.vocab_size, seed=6, device="cuda")
    model.training_step(batch2, 4)
    assert model.ran_ahead is False and len(model._step_peak) == 2
    model.training_step(batch2, 5)
    assert model.ran_ahead is expect
    torch.cuda.synchronize()


def test_monot5_factory_scores_a_tokenised_batch():
    """MonoT5Model.init (reference distill/wrapper.py:16-22) builds model and tokenizer; scores are log-probabilities."""
    from pacednegatives_amd.scorer import MonoT5Model
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        scorer = MonoT5Model.init("t5-tiny")
    scorer.eval()
    enc = scorer.tokenizer(["Query: fast search Document: an index of terms Relevant:", "Query: a Document: b c d Relevant:"],
                           padding=True, return_tensors="pt")
    s = scorer({"input_ids": enc["input_ids"]})
    assert s.shape == (2,) and torch.isfinite(s).all() and (s < 0).all()
    assert (scorer.rel, scorer.nrel) == (1176, 6136)


# ---- checkpoint directories either side of the path (train/train_lce.py:103, eval.py:23) ----------------------
def test_loads_a_checkpoint_written_by_hugging_face(golden_dir):
    """tests/golden/hf_ckpt was written by T5ForConditionalGeneration.save_pretrained (tools/make_golden_hf_ckpt.py)."""
    from pacednegatives_amd.model import T5ForConditionalGenerationHIP
    from test_engine_gpu import B_LOGITS_HF
    g = np.load(os.path.join(golden_dir, "hf_ckpt_logits.npz"))
    m = T5ForConditionalGenerationHIP.from_pretrained(os.path.join(golden_dir, "hf_ckpt"))
    m.eval()
    assert (m.config.d_model, m.config.num_heads, m.config.vocab_size, m.config.dropout_rate) == (64, 2, 6144, 0.1)
    got = m(input_ids=torch.from_numpy(g["ids"]), labels=torch.from_numpy(g["labels"])).logits.cpu()
    within("logits of a Hugging Face-written checkpoint vs its fp32 logits", relerr(got, g["logits"]), B_LOGITS_HF)


def test_hugging_face_loads_what_save_pretrained_writes(golden_dir, tmp_path):
    """Same tensor names, shapes and dtype as the Hugging Face-written fixture; transformers loads the directory with no
    missing or unexpected keys and computes the same logits (what MonoT5ReRanker(model=dir) in eval.py:23 relies on)."""
    from safetensors import safe_open
    from pacednegatives_amd.model import T5ForConditionalGenerationHIP
    from test_engine_gpu import B_LOGITS_HF
    src = os.path.join(golden_dir, "hf_ckpt")
    m = T5ForConditionalGenerationHIP.from_pretrained(src)
    out = str(tmp_path / "finalmodel")
    m.save_pretrained(out)
    with safe_open(os.path.join(src, "model.safetensors"), "pt") as fa, safe_open(os.path.join(out, "model.safetensors"), "pt") as fb:
        assert set(fa.keys()) == set(fb.keys())
        assert fb.metadata() == {"format": "pt"}
        for k in fa.keys():
            a, b = fa.get_tensor(k), fb.get_tensor(k)
            assert a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b), k       # fp32 master copy: a lossless round trip
    import json
    ca, cb = json.load(open(os.path.join(src, "config.json"))), json.load(open(os.path.join(out, "config.json")))
    for key in ("architectures", "model_type", "vocab_size", "d_model", "d_kv", "d_ff", "num_layers", "num_decoder_layers", "num_heads",
                "relative_attention_num_buckets", "relative_attention_max_distance", "dropout_rate", "layer_norm_epsilon",
                "feed_forward_proj", "is_encoder_decoder", "tie_word_embeddings", "pad_token_id", "eos_token_id", "decoder_start_token_id"):
        assert ca[key] == cb[key], key
    from transformers import T5ForConditionalGeneration
    hf, info = T5ForConditionalGeneration.from_pretrained(out, output_loading_info=True)
    assert not info["missing_keys"] and not info["unexpected_keys"] and not info.get("mismatched_keys"), info
    g = np.load(os.path.join(golden_dir, "hf_ckpt_logits.npz"))
    ids, labels = torch.from_numpy(g["ids"]), torch.from_numpy(g["labels"])
    with torch.no_grad():
        want = hf.eval()(input_ids=ids, labels=labels).logits
    assert torch.equal(want, torch.from_numpy(g["logits"])) or relerr(want, g["logits"]) < 1e-5
    m.eval()
    within("this package vs transformers on the directory this package wrote", relerr(m(input_ids=ids, labels=labels).logits.cpu(), want), B_LOGITS_HF)

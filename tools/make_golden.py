#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference itself.

Runs ONLY in the build container (needs /root/reference and Hugging Face transformers); the
resulting small .npz/.json files are committed and are what the tests read.  Nothing here is
imported by the product or by the tests.

What is executed from the reference (PYTHONPATH=/root/reference):
  * pacednegatives.dataloader.LCEDataset          (sample / get / format / __getitem__)
  * pacednegatives.utilities.loss.init_LCEcrossentropy
  * pacednegatives.lceT5.LCEWeights, LCEModel.pair_loss   (module imported with three in-process
    stand-ins for packages absent from the container: `lightning`, `ir_datasets`, and the removed
    `transformers.AdamW` name; none of them takes part in the recorded arithmetic)
and from Hugging Face transformers: T5ForConditionalGeneration on a seeded random-init tiny config.
"""
import json
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)

from oracle.t5_ref import T5Dims, init_params  # noqa: E402  (only for seeded parameters / dims)

TINY = T5Dims(vocab_size=6400, d_model=128, d_kv=64, num_heads=2, d_ff=256, num_layers=2, num_decoder_layers=2)


def import_reference_lcet5():
    import torch.nn as nn
    import transformers
    lt = types.ModuleType("lightning")
    lt.LightningModule = nn.Module
    lt.LightningDataModule = object
    lt.Callback = object
    sys.modules["lightning"] = lt
    sys.modules["ir_datasets"] = types.ModuleType("ir_datasets")
    if not hasattr(transformers, "AdamW"):
        transformers.AdamW = torch.optim.AdamW
    import pacednegatives.lceT5 as ref
    return ref


class FakeCorpus:
    def __init__(self, n_docs, n_queries):
        self.docs = [SimpleNamespace(doc_id=f"d{i}", text=f"document text number {i}") for i in range(n_docs)]
        self.queries = [SimpleNamespace(query_id=f"q{i}", text=f"what is query {i}") for i in range(n_queries)]

    def docs_iter(self):
        return [{"doc_id": d.doc_id, "text": d.text} for d in self.docs]

    def queries_iter(self):
        return [{"query_id": q.query_id, "text": q.text} for q in self.queries]


def golden_sampler():
    from pacednegatives.dataloader import LCEDataset
    out = {}
    items = {}
    for n_neg in (30, 1000):
        corpus = FakeCorpus(n_neg + 4, 3)
        pairs = [[f"q{i}", f"d{n_neg + i}"] for i in range(3)]
        neg_idx = [[f"d{j}" for j in range(n_neg)] for _ in range(3)]
        for n in (2, 8, 16):
            for var in (0.01,):
                ds = LCEDataset(pairs, neg_idx, corpus, None, 0.5, 4, var, n, 1e-10, 1.0 - 1e-10)
                for wi, w in enumerate((1e-10, 0.3466, 0.9, 1.0 - 1e-10, 0.0, 1.5)):
                    for seed in (0, 1, 2):
                        np.random.seed(seed)
                        try:
                            idx = np.stack([ds.sample(w) for _ in range(4)])
                        except SystemExit:
                            idx = np.full((4, n), -1)
                        except ValueError:
                            idx = np.full((4, n), -1)
                        out[f"nneg{n_neg}_n{n}_w{wi}_s{seed}"] = idx.astype(np.int64)
                if n == 2 and n_neg == 30:
                    np.random.seed(5)
                    ds.weight = 0.4
                    items["item0"] = ds[0]
                    items["item2"] = ds[2]
                    items["len"] = len(ds)
    # a pace so low that fewer than n probabilities are non-zero: the reference prints and exits
    corpus = FakeCorpus(1004, 1)
    ds = LCEDataset([["q0", "d1000"]], [[f"d{j}" for j in range(1000)]], corpus, None, 0.5, 4, 0.01, 41, 1e-10, 1.0 - 1e-10)
    np.random.seed(0)
    try:
        ds.sample(1e-10)
        out["fail_nneg1000_n41"] = np.array([0])
    except SystemExit:
        out["fail_nneg1000_n41"] = np.array([1])
    out["weights"] = np.array([1e-10, 0.3466, 0.9, 1.0 - 1e-10, 0.0, 1.5])
    np.savez_compressed(os.path.join(OUT, "sampler.npz"), **out)
    with open(os.path.join(OUT, "items.json"), "w") as f:
        json.dump(items, f, indent=1)


def golden_weights_and_loss(ref):
    from pacednegatives.utilities.loss import init_LCEcrossentropy
    out = {}
    g = torch.Generator().manual_seed(11)
    for ei, eta in enumerate((0.1, 0.4, 1.0)):
        loss = torch.rand(16, generator=g) * 2.0
        w = ref.LCEWeights(SimpleNamespace(eta=eta))
        lw = w(loss)
        (lw * loss).mean().backward()
        out[f"w_loss{ei}"] = loss.numpy()
        out[f"w_out{ei}"] = lw.detach().numpy()
        out[f"w_grad{ei}"] = w.eta.grad.numpy().copy()
        out[f"w_eta{ei}"] = np.array([eta], np.float32)
    V, T = 64, 2
    for ci, (B, n) in enumerate(((4, 2), (4, 4), (2, 16), (2, 32))):      # (2, 32): K = 32 negatives, the t5-large configuration
        pl = torch.randn(B, T, V, generator=g)
        nl = torch.randn(B * n, T, V, generator=g)
        op = torch.tensor([[7, 1]] * B)
        on = torch.tensor([[9, 1]] * (B * n))
        if ci == 1:
            on[1, 1] = -100
            op[2, 0] = -100
        out[f"l_pl{ci}"], out[f"l_nl{ci}"] = pl.numpy(), nl.numpy()
        out[f"l_op{ci}"], out[f"l_on{ci}"] = op.numpy(), on.numpy()
        for um in (False, True):
            self_ns = SimpleNamespace(loss_fn=torch.nn.CrossEntropyLoss(ignore_index=-100, reduction="none"),
                                      hparams=SimpleNamespace(use_mean=um, n=n))
            # caller's reshape (lceT5.py:134) then pair_loss (lceT5.py:116-123)
            ce = ref.LCEModel.pair_loss(self_ns, pl, nl.view(-1, n, V), op, on)
            out[f"l_ce{ci}_{int(um)}"] = ce.numpy()
            alt = init_LCEcrossentropy(ignore_index=-100, use_mean=um)(pl, nl.view(-1, n, V), op, on)
            out[f"l_alt{ci}_{int(um)}"] = alt.numpy()
    np.savez_compressed(os.path.join(OUT, "weights_loss.npz"), **out)


def hf_tiny(seed=0, dropout=0.0):
    from transformers import T5Config, T5ForConditionalGeneration
    cfg = T5Config(vocab_size=TINY.vocab_size, d_model=TINY.d_model, d_kv=TINY.d_kv, d_ff=TINY.d_ff,
                   num_layers=TINY.num_layers, num_decoder_layers=TINY.num_decoder_layers,
                   num_heads=TINY.num_heads, dropout_rate=dropout, decoder_start_token_id=0,
                   feed_forward_proj="relu", tie_word_embeddings=True)
    m = T5ForConditionalGeneration(cfg)
    params = init_params(TINY, seed)
    sd = {k: v.clone() for k, v in params.items()}
    missing = m.load_state_dict(sd, strict=False)
    assert set(missing.missing_keys) <= {"lm_head.weight", "encoder.embed_tokens.weight", "decoder.embed_tokens.weight"}, missing
    assert not missing.unexpected_keys, missing
    m.train()   # Lightning trains in train mode; dropout_rate is 0 for the fixtures
    return m


def synth_batch(B, n, S, seed):
    g = torch.Generator().manual_seed(seed)
    pos = torch.randint(2, TINY.vocab_size, (B, S), generator=g)
    neg = torch.randint(2, TINY.vocab_size, (B * n, S), generator=g)
    pos[:, -1] = 1
    neg[:, -1] = 1
    pl = torch.tensor([[1176, 1]] * B)
    nl = torch.tensor([[6136, 1]] * (B * n))
    return pos, pl, neg, nl


def golden_t5(ref):
    out = {}
    m = hf_tiny(0)
    for ci, (B, n, S) in enumerate(((2, 2, 16), (1, 3, 37))):
        pos, plab, neg, nlab = synth_batch(B, n, S, 100 + ci)
        o = m(input_ids=neg, labels=nlab, output_hidden_states=True)
        out[f"t5_ids{ci}"] = neg.numpy()
        out[f"t5_labels{ci}"] = nlab.numpy()
        out[f"t5_logits{ci}"] = o.logits.detach().numpy()
        out[f"t5_enc_out{ci}"] = o.encoder_last_hidden_state.detach().numpy()
        for l, hs in enumerate(o.encoder_hidden_states):
            out[f"t5_enc_h{l}_{ci}"] = hs.detach().numpy()
        for l, hs in enumerate(o.decoder_hidden_states):
            out[f"t5_dec_h{l}_{ci}"] = hs.detach().numpy()
    # position bias of block 0 (HF:264-279)
    att = m.encoder.block[0].layer[0].SelfAttention
    for S in (16, 128, 256):
        out[f"bias_enc{S}"] = att.compute_bias(S, S).detach().numpy()
    out["bias_dec2"] = m.decoder.block[0].layer[0].SelfAttention.compute_bias(2, 2).detach().numpy()
    np.savez_compressed(os.path.join(OUT, "t5_forward.npz"), **out)

    # ---- one full training step following lceT5.py:125-168, math by the imported pieces
    B, n, S = 2, 3, 24
    eta0, lr, warm, total = 0.3466, 1e-3, 2, 10
    out = {}
    for um in (False, True):
        m = hf_tiny(0)
        pos, plab, neg, nlab = synth_batch(B, n, S, 7)
        weights = ref.LCEWeights(SimpleNamespace(eta=eta0))
        self_ns = SimpleNamespace(loss_fn=torch.nn.CrossEntropyLoss(ignore_index=-100, reduction="none"),
                                  hparams=SimpleNamespace(use_mean=um, n=n))
        meta_opt = torch.optim.AdamW(weights.parameters(), lr=lr, eps=1e-6, weight_decay=0.0)
        opt = torch.optim.AdamW(m.parameters(), lr=lr, eps=1e-6, weight_decay=0.0)
        from transformers import get_linear_schedule_with_warmup
        msch = get_linear_schedule_with_warmup(meta_opt, warm, total)
        sch = get_linear_schedule_with_warmup(opt, warm, total)
        tag = f"step_um{int(um)}_"
        for it in range(3):
            with torch.no_grad():
                plog = m(input_ids=pos, labels=plab, output_hidden_states=False)
                nlog = m(input_ids=neg, labels=nlab).logits
            nlog = nlog.view(-1, n, nlog.size(-1))
            loss = ref.LCEModel.pair_loss(self_ns, plog.logits, nlog, plab, nlab)
            w = weights(loss)
            meta_loss = w * loss
            meta_opt.zero_grad()
            meta_loss.mean().backward()
            out[tag + f"eta_grad{it}"] = weights.eta.grad.numpy().copy()
            meta_opt.step()
            msch.step()
            plog2 = m(input_ids=pos, labels=plab).logits
            nlog2 = m(input_ids=neg, labels=nlab).logits
            nlog2v = nlog2.view(-1, n, nlog2.size(-1))
            main_loss = ref.LCEModel.pair_loss(self_ns, plog2, nlog2v, plab, nlab)
            opt.zero_grad()
            main_loss.mean().backward()
            if it == 0:
                sd = dict(m.named_parameters())
                for k in ("shared.weight", "encoder.block.0.layer.0.SelfAttention.q.weight",
                          "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight",
                          "encoder.block.1.layer.1.DenseReluDense.wi.weight",
                          "encoder.block.1.layer.1.layer_norm.weight",
                          "decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight",
                          "decoder.block.1.layer.1.EncDecAttention.k.weight",
                          "decoder.block.1.layer.2.DenseReluDense.wo.weight",
                          "decoder.final_layer_norm.weight"):
                    g_ = sd[k].grad.numpy().copy()
                    if k == "shared.weight":      # keep the fixture small: touched rows + a norm
                        rows = np.unique(np.concatenate([[0, 1, 1176, 6136], pos.numpy().ravel()[:40]]))
                        out[tag + "grad_rows:" + k] = rows.astype(np.int64)
                        out[tag + "grad_norm:" + k] = np.array([np.sqrt((g_.astype(np.float64) ** 2).sum())])
                        g_ = g_[rows]
                    out[tag + "grad:" + k] = g_
                out[tag + "plogits0"] = plog2.detach().numpy()
                out[tag + "nlogits0"] = nlog2.detach().numpy()
            opt.step()
            sch.step()
            out[tag + f"loss{it}"] = loss.detach().numpy()
            out[tag + f"weights{it}"] = w.detach().numpy()
            out[tag + f"main_loss{it}"] = main_loss.detach().numpy()
            out[tag + f"eta{it}"] = weights.eta.detach().numpy().copy()
        sd = dict(m.named_parameters())
        out[tag + "final:encoder.block.0.layer.0.SelfAttention.q.weight"] = \
            sd["encoder.block.0.layer.0.SelfAttention.q.weight"].detach().numpy().copy()
        out[tag + "final:decoder.block.1.layer.2.DenseReluDense.wo.weight"] = \
            sd["decoder.block.1.layer.2.DenseReluDense.wo.weight"].detach().numpy().copy()
    out["pos"], out["plab"], out["neg"], out["nlab"] = pos.numpy(), plab.numpy(), neg.numpy(), nlab.numpy()
    out["hparams"] = np.array([B, n, S, warm, total], np.int64)
    out["eta0_lr"] = np.array([eta0, lr], np.float64)
    np.savez_compressed(os.path.join(OUT, "train_step.npz"), **out)


def golden_mining():
    """Integer-valued, bf16-exact embeddings: every product and partial sum is exact in fp32, so the
    top-k under (score desc, index asc) is independent of summation order."""
    rng = np.random.RandomState(7)
    Q, Np, D, k = 16, 4096, 64, 50
    q = rng.randint(-8, 9, size=(Q, D)).astype(np.float32)
    p = rng.randint(-8, 9, size=(Np, D)).astype(np.float32)
    s = torch.from_numpy(q) @ torch.from_numpy(p).T
    order = torch.sort(-s, dim=1, stable=True).indices[:, :k]
    np.savez_compressed(os.path.join(OUT, "mining.npz"), q=q, p=p, idx=order.numpy().astype(np.int64),
                        scores=torch.gather(s, 1, order).numpy(), k=np.array([k]))


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(4)
    ref = import_reference_lcet5()
    only = set(sys.argv[1:])          # e.g. `make_golden.py weights_loss` regenerates that file alone
    if not only or "sampler" in only:
        golden_sampler()
    if not only or "weights_loss" in only:
        golden_weights_and_loss(ref)
    if not only or "t5" in only:
        golden_t5(ref)
    if not only or "mining" in only:
        golden_mining()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))

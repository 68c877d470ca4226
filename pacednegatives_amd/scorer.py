"""monoT5 relevance scoring with the HIP T5 engine: the inference counterpart of the training step.

Mirrors ``pacednegatives/distill/wrapper.py:6-38`` (``MonoT5Model``): the score of a "Query: q Document: d
Relevant:" sequence is ``log_softmax(logits[:, 0, (true, false)])[:, 0]`` — the log-probability of "true"
against "false" at the first decoder position.  That position only sees the decoder start token, so the
labels the reference generates (``gen_labels``, :25-26) do not influence it; they are still produced for
interface parity.  The encoder/decoder run through the same kernels as training (no dropout, nothing
saved); instead of the full-vocabulary LM head only the two vocabulary rows that are read are multiplied,
and the 2-way log-softmax is the library's cross-entropy kernel on a 2-column logit matrix.
"""
from __future__ import annotations

import math

import torch

from . import kernels as K

TRUE_ID, FALSE_ID = 1176, 6136          # T5 vocabulary ids of "true" / "false" (pacednegatives/lceT5.py:45-48 builds labels from the same words)


class MonoT5Model:
    def __init__(self, model, tokenizer=None):
        self.model = model
        self.tokenizer = tokenizer
        self.device = model.params.device
        self.rel, self.nrel = TRUE_ID, FALSE_ID
        if tokenizer is not None and hasattr(tokenizer, "encode"):
            try:
                self.rel, self.nrel = int(tokenizer.encode("true")[0]), int(tokenizer.encode("false")[0])
            except Exception:                      # a stand-in tokenizer without these words keeps the published ids
                pass

    @staticmethod
    def init(name_or_dir: str = "t5-base", device="cuda"):
        from .model import T5ForConditionalGenerationHIP
        from .tokenizer import load_tokenizer
        model = T5ForConditionalGenerationHIP.from_pretrained(name_or_dir, device=device)
        return MonoT5Model(model, load_tokenizer(name_or_dir, model.config.vocab_size))

    def save_pretrained(self, path):
        self.model.save_pretrained(path)

    def train(self):
        self.model.train()

    def eval(self):
        self.model.eval()

    def parameters(self):
        return self.model.parameters()

    def gen_labels(self, x):
        """'true' for even rows, 'false' for odd rows, each followed by </s> (wrapper.py:25-26)."""
        n = len(x)
        lab = torch.empty(n, 2, dtype=torch.int64)
        lab[0::2, 0], lab[1::2, 0], lab[:, 1] = self.rel, self.nrel, 1
        return lab

    @torch.no_grad()
    def forward(self, x) -> torch.Tensor:
        """x: {'input_ids': int64 [N, S]} -> float32 [N] on the host (wrapper.py:34-38)."""
        ids = x["input_ids"]
        N = ids.shape[0]
        labels = self.gen_labels(ids)
        eng, P = self.model.engine, self.model.params
        st = eng.forward(ids, labels, save=False, drop_p=0.0, seed=0, want_logits=False)
        y = st.y                                              # [N*2, d]: final-norm output, tied-head scale folded in
        d = y.shape[1]
        pick = torch.tensor([self.rel, self.nrel, self.rel, self.nrel, self.rel, self.nrel, self.rel, self.nrel], device=self.device)
        w2 = P.w("shared").index_select(0, pick).contiguous()  # [8, d]: the two rows the score reads (repeated: N must be a multiple of 4)
        logits2 = torch.empty(N, 8, dtype=torch.float32, device=self.device)
        K.gemm(y.view(N, 2 * d)[:, :d], w2, logits2, epilogue=K.EPI_F32)     # rows n*2+0 only
        nll = torch.empty(N, dtype=torch.float32, device=self.device)
        # the loss kernel works on multiples of 4 columns: over (t, f, t, f) the log-sum-exp is log 2 + LSE(t, f)
        K.cross_entropy(logits2[:, :4], torch.zeros(N, dtype=torch.int64, device=self.device), nll)
        return (math.log(2.0) - nll).cpu()

    __call__ = forward

"""The lceT5 training step on MI355X (mirror of ``pacednegatives/lceT5.py``, Lightning-free).

Same names, constructor arguments and step semantics as the reference module: ``LCEDataModule``
(``lceT5.py:13-67``), ``ChangeDifficulty`` (``:77-80``), ``LCEWeights`` (``:82-99``), ``LCEModel``
(``:101-182``) and the helpers ``batch_iter`` / ``concatenate``.  The arithmetic of a step -- four T5
forwards, the per-token cross-entropy grouped into the LCE loss, the eta-thresholded weights and their
meta-update, the backward pass and both AdamW updates -- runs in the HIP kernels of libpaced_hip.so.
Data-parallel runs (one process per GPU) all-reduce the eta gradient and the flat fp32 gradient
buffer over RCCL (``torch.distributed`` backend "nccl").
"""
from __future__ import annotations

import contextlib
import itertools
import json
import math
import os
from collections import OrderedDict
from typing import Any, List

import torch

from . import kernels as K
from . import parallel
from .dataloader import LCEDataset
from .model import T5ForConditionalGenerationHIP
from .tokenizer import load_tokenizer

F32, BF16 = torch.float32, torch.bfloat16


class _HParams(dict):
    """Dict with attribute access, as Lightning's ``self.hparams``."""
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v


# ----------------------------------------------------------------------------------------- data module
class LCEDataModule:
    def __init__(self, data_dir: str = "path/to/dir", corpus: Any = "msmarco-passage", tokenizer: Any = None,
                 batch_size: int = 32, shuffle=False, use_max=False, var=0.01, n=2, init_weight=0. + 1e-10, num_workers=4,
                 token_store: str = None, propagate_pace: bool = False, seed: int = 0):
        """Arguments as the reference's (``lceT5.py:14-24``) plus three that default to its behaviour:
        ``token_store`` (directory of a ``fastdata.TokenStore``: batches are then assembled in the training process
        from pre-tokenised text, one vectorised draw per batch), ``propagate_pace`` (the pace written by
        ``ChangeDifficulty`` reaches the sampling code; in the reference, worker processes keep the initial pace) and
        ``seed`` of the vectorised sampler."""
        self.data_dir = data_dir
        self.token_store, self.propagate_pace, self.seed = token_store, propagate_pace, seed
        if isinstance(corpus, str):
            try:
                import ir_datasets as irds
            except ImportError as e:
                raise ImportError("corpus was given by name but ir_datasets is not installed; pass an object with "
                                  "docs_iter()/queries_iter() instead") from e
            corpus = irds.load(corpus)
        self.corpus = corpus
        self.tokenizer = tokenizer
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.use_max = use_max
        self.var = var
        self.n = n
        self.num_workers = num_workers
        self.weight = init_weight

    def collate(self):
        """Reference ``lceT5.py:38-54``: tokenise positives and negatives separately, pad each to its own
        longest, labels "true"/"false"; the attention masks are dropped."""
        tokenizer = self.tokenizer

        def collate_fn(batch):
            pos = [x["pos"] for x in batch]
            neg = list(itertools.chain(*[x["neg"] for x in batch]))
            pos_t = tokenizer(pos, padding=True, return_tensors="pt")
            neg_t = tokenizer(neg, padding=True, return_tensors="pt")
            pos_l = tokenizer(["true"] * len(pos_t["input_ids"]), padding=True, return_tensors="pt")["input_ids"]
            neg_l = tokenizer(["false"] * len(neg_t["input_ids"]), padding=True, return_tensors="pt")["input_ids"]
            return ({"input_ids": pos_t["input_ids"], "labels": pos_l}, {"input_ids": neg_t["input_ids"], "labels": neg_l})
        return collate_fn

    def setup(self, stage: str = None):
        with open(self.data_dir, "r") as f:
            records = json.load(f)
        if self.shuffle:
            import random
            random.shuffle(records)
        # what pd.read_json(orient='records', dtype={'query_id': str, 'doc_id_a': str, ...}) gives the reference
        # (lceT5.py:57-62): query and positive ids as strings, the negatives' ids as the JSON holds them
        self.pairs = [[str(r["query_id"]), str(r["doc_id_a"])] for r in records]
        self.neg_idx = [list(r["doc_id_b"]) for r in records]
        self.dataset = LCEDataset(self.pairs, self.neg_idx, self.corpus, self.tokenizer, self.weight, self.batch_size,
                                  var=self.var, n=self.n, min=0. + 1e-10, max=1.0 - 1e-10, use_max=self.use_max)
        if self.propagate_pace:
            self.dataset.share_pace()

    def train_dataloader(self, rank: int = 0, world_size: int = 1):
        """Sequential loader; with ``world_size`` > 1 rank r reads items r, r+W, ... (what Lightning's
        non-shuffling DistributedSampler gives each rank); ``batch_size`` is per rank."""
        from torch.utils.data import DataLoader, Subset
        ds = self.dataset
        if self.token_store is not None:
            from .fastdata import PacedBatchLoader, TokenStore
            return PacedBatchLoader(ds, self.pairs, self.neg_idx, TokenStore.open(self.token_store), self.batch_size,
                                    rank=rank, world_size=world_size, seed=self.seed, propagate_pace=self.propagate_pace)
        view = ds if world_size == 1 else Subset(ds, parallel.shard_indices(len(ds), rank, world_size))
        loader = DataLoader(view, batch_size=self.batch_size, num_workers=self.num_workers, collate_fn=self.collate(),
                            pin_memory=torch.cuda.is_available())
        loader.paced_dataset = ds
        return loader


def batch_iter(iterable, n=1):
    l = len(iterable)
    for ndx in range(0, l, n):
        yield iterable[ndx:min(ndx + n, l)]


def concatenate(*lists):
    return list(itertools.chain(*lists))


class ChangeDifficulty:
    """Reference ``lceT5.py:77-80``: before every batch the dataset's pace becomes min(1-1e-10, eta).
    (With worker processes the workers' dataset copies do not see the write, in the reference as here.)"""

    def on_train_batch_start(self, trainer, pl_module, batch, batch_idx):
        loader = trainer.train_dataloader
        ds = getattr(loader, "paced_dataset", None) or loader.dataset
        ds.weight = min(1 - 1e-10, pl_module.difficulty)


# ------------------------------------------------------------------------------------------- weights
class LCEWeights:
    """eta-thresholded difficulty weights (reference ``lceT5.py:82-99``): w_i = loss_i/eta where
    loss_i <= eta, else 0.  ``eta`` is a 1-element fp32 device tensor updated by the meta optimiser."""

    def __init__(self, hparams, device="cuda"):
        eta0 = hparams["eta"] if isinstance(hparams, dict) else hparams.eta
        self.device = torch.device(device)
        self.eta = torch.tensor([float(eta0)], dtype=F32, device=self.device)
        self.eta_grad = torch.zeros(4, dtype=F32, device=self.device)      # [0] holds d meta_loss / d eta
        self.stats = torch.zeros(4, dtype=F32, device=self.device)         # mean w, mean w*loss, d/d eta, mean loss

    def parameters(self):
        return [self.eta]

    def weighting(self, x, y):
        return x / y

    def forward(self, loss: torch.Tensor) -> torch.Tensor:
        loss = loss.to(self.device, F32).contiguous()
        w = torch.empty_like(loss)
        K.lce_weights(loss, self.eta, w, self.stats)
        self.eta_grad[0:1].copy_(self.stats[2:3])
        return w

    __call__ = forward


# ---------------------------------------------------------------------------------------- optimisers
class FusedAdamW:
    """transformers.AdamW semantics (betas .9/.999, eps 1e-6, weight_decay 0, bias-corrected) as one
    kernel over a flat fp32 buffer; see ``paced_adamw`` in include/paced_hip.h."""

    def __init__(self, p32, g32, m, v, p16=None, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0):
        self.p32, self.g32, self.m, self.v, self.p16 = p32, g32, m, v, p16
        self.base_lr = self.lr = lr
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.t = 0
        self.param_groups = [{"lr": lr}]

        self._early = None     # [lo, hi) ranges already updated (and their gradients cleared) in the step under way
        self._cleared = False

    def zero_grad(self):
        if self._cleared:          # a step made of range updates cleared every range as it went
            self._cleared = False
            return
        self.g32.zero_()

    def _update(self, lo, hi, grad_scale, clear_grad=False):
        b1, b2 = self.betas
        lr = self.param_groups[0]["lr"]
        step_size = lr * math.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)
        p16 = self.p16[lo:hi] if self.p16 is not None else None
        K.adamw(self.p32[lo:hi], self.g32[lo:hi], self.m[lo:hi], self.v[lo:hi], p16, beta1=b1, beta2=b2, eps=self.eps,
                step_size=step_size, decay=lr * self.weight_decay, grad_scale=grad_scale, clear_grad=clear_grad)

    def step_range(self, lo: int, hi: int, grad_scale: float = 1.0):
        """Update elements [lo, hi) now, on the current stream, and clear their gradients: for callers that know the
        range's gradients are final before the rest of the backward pass has run.  The following ``step()`` takes
        the elements no ``step_range`` of the same step covered; ranges must be disjoint."""
        if self._early is None:
            self.t += 1
            self._early = []
        self._update(lo, hi, grad_scale, clear_grad=True)         # the update clears the gradients it has consumed
        self._early.append((lo, hi))

    def pending_ranges(self):
        """[lo, hi) ranges the next ``step()`` will update: everything, or what the step's ``step_range`` calls left."""
        n = self.p32.numel()
        if self._early is None:
            return [(0, n)]
        out, at = [], 0
        for lo, hi in sorted(self._early) + [(n, n)]:
            if lo < at:
                raise RuntimeError("step_range calls of one step overlap")
            if lo > at:
                out.append((at, lo))
            at = hi
        return out

    def step(self, grad_scale: float = 1.0):
        if self._early is None:
            self.t += 1
            self._update(0, self.p32.numel(), grad_scale, clear_grad=True)
            self._cleared = True
            return
        for lo, hi in self.pending_ranges():
            self._update(lo, hi, grad_scale, clear_grad=True)
        self._early = None
        self._cleared = True



class LinearWarmupSchedule:
    """``get_linear_schedule_with_warmup`` (formula restated in the reference at ``pairwrapper.py:16-25``):
    a LambdaLR, so the optimiser starts at base_lr * lambda(0) and each ``step()`` moves to lambda(k)."""

    def __init__(self, optimizer: FusedAdamW, num_warmup_steps: int, num_training_steps: int):
        self.opt, self.warm, self.total, self.k = optimizer, num_warmup_steps, num_training_steps, 0
        self._apply()

    def multiplier(self, k: int) -> float:
        if k < self.warm:
            return float(k) / float(max(1, self.warm))
        return max(0.0, float(self.total - k) / float(max(1, self.total - self.warm)))

    def _apply(self):
        self.opt.param_groups[0]["lr"] = self.opt.base_lr * self.multiplier(self.k)

    def step(self):
        self.k += 1
        self._apply()

    def get_last_lr(self):
        return [self.opt.param_groups[0]["lr"]]


# --------------------------------------------------------------------------------------------- model
class LCEModel:
    def __init__(self, hparams, device=None):
        self.hparams = _HParams()
        for key in hparams.keys():
            self.hparams[key] = hparams[key]
        if not torch.cuda.is_available():
            raise RuntimeError("LCEModel runs its arithmetic in HIP kernels and needs a GPU (no CPU fallback)")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        seed = int(self.hparams.get("seed", 0) or 0)
        self.model = T5ForConditionalGenerationHIP.from_pretrained(self.hparams.model_name, device=self.device, seed=seed)
        self.tokenizer = load_tokenizer(self.hparams.model_name, self.model.config.vocab_size)
        self.weights = LCEWeights(self.hparams, device=self.device)
        self._eta_m = torch.zeros(4, dtype=F32, device=self.device)
        self._eta_v = torch.zeros(4, dtype=F32, device=self.device)
        self._eta_p = torch.zeros(4, dtype=F32, device=self.device)     # 16-byte aligned home of eta
        self._eta_p[0:1].copy_(self.weights.eta)
        self.weights.eta = self._eta_p[0:1]
        self.automatic_optimization = False
        self.dropout = self.hparams.get("dropout", None)
        if self.dropout is None:
            self.dropout = self.model.config.dropout_rate            # Lightning trains in train mode
        self.share_forward = bool(self.hparams.get("share_forward", False))
        # run the no-grad forwards of a step on a second stream beside the forwards that keep state
        self.overlap_forwards = bool(self.hparams.get("overlap_forwards", os.environ.get("PACED_OVERLAP_FORWARDS", "1") != "0"))
        # with overlap_forwards on one rank: AdamW of a segment as soon as its gradients are final (see training_step)
        self.early_update = bool(self.hparams.get("early_update", os.environ.get("PACED_EARLY_UPDATE", "1") != "0"))
        # with overlap_forwards: the no-grad forwards' 256x256 products leave compute units to the decoder (see training_step)
        # (default: 7 of every 8 compute units; 0 = no limit)
        cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        self.nograd_grid = int(self.hparams.get("nograd_grid", os.environ.get("PACED_NOGRAD_GRID", str(cus * 7 // 8))))
        self.nograd_grid_layers = int(self.hparams.get("nograd_grid_layers", os.environ.get("PACED_NOGRAD_GRID_LAYERS", "1000")))
        # with overlap_forwards: training_step returns once the numbers it logs are on the host (see there); False = wait for the step
        self.run_ahead = bool(self.hparams.get("run_ahead", os.environ.get("PACED_RUN_AHEAD", "1") != "0"))
        # memory one step adds at its peak, per batch shape (measured on the first overlapped step of that shape), and
        # the bytes this process may count on (None: what the device reports free plus what torch already holds)
        self._step_peak = {}
        self.memory_limit_bytes = self.hparams.get("memory_limit_bytes", None)
        if self.memory_limit_bytes is None and os.environ.get("PACED_MEMORY_LIMIT"):
            self.memory_limit_bytes = int(os.environ["PACED_MEMORY_LIMIT"])
        self.ran_ahead = False                 # whether the last training_step returned before its backward pass was over
        self._side_stream = self._aux_stream = None
        self._host_vals = None                 # pinned: what a step hands to the host (statistics, eta, mean main loss)
        self.difficulty = float(self.weights.eta.item())
        self.global_step = 0
        self.logged = OrderedDict()
        self._opts = None
        self._base_seed = int(self.hparams.get("dropout_seed", 1234) or 1234)
        self.on_gradients_ready = None

    # ---- Lightning-like plumbing
    def log(self, name, value):
        self.logged[name] = value

    def optimizers(self):
        if self._opts is None:
            self._opts = self.configure_optimizers()
        return self._opts[0]

    def lr_schedulers(self):
        if self._opts is None:
            self._opts = self.configure_optimizers()
        return self._opts[1]

    def configure_optimizers(self):
        """Reference ``lceT5.py:170-182``: AdamW(eta, lr=meta_lr), AdamW(T5, lr=lr), two linear schedules."""
        P = self.model.params
        meta_opt = FusedAdamW(self._eta_p, self.weights.eta_grad, self._eta_m, self._eta_v, None, lr=self.hparams.meta_lr)
        opt = FusedAdamW(P.p32, P.g32, P.m, P.v, P.p16, lr=self.hparams.lr)
        meta_sched = LinearWarmupSchedule(meta_opt, self.hparams.warmup_steps, self.hparams.total_steps)
        sched = LinearWarmupSchedule(opt, self.hparams.warmup_steps, self.hparams.total_steps)
        return [meta_opt, opt], [meta_sched, sched]

    # ---- loss
    def _token_ce(self, logits2d, labels, dlogits=None, gscale=1.0):
        loss = torch.empty(logits2d.shape[0], dtype=F32, device=self.device)
        K.cross_entropy(logits2d, labels.to(self.device, torch.int64).contiguous().view(-1), loss, dlogits,
                        ignore_index=int(self.hparams.ignore_index), gscale=gscale)
        return loss

    def pair_loss(self, p, n, op, on):
        """Reference ``lceT5.py:116-123``.  p [B,T,V]; n [*, n_neg, V] (the caller's view) or [B*n_neg,T,V];
        returns the [B*T] LCE rows: row r = pce[r] + sum|mean of nce[r*n_neg:(r+1)*n_neg]."""
        V = p.shape[-1]
        pce = self._token_ce(p.reshape(-1, V).to(self.device, F32).contiguous(), op)
        nce = self._token_ce(n.reshape(-1, V).to(self.device, F32).contiguous(), on)
        ce = torch.empty_like(pce)
        K.lce_group(pce, nce, ce, int(self.hparams.n), bool(self.hparams.use_mean))
        return ce

    def _two_steps_fit(self, shape_key) -> bool:
        """Whether two steps' activations of this batch shape can be alive at once (see the end of training_step)."""
        peak = self._step_peak.get(shape_key)
        if peak is None:
            return False
        limit = self.memory_limit_bytes
        if limit is None:
            free, _total = torch.cuda.mem_get_info(self.device)
            limit = free + torch.cuda.memory_reserved(self.device)
        return torch.cuda.memory_allocated(self.device) + 2 * peak <= 0.9 * limit

    # ---- the step
    def _forward_pair(self, p, n, save, call, after_encoder=None, gemm_grid=0):
        """Run the positive and negative batches; one packed call when their lengths agree."""
        eng = self.model.engine
        # every rank draws its own masks, as the independent RNG streams of DDP ranks do in the reference
        seed = (self._base_seed * 0x9E3779B1 + parallel.rank() * 0x632BE5AB + self.global_step * 8 + call) & 0xFFFFFFFF
        B = p["input_ids"].shape[0]
        if p["input_ids"].shape[1] == n["input_ids"].shape[1]:
            ids = torch.cat([p["input_ids"].to(self.device), n["input_ids"].to(self.device)], 0)
            lab = torch.cat([p["labels"].to(self.device), n["labels"].to(self.device)], 0)
            st = eng.forward(ids, lab, save=save, drop_p=self.dropout, seed=seed, after_encoder=after_encoder,
                             gemm_grid=gemm_grid, gemm_grid_layers=self.nograd_grid_layers)
            return [st], st.logits[:2 * B], st.logits[2 * B:]
        sp = eng.forward(p["input_ids"], p["labels"], save=save, drop_p=self.dropout, seed=seed,
                         gemm_grid=gemm_grid, gemm_grid_layers=self.nograd_grid_layers)
        sn = eng.forward(n["input_ids"], n["labels"], save=save, drop_p=self.dropout, seed=seed ^ 0x55555555, after_encoder=after_encoder,
                         gemm_grid=gemm_grid, gemm_grid_layers=self.nograd_grid_layers)
        return [sp, sn], sp.logits, sn.logits

    def training_step(self, batch, batch_nb):
        """Reference ``lceT5.py:125-168``, in the same order."""
        p, n = batch
        hp = self.hparams
        nneg, use_mean = int(hp.n), bool(hp.use_mean)
        meta_opt, opt = self.optimizers()
        meta_scheduler, scheduler = self.lr_schedulers()
        world = parallel.world_size()
        B = p["input_ids"].shape[0]
        if n["input_ids"].shape[0] != B * nneg:
            raise ValueError(f"negative batch has {n['input_ids'].shape[0]} rows, expected batch*n = {B * nneg}")
        plab = p["labels"].to(self.device, torch.int64).contiguous().view(-1)
        nlab = n["labels"].to(self.device, torch.int64).contiguous().view(-1)
        ign = int(hp.ignore_index)

        # Nothing in (d) -- the forwards that keep state, the main loss, its backward -- depends on (a) or on eta, and
        # (a) only feeds eta's meta update and the log.  So (a) runs on a second stream, started when (d)'s encoder
        # forward has been enqueued: its large kernels then fill the GPU while (d) is in its two-token decoder forward
        # and backward, and kernel tails on both sides.  The weight gradients of the backward go to a third stream
        # for the same reason.  Values are those of the in-order schedule.  The first step runs in order, so that
        # everything cached on first use exists before two streams read it.
        overlap = self.overlap_forwards and self.global_step > 0 and self.device.type == "cuda"
        shape_key = (B, nneg, int(p["input_ids"].shape[1]), int(n["input_ids"].shape[1]))
        measure_peak = overlap and self.run_ahead and shape_key not in self._step_peak
        if measure_peak:
            torch.cuda.synchronize(self.device)
            torch.cuda.reset_peak_memory_stats(self.device)
            mem_before = torch.cuda.memory_allocated(self.device)
        R = 2 * B
        g_pos = 1.0 / R
        g_neg = (1.0 / R) * ((1.0 / nneg) if use_mean else 1.0)
        # share_forward: with dropout off the no-grad forwards (a) and the forwards that keep state (d) see the same
        # parameters (only eta changes in between) and compute the same logits, so (d)'s loss rows serve (a) as well.
        share = self.share_forward
        if share and self.dropout != 0.0:
            raise ValueError("share_forward needs dropout 0: with dropout on, the reference's two passes draw different masks")

        def no_grad_rows(gemm_grid=0):
            """(a) no-grad forwards and the LCE rows                                        :131-135"""
            states, plog, nlog = self._forward_pair(p, n, save=False, call=0, gemm_grid=gemm_grid)
            pce = torch.empty(2 * B, dtype=F32, device=self.device)
            nce = torch.empty(2 * B * nneg, dtype=F32, device=self.device)
            K.cross_entropy(plog, plab, pce, None, ignore_index=ign)
            K.cross_entropy(nlog, nlab, nce, None, ignore_index=ign)
            rows = torch.empty(2 * B, dtype=F32, device=self.device)
            K.lce_group(pce, nce, rows, nneg, use_mean)
            return rows

        def meta_update(rows):
            """(b) weights and the meta update of eta                                       :137-143"""
            w = self.weights(rows)
            st_ = self.weights.stats.clone()
            parallel.allreduce_sum_(self.weights.eta_grad)          # sum over ranks; averaged by grad_scale below
            meta_opt.step(grad_scale=1.0 / world)
            meta_scheduler.step()
            return w, st_

        def main_backward(after_encoder=None, on_segments_done=None, on_rows=None):
            """(d) forwards with state, loss, backward                                      :149-155
            ``on_rows(rows)`` runs once the LCE rows are enqueued, before the backward pass is."""
            states, plog, nlog = self._forward_pair(p, n, save=True, call=1, after_encoder=after_encoder)
            pce2 = torch.empty(2 * B, dtype=F32, device=self.device)
            nce2 = torch.empty(2 * B * nneg, dtype=F32, device=self.device)
            rows = torch.empty(2 * B, dtype=F32, device=self.device)
            if len(states) == 1:
                dl = torch.empty(states[0].logits.shape, dtype=BF16, device=self.device)
                K.cross_entropy(plog, plab, pce2, dl[:2 * B], ignore_index=ign, gscale=g_pos)
                K.cross_entropy(nlog, nlab, nce2, dl[2 * B:], ignore_index=ign, gscale=g_neg)
                K.lce_group(pce2, nce2, rows, nneg, use_mean)
                if on_rows is not None:
                    on_rows(rows)
                states[0].logits = None
                self.model.engine.backward(states[0], dl, on_segments_done=on_segments_done)
            else:
                dlp = torch.empty(plog.shape, dtype=BF16, device=self.device)
                K.cross_entropy(plog, plab, pce2, dlp, ignore_index=ign, gscale=g_pos)
                dln = torch.empty(nlog.shape, dtype=BF16, device=self.device)
                K.cross_entropy(nlog, nlab, nce2, dln, ignore_index=ign, gscale=g_neg)
                K.lce_group(pce2, nce2, rows, nneg, use_mean)
                if on_rows is not None:
                    on_rows(rows)
                self.model.engine.backward(states[0], dlp)
                self.model.engine.backward(states[1], dln, on_segments_done=on_segments_done)   # the last pass over the gradients
            return rows

        if share:
            self.model.engine.dw_stream = None
            main_loss = main_backward()
            loss = main_loss.clone()
            weights, stats = meta_update(loss)
        elif not overlap:
            self.model.engine.dw_stream = None
            loss = no_grad_rows()
            weights, stats = meta_update(loss)
            main_loss = main_backward()
        else:
            main = torch.cuda.current_stream()
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=self.device)
                self._aux_stream = torch.cuda.Stream(device=self.device)
            aux, box = self._aux_stream, {}

            if self._host_vals is None:
                self._host_vals = torch.zeros(8, dtype=F32).pin_memory()

            def start_no_grad_forwards():
                aux.wait_stream(main)                   # parameters of the last update, and (d)'s encoder forward
                with torch.cuda.stream(aux):
                    # the 256x256 products of these forwards leave compute units to the decoder kernels of the main
                    # stream: a per-call launch option, so no other launch of the process is affected
                    box["loss"] = no_grad_rows(gemm_grid=max(0, self.nograd_grid))
                    box["no_grad_done"] = aux.record_event()
                    # (b) follows (a) on the same stream, as in the reference.  What the host reads of it (the
                    # statistics, eta after the update) is final here, long before the backward pass is over: the
                    # copy to pinned memory is enqueued now and only its event is waited for at the end of the step,
                    # so the host enqueues the next step while the GPU is still in this one's backward pass.
                    box["weights"], box["stats"] = meta_update(box["loss"])
                    self._host_vals[0:4].copy_(box["stats"], non_blocking=True)
                    self._host_vals[4:5].copy_(self.weights.eta, non_blocking=True)
                    box["meta_read"] = aux.record_event()

            def read_main_loss(rows):
                self._host_vals[5:6].copy_(rows.mean().view(1), non_blocking=True)
                box["main_read"] = main.record_event()
            self.model.engine.dw_stream = self._side_stream
            early = None
            if self.early_update and self.on_gradients_ready is None:
                # A segment's gradients are final when its part of the backward pass is over.  Its exchange over the
                # ranks and its AdamW update (HBM-bound, 1.5 ms for the whole model when left to the end of the step)
                # then run under the rest of the backward pass: the decoder's segments in one piece, each encoder layer
                # as it completes.  One rank: on the weight-gradient stream.  Several ranks: on the stream of the no-grad
                # forwards, which has nothing else left to do, so that an all-reduce never sits in front of a weight-
                # gradient product.  Either way the update waits for the chain up to here (gradient pieces the chain
                # writes itself, and its last read of these weights), for the segment's weight-gradient products and
                # for the no-grad forwards, which read every weight.  A communication kernel holding 16-64 compute units
                # for 5 ms of the backward pass costs the step 0.0-0.7 ms (tools/cu_holder_step.py).
                P, side = self.model.params, self._side_stream
                lane = side if world == 1 and not parallel.exchange_forced() else aux

                def early(first, last):
                    lo, hi = P.span(first, last)
                    lane.wait_stream(main)
                    if lane is not side:
                        lane.wait_stream(side)
                    lane.wait_event(box["no_grad_done"])
                    with torch.cuda.stream(lane):
                        parallel.allreduce_sum_(P.g32[lo:hi])
                        opt.step_range(lo, hi, grad_scale=1.0 / world)
            main_loss = main_backward(after_encoder=start_no_grad_forwards, on_segments_done=early, on_rows=read_main_loss)
            main.wait_stream(aux)
            loss, weights, stats = box["loss"], box["weights"], box["stats"]
            for t in (loss, weights, stats):
                t.record_stream(main)
        # main update                                                                        :156-157
        for lo, hi in opt.pending_ranges():                     # the whole flat fp32 buffer, or what no early update took
            parallel.allreduce_sum_(self.model.params.g32[lo:hi])
        if self.on_gradients_ready is not None:                 # tests: the summed gradients, before the update clears them
            self.on_gradients_ready(self)
        opt.step(grad_scale=1.0 / world)
        scheduler.step()
        opt.zero_grad()

        # one host read-back per step: stats of (b), eta after the meta update, mean main loss
        if overlap and not share:
            # The host goes on to enqueue the next step as soon as these two copies have landed.  Its allocations then
            # meet blocks of this step that other streams have not released yet (the backward's operands, handed to the
            # weight-gradient stream), so two steps' activations can be alive at once: where that does not fit the
            # GPU's memory the allocator would stall and re-allocate every step, and the host waits for the whole step.
            # The rule compares twice ONE step's peak (measured on the first overlapped step of this batch shape, which
            # waits for the whole step) plus what is allocated between steps with the memory this process can count on:
            # what the device reports free (other processes' share is gone from it) plus what torch already holds.
            if measure_peak:
                main.synchronize()
                self._step_peak[shape_key] = max(0, torch.cuda.max_memory_allocated(self.device) - mem_before)
                self.ran_ahead = False
            else:
                self.ran_ahead = self.run_ahead and self._two_steps_fit(shape_key)
                if not self.ran_ahead:
                    main.synchronize()
            box["meta_read"].synchronize()
            box["main_read"].synchronize()
            host = self._host_vals[0:6].tolist()
        else:
            host = torch.cat([stats, self.weights.eta, main_loss.mean().view(1)]).tolist()
        avg_weight, meta_loss_mean, _deta, loss_mean, eta, main_mean = host
        self.log("avg_weight", avg_weight)
        self.log("meta_loss", meta_loss_mean)
        self.log("eta", eta)
        self.log("main_loss", loss_mean)          # the reference logs the no-grad loss under this name (:166)
        self.log("lr-AdamW", scheduler.get_last_lr()[0])
        self.log("lr-AdamW-meta", meta_scheduler.get_last_lr()[0])
        self.difficulty = eta
        self.global_step += 1
        tqdm_dict = {"meta_loss": loss_mean, "avg_weight": avg_weight, "main_loss": loss_mean}
        return OrderedDict({"main_loss": main_mean, "meta_loss": meta_loss_mean, "progress_bar": tqdm_dict, "log": tqdm_dict,
                            "loss_rows": loss, "weights": weights, "main_loss_rows": main_loss})

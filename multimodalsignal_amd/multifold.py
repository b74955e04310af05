"""Lockstep training of several LOSO folds as ONE fold batch (include/msig.h msig_*_multi).

The reference trains its 15 folds one after the other (main.py:98-125); each is a fresh model with its own
data split, optimiser, scheduler and early stopping.  At its batch size (64 windows) a fold's train step is ~30
launches that each occupy a few CUs, and fifteen folds on fifteen streams are bound by the command processor's
dispatch rate, not by the CUs.  Here the folds of a rank advance in lockstep: every launch of the step covers all
active folds (blockIdx.z = fold, per-fold arenas — runtime.FoldArena), one gather builds all their batches, one
device op accumulates all their losses, and there is one host sync per epoch.  Everything that is per fold in the
reference stays per fold — weights, BatchNorm statistics, shuffling order, dropout stream, learning-rate schedule,
early stopping, checkpoints, logs.  The forward GRU forms are bit-identical, so the library picks them per launch (layer 0
wave-specialised, layer 1 from 32 tiles per launch on); the BACKWARD form of a launch is by default the one a stand-alone fold of
the same batch size gets (runtime.FoldArena.multi pins msig_multi.form_folds = 1), so each fold's numbers are bit-identical to its
stand-alone run whatever its companions, the grouping or the rank count
(tests/test_trainer_gpu.py::test_lockstep_folds_equal_sequential, ::test_fold_results_do_not_depend_on_sharding); with
`adaptive_forms` the backward form follows the folds still active in a launch (a fold's last bits then depend, reproducibly, on
when its companions stop).  Folds may differ in train / val set size; folds that
stop early leave the batch.
"""
from __future__ import annotations

import ctypes as C
import time
from typing import List

import numpy as np
import torch

from . import _lib as L
from .runtime import FoldArena
from .trainer import Trainer, accuracy_and_weighted_f1


def lockstep_compatible(preps) -> bool:
    """Folds can share launches when they draw from one SubjectStore with one model configuration and one batch size.  Their
    train / val sets may differ in size (WESAD subjects differ by a few windows, dataset.py:17-27): full batches run as one fold
    batch, the folds' ragged last batches as launches over the folds whose batch sizes agree (`launch_plan`)."""
    if not (1 <= len(preps) <= L.MAX_FOLDS):       # a batch of ONE fold is a fold batch too (it can be re-dealt with others later)
        return False
    tr0, va0, _ = preps[0]["loaders"]
    for p in preps:
        tr, va, _ = p["loaders"]
        if (getattr(p["model"], "embedded", False)          # the one-layer 32-unit model keeps its own embedded engine (no arena form)
                or tr.batch_size != tr0.batch_size or va.batch_size != va0.batch_size or tr.store.data_ptr() != tr0.store.data_ptr()
                or p["model"].in_channels != preps[0]["model"].in_channels or p["model"].num_classes != preps[0]["model"].num_classes
                or p["model"].dropout_p != preps[0]["model"].dropout_p):
            return False
    return True


def launch_plan(sizes, bs):
    """The launches of one pass over datasets of `sizes` windows (non-increasing) in batches of `bs`, as (first window, batch
    size, first row, row count) records: at every step the rows that still have a batch form a prefix and their batch sizes are
    non-increasing, so the rows of equal batch size are contiguous runs — each run is one launch over those folds."""
    assert all(sizes[i] >= sizes[i + 1] for i in range(len(sizes) - 1))
    plan, k = [], 0
    while sizes and k * bs < sizes[0]:
        i, r = k * bs, 0
        while r < len(sizes) and sizes[r] > i:
            b, r1 = min(bs, sizes[r] - i), r + 1
            while r1 < len(sizes) and sizes[r1] > i and min(bs, sizes[r1] - i) == b:
                r1 += 1
            plan.append((i, b, r, r1 - r))
            r = r1
        k += 1
    return plan


class LockstepTrainer:
    def __init__(self, preps: List[dict], device, adaptive_forms: bool = False):
        self.preps, self.device = preps, torch.device(device)
        tr0, va0, te0 = preps[0]["loaders"]
        m0 = preps[0]["model"]
        self.n = len(preps)
        self.C, self.K, self.T = m0.in_channels, m0.num_classes, int(tr0.store.shape[2])
        self.arena = FoldArena(self.C, self.K, self.device, self.n, tr0.batch_size, self.T, eval_batch=max(va0.batch_size, te0.batch_size),
                               adaptive_forms=adaptive_forms)
        self.trainers: List[Trainer] = []
        for slot, p in enumerate(preps):
            model = p["model"]
            old = model._engine
            new = self.arena.engine(slot)                      # zeroed storage of arena `slot`
            if p.get("trainer") is not None and old is not None:
                # a fold that has already trained in ANOTHER fold batch (main.run_experiments re-deals the surviving folds between
                # rounds): its Adam moments move with it; model.engine() below moves the parameters and the BatchNorm state
                new.exp_avg.copy_(old.exp_avg)
                new.exp_avg_sq.copy_(old.exp_avg_sq)
            model._engine = new                                # the model's parameters become views into arena `slot`
            t = p.get("trainer") or Trainer(model, p["fold_dir"], p["config"])
            p["trainer"] = t
            model.engine()
            self.trainers.append(t)
        h0 = self.trainers[0].optimizer.hyper
        for t in self.trainers:
            h = t.optimizer.hyper
            if (h["betas"], h["eps"], h["weight_decay"]) != (h0["betas"], h0["eps"], h0["weight_decay"]) or t.epochs != self.trainers[0].epochs:
                raise ValueError("lockstep folds must share betas / eps / weight decay / epoch budget")
        # per-fold running sums of a pass ([sum of CE, #correct], float64): msig_batch.loss_acc of every arena, added to by the
        # loss kernel of each launch a fold takes part in — no accumulation op per launch on this side
        self.acc = self.arena.across("acc", 0, torch.float64, 2)                       # (folds, 2) strided view
        self._layouts, self._eval_orders, self._idx = {}, {}, {}

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- host-side bookkeeping is kept off the per-step path: descriptors, layouts and dropout keys are prepared per epoch ----
    def _layout(self, B, training):
        key = (B, bool(training))
        if key not in self._layouts:
            off = L.workspace_layout(B, self.C, self.T, self.K, training)
            self._layouts[key] = (off, self.arena.batch(B, training, self.trainers[0].model.dropout_p if training else 0.0))
        return self._layouts[key]

    def _gather(self, loader, order_mat, row0, i, b, m):
        """order_mat: (folds, n_max) int64 store positions of the pass; gathers columns i .. i+b of rows row0 .. row0+m.n."""
        wfl = loader.store.shape[1] * loader.store.shape[2]
        L.check(L.lib().msig_gather_windows_multi(loader.store.data_ptr(), loader.store_y.data_ptr(),
                                                  order_mat.data_ptr() + 8 * (row0 * order_mat.shape[1] + i), order_mat.shape[1], b, wfl,
                                                  self.arena.ptr("x"), self.arena.ptr("y"), C.byref(m), self._stream()),
                "msig_gather_windows_multi")

    @staticmethod
    def _stack(rows):
        """(folds, n_max) matrix of the folds' visiting orders; short rows are padded with their own first entry (never gathered:
        a launch only covers columns every one of its rows has)."""
        n_max = max(int(r.numel()) for r in rows)
        if all(int(r.numel()) == n_max for r in rows):
            return torch.stack(rows).contiguous()
        return torch.stack([torch.cat([r, r[:1].expand(n_max - int(r.numel()))]) for r in rows]).contiguous()

    def _zero_acc(self, slots):
        """Zeroes the running sums of the given arenas only: a fold that has stopped runs its test pass on a side stream with its
        own arena's accumulator."""
        key = tuple(slots)
        if key not in self._idx:
            self._idx[key] = torch.tensor(list(slots), dtype=torch.int64, device=self.device)
        self.acc.index_fill_(0, self._idx[key], 0.0)

    def _train_epoch(self, active):
        """One epoch of every active fold.  Folds are visited in order of decreasing training-set size so that the folds of a
        launch are consecutive rows of the order matrix (`launch_plan`); which folds share a launch has no influence on any
        fold's numbers.  Returns per-arena loss sums (indexed by slot) — the epoch's only sync."""
        arena, lib = self.arena, L.lib()
        act = sorted(active, key=lambda f: -len(self.preps[f]["loaders"][0].dataset))
        trs = [self.trainers[f] for f in act]
        loaders = [self.preps[f]["loaders"][0] for f in act]
        order = self._stack([ld.epoch_order() for ld in loaders])                     # (folds, n_max): one stack per epoch
        sizes, bs = [len(ld.dataset) for ld in loaders], loaders[0].batch_size
        for t in trs:
            t.model.train()
        thr = L.dropout_threshold(trs[0].model.dropout_p)
        step0 = [t.optimizer.step_count for t in trs]                                  # folds of unequal size drift apart in step count
        n_steps = [(n + bs - 1) // bs for n in sizes]
        steps = [np.arange(s0 + 1, s0 + 1 + ns) for s0, ns in zip(step0, n_steps)]
        kg = [L.dropout_keys(t.model._seed, st, 1) if thr else np.zeros(len(st), np.uint32) for t, st in zip(trs, steps)]
        kh = [L.dropout_keys(t.model._seed, st, 2) if thr else np.zeros(len(st), np.uint32) for t, st in zip(trs, steps)]
        lrs = [t.optimizer.hyper["lr"] for t in trs]
        h0 = trs[0].optimizer.hyper
        b1, b2, eps, wd = h0["betas"][0], h0["betas"][1], h0["eps"], h0["weight_decay"]
        ea, eas, st = arena.ptr("exp_avg"), arena.ptr("exp_avg_sq"), self._stream()
        multis = {}
        self._zero_acc(act)
        for i, b, r0, nr in launch_plan(sizes, bs):
            k = i // bs
            if (r0, nr) not in multis:                                                 # slots and learning rates: per epoch and row run
                multis[(r0, nr)] = arena.multi(act[r0:r0 + nr], lr=lrs[r0:r0 + nr])
            m = multis[(r0, nr)]
            for j in range(nr):
                m.key_gru[j] = int(kg[r0 + j][k]); m.key_head[j] = int(kh[r0 + j][k]); m.step[j] = int(steps[r0 + j][k])
            self._gather(loaders[0], order, r0, i, b, m)
            _, desc = self._layout(b, True)
            L.check(lib.msig_train_step_multi(C.byref(desc), C.byref(m), ea, eas, b1, b2, eps, wd, int(steps[r0][k]), st), "msig_train_step_multi")
        for t, s0, ns in zip(trs, step0, n_steps):
            t.optimizer.step_count = s0 + ns
        return self.acc[:, 0].cpu().numpy()                   # the epoch's only sync

    def _evaluate(self, active, which):
        """Validation pass of every active fold (loader index `which`): per fold (loss, acc, f1), in the order of `active`."""
        arena, lib = self.arena, L.lib()
        act = sorted(active, key=lambda f: -len(self.preps[f]["loaders"][which].dataset))
        loaders = [self.preps[f]["loaders"][which] for f in act]
        for f in act:
            self.trainers[f].model.eval()
        key = (which, tuple(act))
        if key not in self._eval_orders:                   # validation order is fixed (no shuffling): stack it once per active set
            self._eval_orders = {key: self._stack([ld.epoch_order() for ld in loaders])}
        order = self._eval_orders[key]
        sizes, bs = [len(ld.dataset) for ld in loaders], loaders[0].batch_size
        st = self._stream()
        multis = {}
        self._zero_acc(act)
        preds = [[] for _ in act]
        for i, b, r0, nr in launch_plan(sizes, bs):
            if (r0, nr) not in multis:
                multis[(r0, nr)] = arena.multi(act[r0:r0 + nr])
            m = multis[(r0, nr)]
            self._gather(loaders[0], order, r0, i, b, m)
            off, desc = self._layout(b, False)
            L.check(lib.msig_forward_multi(C.byref(desc), C.byref(m), st), "msig_forward_multi")
            got = arena.across("ws", off[L.WS["PRED"]], torch.int32, b)[act[r0:r0 + nr]]      # (folds of the launch, b) copy
            for j in range(nr):
                preds[r0 + j].append(got[j])
        sums = self.acc[:, 0].cpu().numpy()
        out = {}
        for j, f in enumerate(act):
            ds = loaders[j].dataset
            pred = torch.cat(preds[j]).cpu().numpy().astype(np.int64)
            acc, f1 = accuracy_and_weighted_f1(np.asarray(ds.labels).astype(np.int64), pred)
            out[f] = (float(sums[f]) / len(ds), acc, f1)
        return [out[f] for f in active]

    def run(self, epoch0: int = 0, max_epochs: int = 0, t_start: float = None):
        """Trains every fold to its early stop, then evaluates each on its test subject; returns main.train_fold's dicts.
        With `max_epochs` > 0 it trains epochs epoch0 .. epoch0 + max_epochs - 1 only and returns (infos of the folds that finished, in
        a dict by position, positions of the folds still training): main.run_experiments runs a rank's folds in ROUNDS and re-deals
        the survivors evenly over the fold batches between rounds (which folds share a batch has no influence on any fold's numbers)."""
        import json
        from concurrent.futures import ThreadPoolExecutor
        for t, p in zip(self.trainers, self.preps):
            for ld in p["loaders"]:
                t._check_labels(ld)
        t_start = time.time() if t_start is None else t_start
        dev = self.device

        def finish(f):
            """Checkpoint restore, test pass, confusion matrix, fold_result.json of one fold — on a side stream, off the
            lockstep loop's critical path (a stopped fold's arena is no longer touched by the batch)."""
            t, p = self.trainers[f], self.preps[f]
            torch.cuda.set_device(dev)
            with torch.cuda.stream(torch.cuda.Stream(dev)):
                t._finish_training()
                _, acc, f1 = t.evaluate(p["loaders"][2], is_test=True)
                torch.cuda.current_stream(dev).synchronize()
            info = dict(subject=p["subject"], accuracy=acc, f1_score=f1, seconds=getattr(t, "finished_at", time.time() - t_start),
                        epochs=len(t.history), train_windows_per_s=t.train_windows / max(t.train_seconds, 1e-9), history=t.history)
            (p["fold_dir"] / "fold_result.json").write_text(json.dumps(info))
            return info

        active = list(range(self.n))
        n_train = [len(p["loaders"][0].dataset) for p in self.preps]
        pending = {}
        budget = self.trainers[0].epochs
        last = min(budget, epoch0 + max_epochs) if max_epochs > 0 else budget
        with ThreadPoolExecutor(max_workers=2) as side:
            for epoch in range(epoch0, last):
                if not active:
                    break
                t0 = time.time()
                sums = self._train_epoch(active)
                dt = time.time() - t0
                vals = self._evaluate(active, 1)
                still = []
                for (vl, va, vf), f in zip(vals, active):
                    t = self.trainers[f]
                    t.train_windows += n_train[f]
                    t.train_seconds += dt
                    if not t._end_of_epoch(epoch, float(sums[f]) / n_train[f], dt, n_train[f], vl, va, vf):
                        still.append(f)
                    else:
                        t.finished_at = time.time() - t_start
                        torch.cuda.current_stream(dev).synchronize()       # its last launches are done before the side stream reads
                        pending[f] = side.submit(finish, f)
                active = still
            torch.cuda.current_stream(dev).synchronize()
            if last >= budget:
                for f in active:                                           # ran out of epochs without an early stop
                    pending[f] = side.submit(finish, f)
                active = []
            if max_epochs > 0:
                return {f: fut.result() for f, fut in pending.items()}, active
            return [pending[f].result() for f in range(self.n)]

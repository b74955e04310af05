"""Lockstep training of several LOSO folds as ONE fold batch (include/msig.h msig_*_multi).

The reference trains its 15 folds one after the other (main.py:98-125); each is a fresh model with its own
data split, optimiser, scheduler and early stopping.  At its batch size (64 windows) a fold's train step is ~30
launches that each occupy a few CUs, and fifteen folds on fifteen streams are bound by the command processor's
dispatch rate, not by the CUs.  Here the folds of a rank advance in lockstep: every launch of the step covers all
active folds (blockIdx.z = fold, per-fold arenas — runtime.FoldArena), one gather builds all their batches, one
device op accumulates all their losses, and there is one host sync per epoch.  Everything that is per fold in the
reference stays per fold — weights, BatchNorm statistics, shuffling order, dropout stream, learning-rate schedule,
early stopping, checkpoints, logs — and each fold's numbers are bit-identical to its stand-alone run
(tests/test_trainer_gpu.py::test_lockstep_folds_equal_sequential); folds that stop early leave the batch.
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
    """Folds can share launches when their train / val sets have equal sizes (the synthetic set; WESAD subjects differ by
    a few windows, then the caller falls back to one stream per fold), one SubjectStore and one model configuration."""
    if not (2 <= len(preps) <= L.MAX_FOLDS):
        return False
    tr0, va0, _ = preps[0]["loaders"]
    for p in preps:
        tr, va, _ = p["loaders"]
        if (len(tr.dataset) != len(tr0.dataset) or len(va.dataset) != len(va0.dataset) or tr.batch_size != tr0.batch_size
                or va.batch_size != va0.batch_size or tr.store.data_ptr() != tr0.store.data_ptr()
                or p["model"].in_channels != preps[0]["model"].in_channels or p["model"].num_classes != preps[0]["model"].num_classes
                or p["model"].dropout_p != preps[0]["model"].dropout_p):
            return False
    return True


class LockstepTrainer:
    def __init__(self, preps: List[dict], device):
        self.preps, self.device = preps, torch.device(device)
        tr0, va0, _ = preps[0]["loaders"]
        m0 = preps[0]["model"]
        self.n = len(preps)
        self.C, self.K, self.T = m0.in_channels, m0.num_classes, int(tr0.store.shape[2])
        self.arena = FoldArena(self.C, self.K, self.device, self.n, max(tr0.batch_size, va0.batch_size), self.T)
        self.trainers: List[Trainer] = []
        for slot, p in enumerate(preps):
            model = p["model"]
            model._engine = self.arena.engine(slot)            # the model's parameters become views into arena `slot`
            t = Trainer(model, p["fold_dir"], p["config"])
            model.engine()
            self.trainers.append(t)
        h0 = self.trainers[0].optimizer.hyper
        for t in self.trainers:
            h = t.optimizer.hyper
            if (h["betas"], h["eps"], h["weight_decay"]) != (h0["betas"], h0["eps"], h0["weight_decay"]) or t.epochs != self.trainers[0].epochs:
                raise ValueError("lockstep folds must share betas / eps / weight decay / epoch budget")
        self.acc = torch.zeros(self.n, device=self.device)
        self._layouts, self._eval_orders = {}, {}

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- host-side bookkeeping is kept off the per-step path: descriptors, layouts and dropout keys are prepared per epoch ----
    def _layout(self, B, training):
        key = (B, bool(training))
        if key not in self._layouts:
            off = L.workspace_layout(B, self.C, self.T, self.K, training)
            self._layouts[key] = (self.arena.across("ws", off[L.WS["LOSS"]], torch.float32)[:, 0], off,
                                  self.arena.batch(B, training, self.trainers[0].model.dropout_p if training else 0.0))
        return self._layouts[key]

    def _gather(self, loader, order_mat, i, b, m):
        """order_mat: (folds, n) int64 store positions of the epoch; gathers columns i .. i+b of every row."""
        wfl = loader.store.shape[1] * loader.store.shape[2]
        L.check(L.lib().msig_gather_windows_multi(loader.store.data_ptr(), loader.store_y.data_ptr(), order_mat.data_ptr() + 8 * i,
                                                  order_mat.shape[1], b, wfl, self.arena.ptr("x"), self.arena.ptr("y"), C.byref(m),
                                                  self._stream()), "msig_gather_windows_multi")

    def _train_epoch(self, active):
        arena, lib = self.arena, L.lib()
        trs = [self.trainers[f] for f in active]
        loaders = [self.preps[f]["loaders"][0] for f in active]
        order = torch.stack([ld.epoch_order() for ld in loaders]).contiguous()      # (folds, n): one stack per epoch
        n, bs = order.shape[1], loaders[0].batch_size
        n_steps = (n + bs - 1) // bs
        for t in trs:
            t.model.train()
        step0 = {t.optimizer.step_count for t in trs}
        assert len(step0) == 1, "lockstep folds must have taken the same number of optimiser steps"
        step0 = step0.pop()
        steps = np.arange(step0 + 1, step0 + 1 + n_steps)
        thr = L.dropout_threshold(trs[0].model.dropout_p)
        kg = [L.dropout_keys(t.model._seed, steps, 1) if thr else np.zeros(n_steps, np.uint32) for t in trs]
        kh = [L.dropout_keys(t.model._seed, steps, 2) if thr else np.zeros(n_steps, np.uint32) for t in trs]
        m = arena.multi(active, lr=[t.optimizer.hyper["lr"] for t in trs])           # slots and learning rates: per epoch
        h0 = trs[0].optimizer.hyper
        b1, b2, eps, wd = h0["betas"][0], h0["betas"][1], h0["eps"], h0["weight_decay"]
        ea, eas, st, nf = arena.ptr("exp_avg"), arena.ptr("exp_avg_sq"), self._stream(), len(active)
        self.acc.zero_()
        for k in range(n_steps):
            i = k * bs
            b = min(bs, n - i)
            for j in range(nf):
                m.key_gru[j] = int(kg[j][k]); m.key_head[j] = int(kh[j][k])
            self._gather(loaders[0], order, i, b, m)
            loss, _, desc = self._layout(b, True)
            L.check(lib.msig_train_step_multi(C.byref(desc), C.byref(m), ea, eas, b1, b2, eps, wd, int(steps[k]), st), "msig_train_step_multi")
            self.acc.add_(loss, alpha=float(b))          # every arena's batch loss in one op (inactive arenas: stale values, ignored)
        for t in trs:
            t.optimizer.step_count = step0 + n_steps
        return self.acc.cpu().numpy().astype(np.float64)      # the epoch's only sync

    def _evaluate(self, active, which):
        """Validation pass of every active fold (loader index `which`): per fold (loss, acc, f1)."""
        arena, lib = self.arena, L.lib()
        loaders = [self.preps[f]["loaders"][which] for f in active]
        for f in active:
            self.trainers[f].model.eval()
        key = (which, tuple(active))
        if key not in self._eval_orders:                   # validation order is fixed (no shuffling): stack it once per active set
            self._eval_orders = {key: torch.stack([ld.epoch_order() for ld in loaders]).contiguous()}
        order = self._eval_orders[key]
        n, bs = order.shape[1], loaders[0].batch_size
        m = arena.multi(active)
        st = self._stream()
        self.acc.zero_()
        preds = []
        for i in range(0, n, bs):
            b = min(bs, n - i)
            self._gather(loaders[0], order, i, b, m)
            loss, off, desc = self._layout(b, False)
            L.check(lib.msig_forward_multi(C.byref(desc), C.byref(m), st), "msig_forward_multi")
            self.acc.add_(loss, alpha=float(b))
            preds.append(arena.across("ws", off[L.WS["PRED"]], torch.int32, b)[active])       # (folds, b) copy
        sums = self.acc.cpu().numpy().astype(np.float64)
        pred = torch.cat(preds, dim=1).cpu().numpy().astype(np.int64)
        out = []
        for i, f in enumerate(active):
            ds = loaders[i].dataset
            acc, f1 = accuracy_and_weighted_f1(np.asarray(ds.labels).astype(np.int64), pred[i])
            out.append((float(sums[f]) / len(ds), acc, f1))
        return out

    def run(self):
        """Trains every fold to its early stop, then evaluates each on its test subject; returns main.train_fold's dicts."""
        import json
        from concurrent.futures import ThreadPoolExecutor
        for t, p in zip(self.trainers, self.preps):
            for ld in p["loaders"]:
                t._check_labels(ld)
        t_start = time.time()
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
                        epochs=len(t.history), train_windows_per_s=t.train_windows / max(t.train_seconds, 1e-9))
            (p["fold_dir"] / "fold_result.json").write_text(json.dumps(info))
            return info

        active = list(range(self.n))
        n_train = len(self.preps[0]["loaders"][0].dataset)
        pending = {}
        with ThreadPoolExecutor(max_workers=2) as side:
            for epoch in range(self.trainers[0].epochs):
                if not active:
                    break
                t0 = time.time()
                sums = self._train_epoch(active)
                dt = time.time() - t0
                vals = self._evaluate(active, 1)
                still = []
                for (vl, va, vf), f in zip(vals, active):
                    t = self.trainers[f]
                    t.train_windows += n_train
                    t.train_seconds += dt
                    if not t._end_of_epoch(epoch, float(sums[f]) / n_train, dt, n_train, vl, va, vf):
                        still.append(f)
                    else:
                        t.finished_at = time.time() - t_start
                        torch.cuda.current_stream(dev).synchronize()       # its last launches are done before the side stream reads
                        pending[f] = side.submit(finish, f)
                active = still
            torch.cuda.current_stream(dev).synchronize()
            for f in active:                                               # ran out of epochs without an early stop
                pending[f] = side.submit(finish, f)
            return [pending[f].result() for f in range(self.n)]

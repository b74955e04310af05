"""Drop-in for the reference's ``trainer.py`` driving the fused HIP train step.

Public surface kept (reference ``trainer.py:12-273``): ``EarlyStopping(patience, delta,
checkpoint_path, verbose, log_func)``; ``Trainer(model, fold_output_dir, config)`` with
``.train(train_loader, val_loader)`` and ``.evaluate(loader, is_test, is_val)`` returning
``(loss, acc, f1)`` / ``(loss, acc, f1, preds, labels)``; the same ``config['trainer']``
keys; the same artefacts (``training_log.txt``, ``best_model.pt``,
``test_confusion_matrix.png``) and log line format (plus a windows/s figure).

What differs is where the work happens: one ``msig_train_step`` launch sequence per
mini-batch (zero_grad + forward + CE + backward + Adam, trainer.py:144-149) with no
``.item()`` in the loop — the epoch loss is accumulated on the device and read once per
epoch — and evaluation reads predictions back once per call.

Reference behaviours reproduced on purpose: EarlyStopping treats a
HIGHER monitored value as better although it is fed the validation loss; the class-weight
option is accepted and ignored (the reference's branch is unreachable, trainer.py:81).
"""
from __future__ import annotations

import os
import threading
import time
from pathlib import Path

import numpy as np
import torch
from torch.optim.lr_scheduler import ReduceLROnPlateau

from . import _lib as L
from .models import CnnGruAttentionModel


_PLOT_LOCK = threading.Lock()      # pyplot is not thread-safe; folds may run concurrently


class EarlyStopping:
    """trainer.py:12-39.  `score >= best + delta` counts as an improvement (checkpoint, reset
    the counter); anything lower increments the counter."""

    def __init__(self, patience=7, delta=0, checkpoint_path="checkpoint.pt", verbose=False, log_func=None):
        self.patience, self.delta, self.checkpoint_path = patience, delta, checkpoint_path
        self.verbose, self.log_func = verbose, log_func
        self.counter, self.best_score, self.early_stop = 0, None, False

    def __call__(self, score, model):
        improved = self.best_score is None or not (score < self.best_score + self.delta)
        if improved:
            self.best_score = score
            self.save_checkpoint(model)
            self.counter = 0
            return
        self.counter += 1
        if self.verbose and self.log_func:
            self.log_func(f"EarlyStopping counter: {self.counter}/{self.patience}")
        if self.counter >= self.patience:
            self.early_stop = True

    def save_checkpoint(self, model):
        # the parameters are views into larger flat buffers (one per model, or one arena for a whole fold batch): save
        # copies, not the buffers behind them
        torch.save({k: v.detach().clone() for k, v in model.state_dict().items()}, self.checkpoint_path)


class MsigAdam(torch.optim.Optimizer):
    """torch.optim.Adam(lr, weight_decay) semantics on the model's flat buffers via
    msig_adam_step.  Exists so that `trainer.optimizer` / ReduceLROnPlateau keep working."""

    def __init__(self, model: CnnGruAttentionModel, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__([p for p in model.parameters()], dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.model = model
        self.step_count = 0

    def add_param_group(self, param_group):
        """torch decorates Optimizer.add_param_group with torch._disable_dynamo, whose wrapper imports torch._dynamo at its first
        call: ~0.9 s of interpreter time in front of the first train step of a run whose 15 folds take 6 s (profiles/
        r04_loso_profile.log), for a compiler this path never uses.  The undecorated function does the same bookkeeping."""
        plain = getattr(torch.optim.Optimizer.add_param_group, "__wrapped__", None)
        return plain(self, param_group) if plain is not None else super().add_param_group(param_group)

    @property
    def hyper(self):
        return self.param_groups[0]

    @torch.no_grad()
    def step(self, closure=None):
        """Un-fused use (after loss.backward()): gathers p.grad into the flat gradient buffer."""
        eng = self.model.engine()
        for p, gv in zip(self.model._named(), self.model._grad_views()):
            if p.numel() and p.grad is not None:
                gv.copy_(p.grad)
        self.step_count += 1
        h = self.hyper
        eng.adam_step(h["lr"], h["betas"], h["eps"], h["weight_decay"], self.step_count)


def accuracy_and_weighted_f1(y_true: np.ndarray, y_pred: np.ndarray):
    """sklearn accuracy_score and f1_score(average='weighted') (trainer.py:234-235): per-class
    F1 weighted by true support; a class with no predicted and no true samples scores 0."""
    y_true, y_pred = np.asarray(y_true).astype(np.int64), np.asarray(y_pred).astype(np.int64)
    n = y_true.size
    acc = float((y_true == y_pred).mean()) if n else 0.0
    f1 = 0.0
    for c in np.unique(y_true):
        tp = float(np.sum((y_true == c) & (y_pred == c)))
        fp = float(np.sum((y_true != c) & (y_pred == c)))
        fn = float(np.sum((y_true == c) & (y_pred != c)))
        denom = 2 * tp + fp + fn
        f1 += (np.sum(y_true == c) / n) * (2 * tp / denom if denom > 0 else 0.0)
    return acc, float(f1)


class Trainer:
    def __init__(self, model, fold_output_dir: Path, config):
        self.model, self.fold_dir, self.config = model, Path(fold_output_dir), config
        self.fold_dir.mkdir(parents=True, exist_ok=True)
        self.log_file = self.fold_dir / "training_log.txt"
        with open(self.log_file, "w") as f:
            f.write(f"Training log for run starting at {time.strftime('%Y-%m-%d %H:%M:%S')}\n" + "=" * 50 + "\n")
        if not torch.cuda.is_available():
            raise RuntimeError("Trainer needs an AMD GPU: the multimodalsignal_amd path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.model.to(self.device)
        cfg = config["trainer"]
        self.epochs, self.learning_rate = cfg["epochs"], cfg["learning_rate"]
        self.patience, self.weight_decay = cfg["early_stopping"]["patience"], cfg["weight_decay"]
        self.use_class_weights = cfg.get("use_class_weights", False)      # accepted, inert (trainer.py:81)
        self.verbose = cfg.get("verbose", True)
        self.optimizer = MsigAdam(self.model, lr=self.learning_rate, weight_decay=self.weight_decay)   # trainer.py:68
        self.scheduler = ReduceLROnPlateau(self.optimizer, mode="min", factor=0.1, patience=3)         # trainer.py:72-77
        self.early_stopping = None
        if cfg["early_stopping"]["enabled"]:
            self.early_stopping = EarlyStopping(patience=self.patience, delta=cfg["early_stopping"]["delta"],
                                                checkpoint_path=self.fold_dir / "best_model.pt", verbose=True, log_func=self._log)
        self.history = []
        self._labels_ok = set()
        self.total_start_time = time.time()
        self.train_windows = 0
        self.train_seconds = 0.0

    def _log(self, message):
        if self.verbose:
            print(message)
        with open(self.log_file, "a") as f:
            f.write(message + "\n")

    def _to_device(self, inputs, labels):
        if isinstance(inputs, (list, tuple)):
            raise TypeError("list/tuple inputs (the reference's retired HybridDataset, trainer.py:135-140) are not supported")
        return inputs.to(self.device, non_blocking=True), labels.to(self.device, non_blocking=True)

    def _check_labels(self, loader):
        """A class id outside [0, num_classes) would index past the logits row in the loss kernel (torch's
        CrossEntropyLoss raises for it, trainer.py:147); checked once per dataset, on its host-side label vector —
        e.g. CLASSIFICATION_MODE 'ternary' with NUM_CLASSES left at 2 (main.py:22-23)."""
        ds = getattr(loader, "dataset", None)
        labels = getattr(ds, "labels", None)
        if labels is None or id(ds) in self._labels_ok:
            return
        lab = np.asarray(labels)
        if lab.size and (lab.min() < 0 or lab.max() >= self.model.num_classes):
            raise ValueError(f"label {int(lab.max() if lab.max() >= self.model.num_classes else lab.min())} is outside "
                             f"[0, {self.model.num_classes}): dataset labels do not match the model's num_classes")
        self._labels_ok.add(id(ds))

    # ---- trainer.py:119-191 -------------------------------------------------------------------
    def train(self, train_loader, val_loader):
        eng = self.model.engine()
        self._check_labels(train_loader)
        self._check_labels(val_loader)
        n_train = len(train_loader.dataset)
        for epoch in range(self.epochs):
            t0 = time.time()
            self.model.train()
            eng.loss_acc.zero_()
            for inputs, labels in train_loader:
                x, y = self._to_device(inputs, labels)
                h = self.optimizer.hyper
                self.optimizer.step_count += 1
                eng.train_step(x, y, lr=h["lr"], betas=h["betas"], eps=h["eps"], weight_decay=h["weight_decay"],
                               step=self.optimizer.step_count, dropout_p=self.model.dropout_p, seed=self.model._seed)
                # running_loss += loss.item() * batch (trainer.py:152) happens inside the step: the loss kernel adds to eng.loss_acc
            train_loss = float(eng.loss_acc[0].item()) / n_train                        # the epoch's only sync
            dt = time.time() - t0
            self.train_windows += n_train
            self.train_seconds += dt
            val_loss, val_acc, val_f1, _, _ = self.evaluate(val_loader, is_val=True)
            if self._end_of_epoch(epoch, train_loss, dt, n_train, val_loss, val_acc, val_f1):
                break
        self._finish_training()

    def _end_of_epoch(self, epoch, train_loss, dt, n_train, val_loss, val_acc, val_f1) -> bool:
        """Scheduler step, history, log line, early stopping (trainer.py:160-185); True = stop training."""
        self.scheduler.step(val_loss)
        self.history.append(dict(epoch=epoch + 1, train_loss=train_loss, val_loss=val_loss, val_acc=val_acc, val_f1=val_f1,
                                 lr=self.optimizer.hyper["lr"], seconds=dt))
        self._log(f"Epoch {epoch + 1}/{self.epochs} | 耗时: {dt:.2f}s | 训练损失: {train_loss:.4f} | 验证损失: {val_loss:.4f} | "
                  f"验证Acc: {val_acc:.4f} | 验证F1: {val_f1:.4f} | {n_train / max(dt, 1e-9):.0f} windows/s")
        if self.early_stopping:
            self.early_stopping(val_loss, self.model)
            if self.early_stopping.early_stop:
                self._log("触发早停")
                return True
        return False

    def _finish_training(self):
        if self.early_stopping and self.early_stopping.early_stop:
            self._log(f"加载性能最佳的模型权重从: {self.early_stopping.checkpoint_path}")
            self.model.load_state_dict(torch.load(self.early_stopping.checkpoint_path, weights_only=True))
        self._log(f"--- 训练完成 --- 总训练时长: {time.time() - self.total_start_time:.2f}秒")

    # ---- trainer.py:193-247 -------------------------------------------------------------------
    def evaluate(self, data_loader, is_test=False, is_val=False):
        eng = self.model.engine()
        self._check_labels(data_loader)
        self.model.eval()
        eng.loss_acc.zero_()
        preds, labs = [], []
        for inputs, labels in data_loader:
            x, y = self._to_device(inputs, labels)
            eng.forward(x, y, training=False)              # the loss kernel adds loss * batch to eng.loss_acc (trainer.py:221)
            preds.append(eng.region("PRED", torch.int32, (y.shape[0],)).clone())
            labs.append(y.clone())          # DeviceLoader reuses its batch buffers
        all_preds = torch.cat(preds).cpu().numpy().astype(np.int64)
        all_labels = torch.cat(labs).cpu().numpy().astype(np.int64)
        loss = float(eng.loss_acc[0].item()) / len(data_loader.dataset)
        acc, f1 = accuracy_and_weighted_f1(all_labels, all_preds)
        if is_test:
            self.plot_confusion_matrix(all_labels, all_preds, filename="test_confusion_matrix.png")
            self._log("\n--- 最终测试结果 (模型原始输出) ---")
            self._log(f"测试损失: {loss:.4f} | 测试Acc: {acc:.4f} | 测试F1: {f1:.4f}")
            return loss, acc, f1
        if is_val:
            return loss, acc, f1, list(all_preds), list(all_labels)
        return loss, acc, f1

    def plot_confusion_matrix(self, true_labels, pred_labels, filename="confusion_matrix.png"):
        if os.environ.get("MSIG_NO_PLOT") == "1":          # diagnostic only (tools/profile_loso.py): what the PNGs cost the LOSO wall-clock
            return
        try:
            _PLOT_LOCK.acquire()
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            classes = np.unique(np.concatenate([true_labels, pred_labels]))
            cm = np.zeros((classes.size, classes.size), dtype=np.int64)
            for t, p in zip(true_labels, pred_labels):
                cm[np.searchsorted(classes, t), np.searchsorted(classes, p)] += 1
            names = ["Non-Stress", "Stress"] if len(np.unique(true_labels)) == 2 else ["Neutral/Baseline", "Amusement", "Stress/TSST"]
            fig, ax = plt.subplots(figsize=(8, 6))
            ax.imshow(cm, cmap="Blues")
            for i in range(cm.shape[0]):
                for j in range(cm.shape[1]):
                    ax.text(j, i, str(cm[i, j]), ha="center", va="center")
            ax.set_xticks(range(classes.size)); ax.set_yticks(range(classes.size))
            ax.set_xticklabels(names[:classes.size]); ax.set_yticklabels(names[:classes.size])
            ax.set_xlabel("Predicted Label"); ax.set_ylabel("True Label"); ax.set_title("Confusion Matrix")
            path = self.fold_dir / filename
            fig.savefig(path)
            plt.close(fig)
            self._log(f"混淆矩阵已保存至: {path}")
        except Exception as e:   # plotting must never fail a fold (trainer.py:272-273)
            self._log(f"保存混淆矩阵失败: {e}")
        finally:
            _PLOT_LOCK.release()

"""Drop-in for the reference's ``dataset.py`` plus a GPU-resident batch source.

``WesadDataset`` keeps the reference's constructor, attributes (``.data`` (N,T,C)
float64, ``.labels``) and item format ((C,T) float32, int64 scalar) and applies the same
label maps and per-subject normalisation (reference ``dataset.py:9-65``).  What is new is
``DeviceLoader``: the whole (N,C,T) fp32 store lives in HBM once (≈4 k windows < 1 GB),
shuffling is a device permutation and a batch is one ``msig_gather_windows`` launch —
replacing the per-item cast/permute + DataLoader collate + H2D copy of every step
(``dataset.py:62-65``, ``main.py:112-114``, ``trainer.py:140-142``).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

import numpy as np
import torch
from torch.utils.data import Dataset

from . import _lib as L

LABEL_MODES = ("stress_binary", "ternary", "amusement_binary")


def map_labels(y_raw: np.ndarray, mode: str) -> np.ndarray:
    """WESAD protocol labels 1=baseline 2=TSST 3=amusement 4=meditation -> class ids
    (dataset.py:29-34)."""
    if mode == "stress_binary":
        return (y_raw == 2).astype(np.int64)
    if mode == "ternary":
        out = np.zeros_like(y_raw, dtype=np.int64)
        out[y_raw == 3] = 1
        out[y_raw == 2] = 2
        return out
    if mode == "amusement_binary":
        # The second model of the hierarchical experiment (main.py:183-186) asks for this mode, which the reference's dataset.py
        # does not define (it raises).  Defined here as the obvious map: amusement (raw 3) -> 1, baseline
        # (raw 1) -> 0, every other window is dropped (label -1; WesadDataset removes those rows after the per-subject
        # normalisation, which like every other mode uses ALL of the subject's windows).
        out = np.full_like(y_raw, -1, dtype=np.int64)
        out[y_raw == 1] = 0
        out[y_raw == 3] = 1
        return out
    raise ValueError(f"Unknown classification_mode: {mode}")


def normalise_subject(x: np.ndarray, names) -> np.ndarray:
    """Per-subject, per-channel z-score over all of the subject's windows, std + 1e-8;
    the channel literally named 'chest_EDA' is log1p-transformed first (dataset.py:36-48).
    `x` is (N,T,C) float64 and is modified in place."""
    mu = x.mean(axis=(0, 1))
    sd = x.std(axis=(0, 1)) + 1e-8
    for ch, name in enumerate(names):
        if name == "chest_EDA":
            lg = np.log1p(x[:, :, ch])
            x[:, :, ch] = (lg - lg.mean()) / (lg.std() + 1e-8)
        else:
            x[:, :, ch] = (x[:, :, ch] - mu[ch]) / sd[ch]
    return x


class WesadDataset(Dataset):
    def __init__(self, data_path: Path, subjects: list, channels_to_use: list, all_channel_names: list,
                 classification_mode="stress_binary", cache: Optional[dict] = None):
        """`cache` (optional, not in the reference): a dict shared between datasets of one run; a
        subject's normalised windows depend only on that subject, so the 3 x 15 datasets of a LOSO run
        can load and normalise each subject once instead of 45 times."""
        data_path = Path(data_path)
        self.classification_mode = classification_mode
        self.data_list, self.labels_list = [], []
        cols = [all_channel_names.index(ch) for ch in channels_to_use]
        for sid in subjects:
            key = (str(data_path), sid, tuple(cols), classification_mode)
            if cache is not None and key in cache:
                x, y = cache[key]
            else:
                fx, fy = data_path / f"{sid}_X.npy", data_path / f"{sid}_y.npy"
                if not (fx.exists() and fy.exists()):
                    print(f"Warning: Skipping subject {sid} for data, file not found.")
                    continue
                x = np.load(fx)[:, :, cols]                       # fancy index -> private float64 copy
                y = map_labels(np.load(fy), classification_mode)
                x = normalise_subject(x, [all_channel_names[i] for i in cols])
                if (y < 0).any():                                  # amusement_binary: windows of the other protocol phases are dropped
                    x, y = x[y >= 0], y[y >= 0]
                if cache is not None:
                    cache[key] = (x, y)
            self.data_list.append(x)
            self.labels_list.append(y)
        if not self.data_list:
            raise ValueError(f"No data loaded for subjects: {subjects}. Check paths and data existence.")
        self.data = np.concatenate(self.data_list, axis=0)
        self.labels = np.concatenate(self.labels_list, axis=0)
        self._dev_cache = None

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, idx):
        x = torch.from_numpy(self.data[idx]).float().permute(1, 0)
        return x, torch.tensor(self.labels[idx], dtype=torch.long)

    def device_tensors(self, device):
        """(N,C,T) fp32 windows and (N,) int64 labels in HBM (uploaded once, then cached)."""
        device = torch.device(device)
        if self._dev_cache is None or self._dev_cache[0].device != device:
            x = torch.from_numpy(np.ascontiguousarray(self.data.transpose(0, 2, 1), dtype=np.float32))
            self._dev_cache = (x.to(device), torch.from_numpy(self.labels.astype(np.int64)).to(device))
        return self._dev_cache


class SubjectStore:
    """All subjects of a run, normalised once and resident in HBM as ONE (N_total, C, T) fp32 store.

    A subject's normalised windows depend only on that subject (dataset.py:36-48), so the 45 datasets of
    a LOSO run are just index subsets of this store: no per-fold host concatenation, no per-fold upload.
    `normalise="host"` reproduces the reference's float64 numpy arithmetic and casts to fp32 exactly like
    `__getitem__` (dataset.py:63); `normalise="device"` does the reduction, the optional log1p, the z-score
    and the (N,T,C)->(N,C,T) transposition on the GPU in float64 (torch ops on the raw upload)."""

    def __init__(self, data_path: Path, subjects: list, channels_to_use: list, all_channel_names: list,
                 classification_mode="stress_binary", device="cuda", normalise="host"):
        self.device = torch.device(device)
        data_path = Path(data_path)
        cols = [all_channel_names.index(ch) for ch in channels_to_use]
        names = [all_channel_names[i] for i in cols]
        if normalise not in ("host", "device"):
            raise ValueError(f"normalise must be 'host' or 'device', got {normalise!r}")
        if classification_mode == "amusement_binary":
            raise NotImplementedError("amusement_binary drops windows per subject: use WesadDataset (the hierarchical driver does)")
        xs, ys, self.ranges, start = [], [], {}, 0
        present = []
        for sid in subjects:
            fx, fy = data_path / f"{sid}_X.npy", data_path / f"{sid}_y.npy"
            if not (fx.exists() and fy.exists()):
                print(f"Warning: Skipping subject {sid} for data, file not found.")
                continue
            present.append((sid, fx, fy))

        def host_windows(fx):        # the reference's float64 arithmetic, then its fp32 cast (dataset.py:36-48, :63)
            x = normalise_subject(np.load(fx)[:, :, cols], names)
            return np.ascontiguousarray(x.transpose(0, 2, 1), dtype=np.float32)

        # A subject's windows depend on that subject alone: the host path reads and normalises a few subjects at a time on
        # threads (numpy's reductions and copies run outside the interpreter lock; 0.9 -> 0.35 s for 15 x 270 windows) and
        # uploads them in subject order — the store is the same bit for bit.
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 1))) as pool:
            host = pool.map(host_windows, [fx for _, fx, _ in present]) if normalise == "host" else iter(())
            for sid, fx, fy in present:
                y = map_labels(np.load(fy), classification_mode)
                if normalise == "host":
                    xd = torch.from_numpy(next(host)).to(self.device)
                else:
                    xd = normalise_subject_device(torch.from_numpy(np.load(fx)).to(self.device), cols, names)
                xs.append(xd)
                ys.append(torch.from_numpy(y.astype(np.int64)))
                self.ranges[sid] = (start, start + len(y))
                start += len(y)
        if not xs:
            raise ValueError(f"No data loaded for subjects: {subjects}. Check paths and data existence.")
        self.x = torch.cat(xs, dim=0).contiguous()
        self.labels_host = torch.cat(ys).numpy()
        self.y = torch.from_numpy(self.labels_host).to(self.device)

    def view(self, subjects: list) -> "StoreView":
        return StoreView(self, subjects)


def normalise_subject_device(raw: torch.Tensor, cols, names) -> torch.Tensor:
    """(N,T,C_all) raw float64 device tensor -> normalised (N,C,T) fp32 through msig_normalise_subject
    (float64 reduction, optional log1p, z-score, cast and transposition in HIP)."""
    if not raw.is_cuda or raw.dtype != torch.float64 or raw.dim() != 3:
        raise ValueError("normalise_subject_device needs a (N,T,C_all) float64 GPU tensor")
    raw = raw.contiguous()
    N, T, C_all = raw.shape
    mask = 0
    for c, name in enumerate(names):
        if name == "chest_EDA":
            mask |= 1 << c
    out = torch.empty((N, len(cols), T), dtype=torch.float32, device=raw.device)
    scratch = torch.empty(L.lib().msig_normalise_scratch_bytes(), dtype=torch.uint8, device=raw.device)
    carr = (C.c_int32 * len(cols))(*[int(c) for c in cols])
    st = C.c_void_p(torch.cuda.current_stream(raw.device).cuda_stream)
    L.check(L.lib().msig_normalise_subject(raw.data_ptr(), N, T, C_all, carr, len(cols), mask, out.data_ptr(), scratch.data_ptr(), st),
            "msig_normalise_subject")
    return out


class StoreView:
    """The windows of some subjects inside a SubjectStore; quacks like a WesadDataset for DeviceLoader
    and Trainer (`len`, `.labels`, `device_tensors`)."""

    def __init__(self, store: SubjectStore, subjects: list):
        idx = []
        for sid in subjects:
            if sid not in store.ranges:
                print(f"Warning: Skipping subject {sid} for data, file not found.")
                continue
            a, b = store.ranges[sid]
            idx.append(np.arange(a, b, dtype=np.int64))
        if not idx:
            raise ValueError(f"No data loaded for subjects: {subjects}. Check paths and data existence.")
        self.store = store
        self.index_host = np.concatenate(idx)
        self.index = torch.from_numpy(self.index_host).to(store.device)
        self.labels = store.labels_host[self.index_host]

    def __len__(self):
        return len(self.index_host)

    def __getitem__(self, i):
        j = int(self.index_host[i])
        return self.store.x[j].cpu(), torch.tensor(self.labels[i], dtype=torch.long)

    def device_tensors(self, device):
        return self.store.x, self.store.y


class DeviceLoader:
    """Iterates (x, y) device batches of a WesadDataset without touching the host per step.
    Same iteration contract as ``DataLoader(ds, batch_size, shuffle)`` (no drop_last)."""

    def __init__(self, dataset: WesadDataset, batch_size: int, shuffle: bool, device, seed: Optional[int] = None):
        self.dataset, self.batch_size, self.shuffle = dataset, int(batch_size), bool(shuffle)
        self.device = torch.device(device)
        self.store, self.store_y = dataset.device_tensors(self.device)
        self.index = getattr(dataset, "index", None)      # StoreView: positions inside a shared SubjectStore
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(torch.initial_seed() if seed is None else seed)
        self._bufs = {}

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def epoch_order(self) -> torch.Tensor:
        """Store positions of this epoch's windows, in visiting order (advances the shuffler like one `iter()`)."""
        n = len(self.dataset)
        order = torch.randperm(n, device=self.device, generator=self.gen) if self.shuffle else torch.arange(n, device=self.device)
        if self.index is not None:
            order = self.index[order]
        return order

    def __iter__(self):
        n = len(self.dataset)
        order = self.epoch_order()
        wfl = self.store.shape[1] * self.store.shape[2]
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        for i in range(0, n, self.batch_size):
            idx = order[i:i + self.batch_size].contiguous()
            b = idx.numel()
            if b not in self._bufs:   # two alternating buffers per batch size: the previous batch may still be in flight
                self._bufs[b] = [(torch.empty((b,) + tuple(self.store.shape[1:]), device=self.device),
                                  torch.empty(b, dtype=torch.int64, device=self.device)) for _ in range(2)]
            ox, oy = self._bufs[b][(i // self.batch_size) & 1]
            L.check(L.lib().msig_gather_windows(self.store.data_ptr(), self.store_y.data_ptr(), idx.data_ptr(), b, wfl,
                                                ox.data_ptr(), oy.data_ptr(), st), "msig_gather_windows")
            yield ox, oy

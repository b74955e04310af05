"""Fold sharding for Leave-One-Subject-Out cross-validation.

The 15 folds of the reference's loop (main.py:98-125) share nothing, so they are dealt to
ranks round-robin (fold k -> rank k mod world), one process per GPU, no collective on the
data path.  The only exchange is one all_gather of a (max_local_folds, 3) float64 tensor
[fold index, accuracy, weighted F1] per rank — RCCL ("nccl") on GPUs, gloo in CPU tests."""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch


def folds_for_rank(n_folds: int, world: int, rank: int) -> List[int]:
    return list(range(rank, n_folds, world))


def split_train_val(subjects: Sequence[str], test_subject: str, seed: int = 42):
    """11/3 train/validation split of the 14 remaining subjects: sklearn
    train_test_split(test_size=0.2, random_state=seed) (main.py:102-103)."""
    # train_test_split is ShuffleSplit(n_splits=1): n_test = ceil(0.2 n), one RandomState(seed).permutation(n), test = its first
    # n_test entries, train = the rest in permutation order.  Restated (the sklearn import alone costs ~0.6 s of a 10 s LOSO run);
    # the split table of all 15 folds is pinned to the reference's own in tests/test_host_logic.py (golden loso_splits.json).
    import math
    import numpy as np
    rest = [s for s in subjects if s != test_subject]
    n = len(rest)
    n_test = int(math.ceil(0.2 * n))
    perm = np.random.RandomState(seed).permutation(n)
    return [rest[i] for i in perm[n_test:]], [rest[i] for i in perm[:n_test]]


def gather_fold_metrics(local: Dict[int, tuple], n_folds: int, world: int, device) -> Dict[int, tuple]:
    """local: {fold index: (accuracy, f1)} -> the same dict for ALL folds on every rank."""
    if world == 1:
        return dict(local)
    import torch.distributed as dist
    max_local = (n_folds + world - 1) // world
    mine = torch.full((max_local, 3), -1.0, dtype=torch.float64, device=device)
    for i, (k, (acc, f1)) in enumerate(sorted(local.items())):
        mine[i, 0], mine[i, 1], mine[i, 2] = float(k), float(acc), float(f1)
    bucket = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bucket, mine)
    out = {}
    for t in bucket:
        for row in t.cpu().tolist():
            if row[0] >= 0:
                out[int(row[0])] = (row[1], row[2])
    return out

"""CPU tier, world_size 2 over gloo: the fold-metric exchange of the sharded LOSO loop."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimodalsignal_amd.loso import folds_for_rank, gather_fold_metrics
    n_folds = 15
    local = {k: (0.5 + k / 100.0, 0.25 + k / 200.0) for k in folds_for_rank(n_folds, world, rank)}
    allm = gather_fold_metrics(local, n_folds, world, torch.device("cpu"))
    ok = sorted(allm) == list(range(n_folds)) and all(abs(allm[k][0] - (0.5 + k / 100.0)) < 1e-12 and
                                                       abs(allm[k][1] - (0.25 + k / 200.0)) < 1e-12 for k in allm)
    ret[rank] = bool(ok) and len(local) in (7, 8)
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_fold_metrics_gather_world2():
    world, port = 2, 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_single_rank_gather_is_identity():
    from multimodalsignal_amd.loso import gather_fold_metrics
    local = {0: (0.9, 0.8), 3: (0.7, 0.6)}
    assert gather_fold_metrics(local, 15, 1, torch.device("cpu")) == local

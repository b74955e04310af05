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


# ---- the sharded driver end to end (main.run_experiments, world 2): CPU stubs stand in for the GPU-side fold body ----
def _driver_worker(rank, world, port, out_dir, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimodalsignal_amd import main as M

    class _Store:                      # SubjectStore stand-in: nothing to load on a CPU-only box
        def __init__(self, *a, **k):
            pass

    def _prepare(fold_idx, subject, run_output_dir, device, names, cfg, cache=None):
        return dict(fold=fold_idx, subject=subject)

    def _train(prep, device):          # deterministic per-fold metrics, so every rank can check every row
        k = prep["fold"]
        return dict(subject=prep["subject"], accuracy=0.60 + k / 100.0, f1_score=0.50 + k / 200.0, seconds=0.0, epochs=k + 1,
                    train_windows_per_s=1.0)

    M.SubjectStore, M.prepare_fold, M.train_fold = _Store, _prepare, _train
    cfg = M.default_cfg()
    cfg.update(concurrent_folds=1, gather_device=torch.device("cpu"))
    results, wall = M.run_simple_experiment(out_dir, torch.device("cpu"), ["chest_ECG", "chest_EDA", "chest_Resp"], cfg, rank, world)
    ok = [r["subject"] for r in results] == M.ALL_SUBJECTS                       # 15 rows, in subject order, on every rank
    ok = ok and all(abs(r["accuracy"] - (0.60 + k / 100.0)) < 1e-12 and abs(r["f1_score"] - (0.50 + k / 200.0)) < 1e-12
                    for k, r in enumerate(results))
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_sharded_driver_world2_writes_one_summary(tmp_path):
    """main.py:98-156 sharded over two ranks: folds dealt round-robin, one all_gather, 15 rows in subject order on both
    ranks and ONE cv_summary.txt, written by rank 0, listing every subject (the reference's summary, main.py:129-156)."""
    world, port = 2, 31500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_driver_worker, args=(world, port, str(tmp_path), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}
    summaries = list(tmp_path.rglob("cv_summary.txt"))
    assert len(summaries) == 1
    text = summaries[0].read_text(encoding="utf-8")
    from multimodalsignal_amd.main import ALL_SUBJECTS
    rows = [ln for ln in text.splitlines() if ln.strip().startswith("- 测试")]
    assert [ln.split()[2].rstrip(":") for ln in rows] == ALL_SUBJECTS
    assert "Accuracy = 0.6000" in rows[0] and "Accuracy = 0.7400" in rows[14] and "on 2 GPU(s)" in text

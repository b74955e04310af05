"""GPU tier: SURVEY.md section 4 tier 4 — 1-GPU and N-GPU runs of the same folds with the same per-fold seeds produce identical
per-fold metrics.  Two real ranks share the box's one GPU (gloo in place of RCCL); each rank runs its own SubjectStore, fold
batches (LockstepTrainer) and the metric gather, exactly as on an 8-GPU node."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _run_main(world, out, data, extra=()):
    env = dict(os.environ, MSIG_DIST_BACKEND="gloo", PYTHONPATH=str(ROOT))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    args = ["--synthetic", str(data), "--synthetic-windows", "30", "--window-spread", "4", "--samples", "256", "--difficulty", "4",
            "--subjects", "S2", "S3", "S4", "S5", "S6", "S7", "S8", "--epochs", "8", "--patience", "1", "2", "3", "4", "--batch-size", "16",
            "--out", str(out), *extra]
    if world == 1:
        cmd = [sys.executable, "-m", "multimodalsignal_amd.main", *args]
    else:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "-m", "multimodalsignal_amd.main", *args]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=800, cwd=str(ROOT))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    runs = sorted(Path(out).glob("simple_binary/run_*"))
    assert len(runs) == 1, runs
    return runs[0]


@pytest.mark.timeout(1500)
def test_fold_results_do_not_depend_on_sharding(tmp_path):
    data = tmp_path / "w"
    r1 = _run_main(1, tmp_path / "o1", data)
    r2 = _run_main(2, tmp_path / "o2", data)          # folds dealt round-robin: rank 0 gets 4, rank 1 gets 3
    subs = ["S2", "S3", "S4", "S5", "S6", "S7", "S8"]
    rows = {}
    for tag, run in (("w1", r1), ("w2", r2)):
        infos = [json.loads((run / f"fold_test_on_{s}" / "fold_result.json").read_text()) for s in subs]
        rows[tag] = [(i["subject"], i["accuracy"], i["f1_score"], i["epochs"]) for i in infos]
        assert (run / "cv_summary.txt").read_text(encoding="utf-8").count("测试 S") == len(subs)
    assert rows["w1"] == rows["w2"]
    assert len({r[3] for r in rows["w1"]}) > 1, "folds should stop at different epochs for this test to bite"
    for s in subs:
        a = torch.load(r1 / f"fold_test_on_{s}" / "best_model.pt", weights_only=True, map_location="cpu")
        b = torch.load(r2 / f"fold_test_on_{s}" / "best_model.pt", weights_only=True, map_location="cpu")
        for k in a:
            assert torch.equal(a[k], b[k]), (s, k)
    # the summary lines (mean +- std over folds) agree as well
    pick = lambda run: [ln for ln in (run / "cv_summary.txt").read_text(encoding="utf-8").splitlines() if ln.startswith("平均")]
    assert pick(r1) == pick(r2)

"""Diagnostic: wall-clock of the bench's synthetic 15-fold LOSO for several (kernel forms, lockstep groups) settings in one process.
usage: python tools/loso_groups.py auto:1 auto:2 auto:4 split:4"""
import os, sys, time, contextlib
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import torch
from multimodalsignal_amd import _lib as L
from multimodalsignal_amd import main as M
from multimodalsignal_amd.synth import CHANNELS6, make_synthetic_wesad
data = Path("/tmp/msig_bench_loso/data_w270_t3840_d2")
if not (data / "_channel_names.txt").exists():
    make_synthetic_wesad(data, windows_per_subject=270, T=3840, difficulty=2.0)
names = (data / "_channel_names.txt").read_text().split()
dev = torch.device("cuda:0")
for i, spec in enumerate(sys.argv[1:] or ["auto:4"]):
    form, groups = spec.split(":")
    L.set_kernel_form(*(("split", "split") if form == "split" else ("ws", "b3") if form == "ws" else ("auto", "auto")))
    cfg = M.default_cfg(); cfg.update(data_path=data, channels=list(CHANNELS6), lockstep_groups=int(groups))
    torch.manual_seed(cfg["seed"])
    with contextlib.redirect_stdout(sys.stderr):
        results, wall = M.run_simple_experiment(Path(f"/tmp/msig_bench_loso/groups_run{i}"), dev, names, cfg)
    import json
    infos = [json.loads(p.read_text()) for p in sorted(Path(f"/tmp/msig_bench_loso/groups_run{i}").glob("fold_test_on_*/fold_result.json"))]
    ep = sum(x["epochs"] for x in infos)
    print(f"{spec:10s} wall {wall:6.2f} s  mean acc {np.mean([r['accuracy'] for r in results]):.4f}  epochs {ep}  longest fold {max(x['epochs'] for x in infos)}"
          f"  train windows/s per fold {np.mean([x['train_windows_per_s'] for x in infos]):.0f}", flush=True)

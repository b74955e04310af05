"""Diagnostic: cProfile of the lockstep LOSO run on the bench's synthetic dataset (host-side overhead per super-step)."""
import cProfile, io, os, pstats, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from multimodalsignal_amd import main as M
from multimodalsignal_amd.synth import CHANNELS6, make_synthetic_wesad
data = Path("/tmp/msig_bench_loso/data_w270_t3840_d2")
if not (data / "_channel_names.txt").exists():
    make_synthetic_wesad(data, windows_per_subject=270, T=3840, difficulty=2.0)
names = (data / "_channel_names.txt").read_text().split()
cfg = M.default_cfg(); cfg.update(data_path=data, channels=list(CHANNELS6))
dev = torch.device("cuda:0")
pr = cProfile.Profile()
t0 = time.time(); pr.enable()
M.run_simple_experiment(Path("/tmp/msig_bench_loso/prof_run"), dev, names, cfg)
pr.disable(); print("wall", time.time() - t0)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35); print(s.getvalue())

"""Diagnostic: time of the raw preprocessing branch for one WESAD-sized recording (≈ 100 min of RespiBAN at 700 Hz,
8 columns -> 128 Hz, 60 s / 10 s windows) on the GPU path vs scipy on the host."""
import sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodalsignal_amd import preprocess as G

n = 700 * 60 * 100 + 317
rs = np.random.RandomState(0)
chest = {"ACC": rs.randn(n, 3), "ECG": rs.randn(n, 1), "EDA": rs.rand(n, 1) + 2, "EMG": rs.randn(n, 1), "Resp": rs.randn(n, 1), "Temp": 30 + rs.randn(n, 1)}
protocol = [("Base", 5.0, 25.0), ("TSST", 30.0, 42.0), ("Medi 1", 45.0, 52.0), ("Fun", 60.0, 66.5), ("Medi 2", 70.0, 77.0)]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    X, y = G.preprocess_recording(chest, protocol)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"GPU path (upload + resample + windows + download), run {rep}: {t1 - t0:.2f} s, X {X.shape}", flush=True)
rec = torch.from_numpy(np.concatenate([chest[c].reshape(n, -1) for c in G.CHEST_CHANNELS], axis=1)).cuda()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    yy = G.resample_device(rec, int(n * 128 / 700)); torch.cuda.synchronize()
    print(f"  resample alone on resident data (power-of-two plans cached): {time.perf_counter() - t0:.3f} s", flush=True)
from scipy import signal
t0 = time.perf_counter()
ref = [signal.resample(chest[c], int(n * 128 / 700)) if chest[c].shape[1] == 1 else
       np.column_stack([signal.resample(chest[c][:, i], int(n * 128 / 700)) for i in range(3)]) for c in G.CHEST_CHANNELS]
print(f"scipy.signal.resample on the host, 8 columns: {time.perf_counter() - t0:.2f} s")

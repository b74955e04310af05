"""Diagnostic: per-kernel HIP-event times (us per launch) of the fold-batched train step at B = 64 for F folds, latency forms.
usage: [KT_B=64] [KT_FWD=auto|split|ws] [KT_BWD=auto|split] python tools/kernel_times.py F [filter-substring]"""
import ctypes as C, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from multimodalsignal_amd import _lib as L
from multimodalsignal_amd.runtime import FoldArena
F = int(sys.argv[1]); flt = sys.argv[2] if len(sys.argv) > 2 else ""
dev = torch.device("cuda:0")
L.set_kernel_form(os.environ.get("KT_FWD", "auto"), os.environ.get("KT_BWD", "auto"))
B = int(os.environ.get("KT_B", "64"))
ar = FoldArena(6, 2, dev, F, B, 3840)
for s in range(F):
    ar.engine(s).params.normal_(0, 0.05)
    ar.view(s, "x", torch.float32).normal_()
    ar.view(s, "y", torch.int64).random_(0, 2)
m = ar.multi(list(range(F)), [1] * F, [2] * F, [1e-3] * F)
desc = ar.batch(B, True, 0.5)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
def run(n, k0):
    for k in range(n):
        L.check(L.lib().msig_train_step_multi(C.byref(desc), C.byref(m), ar.ptr("exp_avg"), ar.ptr("exp_avg_sq"), 0.9, 0.999, 1e-8, 1e-4, k0 + k + 1, st), "step")
    torch.cuda.synchronize()
run(10, 0)
import time
t0 = time.perf_counter(); run(200, 10); dt = time.perf_counter() - t0
L.profile_enable(True); run(20, 300); rep = L.profile_report(); L.profile_enable(False)
print(f"F={F} B={B} [fwd {os.environ.get('KT_FWD', 'auto')}]: {1e3*dt/200:.3f} ms/step; " + ", ".join(f"{k} {1e3*ms/20:.1f}" for k, (c, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1]) if flt in k))

"""Diagnostic: ms per fold-batched train step (msig_train_step_multi) as a function of the number of folds, alone and with
several fold batches running concurrently on separate streams (what bounds the lockstep LOSO run?)."""
import ctypes as C, sys, threading, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
from multimodalsignal_amd import _lib as L
from multimodalsignal_amd.runtime import FoldArena
dev = torch.device("cuda:0")
Cc, K, T, B = 6, 2, 3840, 64
FORM = sys.argv[1] if len(sys.argv) > 1 else "auto"          # auto | split | ws (throughput forms: gru_fwd_ws + gru_bwd_b3)
L.set_kernel_form(*{"auto": ("auto", "auto"), "split": ("split", "split"), "ws": ("ws", "b3")}[FORM])
print(f"kernel forms: {FORM}", flush=True)
QUICK = len(sys.argv) > 2 and sys.argv[2] == "quick"


def make(F):
    ar = FoldArena(Cc, K, dev, F, B, T)
    for s in range(F):
        e = ar.engine(s)
        e.params.normal_(0, 0.05)
        ar.view(s, "x", torch.float32).normal_()
        ar.view(s, "y", torch.int64).random_(0, 2)
    return ar


def run(ar, F, n, stream):
    m = ar.multi(list(range(F)), [1] * F, [2] * F, [1e-3] * F)
    desc = ar.batch(B, True, 0.5)
    with torch.cuda.stream(stream):
        st = C.c_void_p(stream.cuda_stream)
        for k in range(n):
            L.check(L.lib().msig_train_step_multi(C.byref(desc), C.byref(m), ar.ptr("exp_avg"), ar.ptr("exp_avg_sq"), 0.9, 0.999, 1e-8, 1e-4, k + 1, st), "step")
        stream.synchronize()


for F in ((4, 8, 15) if QUICK else (1, 2, 5, 8, 15)):
    ar = make(F); s = torch.cuda.Stream(dev)
    run(ar, F, 20, s)
    t0 = time.perf_counter(); run(ar, F, 200, s); dt = time.perf_counter() - t0
    print(f"F={F:2d} alone: {1e3 * dt / 200:.3f} ms per super-step = {1e3 * dt / 200 / F:.3f} ms per fold-step", flush=True)
for G, F in (((2, 8), (4, 4)) if QUICK else ((3, 5), (2, 8), (5, 3), (15, 1))):
    ars = [make(F) for _ in range(G)]; ss = [torch.cuda.Stream(dev) for _ in range(G)]
    for a, s in zip(ars, ss): run(a, F, 10, s)
    ths = [threading.Thread(target=run, args=(a, F, 200, s)) for a, s in zip(ars, ss)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    print(f"{G} batches x {F} folds concurrently: {1e3 * dt / 200:.3f} ms per round of {G * F} fold-steps = {1e3 * dt / 200 / (G * F):.3f} ms per fold-step", flush=True)
# per-kernel times of one fold-batched step at F = 15 and F = 1 (HIP events around every launch; adds bubbles)
for F in ((15,) if QUICK else (15, 1)):
    ar = make(F); s = torch.cuda.Stream(dev)
    run(ar, F, 5, s)
    L.profile_enable(True)
    run(ar, F, 10, s)
    rep = L.profile_report(); L.profile_enable(False)
    tot = sum(ms for _, ms in rep.values()) / 10
    print(f"F={F}: sum of kernel times {tot:.3f} ms per step; " + ", ".join(f"{k} {ms / 10:.3f}" for k, (c, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1])[:16]))

"""Diagnostic: does a HIP stream priority buy a latency-bound fold its stand-alone step time when other folds run beside it?
G single-fold train-step loops (msig_train_step_multi, F = 1, B = 64) on G streams, stream 0 created with the HIGH priority
(PROBE_PRIO=1) or like the others (PROBE_PRIO=0); prints every stream's own time per step.
    PROBE_PRIO=1 python tools/stream_priority_probe.py [G ...]"""
import ctypes as C, os, sys, threading, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from multimodalsignal_amd import _lib as L
from multimodalsignal_amd.runtime import FoldArena
dev = torch.device("cuda:0")
prio = int(os.environ.get("PROBE_PRIO", "1"))
Fs = [int(f) for f in os.environ.get("PROBE_FOLDS", "1").split(",")]       # folds of stream 0, of the others


def make(F):
    ar = FoldArena(6, 2, dev, F, 64, 3840)
    for s in range(F):
        ar.engine(s).params.normal_(0, 0.05)
        ar.view(s, "x", torch.float32).normal_()
        ar.view(s, "y", torch.int64).random_(0, 2)
    return ar


def run(ar, F, n, stream, out, i):
    m = ar.multi(list(range(F)), [1] * F, [2] * F, [1e-3] * F)
    desc = ar.batch(64, True, 0.5)
    st = C.c_void_p(stream.cuda_stream)
    t0 = time.perf_counter()
    for k in range(n):
        L.check(L.lib().msig_train_step_multi(C.byref(desc), C.byref(m), ar.ptr("exp_avg"), ar.ptr("exp_avg_sq"), 0.9, 0.999, 1e-8, 1e-4, k + 1, st), "step")
    stream.synchronize()
    out[i] = (time.perf_counter() - t0) / n


print("priority range (least, greatest):", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
for G in [int(g) for g in (sys.argv[1:] or ["1", "2", "3", "4"])]:
    fl = [Fs[0]] + [Fs[-1]] * (G - 1)
    ars = [make(f) for f in fl]
    ss = [torch.cuda.Stream(dev, priority=-1 if (prio and i == 0) else 0) for i in range(G)]
    out = [0.0] * G
    for i, (a, s) in enumerate(zip(ars, ss)):
        run(a, fl[i], 20, s, out, i)
    n = [400] + [4000] * (G - 1)                   # the others keep running for as long as stream 0 is timed ...
    stop = threading.Event()
    ths = [threading.Thread(target=run, args=(a, fl[i], n[i] if i == 0 else 400, s, out, i)) for i, (a, s) in enumerate(zip(ars, ss))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    print(f"stream 0 {'HIGH' if prio else 'normal'} priority, {G} streams, folds {fl}: ms per step of each stream " + ", ".join(f"{1e3 * o:.3f}" for o in out), flush=True)

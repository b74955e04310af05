"""Diagnostic: G single-fold train-step loops (msig_train_step_multi, F = 1, B = 64, latency forms) on G HIP streams at once — what
does a latency-bound fold lose when other folds run beside it on other streams (the LOSO's late epochs: few folds left, one or
two per fold batch)?   python tools/stream_interference_probe.py [F]"""
import ctypes as C, os, sys, threading, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from multimodalsignal_amd import _lib as L
from multimodalsignal_amd.runtime import FoldArena
dev = torch.device("cuda:0")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L.set_kernel_form("split", "split")


def make():
    ar = FoldArena(6, 2, dev, F, 64, 3840)
    for s in range(F):
        ar.engine(s).params.normal_(0, 0.05)
        ar.view(s, "x", torch.float32).normal_()
        ar.view(s, "y", torch.int64).random_(0, 2)
    return ar


def run(ar, n, stream):
    m = ar.multi(list(range(F)), [1] * F, [2] * F, [1e-3] * F)
    desc = ar.batch(64, True, 0.5)
    st = C.c_void_p(stream.cuda_stream)
    for k in range(n):
        L.check(L.lib().msig_train_step_multi(C.byref(desc), C.byref(m), ar.ptr("exp_avg"), ar.ptr("exp_avg_sq"), 0.9, 0.999, 1e-8, 1e-4, k + 1, st), "step")
    stream.synchronize()


for G in [int(g) for g in os.environ.get("PROBE_STREAMS", "1,2,3,4,6,8").split(",")]:
    ars = [make() for _ in range(G)]
    ss = [torch.cuda.Stream(dev) for _ in range(G)]
    for a, s in zip(ars, ss):
        run(a, 20, s)
    ths = [threading.Thread(target=run, args=(a, 300, s)) for a, s in zip(ars, ss)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    print(f"[GPU_MAX_HW_QUEUES={os.environ['GPU_MAX_HW_QUEUES']}] {G} stream(s) x {F} fold(s): {1e3 * dt / 300:.3f} ms per step of every stream ({1e3 * dt / 300 / (G * F):.3f} ms per fold-step)", flush=True)

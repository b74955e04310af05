"""Diagnostic: N fold-batched train steps (msig_train_step_multi, latency forms) of F folds x B = 64 and nothing else — the program
to put under rocprofv3 when the question is what a kernel does inside a fold batch (tools/multi_step_probe.py times, this one
is for counters):   rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES ... -- python3 tools/fold_batch_steps.py 15 20"""
import ctypes as C, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from multimodalsignal_amd import _lib as L
from multimodalsignal_amd.runtime import FoldArena
F, N = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
L.set_kernel_form("split", "split")
ar = FoldArena(6, 2, dev, F, 64, 3840)
for s in range(F):
    ar.engine(s).params.normal_(0, 0.05)
    ar.view(s, "x", torch.float32).normal_()
    ar.view(s, "y", torch.int64).random_(0, 2)
m = ar.multi(list(range(F)), [1] * F, [2] * F, [1e-3] * F)
desc = ar.batch(64, True, 0.5)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for k in range(N):
    L.check(L.lib().msig_train_step_multi(C.byref(desc), C.byref(m), ar.ptr("exp_avg"), ar.ptr("exp_avg_sq"), 0.9, 0.999, 1e-8, 1e-4, k + 1, st), "step")
torch.cuda.synchronize()

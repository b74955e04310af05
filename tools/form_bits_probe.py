"""Diagnostic: are the forward forms bit-identical?  One fold-batched train step (B = 64, C = 6, T = 3840) from the same state
under MSIG forward forms `split` (projection + recurrence) and `ws` (wave-specialised), per layer where MSIG_DIAG_WS_LAYERS says.
Prints the number of differing words in parameters (after Adam), gradients, BatchNorm state and the loss accumulator."""
import ctypes as C, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from multimodalsignal_amd import _lib as L
from multimodalsignal_amd.runtime import FoldArena
F = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
def run(fwd):
    L.set_kernel_form(fwd, "split")
    torch.manual_seed(0)
    ar = FoldArena(6, 2, dev, F, 64, 3840)
    for s in range(F):
        ar.engine(s).params.normal_(0, 0.05)
        ar.view(s, "x", torch.float32).normal_()
        ar.view(s, "y", torch.int64).random_(0, 2)
    m = ar.multi(list(range(F)), [1] * F, [2] * F, [1e-3] * F)
    desc = ar.batch(64, True, 0.5)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for k in range(3):
        L.check(L.lib().msig_train_step_multi(C.byref(desc), C.byref(m), ar.ptr("exp_avg"), ar.ptr("exp_avg_sq"), 0.9, 0.999, 1e-8, 1e-4, k + 1, st), "step")
    torch.cuda.synchronize()
    return {k: torch.stack([ar.view(s, k, torch.int32).clone() for s in range(F)]) for k in ("params", "grads", "bn_state", "acc")}
a = run("split")
for other in ("ws", "auto"):
    b = run(other)
    for k in a:
        d = (a[k] != b[k])
        print(f"split vs {other}: {k}: {int(d.sum())} of {d.numel()} words differ")
        if k == "grads" and d.any():
            fa, fb = a[k].view(torch.float32), b[k].view(torch.float32)
            print("   max |diff| / max |grad| =", float((fa - fb).abs().max() / fa.abs().max()))

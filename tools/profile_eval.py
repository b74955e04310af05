"""Diagnostic: per-kernel times of the eval-mode forward (no stash stores) vs training forward."""
import sys, torch
sys.path.insert(0, ".")
from multimodalsignal_amd import _lib as L
from multimodalsignal_amd.models import CnnGruAttentionModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = CnnGruAttentionModel(6, 2).to(dev)
eng = m.engine()
x = torch.randn(B, 6, 3840, device=dev)
y = (torch.rand(B, device=dev) < 0.2).long()
for training in (False, True):
    for _ in range(2):
        eng.forward(x, y, training=training, dropout_p=0.5, seed=1, step=1)
    torch.cuda.synchronize()
    L.profile_enable(True)
    for _ in range(3):
        eng.forward(x, y, training=training, dropout_p=0.5, seed=1, step=1)
    torch.cuda.synchronize()
    rep = L.profile_report()
    L.profile_enable(False)
    print("training" if training else "eval", {k: round(v[1] / 3, 3) for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1])[:8]})

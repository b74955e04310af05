"""Stock-PyTorch CPU model with the reference's module graph.  TEST INFRASTRUCTURE ONLY.

Used for (a) the ``cpu_baseline`` leg of bench.py — what the reference's CPU path costs on
the GPU box's host cores (kind "port": the reference's own files do not travel) — and
(b) trainer-level parity tests that need a CPU model with the reference's state_dict.
It is the same eight stock ``torch.nn`` layers the reference composes (models.py:15-22,
45-71), assembled from a table instead of being spelled out; its state_dict keys and
outputs are checked against the golden vectors in tests/test_oracle_golden.py.
"""
import torch
import torch.nn as nn

_CNN_TABLE = (  # (kind, args) in cnn_encoder order, models.py:45-54
    ("conv", dict(out=16, k=7, s=2, p=3)), ("bn", {}), ("relu", {}), ("pool", {}),
    ("conv", dict(out=32, k=5, s=2, p=2)), ("bn", {}), ("relu", {}), ("pool", {}),
)


class _Gate(nn.Module):
    def __init__(self, C, r=4):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool1d(1)
        self.fc = nn.Sequential(nn.Linear(C, C // r, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(C // r, C, bias=False), nn.Sigmoid())

    def forward(self, x):
        s = self.fc(self.avg_pool(x).flatten(1))
        return x * s.unsqueeze(-1)


class CpuCnnGru(nn.Module):
    def __init__(self, in_channels, num_classes, hidden=64, layers=2, dropout=0.5):
        super().__init__()
        self.channel_attention = _Gate(in_channels)
        mods, cin = [], in_channels
        for kind, a in _CNN_TABLE:
            if kind == "conv":
                mods.append(nn.Conv1d(cin, a["out"], a["k"], a["s"], a["p"], bias=False))
                cin = a["out"]
            elif kind == "bn":
                mods.append(nn.BatchNorm1d(cin))
            elif kind == "relu":
                mods.append(nn.ReLU())
            else:
                mods.append(nn.MaxPool1d(3, 2, 1))
        self.cnn_encoder = nn.Sequential(*mods)
        self.gru = nn.GRU(cin, hidden, layers, batch_first=True, bidirectional=True, dropout=dropout if layers > 1 else 0)
        self.classifier = nn.Sequential(nn.Linear(2 * hidden, 64), nn.ReLU(), nn.Dropout(dropout), nn.Linear(64, num_classes))

    def forward(self, x):
        seq = self.cnn_encoder(self.channel_attention(x)).transpose(1, 2)
        out, _ = self.gru(seq)
        return self.classifier(out[:, -1])


def time_train_steps(batch=64, C=6, T=3840, K=2, budget_s=15.0, min_steps=3, threads=None):
    """Times full CPU train steps (fwd + CE + bwd + Adam, trainer.py:144-149) on synthetic
    N(0,1) windows for about `budget_s` seconds.  Returns dict(value, steps, threads, ms_per_step)."""
    import os
    import time
    if threads is None:
        # torch's default (= all hardware threads) oversubscribes this small model badly on many-core hosts
        # (128 threads: 3.8 s/step vs 0.64 s/step on 8): calibrate with one step per candidate, keep the fastest.
        ncpu = os.cpu_count() or 1
        cands = sorted({c for c in (8, 16, 32) if c <= ncpu}) or [ncpu]      # more threads only get slower on this model
    else:
        cands = [threads]
    torch.manual_seed(0)
    m = CpuCnnGru(C, K)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-4)
    crit = nn.CrossEntropyLoss()
    x = torch.randn(batch, C, T)
    y = (torch.rand(batch) < 0.2).long()
    m.train()

    def step():
        opt.zero_grad()
        loss = crit(m(x), y)
        loss.backward()
        opt.step()
        return loss.item()

    best = None
    for c in cands:
        torch.set_num_threads(c)
        step()  # warm-up at this thread count
        t1 = time.perf_counter()
        step()
        dt1 = time.perf_counter() - t1
        if best is None or dt1 < best[1]:
            best = (c, dt1)
    torch.set_num_threads(best[0])
    t0 = time.perf_counter()
    n = 0
    while n < min_steps or (time.perf_counter() - t0) < budget_s:
        step()
        n += 1
    dt = time.perf_counter() - t0
    return dict(value=batch * n / dt, steps=n, threads=torch.get_num_threads(), ms_per_step=1e3 * dt / n)

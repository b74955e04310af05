"""CPU oracle for the CnnGruAttentionModel hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement of the arithmetic that the reference
delegates to stock ``torch.nn`` modules.  It is imported only by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``; the
product (``multimodalsignal_amd``) never imports it and has no CPU fallback.

Parity status: PINNED.  The reference has no tests of its own, so
the pin is the reference itself, run in the build container by
``tests/golden/make_golden.py`` (imports ``/root/reference/models.py`` etc. and
stores inputs/outputs under ``tests/golden/``); ``tests/test_oracle_golden.py``
checks every function below against those vectors.

Every function cites the reference line it restates (paths are into
``/root/reference``).  All functions are dtype-generic (float32 or float64) and
are written with differentiable tensor ops, so ``torch.autograd`` on this
restatement is the backward oracle (the reference's backward is autograd over
the same graph, ``trainer.py:148``).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch

GRU_HIDDEN = 64
CNN1_OUT = 16
CNN2_OUT = 32
HEAD_HIDDEN = 64


# --------------------------------------------------------------------------- #
# Parameter inventory (models.py:39-71; state_dict key set in SURVEY.md §8a a8)
# --------------------------------------------------------------------------- #
def param_specs(C: int, K: int, hidden: int = GRU_HIDDEN, layers: int = 2) -> "OrderedDict[str, tuple]":
    """Learnable tensors in ``nn.Module.parameters()`` order (models.py:43-71).  ``hidden`` / ``layers``: gru_hidden_size /
    gru_num_layers (models.py:39-40); the hierarchical experiment's second model uses 32 / 1 (main.py:35-40)."""
    H, G = hidden, 3 * hidden
    specs = OrderedDict()
    specs["channel_attention.fc.0.weight"] = (C // 4, C)      # models.py:18
    specs["channel_attention.fc.2.weight"] = (C, C // 4)      # models.py:20
    specs["cnn_encoder.0.weight"] = (CNN1_OUT, C, 7)          # models.py:46
    specs["cnn_encoder.1.weight"] = (CNN1_OUT,)               # models.py:47
    specs["cnn_encoder.1.bias"] = (CNN1_OUT,)
    specs["cnn_encoder.4.weight"] = (CNN2_OUT, CNN1_OUT, 5)   # models.py:50
    specs["cnn_encoder.5.weight"] = (CNN2_OUT,)               # models.py:51
    specs["cnn_encoder.5.bias"] = (CNN2_OUT,)
    for layer, isz in ((0, CNN2_OUT), (1, 2 * H))[:layers]:   # models.py:56-63
        for sfx in ("", "_reverse"):
            specs[f"gru.weight_ih_l{layer}{sfx}"] = (G, isz)
            specs[f"gru.weight_hh_l{layer}{sfx}"] = (G, H)
            specs[f"gru.bias_ih_l{layer}{sfx}"] = (G,)
            specs[f"gru.bias_hh_l{layer}{sfx}"] = (G,)
    specs["classifier.0.weight"] = (HEAD_HIDDEN, 2 * H)       # models.py:67
    specs["classifier.0.bias"] = (HEAD_HIDDEN,)
    specs["classifier.3.weight"] = (K, HEAD_HIDDEN)           # models.py:70
    specs["classifier.3.bias"] = (K,)
    return specs


def buffer_specs() -> "OrderedDict[str, tuple]":
    specs = OrderedDict()
    for idx, ch in ((1, CNN1_OUT), (5, CNN2_OUT)):
        specs[f"cnn_encoder.{idx}.running_mean"] = (ch,)
        specs[f"cnn_encoder.{idx}.running_var"] = (ch,)
        specs[f"cnn_encoder.{idx}.num_batches_tracked"] = ()
    return specs


def stage_lengths(T: int):
    """Temporal sizes after each stage (models.py:46-53)."""
    L1 = (T + 2 * 3 - 7) // 2 + 1
    P1 = (L1 + 2 * 1 - 3) // 2 + 1
    L2 = (P1 + 2 * 2 - 5) // 2 + 1
    TP = (L2 + 2 * 1 - 3) // 2 + 1
    return L1, P1, L2, TP


# --------------------------------------------------------------------------- #
# Counter-based dropout mask shared with the HIP kernels (design of this repo;
# the reference uses torch's Philox stream, which cannot be reproduced — see
# SURVEY.md §5.1-7 — so parity under dropout is exact only against THIS mask).
# --------------------------------------------------------------------------- #
def _fmix32(h: np.ndarray) -> np.ndarray:
    h = h.astype(np.uint32)
    h ^= h >> np.uint32(16)
    h = (h * np.uint32(0x85EBCA6B)).astype(np.uint32)
    h ^= h >> np.uint32(13)
    h = (h * np.uint32(0xC2B2AE35)).astype(np.uint32)
    h ^= h >> np.uint32(16)
    return h


def dropout_key(seed: int, step: int, stream: int) -> int:
    """32-bit per-(seed, step, stream) key; computed on the host in the product."""
    with np.errstate(over="ignore"):
        lo = np.uint32(seed & 0xFFFFFFFF)
        hi = np.uint32((seed >> 32) & 0xFFFFFFFF)
        a = np.uint32((step * 0x9E3779B9) & 0xFFFFFFFF)
        b = np.uint32((stream * 0x7F4A7C15) & 0xFFFFFFFF)
        inner = _fmix32(np.array([(int(a) + int(b) + int(hi)) & 0xFFFFFFFF], dtype=np.uint32))[0]
        return int(_fmix32(np.array([lo ^ inner], dtype=np.uint32))[0])


def dropout_threshold(p: float) -> int:
    """p is quantised to 1/256: an element is kept iff its byte >= thr."""
    return int(round(float(p) * 256.0))


def dropout_keep(key: int, n: int, thr: int) -> np.ndarray:
    """keep[e] for flat element index e: byte (e&3) of fmix32((e>>2) ^ key) >= thr."""
    idx = np.arange(n, dtype=np.uint64)
    word = _fmix32(((idx >> np.uint64(2)).astype(np.uint32)) ^ np.uint32(key))
    byte = (word >> ((idx & np.uint64(3)).astype(np.uint32) * np.uint32(8))) & np.uint32(0xFF)
    return byte >= np.uint32(thr)


def dropout_scale(thr: int) -> float:
    return 0.0 if thr >= 256 else 256.0 / (256.0 - thr)


STREAM_GRU = 1   # inter-layer GRU dropout (models.py:62)
STREAM_HEAD = 2  # classifier dropout (models.py:69)


# --------------------------------------------------------------------------- #
# Forward pieces
# --------------------------------------------------------------------------- #
def channel_gate(x, W1, W2):
    """ChannelAttention.forward, models.py:24-31 (without the final multiply)."""
    m = x.mean(dim=2)                                  # AdaptiveAvgPool1d(1), :28
    a1 = m @ W1.t()                                    # Linear(C, C//4), :18
    hid = torch.clamp_min(a1, 0)                       # ReLU, :19
    s = torch.sigmoid(hid @ W2.t())                    # Linear + Sigmoid, :20-21
    return m, a1, s


def conv1d_strided(x, w, stride, pad):
    """nn.Conv1d(bias=False) as an explicit gather + contraction (models.py:46,50)."""
    B, Cin, L = x.shape
    Cout, _, Kw = w.shape
    Lout = (L + 2 * pad - Kw) // stride + 1
    zpad = torch.zeros(B, Cin, pad, dtype=x.dtype)
    xp = torch.cat([zpad, x, zpad], dim=2) if pad else x              # zero padding
    idx = (torch.arange(Lout)[:, None] * stride + torch.arange(Kw)[None, :])  # (Lout,Kw)
    cols = xp[:, :, idx]                               # (B,Cin,Lout,Kw)
    return torch.einsum("bclk,ock->bol", cols, w)


def batchnorm(y, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5):
    """nn.BatchNorm1d (models.py:47,51).  Train: biased batch var for the output,
    unbiased for running_var (SURVEY.md §8c known-answer facts).  Returns
    (out, new_running_mean, new_running_var)."""
    if training:
        n = y.shape[0] * y.shape[2]
        mean = y.mean(dim=(0, 2))
        var = ((y - mean[None, :, None]) ** 2).mean(dim=(0, 2))
        new_rm = (1 - momentum) * running_mean + momentum * mean.detach()
        new_rv = (1 - momentum) * running_var + momentum * var.detach() * (n / max(n - 1, 1))
    else:
        mean, var = running_mean, running_var
        new_rm, new_rv = running_mean, running_var
    out = (y - mean[None, :, None]) / torch.sqrt(var[None, :, None] + eps)
    out = out * gamma[None, :, None] + beta[None, :, None]
    return out, new_rm, new_rv


def pool_windows(a):
    """(B,C,L) -> (B,C,Lout,3): the three candidates of every MaxPool1d(3, 2, 1) window, -inf where padded."""
    B, Cc, L = a.shape
    Lout = (L + 2 - 3) // 2 + 1
    ninf = torch.full((B, Cc, 1), -float("inf"), dtype=a.dtype)
    ap = torch.cat([ninf, a, ninf], dim=2)
    idx = torch.arange(Lout)[:, None] * 2 + torch.arange(3)[None, :]
    return ap[:, :, idx]


def first_argmax(win):
    """Index (0..2) of the FIRST maximal candidate of every window: MaxPool1d's tie rule."""
    m = win.max(dim=3, keepdim=True).values
    return (win == m).to(torch.int8).argmax(dim=3)


def relu_maxpool(z, choice=None):
    """ReLU then MaxPool1d(kernel 3, stride 2, pad 1 with -inf) (models.py:48-49,52-53).

    ``choice`` (B,C,Lout) in {0,1,2}, when given, replaces the argmax: the window's output is the chosen candidate.
    The parity tests use it for the one thing no two fp32 implementations can agree on — which of two candidates that
    are equal to within fp32 resolution is "the" maximum (tests/gpu_common.py: only such near-ties are ever overridden)."""
    win = pool_windows(torch.clamp_min(z, 0))
    if choice is None:
        return win.max(dim=3).values
    return win.gather(3, choice.to(torch.int64)[..., None]).squeeze(3)


def gru_cell(x_t, h, W_ih, W_hh, b_ih, b_hh):
    """One nn.GRU cell step, gate order r,z,n."""
    H = h.shape[1]
    gi = x_t @ W_ih.t() + b_ih
    gh = h @ W_hh.t() + b_hh
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * h


def gru_direction(xs, W_ih, W_hh, b_ih, b_hh, reverse):
    """One direction of one nn.GRU layer over (B,T',I) with h0 = 0 (models.py:78)."""
    B, TP, _ = xs.shape
    h = torch.zeros(B, W_hh.shape[1], dtype=xs.dtype)
    outs = [None] * TP
    order = range(TP - 1, -1, -1) if reverse else range(TP)
    for t in order:
        h = gru_cell(xs[:, t, :], h, W_ih, W_hh, b_ih, b_hh)
        outs[t] = h
    return torch.stack(outs, dim=1)


def cross_entropy(logits, labels):
    """nn.CrossEntropyLoss(reduction='mean') (trainer.py:69,147)."""
    lse = torch.logsumexp(logits, dim=1)
    picked = logits.gather(1, labels.view(-1, 1)).squeeze(1)
    return (lse - picked).mean()


def forward(params: Dict[str, torch.Tensor], buffers: Dict[str, torch.Tensor], x: torch.Tensor,
            *, training: bool, dropout_p: float = 0.0, seed: int = 0, step: int = 0,
            full_reverse_top: bool = False, pool_choice=None):
    """CnnGruAttentionModel.forward (models.py:73-81), returning every stage.

    ``full_reverse_top=True`` runs the top layer's reverse direction over all
    T' steps exactly as nn.GRU does; the default evaluates only the single
    reverse step that ``outputs[:, -1, :]`` consumes (models.py:79; SURVEY.md
    §2.2 K13).  Both give identical logits — that equality is itself tested.
    """
    st = OrderedDict()
    p = params
    B, C, T = x.shape
    H = p["gru.weight_hh_l0"].shape[1]                                    # gru_hidden_size (models.py:58)
    two_layers = "gru.weight_ih_l1" in p                                  # gru_num_layers 2 (reference default) or 1
    m, a1, s = channel_gate(x, p["channel_attention.fc.0.weight"], p["channel_attention.fc.2.weight"])
    st["gate_mean"], st["gate_pre"], st["gate_s"] = m, a1, s
    xs = x * s[:, :, None]                                               # models.py:31
    y1 = conv1d_strided(xs, p["cnn_encoder.0.weight"], 2, 3)             # models.py:46
    st["conv1"] = y1
    z1, rm1, rv1 = batchnorm(y1, p["cnn_encoder.1.weight"], p["cnn_encoder.1.bias"],
                             buffers["cnn_encoder.1.running_mean"], buffers["cnn_encoder.1.running_var"], training)
    st["bn1"] = z1
    p1 = relu_maxpool(z1, pool_choice and pool_choice.get("pool1"))       # models.py:48-49
    st["pool1"] = p1
    y2 = conv1d_strided(p1, p["cnn_encoder.4.weight"], 2, 2)             # models.py:50
    st["conv2"] = y2
    z2, rm2, rv2 = batchnorm(y2, p["cnn_encoder.5.weight"], p["cnn_encoder.5.bias"],
                             buffers["cnn_encoder.5.running_mean"], buffers["cnn_encoder.5.running_var"], training)
    st["bn2"] = z2
    p2 = relu_maxpool(z2, pool_choice and pool_choice.get("pool2"))       # models.py:52-53
    st["pool2"] = p2
    seq = p2.permute(0, 2, 1)                                            # models.py:77
    TP = seq.shape[1]

    def W(layer, sfx):
        return (p[f"gru.weight_ih_l{layer}{sfx}"], p[f"gru.weight_hh_l{layer}{sfx}"],
                p[f"gru.bias_ih_l{layer}{sfx}"], p[f"gru.bias_hh_l{layer}{sfx}"])

    h0f = gru_direction(seq, *W(0, ""), reverse=False)
    h0r = gru_direction(seq, *W(0, "_reverse"), reverse=True)
    l0 = torch.cat([h0f, h0r], dim=2)                                    # (B,T',128)
    st["gru_l0"] = l0
    thr = dropout_threshold(dropout_p) if training else 0
    if two_layers:
        if thr > 0:                                                      # nn.GRU(dropout=...) models.py:62
            keep = dropout_keep(dropout_key(seed, step, STREAM_GRU), l0.numel(), thr)
            mask = torch.from_numpy(keep.reshape(tuple(l0.shape))).to(l0.dtype) * dropout_scale(thr)
            l0d = l0 * mask
        else:
            l0d = l0
        st["gru_l0_dropped"] = l0d
        h1f = gru_direction(l0d, *W(1, ""), reverse=False)
        st["gru_l1_fwd"] = h1f
        if full_reverse_top:
            h1r_last = gru_direction(l0d, *W(1, "_reverse"), reverse=True)[:, -1, :]
        else:
            h1r_last = gru_cell(l0d[:, TP - 1, :], torch.zeros(B, H, dtype=x.dtype), *W(1, "_reverse"))
        st["gru_l1_rev_last"] = h1r_last
        feat = torch.cat([h1f[:, -1, :], h1r_last], dim=1)               # outputs[:, -1, :], models.py:79
    else:
        # one layer: nn.GRU applies its dropout between layers only, so none here; outputs[:, -1, :] is the forward direction's
        # final state and the reverse direction's FIRST step (models.py:78-79)
        st["gru_l0_dropped"] = l0
        feat = l0[:, TP - 1, :]
    st["feat"] = feat
    hid = torch.clamp_min(feat @ p["classifier.0.weight"].t() + p["classifier.0.bias"], 0)  # models.py:67-68
    if thr > 0:                                                          # nn.Dropout, models.py:69
        keep = dropout_keep(dropout_key(seed, step, STREAM_HEAD), hid.numel(), thr)
        hid = hid * (torch.from_numpy(keep.reshape(tuple(hid.shape))).to(hid.dtype) * dropout_scale(thr))
    st["cls_hidden"] = hid
    logits = hid @ p["classifier.3.weight"].t() + p["classifier.3.bias"]  # models.py:70
    st["logits"] = logits
    new_buffers = dict(buffers)
    if training:
        new_buffers["cnn_encoder.1.running_mean"] = rm1
        new_buffers["cnn_encoder.1.running_var"] = rv1
        new_buffers["cnn_encoder.5.running_mean"] = rm2
        new_buffers["cnn_encoder.5.running_var"] = rv2
        for k in ("cnn_encoder.1.num_batches_tracked", "cnn_encoder.5.num_batches_tracked"):
            new_buffers[k] = buffers[k] + 1
    return st, new_buffers


RETAINED = ("gate_s", "conv1", "bn1", "pool1", "conv2", "bn2", "pool2", "gru_l0_dropped", "feat", "logits")


def loss_and_grads(params, buffers, x, labels, retain=False, **fw):
    """Forward + CrossEntropy + backward (trainer.py:146-148).  Returns
    (loss, grads dict, stages, new_buffers); with retain=True the grads dict also
    holds "stage/<name>" gradients of the intermediate tensors in RETAINED."""
    leaf = {k: v.detach().clone().requires_grad_(v.numel() > 0) for k, v in params.items()}
    st, nb = forward(leaf, buffers, x, training=True, **fw)
    if retain:
        for k in RETAINED:
            if st[k].requires_grad:          # gate_s is a constant when C < 4
                st[k].retain_grad()
    loss = cross_entropy(st["logits"], labels)
    loss.backward()
    grads = {k: (v.grad.detach() if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
    if retain:
        for k in RETAINED:
            if st[k].requires_grad:
                grads["stage/" + k] = st[k].grad.detach()
    return loss.detach(), grads, st, nb


def adam_step(params, grads, exp_avg, exp_avg_sq, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
              weight_decay=1e-4):
    """torch.optim.Adam single step, L2 weight decay folded into the gradient
    (trainer.py:68,149; SURVEY.md §8c: p=1,g=0 -> 0.9990000725).  ``step`` is the
    1-based step count after the increment.  Updates dicts in place."""
    b1, b2 = betas
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    for k, pt in params.items():
        g = grads[k] + weight_decay * pt
        exp_avg[k].mul_(b1).add_(g, alpha=1 - b1)
        exp_avg_sq[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = exp_avg_sq[k].sqrt() / math.sqrt(bc2) + eps
        pt.addcdiv_(exp_avg[k], denom, value=-(lr / bc1))


def init_params(C: int, K: int, seed: int, dtype=torch.float32, hidden: int = GRU_HIDDEN, layers: int = 2):
    """Deterministic test weights (NOT the reference initialiser — tests only)."""
    g = torch.Generator().manual_seed(seed)
    out = OrderedDict()
    for name, shape in param_specs(C, K, hidden, layers).items():
        if name in ("cnn_encoder.1.weight", "cnn_encoder.5.weight"):
            t = 1.0 + 0.2 * torch.randn(shape, generator=g)
        elif len(shape) == 1:
            t = 0.1 * torch.randn(shape, generator=g)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
            t = torch.randn(shape, generator=g) / math.sqrt(max(fan_in, 1))
        out[name] = t.to(dtype)
    return out


def init_buffers(dtype=torch.float32):
    out = OrderedDict()
    for name, shape in buffer_specs().items():
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros((), dtype=torch.int64)
        elif name.endswith("running_var"):
            out[name] = torch.ones(shape, dtype=dtype)
        else:
            out[name] = torch.zeros(shape, dtype=dtype)
    return out

"""Drop-in for the reference's ``models.py`` on MI355X.

Same public surface (reference ``models.py:7-81``): ``ChannelAttention(in_channels,
reduction_ratio=4)`` and ``CnnGruAttentionModel(in_channels, num_classes,
cnn_out_channels=32, gru_hidden_size=64, gru_num_layers=2, dropout=0.5)`` are
``nn.Module``s with the reference's ``state_dict`` keys, shapes and default
initialisers (drawn in the same order, so the same ``torch.manual_seed`` gives the
same initial weights).  The sub-modules are parameter CONTAINERS only: all arithmetic
runs in libmsig_hip.so on the flat parameter buffer the parameters are views of.
There is no CPU fallback — a CPU input raises.
"""
from __future__ import annotations

import itertools
import math

import torch
import torch.nn as nn

from . import _lib as L
from .runtime import EmbeddedEngine, Engine

_instance_counter = itertools.count()


class ChannelAttention(nn.Module):
    """Container for the squeeze-excite gate's two bias-free Linear layers (models.py:12-22)."""

    def __init__(self, in_channels, reduction_ratio=4):
        super().__init__()
        hidden = in_channels // reduction_ratio
        self.avg_pool = nn.AdaptiveAvgPool1d(1)
        self.fc = nn.Sequential(nn.Linear(in_channels, hidden, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(hidden, in_channels, bias=False), nn.Sigmoid())

    def forward(self, x):
        """models.py:24-31 on its own: x * sigmoid(fc(mean_T(x)))[:, :, None] through msig_channel_attention (gate_kernel + one
        scaling pass).  Inside CnnGruAttentionModel the product is never written (the gate is folded into conv1's taps); this
        stand-alone forward is inference-only — gradients flow through the model's fused path, not through this call."""
        import ctypes as C
        if not x.is_cuda:
            raise RuntimeError("ChannelAttention.forward needs a GPU tensor: the MI355X path has no CPU fallback")
        if x.dtype != torch.float32 or x.dim() != 3 or x.shape[1] != self.fc[0].in_features:
            raise ValueError(f"expected float32 (B,{self.fc[0].in_features},T) input, got {x.dtype} {tuple(x.shape)}")
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise RuntimeError("the stand-alone ChannelAttention.forward is inference-only: call it under torch.no_grad(), "
                               "or train through CnnGruAttentionModel")
        x = x.contiguous()
        B, Cc, T = x.shape
        w1, w2 = self.fc[0].weight, self.fc[2].weight
        if w1.device != x.device:
            raise RuntimeError(f"ChannelAttention weights are on {w1.device}, input on {x.device}")
        out = torch.empty_like(x)
        s = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
        scratch = torch.empty(B * (Cc + Cc // 4) + 4, dtype=torch.float32, device=x.device)
        st = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        L.check(L.lib().msig_channel_attention(x.data_ptr(), w1.contiguous().data_ptr() if w1.numel() else None,
                                               w2.contiguous().data_ptr() if w2.numel() else None, B, Cc, T, out.data_ptr(),
                                               s.data_ptr(), scratch.data_ptr(), st), "msig_channel_attention")
        return out


class _GruParams(nn.Module):
    """nn.GRU's parameters (names, shapes, U(-1/sqrt(H), 1/sqrt(H)) init in nn.GRU's order)
    without nn.GRU itself, so no MIOpen weight flattening ever touches them."""

    def __init__(self, input_size, hidden_size, num_layers):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers, self.bidirectional = input_size, hidden_size, num_layers, True
        for layer in range(num_layers):
            isz = input_size if layer == 0 else 2 * hidden_size
            for sfx in ("", "_reverse"):
                self.register_parameter(f"weight_ih_l{layer}{sfx}", nn.Parameter(torch.empty(3 * hidden_size, isz)))
                self.register_parameter(f"weight_hh_l{layer}{sfx}", nn.Parameter(torch.empty(3 * hidden_size, hidden_size)))
                self.register_parameter(f"bias_ih_l{layer}{sfx}", nn.Parameter(torch.empty(3 * hidden_size)))
                self.register_parameter(f"bias_hh_l{layer}{sfx}", nn.Parameter(torch.empty(3 * hidden_size)))
        stdv = 1.0 / math.sqrt(hidden_size)
        for w in self.parameters():
            nn.init.uniform_(w, -stdv, stdv)

    def forward(self, *a, **k):
        raise RuntimeError("the GRU runs inside libmsig_hip.so; call CnnGruAttentionModel")


class _MsigFunction(torch.autograd.Function):
    """model(inputs) with autograd: forward = msig_forward, backward = msig_backward."""

    @staticmethod
    def forward(ctx, model, x, *params):
        eng = model._engine
        training = model.training
        step = model._bump_step() if training else 0
        b = eng.forward(x, None, training=training, dropout_p=model.dropout_p, seed=model._seed, step=step)
        ctx.model, ctx.batch, ctx.token, ctx.training = model, b, model._bump_token(), training
        return eng.region("LOGITS", torch.float32, (x.shape[0], model.num_classes)).clone()

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        if not ctx.training:
            raise RuntimeError("backward through a model.eval() forward is not supported (no stash is kept)")
        if ctx.token != model._token:
            raise RuntimeError("the activations of this forward were overwritten by a later forward of the same model")
        eng = model._engine
        eng.backward(ctx.batch, dlogits)
        if model.embedded:
            grads = [g.clone() for g in eng.gather_grads().values()]
        else:
            grads = [eng.param_view(i, eng.grads).clone() for i in range(L.NPARAM)]
        return (None, None, *grads)


class CnnGruAttentionModel(nn.Module):
    def __init__(self, in_channels, num_classes, cnn_out_channels=32, gru_hidden_size=64, gru_num_layers=2, dropout=0.5):
        super().__init__()
        if (cnn_out_channels, gru_hidden_size, gru_num_layers) not in ((32, 64, 2), (32, 32, 1)):
            raise NotImplementedError(
                "the HIP path covers the reference's two configurations: cnn_out_channels=32 with gru_hidden_size=64, "
                "gru_num_layers=2 (main.py:48-55) or gru_hidden_size=32, gru_num_layers=1 (the hierarchical experiment's second "
                f"model, main.py:35-40); got {(cnn_out_channels, gru_hidden_size, gru_num_layers)}")
        # the 32-unit one-layer model runs embedded in the 64-unit kernels (runtime.EmbeddedEngine): same state_dict as the reference's
        self.embedded = (gru_hidden_size, gru_num_layers) == (32, 1)
        self.gru_hidden_size, self.gru_num_layers = gru_hidden_size, gru_num_layers
        if not (1 <= in_channels <= L.MAX_C and 2 <= num_classes <= L.MAX_K):
            raise ValueError(f"in_channels must be 1..{L.MAX_C} and num_classes 2..{L.MAX_K}")
        self.in_channels, self.num_classes, self.dropout_p = in_channels, num_classes, float(dropout)
        # containers, created in the reference's order (models.py:43-71) so that the RNG stream matches
        self.channel_attention = ChannelAttention(in_channels)
        self.cnn_encoder = nn.Sequential(
            nn.Conv1d(in_channels, 16, kernel_size=7, stride=2, padding=3, bias=False), nn.BatchNorm1d(16), nn.ReLU(),
            nn.MaxPool1d(kernel_size=3, stride=2, padding=1),
            nn.Conv1d(16, cnn_out_channels, kernel_size=5, stride=2, padding=2, bias=False), nn.BatchNorm1d(cnn_out_channels),
            nn.ReLU(), nn.MaxPool1d(kernel_size=3, stride=2, padding=1))
        self.gru = _GruParams(cnn_out_channels, gru_hidden_size, gru_num_layers)
        self.classifier = nn.Sequential(nn.Linear(2 * gru_hidden_size, 64), nn.ReLU(), nn.Dropout(dropout), nn.Linear(64, num_classes))
        self._engine = None
        self._seed = (torch.initial_seed() * 0x9E3779B97F4A7C15 + next(_instance_counter)) % (1 << 64)
        self._step = 0
        self._token = 0

    # ---- binding of nn.Parameters / buffers to the engine's flat buffers ----------------------
    def _named(self):
        sd_params = dict(self.named_parameters())
        return [sd_params[k] for k in L.PARAM_KEYS if k in sd_params]        # the one-layer model has no *_l1* tensors

    def _grad_views(self):
        """Where the optimiser puts p.grad (same order as _named()) for the un-fused path (MsigAdam.step)."""
        eng = self._engine
        if self.embedded:
            return list(eng.small_views(eng.small_grads).values())
        return [eng.param_view(i, eng.grads) for i in range(L.NPARAM)]

    def engine(self) -> Engine:
        """Returns the Engine whose flat buffers the parameters are views of (re-binding after
        .to(), load_state_dict(assign=True) or anything else that replaced a parameter's storage)."""
        plist = self._named()
        dev = self.classifier[0].weight.device
        if dev.type != "cuda":
            raise RuntimeError(f"model is on {dev}: move it to the GPU (model.to('cuda')); there is no CPU fallback")
        if self._engine is None or self._engine.device != dev:
            self._engine = (EmbeddedEngine(self.in_channels, self.num_classes, dev, self.gru_hidden_size) if self.embedded
                            else Engine(self.in_channels, self.num_classes, dev))
        eng = self._engine
        views = list(eng.small_views().values()) if self.embedded else [eng.param_view(i) for i in range(len(plist))]
        for p, view in zip(plist, views):
            if p.numel() and (p.data_ptr() != view.data_ptr() or p.device != dev):
                view.copy_(p.data)
                p.data = view
        bv = eng.bn_views()
        for idx in (1, 5):
            bn = self.cnn_encoder[idx]
            for name in ("running_mean", "running_var", "num_batches_tracked"):
                cur, view = bn._buffers[name], bv[f"cnn_encoder.{idx}.{name}"]
                if cur.data_ptr() != view.data_ptr() or cur.device != dev:
                    view.copy_(cur)
                    bn._buffers[name] = view
        return eng

    def set_dropout_seed(self, seed: int):
        """Dropout masks are a pure function of (seed, step, element): fix the seed for reproducible runs
        (by default it derives from torch.initial_seed() and the order of construction)."""
        self._seed = int(seed) % (1 << 64)
        self._step = 0

    def _bump_step(self):
        self._step += 1
        return self._step

    def _bump_token(self):
        self._token += 1
        return self._token

    def forward(self, x):
        if isinstance(x, (list, tuple)):
            raise TypeError("this model takes one (B, C, T) tensor (trainer.py:135-140's list branch is for a dataset "
                            "the reference no longer ships)")
        if not x.is_cuda:
            raise RuntimeError("CnnGruAttentionModel.forward needs a GPU tensor: the MI355X path has no CPU fallback")
        self.engine()
        return _MsigFunction.apply(self, x, *self._named())

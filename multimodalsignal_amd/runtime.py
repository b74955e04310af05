"""Device-side state of one model instance and thin wrappers over the C ABI.

`Engine` owns (through torch tensors) the flat parameter / gradient / Adam
buffers, the BatchNorm buffers and a cache of workspaces, builds the
`msig_batch` descriptor for every call and launches on torch's current HIP
stream.  It holds no arithmetic of its own.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from . import _lib as L


def _require_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"{what} is on {t.device}: the multimodalsignal_amd path runs only on an AMD GPU through "
            "libmsig_hip.so (there is no CPU fallback; the CPU restatement lives in oracle/ and is test-only).")


class Engine:
    def __init__(self, in_channels: int, num_classes: int, device: torch.device, storage: Optional[dict] = None):
        """`storage` (FoldArena.engine): pre-allocated flat tensors "params", "grads", "exp_avg", "exp_avg_sq", "bn_state",
        "bn_count" and a "ws" byte region to use instead of allocating — the buffers of one arena of a fold batch."""
        if not (1 <= in_channels <= L.MAX_C) or not (2 <= num_classes <= L.MAX_K):
            raise ValueError(f"unsupported in_channels={in_channels} / num_classes={num_classes}")
        self.C, self.K = in_channels, num_classes
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("Engine needs a cuda (HIP) device")
        L.lib()
        self.layout = L.param_layout(self.C, self.K)
        self.shapes = L.param_shapes(self.C, self.K)
        self.n_flat = self.layout[-1]
        self._ws_region = None
        if storage is None:
            self.params = torch.zeros(self.n_flat, dtype=torch.float32, device=self.device)
            self.grads = torch.zeros(self.n_flat, dtype=torch.float32, device=self.device)
            self.bn_state = torch.zeros(L.BN_STATE_FLOATS, dtype=torch.float32, device=self.device)
            self.bn_count = torch.zeros(2, dtype=torch.int64, device=self.device)
            self.exp_avg: Optional[torch.Tensor] = None
            self.exp_avg_sq: Optional[torch.Tensor] = None
            self.loss_acc = torch.zeros(2, dtype=torch.float64, device=self.device)      # msig_batch.loss_acc: [sum of CE, #correct] of a pass
        else:
            self.params, self.grads = storage["params"], storage["grads"]
            self.bn_state, self.bn_count = storage["bn_state"], storage["bn_count"]
            self.exp_avg, self.exp_avg_sq = storage["exp_avg"], storage["exp_avg_sq"]
            self._ws_region = storage["ws"]
            self.loss_acc = storage["acc"]
            for t in (self.params, self.grads, self.exp_avg, self.exp_avg_sq, self.bn_state, self.bn_count, self.loss_acc):
                t.zero_()
        self.bn_state[16:32] = 1.0
        self.bn_state[64:96] = 1.0
        self.gru_layers = 2                     # msig_batch.gru_layers: EmbeddedEngine (the one-layer, 32-unit model) sets 1
        self._ws: Dict[Tuple[int, int, bool], Tuple[torch.Tensor, list]] = {}
        self._ws_pool: Dict[bool, torch.Tensor] = {}
        self._last: Optional[Tuple[int, int, bool]] = None
        self._keep = None

    # ---- views -----------------------------------------------------------------
    def _numel(self, i):
        n = 1
        for s in self.shapes[i]:
            n *= s
        return n

    def param_view(self, i: int, flat: Optional[torch.Tensor] = None) -> torch.Tensor:
        flat = self.params if flat is None else flat
        o = self.layout[i]
        return flat[o:o + self._numel(i)].view(self.shapes[i])

    def named_param_views(self, flat: Optional[torch.Tensor] = None):
        return {k: self.param_view(i, flat) for i, k in enumerate(L.PARAM_KEYS)}

    def bn_views(self):
        s = self.bn_state
        return {"cnn_encoder.1.running_mean": s[0:16], "cnn_encoder.1.running_var": s[16:32],
                "cnn_encoder.5.running_mean": s[32:64], "cnn_encoder.5.running_var": s[64:96],
                "cnn_encoder.1.num_batches_tracked": self.bn_count[0], "cnn_encoder.5.num_batches_tracked": self.bn_count[1]}

    def load_named(self, named: Dict[str, torch.Tensor]):
        """Copies a reference-style state_dict (any subset) into the flat buffers."""
        pv, bv = self.named_param_views(), self.bn_views()
        for k, v in named.items():
            dst = pv.get(k, bv.get(k))
            if dst is None:
                raise KeyError(k)
            dst.copy_(torch.as_tensor(v).to(dst.dtype).reshape(dst.shape))

    # ---- workspace ---------------------------------------------------------------
    def workspace(self, B: int, T: int, training: bool):
        key = (B, T, bool(training))
        if key not in self._ws:
            off = L.workspace_layout(B, self.C, T, self.K, training)
            if self._ws_region is not None:          # arena mode: every shape shares the arena's one workspace region
                if off[-1] > self._ws_region.numel():
                    raise RuntimeError(f"fold arena workspace too small for B={B}, T={T}")
                buf = self._ws_region[:off[-1]]
            else:
                # One allocation per mode, sized for the largest shape seen: the ragged last batch of an epoch lays its regions
                # out in the full batch's buffer instead of allocating a second one (13.6 GB each at B = 8192).  Training and
                # evaluation keep separate buffers — an evaluation between a training forward and its backward (autograd path)
                # must not overwrite the stash.  Growing frees the smaller buffer: layouts cached for it are dropped, and a
                # HIP graph captured on it must be re-captured (a graph holds raw pointers).
                pool = self._ws_pool.get(bool(training))
                if pool is None or pool.numel() < off[-1]:
                    for k in [k for k in self._ws if k[2] == bool(training)]:
                        del self._ws[k]
                    if self._last is not None and self._last[2] == bool(training):
                        self._last, self._keep = None, None      # region() must not resolve to a purged layout
                    pool = None
                    self._ws_pool[bool(training)] = pool = torch.empty(off[-1], dtype=torch.uint8, device=self.device)
                buf = pool[:off[-1]]
            lo = off[L.WS["LOSS"]]
            buf[lo:lo + 16].zero_()
            self._ws[key] = (buf, off)
        return self._ws[key]

    def drop_workspaces(self):
        self._ws.clear()
        self._ws_pool.clear()
        self._last = None

    def region(self, name: str, dtype=torch.float32, shape=None, key=None) -> torch.Tensor:
        """A typed view of a workspace region of the last (or given) call."""
        key = key or self._last
        buf, off = self._ws[key]
        i = L.WS[name]
        raw = buf[off[i]:off[i + 1]].view(dtype)
        if shape is not None:
            n = 1
            for s in shape:
                n *= s
            raw = raw[:n].view(*shape)
        return raw

    # ---- descriptor ----------------------------------------------------------------
    def _batch(self, x: torch.Tensor, labels: Optional[torch.Tensor], training: bool, dropout_p: float,
               seed: int, step: int) -> L.Batch:
        _require_gpu(x, "input batch")
        if x.dtype != torch.float32 or x.dim() != 3 or x.shape[1] != self.C:
            raise ValueError(f"expected float32 (B,{self.C},T) input, got {x.dtype} {tuple(x.shape)}")
        x = x.contiguous()
        B, _, T = x.shape
        if labels is not None:
            _require_gpu(labels, "labels")
            labels = labels.to(torch.int64).contiguous()
        buf, off = self.workspace(B, T, training)
        thr = L.dropout_threshold(dropout_p) if training else 0
        b = L.Batch()
        b.shape = L.Shape(B, self.C, T, self.K)
        b.training = int(training)
        b.bn_momentum, b.bn_eps = 0.1, 1e-5
        b.dropout_thr = thr
        b.key_gru = L.dropout_key(seed, step, 1) if thr else 0
        b.key_head = L.dropout_key(seed, step, 2) if thr else 0
        b.x = x.data_ptr()
        b.labels = labels.data_ptr() if labels is not None else None
        b.params = self.params.data_ptr()
        b.grads = self.grads.data_ptr()
        b.bn_state = self.bn_state.data_ptr()
        b.bn_count = self.bn_count.data_ptr()
        b.ws = buf.data_ptr()
        b.ws_bytes = buf.numel()
        b.gru_layers = self.gru_layers
        b.loss_acc = self.loss_acc.data_ptr()
        L.apply_forms(b)
        self._last = (B, T, bool(training))
        self._keep = (x, labels)
        return b

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- calls -------------------------------------------------------------------------
    def forward(self, x, labels=None, training=False, dropout_p=0.0, seed=0, step=0) -> L.Batch:
        """model(inputs) [+ criterion]: logits land in region('LOGITS'); returns the descriptor
        that a following backward() must be given."""
        b = self._batch(x, labels, training, dropout_p, seed, step)
        L.check(L.lib().msig_forward(C.byref(b), self._stream()), "msig_forward")
        return b

    def backward(self, b: L.Batch, dlogits: Optional[torch.Tensor] = None):
        ptr = None
        if dlogits is not None:
            _require_gpu(dlogits, "dlogits")
            dlogits = dlogits.to(torch.float32).contiguous()
            ptr = dlogits.data_ptr()
        L.check(L.lib().msig_backward(C.byref(b), ptr, self._stream()), "msig_backward")

    def ensure_adam_state(self):
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.params)
            self.exp_avg_sq = torch.zeros_like(self.params)

    def adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, step=1):
        self.ensure_adam_state()
        L.check(L.lib().msig_adam_step(self.params.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(),
                                       self.exp_avg_sq.data_ptr(), self.n_flat, lr, betas[0], betas[1], eps,
                                       weight_decay, step, self._stream()), "msig_adam_step")

    def train_step(self, x, labels, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, step=1,
                   dropout_p=0.0, seed=0) -> None:
        """optimizer.zero_grad(); loss = criterion(model(x), y); loss.backward(); optimizer.step()
        (trainer.py:144-149) as one asynchronous call; the batch loss is left in region('LOSS')[0] and added, times the batch size,
        to loss_acc[0] (loss_acc[1] += correctly classified windows): the caller zeroes loss_acc when an epoch starts."""
        self.ensure_adam_state()
        b = self._batch(x, labels, True, dropout_p, seed, step)
        L.check(L.lib().msig_train_step(C.byref(b), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), lr,
                                        betas[0], betas[1], eps, weight_decay, step, self._stream()), "msig_train_step")

    def stage(self, name: str, b: L.Batch):
        """Runs a single stage launcher by name (tests / profiling)."""
        fn = getattr(L.lib(), f"msig_{name}")
        if name == "head_ce_bwd":
            L.check(fn(C.byref(b), None, self._stream()), name)
        else:
            L.check(fn(C.byref(b), self._stream()), name)


class EmbeddedEngine(Engine):
    """The hierarchical experiment's second model (main.py:35-40: gru_hidden_size = 32, gru_num_layers = 1) on the kernels of the
    reference configuration.  Its GRU is embedded in layer 0 of the 64-unit layout: unit u of gate g sits at row g * 64 + u of the
    padded weight / bias tensors, every other row and column is zero.  A padded unit then has r = z = 1/2, n = 0, so its state stays
    exactly 0 from h_0 = 0, it feeds nothing into the real units (its W_hh columns multiply zeros), and its own weights receive exactly
    zero gradient (their gate gradients and its state are zero) — Adam with L2 decay keeps them at zero.  Layer 1 is skipped by the
    library (msig_batch.gru_layers = 1): outputs[:, -1, :] is layer 0's output at the last position, whose 64 + 64 columns hold the
    2 x 32 real features at columns 0..31 and 64..95 (classifier.0.weight is embedded the same way).

    The model's nn.Parameters are views of ONE contiguous buffer `small` (reference shapes, reference order); `index` maps its
    elements into the padded flat buffer.  scatter() / gather() move values between the two around every library call."""

    def __init__(self, in_channels: int, num_classes: int, device: torch.device, hidden: int):
        super().__init__(in_channels, num_classes, device)
        if hidden != 32:
            raise NotImplementedError("embedded GRU: hidden size 32 (main.py:38)")
        self.gru_layers = 1
        self.hidden = hidden
        H, dev = hidden, self.device
        self.small_shapes, idx = [], []
        keys = L.PARAM_KEYS
        gate_rows = torch.cat([torch.arange(g * 64, g * 64 + H) for g in range(3)])            # rows of the real units
        for i, k in enumerate(keys):
            o, shape = self.layout[i], self.shapes[i]
            if k.startswith("gru."):
                if "_l1" in k:
                    continue                                            # no second layer in the embedded model
                if k.startswith("gru.weight_ih"):
                    sm = (3 * H, shape[1]); ii = o + gate_rows[:, None] * shape[1] + torch.arange(shape[1])[None, :]
                elif k.startswith("gru.weight_hh"):
                    sm = (3 * H, H); ii = o + gate_rows[:, None] * shape[1] + torch.arange(H)[None, :]
                else:
                    sm = (3 * H,); ii = o + gate_rows
            elif k == "classifier.0.weight":                            # (64, 128): real features at columns 0..31 (forward) and 64..95 (reverse)
                cols = torch.cat([torch.arange(0, H), torch.arange(64, 64 + H)])
                sm = (shape[0], 2 * H); ii = o + torch.arange(shape[0])[:, None] * shape[1] + cols[None, :]
            else:
                sm, ii = tuple(shape), o + torch.arange(self._numel(i))
            self.small_shapes.append((k, sm))
            idx.append(ii.reshape(-1).to(torch.int64))
        self.index = torch.cat(idx).to(dev)
        self.small = torch.zeros(int(self.index.numel()), dtype=torch.float32, device=dev)
        self.small_grads = torch.zeros_like(self.small)

    def small_views(self, flat: Optional[torch.Tensor] = None):
        flat = self.small if flat is None else flat
        out, at = {}, 0
        for k, sm in self.small_shapes:
            n = 1
            for d in sm:
                n *= d
            out[k] = flat[at:at + n].view(sm)
            at += n
        return out

    def scatter(self):
        """model parameters -> padded flat buffer (the padding stays zero)."""
        self.params.zero_()
        self.params.index_copy_(0, self.index, self.small)

    def gather(self):
        self.small.copy_(self.params.index_select(0, self.index))

    def gather_grads(self):
        self.small_grads.copy_(self.grads.index_select(0, self.index))
        return self.small_views(self.small_grads)

    # ---- the library calls, with the embedding maintained around them ----
    def forward(self, x, labels=None, training=False, dropout_p=0.0, seed=0, step=0):
        self.scatter()
        return super().forward(x, labels, training, dropout_p, seed, step)

    def train_step(self, x, labels, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, step=1, dropout_p=0.0, seed=0):
        self.scatter()
        super().train_step(x, labels, lr, betas, eps, weight_decay, step, dropout_p, seed)
        self.gather()

    def adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, step=1):
        """Un-fused use: the caller has put the model's gradients into `small_grads` (MsigAdam.step)."""
        self.scatter()
        self.grads.zero_()
        self.grads.index_copy_(0, self.index, self.small_grads)
        super().adam_step(lr, betas, eps, weight_decay, step)
        self.gather()


class FoldArena:
    """Device memory of a fold batch (include/msig.h msig_multi): `n` identical arenas, `stride` bytes apart, each holding one
    model's parameters, gradients, Adam moments, BatchNorm state, one workspace region, one input batch and its labels at the
    same offsets — which is all msig_*_multi needs to run the same step for several folds in one set of launches."""

    def __init__(self, in_channels: int, num_classes: int, device, n: int, train_batch: int, T: int, eval_batch: int = 0,
                 adaptive_forms: bool = False):
        if not (1 <= n <= L.MAX_FOLDS):
            raise ValueError(f"1..{L.MAX_FOLDS} folds per arena set")
        eval_batch = int(eval_batch) or int(train_batch)
        self.C, self.K, self.n, self.T = in_channels, num_classes, n, T
        self.max_batch = max(int(train_batch), eval_batch)
        self.max_train_batch = int(train_batch)
        self.adaptive_forms = bool(adaptive_forms)
        self.device = torch.device(device)
        self.n_flat = L.param_layout(in_channels, num_classes)[-1]
        self.ws_bytes = self.workspace_bytes(train_batch, eval_batch, in_channels, T, num_classes)
        sizes = [("params", self.n_flat * 4), ("grads", self.n_flat * 4), ("exp_avg", self.n_flat * 4), ("exp_avg_sq", self.n_flat * 4),
                 ("bn_state", L.BN_STATE_FLOATS * 4), ("bn_count", 16), ("acc", 16), ("x", self.max_batch * in_channels * T * 4), ("y", self.max_batch * 8),
                 ("ws", self.ws_bytes)]
        self.off, at = {}, 0
        for name, nbytes in sizes:
            self.off[name] = (at, nbytes)
            at += (nbytes + 255) // 256 * 256
        self.stride = at
        self.mem = torch.zeros((n, self.stride), dtype=torch.uint8, device=self.device)

    @staticmethod
    def workspace_bytes(train_batch: int, eval_batch: int, in_channels: int, T: int, num_classes: int) -> int:
        """Bytes of an arena's workspace region (allocates nothing): it serves training steps of at most `train_batch` windows and
        evaluation passes of at most `eval_batch` (the evaluation layout has no stash and no gradient scratch: a large
        --eval-batch-size must not be priced as a training batch).  msig_workspace_layout is NOT monotonic in B — the projection region
        WS_GI exists only below 192 batch tiles, so a ragged last batch just under 3072 windows needs MORE than the full batch above
        it: the maximum over every batch size that can occur."""
        def need(bs, training):
            cand = [int(bs)] + ([191 * 16] if bs >= 192 * 16 else [])
            return max(L.workspace_layout(c, in_channels, T, num_classes, training)[-1] for c in cand)
        return max(need(train_batch, True), need(int(eval_batch) or int(train_batch), False))

    def view(self, slot: int, name: str, dtype=torch.uint8) -> torch.Tensor:
        o, nb = self.off[name]
        return self.mem[slot, o:o + nb].view(dtype)

    def across(self, name: str, byte_offset: int, dtype, count: int = 1) -> torch.Tensor:
        """(n, count) strided view of `count` elements at `byte_offset` inside region `name` of every arena."""
        o, _ = self.off[name]
        esz = torch.empty(0, dtype=dtype).element_size()
        flat = self.mem.view(dtype)                                    # (n, stride / esz)
        start = (o + byte_offset) // esz
        return flat[:, start:start + count]

    def engine(self, slot: int) -> Engine:
        st = {k: self.view(slot, k, torch.float32) for k in ("params", "grads", "exp_avg", "exp_avg_sq", "bn_state")}
        st["bn_count"] = self.view(slot, "bn_count", torch.int64)
        st["acc"] = self.view(slot, "acc", torch.float64)
        st["ws"] = self.view(slot, "ws")
        return Engine(self.C, self.K, self.device, storage=st)

    def ptr(self, name: str) -> int:
        return self.mem.data_ptr() + self.off[name][0]                  # arena 0's buffer

    def batch(self, B: int, training: bool, dropout_p: float, with_labels: bool = True) -> L.Batch:
        """msig_batch describing arena 0 (the *_multi calls shift every pointer by slot * stride)."""
        if B > (self.max_train_batch if training else self.max_batch):
            raise ValueError(f"batch {B} exceeds the arena's {self.max_train_batch if training else self.max_batch}")
        b = L.Batch()
        b.shape = L.Shape(B, self.C, self.T, self.K)
        b.training = int(training)
        b.bn_momentum, b.bn_eps = 0.1, 1e-5
        b.dropout_thr = L.dropout_threshold(dropout_p) if training else 0
        b.key_gru = b.key_head = 0
        b.x, b.labels = self.ptr("x"), (self.ptr("y") if with_labels else None)
        b.params, b.grads = self.ptr("params"), self.ptr("grads")
        b.bn_state, b.bn_count = self.ptr("bn_state"), self.ptr("bn_count")
        b.ws, b.ws_bytes = self.ptr("ws"), self.ws_bytes
        b.loss_acc = self.ptr("acc")
        b.gru_layers = 2
        L.apply_forms(b)
        return b

    def multi(self, slots, key_gru=None, key_head=None, lr=None, steps=None) -> L.Multi:
        m = L.Multi()
        m.n, m.stride_bytes = len(slots), self.stride
        # The BACKWARD GRU kernel form is chosen as for ONE stand-alone fold of this batch size, whatever the number of folds in the
        # launch: a fold's bits must not depend on which companions share its launches, on when they stop early, on --lockstep-groups
        # or on how many ranks the folds are dealt to, and the backward forms round differently (they group the dW partials
        # differently).  adaptive_forms=True lets it follow the folds still active in each launch instead (msig.h: fused backward
        # kernels from 12 tiles per launch on).  The FORWARD forms are bit-identical since round 5: the library picks per layer and
        # per launch (gru.hip fwd_form) and this pin does not enter.
        m.form_folds = 0 if self.adaptive_forms else 1
        for i, s in enumerate(slots):
            m.slot[i] = int(s)
            m.key_gru[i] = int(key_gru[i]) if key_gru is not None else 0
            m.key_head[i] = int(key_head[i]) if key_head is not None else 0
            m.lr[i] = float(lr[i]) if lr is not None else 0.0
            m.step[i] = int(steps[i]) if steps is not None else 0
        return m

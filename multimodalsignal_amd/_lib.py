"""ctypes binding of libmsig_hip.so (include/msig.h).

There is deliberately no fallback: if the shared library is missing the import
of anything that computes raises, and every launcher raises RuntimeError on a
non-zero status.  PyTorch is used only to own device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("MSIG_LIB", _HERE / "libmsig_hip.so"))

# ---- mirrors of the enums in include/msig.h (tests/test_cabi.py checks them against the header)
P_GATE_W1, P_GATE_W2, P_CONV1_W, P_BN1_G, P_BN1_B, P_CONV2_W, P_BN2_G, P_BN2_B, P_GRU = range(9)
P_CLS0_W, P_CLS0_B, P_CLS3_W, P_CLS3_B, NPARAM = P_GRU + 16, P_GRU + 17, P_GRU + 18, P_GRU + 19, P_GRU + 20

WS_NAMES = [
    "GATE_MEAN", "GATE_PRE", "GATE_S", "Y1", "BN1_PART", "BN1_STAT", "P1", "Y2", "BN2_PART", "BN2_STAT", "P2",
    "H0", "H1", "STASH0", "STASH1", "STASH1R", "FEAT", "HID", "LOGITS", "PROBS", "PRED", "LOSS", "DLOGITS",
    "DFEAT", "DH0", "DX0", "DY2", "DP1", "DS", "BNB_PART", "BNB_STAT", "GRAD_PART", "GI", "POOLC1", "POOLC2", "G1W", "GATE_EO",
]
WS = {n: i for i, n in enumerate(WS_NAMES)}
NWS = len(WS_NAMES)
BN_STATE_FLOATS = 96
MAX_C, MAX_K = 16, 16

# state_dict key of every parameter tensor, in msig_param order (== nn.Module.parameters() order)
PARAM_KEYS = (
    ["channel_attention.fc.0.weight", "channel_attention.fc.2.weight", "cnn_encoder.0.weight",
     "cnn_encoder.1.weight", "cnn_encoder.1.bias", "cnn_encoder.4.weight", "cnn_encoder.5.weight",
     "cnn_encoder.5.bias"]
    + [f"gru.{w}_l{layer}{sfx}" for layer in (0, 1) for sfx in ("", "_reverse")
       for w in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    + ["classifier.0.weight", "classifier.0.bias", "classifier.3.weight", "classifier.3.bias"]
)
assert len(PARAM_KEYS) == NPARAM

ERRORS = {-1: "MSIG_E_NULL (required pointer is NULL)", -2: "MSIG_E_SHAPE (unsupported B/C/T/K or mode)",
          -3: "MSIG_E_ALIGN (buffer not 16-byte aligned)", -4: "MSIG_E_WORKSPACE (workspace too small)",
          -5: "MSIG_E_FORM (fwd_form / bwd_form is not a kernel form this call can run)"}


class Shape(C.Structure):
    _fields_ = [("B", C.c_int32), ("C", C.c_int32), ("T", C.c_int32), ("K", C.c_int32)]


class Batch(C.Structure):
    _fields_ = [
        ("shape", Shape), ("training", C.c_int32), ("bn_momentum", C.c_float), ("bn_eps", C.c_float),
        ("dropout_thr", C.c_int32), ("key_gru", C.c_uint32), ("key_head", C.c_uint32),
        ("x", C.c_void_p), ("labels", C.c_void_p), ("params", C.c_void_p), ("grads", C.c_void_p),
        ("bn_state", C.c_void_p), ("bn_count", C.c_void_p), ("ws", C.c_void_p), ("ws_bytes", C.c_int64),
        ("gru_layers", C.c_int32), ("fwd_form", C.c_int16), ("bwd_form", C.c_int16),
        ("loss_acc", C.c_void_p),
    ]


MAX_FOLDS = 16
ABI_VERSION = 4       # include/msig.h MSIG_ABI_VERSION


class Multi(C.Structure):
    """msig_multi (include/msig.h): a fold batch — arenas `stride_bytes` apart, per-fold dropout keys, learning rates and optimiser step counts."""
    _fields_ = [("n", C.c_int32), ("slot", C.c_int32 * MAX_FOLDS), ("stride_bytes", C.c_int64),
                ("key_gru", C.c_uint32 * MAX_FOLDS), ("key_head", C.c_uint32 * MAX_FOLDS), ("lr", C.c_float * MAX_FOLDS),
                ("form_folds", C.c_int32), ("step", C.c_int64 * MAX_FOLDS)]


_lib = None


def _loaded_hip_runtime() -> str:
    """Path of the libamdhip64 this process has ALREADY mapped (torch's bundled copy once torch is imported), else the soname for
    the loader's search path.  MSIG_HIP_RUNTIME overrides.  dlopen of that path returns the mapped object; RTLD_GLOBAL then makes
    its symbols visible to libmsig_hip.so, which names no HIP runtime of its own."""
    override = os.environ.get("MSIG_HIP_RUNTIME")
    if override:
        return override
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(None, 1)[-1]
                if "libamdhip64" in os.path.basename(path):
                    return path
    except OSError:
        pass
    return "libamdhip64.so.7"


def lib() -> C.CDLL:
    """Loads libmsig_hip.so once.  Raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `make -C multimodalsignal_amd/csrc` "
                "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
        # libmsig_hip.so has no DT_NEEDED on the HIP runtime (csrc/Makefile: -no-hip-rt): it binds to the copy this process already
        # uses.  torch first — its wheel bundles its own libamdhip64 — then that very object, found BY PATH in /proc/self/maps
        # (_loaded_hip_runtime: dlopen of a mapped path returns the mapped object), is promoted to the global symbol scope, where
        # the loader resolves this library's hip* symbols.  Not by soname: a soname lookup walks the search path and may return
        # /opt/rocm's copy — a second runtime in the process, round 3's hipErrorNoDevice (build() + smoke() in one process).  Only
        # a process with no runtime mapped at all falls back to the soname (ld.so.conf has /opt/rocm/lib on this image).
        import torch  # noqa: F401
        rt = _loaded_hip_runtime()
        try:
            C.CDLL(rt, mode=C.RTLD_GLOBAL)
        except OSError as e:
            raise RuntimeError(f"cannot open the HIP runtime {rt!r} ({e}): libmsig_hip.so binds to the libamdhip64 the process already uses; "
                               "set MSIG_HIP_RUNTIME=/path/to/libamdhip64.so to name it explicitly") from e
        L = C.CDLL(str(LIB_PATH))
        vp, i64p = C.c_void_p, C.POINTER(C.c_int64)
        L.msig_abi_version.restype = C.c_int
        L.msig_struct_bytes.argtypes = [C.c_int32]
        L.msig_struct_bytes.restype = C.c_int64
        if L.msig_abi_version() != ABI_VERSION or L.msig_struct_bytes(0) != C.sizeof(Batch) or L.msig_struct_bytes(1) != C.sizeof(Multi):
            raise RuntimeError(f"{LIB_PATH} is ABI {L.msig_abi_version()} with msig_batch / msig_multi of {L.msig_struct_bytes(0)} / "
                               f"{L.msig_struct_bytes(1)} bytes; this binding is ABI {ABI_VERSION} with {C.sizeof(Batch)} / {C.sizeof(Multi)}: "
                               "rebuild the library (make -C multimodalsignal_amd/csrc)")
        L.msig_stage_lengths.argtypes = [C.c_int, C.POINTER(C.c_int32)]
        L.msig_param_layout.argtypes = [C.c_int, C.c_int, i64p]
        L.msig_workspace_layout.argtypes = [C.POINTER(Shape), C.c_int, i64p]
        L.msig_workspace_bytes.argtypes = [C.POINTER(Shape), C.c_int]
        L.msig_workspace_bytes.restype = C.c_int64
        for name in ("msig_frontend_fwd", "msig_gru_fwd", "msig_head_ce_fwd", "msig_gru_bwd", "msig_frontend_bwd",
                     "msig_forward"):
            getattr(L, name).argtypes = [C.POINTER(Batch), vp]
        for name in ("msig_head_ce_bwd", "msig_backward"):
            getattr(L, name).argtypes = [C.POINTER(Batch), vp, vp]
        L.msig_adam_step.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.c_float, C.c_int64, vp]
        L.msig_train_step.argtypes = [C.POINTER(Batch), vp, vp, C.c_float, C.c_float, C.c_float, C.c_float,
                                      C.c_float, C.c_int64, vp]
        L.msig_dropout_key.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        L.msig_dropout_key.restype = C.c_uint32
        L.msig_gather_windows.argtypes = [vp, vp, vp, C.c_int32, C.c_int64, vp, vp, vp]
        L.msig_normalise_scratch_bytes.restype = C.c_int64
        L.msig_normalise_subject.argtypes = [vp, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_uint32, vp, vp, vp]
        L.msig_channel_attention.argtypes = [vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp]
        L.msig_profile_enable.argtypes = [C.c_int]
        L.msig_forward_multi.argtypes = [C.POINTER(Batch), C.POINTER(Multi), vp]
        L.msig_train_step_multi.argtypes = [C.POINTER(Batch), C.POINTER(Multi), vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int64, vp]
        L.msig_gather_windows_multi.argtypes = [vp, vp, vp, C.c_int64, C.c_int32, C.c_int64, vp, vp, C.POINTER(Multi), vp]
        L.msig_profile_report.argtypes = [C.c_char_p, C.c_int64]
        L.msig_profile_report.restype = C.c_int64
        _lib = L
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = ERRORS.get(rc, f"hipError_t {rc}" if rc > 0 else f"error {rc}")
        raise RuntimeError(f"{what} failed: {msg}")


def stage_lengths(T: int):
    out = (C.c_int32 * 4)()
    check(lib().msig_stage_lengths(T, out), "msig_stage_lengths")
    return tuple(int(v) for v in out)


def param_layout(Cin: int, K: int):
    """Offsets (floats) of the NPARAM tensors in the flat buffer; last entry = total."""
    off = (C.c_int64 * (NPARAM + 1))()
    check(lib().msig_param_layout(Cin, K, off), "msig_param_layout")
    return [int(v) for v in off]


def param_shapes(Cin: int, K: int):
    G, H = 192, 64
    shapes = [(Cin // 4, Cin), (Cin, Cin // 4), (16, Cin, 7), (16,), (16,), (32, 16, 5), (32,), (32,)]
    for layer in (0, 1):
        for _ in range(2):
            shapes += [(G, 128 if layer else 32), (G, H), (G,), (G,)]
    shapes += [(64, 128), (64,), (K, 64), (K,)]
    return shapes


def workspace_layout(B: int, Cin: int, T: int, K: int, training: bool):
    sh = Shape(B, Cin, T, K)
    off = (C.c_int64 * (NWS + 1))()
    check(lib().msig_workspace_layout(C.byref(sh), int(training), off), "msig_workspace_layout")
    return [int(v) for v in off]


def dropout_key(seed: int, step: int, stream_id: int) -> int:
    return int(lib().msig_dropout_key(seed & (2 ** 64 - 1), step & (2 ** 64 - 1), stream_id))


def _fmix32_np(h):
    import numpy as np
    h = h.astype(np.uint64)
    h ^= h >> np.uint64(16); h = (h * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13); h = (h * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    return h


def dropout_keys(seed: int, steps, stream_id: int):
    """msig_dropout_key for an array of steps at once (same mixing; tests/test_host_logic.py checks it against the C function)."""
    import numpy as np
    steps = np.asarray(steps, dtype=np.uint64)
    lo, hi = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    a = (steps * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
    b = np.uint64((stream_id * 0x7F4A7C15) & 0xFFFFFFFF)
    inner = _fmix32_np((a + b + hi) & np.uint64(0xFFFFFFFF))
    return _fmix32_np(lo ^ inner).astype(np.uint32)


def dropout_threshold(p: float) -> int:
    return int(round(float(p) * 256.0))


FORM_AUTO = -1
FWD_FORMS = {"auto": -1, "split": 0, "fused": 1, "b3": 1, "fp32": 2, "ws": 3}       # msig.h MSIG_FWD_*
BWD_FORMS = {"auto": -1, "split": 0, "fused": 1, "b3": 2, "b4": 3, "b5": 4, "b6": 5}                   # msig.h MSIG_BWD_*

# Kernel forms are per call in the C ABI (msig_batch.fwd_form / bwd_form = enumerator + 1, 0 = the library's default).  This
# binding keeps a default pair that runtime.Engine / FoldArena put into every descriptor they build (diagnostics / tests).
_default_forms = [0, 0]


def set_kernel_form(fwd="auto", bwd="auto"):
    """Default GRU kernel forms of the descriptors this binding builds from now on; "auto" = the library picks by batch size."""
    _default_forms[0], _default_forms[1] = FWD_FORMS[fwd] + 1, BWD_FORMS[bwd] + 1


def apply_forms(b: "Batch", fwd=None, bwd=None):
    """Writes the kernel forms into a descriptor: the given names, else the binding's defaults (set_kernel_form)."""
    b.fwd_form = _default_forms[0] if fwd is None else FWD_FORMS[fwd] + 1
    b.bwd_form = _default_forms[1] if bwd is None else BWD_FORMS[bwd] + 1
    return b


def profile_enable(on: bool):
    check(lib().msig_profile_enable(int(on)), "msig_profile_enable")


def profile_report():
    """{kernel name: (launches, total_ms)} for everything launched since profile_enable(True)."""
    buf = C.create_string_buffer(1 << 16)
    n = lib().msig_profile_report(buf, len(buf))
    if n < 0:
        raise RuntimeError(f"msig_profile_report failed: {n}")
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.split()
        out[name] = (int(cnt), float(ms))
    return out

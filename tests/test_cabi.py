"""CPU tier: libmsig_hip.so loads without a GPU, exports every symbol include/msig.h
declares, and its host-only layout functions agree with the oracle's inventory."""
import ctypes as C
import re

import numpy as np
import pytest

from conftest import ROOT
from oracle import cnn_gru_oracle as O
from multimodalsignal_amd import _lib as L

HEADER = (ROOT / "include" / "msig.h").read_text()


def test_library_exports_every_declared_symbol():
    lib = L.lib()
    names = sorted(set(re.findall(r"\b(msig_[a-z0-9_]+)\s*\(", HEADER)))
    assert len(names) == 26, names          # ABI 4: msig_set_kernel_form is gone (kernel forms are per call)
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/msig.h but not exported"
    assert lib.msig_abi_version() == int(re.search(r"#define MSIG_ABI_VERSION (\d+)", HEADER).group(1)) == L.ABI_VERSION
    # the ctypes mirrors of the two descriptor structs have the library's own sizes (also checked at load time by _lib.lib())
    import ctypes as C
    assert lib.msig_struct_bytes(0) == C.sizeof(L.Batch) and lib.msig_struct_bytes(1) == C.sizeof(L.Multi) and lib.msig_struct_bytes(2) == -1


def test_python_enum_mirrors_match_header():
    ws_block = HEADER[HEADER.index("enum msig_ws"):HEADER.index("MSIG_NWS")]
    ws = re.findall(r"MSIG_WS_([A-Z0-9_]+)", ws_block)
    assert ws == L.WS_NAMES
    p_block = HEADER[HEADER.index("enum msig_param"):HEADER.index("MSIG_NPARAM")]
    p_names = re.findall(r"^\s*MSIG_P_([A-Z0-9_]+)", p_block, flags=re.M)
    assert p_names == ["GATE_W1", "GATE_W2", "CONV1_W", "BN1_G", "BN1_B", "CONV2_W", "BN2_G", "BN2_B", "GRU",
                       "CLS0_W", "CLS0_B", "CLS3_W", "CLS3_B"]
    assert L.NPARAM == 28 and L.P_CLS0_W == L.P_GRU + 16
    assert int(re.search(r"#define MSIG_BN_STATE_FLOATS (\d+)", HEADER).group(1)) == L.BN_STATE_FLOATS
    # kernel forms (msig_batch.fwd_form / bwd_form = enumerator + 1): the binding's tables name exactly the header's enumerators, with their values
    for table, prefix, alias in ((L.FWD_FORMS, "MSIG_FWD_", {"LATENCY": "split", "B3": "fused", "FP32": "fp32", "WS": "ws"}),
                                 (L.BWD_FORMS, "MSIG_BWD_", {"SPLIT": "split", "FUSED": "fused", "B3": "b3", "B4": "b4", "B5": "b5", "B6": "b6"})):
        enum = dict((n, int(v)) for n, v in re.findall(prefix + r"([A-Z0-9]+) = (\d+)", HEADER))
        assert set(enum) == set(alias), (prefix, sorted(enum))
        for n, v in enum.items():
            assert table[alias[n]] == v, (prefix + n, v, table)
        assert {v for k, v in table.items() if k != "auto"} == set(enum.values())       # extra names ("b3" for FUSED) are aliases of enumerators
        assert table["auto"] == -1                                                      # + 1 = 0 = "the library's default" in the descriptor
    errs = dict((n, int(v)) for n, v in re.findall(r"#define MSIG_E_([A-Z]+)\s+\((-\d+)\)", HEADER))
    assert set(errs.values()) == set(L.ERRORS) and all(f"MSIG_E_{n}" in L.ERRORS[v] for n, v in errs.items())


def test_kernel_forms_are_per_descriptor_and_checked():
    """ABI 4: the forms live in the descriptor (no process-global setter); a value that names no form is MSIG_E_FORM before any launch."""
    lib = L.lib()
    assert not hasattr(lib, "msig_set_kernel_form")
    try:
        L.set_kernel_form("ws", "b6")
        b = L.apply_forms(L.Batch())
        assert (b.fwd_form, b.bwd_form) == (L.FWD_FORMS["ws"] + 1, L.BWD_FORMS["b6"] + 1)
        b2 = L.apply_forms(L.Batch(), "split", "split")              # explicit names win over the binding's default pair
        assert (b2.fwd_form, b2.bwd_form) == (1, 1)
    finally:
        L.set_kernel_form("auto", "auto")
    assert (L.apply_forms(L.Batch()).fwd_form, L.apply_forms(L.Batch()).bwd_form) == (0, 0)
    # argument order of the checks: NULL buffers first (-1), then — with buffers present — the form fields (-5), nothing launched
    keep = (C.c_char * 4096)()
    addr = (C.addressof(keep) + 255) // 256 * 256
    b = L.Batch()
    b.shape = L.Shape(4, 6, 256, 2)
    for f in ("x", "params", "grads", "bn_state", "bn_count", "ws"):
        setattr(b, f, addr)
    b.ws_bytes = 1 << 40
    for fwd, bwd in ((9, 0), (0, 9), (-1, 0), (0, L.BWD_FORMS["b6"] + 2)):
        b.fwd_form, b.bwd_form = fwd, bwd
        assert lib.msig_forward(C.byref(b), None) == -5 and lib.msig_backward(C.byref(b), None, None) == -5


def test_integration_md_binding_stub_matches_the_library():
    """INTEGRATION.md section B shows the ctypes stub a maintainer would paste into the reference.  Its struct mirrors and its
    load-time assertion are executed here against the built library, so the document cannot go stale silently (round 3: it had)."""
    text = (ROOT / "INTEGRATION.md").read_text()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "class Batch(C.Structure)" in b)
    head = stub[:stub.index("def train_step")]                   # imports, CDLL, struct mirrors, the ABI / size assertion
    assert "msig_abi_version() ==" in head and "msig_struct_bytes(0) == C.sizeof(Batch)" in head
    import os
    cwd = os.getcwd()
    os.chdir(ROOT)                                               # the stub loads the library by its in-tree relative path
    try:
        ns = {}
        exec(compile(head, "INTEGRATION.md#B", "exec"), ns)      # raises AssertionError when the mirror and the library disagree
    finally:
        os.chdir(cwd)
    assert [f[0] for f in ns["Batch"]._fields_] == [f[0] for f in L.Batch._fields_]
    assert C.sizeof(ns["Batch"]) == C.sizeof(L.Batch) == L.lib().msig_struct_bytes(0)
    assert f"== {L.ABI_VERSION} " in head or f"== {L.ABI_VERSION}\n" in head


def test_fold_arena_workspace_covers_ragged_batches_below_the_latency_threshold():
    """msig_workspace_layout is not monotonic in B (the projection region exists only below 192 batch tiles): an arena sized for
    B = 3072 alone would be too small for a ragged last batch of 3056 windows.  FoldArena.workspace_bytes (what the arena allocates;
    no GPU needed) must cover every batch size up to the arena's, in training and in evaluation."""
    from multimodalsignal_amd.runtime import FoldArena
    small, big = L.workspace_layout(3056, 6, 3840, 2, True)[-1], L.workspace_layout(3072, 6, 3840, 2, True)[-1]
    assert small > big
    for tb, eb in ((3072, 0), (3073, 0), (3056, 0), (64, 3072), (64, 4096), (4000, 1024)):
        ws = FoldArena.workspace_bytes(tb, eb, 6, 3840, 2)
        for b_ in (1, 17, 64, 3055, 3056, 3057, 3071, 3072, 3073, 4000, 4096):
            if b_ <= tb:
                assert ws >= L.workspace_layout(b_, 6, 3840, 2, True)[-1], (tb, eb, b_, "training")
            if b_ <= (eb or tb):
                assert ws >= L.workspace_layout(b_, 6, 3840, 2, False)[-1], (tb, eb, b_, "evaluation")
    # and the library refuses an undersized workspace instead of running past it (MSIG_E_WORKSPACE = -4, checked before any launch)
    b = L.Batch()
    b.shape = L.Shape(3056, 6, 3840, 2)
    b.training = 1
    b.x = b.params = b.grads = b.bn_state = b.bn_count = b.ws = 4096       # any non-NULL, aligned address: rejected before it is used
    b.ws_bytes = big
    assert L.lib().msig_forward(C.byref(b), None) == -4


@pytest.mark.parametrize("C_,K", [(1, 2), (2, 3), (3, 2), (4, 2), (6, 2), (8, 3), (16, 16)])
def test_param_layout_matches_oracle_inventory(C_, K):
    off = L.param_layout(C_, K)
    specs = O.param_specs(C_, K)
    assert list(specs.keys()) == L.PARAM_KEYS
    assert [tuple(s) for s in specs.values()] == [tuple(s) for s in L.param_shapes(C_, K)]
    end = 0
    for i, shp in enumerate(specs.values()):
        n = int(np.prod(shp)) if len(shp) else 1
        assert off[i] % 4 == 0 and off[i] >= end
        end = off[i] + n
    assert off[-1] >= end and off[-1] % 4 == 0
    assert off[-1] - sum(int(np.prod(s)) for s in specs.values()) < 4 * L.NPARAM


def test_stage_lengths_and_workspace_layout():
    for T in (16, 136, 200, 256, 3840, 7680):
        assert L.stage_lengths(T) == O.stage_lengths(T)
    assert L.stage_lengths(3840) == (1920, 960, 480, 240)
    for training in (False, True):
        off = L.workspace_layout(64, 6, 3840, 2, training)
        assert len(off) == L.NWS + 1 and all(o % 256 == 0 for o in off) and all(b >= a for a, b in zip(off, off[1:]))
    ev, tr = L.workspace_layout(64, 6, 3840, 2, False), L.workspace_layout(64, 6, 3840, 2, True)
    i = L.WS["STASH0"]
    assert ev[i + 1] - ev[i] == 0 and tr[i + 1] - tr[i] == 2 * 4 * 240 * 4096 * 4
    big = L.workspace_layout(8192, 6, 3840, 2, True)[-1]
    assert 8e9 < big < 40e9          # fits one MI355X many times over


def test_argument_errors_are_reported_not_launched():
    lib = L.lib()
    off = (C.c_int64 * (L.NWS + 1))()
    for bad in (L.Shape(0, 6, 3840, 2), L.Shape(4, 0, 3840, 2), L.Shape(4, 17, 3840, 2), L.Shape(4, 6, 8, 2), L.Shape(4, 6, 3840, 1)):
        assert lib.msig_workspace_layout(C.byref(bad), 1, off) == -2
    assert lib.msig_workspace_layout(None, 1, off) == -1
    assert lib.msig_param_layout(6, 2, None) == -1
    with pytest.raises(RuntimeError, match="MSIG_E_SHAPE"):
        L.param_layout(99, 2)
    b = L.Batch()
    b.shape = L.Shape(4, 6, 256, 2)
    assert lib.msig_forward(C.byref(b), None) == -1          # NULL buffers: refused before any launch
    assert lib.msig_adam_step(None, None, None, None, 16, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, None) == -1


def test_dropout_key_matches_oracle():
    for seed, step, stream in ((0, 0, 1), (42, 7, 1), (42, 7, 2), (2 ** 63 + 12345, 10 ** 9, 2), (99, 123456789012, 1)):
        assert L.dropout_key(seed, step, stream) == O.dropout_key(seed, step, stream)
    assert L.dropout_threshold(0.5) == O.dropout_threshold(0.5) == 128

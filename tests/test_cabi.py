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
    assert len(names) == 27, names
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
    # kernel forms (msig_set_kernel_form): the binding's tables name exactly the header's enumerators, with their values
    for table, prefix, alias in ((L.FWD_FORMS, "MSIG_FWD_", {"LATENCY": "split", "B3": "fused", "FP32": "fp32", "WS": "ws"}),
                                 (L.BWD_FORMS, "MSIG_BWD_", {"SPLIT": "split", "FUSED": "fused", "B3": "b3", "B4": "b4", "B5": "b5", "B6": "b6", "B7": "b7"})):
        enum = dict((n, int(v)) for n, v in re.findall(prefix + r"([A-Z0-9]+) = (\d+)", HEADER))
        assert set(enum) == set(alias), (prefix, sorted(enum))
        for n, v in enum.items():
            assert table[alias[n]] == v, (prefix + n, v, table)
        assert {v for k, v in table.items() if k != "auto"} == set(enum.values())       # extra names ("b3" for FUSED) are aliases of enumerators
        assert table["auto"] == int(re.search(r"#define MSIG_FORM_AUTO \((-?\d+)\)", HEADER).group(1))


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

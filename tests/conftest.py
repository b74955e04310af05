import json
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden_model(name):
    """Returns (meta, params, arrays) for a model_* fixture; resolves shared weights / seeded x."""
    z = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    arrays = {k: z[k] for k in z.files if k != "meta"}
    src = arrays
    if meta.get("weights_from"):
        zz = np.load(GOLDEN / f"{meta['weights_from']}.npz", allow_pickle=False)
        src = {k: zz[k] for k in zz.files}
    params = {k[len("param/"):]: v for k, v in src.items() if k.startswith("param/")}
    if "x" not in arrays:
        arrays["x"] = golden_x(meta)
    return meta, params, arrays


def golden_x(meta):
    """Re-creates the input of a fixture that stores only its seed (see make_golden.model_case)."""
    rs = np.random.RandomState(meta["xseed"])
    B, C, T = meta["B"], meta["C"], meta["T"]
    x = rs.randn(B, C, T).astype(np.float32)
    x = x * (0.5 + rs.rand(1, C, 1).astype(np.float32)) + rs.randn(1, C, 1).astype(np.float32)
    return x


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

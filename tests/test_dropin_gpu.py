"""GPU tier: the reference-side binding of INTEGRATION.md section A — `dropin/` first on sys.path, then exactly the imports and the
fold body of the reference's main.py (main.py:10-12, 105-124): stock torch DataLoaders over WesadDataset objects handed to
Trainer.train / Trainer.evaluate.  Runs in a child interpreter so that the top-level module names `models`, `trainer`,
`dataset` resolve the way they do for the reference's driver."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import GOLDEN

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu

DRIVER = r'''
import json, sys
from pathlib import Path
import numpy as np
import torch
from torch.utils.data import DataLoader
from dataset import WesadDataset                      # main.py:10
from models import CnnGruAttentionModel               # main.py:11
from trainer import Trainer                           # main.py:12
import models, trainer, dataset
assert all(Path(m.__file__).parent.name == "dropin" for m in (models, trainer, dataset)), "the dropin shims must be what `import models` finds"

data, out, shuffle = Path(sys.argv[1]), Path(sys.argv[2]), sys.argv[3] == "1"
names = ["chest_ECG", "chest_EDA"]
BATCH_SIZE, NUM_WORKERS = 16, 0
train_ds = WesadDataset(data, ["S2", "S3", "S4"], names, names, classification_mode="stress_binary")      # main.py:105-110
val_ds = WesadDataset(data, ["S5"], names, names, classification_mode="stress_binary")
test_ds = WesadDataset(data, ["S6"], names, names, classification_mode="stress_binary")
torch.manual_seed(1234)
train_loader = DataLoader(train_ds, batch_size=BATCH_SIZE, shuffle=shuffle, num_workers=NUM_WORKERS)       # main.py:112-114
val_loader = DataLoader(val_ds, batch_size=BATCH_SIZE, shuffle=False, num_workers=NUM_WORKERS)
test_loader = DataLoader(test_ds, batch_size=BATCH_SIZE, shuffle=False, num_workers=NUM_WORKERS)
model = CnnGruAttentionModel(in_channels=len(names), num_classes=2, cnn_out_channels=32, gru_hidden_size=64, gru_num_layers=2, dropout=0.0)
config_dict = {'trainer': {'epochs': 2, 'learning_rate': 0.001, 'early_stopping': {'enabled': True, 'patience': 20, 'delta': 0},
                           'weight_decay': 1e-4}}                                                          # main.py:119-121
trainer_ = Trainer(model, out, config_dict)
trainer_.train(train_loader, val_loader)
_, test_acc, test_f1 = trainer_.evaluate(test_loader, is_test=True)                                        # main.py:124
hist = [[h["train_loss"], h["val_loss"], h["val_acc"], h["val_f1"]] for h in trainer_.history]
sd = {k: v.detach().cpu().numpy().tolist() for k, v in model.state_dict().items() if k in ("classifier.3.weight", "cnn_encoder.1.running_var")}
print("RESULT " + json.dumps({"hist": hist, "test": [test_acc, test_f1], "sd": sd}))
'''


def _write_subjects(z, td):
    for k in z.files:
        if k.startswith("raw/"):
            np.save(Path(td) / f"{k[4:]}.npy", z[k])


def _run_driver(tmp_path, data, tag, shuffle):
    script = tmp_path / f"driver_{tag}.py"
    script.write_text(DRIVER)
    env = dict(os.environ, PYTHONPATH=str(ROOT / "dropin"), MPLBACKEND="Agg")       # dropin/ first, nothing else
    r = subprocess.run([sys.executable, str(script), str(data), str(tmp_path / f"fold_{tag}"), "1" if shuffle else "0"],
                       capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert len(line) == 1, r.stdout[-2000:]
    return json.loads(line[0][7:])


@pytest.mark.timeout(900)
def test_reference_driver_body_through_dropin_modules(tmp_path):
    import torch
    from multimodalsignal_amd.dataset import DeviceLoader, WesadDataset
    from multimodalsignal_amd.models import CnnGruAttentionModel
    from multimodalsignal_amd.trainer import Trainer
    z = np.load(GOLDEN / "trainer_e2e.npz", allow_pickle=False)
    data = tmp_path / "data"
    data.mkdir()
    _write_subjects(z, data)
    got = _run_driver(tmp_path, data, "stock", shuffle=False)
    # the same two epochs through the GPU-resident DeviceLoader path, in this process
    dev = torch.device("cuda:0")
    names = ["chest_ECG", "chest_EDA"]
    mk = lambda s: WesadDataset(data, s, names, names, classification_mode="stress_binary")
    torch.manual_seed(1234)
    model = CnnGruAttentionModel(in_channels=2, num_classes=2, dropout=0.0)
    cfg = {"trainer": {"epochs": 2, "learning_rate": 1e-3, "early_stopping": {"enabled": True, "patience": 20, "delta": 0},
                       "weight_decay": 1e-4, "verbose": False}}
    t = Trainer(model, tmp_path / "fold_dev", cfg)
    t.train(DeviceLoader(mk(["S2", "S3", "S4"]), 16, False, dev), DeviceLoader(mk(["S5"]), 16, False, dev))
    _, acc, f1 = t.evaluate(DeviceLoader(mk(["S6"]), 16, False, dev), is_test=True)
    want = [[h["train_loss"], h["val_loss"], h["val_acc"], h["val_f1"]] for h in t.history]
    assert got["hist"] == want            # same windows in the same order through the same kernels: identical, not merely close
    assert got["test"] == [acc, f1]
    sd = model.state_dict()
    for k, v in got["sd"].items():
        np.testing.assert_array_equal(np.asarray(v, dtype=np.float32), sd[k].cpu().numpy(), err_msg=k)
    for f in ("training_log.txt", "best_model.pt", "test_confusion_matrix.png"):
        assert (tmp_path / "fold_stock" / f).exists(), f
    # and with the reference's shuffle=True (torch's own sampler decides the order): trains, finite, artefacts written
    sh = _run_driver(tmp_path, data, "shuffled", shuffle=True)
    assert np.isfinite(np.asarray(sh["hist"])).all() and len(sh["hist"]) == 2 and 0.0 <= sh["test"][0] <= 1.0


def test_integration_md_stub_trains_a_step_like_the_binding():
    """INTEGRATION.md section B's ctypes stub, executed verbatim: its train_step on raw flat buffers leaves the same weights and
    the same loss as runtime.Engine.train_step (this repo's own binding) — the document is a working binding, not prose."""
    import re
    import numpy as np
    import torch
    from conftest import ROOT
    from multimodalsignal_amd import _lib as L
    from multimodalsignal_amd.runtime import Engine
    from oracle import cnn_gru_oracle as O
    text = (ROOT / "INTEGRATION.md").read_text()
    stub = next(b for b in re.findall(r"```python\n(.*?)```", text, flags=re.S) if "class Batch(C.Structure)" in b)
    import os
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        ns = {}
        exec(compile(stub, "INTEGRATION.md#B", "exec"), ns)
    finally:
        os.chdir(cwd)
    dev = torch.device("cuda:0")
    B, C, K, T = 12, 6, 2, 256
    params = O.init_params(C, K, seed=9)
    rs = np.random.RandomState(4)
    x = torch.as_tensor(rs.randn(B, C, T).astype(np.float32)).to(dev)
    y = torch.as_tensor(rs.randint(0, K, size=(B,)).astype(np.int64)).to(dev)
    ref = Engine(C, K, dev)
    ref.load_named(params)
    ref.ensure_adam_state()
    flat = ref.params.clone()                                   # the stub's caller owns plain flat buffers in msig_param_layout order
    grads, m, v = torch.zeros_like(flat), torch.zeros_like(flat), torch.zeros_like(flat)
    bn_state, bn_count = ref.bn_state.clone(), ref.bn_count.clone()
    ws = ns["train_step"](x, y, flat, grads, m, v, bn_state, bn_count, step=1, lr=1e-3, wd=1e-4, p=0.5, seed=77)
    ref.train_step(x, y, lr=1e-3, weight_decay=1e-4, step=1, dropout_p=0.5, seed=77)
    torch.cuda.synchronize()
    assert torch.equal(flat, ref.params) and torch.equal(bn_state, ref.bn_state) and torch.equal(m, ref.exp_avg)
    off = L.workspace_layout(B, C, T, K, True)[L.WS["LOSS"]]
    assert float(ws[off:off + 4].view(torch.float32)[0]) == float(ref.region("LOSS")[0]) > 0

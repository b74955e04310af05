#!/usr/bin/env python3
"""Regenerates the golden vectors in this directory FROM THE REFERENCE ITSELF.

Runs only in the build container (needs ``/root/reference``); the fixtures it
writes are data (inputs + the reference's outputs) and are what travels to the
GPU box.  Nothing here is imported by the product.

    python tests/golden/make_golden.py

What is imported from the reference: ``models.py`` (the model),
``dataset.py`` (WesadDataset), ``trainer.py`` (Trainer/EarlyStopping; needs an
empty ``seaborn`` stand-in module because that plotting package is absent).
"""
import json
import os
import re
import sys
import tempfile
import types
from pathlib import Path

os.environ.setdefault("MPLBACKEND", "Agg")
import numpy as np
import torch

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(REF))
_sns = types.ModuleType("seaborn")
_sns.heatmap = lambda *a, **k: None
sys.modules.setdefault("seaborn", _sns)

import models as ref_models      # noqa: E402
import dataset as ref_dataset    # noqa: E402
import trainer as ref_trainer    # noqa: E402

torch.set_num_threads(4)


def np_state(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def perturb_norm_layers(model, seed):
    """Make BN affine params / running stats non-trivial so tests can see them."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for idx in (1, 5):
            bn = model.cnn_encoder[idx]
            bn.weight.copy_(1.0 + 0.3 * torch.randn(bn.weight.shape, generator=g))
            bn.bias.copy_(0.2 * torch.randn(bn.bias.shape, generator=g))
            bn.running_mean.copy_(0.1 * torch.randn(bn.running_mean.shape, generator=g))
            bn.running_var.copy_(0.5 + torch.rand(bn.running_var.shape, generator=g))


def model_case(name, C, K, T, B, wseed, xseed, store_stages=True, store_x=True, weights_from=None, **model_kwargs):
    torch.manual_seed(wseed)
    model = ref_models.CnnGruAttentionModel(in_channels=C, num_classes=K, dropout=0.0, **model_kwargs)
    perturb_norm_layers(model, wseed + 1)
    rs = np.random.RandomState(xseed)
    x = rs.randn(B, C, T).astype(np.float32)
    # give channels different offsets/scales so the gate sees distinct means
    x = x * (0.5 + rs.rand(1, C, 1).astype(np.float32)) + rs.randn(1, C, 1).astype(np.float32)
    y = rs.randint(0, K, size=(B,)).astype(np.int64)
    out = {}
    init_sd = np_state(model.state_dict())
    if weights_from is None:
        for k, v in init_sd.items():
            out["param/" + k] = v
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)

    model.eval()
    with torch.no_grad():
        out["eval_logits"] = model(xt).numpy()

    stages = {}
    hooks = []

    def grab(key):
        def fn(_m, _i, o):
            stages[key] = (o[0] if isinstance(o, tuple) else o).detach().numpy().copy()
        return fn

    hooks.append(model.channel_attention.fc.register_forward_hook(grab("gate_s")))
    for idx, key in ((0, "conv1"), (1, "bn1"), (3, "pool1"), (4, "conv2"), (5, "bn2"), (7, "pool2")):
        hooks.append(model.cnn_encoder[idx].register_forward_hook(grab(key)))
    hooks.append(model.gru.register_forward_hook(grab("gru_out")))
    hooks.append(model.classifier[2].register_forward_hook(grab("cls_hidden")))

    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)   # trainer.py:68
    crit = torch.nn.CrossEntropyLoss()                                        # trainer.py:69
    for step in (1, 2, 3):
        opt.zero_grad()
        logits = model(xt)
        loss = crit(logits, yt)
        loss.backward()
        if step == 1:
            out["train_logits"] = logits.detach().numpy().copy()
            out["train_loss"] = np.float64(loss.item())
            for k, p in model.named_parameters():
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                if store_stages or p.numel() <= 4096:
                    out["grad/" + k] = g.detach().numpy().copy()
                out["gradnorm/" + k] = np.float64(g.double().norm().item())
            if store_stages:
                for k, v in stages.items():
                    out["stage/" + k] = v
            else:
                out["stage/gate_s"] = stages["gate_s"]
                out["stage/gru_last"] = stages["gru_out"][:, -1, :]
        opt.step()
        if step == 1:
            for k, v in np_state(model.state_dict()).items():
                if "running" in k or "num_batches" in k or store_stages:
                    out["after1/" + k] = v
        out[f"loss_step{step}"] = np.float64(loss.item())
    if store_stages:
        for k, v in np_state(model.state_dict()).items():
            out["after3/" + k] = v
    for h in hooks:
        h.remove()
    if store_x:
        out["x"] = x
    out["y"] = y
    out["meta"] = np.array(json.dumps(dict(C=C, K=K, T=T, B=B, wseed=wseed, xseed=xseed,
                                           weights_from=weights_from, store_x=store_x, **model_kwargs)))
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print("wrote", name, sum(v.nbytes for v in out.values() if hasattr(v, "nbytes")) // 1024, "KiB raw")


def spec_case():
    spec = {}
    for C in (1, 2, 3, 4, 6):
        for K in (2, 3):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                m = ref_models.CnnGruAttentionModel(in_channels=C, num_classes=K)
            spec[f"C{C}_K{K}"] = {
                "state_dict": [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()],
                "parameters": [k for k, _ in m.named_parameters()],
                "n_params": int(sum(p.numel() for p in m.parameters())),
            }
    # init statistics: bounds of the default initialisers
    torch.manual_seed(0)
    m = ref_models.CnnGruAttentionModel(in_channels=6, num_classes=2)
    bounds = {k: [float(v.min()), float(v.max())] for k, v in m.state_dict().items() if v.dtype.is_floating_point and v.numel()}
    spec["init_bounds_C6_K2"] = bounds
    (OUT / "state_dict_specs.json").write_text(json.dumps(spec, indent=1))
    print("wrote state_dict_specs.json")


def dataset_case():
    names = ["chest_ACC_x", "chest_ECG", "chest_EDA", "chest_Resp"]
    rs = np.random.RandomState(5)
    raw = {}
    for sid, n in (("S2", 5), ("S3", 4)):
        X = rs.randn(n, 64, 4) * np.array([1.0, 0.3, 0.5, 2.0]) + np.array([0.0, 0.1, 3.0, -1.0])
        X[:, :, 2] = np.abs(X[:, :, 2]) + 0.05           # EDA is positive (log1p)
        yv = rs.choice([1, 2, 3, 4], size=n)
        yv[0] = 2
        raw[sid] = (X.astype(np.float64), yv.astype(np.int64))
    out = {}
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        for sid, (X, yv) in raw.items():
            np.save(td / f"{sid}_X.npy", X)
            np.save(td / f"{sid}_y.npy", yv)
            out[f"raw/{sid}_X"] = X
            out[f"raw/{sid}_y"] = yv
        for mode in ("stress_binary", "ternary"):
            for chans in (["chest_ECG", "chest_EDA"], ["chest_Resp", "chest_ACC_x", "chest_EDA"]):
                ds = ref_dataset.WesadDataset(td, ["S2", "S9", "S3"], chans, names, classification_mode=mode)
                key = f"{mode}/{'+'.join(chans)}"
                out[key + "/data"] = ds.data
                out[key + "/labels"] = ds.labels
                xi, yi = ds[3]
                out[key + "/item3_x"] = xi.numpy()
                out[key + "/item3_y"] = yi.numpy()
                out[key + "/len"] = np.int64(len(ds))
        try:
            ref_dataset.WesadDataset(td, ["S2"], ["chest_ECG"], names, classification_mode="amusement_binary")
            out["bad_mode_raises"] = np.array("no")
        except ValueError as e:
            out["bad_mode_raises"] = np.array(str(e))
        try:
            ref_dataset.WesadDataset(td, ["S9"], ["chest_ECG"], names)
            out["empty_raises"] = np.array("no")
        except ValueError as e:
            out["empty_raises"] = np.array(str(e))
    out["channel_names"] = np.array(json.dumps(names))
    np.savez_compressed(OUT / "dataset.npz", **out)
    print("wrote dataset")


def control_case():
    """EarlyStopping (trainer.py:12-39) and ReduceLROnPlateau (trainer.py:72-77) traces."""
    traces = {}

    class FakeModel:
        def state_dict(self):
            return {}

    seqs = {
        "falling": [1.0, 0.9, 0.8, 0.7, 0.6, 0.5, 0.4],
        "rising": [0.5, 0.6, 0.7, 0.8, 0.9],
        "mixed": [0.7, 0.6, 0.8, 0.75, 0.7, 0.9, 0.85, 0.8, 0.7, 0.6, 0.5, 0.4],
        "equal": [0.5, 0.5, 0.5, 0.4],
    }
    for name, seq in seqs.items():
        for patience in (3, 20):
            with tempfile.TemporaryDirectory() as td:
                es = ref_trainer.EarlyStopping(patience=patience, delta=0, checkpoint_path=Path(td) / "c.pt")
                saves = []
                es.save_checkpoint = lambda model, _s=saves: _s.append(1)
                rows = []
                for v in seq:
                    n0 = len(saves)
                    es(v, FakeModel())
                    rows.append([v, es.counter, es.best_score, bool(len(saves) > n0), bool(es.early_stop)])
                    if es.early_stop:
                        break
                traces[f"es/{name}/p{patience}"] = rows
    lin = torch.nn.Linear(1, 1)
    opt = torch.optim.Adam(lin.parameters(), lr=1e-3, weight_decay=1e-4)
    sch = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.1, patience=3)
    vals = [1.0, 0.9, 0.95, 0.93, 0.92, 0.91, 0.90005, 0.95, 0.96, 0.97, 0.98, 0.5, 0.6, 0.6, 0.6, 0.6, 0.6]
    rows = []
    for v in vals:
        sch.step(v)
        rows.append([v, opt.param_groups[0]["lr"]])
    traces["plateau"] = rows
    (OUT / "trainer_control.json").write_text(json.dumps(traces, indent=1))
    print("wrote trainer_control.json")


def split_case():
    from sklearn.model_selection import train_test_split
    subs = [f"S{i}" for i in range(2, 18) if i != 12]          # main.py:67
    table = {}
    for s in subs:
        tv = [q for q in subs if q != s]
        tr, va = train_test_split(tv, test_size=0.2, random_state=42)   # main.py:102-103
        table[s] = {"train": tr, "val": va}
    (OUT / "loso_splits.json").write_text(json.dumps(table, indent=1))
    print("wrote loso_splits.json")


def metrics_case():
    from sklearn.metrics import accuracy_score, f1_score
    rs = np.random.RandomState(3)
    cases = []
    for n, k in ((50, 2), (37, 3), (10, 2), (8, 2)):
        yt = rs.randint(0, k, size=n)
        yp = rs.randint(0, k, size=n)
        cases.append((yt, yp))
    cases.append((np.zeros(9, dtype=int), np.zeros(9, dtype=int)))          # single class, all right
    cases.append((np.zeros(9, dtype=int), np.array([0, 1, 0, 0, 1, 0, 0, 0, 0])))  # single true class
    cases.append((np.array([0, 1, 1, 0, 1]), np.zeros(5, dtype=int)))       # never predicts 1
    import warnings
    rows = []
    for yt, yp in cases:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rows.append({"y_true": yt.tolist(), "y_pred": yp.tolist(),
                         "accuracy": float(accuracy_score(yt, yp)),
                         "f1_weighted": float(f1_score(yt, yp, average="weighted"))})   # trainer.py:234-235
    (OUT / "metrics.json").write_text(json.dumps(rows))
    print("wrote metrics.json")


def trainer_e2e_case():
    """Full-batch, dropout-free reference Trainer run on a tiny synthetic dataset."""
    names = ["chest_ECG", "chest_EDA"]
    rs = np.random.RandomState(21)
    T = 256
    out = {}
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        for sid, n in (("S2", 12), ("S3", 12), ("S4", 12), ("S5", 10), ("S6", 9)):
            yv = rs.choice([1, 2, 3, 4], size=n)
            yv[:3] = 2
            X = rs.randn(n, T, 2)
            tt = np.arange(T) / 64.0
            X[:, :, 0] += (yv == 2)[:, None] * 0.8 * np.sin(2 * np.pi * 1.5 * tt)[None, :]
            X[:, :, 1] = np.abs(X[:, :, 1]) + 0.1 + (yv == 2)[:, None] * 0.5
            np.save(td / f"{sid}_X.npy", X)
            np.save(td / f"{sid}_y.npy", yv)
            out[f"raw/{sid}_X"] = X
            out[f"raw/{sid}_y"] = yv
        mk = lambda subj: ref_dataset.WesadDataset(td, subj, names, names, classification_mode="stress_binary")
        tr, va, te = mk(["S2", "S3", "S4"]), mk(["S5"]), mk(["S6"])
        from torch.utils.data import DataLoader
        torch.manual_seed(1234)
        model = ref_models.CnnGruAttentionModel(in_channels=2, num_classes=2, dropout=0.0)
        for k, v in np_state(model.state_dict()).items():
            out["init/" + k] = v
        cfg = {"trainer": {"epochs": 6, "learning_rate": 1e-3,
                           "early_stopping": {"enabled": True, "patience": 20, "delta": 0},
                           "weight_decay": 1e-4}}
        fold = td / "fold"
        t = ref_trainer.Trainer(model, fold, cfg)
        rec = []
        orig_eval = t.evaluate

        def wrapped(loader, is_test=False, is_val=False):
            r = orig_eval(loader, is_test=is_test, is_val=is_val)
            rec.append([float(r[0]), float(r[1]), float(r[2])])
            return r

        t.evaluate = wrapped
        t.train(DataLoader(tr, batch_size=64, shuffle=True), DataLoader(va, batch_size=64, shuffle=False))
        test_loss, test_acc, test_f1 = t.evaluate(DataLoader(te, batch_size=64, shuffle=False), is_test=True)
        log = (fold / "training_log.txt").read_text()
        tl = [float(x) for x in re.findall(r"训练损失: ([0-9.]+)", log)]
        out["val_epochs"] = np.array(rec[:-1], dtype=np.float64)
        out["test"] = np.array(rec[-1], dtype=np.float64)
        out["train_loss_4dp"] = np.array(tl, dtype=np.float64)
        out["ckpt_exists"] = np.array((fold / "best_model.pt").exists())
        out["log_text"] = np.array(log)
        for k, v in np_state(model.state_dict()).items():
            out["final/" + k] = v
    np.savez_compressed(OUT / "trainer_e2e.npz", **out)
    print("wrote trainer_e2e; val epochs:\n", out["val_epochs"], "\ntest", out["test"])


def m2_case():
    """The hierarchical experiment's second model (main.py:35-40: gru_hidden_size=32, gru_num_layers=1, three chest channels)."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # nn.GRU warns that dropout has no effect with one layer
        model_case("model_m2_c3_k2_t512", C=3, K=2, T=512, B=5, wseed=21, xseed=22, gru_hidden_size=32, gru_num_layers=1)


if __name__ == "__main__":
    import sys
    import warnings
    warnings.filterwarnings("ignore", message="Initializing zero-element tensors is a no-op")
    if "--only-m2" in sys.argv:                  # added in round 3: generates this one fixture, leaves the others untouched
        m2_case()
        sys.exit(0)
    spec_case()
    model_case("model_c6_k2_t512", C=6, K=2, T=512, B=5, wseed=7, xseed=11)
    model_case("model_c2_k3_t256", C=2, K=3, T=256, B=3, wseed=8, xseed=12)
    model_case("model_c4_k2_t200", C=4, K=2, T=200, B=2, wseed=9, xseed=13)
    # full-size window: same weights as model_c6_k2_t512 (same wseed), x regenerated from its seed
    model_case("model_c6_k2_t3840", C=6, K=2, T=3840, B=2, wseed=7, xseed=14, store_stages=False,
               store_x=False, weights_from="model_c6_k2_t512")
    dataset_case()
    control_case()
    split_case()
    metrics_case()
    trainer_e2e_case()
    m2_case()

"""CPU tier: the Python host side (dataset, early stopping, scheduler, metrics, splits,
model container) against traces recorded from the reference (tests/golden/)."""
import json
import tempfile
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import GOLDEN


def test_dataset_matches_reference():
    from multimodalsignal_amd.dataset import WesadDataset
    z = np.load(GOLDEN / "dataset.npz", allow_pickle=False)
    names = json.loads(str(z["channel_names"]))
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        for sid in ("S2", "S3"):
            np.save(td / f"{sid}_X.npy", z[f"raw/{sid}_X"])
            np.save(td / f"{sid}_y.npy", z[f"raw/{sid}_y"])
        for mode in ("stress_binary", "ternary"):
            for chans in (["chest_ECG", "chest_EDA"], ["chest_Resp", "chest_ACC_x", "chest_EDA"]):
                ds = WesadDataset(td, ["S2", "S9", "S3"], chans, names, classification_mode=mode)   # S9 missing: skipped
                key = f"{mode}/{'+'.join(chans)}"
                assert len(ds) == int(z[key + "/len"])
                np.testing.assert_allclose(ds.data, z[key + "/data"], rtol=0, atol=1e-12)
                assert ds.data.dtype == np.float64 and (ds.labels == z[key + "/labels"]).all()
                xi, yi = ds[3]
                assert xi.dtype == torch.float32 and tuple(xi.shape) == (len(chans), 64) and yi.dtype == torch.int64
                np.testing.assert_array_equal(xi.numpy(), z[key + "/item3_x"])
                assert int(yi) == int(z[key + "/item3_y"])
        with pytest.raises(ValueError, match="Unknown classification_mode"):
            WesadDataset(td, ["S2"], ["chest_ECG"], names, classification_mode="quaternary")
        # 'amusement_binary' (main.py:183-186) raises in the reference's dataset.py; here: amusement (raw 3) -> 1, baseline (raw 1) -> 0,
        # every other window dropped AFTER the per-subject normalisation over all of the subject's windows
        raw_y = np.load(Path(td) / "S2_y.npy")
        full = WesadDataset(td, ["S2"], ["chest_ECG"], names, classification_mode="ternary")
        amu = WesadDataset(td, ["S2"], ["chest_ECG"], names, classification_mode="amusement_binary")
        keep = (raw_y == 1) | (raw_y == 3)
        assert len(amu) == int(keep.sum()) and (amu.labels == (raw_y[keep] == 3)).all()
        np.testing.assert_array_equal(amu.data, full.data[keep])
        with pytest.raises(ValueError, match="No data loaded"):
            WesadDataset(td, ["S9"], ["chest_ECG"], names)


def test_early_stopping_traces_match_reference():
    from multimodalsignal_amd.trainer import EarlyStopping
    traces = json.loads((GOLDEN / "trainer_control.json").read_text())

    class FakeModel:
        def state_dict(self):
            return {}

    n = 0
    for key, rows in traces.items():
        if not key.startswith("es/"):
            continue
        patience = int(key.rsplit("p", 1)[1])
        es = EarlyStopping(patience=patience, delta=0, checkpoint_path="unused")
        saves = []
        es.save_checkpoint = lambda model, _s=saves: _s.append(1)
        for v, counter, best, saved, stop in rows:
            before = len(saves)
            es(v, FakeModel())
            assert (es.counter, es.best_score, len(saves) > before, es.early_stop) == (counter, best, saved, stop), (key, v)
            n += 1
    assert n > 40
    # the inverted comparison: a falling loss never checkpoints after epoch 1
    rows = traces["es/falling/p3"]
    assert [r[3] for r in rows] == [True, False, False, False] and rows[-1][4] is True


def test_plateau_scheduler_on_msig_adam_matches_reference():
    from multimodalsignal_amd.models import CnnGruAttentionModel
    from multimodalsignal_amd.trainer import MsigAdam
    from torch.optim.lr_scheduler import ReduceLROnPlateau
    traces = json.loads((GOLDEN / "trainer_control.json").read_text())
    opt = MsigAdam(CnnGruAttentionModel(6, 2), lr=1e-3, weight_decay=1e-4)
    sch = ReduceLROnPlateau(opt, mode="min", factor=0.1, patience=3)
    for v, lr in traces["plateau"]:
        sch.step(v)
        assert opt.hyper["lr"] == pytest.approx(lr, rel=1e-12)


def test_metrics_match_sklearn():
    from multimodalsignal_amd.trainer import accuracy_and_weighted_f1
    for row in json.loads((GOLDEN / "metrics.json").read_text()):
        acc, f1 = accuracy_and_weighted_f1(np.array(row["y_true"]), np.array(row["y_pred"]))
        assert acc == pytest.approx(row["accuracy"], abs=1e-12) and f1 == pytest.approx(row["f1_weighted"], abs=1e-12)


def test_loso_split_table_matches_reference():
    from multimodalsignal_amd.loso import folds_for_rank, split_train_val
    from multimodalsignal_amd.main import ALL_SUBJECTS
    table = json.loads((GOLDEN / "loso_splits.json").read_text())
    assert list(table.keys()) == ALL_SUBJECTS and len(ALL_SUBJECTS) == 15
    for s in ALL_SUBJECTS:
        tr, va = split_train_val(ALL_SUBJECTS, s, 42)
        assert tr == table[s]["train"] and va == table[s]["val"] and len(tr) == 11 and len(va) == 3
    for world in (1, 2, 4, 8):
        got = sorted(k for r in range(world) for k in folds_for_rank(15, world, r))
        assert got == list(range(15))
    assert max(len(folds_for_rank(15, 8, r)) for r in range(8)) == 2


def test_model_container_matches_reference_state_dict_and_init():
    import warnings
    from multimodalsignal_amd.models import CnnGruAttentionModel
    spec = json.loads((GOLDEN / "state_dict_specs.json").read_text())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for C in (1, 2, 3, 4, 6):
            for K in (2, 3):
                m = CnnGruAttentionModel(C, K)
                ref = spec[f"C{C}_K{K}"]
                got = [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()]
                assert got == ref["state_dict"]
                assert [k for k, _ in m.named_parameters()] == ref["parameters"]
                assert sum(p.numel() for p in m.parameters()) == ref["n_params"]
        # same torch seed -> bit-identical initial weights (same initialisers, same draw order)
        z = np.load(GOLDEN / "trainer_e2e.npz", allow_pickle=False)
        torch.manual_seed(1234)
        m = CnnGruAttentionModel(in_channels=2, num_classes=2, dropout=0.0)
        for k, v in m.state_dict().items():
            np.testing.assert_array_equal(v.numpy(), z["init/" + k], err_msg=k)


def test_no_cpu_fallback_and_unsupported_configs_fail_loudly():
    from multimodalsignal_amd.models import CnnGruAttentionModel
    m = CnnGruAttentionModel(6, 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(2, 6, 256))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.engine()
    m2 = CnnGruAttentionModel(3, 2, gru_hidden_size=32, gru_num_layers=1)     # main.py:35-40's M2 variant: supported (embedded engine)
    assert m2.embedded and tuple(m2.state_dict()["gru.weight_hh_l0_reverse"].shape) == (96, 32) and "gru.weight_ih_l1" not in m2.state_dict()
    assert tuple(m2.state_dict()["classifier.0.weight"].shape) == (64, 64)
    for bad in (dict(gru_hidden_size=48), dict(gru_num_layers=3), dict(cnn_out_channels=64), dict(gru_hidden_size=32, gru_num_layers=2)):
        with pytest.raises(NotImplementedError):
            CnnGruAttentionModel(6, 2, **bad)
    with pytest.raises(ValueError):
        CnnGruAttentionModel(40, 2)


def test_synthetic_generator_writes_the_preprocess_format(tmp_path):
    from multimodalsignal_amd.synth import make_synthetic_wesad
    from multimodalsignal_amd.dataset import WesadDataset
    d = make_synthetic_wesad(tmp_path / "w", subjects=["S2", "S3"], windows_per_subject=6, T=128)
    names = (d / "_channel_names.txt").read_text().split()
    x = np.load(d / "S2_X.npy")
    assert x.shape == (6, 128, 6) and x.dtype == np.float64 and set(np.load(d / "S2_y.npy")) <= {1, 2, 3, 4}
    ds = WesadDataset(d, ["S2", "S3"], names, names)
    assert ds.data.shape == (12, 128, 6) and np.isfinite(ds.data).all()
    np.testing.assert_allclose(ds.data[:6].mean(axis=(0, 1)), 0, atol=1e-9)


def test_vectorised_dropout_keys_match_the_c_function():
    """_lib.dropout_keys (numpy, used by the lockstep trainer once per epoch) == msig_dropout_key for every step."""
    from multimodalsignal_amd import _lib as L
    for seed in (0, 42, 0x9E3779B97F4A7C15, (1 << 64) - 1, 123456789012345):
        steps = np.array([0, 1, 2, 47, 48, 1000, 2 ** 31, 2 ** 40 + 17], dtype=np.uint64)
        for stream in (1, 2):
            got = L.dropout_keys(seed, steps, stream)
            want = [L.dropout_key(seed, int(s), stream) for s in steps]
            assert [int(v) for v in got] == want, (seed, stream)


def test_loso_scaling_model_reproduces_one_gpu_and_is_bounded_by_the_longest_fold(tmp_path, capsys):
    """tools/loso_scaling_model.py (DESIGN.md section 6's prediction table): calibrated on a bench line it reproduces the 1-GPU
    wall-clock, predicts non-increasing wall-clocks for 2, 4, 8 GPUs, and never goes below the longest fold run alone."""
    import importlib.util
    import json
    import sys
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("loso_scaling_model", ROOT / "tools" / "loso_scaling_model.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    epochs = [21, 21, 21, 21, 29, 21, 55, 21, 21, 21, 30, 58, 32, 70, 47]
    line = {"b64": {"ms_per_step": 1.04}, "loso": {"wall_s": 6.44, "epochs_per_fold": epochs, "train_steps_per_epoch": 47, "fixed_s": 0.5}}
    f = tmp_path / "bench.json"
    f.write_text(json.dumps(line))
    old = sys.argv
    sys.argv = ["loso_scaling_model.py", str(f)]
    try:
        mod.main()
    finally:
        sys.argv = old
    rows = [ln.split("|") for ln in capsys.readouterr().out.splitlines() if ln.startswith("| ") and ln.split("|")[1].strip().isdigit()]
    walls = {int(r[1]): float(r[3]) for r in rows}
    assert set(walls) == {1, 2, 4, 8} and abs(walls[1] - 6.44) < 0.02
    assert walls[1] >= walls[2] >= walls[4] >= walls[8]
    alone = 0.5 + mod.rank_wall([70], 1.04, 0.0, 47, 2.0)
    assert walls[8] >= alone - 1e-9 and walls[8] < walls[1]
    # the lockstep model itself: one fold alone is epochs x (steps x s1 + eval); two folds stretch the shared epochs by (1 + k)
    assert abs(mod.rank_wall([10], 1.0, 0.1, 50, 2.0) - 10 * 52e-3) < 1e-12
    assert abs(mod.rank_wall([10, 4], 1.0, 0.1, 50, 2.0) - (4 * 52e-3 * 1.1 + 6 * 52e-3)) < 1e-12


def test_early_stopping_and_scheduler_replay_the_reference_bench_setting_runs():
    """The reference's inverted early stopping in the regime where it fires (fixture loso_parity_bench_ref.json: the reference's own
    validation-loss curves of three bench-setting folds, patience 20, budget 100): this repo's EarlyStopping fed the reference's
    curve stops at the reference's epoch and writes its last checkpoint at the reference's epoch — every run, the self check included."""
    import json
    from conftest import GOLDEN
    from multimodalsignal_amd.trainer import EarlyStopping
    fx = json.loads((GOLDEN / "loso_parity_bench_ref.json").read_text())
    runs = [(s, f) for s, f in fx["folds"].items()] + [(s, f) for s, f in ((fx.get("reference_self_check") or {}).get("folds") or {}).items()]
    assert len(runs) >= 3
    for sid, f in runs:
        es = EarlyStopping(patience=fx["training"]["patience"], delta=0)
        saved = []
        es.save_checkpoint = lambda model, _s=saved: _s.append(1)
        stop, ck = None, 0
        for ep, v in enumerate(f["val"], 1):
            n = len(saved)
            es(v[0], None)
            if len(saved) > n:
                ck = ep
            if es.early_stop:
                stop = ep
                break
        assert (stop or fx["training"]["epochs"]) == f["epochs"], (sid, stop, f["epochs"])
        assert ck == f["checkpoint_epoch"], (sid, ck, f["checkpoint_epoch"])

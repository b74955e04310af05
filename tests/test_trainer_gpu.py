"""GPU tier: the host-side drop-ins (models.py / trainer.py / dataset.py / main.py) on the
HIP path, against the reference's recorded Trainer run and the stock-torch CPU model."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden_model

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _write_subjects(z, td):
    for k in z.files:
        if k.startswith("raw/"):
            np.save(Path(td) / f"{k[4:]}.npy", z[k])


def test_trainer_reproduces_reference_run(tmp_path):
    """Full-batch, dropout-free 6-epoch run recorded from the reference Trainer (CPU torch):
    same initial weights, same windows -> same per-epoch validation loss/acc/F1 and test metrics."""
    from multimodalsignal_amd.dataset import DeviceLoader, WesadDataset
    from multimodalsignal_amd.models import CnnGruAttentionModel
    from multimodalsignal_amd.trainer import Trainer
    z = np.load(GOLDEN / "trainer_e2e.npz", allow_pickle=False)
    _write_subjects(z, tmp_path)
    names = ["chest_ECG", "chest_EDA"]
    mk = lambda s: WesadDataset(tmp_path, s, names, names, classification_mode="stress_binary")
    tr, va, te = mk(["S2", "S3", "S4"]), mk(["S5"]), mk(["S6"])
    torch.manual_seed(1234)
    model = CnnGruAttentionModel(in_channels=2, num_classes=2, dropout=0.0)
    cfg = {"trainer": {"epochs": 6, "learning_rate": 1e-3, "early_stopping": {"enabled": True, "patience": 20, "delta": 0},
                       "weight_decay": 1e-4, "verbose": False}}
    t = Trainer(model, tmp_path / "fold", cfg)
    t.train(DeviceLoader(tr, 64, True, DEV, seed=1), DeviceLoader(va, 64, False, DEV))
    got = np.array([[h["val_loss"], h["val_acc"], h["val_f1"]] for h in t.history])
    ref = z["val_epochs"]
    np.testing.assert_allclose(got[:, 0], ref[:, 0], rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(got[:, 1:], ref[:, 1:], atol=1e-9)
    np.testing.assert_allclose([h["train_loss"] for h in t.history], z["train_loss_4dp"], atol=2e-3)
    loss, acc, f1 = t.evaluate(DeviceLoader(te, 64, False, DEV), is_test=True)
    assert abs(loss - z["test"][0]) < 2e-3 and acc == pytest.approx(z["test"][1]) and f1 == pytest.approx(z["test"][2])
    assert (tmp_path / "fold" / "best_model.pt").exists() == bool(z["ckpt_exists"])
    assert (tmp_path / "fold" / "training_log.txt").read_text().count("Epoch ") == 6
    # torch's stock DataLoader (what the reference's main.py builds) is accepted too
    from torch.utils.data import DataLoader
    l2, a2, f2 = t.evaluate(DataLoader(te, batch_size=4, shuffle=False))
    assert abs(l2 - loss) < 1e-5 and a2 == acc and f2 == f1
    # final weights close to the reference's
    sd = model.state_dict()
    for k in ("classifier.3.weight", "gru.weight_hh_l1", "cnn_encoder.0.weight", "cnn_encoder.5.running_var"):
        np.testing.assert_allclose(sd[k].cpu().numpy(), z["final/" + k], rtol=5e-2, atol=6e-3, err_msg=k)


def test_autograd_path_equals_fused_step():
    """model(x) -> torch CrossEntropyLoss -> loss.backward() -> optimizer.step() (the reference's
    literal loop, trainer.py:144-149) gives the same update as msig_train_step."""
    from multimodalsignal_amd.models import CnnGruAttentionModel
    from multimodalsignal_amd.trainer import MsigAdam
    meta, params_np, g = load_golden_model("model_c6_k2_t512")
    sd = {k: torch.as_tensor(v) for k, v in params_np.items()}
    x, y = torch.as_tensor(g["x"]).to(DEV), torch.as_tensor(g["y"]).to(DEV)
    ma = CnnGruAttentionModel(6, 2, dropout=0.0); ma.load_state_dict(sd); ma.to(DEV).train()
    mb = CnnGruAttentionModel(6, 2, dropout=0.0); mb.load_state_dict(sd); mb.to(DEV).train()
    opt = MsigAdam(ma, lr=1e-3, weight_decay=1e-4)
    crit = torch.nn.CrossEntropyLoss()
    eb = mb.engine()
    for step in (1, 2):
        opt.zero_grad()
        logits = ma(x)
        loss = crit(logits, y)
        loss.backward()
        opt.step()
        eb.train_step(x, y, lr=1e-3, weight_decay=1e-4, step=step)
        torch.cuda.synchronize()
        assert abs(float(loss) - float(eb.region("LOSS")[0])) < 1e-6
        assert abs(float(loss) - float(g[f"loss_step{step}"])) < 1e-4
    for (ka, va), (kb, vb) in zip(ma.state_dict().items(), mb.state_dict().items()):
        assert ka == kb
        np.testing.assert_allclose(va.cpu().numpy(), vb.cpu().numpy(), rtol=1e-4, atol=3e-6, err_msg=ka)   # torch CE vs ce_kernel: 1-ulp dlogits, amplified by Adam at |g|~eps
    # torch's own Adam on the (flat-buffer-view) parameters also works: a drop-in must not care
    mc = CnnGruAttentionModel(6, 2, dropout=0.0); mc.load_state_dict(sd); mc.to(DEV).train()
    topt = torch.optim.Adam(mc.parameters(), lr=1e-3, weight_decay=1e-4)
    mc.engine()
    topt.zero_grad(); crit(mc(x), y).backward(); topt.step()
    np.testing.assert_allclose(mc.state_dict()["classifier.0.weight"].cpu().numpy(), g["after1/classifier.0.weight"], rtol=1e-3, atol=2e-5)
    # eval forward through nn.Module.__call__
    ma.eval()
    with torch.no_grad():
        out = ma(x)
    assert out.shape == (5, 2) and torch.isfinite(out).all()


def test_checkpoints_interchange_with_reference_module_graph(tmp_path):
    from multimodalsignal_amd.models import CnnGruAttentionModel
    from oracle.cpu_model import CpuCnnGru
    meta, params_np, g = load_golden_model("model_c6_k2_t512")
    m = CnnGruAttentionModel(6, 2)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in params_np.items()})
    m.to(DEV).eval()
    with torch.no_grad():
        mine = m(torch.as_tensor(g["x"]).to(DEV)).cpu().numpy()
    np.testing.assert_allclose(mine, g["eval_logits"], rtol=2e-4, atol=5e-5)
    torch.save(m.state_dict(), tmp_path / "best_model.pt")                 # trainer.py:38-39
    ref = CpuCnnGru(6, 2)
    ref.load_state_dict(torch.load(tmp_path / "best_model.pt", weights_only=True, map_location="cpu"))   # trainer.py:187
    ref.eval()
    with torch.no_grad():
        np.testing.assert_allclose(ref(torch.as_tensor(g["x"])).numpy(), mine, rtol=2e-4, atol=5e-5)
    m2 = CnnGruAttentionModel(6, 2).to(DEV)
    m2.load_state_dict(torch.load(tmp_path / "best_model.pt", weights_only=True))
    m2.eval()
    with torch.no_grad():
        np.testing.assert_array_equal(m2(torch.as_tensor(g["x"]).to(DEV)).cpu().numpy(), mine)


def test_device_loader_covers_every_window_once(tmp_path):
    from multimodalsignal_amd.dataset import DeviceLoader, WesadDataset
    from multimodalsignal_amd.synth import make_synthetic_wesad
    d = make_synthetic_wesad(tmp_path / "w", subjects=["S2", "S3"], windows_per_subject=37, T=128)
    names = (d / "_channel_names.txt").read_text().split()
    ds = WesadDataset(d, ["S2", "S3"], names, names)
    ld = DeviceLoader(ds, 16, True, DEV, seed=5)
    seen, ys = [], []
    for xb, yb in ld:
        seen.append(xb.clone()); ys.append(yb.clone())
    X = torch.cat(seen).cpu().numpy(); Y = torch.cat(ys).cpu().numpy()
    assert X.shape == (74, 6, 128) and len(ld) == 5
    ref = ds.data.transpose(0, 2, 1).astype(np.float32)
    order = [int(np.argmin(np.abs(ref - X[i]).reshape(74, -1).sum(1))) for i in range(74)]
    assert sorted(order) == list(range(74)) and order != list(range(74))
    assert (ds.labels[order] == Y).all()
    first = torch.cat([xb.clone() for xb, _ in DeviceLoader(ds, 16, False, DEV)]).cpu().numpy()
    np.testing.assert_array_equal(first, ref)


def test_loso_driver_small_synthetic_run(tmp_path):
    from multimodalsignal_amd import main as M
    from multimodalsignal_amd.synth import make_synthetic_wesad, CHANNELS6
    subs = ["S2", "S3", "S4", "S5", "S6"]
    d = make_synthetic_wesad(tmp_path / "w", subjects=subs, windows_per_subject=24, T=256)
    cfg = M.default_cfg()
    cfg.update(data_path=d, channels=list(CHANNELS6), subjects=subs, epochs=3, patience=20, batch_size=16)
    names = (d / "_channel_names.txt").read_text().split()
    results, wall = M.run_simple_experiment(tmp_path / "run", DEV, names, cfg)
    assert [r["subject"] for r in results] == subs and all(0.0 <= r["accuracy"] <= 1.0 for r in results)
    txt = (tmp_path / "run" / "cv_summary.txt").read_text(encoding="utf-8")
    assert "平均准确率" in txt and txt.count("测试 S") == 5
    for s in subs:
        fd = tmp_path / "run" / f"fold_test_on_{s}"
        assert (fd / "training_log.txt").exists() and (fd / "best_model.pt").exists() and (fd / "fold_result.json").exists()


def test_concurrent_folds_equal_sequential(tmp_path):
    """Folds trained concurrently on separate HIP streams give exactly the sequential results
    (seeding / initialisation stay sequential; kernels have no atomics)."""
    from multimodalsignal_amd import main as M
    from multimodalsignal_amd.synth import make_synthetic_wesad, CHANNELS6
    subs = ["S2", "S3", "S4", "S5"]
    d = make_synthetic_wesad(tmp_path / "w", subjects=subs, windows_per_subject=20, T=256, difficulty=2.0)
    names = (d / "_channel_names.txt").read_text().split()
    out = {}
    for conc in (1, 4):
        cfg = M.default_cfg()
        cfg.update(data_path=d, channels=list(CHANNELS6), subjects=subs, epochs=3, patience=20, batch_size=16, concurrent_folds=conc,
                   lockstep=False)
        results, _ = M.run_simple_experiment(tmp_path / f"run{conc}", DEV, names, cfg)
        out[conc] = [(r["subject"], r["accuracy"], r["f1_score"]) for r in results]
        logs = [(tmp_path / f"run{conc}" / f"fold_test_on_{s}" / "training_log.txt").read_text() for s in subs]
        out[(conc, "loss")] = [ln.split("训练损失: ")[1].split(" |")[0] for lg in logs for ln in lg.splitlines() if "训练损失" in ln]
    assert out[1] == out[4]
    assert out[(1, "loss")] == out[(4, "loss")] and len(out[(1, "loss")]) == 12


@pytest.mark.parametrize("forms,spread", [(("auto", "auto"), 0), (("ws", "b3"), 0), (("auto", "auto"), 6), (("ws", "b4"), 6)])
def test_lockstep_folds_equal_sequential(tmp_path, forms, spread):
    """Folds trained in LOCKSTEP as one fold batch (msig_train_step_multi / msig_forward_multi / msig_gather_windows_multi: every
    launch covers all folds, per-fold arenas, blockIdx.z = fold) give exactly the sequential per-fold results — metrics, per-epoch
    training / validation numbers as logged, early-stopping epochs, checkpointed weights — including folds that stop early and
    leave the batch while others continue (per-fold patience 1..4 here), a ragged last batch, dropout and the LR schedule.
    forms = ("ws", "b3"): the throughput-form GRU kernels, which a fold batch selects by itself from 12 tiles over the launch's folds on
    (their FOLDS instantiations: gru_fwd_ws<.., true>, gru_bwd_b3<.., true>), against the same forms run fold by fold.
    spread > 0: subjects of UNEQUAL size (37 +- 6 windows, as real WESAD's differ) — the folds' train / val sets then differ in
    size, take different numbers of steps per epoch (per-fold Adam step counts, msig_multi.step) and end on ragged batches of
    different sizes, which run as launches over the folds whose batch sizes agree (multifold.launch_plan)."""
    from multimodalsignal_amd import _lib as L
    from multimodalsignal_amd import main as M
    from multimodalsignal_amd.dataset import SubjectStore
    from multimodalsignal_amd.multifold import LockstepTrainer, lockstep_compatible
    from multimodalsignal_amd.synth import make_synthetic_wesad, CHANNELS6
    subs = ["S2", "S3", "S4", "S5", "S6"]
    d = make_synthetic_wesad(tmp_path / "w", subjects=subs, windows_per_subject=37, T=320, difficulty=4.0, window_spread=spread)
    names = (d / "_channel_names.txt").read_text().split()
    out = {}
    if spread:
        sizes = [len(np.load(d / f"{s_}_y.npy")) for s_ in subs]
        assert len(set(sizes)) > 2, sizes
    L.set_kernel_form(*forms)
    for mode in ("seq", "lock"):
        base = M.default_cfg()
        base.update(data_path=d, channels=list(CHANNELS6), subjects=subs, epochs=12, batch_size=16)
        store = SubjectStore(d, subs, base["channels"], names, classification_mode=base["mode"], device=DEV)
        preps = [M.prepare_fold(k, sid, tmp_path / mode, DEV, names, dict(base, patience=1 + k % 4), store) for k, sid in enumerate(subs)]
        if mode == "seq":
            infos = [M.train_fold(p, DEV) for p in preps]
        else:
            assert lockstep_compatible(preps)
            infos = LockstepTrainer(preps, DEV).run()
        out[mode] = [(i["subject"], i["accuracy"], i["f1_score"], i["epochs"]) for i in infos]
        logs = [(tmp_path / mode / f"fold_test_on_{s}" / "training_log.txt").read_text() for s in subs]
        out[mode, "epochs"] = [[ln.split(" | 耗时")[0] + ln.split("s |", 1)[1].rsplit(" | ", 1)[0] for ln in lg.splitlines() if "训练损失" in ln] for lg in logs]
        out[mode, "w"] = [torch.load(tmp_path / mode / f"fold_test_on_{s}" / "best_model.pt", weights_only=True) for s in subs]
    L.set_kernel_form("auto", "auto")
    assert out["seq"] == out["lock"]
    assert out["seq", "epochs"] == out["lock", "epochs"]
    assert len({i[3] for i in out["seq"]}) > 1, "the folds should stop at different epochs for this test to bite"
    for a, b in zip(out["seq", "w"], out["lock", "w"]):
        for k in a:
            assert torch.equal(a[k], b[k]), k


def test_fold_results_do_not_depend_on_grouping(tmp_path):
    """at the reference's B = 64 (4 tiles) a fold batch of >= 3 folds used to train on the throughput-form GRU
    kernels and to switch to the latency forms once early stopping left it fewer — a fold's bits depended on its companions.  The
    form is now pinned per run (FoldArena.multi: msig_multi.form_folds = 1): one fold batch of six, three batches of two, and six
    stand-alone folds give identical metrics, stop epochs and checkpoints, with folds dropping across the old 12-tile threshold."""
    from multimodalsignal_amd import main as M
    from multimodalsignal_amd.synth import make_synthetic_wesad, CHANNELS6
    subs = ["S2", "S3", "S4", "S5", "S6", "S7"]
    d = make_synthetic_wesad(tmp_path / "w", subjects=subs, windows_per_subject=40, T=256, difficulty=4.0, window_spread=4)
    names = (d / "_channel_names.txt").read_text().split()
    out = {}
    for tag, kw in (("one", dict(lockstep_groups=1)), ("three", dict(lockstep_groups=3)), ("seq", dict(concurrent_folds=1))):
        cfg = M.default_cfg()
        cfg.update(data_path=d, channels=list(CHANNELS6), subjects=subs, epochs=8, patience=[1, 2, 3, 4], batch_size=64, **kw)
        M.run_simple_experiment(tmp_path / tag, DEV, names, cfg)
        infos = [json.loads((tmp_path / tag / f"fold_test_on_{s}" / "fold_result.json").read_text()) for s in subs]
        out[tag] = [(i["subject"], i["accuracy"], i["f1_score"], i["epochs"]) for i in infos]
        out[tag, "w"] = [torch.load(tmp_path / tag / f"fold_test_on_{s}" / "best_model.pt", weights_only=True) for s in subs]
    assert out["one"] == out["three"] == out["seq"]
    assert len({i[3] for i in out["one"]}) > 1, "the folds should stop at different epochs for this test to bite"
    for tag in ("three", "seq"):
        for a, b in zip(out["one", "w"], out[tag, "w"]):
            for k in a:
                assert torch.equal(a[k], b[k]), (tag, k)


def test_hierarchical_experiment_small_synthetic_run(tmp_path):
    """The reference's run_hierarchical_experiment (main.py:159-247): per fold M1 (stress vs rest, reference configuration) and M2
    (amusement vs baseline, gru_hidden_size 32 / gru_num_layers 1 on the embedded engine), then the three-class decision."""
    from multimodalsignal_amd import main as M
    from multimodalsignal_amd.synth import make_synthetic_wesad
    subs = ["S2", "S3", "S4", "S5", "S6"]
    d = make_synthetic_wesad(tmp_path / "w", subjects=subs, windows_per_subject=40, T=256, difficulty=2.0)
    names = (d / "_channel_names.txt").read_text().split()
    cfg = M.default_cfg()
    cfg.update(data_path=d, subjects=subs, epochs=2, patience=20, batch_size=16)
    results, wall = M.run_hierarchical_experiment(tmp_path / "run", DEV, names, cfg)
    assert [r["subject"] for r in results] == subs
    assert all(0.0 <= r["m1_accuracy"] <= 1.0 and 0.0 <= r["ternary_accuracy"] <= 1.0 for r in results)
    txt = (tmp_path / "run" / "hierarchical_summary.txt").read_text(encoding="utf-8")
    assert txt.count("测试 S") == 5 and "平均三分类准确率" in txt
    for s_ in subs:
        fd = tmp_path / "run" / f"fold_test_on_{s_}"
        assert (fd / "model_m1" / "best_model.pt").exists() and (fd / "model_m2" / "best_model.pt").exists()
        sd2 = torch.load(fd / "model_m2" / "best_model.pt", weights_only=True)
        assert tuple(sd2["gru.weight_hh_l0"].shape) == (96, 32) and "gru.weight_ih_l1" not in sd2       # the reference's M2 state_dict
        row = json.loads((fd / "fold_result.json").read_text())
        assert row["n"] > 0 and 0 <= row["correct"] <= row["n"]


def test_ablation_sweep_equals_separate_runs(tmp_path):
    """The channel-ablation sweep (BASELINE.json config 4: sets x folds as one sharded job) gives, per set,
    exactly what a separate LOSO run with those channels gives, and writes one cv_summary.txt per set."""
    from multimodalsignal_amd import main as M
    from multimodalsignal_amd.synth import make_synthetic_wesad
    subs = ["S2", "S3", "S4", "S5"]
    d = make_synthetic_wesad(tmp_path / "w", subjects=subs, windows_per_subject=20, T=256, difficulty=2.0)
    names = (d / "_channel_names.txt").read_text().split()
    sets = M.ablation_sets(names)
    assert list(sets) == ["ecg_only", "eda_only", "chest_only", "wrist_only"]
    assert sets["ecg_only"] == ["chest_ECG"] and sets["wrist_only"] == ["wrist_BVP", "wrist_EDA"] and len(sets["chest_only"]) == 4
    base = M.default_cfg()
    base.update(data_path=d, subjects=subs, epochs=2, patience=20, batch_size=16, concurrent_folds=8)
    sweep, _ = M.run_experiments(tmp_path / "sweep", DEV, names, {n: dict(base, channels=ch) for n, ch in sets.items()})
    for n, ch in sets.items():
        assert (tmp_path / "sweep" / n / "cv_summary.txt").exists()
        assert (tmp_path / "sweep" / n / "fold_test_on_S3" / "best_model.pt").exists()
        single, _ = M.run_simple_experiment(tmp_path / f"single_{n}", DEV, names, dict(base, channels=ch, concurrent_folds=1))
        assert [(r["subject"], r["accuracy"], r["f1_score"]) for r in sweep[n]] == \
               [(r["subject"], r["accuracy"], r["f1_score"]) for r in single], n


def test_subject_store_matches_wesad_dataset(tmp_path):
    """One HBM-resident store for all subjects == the per-fold WesadDataset path, window for window;
    on-device normalisation agrees with the reference's float64 numpy arithmetic to fp32 rounding."""
    from multimodalsignal_amd.dataset import DeviceLoader, SubjectStore, WesadDataset
    from multimodalsignal_amd.synth import make_synthetic_wesad
    subs = ["S2", "S3", "S4"]
    d = make_synthetic_wesad(tmp_path / "w", subjects=subs, windows_per_subject=9, T=128, difficulty=2.0)
    names = (d / "_channel_names.txt").read_text().split()
    use = ["chest_EDA", "chest_ECG", "wrist_EDA"]          # includes the log1p channel, in a non-native order
    store = SubjectStore(d, subs + ["S9"], use, names, device=DEV)     # S9 missing: skipped with a warning
    ref = WesadDataset(d, ["S4", "S2"], use, names)
    view = store.view(["S4", "S2"])
    assert len(view) == len(ref) == 18 and (view.labels == ref.labels).all()
    got = torch.cat([xb.clone() for xb, _ in DeviceLoader(view, 5, False, DEV)]).cpu().numpy()
    want = torch.cat([xb.clone() for xb, _ in DeviceLoader(ref, 5, False, DEV)]).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(got, ref.data.transpose(0, 2, 1).astype(np.float32))
    xi, yi = view[3]
    np.testing.assert_array_equal(xi.numpy(), ref[3][0].numpy())
    dev_store = SubjectStore(d, subs, use, names, device=DEV, normalise="device")
    np.testing.assert_allclose(dev_store.x.cpu().numpy(), store.x.cpu().numpy(), rtol=0, atol=2e-6)
    with pytest.raises(ValueError, match="No data loaded"):
        store.view(["S9"])


def test_labels_outside_the_class_range_raise(tmp_path):
    """CLASSIFICATION_MODE 'ternary' (labels 0..2) with a 2-class model: torch's CrossEntropyLoss raises (trainer.py:147);
    here the loss kernel would index past the logits row, so Trainer refuses the dataset up front."""
    from multimodalsignal_amd.dataset import DeviceLoader, WesadDataset
    from multimodalsignal_amd.models import CnnGruAttentionModel
    from multimodalsignal_amd.trainer import Trainer
    z = np.load(GOLDEN / "trainer_e2e.npz", allow_pickle=False)
    _write_subjects(z, tmp_path)
    names = ["chest_ECG", "chest_EDA"]
    ds = WesadDataset(tmp_path, ["S2", "S3"], names, names, classification_mode="ternary")
    assert ds.labels.max() == 2
    cfg = {"trainer": {"epochs": 1, "learning_rate": 1e-3, "early_stopping": {"enabled": False, "patience": 1, "delta": 0},
                       "weight_decay": 0.0, "verbose": False}}
    t = Trainer(CnnGruAttentionModel(in_channels=2, num_classes=2), tmp_path / "fold", cfg)
    with pytest.raises(ValueError, match="outside"):
        t.train(DeviceLoader(ds, 16, False, DEV), DeviceLoader(ds, 16, False, DEV))
    with pytest.raises(ValueError, match="outside"):
        t.evaluate(DeviceLoader(ds, 16, False, DEV))
    t3 = Trainer(CnnGruAttentionModel(in_channels=2, num_classes=3), tmp_path / "fold3", cfg)       # the matching model trains
    t3.train(DeviceLoader(ds, 16, False, DEV), DeviceLoader(ds, 16, False, DEV))


def test_ragged_batch_shares_the_full_batch_workspace():
    """One workspace allocation per mode: the ragged last batch of an epoch lays its regions out in the full batch's buffer, a
    smaller-then-larger sequence grows it, and the step's result does not depend on what shared the buffer before."""
    from multimodalsignal_amd.models import CnnGruAttentionModel
    meta, params_np, g = load_golden_model("model_c6_k2_t512")
    sd = {k: torch.as_tensor(v) for k, v in params_np.items()}
    x, y = torch.as_tensor(g["x"]).to(DEV), torch.as_tensor(g["y"]).to(DEV)
    ma = CnnGruAttentionModel(6, 2, dropout=0.0); ma.load_state_dict(sd); ma.to(DEV).train()
    mb = CnnGruAttentionModel(6, 2, dropout=0.0); mb.load_state_dict(sd); mb.to(DEV).train()
    ea, eb = ma.engine(), mb.engine()
    B, T = x.shape[0], x.shape[2]
    full, _ = ea.workspace(B, T, True)
    ragged, _ = ea.workspace(B - 2, T, True)
    assert ragged.data_ptr() == full.data_ptr() and ragged.numel() < full.numel()
    assert ea.workspace(B, T, False)[0].data_ptr() != full.data_ptr()          # evaluation never shares the training buffer
    ea.train_step(x[:B - 2], y[:B - 2], lr=1e-3, weight_decay=1e-4, step=1)    # ragged first ...
    ea.train_step(x, y, lr=1e-3, weight_decay=1e-4, step=2)                    # ... then the full batch in the same buffer
    eb.workspace(B - 2, T, True)
    eb.train_step(x[:B - 2], y[:B - 2], lr=1e-3, weight_decay=1e-4, step=1)    # here the buffer has to grow between the steps
    small_ptr = eb.workspace(B - 2, T, True)[0].data_ptr()
    eb.train_step(x, y, lr=1e-3, weight_decay=1e-4, step=2)
    assert eb.workspace(B - 2, T, True)[0].data_ptr() == eb.workspace(B, T, True)[0].data_ptr()
    torch.cuda.synchronize()
    del small_ptr
    for (ka, va), (kb, vb) in zip(ma.state_dict().items(), mb.state_dict().items()):
        assert torch.equal(va, vb), ka

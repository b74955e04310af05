"""GPU tier, north_star's accuracy criterion: per-fold accuracy of the HIP path against the REFERENCE's
own CPU runs on identical windows, at the reference's defaults (B = 64, dropout 0.5, shuffled batches, Adam), over several
seeds per side — one seed cannot separate a few pp of per-fold difference from the spread of the random streams, which
differ by construction (torch's Philox dropout / DataLoader shuffling vs the counter-based masks and the device shuffler).

Fixture: tests/golden/loso_parity_ref.json (tests/golden/make_parity_fixture.py; the reference imported in the build
container).  The synthetic dataset is regenerated here from the generator arguments the fixture records."""
import json
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(1500)
def test_loso_accuracy_matches_reference_over_seeds(tmp_path_factory):
    sys.path.insert(0, str(ROOT / "tools"))
    import loso_parity as LP
    fx = json.loads((GOLDEN / "loso_parity_ref.json").read_text())
    ds, tr = fx["dataset"], fx["training"]
    data = Path("/tmp") / f"msig_parity_w{ds['windows_per_subject']}_t{ds['T']}_d{ds['difficulty']}"
    LP.ensure_data(data, ds["windows_per_subject"], ds["T"], ds["difficulty"])
    out = tmp_path_factory.mktemp("parity")
    folds = list(fx["runs"][0]["folds"])
    ref = {f: np.array([r["folds"][f]["acc"] for r in fx["runs"]]) for f in folds}
    got = {f: [] for f in folds}
    got_f1s = []
    for run in fx["runs"]:
        res = LP.run_side("gpu", data, out, run["seed_base"], folds, tr["epochs"], tr["batch"], tr["dropout"], log=lambda *a: None)
        for r in res:
            got[r["subject"]].append(r["acc"])
            got_f1s.append(r["f1"])
    got = {f: np.array(v) for f, v in got.items()}
    ref_f1 = float(np.mean([r["folds"][f]["f1"] for r in fx["runs"] for f in folds]))
    n = len(fx["runs"])
    ref_mean, got_mean = float(np.mean([ref[f].mean() for f in folds])), float(np.mean([got[f].mean() for f in folds]))
    print(f"\nmean LOSO accuracy over {n} seeds x {len(folds)} folds: reference {ref_mean:.4f}  HIP {got_mean:.4f}  (diff {100 * (got_mean - ref_mean):+.2f} pp)")
    for f in folds:
        print(f"  {f}: reference {ref[f].mean():.3f} [{ref[f].min():.2f} .. {ref[f].max():.2f}]   HIP {got[f].mean():.3f} [{got[f].min():.2f} .. {got[f].max():.2f}]")
    got_f1 = float(np.mean(got_f1s))
    print(f"mean weighted F1: reference {ref_f1:.4f}  HIP {got_f1:.4f}  (diff {100 * (got_f1 - ref_f1):+.2f} pp)")
    # north_star: mean LOSO accuracy — and weighted F1 — within +-0.5 pp of the reference's CPU run on identical windows
    assert abs(got_mean - ref_mean) <= 0.005, (got_mean, ref_mean)
    assert abs(got_f1 - ref_f1) <= 0.005, (got_f1, ref_f1)
    # and every fold's mean inside the interval the reference's own seeds span (widened by one test-window: 1 / n_test)
    slack = 1.0 / ds["windows_per_subject"]
    for f in folds:
        assert ref[f].min() - slack <= got[f].mean() <= ref[f].max() + slack, (f, got[f].mean(), ref[f].min(), ref[f].max())


@pytest.mark.timeout(900)
def test_deterministic_loso_matches_reference_fold_by_fold(tmp_path):
    """per-fold accuracy AND weighted F1 of the SHIPPED LOSO path — `run_experiments`, LockstepTrainer, default
    fold grouping, pinned kernel forms — against the reference's own CPU run (fixture: tests/golden/loso_parity_det_ref.json, made by
    tests/golden/make_parity_fixture.py --deterministic from the imported reference) in the deterministic setting: dropout 0,
    unshuffled batches of 64, all 15 folds, 20 epochs, bit-identical initial weights (fold k seeded 42 + k on both sides).  Both sides
    then run the same arithmetic on the same batches; what differs is fp32 rounding (summation order, split-bf16 MFMA vs torch's CPU
    GEMMs), which 360 Adam steps amplify — so the validation-loss curves are compared with a tolerance that grows with the epoch, and
    the test metrics to within one window of the 100-window test subject."""
    from multimodalsignal_amd import main as M
    from multimodalsignal_amd.synth import ALL_SUBJECTS, CHANNELS6, make_synthetic_wesad
    fx = json.loads((GOLDEN / "loso_parity_det_ref.json").read_text())
    ds, tr = fx["dataset"], fx["training"]
    data = Path("/tmp") / f"msig_parity_w{ds['windows_per_subject']}_t{ds['T']}_d{ds['difficulty']}"
    if not (data / "_channel_names.txt").exists():
        make_synthetic_wesad(data, windows_per_subject=ds["windows_per_subject"], T=ds["T"], difficulty=ds["difficulty"])
    names = (data / "_channel_names.txt").read_text().split()
    dev = torch.device("cuda:0")
    cfg = M.default_cfg()
    cfg.update(data_path=data, channels=list(CHANNELS6), epochs=tr["epochs"], batch_size=tr["batch"], patience=tr["patience"], lr=tr["lr"],
               weight_decay=tr["weight_decay"], seed=tr["seed_base"], shuffle=False, subjects=list(ALL_SUBJECTS),
               model_params=dict(cnn_out_channels=32, gru_hidden_size=64, gru_num_layers=2, dropout=0.0))
    assert cfg["concurrent_folds"] == 15 and cfg.get("lockstep", True)          # the shipped path: lockstep fold batches, default groups
    results, wall = M.run_simple_experiment(tmp_path, dev, names, cfg)
    assert [r["subject"] for r in results] == list(fx["folds"]) == list(ALL_SUBJECTS)
    n_test = ds["windows_per_subject"]
    worst = dict(acc=0.0, f1=0.0, val=0.0)
    rows, all_dv = [], []
    for r in results:
        ref = fx["folds"][r["subject"]]
        hist = json.loads((tmp_path / f"fold_test_on_{r['subject']}" / "fold_result.json").read_text())["history"]
        assert len(hist) == len(ref["val"]) == tr["epochs"], (r["subject"], len(hist))       # patience 20 over 20 epochs: no early stop on either side
        dv = [abs(h["val_loss"] - v[0]) for h, v in zip(hist, ref["val"])]
        da = [abs(h["val_acc"] - v[1]) for h, v in zip(hist, ref["val"])]
        all_dv.append(dv)
        rows.append((r["subject"], r["accuracy"], ref["acc"], r["f1_score"], ref["f1"], max(dv), dv[0], max(da)))
        worst["acc"] = max(worst["acc"], abs(r["accuracy"] - ref["acc"]))
        worst["f1"] = max(worst["f1"], abs(r["f1_score"] - ref["f1"]))
        worst["val"] = max(worst["val"], max(dv))
    print(f"\n15-fold deterministic LOSO on the HIP path in {wall:.1f} s (reference CPU: {sum(f['seconds_cpu'] for f in fx['folds'].values()):.0f} s)")
    for row in rows:
        print("  %-4s acc %.4f (ref %.4f)  f1 %.4f (ref %.4f)  max |val loss diff| %.2e (epoch 1: %.2e)  max |val acc diff| %.4f" % row)
    per_epoch = np.max(np.array(all_dv), axis=0)
    print("  worst |val loss diff| over the folds, per epoch: " + " ".join(f"{v:.1e}" for v in per_epoch))
    mean_acc, mean_f1 = float(np.mean([r["accuracy"] for r in results])), float(np.mean([r["f1_score"] for r in results]))
    print(f"  mean acc {mean_acc:.4f} (ref {fx['summary']['mean_acc']:.4f})  mean F1 {mean_f1:.4f} (ref {fx['summary']['mean_f1']:.4f})  worst {worst}")
    # the yardstick: the reference against itself with another thread count (same code, another summation order)
    sc = fx.get("reference_self_check")
    if sc:
        for sid, f in sc["folds"].items():
            own = [abs(a[0] - b[0]) for a, b in zip(f["val"], fx["folds"][sid]["val"])]
            print(f"  reference with {sc['threads']} thread(s) vs the fixture's run, {sid}: acc {f['acc']:.2f} vs {fx['folds'][sid]['acc']:.2f}, "
                  f"max |val loss diff| {max(own):.1e} (epoch 1: {own[0]:.1e})")
        self_worst = max(max(abs(a[0] - b[0]) for a, b in zip(f["val"], fx["folds"][sid]["val"])) for sid, f in sc["folds"].items())
        assert worst["val"] <= 1.5 * self_worst, (worst["val"], self_worst)       # the HIP path is no further from the reference than the reference from itself
    # (1) the curves: epoch 1 agrees to fp32 noise (observed 1e-8 ... 7e-6 over the folds), and the gap may grow no faster than two
    #     fp32 trajectories of the same training run separate anyway (observed worst over the folds, per epoch: 6.7e-6 1.4e-5 8.5e-5
    #     2.6e-4 1.2e-3 5.4e-3 5.7e-3 1.2e-2 2.5e-2, then a plateau at 2.2e-2: a factor ~4 per epoch — 18 Adam steps — until the
    #     trajectories are as far apart as two fp32 runs of one training get; per fold the largest gap over the 20 epochs is
    #     5e-6 ... 2.5e-2, median 5e-3).  A wrong kernel shows in epoch 1, not in epoch 9.
    limit = np.minimum(DET_VAL_CAP, DET_VAL_TOL_EPOCH1 * DET_VAL_GROWTH ** np.arange(tr["epochs"]))
    assert (per_epoch <= limit).all(), (per_epoch, limit)
    # (2) the metrics of every fold: within DET_WINDOWS windows of its 100-window test subject, most folds identical
    same = 0
    for sid, acc, racc, f1, rf1, dvmax, dv0, damax in rows:
        assert abs(acc - racc) <= DET_WINDOWS / n_test + 1e-9, (sid, acc, racc)
        assert abs(f1 - rf1) <= DET_F1_TOL, (sid, f1, rf1)
        same += int(round(acc * n_test) == round(racc * n_test) and abs(f1 - rf1) < 1e-9)
    assert same >= DET_SAME_FOLDS, same
    # (3) north_star: within 0.5 pp of the reference's CPU run, accuracy and F1 (observed: equal to four digits, 0.7427 / 0.7253)
    assert abs(mean_acc - fx["summary"]["mean_acc"]) <= 0.005 and abs(mean_f1 - fx["summary"]["mean_f1"]) <= 0.005


# tolerances of the deterministic comparison (set from the observed run, see the test's printout in DESIGN.md section 2)
DET_VAL_TOL_EPOCH1 = 1.5e-5      # epoch 1: observed <= 6.7e-6 (14 of 15 folds <= 7.4e-8)
DET_VAL_GROWTH = 4.0             # allowed growth of the gap per epoch (observed ~4 until saturation)
DET_VAL_CAP = 4e-2               # saturation (observed 2.5e-2)
DET_WINDOWS = 3                  # test accuracy within 3 of 100 windows (observed: 10 folds identical, four off by 1, one by 3)
DET_F1_TOL = 0.03
DET_SAME_FOLDS = 8               # folds whose test accuracy AND weighted F1 equal the reference's exactly (observed 10 of 15)


# ---- the bench's own LOSO regime: early stopping fires (round 5) ------------------------------------------------
BENCH_VAL_TOL_EPOCH1 = 3e-5       # validation loss after epoch 1 (48 Adam steps): observed 2e-9 (S5), 1.2e-5 (S3)
BENCH_VAL_GROWTH = 16.0           # allowed growth of the gap per epoch of 48 Adam steps (observed x 9 ... x 12: 1.2e-5, 1.5e-4, 1.4e-3), first four epochs
BENCH_VAL_CAP = 5e-2              # ... up to the distance two fp32 runs of one training end up at anyway


@pytest.mark.timeout(900)
def test_bench_setting_early_stopping_matches_reference(tmp_path):
    """The regime the bench's LOSO block runs in — 270 +- 20 windows per subject, difficulty 2, B = 64, patience 20, epoch budget 100,
    where the reference's inverted early stopping FIRES — in the deterministic setting, through the shipped path (run_experiments,
    lockstep fold batch), for three folds of the 15-fold split: S3 (the HIP path stops it at the earliest possible epoch, 21), S7 and S5
    (fixture tests/golden/loso_parity_bench_ref.json: the reference's own CPU runs, 1.4 CPU-hours).

    What can be asserted, and what cannot: a fold whose validation loss stays below its first-epoch value stops at epoch 21 on every
    trajectory — there (S3) stop epoch, checkpoint epoch, test accuracy and F1 must equal the reference's.  A fold whose loss climbs
    back past an earlier "best" (the inverted rule resets its counter whenever the loss RISES to a new maximum) stops at an epoch that
    depends on last-bit differences amplified by 48 Adam steps per epoch: the reference does not reproduce ITS OWN stop epoch there
    (`reference_self_check`: the same code with 2 instead of 3 threads), so for those folds the test bounds the curves while they are
    comparable and requires the stop epoch only if the reference agrees with itself."""
    from multimodalsignal_amd import main as M
    from multimodalsignal_amd.synth import ALL_SUBJECTS, CHANNELS6, make_synthetic_wesad
    from multimodalsignal_amd.trainer import EarlyStopping
    fx = json.loads((GOLDEN / "loso_parity_bench_ref.json").read_text())
    ds, tr = fx["dataset"], fx["training"]
    data = Path("/tmp") / f"msig_parity_w{ds['windows_per_subject']}_s{ds['window_spread']}_t{ds['T']}_d{ds['difficulty']}"
    if not (data / "_channel_names.txt").exists():
        make_synthetic_wesad(data, windows_per_subject=ds["windows_per_subject"], T=ds["T"], difficulty=ds["difficulty"], window_spread=ds["window_spread"])
    names = (data / "_channel_names.txt").read_text().split()
    dev = torch.device("cuda:0")
    cfg = M.default_cfg()
    cfg.update(data_path=data, channels=list(CHANNELS6), epochs=tr["epochs"], batch_size=tr["batch"], patience=tr["patience"], lr=tr["lr"],
               weight_decay=tr["weight_decay"], seed=tr["seed_base"], shuffle=False, subjects=list(ALL_SUBJECTS), only_subjects=list(fx["folds"]),
               eval_batch_size=0,                          # validation in batches of 64, as the reference's loaders
               model_params=dict(cnn_out_channels=32, gru_hidden_size=64, gru_num_layers=2, dropout=0.0))
    results, wall = M.run_simple_experiment(tmp_path, dev, names, cfg)
    assert sorted(r["subject"] for r in results) == sorted(fx["folds"])
    sc = (fx.get("reference_self_check") or {}).get("folds", {})
    print(f"\nbench-setting folds on the HIP path in {wall:.1f} s (reference CPU: {sum(f['seconds_cpu'] for f in fx['folds'].values()):.0f} s)")
    for r in results:
        sid, ref = r["subject"], fx["folds"][r["subject"]]
        info = json.loads((tmp_path / f"fold_test_on_{sid}" / "fold_result.json").read_text())
        hist = info["history"]
        vl = [h["val_loss"] for h in hist]
        # our own rule on our own curve (the checkpoint epoch is not stored: replay)
        es, saved, ck = EarlyStopping(patience=tr["patience"], delta=0), [], 0
        es.save_checkpoint = lambda model, _s=saved: _s.append(1)
        for ep, v in enumerate(vl, 1):
            n_saved = len(saved)
            es(v, None)
            if len(saved) > n_saved:
                ck = ep
        n_cmp = min(len(vl), len(ref["val"]))
        gap = [abs(a - b[0]) for a, b in zip(vl[:n_cmp], ref["val"][:n_cmp])]
        own = None
        if sid in sc:
            m = min(n_cmp, len(sc[sid]["val"]))
            own = [abs(a[0] - b[0]) for a, b in zip(sc[sid]["val"][:m], ref["val"][:m])]
        print(f"  {sid}: HIP {len(vl)} epochs (checkpoint {ck}), reference {ref['epochs']} (checkpoint {ref['checkpoint_epoch']})"
              + (f", reference with {fx['reference_self_check']['threads']} threads {sc[sid]['epochs']} (checkpoint {sc[sid]['checkpoint_epoch']})" if sid in sc else "")
              + f"; acc {r['accuracy']:.4f} vs {ref['acc']:.4f}, f1 {r['f1_score']:.4f} vs {ref['f1']:.4f}; |val loss gap| epochs 1-6: "
              + " ".join(f"{g:.1e}" for g in gap[:6]) + (" | reference vs itself: " + " ".join(f"{g:.1e}" for g in own[:6]) if own else ""))
        # a wrong kernel shows in epoch 1; after that the two fp32 trajectories separate by about an order of magnitude per epoch (the
        # reference's own two runs — the same code on 2 and 3 threads — stay closer for a few epochs, 1e-8 ... 8e-4 over epochs 2-6,
        # and are 5e-2 apart by epoch 8: they differ in a summation order, the HIP path in every contraction)
        for e in range(min(4, n_cmp)):
            assert gap[e] <= min(BENCH_VAL_CAP, BENCH_VAL_TOL_EPOCH1 * BENCH_VAL_GROWTH ** e), (sid, e, gap[:4])
        robust = ref["early_stop"] and ref["epochs"] == tr["patience"] + 1 and ref["checkpoint_epoch"] == 1
        agrees_with_itself = sid in sc and sc[sid]["epochs"] == ref["epochs"] and sc[sid]["checkpoint_epoch"] == ref["checkpoint_epoch"]
        if robust or agrees_with_itself:
            n_te = len(np.load(data / f"{sid}_y.npy"))
            assert len(vl) == ref["epochs"] and ck == ref["checkpoint_epoch"], (sid, len(vl), ck, ref["epochs"], ref["checkpoint_epoch"])
            assert abs(r["accuracy"] - ref["acc"]) <= 1.0 / n_te + 1e-9 and abs(r["f1_score"] - ref["f1"]) <= 0.01, (sid, r, ref["acc"], ref["f1"])
    assert any(fx["folds"][s_]["early_stop"] and fx["folds"][s_]["epochs"] == tr["patience"] + 1 for s_ in fx["folds"])      # the fixture holds a robust fold

#!/usr/bin/env python3
"""Accuracy-parity experiment: the same LOSO folds on the same synthetic WESAD-shaped windows, trained (a) by this repo
on the GPU and (b) by the REFERENCE ITSELF on CPU (imports /root/reference — build container only).

    python tools/loso_parity.py --side gpu --out gpurun_out/parity_gpu.json [options]
    python tools/loso_parity.py --side ref --out /tmp/parity_ref.json [options]

Fold k of a run is seeded `seed_base + k` on both sides (model initialisation — bit-identical on both sides —, batch
shuffling and dropout, whose random streams differ by construction: SURVEY.md §5.1-7).  tests/golden/make_parity_fixture.py
turns a set of --side ref runs into the committed fixture that tests/test_accuracy_parity_gpu.py checks the HIP path against.
"""
import argparse, json, os, sys, time, types
from pathlib import Path
import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

DEFAULTS = dict(windows=100, samples=3840, difficulty=3.0, folds=["S2", "S5", "S9", "S13", "S17"], epochs=10, batch=64, dropout=0.5)


def ensure_data(data, windows, samples, difficulty, window_spread=0):
    from multimodalsignal_amd.synth import make_synthetic_wesad
    data = Path(data)
    if not (data / "_channel_names.txt").exists():
        make_synthetic_wesad(data, windows_per_subject=windows, T=samples, difficulty=difficulty, window_spread=window_spread)
    return (data / "_channel_names.txt").read_text().split()


def run_side(side, data, out_dir, seed_base=42, folds=DEFAULTS["folds"], epochs=10, batch=64, dropout=0.5, threads=8, device=None, log=print, shuffle=True):
    """Trains the given LOSO folds; returns [{subject, acc, f1, test_loss, seconds, history}]."""
    from multimodalsignal_amd.synth import ALL_SUBJECTS, CHANNELS6
    from multimodalsignal_amd.loso import split_train_val
    names = (Path(data) / "_channel_names.txt").read_text().split()
    cfgT = {"trainer": {"epochs": epochs, "learning_rate": 1e-3, "early_stopping": {"enabled": True, "patience": 20, "delta": 0},
                        "weight_decay": 1e-4, "verbose": False}}
    out_dir = Path(out_dir)
    results = []
    for sid in folds:
        k = ALL_SUBJECTS.index(sid)
        tr_s, va_s = split_train_val(ALL_SUBJECTS, sid, 42)
        torch.manual_seed(seed_base + k)
        t0 = time.time()
        if side == "gpu":
            from multimodalsignal_amd.dataset import DeviceLoader, WesadDataset
            from multimodalsignal_amd.models import CnnGruAttentionModel
            from multimodalsignal_amd.trainer import Trainer
            dev = device or torch.device("cuda:0")
            mk = lambda s: WesadDataset(data, s, CHANNELS6, names)
            tr, va, te = mk(tr_s), mk(va_s), mk([sid])
            model = CnnGruAttentionModel(6, 2, dropout=dropout)
            model.set_dropout_seed((seed_base + k) * 0x9E3779B97F4A7C15 + 12345)
            t = Trainer(model, out_dir / f"parity_fold_{sid}", cfgT)
            t.train(DeviceLoader(tr, batch, shuffle, dev, seed=seed_base + k), DeviceLoader(va, batch, False, dev))
            loss, acc, f1 = t.evaluate(DeviceLoader(te, batch, False, dev), is_test=True)
            hist = [[h["train_loss"], h["val_loss"], h["val_acc"]] for h in t.history]
        else:
            os.environ.setdefault("MPLBACKEND", "Agg")
            torch.set_num_threads(threads)
            sys.dont_write_bytecode = True           # nothing is written under /root/reference
            sys.path.insert(0, "/root/reference")
            sns = types.ModuleType("seaborn"); sns.heatmap = lambda *a, **k: None
            sys.modules.setdefault("seaborn", sns)
            import dataset as rd, models as rm, trainer as rt
            from torch.utils.data import DataLoader
            mk = lambda s: rd.WesadDataset(data, s, CHANNELS6, names)
            tr, va, te = mk(tr_s), mk(va_s), mk([sid])
            model = rm.CnnGruAttentionModel(6, 2, dropout=dropout)
            t = rt.Trainer(model, out_dir / f"parity_ref_fold_{sid}", cfgT)
            rec = []
            orig = t.evaluate
            def wrapped(loader, is_test=False, is_val=False, _o=orig, _r=rec):
                r = _o(loader, is_test=is_test, is_val=is_val); _r.append([float(r[0]), float(r[1]), float(r[2])]); return r
            t.evaluate = wrapped
            t.train(DataLoader(tr, batch_size=batch, shuffle=shuffle), DataLoader(va, batch_size=batch, shuffle=False))
            loss, acc, f1 = t.evaluate(DataLoader(te, batch_size=batch, shuffle=False), is_test=True)
            hist = [[None, r[0], r[1], r[2]] for r in rec[:-1]]
        results.append(dict(subject=sid, acc=float(acc), f1=float(f1), test_loss=float(loss), seconds=time.time() - t0, history=hist))
        log(sid, results[-1]["acc"], results[-1]["f1"], f"{results[-1]['seconds']:.1f}s")
    return results


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", choices=["gpu", "ref"], required=True)
    ap.add_argument("--data", type=Path, default=Path("/tmp/wesad_parity"))
    ap.add_argument("--windows", type=int, default=DEFAULTS["windows"])
    ap.add_argument("--samples", type=int, default=DEFAULTS["samples"])
    ap.add_argument("--difficulty", type=float, default=DEFAULTS["difficulty"])
    ap.add_argument("--window-spread", type=int, default=0, help="subjects get --windows +- this many windows (bench.py's LOSO set: 270 +- 20, difficulty 2)")
    ap.add_argument("--folds", nargs="+", default=DEFAULTS["folds"])
    ap.add_argument("--epochs", type=int, default=DEFAULTS["epochs"])
    ap.add_argument("--batch", type=int, default=DEFAULTS["batch"])
    ap.add_argument("--dropout", type=float, default=DEFAULTS["dropout"])
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--no-shuffle", action="store_true", help="training batches in dataset order (with --dropout 0: the deterministic setting)")
    ap.add_argument("--seed-base", type=int, nargs="+", default=[42], help="fold k is seeded seed_base + k (model init, shuffling, dropout); several = several runs")
    ap.add_argument("--out", type=Path, required=True)
    args = ap.parse_args()
    ensure_data(args.data, args.windows, args.samples, args.difficulty, args.window_spread)
    runs = []
    t_all = time.time()
    for sb in args.seed_base:
        res = run_side(args.side, args.data, args.out.parent, sb, args.folds, args.epochs, args.batch, args.dropout, args.threads, shuffle=not args.no_shuffle,
                       log=lambda *a: print(f"[seed {sb}]", *a, flush=True))
        runs.append(dict(seed_base=sb, results=res, mean_acc=float(np.mean([r["acc"] for r in res]))))
        doc = dict(args={k: str(v) for k, v in vars(args).items()}, runs=runs, wall_s=time.time() - t_all)
        if len(runs) == 1:       # single-run files keep the round-1 layout
            doc.update(results=res, mean_acc=runs[0]["mean_acc"])
        args.out.write_text(json.dumps(doc, indent=1))
    print("mean acc per run", [round(r["mean_acc"], 4) for r in runs])


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Accuracy-parity experiment (evidence for DESIGN.md, not a test): the same LOSO folds on the same
synthetic WESAD-shaped windows, trained (a) by this repo on the GPU and (b) by the REFERENCE ITSELF on
CPU (imports /root/reference — build container only).

    python tools/loso_parity.py --side gpu --out gpurun_out/parity_gpu.json [options]
    python tools/loso_parity.py --side ref --out /tmp/parity_ref.json [options]
"""
import argparse, json, os, sys, time, types
from pathlib import Path
import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
ap = argparse.ArgumentParser()
ap.add_argument("--side", choices=["gpu", "ref"], required=True)
ap.add_argument("--data", type=Path, default=Path("/tmp/wesad_parity"))
ap.add_argument("--windows", type=int, default=100)
ap.add_argument("--samples", type=int, default=3840)
ap.add_argument("--difficulty", type=float, default=3.0)
ap.add_argument("--folds", nargs="+", default=["S2", "S5", "S9", "S13", "S17"])
ap.add_argument("--epochs", type=int, default=10)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--dropout", type=float, default=0.5)
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--seed-base", type=int, default=42, help="fold k is seeded seed_base + k (model init, shuffling, dropout)")
ap.add_argument("--out", type=Path, required=True)
args = ap.parse_args()

from multimodalsignal_amd.synth import ALL_SUBJECTS, CHANNELS6, make_synthetic_wesad
from multimodalsignal_amd.loso import split_train_val
if not (args.data / "_channel_names.txt").exists():
    make_synthetic_wesad(args.data, windows_per_subject=args.windows, T=args.samples, difficulty=args.difficulty)
names = (args.data / "_channel_names.txt").read_text().split()
cfgT = {"trainer": {"epochs": args.epochs, "learning_rate": 1e-3, "early_stopping": {"enabled": True, "patience": 20, "delta": 0},
                    "weight_decay": 1e-4, "verbose": False}}
results = []
t_all = time.time()
for sid in args.folds:
    k = ALL_SUBJECTS.index(sid)
    tr_s, va_s = split_train_val(ALL_SUBJECTS, sid, 42)
    torch.manual_seed(args.seed_base + k)
    t0 = time.time()
    if args.side == "gpu":
        from multimodalsignal_amd.dataset import DeviceLoader, WesadDataset
        from multimodalsignal_amd.models import CnnGruAttentionModel
        from multimodalsignal_amd.trainer import Trainer
        dev = torch.device("cuda:0")
        mk = lambda s: WesadDataset(args.data, s, CHANNELS6, names)
        tr, va, te = mk(tr_s), mk(va_s), mk([sid])
        model = CnnGruAttentionModel(6, 2, dropout=args.dropout)
        t = Trainer(model, args.out.parent / f"parity_fold_{sid}", cfgT)
        t.train(DeviceLoader(tr, args.batch, True, dev, seed=args.seed_base + k), DeviceLoader(va, args.batch, False, dev))
        loss, acc, f1 = t.evaluate(DeviceLoader(te, args.batch, False, dev), is_test=True)
        hist = [[h["train_loss"], h["val_loss"], h["val_acc"]] for h in t.history]
    else:
        os.environ.setdefault("MPLBACKEND", "Agg")
        torch.set_num_threads(args.threads)
        sys.path.insert(0, "/root/reference")
        sns = types.ModuleType("seaborn"); sns.heatmap = lambda *a, **k: None
        sys.modules.setdefault("seaborn", sns)
        import dataset as rd, models as rm, trainer as rt
        from torch.utils.data import DataLoader
        mk = lambda s: rd.WesadDataset(args.data, s, CHANNELS6, names)
        tr, va, te = mk(tr_s), mk(va_s), mk([sid])
        model = rm.CnnGruAttentionModel(6, 2, dropout=args.dropout)
        t = rt.Trainer(model, args.out.parent / f"parity_ref_fold_{sid}", cfgT)
        rec = []
        orig = t.evaluate
        def wrapped(loader, is_test=False, is_val=False, _o=orig, _r=rec):
            r = _o(loader, is_test=is_test, is_val=is_val); _r.append([float(r[0]), float(r[1])]); return r
        t.evaluate = wrapped
        t.train(DataLoader(tr, batch_size=args.batch, shuffle=True), DataLoader(va, batch_size=args.batch, shuffle=False))
        loss, acc, f1 = t.evaluate(DataLoader(te, batch_size=args.batch, shuffle=False), is_test=True)
        hist = [[None, r[0], r[1]] for r in rec[:-1]]
    results.append(dict(subject=sid, acc=float(acc), f1=float(f1), test_loss=float(loss), seconds=time.time() - t0, history=hist))
    print(sid, results[-1]["acc"], results[-1]["f1"], f"{results[-1]['seconds']:.1f}s", flush=True)
    args.out.write_text(json.dumps(dict(args={k: str(v) for k, v in vars(args).items()}, results=results,
                                        mean_acc=float(np.mean([r["acc"] for r in results])), wall_s=time.time() - t_all), indent=1))
print("mean acc", np.mean([r["acc"] for r in results]))

"""LOSO experiment driver — the reference's ``main.py`` (run_simple_experiment,
main.py:91-156) with the 15 folds sharded over the GPUs of one node.

    python -m multimodalsignal_amd.main --data ./data/chest_raw                 # 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 -m multimodalsignal_amd.main --data ...
    python -m multimodalsignal_amd.main --synthetic /tmp/wesad_synth             # no dataset needed

Same module-level constants as the reference (edit them or use the flags), same fold
directories / ``cv_summary.txt``.  Differences, all deliberate: every fold gets an explicit
seed (SEED + fold index) because a sharded run cannot share one global RNG stream
(SURVEY.md §5.1-7); data live in HBM (DeviceLoader); rank 0 writes the summary after one
RCCL all_gather of the per-fold metrics.  The hierarchical experiment (main.py:159-247), which the
reference cannot run itself, is run_hierarchical_experiment / --hierarchical.
"""
from __future__ import annotations

import argparse
import json
import os
import time
import warnings
from datetime import datetime
from pathlib import Path

import numpy as np
import torch

from .dataset import DeviceLoader, SubjectStore, WesadDataset
from .loso import folds_for_rank, gather_fold_metrics, split_train_val
from .models import CnnGruAttentionModel
from .trainer import Trainer

warnings.filterwarnings("ignore", message="Initializing zero-element tensors is a no-op")

# ---- configuration (main.py:20-67) -------------------------------------------------------------
RUN_NAME = "simple_binary"
CLASSIFICATION_MODE = "stress_binary"
NUM_CLASSES = 2
MODEL_TO_USE = "cnn_gru_attention"
CHANNELS_TO_USE = ["chest_ECG", "chest_EDA", "chest_Resp"]
MODEL_PARAMS = {"cnn_gru_attention": {"cnn_out_channels": 32, "gru_hidden_size": 64, "gru_num_layers": 2, "dropout": 0.5}}
PROCESSED_DATA_PATH = Path("./data")
EARLY_DATA_PATH = PROCESSED_DATA_PATH / "chest_raw"
SEED = 42
NUM_WORKERS = 0
EPOCHS = 100
BATCH_SIZE = 64
LEARNING_RATE = 0.001
PATIENCE = 20
WEIGHTS_DECAY = 1e-4
ALL_SUBJECTS = [f"S{i}" for i in range(2, 18) if i != 12]
# Batch size of the validation / test passes.  The reference evaluates in batches of BATCH_SIZE (main.py:113-114); evaluation runs in
# eval mode (running BatchNorm statistics, no dropout), so every window's logits — and with them accuracy, F1 and the confusion matrix —
# are the same whatever the batching; only the summation order of the reported loss changes (~1e-7 relative).  One launch sequence
# over a subject's ~270 windows instead of five is what the latency-bound B = 64 regime wants (0 = BATCH_SIZE, the reference's loaders).
EVAL_BATCH_SIZE = 1024
# Concurrent HIP streams per GPU.  Measured (profiles/r05_stream_sweep.log: G single-fold step loops on G streams, GPU_MAX_HW_QUEUES
# 4 / 8 / 16 / 24): 1.04 / 1.12 / 1.19 / 1.26 ms per step with 1 / 2 / 3 / 4 streams, then 1.9-2.3 ms with FIVE whatever the queue
# count — the command processor's four pipes each work on one queue at a time, and a stream whose next launch waits behind a
# 200-us recurrence of another stream on the same pipe waits for all of it.  Nothing in this driver runs more than four streams of
# training at once: four fold batches per configuration (+ a side stream for finished folds' short test passes), one per configuration
# in a sweep, and at most four single-fold streams with --no-lockstep (round 4 ran fifteen there).
MAX_TRAIN_STREAMS = 4


def prepare_fold(fold_idx, subject_to_test, run_output_dir, device, all_channel_names, cfg, cache=None):
    """Everything of one fold that touches global state (RNG seeding, model initialisation, host data):
    done sequentially in the main thread so that concurrent folds stay deterministic."""
    fold_dir = Path(run_output_dir) / f"fold_test_on_{subject_to_test}"
    fold_dir.mkdir(parents=True, exist_ok=True)
    train_subjects, val_subjects = split_train_val(cfg["subjects"], subject_to_test, cfg["seed"])
    if isinstance(cache, SubjectStore):       # one HBM-resident store for the whole run: datasets are index subsets
        mk = cache.view
    else:
        mk = lambda subj: WesadDataset(cfg["data_path"], subj, cfg["channels"], all_channel_names,
                                       classification_mode=cfg["mode"], cache=cache)
    train_ds, val_ds, test_ds = mk(train_subjects), mk(val_subjects), mk([subject_to_test])
    fold_seed = cfg["seed"] + fold_idx
    torch.manual_seed(fold_seed)
    # Evaluation runs in eval mode (running BN statistics, no dropout), so its predictions do not depend on how the
    # windows are batched; only the summation order of the reported loss does (~1e-7 relative).
    ebs = int(cfg.get("eval_batch_size") or cfg["batch_size"])
    # cfg["shuffle"] = False: training batches in dataset order (with dropout 0 the run is deterministic up to rounding — the
    # setting tests/test_accuracy_parity_gpu.py compares fold by fold with the reference's CPU run); default as main.py:112
    loaders = (DeviceLoader(train_ds, cfg["batch_size"], bool(cfg.get("shuffle", True)), device, seed=fold_seed),
               DeviceLoader(val_ds, ebs, False, device), DeviceLoader(test_ds, ebs, False, device))
    model = CnnGruAttentionModel(in_channels=len(cfg["channels"]), num_classes=cfg["num_classes"], **cfg["model_params"])
    model.set_dropout_seed(fold_seed * 0x9E3779B97F4A7C15 + 12345)
    pat = cfg["patience"]
    if isinstance(pat, (list, tuple)):       # a per-fold cycle of patiences (tests: folds that stop at different epochs); an int as in main.py:66
        pat = int(pat[fold_idx % len(pat)])
    config_dict = {"trainer": {"epochs": cfg["epochs"], "learning_rate": cfg["lr"],
                               "early_stopping": {"enabled": True, "patience": pat, "delta": 0},
                               "weight_decay": cfg["weight_decay"], "verbose": cfg["verbose"]}}
    return dict(fold=fold_idx, subject=subject_to_test, fold_dir=fold_dir, loaders=loaders, model=model, config=config_dict)


def train_fold(prep, device):
    """The reference's per-fold body (main.py:116-125) on the current HIP stream."""
    train_loader, val_loader, test_loader = prep["loaders"]
    trainer = Trainer(prep["model"], prep["fold_dir"], prep["config"])
    t0 = time.time()
    trainer.train(train_loader, val_loader)
    _, test_acc, test_f1 = trainer.evaluate(test_loader, is_test=True)
    info = dict(subject=prep["subject"], accuracy=test_acc, f1_score=test_f1, seconds=time.time() - t0,
                epochs=len(trainer.history), train_windows_per_s=trainer.train_windows / max(trainer.train_seconds, 1e-9),
                history=trainer.history)
    (prep["fold_dir"] / "fold_result.json").write_text(json.dumps(info))      # survives a crash of another fold
    return info


def run_fold(fold_idx, subject_to_test, run_output_dir, device, all_channel_names, cfg, cache=None):
    """One iteration of the reference's fold loop (main.py:99-125)."""
    return train_fold(prepare_fold(fold_idx, subject_to_test, run_output_dir, device, all_channel_names, cfg, cache), device)


def write_summary(run_output_dir, results, cfg, wall_s, world):
    accs = [r["accuracy"] for r in results]
    f1s = [r["f1_score"] for r in results]
    path = Path(run_output_dir) / "cv_summary.txt"
    with open(path, "w", encoding="utf-8") as f:
        f.write("实验配置:\n")
        for k, v in (("MODEL_TO_USE", MODEL_TO_USE), ("RUN_NAME", RUN_NAME), ("SEED", cfg["seed"]), ("CHANNELS_TO_USE", cfg["channels"]),
                     ("EPOCHS", cfg["epochs"]), ("BATCH_SIZE", cfg["batch_size"]), ("LEARNING_RATE", cfg["lr"]),
                     ("NUM_WORKERS", NUM_WORKERS), ("PATIENCE", cfg["patience"]), ("NUM_CLASSES", cfg["num_classes"]),
                     ("MODEL_PARAMS", {MODEL_TO_USE: cfg["model_params"]})):
            f.write(f"{k}: {v}\n")
        f.write("\n每个折叠的详细结果:\n")
        for r in results:
            f.write(f"  - 测试 {r['subject']}: Accuracy = {r['accuracy']:.4f}, F1-score = {r['f1_score']:.4f}\n")
        f.write("\n最终平均性能:\n")
        f.write(f"平均准确率 (Accuracy): {np.mean(accs):.4f} ± {np.std(accs):.4f}\n")
        f.write(f"平均 F1 分数 (Weighted F1-score): {np.mean(f1s):.4f} ± {np.std(f1s):.4f}\n")
        f.write(f"\nLOSO wall-clock: {wall_s:.1f} s on {world} GPU(s)\n")
    return path


def run_experiments(run_output_dir, device, all_channel_names, cfgs, rank=0, world=1):
    """Runs the LOSO loop of every configuration in `cfgs` ({name: cfg}) as ONE sharded job: the work units
    are (configuration, fold) pairs — 15 for a plain run, 4 x 15 = 60 for the channel-ablation sweep — dealt
    round-robin to the ranks and trained concurrently within a rank.  A configuration named "" writes into
    `run_output_dir` itself, any other into `run_output_dir/<name>`.  Returns {name: results}, wall seconds."""
    names = list(cfgs)
    cfg0 = cfgs[names[0]]
    t0 = time.time()
    # matplotlib (the confusion-matrix plots) costs ~0.4 s of interpreter time the first time it is imported: started here on a
    # thread, it runs while this thread reads and normalises the subjects' files (numpy, mostly outside the interpreter lock)
    # instead of in front of the first fold's test pass.  (Round 3 also warmed torch._dynamo here, which torch.optim's first
    # Optimizer pulled in — 0.9 s that still ended up in front of the first train step; trainer.MsigAdam no longer triggers it.)
    import threading

    def _warm_imports():
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot  # noqa: F401
        except Exception:
            pass
    warm = threading.Thread(target=_warm_imports, daemon=True)
    warm.start()
    stores = {n: SubjectStore(c["data_path"], c["subjects"], c["channels"], all_channel_names, classification_mode=c["mode"],
                              device=device, normalise=c.get("normalise", "host")) for n, c in cfgs.items()}
    t_data = time.time() - t0
    # unit u = (fold-major, configuration-minor): neighbouring units of one fold index go to different ranks
    units = [(n, k) for k in range(max(len(c["subjects"]) for c in cfgs.values())) for n in names if k < len(cfgs[n]["subjects"])]
    mine = folds_for_rank(len(units), world, rank)
    if cfg0.get("only_subjects"):       # a subset of the LOSO's folds (tests: the splits stay those of the full subject list)
        keep = set(cfg0["only_subjects"])
        mine = [u for u in mine if cfgs[units[u][0]]["subjects"][units[u][1]] in keep]
    conc = max(1, min(int(cfg0.get("concurrent_folds", 1)), len(mine)))
    out_dir = {n: (Path(run_output_dir) / n if n else Path(run_output_dir)) for n in names}
    local = {}

    def report(u, info):
        n, k = units[u]
        local[u] = (info["accuracy"], info["f1_score"])
        tag = f"{n}/" if n else ""
        print(f"[rank {rank}] {tag}fold {k} ({cfgs[n]['subjects'][k]}): acc {info['accuracy']:.4f} f1 {info['f1_score']:.4f} "
              f"{info['epochs']} epochs {info['seconds']:.1f}s {info['train_windows_per_s']:.0f} windows/s", flush=True)

    def prep(u):
        n, k = units[u]
        return prepare_fold(k, cfgs[n]["subjects"][k], out_dir[n], device, all_channel_names, cfgs[n], stores[n])

    lockstep_done = False
    if conc > 1 and mine and cfg0.get("lockstep", True):
        # Folds of one configuration advance in LOCKSTEP as one fold batch: every launch of the step covers all of them
        # (multifold.LockstepTrainer, msig_*_multi) — at B = 64 fifteen streams are bound by the command processor, one set of
        # launches is not.  The folds' splits may differ in size (real WESAD): full batches share launches, ragged last batches
        # run over the folds whose batch sizes agree.  At most `concurrent_folds` folds are resident at a time.
        from .multifold import LockstepTrainer, lockstep_compatible
        groups = {}
        for u in mine:
            groups.setdefault(units[u][0], []).append(u)
        # Each configuration's folds are dealt round-robin into `lockstep_groups` fold batches (default 4), each advancing in
        # lockstep on its own HIP stream: one batch of 15 is bound by the latency of its ~30 dependent launches per step
        # (2.1 ms at 15 folds, 0.96 ms at one) and runs as many epochs as its slowest fold; several smaller batches overlap
        # each other's latency (and each other's per-epoch host work and sync), let early finishers free their share sooner and
        # — what decides the wall-clock — keep the few folds that train longest in SMALL batches: a step costs 0.96 + 0.11 ms per
        # further fold of its batch, and which folds stop late is not known when they are dealt.  As many batches as the command
        # processor has pipes (MAX_TRAIN_STREAMS = 4).  Round 5, bench LOSO, the same 558 fold-epochs: 7.44 s with 1 batch, 7.03 with
        # 2, 7.06-7.12 with 3 (round 2-4's default: its deal puts the folds of 76, 70 and 59 epochs into one batch), 6.59-6.69 with 4
        # (profiles/r05_loso_groups.log) — the short test pass of a finished fold on the side stream is a fifth stream for a few
        # milliseconds and is inside those numbers.
        ng = max(1, int(cfg0.get("lockstep_groups", 4)))
        adaptive = bool(cfg0.get("adaptive_forms", False))
        preps = {u: prep(u) for u in mine}               # sequential: seeding / initialisation order as in every other mode
        waves = []                                       # lists of chunks; the chunks of a wave run concurrently, waves one after another
        glist = list(groups.values())
        if len(glist) > 1:
            # A sweep (channel ablation: 4 configurations x 15 folds): the configurations' fold batches run CONCURRENTLY, one
            # stream each, at most MAX_TRAIN_STREAMS at a time — round 4 ran them as sequential waves of three streams, each wave
            # ending in a multi-second tail with one or two folds left on an otherwise idle GPU.
            for c0 in range(0, len(glist), MAX_TRAIN_STREAMS):
                wave = []
                for g in glist[c0:c0 + MAX_TRAIN_STREAMS]:
                    wave += [g[i:i + 16] for i in range(0, len(g), 16)]
                waves.append(wave)
        else:
            for g in glist:
                for w0 in range(0, len(g), conc):
                    gw = g[w0:w0 + conc]
                    k = min(ng, max(1, len(gw) // 2), MAX_TRAIN_STREAMS)
                    parts = [gw[i::k] for i in range(k)]
                    waves.append([part[i:i + 16] for part in parts for i in range(0, len(part), 16)])
        chunk_preps = [[[(u, preps[u]) for u in ch] for ch in wv] for wv in waves]
        if all(lockstep_compatible([p for _, p in ch]) for wv in chunk_preps for ch in wv):
            torch.cuda.synchronize(device)
            t_lock0 = time.time()

            streams = {}                                          # one HIP stream per fold-batch position, reused across rounds

            def work(args):
                ch, epoch0, n_ep, pos_ = args
                torch.cuda.set_device(device)
                if pos_ not in streams:
                    streams[pos_] = torch.cuda.Stream(device)
                with torch.cuda.stream(streams[pos_]):
                    lt = LockstepTrainer([p for _, p in ch], device, adaptive_forms=adaptive)
                    done, alive = lt.run(epoch0=epoch0, max_epochs=n_ep, t_start=t_lock0)
                    torch.cuda.current_stream(device).synchronize()
                return ch, done, alive, lt

            from concurrent.futures import ThreadPoolExecutor

            def run_round(chunks, epoch0, n_ep):
                """One round: every chunk (fold batch) trains epochs epoch0 .. epoch0 + n_ep - 1 on its own stream.  Returns the units
                still training (with their preps, which now carry their Trainer) — and keeps the finished batches' arenas alive
                until the survivors have been re-dealt (their state is copied out of them)."""
                jobs = [(ch, epoch0, n_ep, i) for i, ch in enumerate(chunks)]
                if len(jobs) == 1:
                    res = [work(jobs[0])]
                else:
                    with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
                        res = list(ex.map(work, jobs))
                alive_units = []
                for ch, done, alive, lt in res:
                    for pos, info in done.items():
                        report(ch[pos][0], info)
                    alive_units += [ch[pos] for pos in alive]
                return sorted(alive_units, key=lambda up: up[0]), res

            budget = int(cfg0["epochs"])
            pats = cfg0["patience"] if isinstance(cfg0["patience"], (list, tuple)) else [cfg0["patience"]]
            first_round = min(int(p_) for p_ in pats) + 1            # no fold can stop before patience + 1 epochs
            # Re-dealing the surviving folds evenly over the fold batches between rounds — after the first patience + 1 epochs (0) or every n
            # epochs (n) — is BUILT AND OFF (-1, the default): measured on the bench's LOSO (same 558 fold-epochs, same box, profiles/
            # r05_loso_redeal.log) never: 7.30 / 7.23 s, once: 7.41 / 7.37 s, every 16 / 8 epochs: 7.47 / 7.5-7.7 s.  The run is bound by its
            # longest fold's serial chain wherever that fold sits; evening out the batches slows the short folds' batches down and the
            # round boundaries (the batches wait for each other, arenas are rebuilt) cost more than the balance returns.
            redeal = int(cfg0.get("redeal_every", -1))
            for wv in chunk_preps:
                if redeal < 0 or len(names) > 1:
                    run_round(wv, 0, budget)                          # sweeps: one fold batch per configuration, run to the end
                    continue
                # Rounds.  Early stopping thins the fold batches unevenly — the batch that happens to hold the long folds bounds
                # the run (round 4 / first half of round 5: 6.67 s against 6.08 s for its neighbour) — so after the first
                # patience + 1 epochs, and then every `redeal_every`, the folds still training are dealt evenly over the batches
                # again (at most `lockstep_groups`, never more than one batch per fold).  A fold's numbers do not depend on its
                # companions, so this changes nothing but the wall-clock.
                alive_units, keep = run_round(wv, 0, min(first_round, budget))
                epoch0 = min(first_round, budget)
                while alive_units and epoch0 < budget:
                    k = max(1, min(ng, len(alive_units), MAX_TRAIN_STREAMS))
                    chunks = [alive_units[i::k] for i in range(k)]
                    n_ep = min(redeal, budget - epoch0) if redeal > 0 else budget - epoch0
                    alive_units, keep2 = run_round(chunks, epoch0, n_ep)
                    epoch0 += n_ep
                    keep = keep2                                      # the previous round's arenas may go now
                del keep
            lockstep_done = True
        else:
            del preps, chunk_preps
    if lockstep_done:
        pass
    elif conc == 1:
        for u in mine:
            report(u, train_fold(prep(u), device))
    elif mine:
        # At the reference's batch size (64) one fold keeps ~2 % of an MI355X busy (4 batch tiles of a strictly
        # sequential recurrence), so the rank's units run concurrently, each on its own HIP stream.  Seeding,
        # model initialisation and host-side data preparation stay sequential (deterministic); only the
        # training loops overlap (libmsig_hip.so is re-entrant across streams; ctypes releases the GIL).
        from concurrent.futures import ThreadPoolExecutor
        preps = [(u, prep(u)) for u in mine]
        torch.cuda.synchronize(device)      # uploads were issued on this thread's stream

        def work(item):
            u, p = item
            torch.cuda.set_device(device)
            with torch.cuda.stream(torch.cuda.Stream(device)):
                info = train_fold(p, device)
                torch.cuda.current_stream(device).synchronize()
            return u, info

        with ThreadPoolExecutor(max_workers=min(conc, MAX_TRAIN_STREAMS)) as ex:      # beyond four streams the device falls off a cliff
            for u, info in ex.map(work, preps):
                report(u, info)
    if cfg0.get("emulate_rank"):
        # bench.py --emulate-ranks: this process plays rank `rank` of a `world`-GPU job ALONE on its GPU — exactly what that rank
        # executes on an 8-GPU node, less the one ~100-byte all_gather of the fold metrics
        allm = dict(local)
    else:
        allm = gather_fold_metrics(local, len(units), world, cfg0.get("gather_device", device))
    wall = time.time() - t0
    warm.join()            # long done in a real run; a tiny one must not leave an import running at interpreter exit
    results = {n: [] for n in names}
    for u in sorted(allm):
        n, k = units[u]
        results[n].append({"subject": cfgs[n]["subjects"][k], "accuracy": allm[u][0], "f1_score": allm[u][1]})
    if rank == 0 or cfg0.get("emulate_rank"):
        for n in names:
            out_dir[n].mkdir(parents=True, exist_ok=True)
            path = write_summary(out_dir[n], results[n], cfgs[n], wall, world)
            accs = [r["accuracy"] for r in results[n]]
            print(f"交叉验证汇总结果已保存至: {path}")
            print(f"{(n + ': ') if n else ''}平均准确率 (Accuracy): {np.mean(accs):.4f} ± {np.std(accs):.4f}"
                  f" | LOSO wall-clock {wall:.1f}s on {world} GPU(s) ({len(units)} folds in this job; load + normalise + upload {t_data:.1f}s)")
    return results, wall


def run_simple_experiment(run_output_dir, device, all_channel_names, cfg=None, rank=0, world=1):
    """The reference's entry point (main.py:91): one configuration, 15 folds."""
    results, wall = run_experiments(run_output_dir, device, all_channel_names, {"": cfg or default_cfg()}, rank, world)
    return results[""], wall


# ---- hierarchical experiment (main.py:20-40, 159-247) ------------------------------------------------------------------
M1_CHANNELS_TO_USE = ["chest_ECG", "chest_EDA", "chest_Resp"]
M1_MODEL_PARAMS = {"cnn_out_channels": 32, "gru_hidden_size": 64, "gru_num_layers": 2, "dropout": 0.5}
M2_CHANNELS_TO_USE = ["chest_ECG", "chest_EDA", "chest_Resp"]
M2_MODEL_PARAMS = {"cnn_out_channels": 32, "gru_hidden_size": 32, "gru_num_layers": 1, "dropout": 0.5}


def run_hierarchical_experiment(run_output_dir, device, all_channel_names, cfg=None, rank=0, world=1):
    """The reference's run_hierarchical_experiment (main.py:159-247): per LOSO fold a stress-vs-rest model M1 (the reference
    configuration) and an amusement-vs-baseline model M2 (gru_hidden_size 32, gru_num_layers 1: runtime.EmbeddedEngine), then the
    three-class decision `2 if M1 says stress else M2's class` on the test subject.  The reference cannot run this function (its
    dataset.py raises for 'amusement_binary'): the label map is defined in dataset.map_labels, and the
    summary the reference stops short of (overall three-class accuracy / weighted F1, per-fold M1 accuracy) is written to
    hierarchical_summary.txt.  Folds are dealt to the ranks like the simple experiment's; each fold's two models train one after
    the other on the rank's GPU.  Returns (per-fold dicts in subject order, wall seconds)."""
    from .trainer import accuracy_and_weighted_f1
    cfg = dict(cfg or default_cfg())
    m1_ch, m2_ch = list(cfg.get("m1_channels", M1_CHANNELS_TO_USE)), list(cfg.get("m2_channels", M2_CHANNELS_TO_USE))
    m1_par, m2_par = dict(cfg.get("m1_params", M1_MODEL_PARAMS)), dict(cfg.get("m2_params", M2_MODEL_PARAMS))
    subjects, t0, cache = list(cfg["subjects"]), time.time(), {}

    def tcfg_for(k):
        pat = cfg["patience"]
        if isinstance(pat, (list, tuple)):       # a per-fold cycle of patiences, as in prepare_fold
            pat = int(pat[k % len(pat)])
        return {"trainer": {"epochs": cfg["epochs"], "learning_rate": cfg["lr"],
                            "early_stopping": {"enabled": True, "patience": pat, "delta": 0},
                            "weight_decay": cfg["weight_decay"], "verbose": cfg.get("verbose", False)}}
    mk = lambda subj, ch, mode: WesadDataset(cfg["data_path"], subj, ch, all_channel_names, classification_mode=mode, cache=cache)
    bs = cfg["batch_size"]
    ebs = int(cfg.get("eval_batch_size") or bs)              # validation / test passes (per-window results do not depend on it)
    local, rows = {}, {}
    for k in folds_for_rank(len(subjects), world, rank):
        sid = subjects[k]
        fold_dir = Path(run_output_dir) / f"fold_test_on_{sid}"
        fold_dir.mkdir(parents=True, exist_ok=True)
        train_subjects, val_subjects = split_train_val(subjects, sid, cfg["seed"])
        trainers = {}
        for tag, ch, par, mode in (("m1", m1_ch, m1_par, "stress_binary"), ("m2", m2_ch, m2_par, "amusement_binary")):
            torch.manual_seed(cfg["seed"] + 2 * k + (tag == "m2"))
            tr_ds, va_ds = mk(train_subjects, ch, mode), mk(val_subjects, ch, mode)
            if len(tr_ds) == 0 or len(va_ds) == 0:                # main.py:187-189
                print(f"警告: 训练集或验证集在 {mode} 模式下没有数据，跳过此折叠。")
                break
            model = CnnGruAttentionModel(in_channels=len(ch), num_classes=2, **par)
            model.set_dropout_seed((cfg["seed"] + 2 * k + (tag == "m2")) * 0x9E3779B97F4A7C15 + 12345)
            t = Trainer(model, fold_dir / f"model_{tag}", tcfg_for(k))
            t.train(DeviceLoader(tr_ds, bs, True, device, seed=cfg["seed"] + 2 * k + (tag == "m2")), DeviceLoader(va_ds, ebs, False, device))
            trainers[tag] = t
        if len(trainers) < 2:
            continue
        _, m1_acc, m1_f1 = trainers["m1"].evaluate(DeviceLoader(mk([sid], m1_ch, "stress_binary"), ebs, False, device), is_test=True)   # main.py:203-207
        eval_ch = list(dict.fromkeys(m1_ch + m2_ch))              # main.py:211 (a set there: the order is immaterial, the indices follow it)
        tern = mk([sid], eval_ch, "ternary")
        i1, i2 = [eval_ch.index(c) for c in m1_ch], [eval_ch.index(c) for c in m2_ch]
        m1, m2 = trainers["m1"].model.eval(), trainers["m2"].model.eval()
        preds = []
        with torch.no_grad():
            for xb, _ in DeviceLoader(tern, ebs, False, device):
                p1 = torch.argmax(m1(xb[:, i1, :].contiguous()), dim=1)
                p2 = torch.argmax(m2(xb[:, i2, :].contiguous()), dim=1)
                preds.append(torch.where(p1 == 1, torch.full_like(p2, 2), p2))      # main.py:243
        pred = torch.cat(preds).cpu().numpy()
        acc3, f13 = accuracy_and_weighted_f1(np.asarray(tern.labels), pred)
        rows[k] = dict(subject=sid, m1_accuracy=m1_acc, m1_f1=m1_f1, ternary_accuracy=acc3, ternary_f1=f13, n=int(len(pred)),
                       correct=int((pred == np.asarray(tern.labels)).sum()))
        (fold_dir / "fold_result.json").write_text(json.dumps(rows[k]))
        local[k] = (m1_acc, acc3)
        print(f"[rank {rank}] fold {k} ({sid}): M1 acc {m1_acc:.4f} | three-class acc {acc3:.4f} f1 {f13:.4f}", flush=True)
    allm = gather_fold_metrics(local, len(subjects), world, cfg.get("gather_device", device))
    wall = time.time() - t0
    results = [dict(subject=subjects[k], m1_accuracy=allm[k][0], ternary_accuracy=allm[k][1]) for k in sorted(allm)]
    if rank == 0:
        path = Path(run_output_dir) / "hierarchical_summary.txt"
        with open(path, "w", encoding="utf-8") as f:
            f.write(f"M1 {m1_ch} {m1_par}\nM2 {m2_ch} {m2_par}\n\n")
            for r in results:
                f.write(f"  - 测试 {r['subject']}: M1 Accuracy = {r['m1_accuracy']:.4f}, 三分类 Accuracy = {r['ternary_accuracy']:.4f}\n")
            if results:
                f.write(f"\n平均 M1 准确率: {np.mean([r['m1_accuracy'] for r in results]):.4f}\n")
                f.write(f"平均三分类准确率: {np.mean([r['ternary_accuracy'] for r in results]):.4f}\n")
            f.write(f"\nwall-clock: {wall:.1f} s on {world} GPU(s)\n")
        print(f"分层分类汇总结果已保存至: {path}")
    return results, wall


def ablation_sets(all_channel_names):
    """The channel-ablation sweep of BASELINE.json: ECG only, EDA only, every chest channel, every wrist channel
    (sets whose channels the dataset does not have are dropped)."""
    have = list(all_channel_names)
    sets = {"ecg_only": [c for c in have if c == "chest_ECG"], "eda_only": [c for c in have if c == "chest_EDA"],
            "chest_only": [c for c in have if c.startswith("chest_")], "wrist_only": [c for c in have if c.startswith("wrist_")]}
    return {n: ch for n, ch in sets.items() if ch}


def default_cfg():
    return dict(data_path=EARLY_DATA_PATH, channels=list(CHANNELS_TO_USE), mode=CLASSIFICATION_MODE, num_classes=NUM_CLASSES,
                model_params=dict(MODEL_PARAMS[MODEL_TO_USE]), seed=SEED, epochs=EPOCHS, batch_size=BATCH_SIZE, lr=LEARNING_RATE,
                patience=PATIENCE, weight_decay=WEIGHTS_DECAY, subjects=list(ALL_SUBJECTS), verbose=False, concurrent_folds=15,
                eval_batch_size=EVAL_BATCH_SIZE)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--data", type=Path, default=None, help="directory with {sid}_X.npy/_y.npy and _channel_names.txt")
    ap.add_argument("--synthetic", type=Path, default=None, help="generate (if missing) and use a synthetic dataset here")
    ap.add_argument("--synthetic-windows", type=int, default=270)
    ap.add_argument("--samples", type=int, default=3840)
    ap.add_argument("--channels", nargs="+", default=None)
    ap.add_argument("--ablation", action="store_true",
                    help="channel-ablation sweep {ECG-only, EDA-only, chest-only, wrist-only} x all folds as one sharded job")
    ap.add_argument("--sweep", nargs="+", default=None, metavar="NAME=CH1,CH2",
                    help="custom sweep: one LOSO run per named channel set, all folds of all sets sharded together")
    ap.add_argument("--epochs", type=int, default=EPOCHS)
    ap.add_argument("--patience", type=int, nargs="+", default=[PATIENCE], help="early-stopping patience; several values = a per-fold cycle")
    ap.add_argument("--batch-size", type=int, default=BATCH_SIZE)
    ap.add_argument("--eval-batch-size", type=int, default=EVAL_BATCH_SIZE,
                    help="batch size of the validation / test passes (per-window results do not depend on it; 0 = --batch-size, the reference's loaders)")
    ap.add_argument("--subjects", nargs="+", default=None)
    ap.add_argument("--out", type=Path, default=Path("./output"))
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--concurrent-folds", type=int, default=15,
                    help="folds resident per GPU at a time (as lockstep fold batches; with --no-lockstep at most MAX_TRAIN_STREAMS = 4 of them train at once); 1 = sequential")
    ap.add_argument("--lockstep-groups", type=int, default=4, help="fold batches per configuration, each on its own HIP stream")
    ap.add_argument("--hierarchical", action="store_true",
                    help="the reference's hierarchical experiment (main.py:159-247): M1 stress vs rest + M2 amusement vs baseline per fold")
    ap.add_argument("--adaptive-forms", action="store_true",
                    help="let the GRU kernel form of a fold batch follow the folds still active in each launch (faster on one GPU; a "
                         "fold's last bits then depend on its companions — by default they do not depend on grouping or rank count)")
    ap.add_argument("--window-spread", type=int, default=0, help="synthetic set: subjects get --synthetic-windows +- this many windows")
    ap.add_argument("--no-lockstep", action="store_true",
                    help="train concurrent folds on one HIP stream each instead of as one fold batch (msig_*_multi)")
    ap.add_argument("--difficulty", type=float, default=1.0, help="noise scale of the synthetic dataset")
    ap.add_argument("--normalise", choices=["host", "device"], default="host", help="where the per-subject z-score runs")
    args = ap.parse_args(argv)

    # Concurrent folds need their own hardware queues: with the runtime's default of 4, fifteen streams share four
    # queues and serialise (1649 -> 2914 train steps/s at 15 folds with 16 queues, tools/concurrency_probe.py).
    # Read by the HIP runtime when it initialises, so it has to be set before the first GPU call.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    world, rank, local_rank = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    # one process per GPU over RCCL; MSIG_DIST_BACKEND=gloo (and ranks sharing a GPU) only to rehearse on a 1-GPU box
    backend = os.environ.get("MSIG_DIST_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    cfg = default_cfg()
    cfg.update(epochs=args.epochs, patience=args.patience[0] if len(args.patience) == 1 else list(args.patience), batch_size=args.batch_size,
               verbose=args.verbose,
               concurrent_folds=args.concurrent_folds, normalise=args.normalise, eval_batch_size=args.eval_batch_size,
               lockstep=not args.no_lockstep, lockstep_groups=args.lockstep_groups, adaptive_forms=args.adaptive_forms)
    if args.synthetic is not None:
        from .synth import CHANNELS6, make_synthetic_wesad
        if rank == 0 and not (args.synthetic / "_channel_names.txt").exists():
            make_synthetic_wesad(args.synthetic, windows_per_subject=args.synthetic_windows, T=args.samples, difficulty=args.difficulty,
                                 window_spread=args.window_spread)
        if world > 1:
            dist.barrier(device_ids=[local_rank]) if backend == "nccl" else dist.barrier()
        cfg.update(data_path=args.synthetic, channels=list(CHANNELS6))
    elif args.data is not None:
        cfg.update(data_path=args.data)
    if args.channels:
        cfg.update(channels=args.channels)
    if args.subjects:
        cfg.update(subjects=args.subjects)
    torch.manual_seed(cfg["seed"])
    np.random.seed(cfg["seed"])
    stamp = datetime.now().strftime("%Y%m%d_%H%M%S")
    run_output_dir = args.out / RUN_NAME / f"run_{stamp}"
    if rank == 0:
        run_output_dir.mkdir(parents=True, exist_ok=True)
        print(f"====== 运行结果将保存至: {run_output_dir} ====== ({world} GPU(s))")
    if world > 1:
        box = [str(run_output_dir)]
        dist.broadcast_object_list(box, src=0)
        run_output_dir = Path(box[0])
    with open(Path(cfg["data_path"]) / "_channel_names.txt") as f:
        all_channel_names = [ln.strip() for ln in f if ln.strip()]
    cfg["gather_device"] = device if backend == "nccl" else torch.device("cpu")
    sets = None
    if args.ablation:
        sets = ablation_sets(all_channel_names)
    if args.sweep:
        sets = dict(sets or {})
        for item in args.sweep:
            name, _, chans = item.partition("=")
            if not name or not chans:
                ap.error(f"--sweep expects NAME=CH1,CH2,... (got {item!r})")
            sets[name] = chans.split(",")
    if args.hierarchical:
        results, wall = run_hierarchical_experiment(run_output_dir, device, all_channel_names, cfg, rank, world)
    elif sets:
        for n, ch in sets.items():
            if len(ch) > 16:
                ap.error(f"channel set {n!r} has {len(ch)} channels; the HIP path supports at most 16")
        results, wall = run_experiments(run_output_dir, device, all_channel_names,
                                        {n: dict(cfg, channels=list(ch)) for n, ch in sets.items()}, rank, world)
    else:
        results, wall = run_simple_experiment(run_output_dir, device, all_channel_names, cfg, rank, world)
    if world > 1:
        dist.destroy_process_group()
    return results, wall


if __name__ == "__main__":
    main()

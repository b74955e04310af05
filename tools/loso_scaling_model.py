#!/usr/bin/env python3
"""PREDICTED wall-clock of the bench's 15-fold synthetic LOSO on N = 1, 2, 4, 8 GPUs, from quantities a
1-GPU bench line already holds: the per-fold epoch counts, the B = 64 step time of one fold alone, and the 1-GPU wall-clock.

Model.  Folds are dealt round-robin to ranks (loso.folds_for_rank); a rank trains its folds in lockstep, a fold leaves when it
stops early.  With `a` folds active on a GPU one lockstep super-step (one train step of each active fold) takes
    s(a) = s1 * (1 + k * (a - 1))
— s1 = one fold alone (latency-bound: four 240-step recurrences, bench `b64.ms_per_step`); k = the stretch per additional
concurrent fold (shared command processor, bulk kernels that scale with the fold count, chip clocks), CALIBRATED so that the
model reproduces the measured 1-GPU wall-clock.  An epoch of a fold is `steps` train super-steps plus its evaluation pass
(`eval_ms`, one launch sequence per subject set at --eval-batch-size 1024); `fixed_s` is what a run pays once (reading and
uploading the dataset, interpreter start-up of the fold loop, the last fold's test pass and plot).

    python tools/loso_scaling_model.py gpurun_out/r04_bench_final.json
"""
import json
import sys


def rank_wall(epochs, s1_ms, k, steps, eval_ms):
    """Wall-clock (s) of one rank training folds with the given epoch counts in lockstep."""
    ep = sorted(epochs, reverse=True)
    t, done = 0.0, 0
    while ep:
        a = len(ep)
        span = ep[-1] - done                      # epochs until the next fold stops
        t += span * (steps * s1_ms * (1 + k * (a - 1)) + eval_ms * (1 + k * (a - 1)))
        done = ep[-1]
        while ep and ep[-1] == done:
            ep.pop()
    return t / 1e3


def main():
    d = json.load(open(sys.argv[1]))
    lo, b64 = d["loso"], d["b64"]
    epochs = lo["epochs_per_fold"]
    s1 = b64["ms_per_step"]
    steps = lo.get("train_steps_per_epoch", 47)
    eval_ms, fixed = lo.get("eval_ms_per_epoch", 2.0), lo.get("fixed_s", 1.0)
    wall1 = lo["wall_s"]
    lo_k, hi_k = 0.0, 1.0
    for _ in range(60):                            # calibrate k on the measured 1-GPU wall-clock
        k = 0.5 * (lo_k + hi_k)
        if fixed + rank_wall(epochs, s1, k, steps, eval_ms) > wall1:
            hi_k = k
        else:
            lo_k = k
    print(f"epochs per fold {epochs} (total {sum(epochs)}); s1 = {s1:.3f} ms; steps/epoch {steps}; eval {eval_ms} ms/epoch; fixed {fixed} s")
    print(f"calibrated stretch per additional concurrent fold k = {k:.4f}  (1-GPU wall {wall1:.2f} s reproduced)")
    emu = lo.get("emulated", {})
    print("| GPUs | folds per rank | predicted LOSO wall (s) | measured by rank emulation on one GPU (s) | speed-up vs 1 GPU (measured) | bound by |")
    print("|---|---|---|---|---|---|")
    for n in (1, 2, 4, 8):
        walls = [fixed + rank_wall(epochs[r::n], s1, k, steps, eval_ms) for r in range(n)]
        w = max(walls)
        r = walls.index(w)
        if n == 1:
            meas, sp = f"{wall1:.2f} (the run itself)", "1.00 x"
        elif str(n) in emu:
            e = emu[str(n)]
            meas = f"{e['wall_s']:.2f} (" + ", ".join(f"rank {q}: {v['wall_s']:.2f}" for q, v in e["ranks"].items()) + ")"
            sp = f"{wall1 / e['wall_s']:.2f} x"
        else:
            meas, sp = "-", "-"
        print(f"| {n} | {[len(epochs[q::n]) for q in range(n)]} | {w:.2f} | {meas} | {sp} | rank {r}: folds with {sorted(epochs[r::n], reverse=True)} epochs |")
    abl = lo.get("ablation")
    if abl and "epochs_per_fold" in abl:
        # 60 units dealt (fold-major, configuration-minor) round-robin: unit u = 4 * fold + set index -> rank u mod N (main.run_experiments)
        sets = list(abl["epochs_per_fold"])
        units = [abl["epochs_per_fold"][sn][f] for f in range(len(epochs)) for sn in sets]
        print(f"\nchannel-ablation sweep, {len(units)} units ({', '.join(sets)}): measured on 1 GPU {abl['wall_s']:.2f} s; same model (s1, k of the 6-channel LOSO):")
        print("| GPUs | units per rank | predicted wall (s) |")
        print("|---|---|---|")
        for n in (1, 2, 4, 8):
            walls = [fixed + rank_wall(units[r::n], s1, k, steps, eval_ms) for r in range(n)]
            print(f"| {n} | {[len(units[q::n]) for q in range(n)]} | {max(walls):.2f} |")
    floor = fixed + rank_wall([max(epochs)], s1, k, steps, eval_ms)
    print(f"floor (the longest fold alone on a GPU, {max(epochs)} epochs): {floor:.2f} s -> at most {wall1 / floor:.2f} x whatever the GPU count")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Where the wall-clock of the bench's 15-fold LOSO block goes on the HOST side: runs main.run_simple_experiment on the bench's
synthetic dataset under cProfile (every thread of the driver gets its own profiler) and prints the
heaviest functions by cumulative time, plus the per-fold epoch counts.

    python tools/profile_loso.py [--epochs N] [--groups G] [--eval-batch-size E] > gpurun_out/r04_loso_profile.log
"""
import argparse, cProfile, io, json, os, pstats, sys, threading, time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--groups", type=int, default=3)
    ap.add_argument("--eval-batch-size", type=int, default=None)
    ap.add_argument("--windows", type=int, default=270)
    ap.add_argument("--spread", type=int, default=20)
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--dir", type=Path, default=Path("/tmp/msig_bench_loso"))
    args = ap.parse_args()
    from multimodalsignal_amd import main as M
    from multimodalsignal_amd.synth import CHANNELS6, make_synthetic_wesad
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    data = args.dir / f"data_w{args.windows}_s{args.spread}_t3840_d2"
    if not (data / "_channel_names.txt").exists():
        make_synthetic_wesad(data, windows_per_subject=args.windows, T=3840, difficulty=2.0, window_spread=args.spread)
    names = (data / "_channel_names.txt").read_text().split()
    cfg = M.default_cfg()
    cfg.update(data_path=data, channels=list(CHANNELS6), epochs=args.epochs, lockstep_groups=args.groups, eval_batch_size=args.eval_batch_size,
               redeal_every=int(os.environ.get("MSIG_REDEAL", "-1")))        # -1: never re-deal the surviving folds (default); 0: once after the first round; n: every n epochs
    torch.zeros(1, device=dev)                      # runtime initialisation is not the LOSO's
    torch.cuda.synchronize()
    profs = []
    if not args.no_profile:
        orig_run = threading.Thread.run

        def run(self):                               # every thread of the driver (fold-batch workers, side streams) gets its own profiler
            p = cProfile.Profile(); profs.append(p); p.enable()
            try:
                orig_run(self)
            finally:
                p.disable()
        threading.Thread.run = run
        main_prof = cProfile.Profile(); main_prof.enable(); profs.append(main_prof)
    t0 = time.time()
    results, wall = M.run_simple_experiment(args.dir / f"prof_{os.getpid()}", dev, names, cfg)
    if not args.no_profile:
        for p in profs:
            p.disable()
    out = args.dir / f"prof_{os.getpid()}"
    infos = [json.loads(p.read_text()) for p in sorted(out.glob("fold_test_on_*/fold_result.json"))]
    print(f"wall {wall:.2f} s (outer {time.time() - t0:.2f}); epochs per fold {[i['epochs'] for i in infos]} total {sum(i['epochs'] for i in infos)}; "
          f"fold seconds {[round(i['seconds'], 2) for i in infos]}")
    if not args.no_profile:
        st = pstats.Stats(profs[0])
        for p in profs[1:]:
            st.add(p)
        s = io.StringIO()
        st.stream = s
        st.sort_stats("cumulative").print_stats(45)
        print(s.getvalue())
        s = io.StringIO(); st.stream = s
        st.sort_stats("tottime").print_stats(25)
        print(s.getvalue())


if __name__ == "__main__":
    main()

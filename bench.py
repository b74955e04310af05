#!/usr/bin/env python3
"""Headline benchmark: train windows/s of the CnnGruAttentionModel step on MI355X, the full 15-fold
LOSO wall-clock and its mean accuracy (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W          (N=1 default)

With N > 1 and no torch.distributed environment, bench.py starts the N ranks ITSELF (a
`python -m torch.distributed.run --nproc-per-node N ... bench.py ...` child, before anything touches
the GPU) and relays rank 0's JSON line; started under torchrun it is one of the ranks.  A launch whose
WORLD_SIZE differs from --gpus is an error, never a silent 1-GPU run.

A "step" is one full training step (zero_grad, forward, CrossEntropy, backward, Adam —
trainer.py:144-149) over one synthetic batch of (B, 6, 3840) fp32 windows that is already resident
in HBM; dropout (p=0.5) and BatchNorm batch statistics are active exactly as in `model.train()`.
Every rank trains an independent replica on its own GPU (the LOSO folds share nothing —
SURVEY.md §8e), so the aggregate is weak scaling.

Rank 0 prints ONE JSON line; see DESIGN.md §Measurement for every field.  Besides the contract's
fields: `roofline` (dominant contraction kernel), `cpu_baseline`, `kernels` (named as rocprofv3 names
them), `b64` (the same step at the reference's B = 64; `b64.fold_batch`: fifteen such models in one set of launches) and `loso` (synthetic 15-fold LOSO with the
reference's hyper-parameters, folds sharded over the ranks: wall_s, mean_acc, epochs_total).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2516.6   # 1024 SIMDs x 2.4 GHz x 1024 FLOP/clk (v_mfma_f32_16x16x32_bf16: 16384 FLOP / 16 cycles)
CLOCK_GHZ, N_SIMD = 2.4, 1024


def kernel_macs_per_window(C, T):
    """Algorithmic MACs each kernel launch performs per window (SURVEY.md §8d work model): the forward
    contraction it implements, or the dX / dW contraction of the backward.  Keys are the library's
    profile labels = the kernel rocprofv3 shows + the layer (gru_bwd_b3_l0 = gru_bwd_b3<32>)."""
    L1 = (T - 1) // 2 + 1
    P1 = (L1 - 1) // 2 + 1
    L2 = (P1 - 1) // 2 + 1
    TP = (L2 - 1) // 2 + 1
    conv1, conv2 = 16 * C * 7 * L1, 32 * 16 * 5 * L2
    cell0, cell1, rev1 = 192 * (32 + 64), 192 * (128 + 64), 192 * 128
    fwd0, fwd1 = 2 * TP * cell0, TP * cell1 + rev1
    bwd1, bwd0 = TP * (192 * 64 + 192 * 128 + cell1), 2 * TP * (192 * 64 + 192 * 32 + cell0)
    m = {
        "conv1_fwd": conv1, "conv2_fwd": conv2, "pool1_conv2_fwd": conv2,
        "gru_fwd_ws_l0": fwd0, "gru_fwd_ws_l1": fwd1, "gru_fwd_b3_l0": fwd0, "gru_fwd_b3_l1": fwd1, "gru_fwd_seq_l0": fwd0, "gru_fwd_seq_l1": fwd1,
        "gru_fwd_proj_l0": 2 * TP * 192 * 32, "gru_fwd_rec_l0": 2 * TP * 192 * 64,
        "gru_fwd_proj_l1": TP * 192 * 128 + rev1, "gru_fwd_rec_l1": TP * 192 * 64,
        "head_fwd": 64 * 128 + 2 * 64,
        # fused backward = recurrence (dh) + dX + dW contractions of the layer
        "gru_bwd_b3_l1": bwd1, "gru_bwd_b3_l0": bwd0, "gru_bwd_fused_l1": bwd1, "gru_bwd_fused_l0": bwd0,
        # split form (latency form / MSIG_GRU_BWD=split) and the single reverse step of the top layer
        "gru_bwd_seq_l1": TP * 192 * 64, "gru_bwd_seq_l0": 2 * TP * 192 * 64, "gru_bwd_seq_l1rev": 0,
        "gru_bwd_seq4_l1": TP * 192 * 64, "gru_bwd_seq4_l0": 2 * TP * 192 * 64, "gru_bwd_seq4_l1rev": 0,
        "gru_bwd_b4_l0": bwd0, "gru_bwd_b4_l1": bwd1, "gru_bwd_b5_l0": bwd0,
        "gru_bwd_b6_l0": bwd0,           # algorithmic work only: the W_hn h + b_hn it recomputes (2 TP 64 x 64 MACs) is not counted
        "gru_bwd_dx_l1": TP * 192 * 128 + rev1, "gru_bwd_dx_l1rev": rev1, "gru_bwd_dx_l0": 2 * TP * 192 * 32,   # dx_l1: the reverse step folded in
        "gru_bwd_dw_l1": TP * cell1, "gru_bwd_dw_l1rev": rev1, "gru_bwd_dw_l0": 2 * TP * cell0,
        # latency form since round 5: dX + dW of a layer in one launch (layer 1: both directions' dW and the reverse step's dX included)
        "gru_bwd_dxdw_l1": TP * 192 * 128 + rev1 + TP * cell1 + rev1, "gru_bwd_dxdw_l0": 2 * TP * 192 * 32 + 2 * TP * cell0,
        # ... and layer 1's dW rides in layer 0's recurrence launch (gru_bwd4.hip gru_bwd_seq4_dw1)
        "gru_bwd_seq4_l0+dw_l1": 2 * TP * 192 * 64 + TP * cell1 + rev1,
        "conv2_bwd": 2 * conv2, "conv1_bwd": conv1,           # conv2_bwd: dX and dW contractions in one kernel
        "head_bwd": 2 * (64 * 128 + 2 * 64),
        "head_step": 3 * (64 * 128 + 2 * 64),             # few windows: head forward + CrossEntropy + head backward in one launch
    }
    fwd = conv1 + conv2 + fwd0 + fwd1 + 64 * 128 + 2 * 64
    return m, fwd


# kernels whose every contraction runs as split-bf16 (six v_mfma_f32_16x16x32_bf16 per 16x16x32 block of MACs)
SPLIT_BF16 = {"gru_fwd_ws_l0", "gru_fwd_ws_l1", "gru_fwd_b3_l0", "gru_fwd_b3_l1", "gru_bwd_b3_l0", "gru_bwd_b3_l1", "gru_bwd_b4_l0", "gru_bwd_b4_l1",
              "gru_bwd_b5_l0", "gru_bwd_b6_l0", "gru_bwd_seq4_l0", "gru_bwd_seq4_l1", "gru_bwd_dxdw_l0", "gru_bwd_dxdw_l1", "gru_bwd_seq4_l0+dw_l1", "gru_bwd_dx_l1", "gru_bwd_dx_l0", "gru_bwd_dw_l0"}


def fold_batch_step_ms(dev, folds, T, steps):
    """ms per fused train step of `folds` models of B = 64 x (6, T) in one set of launches (synthetic inputs, random-init weights)."""
    import ctypes as C
    import torch
    from multimodalsignal_amd import _lib as L
    from multimodalsignal_amd.runtime import FoldArena
    ar = FoldArena(6, 2, dev, folds, 64, T)
    for s in range(folds):
        ar.engine(s).params.normal_(0, 0.05)
        ar.view(s, "x", torch.float32).normal_()
        ar.view(s, "y", torch.int64).random_(0, 2)
    m = ar.multi(list(range(folds)), [1] * folds, [2] * folds, [1e-3] * folds)
    desc = ar.batch(64, True, 0.5)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def run(n, k0):
        for k in range(n):
            L.check(L.lib().msig_train_step_multi(C.byref(desc), C.byref(m), ar.ptr("exp_avg"), ar.ptr("exp_avg_sq"), 0.9, 0.999, 1e-8, 1e-4, k0 + k + 1, st),
                    "msig_train_step_multi")
        torch.cuda.synchronize(dev)
    run(10, 0)
    t0 = time.perf_counter()
    run(steps, 10)
    dt = time.perf_counter() - t0
    return {"folds": folds, "batch": 64, "steps": steps, "ms_per_step": round(1e3 * dt / steps, 4),
            "value": round(folds * 64 * steps / dt, 1), "unit": "windows/s per GPU"}


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def lib_sha16():
    p = ROOT / "multimodalsignal_amd" / "libmsig_hip.so"
    return hashlib.sha256(p.read_bytes()).hexdigest()[:16] if p.exists() else None


def src_sha16():
    """sha256 over the kernel sources the library is built from (csrc/*.hip, *.h, Makefile, include/msig.h).  The binary's own hash
    is not reproducible across build directories (hipcc embeds paths), the sources' is: the PMC traffic file is matched on this."""
    h = hashlib.sha256()
    csrc = ROOT / "multimodalsignal_amd" / "csrc"
    for f in sorted([*csrc.glob("*.hip"), *csrc.glob("*.h"), csrc / "Makefile", ROOT / "include" / "msig.h"]):
        h.update(f.name.encode()); h.update(f.read_bytes())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps of the headline leg (100 x 8 ms: box-to-box noise is +-3 %%)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--long-steps", type=int, default=100, help="rank 0: a second, longer timed run reported as `long_run` when --steps is shorter (0 disables)")
    ap.add_argument("--batch", type=int, default=8192, help="windows per step per GPU (64 = the reference's training batch)")
    ap.add_argument("--channels", type=int, default=6)
    ap.add_argument("--samples", type=int, default=3840, help="samples per window (60 s @ 64 Hz)")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU-baseline work (0 disables)")
    ap.add_argument("--b64-steps", type=int, default=300, help="timed steps of the B=64 block (0 disables)")
    ap.add_argument("--loso", type=int, default=1, help="1: run the synthetic 15-fold LOSO block (folds sharded over the ranks)")
    ap.add_argument("--loso-windows", type=int, default=270, help="mean windows per subject of the synthetic LOSO set")
    ap.add_argument("--loso-spread", type=int, default=20,
                    help="subjects get --loso-windows +- this many windows (WESAD recordings differ in length; 0 = equal sizes)")
    ap.add_argument("--loso-epochs", type=int, default=None, help="epoch budget of the LOSO block (default: the reference's 100)")
    ap.add_argument("--loso-dir", type=Path, default=Path("/tmp/msig_bench_loso"))
    ap.add_argument("--emulate-ranks", type=str, default="2,4,8",
                    help="1-GPU runs only: for each N listed, play single ranks of an N-GPU LOSO job alone on this GPU (the rank that holds "
                         "the longest fold and its neighbour; both ranks at N = 2) and report their wall-clock as loso.emulated — what "
                         "that rank would execute on an N-GPU node, less one ~100-byte all_gather ('' disables)")
    ap.add_argument("--ablation", type=int, default=1,
                    help="1: run the channel-ablation sweep {ECG-only, EDA-only, chest-only, wrist-only} x 15 folds = 60 units as one job (BASELINE configs[3])")
    ap.add_argument("--dry-run", action="store_true",
                    help="plumbing rehearsal without a GPU: spawns/joins the ranks and prints the line with value null (tests only)")
    args = ap.parse_args()

    # ---- rank launch: before anything touches the GPU (a process that has initialised HIP must not exec/fork workers) ----
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(Path(__file__).resolve())] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        rc = subprocess.run(cmd, env=env).returncode
        if rc != 0:
            print(f"bench.py: the {args.gpus}-rank launch failed (exit {rc}); nothing was measured", file=sys.stderr)
        sys.exit(rc)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as {args.gpus} GPUs")
    # fifteen concurrent folds need their own hardware queues; read by the HIP runtime when it initialises
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

    import torch
    import torch.distributed as dist
    # one process per GPU; MSIG_DIST_BACKEND=gloo (+ fewer GPUs than ranks) is only for rehearsing the
    # multi-rank path on a single-GPU box or (with --dry-run) on a CPU-only one
    backend = os.environ.get("MSIG_DIST_BACKEND", "nccl")
    if args.dry_run:
        if world > 1:
            dist.init_process_group("gloo")
        n_seen = dist.get_world_size() if world > 1 else 1
        t = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
        if world > 1:
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"metric": "train windows/sec", "value": None, "unit": "windows/s", "n_gpus": n_seen, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                              "dtype": "f32", "data": "synthetic", "dry_run": True, "config": {"workload": "none (plumbing rehearsal)"}}))
        if world > 1:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    from multimodalsignal_amd import _lib as L
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)
    n_seen = dist.get_world_size() if world > 1 else 1       # what the process group actually spans

    B, C, T, K = args.batch, args.channels, args.samples, 2
    # random-init weights of the reference architecture (torch default initialisers)
    from multimodalsignal_amd.models import CnnGruAttentionModel
    torch.manual_seed(42 + rank)
    model = CnnGruAttentionModel(C, K).to(dev)
    eng = model.engine()
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn(B, C, T, device=dev, generator=gen)
    y = (torch.rand(B, device=dev, generator=gen) < 0.2).to(torch.int64)

    def step(i, xx=x, yy=y):
        eng.train_step(xx, yy, lr=1e-3, weight_decay=1e-4, step=i, dropout_p=0.5, seed=99)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            if backend == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(v):
        if world == 1:
            return v
        t = torch.tensor([v], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(k, collective=True):
        """k steps between barrier + synchronize pairs (collective=False: this rank alone, synchronize only); per-step durations
        from HIP events on the launch stream (torch's current stream is the stream every kernel of the step is launched on),
        recorded without any host sync inside the region."""
        nonlocal n
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
        sync = barrier if collective else (lambda: torch.cuda.synchronize(dev))
        sync()
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(k):
            n += 1
            step(n)
            ev[i + 1].record()
        sync()
        dt = time.perf_counter() - t0
        per = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(k))
        return dt, per

    def spread(per):
        return {"min": round(per[0], 4), "median": round(per[len(per) // 2], 4), "max": round(per[-1], 4)}

    n = 0
    for _ in range(args.warmup):
        n += 1
        step(n)
    dt, step_ms = timed(args.steps)
    elapsed = max_over_ranks(dt)
    loss_last = float(eng.region("LOSS")[0])
    long_run = None
    if rank == 0 and 0 < args.steps < args.long_steps:
        # the contract's K steps can be a 0.17 s region (K = 20): kernel changes of 1-3 % drown in it, so rank 0 times a longer one too
        dtl, perl = timed(args.long_steps, collective=False)
        long_run = {"steps": args.long_steps, "value_per_gpu": round(B * args.long_steps / dtl, 1), "ms_per_step": round(1e3 * dtl / args.long_steps, 3),
                    "ms_per_step_spread": spread(perl), "ranks": "rank 0 alone (not the contract's max-over-ranks figure)"}

    # ---- per-kernel HIP-event timing (own pass: the events add bubbles) --------------------
    roofline, kernels = None, {}
    macs, fwd_macs = kernel_macs_per_window(C, T)
    if rank == 0 and args.profile_steps > 0:
        L.profile_enable(True)
        for _ in range(args.profile_steps):
            n += 1
            step(n)
        torch.cuda.synchronize(dev)
        rep = L.profile_report()
        L.profile_enable(False)
        total_ms = sum(ms for _, ms in rep.values())
        for name, (cnt, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1]):
            per_step = ms / args.profile_steps
            ent = {"ms_per_step": round(per_step, 4), "launches_per_step": cnt / args.profile_steps}
            if macs.get(name, 0) > 0:
                ent["tflops"] = round(2.0 * macs[name] * B / (per_step * 1e-3) / 1e12, 2)
            kernels[name] = ent
        dom = next(k for k in kernels if macs.get(k, 0) > 0)      # slowest kernel with a contraction
        dom_ms = kernels[dom]["ms_per_step"] / max(kernels[dom]["launches_per_step"], 1)
        flop = 2.0 * macs[dom] * B
        ach = flop / (dom_ms * 1e-3) / 1e12
        # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (not live: PMC needs the profiler).
        # The file records the sha256 of the kernel sources (and of the libmsig_hip.so) it was measured on: other sources, shape or
        # kernel -> null.
        traffic, traffic_src = None, None
        for tf in sorted((ROOT / "profiles").glob("r*_pmc_traffic.json"), reverse=True):
            tj = json.loads(tf.read_text())
            if (tj.get("config") == {"batch": B, "channels": C, "samples": T} and dom in tj.get("kernels", {})
                    and (tj.get("src_sha16") == src_sha16() or tj.get("lib_sha16") == lib_sha16())):
                traffic, traffic_src = tj["kernels"][dom]["bytes_per_launch"], tf.name
                break
        # The bound of a kernel that contracts on split-bf16 is the bf16 matrix pipe at SIX instructions per block of
        # algorithmic MACs (dense bf16 peak / 6 = 419 TFLOP/s of fp32-equivalent work): `peak`/`frac` use it, so that frac is
        # the matrix-pipe occupancy (MFMA issue cycles / (SIMDs x duration x 2.4 GHz)); the fp32-MFMA peak that BASELINE's
        # fp32 path is priced against stays as `frac_fp32_yardstick` (a yardstick such a kernel can exceed, not a bound).
        peak = PEAK_BF16_MFMA_TFLOPS / 6 if dom in SPLIT_BF16 else PEAK_F32_MFMA_TFLOPS
        roofline = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1),
                    "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "avg_launch_ms": round(dom_ms, 4), "flop_per_launch": flop,
                    "frac_fp32_yardstick": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "peak_fp32_mfma": PEAK_F32_MFMA_TFLOPS,
                    "sum_kernel_ms_per_step": round(total_ms / args.profile_steps, 3)}
        if dom in SPLIT_BF16:
            mfma_cycles = macs[dom] * B / 8192.0 * 6 * 16        # six v_mfma_f32_16x16x32_bf16 of 16 cycles per 8192 MACs
            roofline["frac_matrix_pipe"] = round(mfma_cycles / (N_SIMD * dom_ms * 1e-3 * CLOCK_GHZ * 1e9), 4)
            roofline["arithmetic"] = "split-bf16: 3 bf16 pieces per fp32 operand, 6 cross products per block, fp32 accumulate"

    # ---- the same step at the reference's batch size (main.py:63 BATCH_SIZE = 64): latency-bound regime of configs[1..3] ----
    b64 = None
    if args.b64_steps > 0 and rank == 0:
        xs, ys = x[:64].contiguous(), y[:64].contiguous()
        for i in range(30):
            step(n + i + 1, xs, ys)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for i in range(args.b64_steps):
            step(n + 31 + i, xs, ys)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t1
        L.profile_enable(True)
        step(n + 40 + args.b64_steps, xs, ys)
        torch.cuda.synchronize(dev)
        launches = sum(c for c, _ in L.profile_report().values())
        L.profile_enable(False)
        b64 = {"value": round(64 * args.b64_steps / dt, 1), "unit": "windows/s per GPU", "ms_per_step": round(1e3 * dt / args.b64_steps, 4),
               "steps": args.b64_steps, "launches_per_step": launches, "batch": 64}

        # ... and fifteen such steps — fifteen independent models, the LOSO's folds — as ONE fold batch (msig_train_step_multi:
        # every launch covers all of them), the regime of the LOSO's first epochs
        b64["fold_batch"] = fold_batch_step_ms(dev, 15, T, max(50, args.b64_steps // 2))

    # ---- synthetic 15-fold LOSO with the reference's hyper-parameters (main.py:48-67), folds sharded over the ranks ----
    loso = None
    if args.loso:
        del x, y
        eng.drop_workspaces()
        torch.cuda.empty_cache()
        from multimodalsignal_amd import main as M
        from multimodalsignal_amd.synth import CHANNELS6, make_synthetic_wesad
        data = args.loso_dir / f"data_w{args.loso_windows}_s{args.loso_spread}_t{T}_d2"
        if rank == 0 and not (data / "_channel_names.txt").exists():
            make_synthetic_wesad(data, windows_per_subject=args.loso_windows, T=T, difficulty=2.0, window_spread=args.loso_spread)
        barrier()
        names = (data / "_channel_names.txt").read_text().split()
        cfg = M.default_cfg()
        cfg.update(data_path=data, channels=list(CHANNELS6), gather_device=dev if backend == "nccl" else torch.device("cpu"))
        if args.loso_epochs:
            cfg.update(epochs=args.loso_epochs)
        out = args.loso_dir / f"run_{os.getpid() if world == 1 else os.environ.get('MASTER_PORT', '0')}"
        torch.manual_seed(cfg["seed"])
        barrier()
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):         # the per-fold progress lines: stdout carries the JSON line only
            results, wall = M.run_simple_experiment(out, dev, names, cfg, rank, world)
        wall = max_over_ranks(wall)
        if rank == 0:
            import numpy as np
            from multimodalsignal_amd.synth import subject_window_counts
            counts = subject_window_counts(15, args.loso_windows, args.loso_spread)
            infos = [json.loads(p.read_text()) for p in sorted(out.glob("fold_test_on_*/fold_result.json"))]
            by_subject = {i["subject"]: i for i in infos}
            loso = {"wall_s": round(wall, 2), "mean_acc": round(float(np.mean([r["accuracy"] for r in results])), 4),
                    "std_acc": round(float(np.std([r["accuracy"] for r in results])), 4),
                    "mean_f1": round(float(np.mean([r["f1_score"] for r in results])), 4),
                    "folds": len(results), "folds_per_rank": [len(range(r, len(results), world)) for r in range(world)],
                    "epochs_total": int(sum(i["epochs"] for i in infos)),
                    # the early-stopping rule of the reference makes the epoch count a chaotic function of last-bit rounding (DESIGN.md
                    # section 2): wall_s of two builds is comparable only together with this rate
                    "fold_epochs_per_s": round(float(sum(i["epochs"] for i in infos)) / wall, 2),
                    # in subject order; with seconds_per_fold (training time of a fold until its stop, inside its lockstep batch) the
                    # inputs of tools/loso_scaling_model.py, whose predicted wall-clock at N = 1, 2, 4, 8 is in DESIGN.md section 6
                    "epochs_per_fold": [int(by_subject[r["subject"]]["epochs"]) for r in results],
                    "seconds_per_fold": [round(float(by_subject[r["subject"]]["seconds"]), 2) for r in results],
                    "fixed_s": round(wall - max(float(i["seconds"]) for i in infos), 2),
                    "train_steps_per_epoch": int(round(float(np.mean([-(-(sum(counts) - c - 3 * args.loso_windows) // cfg["batch_size"]) for c in counts])))),
                    "eval_batch": int(cfg.get("eval_batch_size") or cfg["batch_size"]),
                    "fold_execution": f"lockstep fold batches (msig_train_step_multi), up to {cfg.get('lockstep_groups', 4)} per rank on separate "
                                      "streams; GRU kernel forms pinned per run (a fold's results do not depend on grouping or rank count)",
                    "data": f"synthetic WESAD-shaped, 15 subjects x {args.loso_windows} +- {args.loso_spread} windows "
                            f"({min(counts)}..{max(counts)}) x (6 ch, {T} samples), difficulty 2",
                    "hyper": {"batch": cfg["batch_size"], "epochs": cfg["epochs"], "patience": cfg["patience"], "lr": cfg["lr"],
                              "weight_decay": cfg["weight_decay"], "dropout": cfg["model_params"]["dropout"]},
                    "includes": "load + normalise + upload of the dataset, training, evaluation, metric gather"}
        # ---- N-GPU LOSO by rank emulation (1-GPU runs): a rank of an N-GPU job trains exactly the folds k = r (mod N) and shares its GPU
        #      with nobody, so running that rank ALONE here measures its wall-clock; the job's wall-clock is the slowest rank's ----
        if world == 1 and args.emulate_ranks:
            ep = loso["epochs_per_fold"]
            longest = max(range(len(ep)), key=lambda k: ep[k])
            emu = {}
            for N in [int(v) for v in args.emulate_ranks.split(",") if v.strip()]:
                ranks = sorted({longest % N, (longest + 1) % N}) if N > 2 else list(range(N))
                per = {}
                for r in ranks:
                    torch.manual_seed(cfg["seed"])
                    with contextlib.redirect_stdout(sys.stderr):
                        res_r, wall_r = M.run_simple_experiment(args.loso_dir / f"emu_{os.getpid()}_{N}_{r}", dev, names, dict(cfg, emulate_rank=True), r, N)
                    per[str(r)] = {"wall_s": round(wall_r, 2), "folds": len(res_r)}
                emu[str(N)] = {"ranks": per, "wall_s": max(v["wall_s"] for v in per.values()),
                               "bound_rank_expected": longest % N, "ranks_not_run": N - len(ranks)}
            loso["emulated"] = emu
            loso["emulated_note"] = ("each listed rank of an N-GPU job run ALONE on this GPU (its folds, its data load, no gather): the N-GPU "
                                     "wall-clock is the slowest rank's; for N > 2 only the rank holding the longest fold and its neighbour were run")
        # ---- channel-ablation sweep (BASELINE configs[3]): 4 channel sets x 15 folds = 60 work units as ONE job ----
        if args.ablation:
            sets = M.ablation_sets(names)
            torch.manual_seed(cfg["seed"])
            barrier()
            with contextlib.redirect_stdout(sys.stderr):
                abl_dir = args.loso_dir / f"abl_{os.getpid() if world == 1 else os.environ.get('MASTER_PORT', '0')}"
                res_a, wall_a = M.run_experiments(abl_dir, dev, names,
                                                  {n: dict(cfg, channels=list(ch)) for n, ch in sets.items()}, rank, world)
            wall_a = max_over_ranks(wall_a)
            if rank == 0:
                loso_abl = {"wall_s": round(wall_a, 2), "units": int(sum(len(v) for v in res_a.values())), "sets": {n: len(ch) for n, ch in sets.items()},
                            "mean_acc": {n: round(float(np.mean([r["accuracy"] for r in v])), 4) for n, v in res_a.items()},
                            "epochs_per_fold": {n: [int(json.loads((abl_dir / n / f"fold_test_on_{r['subject']}" / "fold_result.json").read_text())["epochs"])
                                                    for r in v] for n, v in res_a.items()},
                            "execution": "one fold batch (msig_train_step_multi) per channel set, the four sets concurrently on four streams"}
                loso["ablation"] = loso_abl

    cpu = None
    if rank == 0 and args.cpu_budget > 0:          # every N: rank 0, after the GPU sections (the other ranks wait at the last barrier)
        from oracle.cpu_model import time_train_steps      # reported baseline only; never the product path
        r = time_train_steps(batch=64, C=C, T=T, K=K, budget_s=args.cpu_budget)
        r256 = time_train_steps(batch=256, C=C, T=T, K=K, budget_s=6.0, min_steps=4, threads=r["threads"])      # ~10 steps of 0.6 s (round 3: two)
        phys = None
        try:
            pairs = set()
            pid = cid = None
            for ln in open("/proc/cpuinfo"):
                if ln.startswith("physical id"):
                    pid = ln.split(":")[1].strip()
                elif ln.startswith("core id"):
                    cid = ln.split(":")[1].strip()
                elif not ln.strip():
                    if pid is not None and cid is not None:
                        pairs.add((pid, cid))
                    pid = cid = None
            phys = len(pairs) or None
        except OSError:
            pass
        cpu = {"value": round(r["value"], 2), "unit": "windows/s", "cores": r["threads"], "kind": "port",
               "sample": f"{r['steps']} train steps of B=64 x ({C},{T}) on torch-CPU (stock nn modules, reference module graph), "
                         f"{r['ms_per_step']:.0f} ms/step; thread count calibrated over {{8,16,32}} (more threads are slower on this model)",
               "host_logical_cpus": os.cpu_count(), "host_physical_cores": phys,
               "b256": {"value": round(r256["value"], 2), "steps": r256["steps"], "ms_per_step": round(r256["ms_per_step"], 1)}}

    if rank == 0:
        value = n_seen * B * args.steps / elapsed
        train_flop = 3 * 2.0 * fwd_macs
        out = {
            "metric": "train windows/sec", "value": round(value, 1), "unit": "windows/s", "n_gpus": n_seen,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "ms_per_step_spread": spread(step_ms), "long_run": long_run,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"CnnGruAttentionModel full train step (fwd+CE+bwd+Adam, dropout 0.5, BN batch stats), "
                                   f"B={B} windows/GPU x ({C} ch, {T} samples = 60 s @ 64 Hz), random-init weights; "
                                   "BASELINE.json configs[4] shape, the per-step work of configs[1]",
                       "batch_per_gpu": B, "channels": C, "samples": T, "classes": K, "parallelism": f"replica x{n_seen}",
                       "arithmetic": "fp32 throughout; the GRU contractions run as split products with fp32 accumulation: backward recurrence, dX, dW and "
                                     "the layer-0 input projection as split-bf16 MFMA (three bf16 pieces per fp32 operand, six cross products), the "
                                     "forward recurrences and the layer-1 projection as two-piece fp16 MFMA (pre-scaled operands, three cross products) "
                                     "- error <= the fp32 MFMA chain's (profiles/r01_bf16x3_microbench.log, r05_f16x2_microbench.log); the "
                                     "convolutions and the head on fp32 MFMA"},
            "step_mfma_frac": round(value / n_seen * train_flop / (PEAK_F32_MFMA_TFLOPS * 1e12), 4),
            "train_mflop_per_window": round(train_flop / 1e6, 2),
            "loss_last": round(loss_last, 5), "lib_sha16": lib_sha16(), "src_sha16": src_sha16(),
            "roofline": roofline, "cpu_baseline": cpu, "b64": b64, "loso": loso, "kernels": kernels,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()                      # ranks > 0 wait here while rank 0 times the CPU baseline
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: train windows/s of the CnnGruAttentionModel step on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N=1 default)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full training step (zero_grad, forward, CrossEntropy, backward, Adam —
trainer.py:144-149) over one synthetic batch of (B, 6, 3840) fp32 windows that is already
resident in HBM; dropout (p=0.5) and BatchNorm batch statistics are active exactly as in
`model.train()`.  With N > 1 every rank trains an independent replica on its own GPU (the
LOSO folds share nothing — SURVEY.md §8e), so the aggregate is weak scaling.

Rank 0 prints ONE JSON line; see DESIGN.md §Measurement for every field.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
TRAIN_MFLOP_PER_WINDOW = None    # filled from the per-kernel table below


def kernel_macs_per_window(C, T):
    """Algorithmic MACs each kernel launch performs per window (SURVEY.md §8d work model):
    the forward contraction it implements, or the dX / dW contraction of the backward."""
    L1 = (T - 1) // 2 + 1
    P1 = (L1 - 1) // 2 + 1
    L2 = (P1 - 1) // 2 + 1
    TP = (L2 - 1) // 2 + 1
    conv1, conv2 = 16 * C * 7 * L1, 32 * 16 * 5 * L2
    cell0, cell1, rev1 = 192 * (32 + 64), 192 * (128 + 64), 192 * 128
    m = {
        "conv1_fwd": conv1, "conv2_fwd": conv2,
        "gru_fwd_seq_l0": 2 * TP * cell0, "gru_fwd_seq_l1": TP * cell1 + rev1,
        "head_fwd": 64 * 128 + 2 * 64,
        # fused backward = recurrence (dh) + dX + dW contractions of the layer
        "gru_bwd_fused_l1": TP * (192 * 64 + 192 * 128 + cell1), "gru_bwd_fused_l0": 2 * TP * (192 * 64 + 192 * 32 + cell0),
        # split fallback (MSIG_GRU_BWD=split) and the single reverse step of the top layer
        "gru_bwd_seq_l1": TP * 192 * 64, "gru_bwd_seq_l0": 2 * TP * 192 * 64, "gru_bwd_seq_l1rev": 0,
        "gru_bwd_dx_l1": TP * 192 * 128, "gru_bwd_dx_l1rev": rev1, "gru_bwd_dx_l0": 2 * TP * 192 * 32,
        "gru_bwd_dw_l1": TP * cell1, "gru_bwd_dw_l1rev": rev1, "gru_bwd_dw_l0": 2 * TP * cell0,
        "conv2_bwd_dx": conv2, "conv2_bwd_dw": conv2, "conv1_bwd": conv1,
        "head_bwd": 2 * (64 * 128 + 2 * 64),
    }
    fwd = conv1 + conv2 + 2 * TP * cell0 + TP * cell1 + rev1 + 64 * 128 + 2 * 64
    return m, fwd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8192, help="windows per step per GPU (64 = the reference's training batch)")
    ap.add_argument("--channels", type=int, default=6)
    ap.add_argument("--samples", type=int, default=3840, help="samples per window (60 s @ 64 Hz)")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU-baseline work (0 disables)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from multimodalsignal_amd import _lib as L
    from multimodalsignal_amd.runtime import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # one process per GPU; MSIG_DIST_BACKEND=gloo (+ fewer GPUs than ranks) is only for rehearsing the
    # multi-rank path on a single-GPU box
    backend = os.environ.get("MSIG_DIST_BACKEND", "nccl")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    B, C, T, K = args.batch, args.channels, args.samples, 2
    # random-init weights of the reference architecture (torch default initialisers)
    from multimodalsignal_amd.models import CnnGruAttentionModel
    torch.manual_seed(42 + rank)
    model = CnnGruAttentionModel(C, K).to(dev)
    eng = model.engine()
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn(B, C, T, device=dev, generator=gen)
    y = (torch.rand(B, device=dev, generator=gen) < 0.2).to(torch.int64)

    def step(i):
        eng.train_step(x, y, lr=1e-3, weight_decay=1e-4, step=i, dropout_p=0.5, seed=99)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            if backend == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()
        torch.cuda.synchronize(dev)

    n = 0
    for _ in range(args.warmup):
        n += 1
        step(n)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n += 1
        step(n)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    loss_last = float(eng.region("LOSS")[0])

    # ---- per-kernel HIP-event timing (own pass: the events add bubbles) --------------------
    roofline, kernels = None, {}
    if rank == 0 and args.profile_steps > 0:
        L.profile_enable(True)
        for _ in range(args.profile_steps):
            n += 1
            step(n)
        torch.cuda.synchronize(dev)
        rep = L.profile_report()
        L.profile_enable(False)
        macs, fwd_macs = kernel_macs_per_window(C, T)
        total_ms = sum(ms for _, ms in rep.values())
        for name, (cnt, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1]):
            per_step = ms / args.profile_steps
            ent = {"ms_per_step": round(per_step, 4), "launches_per_step": cnt / args.profile_steps}
            if macs.get(name, 0) > 0:
                ent["tflops"] = round(2.0 * macs[name] * B / (per_step * 1e-3) / 1e12, 2)
            kernels[name] = ent
        dom = next(k for k in kernels if macs.get(k, 0) > 0)      # slowest kernel with a contraction
        dom_ms = kernels[dom]["ms_per_step"] / max(kernels[dom]["launches_per_step"], 1)
        ach = 2.0 * macs[dom] * B / (dom_ms * 1e-3) / 1e12
        # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (not live: PMC needs the
        # profiler); only quoted when the file was taken at this exact shape.
        traffic = None
        tf = ROOT / "profiles" / "r01_pmc_traffic.json"
        if tf.exists():
            tj = json.loads(tf.read_text())
            if tj.get("config") == {"batch": B, "channels": C, "samples": T} and dom in tj.get("kernels", {}):
                traffic = tj["kernels"][dom]["bytes_per_launch"]
        roofline = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                    "avg_launch_ms": round(dom_ms, 4), "flop_per_launch": 2.0 * macs[dom] * B,
                    "sum_kernel_ms_per_step": round(total_ms / args.profile_steps, 3)}

    cpu = None
    if rank == 0 and world == 1 and args.cpu_budget > 0:
        from oracle.cpu_model import time_train_steps      # reported baseline only; never the product path
        r = time_train_steps(batch=64, C=C, T=T, K=K, budget_s=args.cpu_budget)
        cpu = {"value": round(r["value"], 2), "unit": "windows/s", "cores": r["threads"], "kind": "port",
               "sample": f"{r['steps']} train steps of B=64 x ({C},{T}) on torch-CPU (stock nn modules, reference module graph), "
                         f"{r['ms_per_step']:.0f} ms/step"}

    if rank == 0:
        macs, fwd_macs = kernel_macs_per_window(C, T)
        value = world * B * args.steps / elapsed
        train_flop = 3 * 2.0 * fwd_macs
        out = {
            "metric": "train windows/sec", "value": round(value, 1), "unit": "windows/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"CnnGruAttentionModel full train step (fwd+CE+bwd+Adam, dropout 0.5, BN batch stats), "
                                   f"B={B} windows/GPU x ({C} ch, {T} samples = 60 s @ 64 Hz), random-init weights; "
                                   "BASELINE.json configs[4] shape, the per-step work of configs[1]",
                       "batch_per_gpu": B, "channels": C, "samples": T, "classes": K, "parallelism": f"replica x{world}",
                       "arithmetic": "fp32 throughout; the GRU forward and the layer-0 backward recurrence/dX contract on split-bf16 MFMA "
                                     "(three bf16 pieces per fp32 operand, six cross products, fp32 accumulate: error <= the fp32 MFMA chain's, "
                                     "profiles/r01_bf16x3_microbench.log), every other contraction on fp32 MFMA"},
            "step_mfma_frac": round(value / world * train_flop / (PEAK_F32_MFMA_TFLOPS * 1e12), 4),
            "train_mflop_per_window": round(train_flop / 1e6, 2),
            "loss_last": round(loss_last, 5),
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

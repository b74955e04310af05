#!/usr/bin/env python3
"""Per-tensor HBM byte ledger of one train step: who writes and who re-reads every large tensor of the workspace, summed per
kernel and set against the measured traffic (a profiles/r*_pmc_traffic.json made by tools/pmc_traffic.py).

    python tools/byte_ledger.py profiles/r04_pmc_traffic.json > profiles/r04_byte_ledger.md

Sizes follow include/msig.h's workspace regions for the json's shape (B windows x C channels x T samples).  What the table answers:
is a kernel's measured traffic its algorithmic bytes (every operand once), and which tensors cross HBM more than once per step.
"""
import json
import sys

doc = json.load(open(sys.argv[1]))
B, C, T = doc["config"]["batch"], doc["config"]["channels"], doc["config"]["samples"]
L1 = (T + 6 - 7) // 2 + 1; P1 = (L1 + 2 - 3) // 2 + 1; L2 = (P1 + 4 - 5) // 2 + 1; TP = (L2 + 2 - 3) // 2 + 1
NT = (B + 15) // 16
F = 4
vec = NT * TP * 4 * 64 * 16          # one stash vector of one direction: (tile, step, wave, lane) float4
t = {  # tensor -> bytes
    "x": B * C * T * F, "y1": B * L1 * 16 * F, "p1": B * P1 * 16 * F, "y2": B * L2 * 32 * F, "p2": B * TP * 32 * F,
    "H0": B * TP * 128 * F, "H1": B * TP * 64 * F, "stash0 (r, z; 2 dirs)": 2 * 2 * vec, "stash1 (r, z, hn)": 3 * vec,
    "dH0": B * TP * 128 * F, "dX0 (2 dirs)": 2 * B * TP * 32 * F, "dz2": B * L2 * 32 * F, "dP1": B * P1 * 16 * F,
    "poolc1": B * P1 * 4, "poolc2": B * TP * 8, "G1W": B * 32 * ((C * 7 + 15) // 16 * 16) * F,
}
# kernel -> (reads, writes); a tensor listed twice is read by both directions
K = [
    ("gate", ["x"], []),
    ("conv1_fwd", ["x"], ["y1"]),
    ("pool1_conv2_fwd", ["y1"], ["p1", "poolc1", "y2"]),
    ("bn_relu_pool_32", ["y2"], ["p2", "poolc2"]),
    ("gru_fwd_ws_l0", ["p2", "p2"], ["H0", "stash0 (r, z; 2 dirs)"]),
    ("gru_fwd_ws_l1", ["H0"], ["H1", "stash1 (r, z, hn)"]),
    ("gru_bwd_b3_l1", ["stash1 (r, z, hn)", "H1", "H0"], ["dH0"]),
    ("gru_bwd_b6_l0", ["stash0 (r, z; 2 dirs)", "H0", "dH0", "p2", "p2"], ["dX0 (2 dirs)"]),
    ("pool_bn_bwd_pass1_32", ["dX0 (2 dirs)", "poolc2", "y2"], ["dz2"]),
    ("conv2_bwd", ["dz2", "y2", "p1"], ["dP1"]),
    ("conv1_bwd", ["dP1", "poolc1", "y1", "x"], ["G1W"]),
]
meas = {k: v["bytes_per_launch"] for k, v in doc["kernels"].items()}
print(f"# HBM byte ledger of one train step, B = {B} x ({C}, {T})  (measured: {sys.argv[1]}, src_sha16 {doc.get('src_sha16')})\n")
print("| kernel | reads | writes | ledger GB | measured GB | measured / ledger |")
print("|---|---|---|---|---|---|")
tot_l = tot_m = 0.0
uses = {}
for name, rd, wr in K:
    lb = sum(t[x] for x in rd) + sum(t[x] for x in wr)
    m = meas.get(name)
    tot_l += lb; tot_m += m or 0.0
    for x in rd:
        uses.setdefault(x, [[], []])[0].append(name)
    for x in wr:
        uses.setdefault(x, [[], []])[1].append(name)
    fmt = lambda xs: ", ".join(f"{x} {t[x] / 1e9:.2f}" for x in xs) or "—"
    print(f"| `{name}` | {fmt(rd)} | {fmt(wr)} | {lb / 1e9:.2f} | {m / 1e9:.2f} | {m / lb:.2f} |" if m else f"| `{name}` | {fmt(rd)} | {fmt(wr)} | {lb / 1e9:.2f} | n/a | |")
rest = sum(v for k, v in meas.items() if k not in {n for n, _, _ in K})
print(f"| the other launches (head, reductions, Adam, finalize) | | | | {rest / 1e9:.2f} | |")
print(f"| **step** | | | **{tot_l / 1e9:.2f}** | **{(tot_m + rest) / 1e9:.2f}** | {(tot_m + rest) / tot_l:.2f} |")
print("\n| tensor | GB | written by | read by | crossings of HBM per step |")
print("|---|---|---|---|---|")
for x, (rd, wr) in sorted(uses.items(), key=lambda kv: -t[kv[0]] * (len(kv[1][0]) + len(kv[1][1]))):
    n = len(rd) + len(wr)
    print(f"| {x} | {t[x] / 1e9:.2f} | {', '.join(wr) or 'caller'} | {', '.join(rd)} | {n} ({n * t[x] / 1e9:.2f} GB) |")
alg = 2 * t["x"]
print(f"\nSURVEY section 8(d)'s algorithmic figure (the window read twice): {alg / 1e9:.2f} GB; with the tensors BPTT and the BatchNorm "
      f"batch statistics force through memory (each written once and read once): {sum(2 * v for k, v in t.items() if k != 'x') / 1e9 + alg / 1e9:.2f} GB; "
      f"ledger {tot_l / 1e9:.2f} GB; measured {(tot_m + rest) / 1e9:.2f} GB.")

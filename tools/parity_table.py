#!/usr/bin/env python3
"""Worst observed error per stage over a MSIG_PARITY_DUMP file (tests/gpu_common.py), next to the tolerance in force:
   MSIG_PARITY_DUMP=gpurun_out/parity.jsonl python -m pytest tests -m gpu -q ; python tools/parity_table.py gpurun_out/parity.jsonl"""
import json
import sys
from collections import defaultdict

worst, tol, n, worst_ratio = defaultdict(float), {}, defaultdict(int), defaultdict(float)
for ln in open(sys.argv[1]):
    r = json.loads(ln)
    for k, e in r["err"].items():
        key = "grad/*" if k.startswith("grad/") else k
        if e > worst[key]:
            worst[key] = e
        n[key] += 1
        tol[key] = max(tol.get(key, 0.0), r["tol"][k])
        own = r["own"].get(k)
        if own:
            worst_ratio[key] = max(worst_ratio[key], e / max(own, 1e-12))
print(f"{'stage':32s} {'cases':>6s} {'worst err':>11s} {'tolerance':>11s} {'tol/worst':>10s} {'worst err/own':>14s}")
for k in worst:
    w = worst[k]
    print(f"{k:32s} {n[k]:6d} {w:11.3e} {tol[k]:11.1e} {tol[k] / max(w, 1e-30):10.1f} {worst_ratio.get(k, 0.0):14.2f}")

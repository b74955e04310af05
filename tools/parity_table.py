#!/usr/bin/env python3
"""Worst observed error per stage over a MSIG_PARITY_DUMP file (tests/gpu_common.py), next to the fp32 oracle's own error:
   MSIG_PARITY_DUMP=gpurun_out/parity.jsonl python -m pytest tests -m gpu -q ; python tools/parity_table.py gpurun_out/parity.jsonl
With --recheck the tolerances CURRENTLY in tests/gpu_common.py are applied to the recorded (err, own) pairs — which recorded
comparisons would fail under them (calibrating the adaptive tolerances; evaluating a negative-control run offline)."""
import json
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
args = [a for a in sys.argv[1:] if not a.startswith("--")]
recheck = "--recheck" in sys.argv
if recheck:
    import gpu_common as G

worst, tol, n, worst_ratio, fails = defaultdict(float), {}, defaultdict(int), defaultdict(float), []
for path in args:
    for ln in open(path):
        r = json.loads(ln)
        ties = r["err"].get("pool_near_ties_adopted", 0.0)
        for k, e in r["err"].items():
            key = "grad/*" if k.startswith("grad/") else k
            if e > worst[key]:
                worst[key] = e
            n[key] += 1
            own = r["own"].get(k)
            t = r["tol"][k]
            if recheck and own is not None:
                slack = 1.0 + G.TIE_SLACK * ties
                if k.startswith("grad/"):
                    kk = k[5:]
                    front = kk.startswith("cnn_encoder.0") or kk.startswith("cnn_encoder.1") or kk.startswith("channel_attention")
                    t = G.grad_tol(kk, own, slack if front else 1.0)
                else:
                    t = G.stage_tol(k, own, 1.0 + G.TIE_SLACK_DBN1 * ties if k == "d_bn1" else slack if k == "d_gate_s" else 1.0)
            elif recheck:
                t = next((v for kk, v in G.FIXED_TOL.items() if kk in k), t)
            tol[key] = max(tol.get(key, 0.0), t)
            if own:
                worst_ratio[key] = max(worst_ratio[key], e / max(own, 1e-12))
            if not (e <= t):
                fails.append((r["test"].split("::")[-1], r.get("tag"), k, e, t, own))
print(f"{'stage':32s} {'cases':>6s} {'worst err':>11s} {'largest tol':>11s} {'worst err/own':>14s}")
for k in worst:
    print(f"{k:32s} {n[k]:6d} {worst[k]:11.3e} {tol[k]:11.1e} {worst_ratio.get(k, 0.0):14.2f}")
cases = sorted({(f[0], f[1]) for f in fails})
print(f"\n{len(fails)} comparisons over tolerance in {len(cases)} case(s)" + (" (tolerances of tests/gpu_common.py re-applied)" if recheck else " (tolerances as recorded)"))
for f in fails[:60]:
    print("  FAIL %-70s %-8s %-34s err %.3e tol %.1e own %s" % (f[0][:70], f[1], f[2], f[3], f[4], "%.2e" % f[5] if f[5] is not None else "-"))

"""Mean per launch of every counter of a rocprofv3 counter_collection.csv, one row per kernel (sorted by the first counter).

    python tools/pmc_table.py <counter_collection.csv> > table.csv
"""
import csv, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = sorted({c for k in acc for c in acc[k]})
print("kernel,launches," + ",".join(counters))
rows = []
for k, d in acc.items():
    n = max(len(v) for v in d.values())
    rows.append((k, n, [sum(d[c]) / len(d[c]) if d.get(c) else float("nan") for c in counters]))
for k, n, vals in sorted(rows, key=lambda r: -r[2][0] if r[2][0] == r[2][0] else 0):
    print(f"\"{k}\",{n}," + ",".join(f"{v:.6g}" for v in vals))

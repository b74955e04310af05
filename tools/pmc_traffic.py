"""Turns the two rocprofv3 PMC passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, each with --kernel-trace only, same
bench.py command) into per-kernel HBM bytes per launch:  bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (the counters
are in KB; gfx950's FETCH_SIZE counts half of 16-byte-per-lane reads — MI355X_MICROARCH.md, HBM section).

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <out.csv>

The json records the sha256 (first 16 hex digits) of the libmsig_hip.so the passes ran on; bench.py quotes a traffic figure
only while the library it runs is that one.
"""
import csv, hashlib, json, sys
from pathlib import Path
from collections import defaultdict

NAMES = {"void gru_bwd_fused<32>(GruArgs, int)": "gru_bwd_fused_l0", "void gru_bwd_fused<128>(GruArgs, int)": "gru_bwd_fused_l1",
         "void gru_bwd_b3<32>(GruArgs, int)": "gru_bwd_b3_l0", "void gru_bwd_b3<128>(GruArgs, int)": "gru_bwd_b3_l1",
         "void gru_fwd_seq<32, true>(GruArgs)": "gru_fwd_seq_l0", "void gru_fwd_seq<128, true>(GruArgs)": "gru_fwd_seq_l1",
         "void gru_fwd_b3<32, true>(GruArgs)": "gru_fwd_b3_l0", "void gru_fwd_b3<128, true>(GruArgs)": "gru_fwd_b3_l1"}


def means(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    fetch, write = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
    rows, kernels = [], {}
    for k in sorted(fetch, key=lambda k: -(2 * fetch[k][0] + write.get(k, (0, 0))[0])):
        f, n = fetch[k]
        w = write.get(k, (0.0, 0))[0]
        b = (2 * f + w) * 1024
        rows.append((k, n, f, w, b))
        if k in NAMES:
            kernels[NAMES[k]] = {"bytes_per_launch": b, "fetch_kb": f, "write_kb": w}
    so = Path(__file__).resolve().parent.parent / "multimodalsignal_amd" / "libmsig_hip.so"
    json.dump({"config": {"batch": 8192, "channels": 6, "samples": 3840},
               "lib_sha16": hashlib.sha256(so.read_bytes()).hexdigest()[:16],
               "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only, python3 bench.py --cpu-budget 0 "
                         "--steps 3 --warmup 1 --profile-steps 0); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts half of "
                         "16-B/lane streaming reads (MI355X_MICROARCH.md HBM section); calibration: conv1_fwd WRITE = the y1 tensor (1.0066 GB)",
               "kernels": kernels}, open(sys.argv[3], "w"), indent=1)
    with open(sys.argv[4], "w") as fh:
        fh.write("kernel,launches,FETCH_SIZE_KB_mean,WRITE_SIZE_KB_mean,hbm_bytes_per_launch_corrected\n")
        for k, n, f, w, b in rows:
            fh.write(f"\"{k}\",{n},{f:.1f},{w:.1f},{b:.4g}\n")
    for k, n, f, w, b in rows[:12]:
        print(f"{k[:60]:60s} {n:3d} launches  fetch {f/1e6:7.3f} GB(KB-count)  write {w/1e6:7.3f}  corrected {b/1e9:7.3f} GB")


main()

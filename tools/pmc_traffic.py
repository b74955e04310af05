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

import re


def label(kernel_name):
    """rocprofv3 kernel name -> the library's profile label (bench.py `kernels` keys): gru_bwd_b3<32, false>(...) -> gru_bwd_b3_l0."""
    m = re.match(r"void (gru_(?:fwd|bwd)_(?:b3|b4|ws|seq4|seq))<(32|128)[,>]", kernel_name)
    if m:
        return f"{m.group(1)}_l{0 if m.group(2) == '32' else 1}"
    m = re.match(r"void (gru_bwd_b[56])<", kernel_name)          # layer-0 kernels: gru_bwd_b6<false>(...) -> gru_bwd_b6_l0
    if m:
        return f"{m.group(1)}_l0"
    m = re.match(r"(?:void )?(conv1_fwd|conv1_bwd_fin|conv1_bwd|pool1_conv2_fwd|conv2_fwd|conv2_bwd|bn_relu_pool|pool_bn_bwd_pass1)(?:_kernel)?(?:<(\d+))?", kernel_name)
    if m:
        base = m.group(1)
        return f"{base}_{m.group(2)}" if base in ("bn_relu_pool", "pool_bn_bwd_pass1") and m.group(2) else base
    m = re.match(r"(?:void )?(gate|gate_bwd|ce|head_fwd|head_bwd|colsum_adam)_kernel", kernel_name)
    return m.group(1) if m else None


def src_sha16():
    """Same as bench.py src_sha16: sha256 over the kernel sources (the binary's hash depends on the build directory)."""
    root = Path(__file__).resolve().parent.parent
    csrc = root / "multimodalsignal_amd" / "csrc"
    h = hashlib.sha256()
    for f in sorted([*csrc.glob("*.hip"), *csrc.glob("*.h"), csrc / "Makefile", root / "include" / "msig.h"]):
        h.update(f.name.encode()); h.update(f.read_bytes())
    return h.hexdigest()[:16]


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
        if label(k) and label(k) not in kernels:
            kernels[label(k)] = {"bytes_per_launch": b, "fetch_kb": f, "write_kb": w, "kernel": k}
    so = Path(__file__).resolve().parent.parent / "multimodalsignal_amd" / "libmsig_hip.so"
    json.dump({"config": {"batch": 8192, "channels": 6, "samples": 3840},
               "lib_sha16": hashlib.sha256(so.read_bytes()).hexdigest()[:16], "src_sha16": src_sha16(),
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

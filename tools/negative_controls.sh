#!/bin/bash
# Negative controls of the parity gate (run on the GPU box, from the repository root; libraries built beforehand by
#   for k in 1 2 3 4 5 9; do make -C multimodalsignal_amd/csrc negctl NEGCTL=$k; done ).
# Each library drops the a1 * b1 cross term of the split-bf16 product (2^-16 relative per product) in ONE class of contractions
# (msig_dev.h CT_*): 1 backward recurrence, 2 dX, 3 dW, 4 forward recurrence, 5 forward projection, 9 all of them.  The parity
# tests below must FAIL against every one of them and PASS against the product library.  Output: one block per library.
SEL='golden_case_stages or (random_shapes and (ws6 or split) and (40-6-2-512 or 33-3-3-256 or 3100-6-2-64 or 17-6-2-320))'
OUT=${1:-gpurun_out/r04_negative_control.log}
mkdir -p gpurun_out
: > "$OUT"
echo "# tools/negative_controls.sh: python -m pytest tests/test_parity_gpu.py -k \"$SEL\"  (4 golden cases + 4 shapes x the two shipped form sets = 12 cases)" >> "$OUT"
for k in product 1 2 3 4 5 9; do
  if [ "$k" = product ]; then lib=$PWD/multimodalsignal_amd/libmsig_hip.so; else lib=$PWD/multimodalsignal_amd/libmsig_hip_negctl$k.so; fi
  [ -f "$lib" ] || { echo "missing $lib" >> "$OUT"; continue; }
  rm -f gpurun_out/negctl_$k.jsonl
  MSIG_LIB=$lib MSIG_PARITY_DUMP=gpurun_out/negctl_$k.jsonl timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -q -p no:cacheprovider -k "$SEL" > gpurun_out/negctl_$k.txt 2>&1
  echo "## library: $(basename $lib)   ->   $(tail -1 gpurun_out/negctl_$k.txt)" >> "$OUT"
  grep -h "^FAIL " gpurun_out/negctl_$k.txt | sort | uniq -c | sort -rn | head -12 >> "$OUT"
done
tail -n +1 "$OUT"

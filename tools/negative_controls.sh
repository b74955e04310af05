#!/bin/bash
# Negative controls of the parity gate (run on the GPU box, from the repository root; libraries built beforehand by
#   for k in 1 2 3 4 5 9; do make -C multimodalsignal_amd/csrc negctl NEGCTL=$k; done   -> multimodalsignal_amd/csrc/build/ ).
# Each library drops ONE cross term of the split products in ONE class of contractions (msig_dev.h CT_*): split-bf16 without a1 * b1
# (2^-16 relative per product), two-piece fp16 without a1 * b0 (2^-11).  1 backward recurrence, 2 dX, 3 dW, 4 forward recurrence,
# 5 forward projection, 9 all of them.  Every selected case must FAIL against every control and PASS against the product library;
# the script prints per-case verdicts and EXITS NON-ZERO when a control passes a case or the product fails one.
SEL='golden_case_stages or (random_shapes and (ws6 or split) and (40-6-2-512 or 33-3-3-256 or 3100-6-2-64 or 17-6-2-320))'
OUT=${1:-gpurun_out/r05_negative_control.log}
mkdir -p gpurun_out
: > "$OUT"
echo "# tools/negative_controls.sh: python -m pytest tests/test_parity_gpu.py -k \"$SEL\"  (4 golden cases + 4 shapes x the two shipped form sets = 12 cases)" >> "$OUT"
bad=0
for k in product 1 2 3 4 5 9; do
  if [ "$k" = product ]; then lib=$PWD/multimodalsignal_amd/libmsig_hip.so; else lib=$PWD/multimodalsignal_amd/csrc/build/libmsig_hip_negctl$k.so; fi
  [ -f "$lib" ] || { echo "missing $lib" >> "$OUT"; bad=1; continue; }
  MSIG_LIB=$lib timeout -k 10 400 python -m pytest tests/test_parity_gpu.py -q -rA -p no:cacheprovider -k "$SEL" > gpurun_out/negctl_$k.txt 2>&1
  echo "## library: $(basename $lib)   ->   $(tail -1 gpurun_out/negctl_$k.txt)" >> "$OUT"
  grep -E "^(PASSED|FAILED) " gpurun_out/negctl_$k.txt | sed 's/ - .*//' | sort >> "$OUT"
  npass=$(grep -c "^PASSED " gpurun_out/negctl_$k.txt); nfail=$(grep -c "^FAILED " gpurun_out/negctl_$k.txt)
  if [ "$k" = product ]; then [ "$nfail" = 0 ] && [ "$npass" = 12 ] || { echo "!! the product library must pass all 12 cases" >> "$OUT"; bad=1; }
  else [ "$npass" = 0 ] && [ "$nfail" = 12 ] || { echo "!! control $k passed $npass of 12 cases: the gate does not catch it there" >> "$OUT"; bad=1; }; fi
done
tail -n +1 "$OUT"
exit $bad

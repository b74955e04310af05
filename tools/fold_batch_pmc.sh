#!/bin/bash
# Where do the recurrence kernels' cycles go inside a 15-fold batch compared with one fold alone?  (GPU box, repo root.)
# Kernel durations (--kernel-trace --stats) and, in separate passes, wave-cycle counters for F = 1 and F = 15.
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/fold_pmc; rm -rf $O; mkdir -p $O; cd /tmp
for F in 1 15; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$F -- python3 $R/tools/fold_batch_steps.py $F 20 > $O/stats_$F.log 2>&1
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/pmc_$F -- python3 $R/tools/fold_batch_steps.py $F 20 > $O/pmc_$F.log 2>&1
done
cd $R
for F in 1 15; do
  echo "== F = $F: kernel stats (ns)"; s=$(find $O/stats_$F -name "*kernel_stats.csv" | head -1); grep -E "gru_fwd_rec|gru_bwd_seq4|Name" $s | cut -c1-160
  echo "== F = $F: counters per launch"; c=$(find $O/pmc_$F -name "*counter_collection.csv" | head -1); python3 tools/pmc_table.py $c | grep -E "kernel,|gru_fwd_rec|gru_bwd_seq4" | cut -c1-300
done

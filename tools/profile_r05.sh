#!/bin/bash
# Round-5 profile collection on the GPU box (run from the repo root through gpurun): kernel-trace stats of the bench command,
# then separate PMC passes (never combined with other trace domains).  Summaries land in gpurun_out/prof_r05/.
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/prof_r05; mkdir -p $O; cd /tmp
A="--steps 3 --warmup 1 --profile-steps 0 --loso 0 --cpu-budget 0 --b64-steps 0 --long-steps 0"
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --loso 0 --cpu-budget 0 --b64-steps 0 --long-steps 0 > $O/bench_under_rocprof.json 2> $O/stats.log
for c in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  timeout -k 10 250 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/bench.py $A > $O/pmc_$c.log 2>&1
done
timeout -k 10 250 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O/pmc_wave -- python3 $R/bench.py $A > $O/pmc_wave.log 2>&1
timeout -k 10 250 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_insts -- python3 $R/bench.py $A > $O/pmc_insts.log 2>&1
cd $R
f=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $f $w $O/r05_pmc_traffic.json $O/r05_pmc_fetch_write_B8192.csv > $O/traffic.txt
for d in MfmaUtil wave insts; do f=$(find $O/pmc_$d -name "*counter_collection.csv" | head -1); python tools/pmc_table.py $f > $O/r05_pmc_${d}_B8192.csv; done
s=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $s $O/r05_bench_B8192_kernel_stats.csv
head -8 $O/r05_bench_B8192_kernel_stats.csv; cat $O/traffic.txt | head -14; head -5 $O/r05_pmc_MfmaUtil_B8192.csv

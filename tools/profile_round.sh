#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02
# kernel-trace stats of the bench command, then separate PMC passes (never combined with sys/hip traces):
# FETCH_SIZE, WRITE_SIZE, MfmaUtil, instruction counts, LDS bank conflicts.  Summaries land in gpurun_out/prof_<tag>/;
# copy what is to be judged into profiles/.
set -e
TAG=${1:-rXX}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
BENCH="python3 bench.py --cpu-budget 0 --loso 0 --b64-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH --steps 8 --warmup 2 > $OUT/stats_bench.json 2> $OUT/stats.err
PMCB="$BENCH --steps 3 --warmup 1 --profile-steps 0"
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -- $PMCB > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/write -- $PMCB > /dev/null 2> $OUT/write.err
rocprofv3 --kernel-trace --output-format csv --pmc MfmaUtil -d $OUT/mfma -- $PMCB > /dev/null 2> $OUT/mfma.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA -d $OUT/insts -- $PMCB > /dev/null 2> $OUT/insts.err || true
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/lds -- $PMCB > /dev/null 2> $OUT/lds.err || true
F=$(find $OUT/fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py $F $W $OUT/pmc_traffic.json $OUT/pmc_fetch_write.csv > $OUT/pmc_traffic.log
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
for d in mfma insts lds; do
  C=$(find $OUT/$d -name "*counter_collection.csv" | head -1)
  [ -n "$C" ] && python3 tools/pmc_table.py $C > $OUT/pmc_$d.csv || true
done
head -12 $OUT/kernel_stats.csv; cat $OUT/pmc_traffic.log; head -12 $OUT/pmc_mfma.csv; head -12 $OUT/pmc_lds.csv; head -12 $OUT/pmc_insts.csv
# where the wave cycles go (quad-cycles): parked on s_waitcnt / s_barrier, issue stalls, active issue by unit
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM -d $OUT/waves -- $PMCB > /dev/null 2> $OUT/waves.err || true
C=$(find $OUT/waves -name "*counter_collection.csv" | head -1); [ -n "$C" ] && python3 tools/pmc_table.py $C > $OUT/pmc_waves.csv && head -8 $OUT/pmc_waves.csv

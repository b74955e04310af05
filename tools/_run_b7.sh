A="--loso 0 --cpu-budget 0 --long-steps 0 --b64-steps 0"
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -k "ws7 or b7" > gpurun_out/b7_parity.log 2>&1; rc=$?; tail -25 gpurun_out/b7_parity.log | cut -c1-250; [ $rc = 0 ] || exit $rc
MSIG_LIB=$PWD/multimodalsignal_amd/libmsig_hip_stamps.so MSIG_GRU_BWD=b7 timeout -k 10 150 python bench.py --steps 2 --warmup 1 $A --profile-steps 0 > gpurun_out/b7_stamps.log 2>&1 || exit 1
grep -h "stamps b7" gpurun_out/b7_stamps.log | sort | uniq | head -4
MSIG_GRU_BWD=b7 timeout -k 10 120 python bench.py $A --steps 30 --warmup 5 > gpurun_out/b7_bench.log 2>&1 || exit 1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/b7_bench.log').read().strip().splitlines()[-1])
print('b7 ms/step', d['ms_per_step'], d.get('ms_per_step_spread'), 'loss', d.get('loss_last'))
for n,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:5]: print('   ', n, v)
PY

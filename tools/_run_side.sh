timeout -k 10 500 python -m pytest tests/test_parity_gpu.py tests/test_trainer_gpu.py -x -q -k "split or graph or lockstep or fold_batch" > gpurun_out/side_parity.log 2>&1; rc=$?; tail -6 gpurun_out/side_parity.log | cut -c1-220; [ $rc = 0 ] || exit $rc
for ss in 1 0; do
MSIG_SIDE_STREAM=$ss timeout -k 10 100 python bench.py --batch 64 --steps 300 --profile-steps 0 --loso 0 --cpu-budget 0 --b64-steps 0 --long-steps 0 > gpurun_out/side_b64.json 2>&1 || exit 1
python - $ss <<'PY'
import json,sys
b=json.loads(open('gpurun_out/side_b64.json').read().strip().splitlines()[-1])
print('side',sys.argv[1],'B64 step', b['ms_per_step'], b['value'], b.get('ms_per_step_spread'))
PY
MSIG_SIDE_STREAM=$ss timeout -k 10 200 python tools/multi_step_probe.py split quick > gpurun_out/side_probe_$ss.log 2>&1; grep "alone\|concurrently" gpurun_out/side_probe_$ss.log | cut -c1-200
done

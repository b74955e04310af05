timeout -k 10 400 python -m pytest tests/test_parity_gpu.py -x -q -k "split" > gpurun_out/bulk_parity.log 2>&1; rc=$?; tail -15 gpurun_out/bulk_parity.log | cut -c1-220; [ $rc = 0 ] || exit $rc
timeout -k 10 100 python bench.py --batch 64 --steps 200 --profile-steps 20 --loso 0 --cpu-budget 0 --b64-steps 0 --long-steps 0 > gpurun_out/bulk_b64.json 2>&1 || exit 1
python - <<'PY'
import json
b=json.loads(open('gpurun_out/bulk_b64.json').read().strip().splitlines()[-1])
print('B64', b['ms_per_step'], b['value'])
for n,v in sorted(b['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:14]: print(f"{n:24s} {1000*v['ms_per_step']:8.1f} us")
PY
timeout -k 10 200 python tools/multi_step_probe.py split quick > gpurun_out/bulk_probe.log 2>&1; tail -6 gpurun_out/bulk_probe.log | cut -c1-400

A="--loso 0 --cpu-budget 0 --long-steps 0 --b64-steps 0"
MSIG_LIB=$PWD/multimodalsignal_amd/libmsig_hip_stamps.so MSIG_GRU_BWD=b6 timeout -k 10 150 python bench.py --steps 2 --warmup 1 $A --profile-steps 0 > gpurun_out/s6_stamps_b6.log 2>&1 || exit 1
grep -h "stamps b6" gpurun_out/s6_stamps_b6.log | sort | uniq | head -4
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -k "ws6 or b6" > gpurun_out/s6_parity_b6.log 2>&1; rc=$?; tail -3 gpurun_out/s6_parity_b6.log | cut -c1-300; [ $rc = 0 ] || exit $rc
MSIG_GRU_BWD=b6 timeout -k 10 120 python bench.py $A --steps 30 --warmup 5 > gpurun_out/s6_b6.log 2>&1 || exit 1
python - <<'PY'
import json
for f in ['s6_b6']:
    d=json.loads(open(f'gpurun_out/{f}.log').read().strip().splitlines()[-1])
    print(f, 'ms/step', d['ms_per_step'], d.get('ms_per_step_spread'), 'loss', d.get('loss_last'))
    k=d.get('kernels',{})
    for n,v in sorted(k.items(), key=lambda kv:-kv[1].get('ms_per_step',0) if isinstance(kv[1],dict) else 0)[:5]:
        print('   ', n, v)
PY

set -e
mkdir -p gpurun_out/r05
python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "golden or split or fold_batch or fused_step or any_backward or three_fused" > gpurun_out/r05/c2_tests.log 2>&1 || { tail -40 gpurun_out/r05/c2_tests.log; exit 1; }
tail -3 gpurun_out/r05/c2_tests.log
python bench.py --batch 64 --steps 200 --profile-steps 20 --loso 0 --cpu-budget 0 --b64-steps 0 > gpurun_out/r05/c2_b64.json 2> gpurun_out/r05/c2_b64.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/c2_b64.json').read().strip().splitlines()[-1])
print('B64 step', d['ms_per_step'])
for k,v in d['kernels'].items(): print(f"  {k:24s} {1e3*v['ms_per_step']:7.1f} us x{v['launches_per_step']}")
PY
python tools/multi_step_probe.py split quick > gpurun_out/r05/c2_multi.log 2>&1
cat gpurun_out/r05/c2_multi.log

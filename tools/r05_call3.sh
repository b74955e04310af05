set -e
mkdir -p gpurun_out/r05
python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "golden or ws6 or split or fold_batch or lds_images or many_tiles or one_layer or fused_step" > gpurun_out/r05/c3_tests.log 2>&1 || { tail -40 gpurun_out/r05/c3_tests.log; exit 1; }
tail -3 gpurun_out/r05/c3_tests.log
python bench.py --steps 30 --profile-steps 3 --loso 0 --cpu-budget 0 --b64-steps 100 > gpurun_out/r05/c3_bench.json 2> gpurun_out/r05/c3_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/c3_bench.json').read().strip().splitlines()[-1])
print('B8192 step', d['ms_per_step'], 'value', d['value'], 'b64', d['b64']['ms_per_step'])
for k,v in d['kernels'].items(): print(f"  {k:24s} {v['ms_per_step']:7.3f} ms x{v['launches_per_step']}")
print(d['roofline'])
PY

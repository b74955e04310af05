for cap in 128 256 512 0; do
  if [ $cap = 0 ]; then unset MSIG_BULK_CAP; else export MSIG_BULK_CAP=$cap; fi
  timeout -k 10 100 python bench.py --batch 64 --steps 200 --profile-steps 20 --loso 0 --cpu-budget 0 --b64-steps 0 --long-steps 0 > gpurun_out/cap.json 2>&1 || exit 1
  python - $cap <<'PY'
import json,sys
b=json.loads(open('gpurun_out/cap.json').read().strip().splitlines()[-1])
k=b['kernels']
print('cap',sys.argv[1],'step',b['ms_per_step'], {n:round(1000*k[n]['ms_per_step'],1) for n in ['gru_fwd_proj_l1','gru_fwd_proj_l0','gru_bwd_dx_l1','gru_bwd_dx_l0','gru_bwd_dx_l1rev']})
PY
done

timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/f_gpu_suite.log 2>&1; rc=$?; tail -3 gpurun_out/f_gpu_suite.log; [ $rc = 0 ] || exit $rc
timeout -k 10 400 python bench.py > gpurun_out/f_bench_default.json 2> gpurun_out/f_bench_default.err || exit 1
tail -c 600 gpurun_out/f_bench_default.json
timeout -k 10 120 python bench.py --batch 64 --steps 200 --profile-steps 20 --loso 0 --cpu-budget 0 --b64-steps 0 --long-steps 0 > gpurun_out/f_bench_b64.json 2>&1 || exit 1

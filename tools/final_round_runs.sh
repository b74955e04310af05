# Round-end GPU runs (through gpurun, from the repo root): default bench line, B = 64 kernel times, then tools/profile_r05.sh (kernel stats + PMC passes).
mkdir -p gpurun_out
timeout -k 10 500 python bench.py > gpurun_out/f_bench_default.json 2> gpurun_out/f_bench_default.err || exit 1
timeout -k 10 100 python bench.py --batch 64 --steps 200 --profile-steps 20 --loso 0 --cpu-budget 0 --b64-steps 0 --long-steps 0 > gpurun_out/f_bench_b64.json 2>&1 || exit 1
rm -rf gpurun_out/prof_r05
bash tools/profile_r05.sh > gpurun_out/prof_r05_run.log 2>&1; tail -4 gpurun_out/prof_r05_run.log | cut -c1-160

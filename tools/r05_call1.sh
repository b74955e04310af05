set -e
mkdir -p gpurun_out/r05
python tools/loso_parity.py --side gpu --dropout 0 --no-shuffle --epochs 100 --windows 270 --window-spread 20 --difficulty 2 --data /tmp/wesad_benchset --folds S2 S3 S4 S5 S6 S7 S8 S9 S10 S11 S13 S14 S15 S16 S17 --out gpurun_out/r05/det_benchset_gpu.json > gpurun_out/r05/det_benchset_gpu.log 2>&1
python tools/profile_loso.py --no-profile > gpurun_out/r05/loso_plot_on.log 2>&1
python tools/profile_loso.py --no-profile >> gpurun_out/r05/loso_plot_on.log 2>&1
MSIG_NO_PLOT=1 python tools/profile_loso.py --no-profile > gpurun_out/r05/loso_plot_off.log 2>&1
MSIG_NO_PLOT=1 python tools/profile_loso.py --no-profile >> gpurun_out/r05/loso_plot_off.log 2>&1
for Q in 4 8 16 24; do GPU_MAX_HW_QUEUES=$Q PROBE_STREAMS=1,2,3,4,5,6,7,8,10,12,15 python tools/stream_interference_probe.py >> gpurun_out/r05/stream_sweep.log 2>&1; done
tail -3 gpurun_out/r05/loso_plot_on.log gpurun_out/r05/loso_plot_off.log; cat gpurun_out/r05/stream_sweep.log

R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_job.py tests/test_gpu_configs.py -m gpu -x -q -k "bucketed or wide_keys or large_k or group8 or long_and_ragged or probe_counting or three_mask" > $O/r03_run20_tests.log 2>&1 || { tail -30 $O/r03_run20_tests.log; exit 1; }
tail -2 $O/r03_run20_tests.log
cd /tmp
for k in 33 34; do
python3 $R/bench.py -k $k --no-matrix --cpu-sample 0 > $O/r03_k${k}_bench_rolled.json 2>/dev/null
python3 -c "
import json
d=json.load(open('$O/r03_k${k}_bench_rolled.json')); print('k', $k, d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],2) for k,v in d['roofline']['kernels'].items()})"
done

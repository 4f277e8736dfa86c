R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_kernels.py -m gpu -x -q -k "probe_counting or three_mask or wide_keys or bucketed or k33 or group8 or hot or split" > $O/r03_run13_tests.log 2>&1 || { tail -30 $O/r03_run13_tests.log; exit 1; }
tail -2 $O/r03_run13_tests.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_job.py tests/test_gpu_fullsize.py -m gpu -x -q -k "bucketed or k33 or tiled" > $O/r03_run13_tests2.log 2>&1 || { tail -30 $O/r03_run13_tests2.log; exit 1; }
tail -2 $O/r03_run13_tests2.log
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py -k 33 --no-matrix --cpu-sample 0 > $O/r03_k33_bench_packed.json 2>/dev/null
python3 -c "
import json
d=json.load(open('$O/r03_k33_bench_packed.json')); print('k33', d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],2) for k,v in d['roofline']['kernels'].items()})"
bash $R/tools/profile_bench.sh r03_c5 -k 21 -t 5 --reads 20000000 --read-len 150 --steps 1 --warmup 0 --cpu-sample 0 --no-probe-count --kt-steps 1 > $O/r03_c5.log 2>&1
tail -1 $O/r03_c5.log | cut -c1-1800

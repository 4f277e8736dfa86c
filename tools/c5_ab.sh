set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_job.py -x -q -m gpu -k "sliced" > gpurun_out/r02_sliced_tests.log 2>&1 || { tail -30 gpurun_out/r02_sliced_tests.log; exit 1; }
tail -3 gpurun_out/r02_sliced_tests.log
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "c5_regime" > gpurun_out/r02_c5_test.log 2>&1 || { tail -30 gpurun_out/r02_c5_test.log; exit 1; }
tail -3 gpurun_out/r02_c5_test.log
B="--reads 2000000 --read-len 150 -k 21 -t 5 --steps 1 --warmup 0 --no-matrix --cpu-sample 0 --no-probe-count"
for w in 1 2 4 8; do COMMET_SLICE_WORDS=$w timeout -k 10 300 python bench.py $B > gpurun_out/r02_c5_w$w.json 2> gpurun_out/r02_c5_w$w.err; done
COMMET_SLICE_MODE=1 timeout -k 10 300 python bench.py $B > gpurun_out/r02_c5_slots.json 2> gpurun_out/r02_c5_slots.err
python - <<'P'
import json
for n in ("w1","w2","w4","w8","slots"):
    try:
        b=json.load(open(f"gpurun_out/r02_c5_{n}.json"))
        print(n, "ms/step", b["ms_per_step"], {k:round(v["ms_per_step"],1) for k,v in b["roofline"]["kernels"].items()})
    except Exception as e: print(n, "failed", e)
P

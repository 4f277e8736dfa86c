# wide rows of the many-small-chunks regime on the GPU box: tests, then configs[4] on a 2 M x 2 M slice (wide / narrow) and at full size
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
( free -g | head -2; df -h /dev/shm | tail -1; nproc; cat /sys/fs/cgroup/memory.max /sys/fs/cgroup/cpu.max ) > $O/r03_box.txt 2>&1
timeout -k 10 900 python3 -m pytest tests/test_gpu_job.py tests/test_gpu_configs.py -x -q -m gpu -k "wide or c5" > $O/r03_wide_tests.log 2>&1 || { tail -30 $O/r03_wide_tests.log; exit 1; }
tail -3 $O/r03_wide_tests.log
B="-k 21 -t 5 --read-len 150 --steps 1 --warmup 0 --cpu-sample 0 --no-probe-count --no-matrix --kt-steps 1"
cd /tmp
python3 $R/bench.py $B --reads 2000000 > $O/r03_c5_2m_wide.json 2> $O/r03_c5_2m_wide.err && \
COMMET_SLICE_WIDE=1 python3 $R/bench.py $B --reads 2000000 > $O/r03_c5_2m_narrow.json 2> $O/r03_c5_2m_narrow.err && \
timeout -k 10 300 python3 $R/bench.py $B --reads 20000000 > $O/r03_c5_full_wide.json 2> $O/r03_c5_full_wide.err
python3 - <<'P'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/"
for f in ("r03_c5_2m_wide", "r03_c5_2m_narrow", "r03_c5_full_wide"):
    try:
        d = json.load(open(O + f + ".json"))
        print(f, d["value"], d["ms_per_step"], {k: round(v["ms_per_step"], 2) for k, v in d["roofline"]["kernels"].items()})
    except Exception as ex:
        print(f, "failed", ex)
P

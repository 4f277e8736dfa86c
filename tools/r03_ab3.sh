R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/r03_hot_tests.log 2>&1 || { tail -30 $O/r03_hot_tests.log; exit 1; }
tail -2 $O/r03_hot_tests.log
cd /tmp
python3 $R/bench.py --skew 0.1 --no-matrix --cpu-sample 0 > $O/r03_skew_bench2.json 2> /dev/null
python3 $R/bench.py --no-matrix --cpu-sample 0 > $O/r03_uniform_bench2.json 2> /dev/null
python3 - <<'P'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/"
for f in ("r03_uniform_bench2", "r03_skew_bench2"):
    d = json.load(open(O + f + ".json"))
    print(f, d["value"], d["ms_per_step"], {k: round(v["ms_per_step"], 2) for k, v in d["roofline"]["kernels"].items()})
P

# the non-uniform data leg (10 % low-complexity / repeated reads) and tracked profiles of the 2 x 25 M and 2 x 50 M-read jobs
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "skewed" > $O/r03_skew_test.log 2>&1 || { tail -30 $O/r03_skew_test.log; exit 1; }
tail -2 $O/r03_skew_test.log
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --skew 0.1 --no-matrix --cpu-sample 0 > $O/r03_skew_bench.json 2> $O/r03_skew_bench.err
COMMET_TILED=1 python3 $R/bench.py --skew 0.1 --no-matrix --cpu-sample 0 > $O/r03_skew_bench_untiled.json 2> /dev/null
python3 $R/bench.py --no-matrix --cpu-sample 0 > $O/r03_uniform_bench.json 2> /dev/null
python3 - <<'P'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/"
for f in ("r03_uniform_bench", "r03_skew_bench", "r03_skew_bench_untiled"):
    d = json.load(open(O + f + ".json"))
    print(f, d["value"], d["ms_per_step"], {k: round(v["ms_per_step"], 2) for k, v in d["roofline"]["kernels"].items()})
P
bash $R/tools/profile_bench.sh r03_25m --reads 25000000 --cpu-sample 0 --no-probe-count > $O/r03_25m.log 2>&1; tail -3 $O/r03_25m.log | cut -c1-600
bash $R/tools/profile_bench.sh r03_50m --reads 50000000 --cpu-sample 0 --no-probe-count > $O/r03_50m.log 2>&1; tail -3 $O/r03_50m.log | cut -c1-600

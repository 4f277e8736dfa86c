# round 5, call 13: the full GPU suite of the frozen sources (durations), then bench lines for k = 33 and skewed sets
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_tests gpurun_out/r05_final
python3 -m pytest tests -q -m gpu --durations=25 > gpurun_out/r05_tests/full2.log 2>&1 || { tail -40 gpurun_out/r05_tests/full2.log; exit 1; }
tail -32 gpurun_out/r05_tests/full2.log
python3 bench.py --no-matrix --cpu-sample 0 --no-probe-count -k 33 > gpurun_out/r05_final/bench_k33.json 2> gpurun_out/r05_final/bench_k33.err
python3 bench.py --no-matrix --cpu-sample 0 --no-probe-count --skew 0.1 > gpurun_out/r05_final/bench_skew.json 2> gpurun_out/r05_final/bench_skew.err
python3 -c "
import json
for f in ('bench_k33', 'bench_skew'):
    b = json.load(open('gpurun_out/r05_final/' + f + '.json')); print(f, b['value'], b['ms_per_step'], b['detail']['index_kernel_ms'], b['detail']['search_kernel_ms'])"

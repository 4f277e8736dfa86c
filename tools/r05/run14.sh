# round 5, call 14: the configs[4] profile of the round's sources; the configs[3] leg's per-kernel times as bench.py runs it
set -e
cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh r05_c5 --reads 20000000 --read-len 150 -k 21 -t 5 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r05_c5_profile.log 2>&1 || { tail -20 gpurun_out/r05_c5_profile.log; exit 1; }
python3 -c "
import json; b = json.load(open('gpurun_out/r05_c5/bench.json')); print('c5', b['value'], b['ms_per_step'], b['roofline']['kernel'], b['roofline']['frac'], b['roofline'].get('as_executed'))"
mkdir -p gpurun_out/r05_c3_matrix
COMMET_MATRIX_KERNEL_TIMES=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_c3_matrix/bench_kernel_times.json 2> gpurun_out/r05_c3_matrix/bench_kernel_times.err
python3 -c "
import json; b = json.load(open('gpurun_out/r05_c3_matrix/bench_kernel_times.json'))
for n in ('matrix_configs2', 'matrix'):
    m = b['detail'][n]; pr = m['per_rank'][0]
    print(n, m['total_s'], m['jobs_s'], pr['device_ms'], sorted(pr['kernel_ms'].items(), key=lambda kv: -kv[1][1])[:8])"

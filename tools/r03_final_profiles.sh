# the round's final rocprofv3 passes: configs[1] (r03_final) and configs[4] (r03_c5), bench lines priced with them,
# and the default bench line with its matrix leg (configs[2]) and CPU baseline
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
bash $R/tools/profile_bench.sh r03_final --cpu-sample 0 > $O/r03_final.log 2>&1; tail -1 $O/r03_final.log | cut -c1-300
bash $R/tools/profile_bench.sh r03_c5 -k 21 -t 5 --reads 20000000 --read-len 150 --steps 1 --warmup 0 --cpu-sample 0 --no-probe-count --kt-steps 1 > $O/r03_c5.log 2>&1; tail -1 $O/r03_c5.log | cut -c1-300
cd /tmp && python3 $R/bench.py --traffic $O/r03_final/traffic.json > $O/r03_final/bench_with_matrix.json 2> $O/r03_final/bench_with_matrix.err
python3 -c "
import json
d=json.load(open('$O/r03_final/bench_with_matrix.json')); m=d['detail']['matrix']; print(d['value'], d['ms_per_step'], {k:v for k,v in m.items() if k not in ('per_rank','workload','predicted_vs_actual_share')})"

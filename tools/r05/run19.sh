# round 5, call 19: J2 / J3 jobs batched (commet_index_many_and_search): the matrix tests (ABCDE goldens, configs[2] and configs[3] against the CPU
# checker), then the driver's bench command twice
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_multi
mkdir -p $O
python3 -m pytest tests/test_gpu_multi.py tests/test_gpu_matrix.py tests/test_gpu_configs.py tests/test_gpu_cli.py -x -q -m gpu -k "multi or matrix or c3 or c4 or cli" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for i in 1 2; do
  python3 bench.py --gpus 1 --steps 10 --warmup 3 --cpu-sample 0 --no-probe-count > $O/bench_$i.json 2> $O/bench_$i.err
  python3 - $O/bench_$i.json <<'PY'
import json, sys
b = json.load(open(sys.argv[1])); d = b["detail"]
print("value", b["value"])
for n in ("matrix_configs2", "matrix"):
    m = d[n]; pr = m["per_rank"][0]
    print(n, "total_s", m.get("total_s"), "jobs_s", m.get("jobs_s"), "set_wait_s", m.get("set_wait_s"), "device_ms", pr.get("device_ms"), "call_ms", pr.get("call_ms"), "jobs", pr.get("jobs"), m.get("error"))
PY
done
COMMET_MATRIX_KERNEL_TIMES=1 python3 bench.py --gpus 1 --steps 5 --warmup 2 --cpu-sample 0 --no-probe-count > $O/bench_kt.json 2> $O/bench_kt.err
python3 -c "
import json; b = json.load(open('$O/bench_kt.json'))
for n in ('matrix_configs2', 'matrix'):
    m = b['detail'][n]; pr = m['per_rank'][0]
    print(n, m['total_s'], m['jobs_s'], pr['device_ms'], sorted(pr['kernel_ms'].items(), key=lambda kv: -kv[1][1])[:8])"

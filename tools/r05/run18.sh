# round 5, call 18: lists above the cap with their memory set aside by the driver's loader thread: the test, then the driver's bench
# command three times (the first run of a call meets a box whose device memory is untouched)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_large_lists
mkdir -p $O
python3 -m pytest tests/test_gpu_large_list.py tests/test_gpu_matrix.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for i in 1 2 3; do
  python3 bench.py --gpus 1 --steps 10 --warmup 3 --cpu-sample 0 --no-probe-count > $O/bench_reserve_$i.json 2> $O/bench_reserve_$i.err
  python3 - $O/bench_reserve_$i.json <<'PY'
import json, sys
b = json.load(open(sys.argv[1])); d = b["detail"]
for n in ("matrix_configs2", "matrix"):
    m = d[n]; pr = m["per_rank"][0]
    print(n, "total_s", m.get("total_s"), "jobs_s", m.get("jobs_s"), "set_wait_s", m.get("set_wait_s"), "device_ms", pr.get("device_ms"), "call_ms", pr.get("call_ms"), "lists", pr.get("lists_reserved"), m.get("error"))
PY
  grep "large query lists\|set 0 resident" $O/bench_reserve_$i.err | tail -3
done

# round 5, call 32: the host planner's downloads replaced by kernel stores into pinned host memory: tests, then the configs[3] leg's calls, wall against device time
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_zerocopy
mkdir -p $O; rm -f $O/calls*.log
tools/exp/alloc_cost > $O/alloc_cost.log 2>&1; sed -n 2,2p $O/alloc_cost.log
python3 -m pytest tests/test_gpu_job.py tests/test_gpu_multi.py tests/test_gpu_matrix.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for v in a b; do
  COMMET_MATRIX_CALL_LOG=$GRAFT_REPO_ROOT/$O/calls_$v.log python3 bench.py --gpus 1 --steps 3 --warmup 1 --cpu-sample 0 --no-probe-count --no-kernel-times > $O/bench_$v.json 2> $O/bench_$v.err
  python3 - $O/bench_$v.json $v $O/calls_$v.log <<'PY'
import json, sys
b = json.load(open(sys.argv[1])); d = b["detail"]
print("run", sys.argv[2], "value", b["value"])
for n in ("matrix_configs2", "matrix"):
    m = d[n]; pr = m["per_rank"][0]
    print("  ", n, "total_s", m.get("total_s"), "jobs_s", m.get("jobs_s"), "set_wait_s", m.get("set_wait_s"), "device_ms", pr.get("device_ms"), "call_ms", pr.get("call_ms"), m.get("error"))
rows = [l.split() for l in open(sys.argv[3])][-27:]
g = lambda rs: [round(float(r[2]) - float(r[3]) - float(r[4]), 1) for r in rs]
print("   J1 host-only ms:", g([r for i, r in enumerate(rows[:18]) if i % 2 == 0]))
print("   J2 host-only ms:", g([r for i, r in enumerate(rows[:18]) if i % 2 == 1]))
print("   J3 host-only ms:", g(rows[18:]))
PY
done

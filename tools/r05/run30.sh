# round 5, call 30: every library call of the configs[3] leg: wall time against event-timed device time (COMMET_MATRIX_CALL_LOG)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_calls
mkdir -p $O; rm -f $O/calls.log
tools/exp/alloc_cost > $O/alloc_cost.log 2>&1; sed -n 2,2p $O/alloc_cost.log
COMMET_MATRIX_CALL_LOG=$GRAFT_REPO_ROOT/$O/calls.log python3 bench.py --gpus 1 --steps 3 --warmup 1 --cpu-sample 0 --no-probe-count --no-kernel-times > $O/bench.json 2> $O/bench.err
wc -l $O/calls.log
python3 - $O/bench.json <<'PY'
import json, sys
b = json.load(open(sys.argv[1])); d = b["detail"]
for n in ("matrix_configs2", "matrix"):
    m = d[n]; pr = m["per_rank"][0]
    print(n, "total_s", m.get("total_s"), "jobs_s", m.get("jobs_s"), "set_wait_s", m.get("set_wait_s"), "device_ms", pr.get("device_ms"), "call_ms", pr.get("call_ms"), m.get("error"))
PY

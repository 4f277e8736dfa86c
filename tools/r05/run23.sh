# round 5, call 23: the whole GPU suite on the round's sources, then the driver's bench command priced with the committed profiles/r05_final
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_tests
mkdir -p $O
T0=$(date +%s)
python3 -m pytest tests -x -q -m gpu --durations=15 > $O/gpu_suite.log 2>&1; RC=$?
echo "suite rc=$RC in $(( $(date +%s) - T0 )) s"; tail -22 $O/gpu_suite.log
[ $RC -eq 0 ] || exit $RC
O2=gpurun_out/r05_final
mkdir -p $O2
python3 bench.py --gpus 1 --steps 10 --warmup 3 > $O2/bench_driver_command.json 2> $O2/bench_driver_command.progress.log || { tail -20 $O2/bench_driver_command.progress.log; exit 1; }
python3 - $O2/bench_driver_command.json <<'PY'
import json, sys
b = json.load(open(sys.argv[1])); d = b["detail"]
print("value", b["value"], "stale", b["roofline"]["traffic_stale"], "frac", b["roofline"]["frac"])
for n in ("matrix_configs2", "matrix"):
    m = d[n]; pr = m["per_rank"][0]
    print(n, "total_s", m.get("total_s"), "jobs_s", m.get("jobs_s"), "set_wait_s", m.get("set_wait_s"), "device_ms", pr.get("device_ms"), "call_ms", pr.get("call_ms"), m.get("error"))
PY

# round 5, call 22: the final profile of the round's sources (rocprofv3 passes of bench.py, then the driver's own command with its progress log)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_final
rm -rf $O; mkdir -p $O
tools/exp/alloc_cost > $O/alloc_cost.log 2>&1; sed -n 2,3p $O/alloc_cost.log
bash tools/profile_bench.sh r05_final --steps 20 --warmup 5 > $O/profile.log 2>&1 || { tail -20 $O/profile.log; exit 1; }
tail -1 $O/profile.log | cut -c1-400
cd $GRAFT_REPO_ROOT
python3 bench.py --gpus 1 --steps 10 --warmup 3 > $O/bench_driver_command.json 2> $O/bench_driver_command.progress.log || { tail -20 $O/bench_driver_command.progress.log; exit 1; }
python3 - $O/bench_driver_command.json <<'PY'
import json, sys
b = json.load(open(sys.argv[1])); d = b["detail"]
print("value", b["value"], "roofline", b["roofline"], "cpu", b["cpu_baseline"])
for n in ("matrix_configs2", "matrix"):
    m = d[n]; pr = m["per_rank"][0]
    print(n, "total_s", m.get("total_s"), "jobs_s", m.get("jobs_s"), "set_wait_s", m.get("set_wait_s"), "device_ms", pr.get("device_ms"), "call_ms", pr.get("call_ms"), m.get("error"))
PY

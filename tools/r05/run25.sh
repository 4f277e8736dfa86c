# round 5, call 25: after the tidy-up of commet_index_many_and_search's exits: its tests, the final profile again, the driver's command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_final
rm -rf $O; mkdir -p $O
tools/exp/alloc_cost > $O/alloc_cost.log 2>&1; sed -n 2,3p $O/alloc_cost.log
python3 -m pytest tests/test_gpu_multi.py tests/test_gpu_matrix.py tests/test_gpu_cache.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
bash tools/profile_bench.sh r05_final --steps 20 --warmup 5 > $O/profile.log 2>&1 || { tail -20 $O/profile.log; exit 1; }
tail -1 $O/profile.log | cut -c1-300
cd $GRAFT_REPO_ROOT
python3 bench.py --gpus 1 --steps 10 --warmup 3 --traffic $O/traffic.json > $O/bench_driver_command.json 2> $O/bench_driver_command.progress.log || { tail -20 $O/bench_driver_command.progress.log; exit 1; }
python3 - $O/bench_driver_command.json <<'PY'
import json, sys
b = json.load(open(sys.argv[1])); d = b["detail"]
print("value", b["value"], "stale", b["roofline"]["traffic_stale"], b["roofline"]["kernel"], "frac", b["roofline"]["frac"])
for n in ("matrix_configs2", "matrix"):
    m = d[n]; pr = m["per_rank"][0]
    print(n, "total_s", m.get("total_s"), "jobs_s", m.get("jobs_s"), "set_wait_s", m.get("set_wait_s"), "device_ms", pr.get("device_ms"), "call_ms", pr.get("call_ms"), m.get("error"))
PY

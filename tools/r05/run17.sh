# round 5, call 17: the configs[3] leg with query lists for the 50 M-read sets (11 GB each, built for a set's second eligible scan) now that
# the library never frees device memory while the process lives: three runs of the driver's command with COMMET_QUERY_LIST_MAX_GB=16
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_large_lists
mkdir -p $O
for i in 1 2 3; do
  COMMET_QUERY_LIST_MAX_GB=16 python3 bench.py --gpus 1 --steps 10 --warmup 3 --cpu-sample 0 --no-probe-count > $O/bench_ql16_$i.json 2> $O/bench_ql16_$i.err
  python3 - $O/bench_ql16_$i.json <<'PY'
import json, sys
b = json.load(open(sys.argv[1])); d = b["detail"]
for n in ("matrix_configs2", "matrix"):
    m = d[n]; pr = m["per_rank"][0]
    print(n, "total_s", m.get("total_s"), "jobs_s", m.get("jobs_s"), "set_wait_s", m.get("set_wait_s"), "device_ms", pr.get("device_ms"), "call_ms", pr.get("call_ms"), m.get("error"))
PY
done

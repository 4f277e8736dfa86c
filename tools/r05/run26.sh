# round 5, call 26: blocks above 6 GiB from hipMallocAsync (the default now) against COMMET_DEVMEM_POOL=0: which kind of box, then the driver's bench command three times
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_pool6
mkdir -p $O
tools/exp/alloc_cost > $O/alloc_cost.log 2>&1; sed -n 2,3p $O/alloc_cost.log
for v in 1 0 1b; do
  COMMET_DEVMEM_POOL=${v:0:1} python3 bench.py --gpus 1 --steps 10 --warmup 3 --cpu-sample 0 --no-probe-count > $O/bench_p$v.json 2> $O/bench_p$v.err
  python3 - $O/bench_p$v.json $v <<'PY'
import json, sys
b = json.load(open(sys.argv[1])); d = b["detail"]
print("pool", sys.argv[2], "value", b["value"], "cold_first_job_ms", d.get("cold_context_first_job_ms"))
for n in ("matrix_configs2", "matrix"):
    m = d[n]; pr = m["per_rank"][0]
    print("  ", n, "total_s", m.get("total_s"), "jobs_s", m.get("jobs_s"), "set_wait_s", m.get("set_wait_s"), "device_ms", pr.get("device_ms"), "call_ms", pr.get("call_ms"), m.get("error"))
PY
done

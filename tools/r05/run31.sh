# round 5, call 31: ingest workers at nice +10 (the default now) against COMMET_INGEST_NICE=0: the configs[3] leg's calls, wall against device time
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_nice
mkdir -p $O; rm -f $O/calls_*.log
tools/exp/alloc_cost > $O/alloc_cost.log 2>&1; sed -n 2,2p $O/alloc_cost.log
for v in 10 0 10b; do
  COMMET_INGEST_NICE=${v%b} COMMET_MATRIX_CALL_LOG=$GRAFT_REPO_ROOT/$O/calls_$v.log python3 bench.py --gpus 1 --steps 3 --warmup 1 --cpu-sample 0 --no-probe-count --no-kernel-times > $O/bench_$v.json 2> $O/bench_$v.err
  python3 - $O/bench_$v.json $v $O/calls_$v.log <<'PY'
import json, sys
b = json.load(open(sys.argv[1])); d = b["detail"]
print("nice", sys.argv[2], "value", b["value"], "upload_and_pack_s", d.get("upload_and_pack_s"))
for n in ("matrix_configs2", "matrix"):
    m = d[n]; pr = m["per_rank"][0]
    print("  ", n, "total_s", m.get("total_s"), "jobs_s", m.get("jobs_s"), "set_wait_s", m.get("set_wait_s"), "load_s", m.get("load_s"), "parse_s", pr.get("parse_s"), "device_ms", pr.get("device_ms"), "call_ms", pr.get("call_ms"), m.get("error"))
rows = [l.split() for l in open(sys.argv[3])][-27:]
j1 = [r for i, r in enumerate(rows[:18]) if i % 2 == 0]
print("   J1 host-only ms:", [round(float(r[2]) - float(r[3]) - float(r[4]), 1) for r in j1])
PY
done

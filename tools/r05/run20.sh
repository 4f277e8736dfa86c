# round 5, call 20: batched J2 / J3 with small tiled-eligible jobs left on their own: tests, the driver's bench command twice, a 5-rank rehearsal
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_multi
mkdir -p $O
python3 -m pytest tests/test_gpu_multi.py tests/test_gpu_matrix.py tests/test_gpu_configs.py -x -q -m gpu -k "multi or matrix or c3 or c4" > $O/tests2.log 2>&1 || { tail -40 $O/tests2.log; exit 1; }
tail -3 $O/tests2.log
for i in 3 4; do
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
COMMET_FORCE_DEVICE=0 python3 bench.py --gpus 5 --reads 2000000 --matrix-reads 2000000 --steps 3 --warmup 1 > $O/bench_gpus5.json 2> $O/bench_gpus5.err; echo "gpus5 rc=$?"
python3 -c "
import json; b = json.load(open('$O/bench_gpus5.json')); m = b.get('matrix') or {}
print('gpus5', b['n_gpus'], {k: m.get(k) for k in ('error', 'total_s', 'handover')}, [p.get('jobs') for p in b['detail']['matrix']['per_rank']])"

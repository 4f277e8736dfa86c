# round 5, call 29: search_group8_kernel with ONE candidate sweep for the two strands of a filter: parity tests that reach the kernel, then the driver's bench command with per-kernel times
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_g8merge
mkdir -p $O
python3 -m pytest tests/test_gpu_multi.py tests/test_gpu_job.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
COMMET_MATRIX_KERNEL_TIMES=1 python3 bench.py --gpus 1 --steps 5 --warmup 2 --cpu-sample 0 --no-probe-count > $O/bench_kt.json 2> $O/bench_kt.err
python3 - $O/bench_kt.json <<'PY'
import json, sys
b = json.load(open(sys.argv[1])); d = b["detail"]
print("value", b["value"])
for n in ("matrix_configs2", "matrix"):
    m = d[n]; pr = m["per_rank"][0]
    print("  ", n, "total_s", m.get("total_s"), "jobs_s", m.get("jobs_s"), "set_wait_s", m.get("set_wait_s"), "device_ms", pr.get("device_ms"), "call_ms", pr.get("call_ms"), m.get("error"))
    print("     ", sorted(pr.get("kernel_ms", {}).items(), key=lambda kv: -kv[1][1])[:5])
PY

# round 5, call 11: the device-memory cache: parity suites that recycle the most memory, then bench.py as the driver runs it, twice
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_devmem
python3 -m pytest tests/test_gpu_cache.py tests/test_gpu_job.py tests/test_gpu_kernels.py tests/test_gpu_matrix.py tests/test_gpu_server.py tests/test_gpu_cli.py -x -q -m gpu > gpurun_out/r05_devmem/tests.log 2>&1 || { tail -30 gpurun_out/r05_devmem/tests.log; exit 1; }
tail -3 gpurun_out/r05_devmem/tests.log
for i in 1 2; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_devmem/bench_$i.json 2> gpurun_out/r05_devmem/bench_$i.err
  python3 - gpurun_out/r05_devmem/bench_$i.json <<'PY'
import json, sys
b = json.load(open(sys.argv[1]))
d = b["detail"]
print("value", b["value"], "ms/step", b["ms_per_step"], "idx", d["index_kernel_ms"], "srch", d["search_kernel_ms"], "first_job", d["first_job_ms"], "cold", d["cold_context_first_job_ms"])
for name in ("matrix_configs2", "matrix"):
    m = d[name]
    print("  ", name, "total_s", m.get("total_s"), "jobs_s", m.get("jobs_s"), "set_wait_s", m.get("set_wait_s"), "device_ms", m["per_rank"][0].get("device_ms"), m.get("error"))
PY
  grep "matrix leg: done\|jobs of set 0 done" gpurun_out/r05_devmem/bench_$i.err | tail -4
done

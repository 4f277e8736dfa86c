# round 5, call 9: bench.py as the driver runs it, twice (the second with per-kernel times in the matrix legs): is the configs[2]
# leg slower behind the headline than in a fresh process (tools/c2_variance.sh), and in which kernels?
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_c2_variance
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_c2_variance/bench_full_1.json 2> gpurun_out/r05_c2_variance/bench_full_1.err
COMMET_MATRIX_KERNEL_TIMES=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_c2_variance/bench_full_2_kt.json 2> gpurun_out/r05_c2_variance/bench_full_2_kt.err
python3 - <<'PY'
import json
for f in ("bench_full_1", "bench_full_2_kt"):
    b = json.load(open(f"gpurun_out/r05_c2_variance/{f}.json"))
    c2, c3 = b["detail"]["matrix_configs2"], b["detail"]["matrix"]
    print(f, "value", b["value"], "ms/step", b["ms_per_step"], "idx", b["detail"]["index_kernel_ms"], "srch", b["detail"]["search_kernel_ms"], "first_job", b["detail"]["first_job_ms"])
    for name, m in (("c2", c2), ("c3", c3)):
        pr = m["per_rank"][0]
        top = sorted((pr.get("kernel_ms") or {}).items(), key=lambda kv: -kv[1][1])[:8]
        print("  ", name, "total_s", m.get("total_s"), "jobs_s", m.get("jobs_s"), "device_ms", pr.get("device_ms"), {k: v[1] for k, v in top})
PY

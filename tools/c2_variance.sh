# The configs[2] matrix leg (10 x 10 M reads through the driver, one GPU) five times in FRESH processes, per-kernel times on, the
# FASTA files written once: where do a fast and a slow run differ?   bash tools/c2_variance.sh <out dir> [runs]
set -e
cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r05_c2_variance}; N=${2:-5}
mkdir -p $O
export COMMET_BENCH_DIR=/dev/shm/commet_c2var COMMET_BENCH_KEEP=1
for i in $(seq 1 $N); do
  COMMET_MATRIX_KERNEL_TIMES=1 python3 tools/matrix_bench.py 10 10000000 32 > $O/run_kt_$i.json 2> $O/run_kt_$i.err
  python3 tools/matrix_bench.py 10 10000000 32 > $O/run_$i.json 2> $O/run_$i.err
  python3 - $O/run_$i.json $O/run_kt_$i.json <<'PY'
import json, sys
a, b = (json.load(open(p)) for p in sys.argv[1:3])
pr = b["per_rank"][0]
top = sorted(pr.get("kernel_ms", {}).items(), key=lambda kv: -kv[1][1])[:8]
print("run", sys.argv[1], "total_s", round(a["total_s"], 3), "jobs_s", round(a["jobs_s"], 3), "device_ms", round(a["per_rank"][0]["device_ms"], 1),
      "| timed run: total_s", round(b["total_s"], 3), "device_ms", round(pr["device_ms"], 1), {k: v[1] for k, v in top})
PY
done
rm -rf /dev/shm/commet_c2var

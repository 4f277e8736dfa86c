# The configs[3] matrix leg (10 x 50 M reads through the driver, one GPU) in fresh processes under several values of one environment
# knob, the FASTA files written once: what do the parser's threads cost the job thread?
#   bash tools/c3_ingest_ab.sh <out dir> <VARIABLE> <value> [<value> ...]
set -e
cd $GRAFT_REPO_ROOT
O=$1; V=$2; shift 2
mkdir -p $O
export COMMET_BENCH_DIR=/dev/shm/commet_c3ab COMMET_BENCH_KEEP=1
echo "nproc $(nproc), online $(getconf _NPROCESSORS_ONLN)" | tee $O/host.txt
i=0
for v in "$@"; do
  i=$((i+1))
  env $V=$v python3 tools/matrix_bench.py 10 50000000 32 > $O/run_${i}_$v.json 2> $O/run_${i}_$v.err || { tail -5 $O/run_${i}_$v.err; rm -rf $COMMET_BENCH_DIR; exit 1; }
  python3 - $O/run_${i}_$v.json $V $v <<'PY' | tee -a $O/summary.txt
import json, sys
a = json.load(open(sys.argv[1])); p = a["per_rank"][0]
over = sum(r[5] - r[3] - r[4] for r in p.get("job_log", []))
print(sys.argv[2], "=", sys.argv[3], "total_s", round(a["total_s"], 3), "jobs_s", round(a["jobs_s"], 3), "set_wait_s", round(a["set_wait_s"], 3), "parse_s", round(p["parse_s"], 3),
      "call_ms", round(p["call_ms"], 1), "device_ms", round(p["device_ms"], 1), "calls' host time outside the kernels' ms", round(over, 1))
PY
done
rm -rf $COMMET_BENCH_DIR

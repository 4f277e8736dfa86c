# round 5, call 10: why is the configs[2] leg slower behind the headline (device 1500 ms) than in a fresh process (1150 ms)?
# per-kernel stats (rocprofv3 --kernel-trace --stats) of (a) the leg alone in a fresh process, (b) bench.py with the leg behind the
# headline, (c) the same with one index lane
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c2_variance
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export COMMET_BENCH_DIR=/dev/shm/commet_c2var COMMET_BENCH_KEEP=1
python3 $R/tools/matrix_bench.py 10 10000000 32 > $O/fresh0.json 2> $O/fresh0.err      # (writes the files)
rocprofv3 --kernel-trace --stats -d $O/kt_fresh -o p --output-format csv -- python3 $R/tools/matrix_bench.py 10 10000000 32 > $O/fresh_prof.json 2> $O/fresh_prof.err
rm -rf /dev/shm/commet_c2var
B="--steps 20 --warmup 5 --cpu-sample 0 --no-probe-count --matrix-reads 10000000"
rocprofv3 --kernel-trace --stats -d $O/kt_bench -o p --output-format csv -- python3 $R/bench.py $B > $O/bench_prof.json 2> $O/bench_prof.err
COMMET_INDEX_LANES=1 python3 $R/bench.py $B > $O/bench_lanes1.json 2> $O/bench_lanes1.err
python3 $R/bench.py $B > $O/bench_lanes2.json 2> $O/bench_lanes2.err
cp $(find $O/kt_fresh -name "*kernel_stats.csv" | head -1) $O/kernel_stats_fresh.csv
cp $(find $O/kt_bench -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bench.csv
python3 - <<'PY'
import csv, json, os
O = os.environ.get("GRAFT_REPO_ROOT") + "/gpurun_out/r05_c2_variance"
def stats(p):
    return {r["Name"].split("(")[0][-60:]: (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(p))}
a, b = stats(O + "/kernel_stats_fresh.csv"), stats(O + "/kernel_stats_bench.csv")
print("kernel: fresh leg alone (calls, ms) | bench = headline + leg (calls, ms)")
for k in sorted(a, key=lambda k: -a[k][1])[:12]:
    print(f"  {k:62s} {a[k][0]:6d} {a[k][1]:9.1f} | {b.get(k, (0, 0))[0]:6d} {b.get(k, (0, 0))[1]:9.1f}")
for f in ("fresh0", "fresh_prof"):
    m = json.load(open(f"{O}/{f}.json")); print(f, "total_s", round(m["total_s"], 3), "device_ms", round(m["per_rank"][0]["device_ms"], 1))
for f in ("bench_prof", "bench_lanes1", "bench_lanes2"):
    m = json.load(open(f"{O}/{f}.json")); mm = m["detail"]["matrix"]
    print(f, "value", m["value"], "leg total_s", mm["total_s"], "device_ms", mm["per_rank"][0]["device_ms"])
PY
rm -rf $O/kt_fresh $O/kt_bench

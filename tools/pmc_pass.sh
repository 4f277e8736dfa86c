# one rocprofv3 counter pass of a single bench step on the GPU box, summed per kernel:
#   [BENCH_ARGS="--reads 50000000"] bash tools/pmc_pass.sh <name> COUNTER [COUNTER ...]      -> gpurun_out/<name>.csv (+ a table on stdout)
R=$GRAFT_REPO_ROOT
N=$1; shift
cd /tmp && export TMPDIR=/tmp
COMMET_INDEX_LANES=1 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pp_$N -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-probe-count --no-kernel-times --no-matrix $BENCH_ARGS > /tmp/pp_$N.log 2>&1 || { tail -20 /tmp/pp_$N.log; exit 1; }
cp $(find /tmp/pp_$N -name "*counter_collection.csv" | head -1) $R/gpurun_out/$N.csv
python3 - $R/gpurun_out/$N.csv <<'P'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("commet::", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); calls[k] += 1
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", kv[1].get("SQ_WAVES", 0))):
    if calls[k] > 60 or k.startswith("__amd"): continue
    print(k, calls[k], {c: round(x / calls[k] / 1e6, 3) for c, x in v.items()}, "(M per launch)")
P

# round 5, call 33: HIP API trace of the driver's bench command: which calls of the job thread take tens of ms beside a loading set, and what the loader does meanwhile
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_hiptrace
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
$GRAFT_REPO_ROOT/tools/exp/alloc_cost > $O/alloc_cost.log 2>&1; sed -n 2,2p $O/alloc_cost.log
COMMET_MATRIX_CALL_LOG=$O/calls.log rocprofv3 --hip-runtime-trace -d $O/tr -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 2 --warmup 1 --cpu-sample 0 --no-probe-count --no-kernel-times > $O/bench.json 2> $O/bench.err
ls -la $O/tr/* | head
F=$(find $O/tr -name "*hip_api_trace.csv" | head -1)
python3 - $F $O <<'PY'
import csv, sys, collections
f, out = sys.argv[1], sys.argv[2]
rows = []
with open(f) as fh:
    rd = csv.DictReader(fh)
    for r in rd:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], int(r["Thread_Id"])))
print(len(rows), "api calls")
t_end = max(r[1] for r in rows)
# the configs[3] leg = the last ~12 s; long calls (> 8 ms) there by thread
leg0 = t_end - 12_000_000_000
long_calls = [r for r in rows if r[0] > leg0 and r[1] - r[0] > 8_000_000]
by = collections.Counter((r[3], r[2]) for r in long_calls)
tot = collections.Counter()
for r in long_calls: tot[(r[3], r[2])] += (r[1] - r[0]) / 1e6
for k, n in by.most_common(25): print(k, n, "calls", round(tot[k], 1), "ms")
with open(out + "/long_calls.txt", "w") as fh:
    for r in sorted(long_calls): fh.write(f"{(r[0]-leg0)/1e6:10.1f} ms +{(r[1]-r[0])/1e6:8.1f} ms tid {r[3]} {r[2]}\n")
PY
rm -rf $O/tr

# rocprofv3 passes of bench.py on the GPU box: bash tools/profile_bench.sh <name> [extra bench.py args]
#   ->  gpurun_out/<name>/{bench.json,kernel_stats.csv,pmc_fetch_size.csv,pmc_write_size.csv,pmc_sq.csv,traffic.json}
# Copy the directory to profiles/<name> to have bench.py use (and the judge read) it.
set -e
R=$GRAFT_REPO_ROOT
N=${1:-r02_profile}
shift || true
O=$R/gpurun_out/$N
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --no-matrix "$@" > $O/bench.json 2> $O/bench.err
# one index lane here: with two, kernels of both lanes run at once and their durations are not additive
# (the counter passes time ONE workload: no ragged leg beside it — pass --ragged-only to profile the ragged step itself)
B="--cpu-sample 0 --no-probe-count --no-kernel-times --no-matrix --ragged none"
COMMET_INDEX_LANES=1 rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 $B "$@" > $O/kt.log 2>&1
COMMET_INDEX_LANES=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pf -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 $B "$@" > $O/pf.log 2>&1
COMMET_INDEX_LANES=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pw -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 $B "$@" > $O/pw.log 2>&1
COMMET_INDEX_LANES=1 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace -d $O/ps -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 $B "$@" > $O/ps.log 2>&1 || echo "SQ pass failed" >> $O/bench.err
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cp $(find $O/pf -name "*counter_collection.csv" | head -1) $O/pmc_fetch_size.csv
cp $(find $O/pw -name "*counter_collection.csv" | head -1) $O/pmc_write_size.csv
cp $(find $O/ps -name "*counter_collection.csv" | head -1) $O/pmc_sq.csv 2>/dev/null || true
rm -rf $O/kt $O/pf $O/pw $O/ps
cd $R && python3 tools/pmc_summary.py $O
# the bench line again, now priced with the counters just collected (the first one only named the workload)
cd /tmp && python3 $R/bench.py --no-matrix --traffic $O/traffic.json "$@" > $O/bench.json 2> $O/bench.err
cat $O/bench.json

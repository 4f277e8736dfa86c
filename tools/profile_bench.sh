# rocprofv3 passes of bench.py on the GPU box: bash tools/profile_bench.sh <name>  ->  gpurun_out/<name>/{bench.json,kernel_stats.csv,pmc_*.csv,traffic.json}
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r01_profile}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
# one index lane here: with two, kernels of both lanes run at once and their durations are not additive
COMMET_INDEX_LANES=1 rocprofv3 --kernel-trace --stats -d $O/kt -o r01 --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $O/kt.log 2>&1
COMMET_INDEX_LANES=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pf -o r01 --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-probe-count > $O/pf.log 2>&1
COMMET_INDEX_LANES=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pw -o r01 --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-probe-count > $O/pw.log 2>&1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cp $(find $O/pf -name "*counter_collection.csv" | head -1) $O/pmc_fetch_size.csv
cp $(find $O/pw -name "*counter_collection.csv" | head -1) $O/pmc_write_size.csv
rm -rf $O/kt $O/pf $O/pw
cd $R && python3 tools/pmc_summary.py $O
cat $O/bench.json

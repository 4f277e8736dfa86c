# BASELINE configs[4] at full size (k=21, t=5, 2 x 20 M x 150 bp) on the GPU box: bash tools/c5_full.sh -> gpurun_out/r02_c5/
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_c5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--reads 20000000 --read-len 150 -k 21 -t 5 --steps 1 --warmup 0 --no-matrix --cpu-sample 0 --no-probe-count"
python3 $R/bench.py $B --kt-steps 1 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python3 $R/bench.py $B --no-kernel-times > $O/kt.log 2>&1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/kt
cat $O/bench.json

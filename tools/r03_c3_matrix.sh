# BASELINE configs[3] (10 sets x 50 M reads, the whole 10 x 10 matrix) through bench.py's matrix leg on ONE GPU, then the
# rocprofv3 passes of configs[4] with the wide rows:  bash tools/r03_c3_matrix.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --matrix-reads 50000000 > $O/r03_bench_with_matrix_c3.json 2> $O/r03_bench_with_matrix_c3.err
tail -c 1500 $O/r03_bench_with_matrix_c3.json
tail -5 $O/r03_bench_with_matrix_c3.err
bash $R/tools/profile_bench.sh r03_c5 -k 21 -t 5 --reads 20000000 --read-len 150 --steps 1 --warmup 0 --cpu-sample 0 --no-probe-count --kt-steps 1 > $O/r03_c5.log 2>&1
tail -12 $O/r03_c5.log

# round 5, call 6: parity (job scenarios incl. the forced sparse passes, kernels), then scatter1's replicated rank counters and the build's
# deeper unroll on configs[1], then a 50 M-read pair chain with and without the list form of sparse passes
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_ab
python3 -m pytest tests/test_gpu_job.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r05_ab/tests6.log
bash tools/ab_flags.sh "-DS1_REPL=1 -DBUILD_U8=0" "-DS1_REPL=4 -DBUILD_U8=0" "-DS1_REPL=2 -DBUILD_U8=0" "-DS1_REPL=4 -DBUILD_U8=1" 2>&1 | tee gpurun_out/r05_ab/ab6.log
summ() { python3 -c "
import json, sys
o = json.loads(sys.stdin.readlines()[-1])
print(sys.argv[1], {j: (v['total_ms'], v['index_ms'], v['search_ms'], {k: x[1] for k, x in v['kernels'].items() if k.startswith(('search', 'active'))}) for j, v in o.items()})" "$1"; }
COMMET_SPARSE_SEARCH=1 python3 tools/j2_anatomy.py 50000000 2>/dev/null | summ "bitmap form" | tee -a gpurun_out/r05_ab/j2_6.log
python3 tools/j2_anatomy.py 50000000 2>/dev/null | summ "list form (auto)" | tee -a gpurun_out/r05_ab/j2_6.log

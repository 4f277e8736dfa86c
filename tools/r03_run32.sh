R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 1000 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_job.py tests/test_gpu_configs.py tests/test_gpu_matrix.py -m gpu -x -q -k "bucketed or c3 or c4 or matrix or job_matches or long_and_ragged" > $O/r03_run32_tests.log 2>&1 || { tail -30 $O/r03_run32_tests.log; exit 1; }
tail -2 $O/r03_run32_tests.log
python3 tools/r03_j2_anatomy.py

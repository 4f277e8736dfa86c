# round 5, call 16: the group kernels' lane-a gathers batched (GATHER_U windows per round): a 50 M-read pair chain per value
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_ab
summ() { python3 -c "
import json, sys
o = json.loads(sys.stdin.readlines()[-1])
print(sys.argv[1], {j: (v['total_ms'], v['index_ms'], v['search_ms'], {k: x[1] for k, x in v['kernels'].items() if k.startswith(('search', 'active'))}) for j, v in o.items()})" "$1"; }
for f in "-DGATHER_U=1" "-DGATHER_U=2" "-DGATHER_U=4" "-DGATHER_U=8"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC $f -o commet_amd/libcommet_hip.so commet_amd/csrc/capi.hip -lz
  python3 tools/j2_anatomy.py 50000000 2>/dev/null | summ "$f" | tee -a gpurun_out/r05_ab/j2_16.log
done
python3 -m commet_amd.build --force > /dev/null
python3 -m pytest tests/test_gpu_job.py tests/test_gpu_configs.py -x -q -m gpu -k "group8 or job_matches or sparse or long_and_ragged" > gpurun_out/r05_ab/tests16.log 2>&1 || { tail -20 gpurun_out/r05_ab/tests16.log; exit 1; }
tail -2 gpurun_out/r05_ab/tests16.log

# round 5, call 12: search_group8_kernel with FEWER workgroups per CU (more registers per thread): 5 (default) against 4, 3
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_ab
summ() { python3 -c "
import json, sys
o = json.loads(sys.stdin.readlines()[-1])
print(sys.argv[1], {j: (v['total_ms'], v['index_ms'], v['search_ms'], {k: x[1] for k, x in v['kernels'].items() if k.startswith(('search', 'active'))}) for j, v in o.items()})" "$1"; }
for f in "-DG8_WAVES=1" "-DG8_WAVES=4 -DCOMMET_SGPR_CAP=0" "-DG8_WAVES=3 -DCOMMET_SGPR_CAP=0" "-DG8_HEAVY=24" "-DGROUP8_TAIL_WIN=16"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC $f -o commet_amd/libcommet_hip.so commet_amd/csrc/capi.hip -lz
  python3 tools/j2_anatomy.py 50000000 2>/dev/null | summ "$f" | tee -a gpurun_out/r05_ab/j2_12.log
done
python3 -m commet_amd.build --force > /dev/null

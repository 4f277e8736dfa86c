# round 5, call 5: scatter2 with four workgroups per CU (448 threads, 72 VGPRs); search_group8_kernel held to 7 / 8 waves per SIMD;
# search_group_kernel without the LDS copy of the reads (eight workgroups per CU instead of five)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_ab
bash tools/ab_flags.sh "" "-DCOMMET_S2_NT=448 -DS2P_WAVES=7" "-DCOMMET_S2_NT=384 -DS2P_WAVES=6" 2>&1 | tee gpurun_out/r05_ab/ab5.log
summ() { python3 -c "
import json, sys
o = json.loads(sys.stdin.readlines()[-1])
print(sys.argv[1], {j: (v['total_ms'], v['index_ms'], v['search_ms'], {k: x[1] for k, x in v['kernels'].items() if k.startswith('search')}) for j, v in o.items()})" "$1"; }
for w in 1 7 8; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DG8_WAVES=$w -o commet_amd/libcommet_hip.so commet_amd/csrc/capi.hip -lz
  python3 tools/j2_anatomy.py 50000000 2>/dev/null | summ "G8_WAVES=$w" | tee -a gpurun_out/r05_ab/j2_5.log
done
COMMET_NO_STAGE_READS=1 python3 tools/j2_anatomy.py 50000000 2>/dev/null | summ "G8_WAVES=8 NO_STAGE_READS" | tee -a gpurun_out/r05_ab/j2_5.log
python3 -m commet_amd.build --force > /dev/null

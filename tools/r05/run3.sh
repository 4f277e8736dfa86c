# round 5, call 3: parity of the job-level scenarios (every search regime forced), then the direct next-window probe A/B on configs[1]
# and on a 2 x 50 M-read pair chain (J1 group8, J2 / J3 group<2>)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_ab
python3 -m pytest tests/test_gpu_job.py tests/test_gpu_kernels.py -x -q -m gpu --durations=5 2>&1 | tail -12 | tee gpurun_out/r05_ab/tests3.log
bash tools/ab_flags.sh "-DCOMMET_DIRECT_NEXT=0" "-DCOMMET_DIRECT_NEXT=1" 2>&1 | tee gpurun_out/r05_ab/ab3.log
for d in 0 1; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DCOMMET_DIRECT_NEXT=$d -o commet_amd/libcommet_hip.so commet_amd/csrc/capi.hip -lz
  echo "DIRECT_NEXT=$d" | tee -a gpurun_out/r05_ab/j2_3.log
  python3 tools/j2_anatomy.py 50000000 2>&1 | tail -1 | tee -a gpurun_out/r05_ab/j2_3.log
  python3 bench.py --no-matrix --cpu-sample 0 --no-probe-count --steps 3 --kt-steps 2 --reads 50000000 > /tmp/b50.json 2>/tmp/b50.err
  python3 -c "
import json; b = json.load(open('/tmp/b50.json'))
print('2x50M', b['ms_per_step'], {k: round(v['ms_per_step'], 2) for k, v in b['roofline']['kernels'].items() if v['ms_per_step'] > 0.5})" | tee -a gpurun_out/r05_ab/j2_3.log
done
python3 -m commet_amd.build --force > /dev/null

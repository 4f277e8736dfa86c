# ablation timings of the tiled search's replay kernel on the GPU box (builds two throw-away libraries)
set -e
cd $GRAFT_REPO_ROOT
cp commet_amd/libcommet_hip.so /tmp/lib_ok.so
for a in 1024 8192 16384 32768; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DCOMMET_TQ_ABLATE=$a -o commet_amd/libcommet_hip.so commet_amd/csrc/capi.hip -lz
  python3 bench.py --no-matrix --cpu-sample 0 --no-probe-count --steps 2 > gpurun_out/r02_tq_abl_$a.json 2> gpurun_out/r02_tq_abl_$a.err || true
  python3 -c "
import json;b=json.load(open('gpurun_out/r02_tq_abl_$a.json'));print($a, {k:round(v['ms_per_step'],2) for k,v in b['roofline']['kernels'].items() if k.startswith('tq')})"
done
cp /tmp/lib_ok.so commet_amd/libcommet_hip.so

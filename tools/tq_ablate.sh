# ablation timings of the tiled search's replay kernel on the GPU box (throw-away libraries; results are wrong on purpose)
#   bash tools/tq_ablate.sh 0 512 1024 8192 32768
cd $GRAFT_REPO_ROOT
for a in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DCOMMET_TQ_ABLATE=$a -o commet_amd/libcommet_hip.so commet_amd/csrc/capi.hip -lz || exit 1
  python3 bench.py --no-matrix --cpu-sample 0 --no-probe-count --steps 2 --kt-steps 2 > /tmp/o.json 2> /tmp/o.err || { tail -3 /tmp/o.err; exit 1; }
  python3 -c "
import json;b=json.load(open('/tmp/o.json'));print($a, b['detail']['shared'], {k:round(v['ms_per_step'],2) for k,v in b['roofline']['kernels'].items() if k.startswith('tq')})"
done

# ablation timings of scatter2 on the GPU box (throw-away libraries; results are wrong on purpose)
cd $GRAFT_REPO_ROOT
for a in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DS2_PREFETCH=0 -DCOMMET_ABLATE=$a -o commet_amd/libcommet_hip.so commet_amd/csrc/capi.hip -lz || exit 1
  COMMET_S2_WGS_PER_CU=1000 python3 bench.py --no-matrix --cpu-sample 0 --no-probe-count --steps 2 --kt-steps 2 > /tmp/o.json 2>/tmp/o.err || { tail -5 /tmp/o.err; exit 1; }
  python3 -c "
import json,sys;b=json.load(open('/tmp/o.json'));print(sys.argv[1:], {k:round(v['ms_per_step'],2) for k,v in b['roofline']['kernels'].items() if k.startswith('part_s')})" "$a"
done

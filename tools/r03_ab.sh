# A/B on the GPU box: (1) tiled search in 1..4 parts (replay of a part beside the next part's probe), (2) scatter2 ablations on skewed data
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp
for p in 1 2 3 4 6; do
  COMMET_TQ_PARTS=$p python3 $R/bench.py --no-matrix --cpu-sample 0 --no-probe-count --no-kernel-times --steps 10 > $O/r03_parts_$p.json 2>/dev/null
  python3 -c "
import json,sys;b=json.load(open('$O/r03_parts_$p.json'));print('parts', $p, b['value'], b['ms_per_step'], b['detail']['index_kernel_ms'], b['detail']['search_kernel_ms'])"
done
cd $R
for a in 0 128 32 256; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DCOMMET_ABLATE=$a -o commet_amd/libcommet_hip.so commet_amd/csrc/capi.hip -lz || exit 1
  (cd /tmp && python3 $R/bench.py --skew 0.1 --no-matrix --cpu-sample 0 --no-probe-count --steps 2 --kt-steps 2 > /tmp/o.json 2>/tmp/o.err) || { tail -5 /tmp/o.err; exit 1; }
  python3 -c "
import json,sys;b=json.load(open('/tmp/o.json'));print('ablate', sys.argv[1:], {k:round(v['ms_per_step'],2) for k,v in b['roofline']['kernels'].items() if k.startswith('part_s')})" "$a"
done

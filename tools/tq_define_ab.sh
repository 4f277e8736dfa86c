# A/B runs of compile-time constants on the GPU box: rebuilds the library with each -D set (and environment), runs the configs[1] step.
#   bash tools/tq_define_ab.sh "-DTQ_TAIL_FIRST=4" "-DTQ_TAIL_FIRST=32;COMMET_X=1" ...
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  d="${spec%%;*}"; e="A=1"; [ "$spec" != "$d" ] && e="${spec#*;}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC $d -o commet_amd/libcommet_hip.so commet_amd/csrc/capi.hip -lz || exit 1
  env $e python3 bench.py --no-matrix --cpu-sample 0 --no-probe-count --steps ${STEPS:-5} --kt-steps 2 $BENCH_ARGS > /tmp/o.json 2>/tmp/o.err || { tail -5 /tmp/o.err; exit 1; }
  python3 -c "
import json,sys;b=json.load(open('/tmp/o.json'));print(sys.argv[1:], b['value'], b['ms_per_step'], b['detail']['shared'], b['detail']['index_kernel_ms'], b['detail']['search_kernel_ms'], {k:round(v['ms_per_step'],2) for k,v in b['roofline']['kernels'].items() if k.startswith('${KFILTER:-part_}')})" "$spec"
done

# Timing ablations and compile-time A/Bs on the GPU box: throw-away libraries built with -D<MACRO>=<value>, one bench.py run each,
# the kernels' times side by side.  Results of ablation builds are wrong on purpose; the shipped library is rebuilt at the end.
#   bash tools/ablate.sh <MACRO> <value> [<value> ...] [-- <bench.py args>]
#   e.g.  bash tools/ablate.sh COMMET_TQ_ABLATE 0 512 1024 8192 32768            (tile_search.hpp: replay phases)
#         bash tools/ablate.sh COMMET_ABLATE 0 32 64 128 256 -- --skew 0.1       (index_part.hpp: scatter phases, skewed data)
#         bash tools/ablate.sh COMMET_S1_NT 512 1024                             (scatter1 workgroup size)
set -e
cd $GRAFT_REPO_ROOT
M=$1; shift
VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done
[ "$1" = "--" ] && shift
for v in "${VALS[@]}"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -D$M=$v -o commet_amd/libcommet_hip.so commet_amd/csrc/capi.hip -lz
  python3 bench.py --no-matrix --cpu-sample 0 --no-probe-count --steps 3 --kt-steps 2 "$@" > /tmp/ab.json 2> /tmp/ab.err || { tail -5 /tmp/ab.err; exit 1; }
  python3 -c "
import json, sys; b = json.load(open('/tmp/ab.json'))
print(sys.argv[1], '=', sys.argv[2], 'ms/step', b['ms_per_step'], 'shared', b['detail']['shared'], {k: round(v['ms_per_step'], 2) for k, v in b['roofline']['kernels'].items() if v['ms_per_step'] > 0.05})" $M $v
done
python3 -m commet_amd.build --force > /dev/null

# Compile-time A/Bs on the GPU box with several macros at once: one throw-away library per flag set, one bench.py run each, the
# kernels' times side by side; the shipped library is rebuilt at the end.
#   bash tools/ab_flags.sh "<flags of build 1>" "<flags of build 2>" ... [-- <bench.py args>]
#   e.g.  bash tools/ab_flags.sh "" "-DCOMMET_TQ_REPLAY_SGPR=80" "-DCOMMET_TQ_REPLAY_SGPR=80 -DTQ_COLLECT_U=8"
set -e
cd $GRAFT_REPO_ROOT
SETS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do SETS+=("$1"); shift; done
[ "$1" = "--" ] && shift
for f in "${SETS[@]}"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC $f -o commet_amd/libcommet_hip.so commet_amd/csrc/capi.hip -lz
  python3 bench.py --no-matrix --cpu-sample 0 --no-probe-count --steps 10 --kt-steps 3 "$@" > /tmp/ab.json 2> /tmp/ab.err || { tail -5 /tmp/ab.err; exit 1; }
  python3 -c "
import json, sys; b = json.load(open('/tmp/ab.json'))
print('[' + sys.argv[1] + ']', 'ms/step', b['ms_per_step'], 'reads/s', b['value'], 'shared', b['detail']['shared'], 'idx', b['detail']['index_kernel_ms'], 'srch', b['detail']['search_kernel_ms'], {k: round(v['ms_per_step'], 2) for k, v in b['roofline']['kernels'].items() if v['ms_per_step'] > 0.05})" "$f"
done
python3 -m commet_amd.build --force > /dev/null

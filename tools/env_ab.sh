# Run-time A/Bs on the GPU box: bench.py under every value of one environment knob (read once in commet_create, capi/context.hpp),
# one line per value with the step and the kernels' times.
#   bash tools/env_ab.sh <VARIABLE> <value> [<value> ...] [-- <bench.py args>]
#   e.g.  bash tools/env_ab.sh COMMET_TQ_WPX 16 32 64 128                                            (probe workgroups per XCD)
#         bash tools/env_ab.sh COMMET_TQ_PARTS 1 2 3 4
#         bash tools/env_ab.sh COMMET_SLICE_WIDE 1 2 -- --reads 20000000 --read-len 150 -k 21 -t 5   (configs[4]: narrow tables / wide rows)
#         bash tools/env_ab.sh COMMET_INDEX_LANES 1 2
set -e
cd $GRAFT_REPO_ROOT
V=$1; shift
VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done
[ "$1" = "--" ] && shift
for v in "${VALS[@]}"; do
  env $V=$v python3 bench.py --no-matrix --cpu-sample 0 --no-probe-count --steps 5 --kt-steps 2 "$@" > /tmp/ab.json 2> /tmp/ab.err || { tail -5 /tmp/ab.err; exit 1; }
  python3 -c "
import json, sys; b = json.load(open('/tmp/ab.json'))
print(sys.argv[1], '=', sys.argv[2], 'ms/step', b['ms_per_step'], 'reads/s', b['value'], {k: round(v['ms_per_step'], 2) for k, v in b['roofline']['kernels'].items() if v['ms_per_step'] > 0.05})" $V $v
done

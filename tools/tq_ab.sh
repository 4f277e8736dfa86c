# A/B runs of the configs[1] step under environment switches (one line per run):  bash tools/tq_ab.sh "COMMET_PIPELINE=2" "COMMET_PIPELINE=1" ...
cd $GRAFT_REPO_ROOT
run() { env $1 python3 bench.py --no-matrix --cpu-sample 0 --no-probe-count --steps 5 --kt-steps 2 > /tmp/o.json 2>/tmp/o.err || { tail -5 /tmp/o.err; return 1; }; python3 -c "
import json,sys;b=json.load(open('/tmp/o.json'));print(sys.argv[1:], b['value'], b['ms_per_step'], b['detail']['shared'], b['detail']['index_kernel_ms'], b['detail']['search_kernel_ms'], {k:round(v['ms_per_step'],2) for k,v in b['roofline']['kernels'].items()})" "$1"; }
for cfg in "$@"; do run "$cfg" || exit 1; done

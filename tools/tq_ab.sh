cd $GRAFT_REPO_ROOT
run() { env "$@" python3 bench.py --no-matrix --cpu-sample 0 --no-probe-count --steps 3 --kt-steps 2 > /tmp/o.json 2>/tmp/o.err; python3 -c "
import json,sys;b=json.load(open('/tmp/o.json'));print(sys.argv[1:], b['value'], b['ms_per_step'], b['detail']['shared'], {k:round(v['ms_per_step'],2) for k,v in b['roofline']['kernels'].items() if k.startswith('tq') or k.startswith('search')})" "$@"; }
run COMMET_TILED=2

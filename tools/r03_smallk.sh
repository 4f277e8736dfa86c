# 10 M-read sets at small k (the many-small-chunks regime in auto mode): where the time goes
R=$GRAFT_REPO_ROOT
cd /tmp
for k in 24 23 22; do
python3 $R/bench.py -k $k --no-matrix --cpu-sample 0 --no-probe-count --steps 2 --warmup 1 --kt-steps 1 > /tmp/o.json 2>/tmp/o.err || { tail -3 /tmp/o.err; continue; }
python3 -c "
import json
d=json.load(open('/tmp/o.json')); print('k', $k, 'chunks', d['detail']['chunks'], d['value'], d['ms_per_step'], d['detail']['shared'], {k:round(v['ms_per_step'],2) for k,v in d['roofline']['kernels'].items() if v['ms_per_step']>0.5})"
done

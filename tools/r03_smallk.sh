# 10 M-read sets at small k (the many-small-chunks regime in auto mode: probe, then narrow tables or wide rows)
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_job.py -m gpu -x -q -k "auto_mode or wide or c5 or sliced" 2>&1 | tail -3
cd /tmp
for k in 24 22 20 18 16; do
python3 $R/bench.py -k $k --no-matrix --cpu-sample 0 --no-probe-count --steps 2 --warmup 1 --kt-steps 1 > /tmp/o.json 2>/tmp/o.err || { tail -3 /tmp/o.err; continue; }
python3 -c "
import json
d=json.load(open('/tmp/o.json')); print('k', $k, 'chunks', d['detail']['chunks'], d['value'], d['ms_per_step'], d['detail']['shared'], {k:(round(v['ms_per_step'],2), v['launches_per_step']) for k,v in d['roofline']['kernels'].items() if v['ms_per_step']>0.5})"
done
python3 $R/bench.py -k 21 -t 5 --reads 20000000 --read-len 150 --no-matrix --cpu-sample 0 --no-probe-count --steps 1 --warmup 0 --kt-steps 1 > /tmp/o.json 2>/tmp/o.err
python3 -c "
import json
d=json.load(open('/tmp/o.json')); print('C5', d['ms_per_step'], {k:(round(v['ms_per_step'],2), v['launches_per_step']) for k,v in d['roofline']['kernels'].items() if v['ms_per_step']>0.5})"

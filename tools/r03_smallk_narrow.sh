R=$GRAFT_REPO_ROOT
cd /tmp
for k in 24 22 20; do
COMMET_SLICE_WIDE=1 timeout -k 10 300 python3 $R/bench.py -k $k --no-matrix --cpu-sample 0 --no-probe-count --steps 1 --warmup 0 --no-kernel-times > /tmp/o.json 2>/tmp/o.err || { tail -3 /tmp/o.err; continue; }
python3 -c "
import json
d=json.load(open('/tmp/o.json')); print('narrow k', $k, 'chunks', d['detail']['chunks'], d['value'], d['ms_per_step'])"
done

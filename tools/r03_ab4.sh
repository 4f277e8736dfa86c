# narrow tables vs wide rows for jobs of 8..256 chunk filters (k = 21, t = 5, 150 bp)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp
B="-k 21 -t 5 --read-len 150 --steps 3 --warmup 1 --cpu-sample 0 --no-probe-count --no-matrix --no-kernel-times"
for n in 30000 100000 200000 450000; do
  for w in 0 2; do
    COMMET_SLICE_WIDE=$w python3 $R/bench.py $B --reads $n > /tmp/o.json 2>/dev/null
    python3 -c "
import json; d=json.load(open('/tmp/o.json')); print('reads', $n, 'wide', $w, 'chunks', d['detail']['chunks'], 'ms', d['ms_per_step'])"
  done
done

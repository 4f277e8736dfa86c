cd $GRAFT_REPO_ROOT
for m in 2 1; do
COMMET_TILED=$m COMMET_BENCH_KEEP=1 python3 tools/matrix_bench.py 10 10000000 32 > /tmp/m.json 2>/tmp/m.err || tail -5 /tmp/m.err
python3 -c "
import json;b=json.load(open('/tmp/m.json'));print('tiled mode', $m, {k:(round(v,3) if isinstance(v,float) else v) for k,v in b.items() if k in ('filter_s','load_s','jobs_s','reads_per_s','total_s')})"
done
rm -rf /tmp/commet_matrix_10_10000000

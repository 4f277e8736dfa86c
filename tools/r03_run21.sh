# configs[3] on one GPU with query lists allowed for the 50 M-read sets (11 GB each)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp
COMMET_QUERY_LIST_GB=150 COMMET_QUERY_LIST_MAX_GB=16 python3 $R/bench.py --matrix-reads 50000000 --cpu-sample 0 --no-probe-count --steps 3 > $O/r03_c3_biglists.json 2> $O/r03_c3_biglists.err
tail -3 $O/r03_c3_biglists.err
python3 -c "
import json
d=json.load(open('$O/r03_c3_biglists.json')); m=d['detail']['matrix']; print({k:v for k,v in m.items() if k not in ('per_rank','workload','predicted_vs_actual_share')}); print(m['per_rank'])"
COMMET_QUERY_LIST_GB=150 COMMET_QUERY_LIST_MAX_GB=16 python3 $R/bench.py --reads 50000000 --no-matrix --cpu-sample 0 --no-probe-count --steps 3 > $O/r03_50m_biglist.json 2>/dev/null
python3 -c "
import json
d=json.load(open('$O/r03_50m_biglist.json')); print('50M pair', d['value'], d['ms_per_step'], d['detail']['first_job_ms'], d['detail']['query_list_bytes'], {k:round(v['ms_per_step'],2) for k,v in d['roofline']['kernels'].items() if v['ms_per_step']>1})"

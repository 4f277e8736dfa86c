# round 5, call 1: baseline of this round's box + the L2-fold timing bounds of the replay (COMMET_TQ_ABLATE 262144 / 524288)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_base
python3 bench.py --steps 20 --warmup 3 --no-matrix > gpurun_out/r05_base/bench.json 2> gpurun_out/r05_base/bench.err
python3 - <<'PY'
import json
b = json.load(open('gpurun_out/r05_base/bench.json'))
print('base', b['value'], b['ms_per_step'], {k: round(v['ms_per_step'], 2) for k, v in b['roofline']['kernels'].items() if v['ms_per_step'] > 0.05})
print({k: b['detail'].get(k) for k in ('index_kernel_ms', 'search_kernel_ms', 'first_job_ms')})
PY
COMMET_JOB_VERBOSE=1 python3 bench.py --no-matrix --cpu-sample 0 --no-probe-count --no-kernel-times --steps 5 > /dev/null 2> gpurun_out/r05_base/jobverbose.err || true
grep '\[job\]' gpurun_out/r05_base/jobverbose.err | tail -4
bash tools/ablate.sh COMMET_TQ_ABLATE 0 262144 524288 786432 > gpurun_out/r05_base/ablate.log 2>&1
cat gpurun_out/r05_base/ablate.log

# tq_probe_kernel: threads per workgroup x workgroups per XCD (configs[1]); prints the probe's and the step's ms.
# COMMET_TQ_PNT existed only in the experiment's build (kernel with blockDim.x strides and __launch_bounds__(1024), launch with
# dim3(c->tq_pnt)); the shipped kernel is fixed at 256 threads, COMMET_TQ_WPX still works.
R=$GRAFT_REPO_ROOT
cd /tmp
for wpx in 64 32 16; do for pnt in 256 512 1024; do
COMMET_TQ_WPX=$wpx COMMET_TQ_PNT=$pnt python3 $R/bench.py --no-matrix --cpu-sample 0 --no-probe-count --steps 10 --warmup 2 --kt-steps 3 > /tmp/o.json 2>/tmp/o.err || { tail -3 /tmp/o.err; exit 1; }
python3 -c "
import json
d=json.load(open('/tmp/o.json')); k=d['roofline']['kernels']; print('wpx', $wpx, 'threads', $pnt, 'probe', round(k['tq_probe_kernel']['ms_per_step'],3), 'replay', round(k['tq_replay_kernel']['ms_per_step'],3), 'step', d['ms_per_step'], 'shared', d['detail']['shared'])"
done; done

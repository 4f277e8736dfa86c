# round 5, call 8: the probe's address prefetch and its grid (workgroups per XCD), then the configs[2] leg five times in fresh processes
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_ab gpurun_out/r05_c2_variance
bash tools/ab_flags.sh "-DTQ_PROBE_PREFETCH=0" "-DTQ_PROBE_PREFETCH=1" 2>&1 | tee gpurun_out/r05_ab/ab8.log
bash tools/env_ab.sh COMMET_TQ_WPX 32 64 96 128 256 2>&1 | tee -a gpurun_out/r05_ab/ab8.log
bash tools/c2_variance.sh gpurun_out/r05_c2_variance 5 2>&1 | tee gpurun_out/r05_c2_variance/summary.log

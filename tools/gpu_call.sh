# One parameterised GPU-box script instead of a script per call (rounds 1-5 kept thirty of those; git history has them).
#   gpurun -- 'bash tools/gpu_call.sh NAME STEP [STEP ...]'      results under gpurun_out/NAME/
# STEPs, run in order, the call stops at the first one that fails:
#   tests[=PYTEST ARGS]     python -m pytest -m gpu -x -q [ARGS | tests]          -> tests.log
#   bench[=ARGS]            python bench.py ARGS                                   -> bench[_<n>].json / .err
#   profile[=ARGS]          tools/profile_bench.sh NAME/profile ARGS (rocprofv3 kernel trace + PMC passes -> traffic.json)
#   fuzz=N:SEED             tools/fuzz_gpu.py N scenarios from SEED                -> fuzz_SEED.log
#   env:K=V                 exported for the steps that follow
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=$1; shift
O=$R/gpurun_out/$N
mkdir -p $O
cd $R
nb=0
for step in "$@"; do
  kind=${step%%=*}; arg=""; [ "$kind" != "$step" ] && arg=${step#*=}
  case $kind in
    env:*) export "${step#env:}";;
    tests) timeout -k 10 1100 python3 -m pytest -m gpu -x -q --durations=10 ${arg:-tests} > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }; tail -3 $O/tests.log;;
    bench) nb=$((nb+1)); f=$O/bench$([ $nb -gt 1 ] && echo _$nb || true); python3 bench.py $arg > $f.json 2> $f.err || { tail -20 $f.err; exit 1; }; tail -2 $f.err;;
    profile) bash tools/profile_bench.sh $N/profile $arg > $O/profile.log 2>&1 || { tail -30 $O/profile.log; exit 1; }; (tail -1 $O/profile.log | cut -c1-300) || true;;
    fuzz) n=${arg%%:*}; s=${arg#*:}; timeout -k 10 1100 python3 tools/fuzz_gpu.py $s $n > $O/fuzz_$s.log 2>&1 || { tail -20 $O/fuzz_$s.log; exit 1; }; tail -2 $O/fuzz_$s.log;;
    *) echo "unknown step $step"; exit 2;;
  esac
done

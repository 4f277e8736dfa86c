# round 5, call 15: the multi-rank path rehearsed on ONE GPU (COMMET_FORCE_DEVICE=0; at most 6 processes may use the card on this
# pool: 5 ranks + the canary): 5 ranks end to end, 2 ranks on the default configs[3] matrix, and the failure path (4 ranks x 50 M reads
# do not fit one device: one rank's allocation fails -> top-level matrix error, exit codes)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_n_rehearsal
mkdir -p $O
export COMMET_FORCE_DEVICE=0
python3 bench.py --gpus 5 --reads 2000000 --matrix-reads 2000000 --steps 5 --warmup 1 > $O/bench_gpus5_2m_reads_on_one_gpu.json 2> $O/bench_gpus5_2m_reads_on_one_gpu.progress.log; echo "gpus5 rc=$?" | tee $O/exit_codes.log
python3 bench.py --gpus 2 > $O/bench_gpus2_default_configs3_on_one_gpu.json 2> $O/bench_gpus2_default_configs3_on_one_gpu.progress.log; echo "gpus2 rc=$?" | tee -a $O/exit_codes.log
python3 bench.py --gpus 4 > $O/bench_gpus4_default_out_of_memory.json 2> $O/bench_gpus4_default_out_of_memory.progress.log; echo "gpus4 (expected: one rank out of memory) rc=$?" | tee -a $O/exit_codes.log
python3 - <<'PY'
import json
O = "gpurun_out/r05_n_rehearsal/"
for f in ("bench_gpus5_2m_reads_on_one_gpu", "bench_gpus2_default_configs3_on_one_gpu", "bench_gpus4_default_out_of_memory"):
    try:
        b = json.load(open(O + f + ".json"))
    except Exception as ex:
        print(f, "NO LINE", ex); continue
    m = b.get("matrix") or {}
    print(f, "n_gpus", b["n_gpus"], "value", b["value"], "matrix", {k: m.get(k) for k in ("error", "total_s", "handover", "world")})
    d = (b["detail"].get("matrix") or {})
    for pr in d.get("per_rank", []):
        print("   rank", pr.get("rank"), {k: pr.get(k) for k in ("handover", "ipc_canary", "sets_parsed", "sets_loaded", "backend", "torch_loaded", "jobs", "set_wait_s")})
PY
ls /dev/shm | head

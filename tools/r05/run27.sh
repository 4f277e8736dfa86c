# round 5, call 27: the N > 1 forms of bench.py on the round's last sources, every rank on the one device (COMMET_FORCE_DEVICE=0):
# the driver's torchrun form at two ranks, the self-launched form at five
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_n_rehearsal2
mkdir -p $O
export COMMET_FORCE_DEVICE=0
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 5 --warmup 2 --reads 5000000 --matrix-reads 5000000 > $O/bench_torchrun2.json 2> $O/bench_torchrun2.progress.log; echo "torchrun2 rc=$?"
python3 bench.py --gpus 5 --reads 2000000 --matrix-reads 2000000 --steps 3 --warmup 1 > $O/bench_gpus5.json 2> $O/bench_gpus5.progress.log; echo "gpus5 rc=$?"
python3 - <<'PY'
import json
for f in ("bench_torchrun2", "bench_gpus5"):
    b = json.loads(open(f"gpurun_out/r05_n_rehearsal2/{f}.json").read().strip().splitlines()[-1]); m = b["detail"]["matrix"]
    print(f, b["n_gpus"], b["value"], b["scaling"], b.get("matrix"), m.get("total_s"), m.get("handover"), [(p.get("jobs"), p.get("backend"), p.get("torch_loaded")) for p in m["per_rank"]])
PY
ls /dev/shm | head

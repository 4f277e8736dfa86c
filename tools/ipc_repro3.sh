# NOT RUN YET in this form (round 3 ran out of GPU minutes).
# the N x N driver alone (no bench.py context before it) on 4 sets of 50 M reads, two ranks on one device, device-to-device hand-over
R=$GRAFT_REPO_ROOT
W=/dev/shm/commet_repro3
rm -rf $W; mkdir -p $W/out
cd $R
python3 - <<PY
import sys; sys.path.insert(0, "$R")
from commet_amd import synth
for s in range(4):          # (one after the other: a spawn pool cannot start from a script read from stdin)
    synth.write_set_fasta((s, 50_000_000, 100, "$W/set%d.fa" % s))
open("$W/sets.txt", "w").write("".join("S%d: $W/set%d.fa\n" % (s, s) for s in range(4)))
print("files written", flush=True)
PY
COMMET_MATRIX_IPC=${IPC:-1} COMMET_FORCE_DEVICE=0 COMMET_DIST_TIMEOUT_S=60 timeout -k 10 120 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 -m commet_amd.matrix $W/sets.txt -k 32 -t 2 -o $W/out/ 2>&1 | grep -v "amdgpu.ids\|hostname of the client" | tail -12
echo "rc=${PIPESTATUS[0]}"
rm -rf $W

"""scatter1's time job after job in ONE fresh context (its workspaces allocated by the first job): does it settle, and how fast?
  python tools/s1_drift.py [jobs]"""
import os, sys
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/commet_amd") else os.environ.get("GRAFT_REPO_ROOT", "."))
import commet_amd
from commet_amd import synth
jobs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n, L = 10_000_000, 100
b0, o0 = synth.synth_set(0, n, L)
b1, o1 = synth.synth_set(1, n, L)
for ctxno in range(2):
    with commet_amd.Context(k=32, t=2) as ctx:
        irs = commet_amd.ReadSet.from_files(ctx, [(b0, o0)])
        qrs = commet_amd.ReadSet.from_files(ctx, [(b1, o1)])
        ctx.set_option("kernel_timing", 1)
        row = []
        for j in range(jobs):
            ctx.index_and_search(irs, [qrs])
            kt = ctx.kernel_times()
            row.append((round(kt["part_scatter1_kernel"][1], 2), round(kt["part_scatter2_packed_kernel"][1], 2), round(sum(v[1] for v in kt.values()), 1)))
        print("context", ctxno, "scatter1 / scatter2 / all kernels, ms per job:", flush=True)
        print("  " + " ".join(f"{a}/{b}/{c}" for a, b, c in row), flush=True)

import os, sys
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/commet_amd") else os.environ.get("GRAFT_REPO_ROOT", "."))
import commet_amd
from commet_amd import synth
n, L = 10_000_000, 100
b0, o0 = synth.synth_set(0, n, L)
b1, o1 = synth.synth_set(1, n, L)
with commet_amd.Context(k=32, t=2) as ctx:
    qrs = commet_amd.ReadSet.from_files(ctx, [(b1, o1)])
    for rep in range(6):
        irs = commet_amd.ReadSet.from_files(ctx, [(b0, o0)])
        ctx.index_and_search(irs, [qrs])
        ctx.set_option("kernel_timing", 1)
        for _ in range(3):
            ctx.index_and_search(irs, [qrs])
        kt = ctx.kernel_times()
        ctx.set_option("kernel_timing", 0)
        print("same ctx, new index set", rep, {k: round(v[1] / 3, 2) for k, v in kt.items() if k.startswith("part_s") or k.startswith("part_h")}, flush=True)
        irs.close()

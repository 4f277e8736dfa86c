"""Cold-set ingest rates on the GPU box: python tools/ingest_bench.py [reads] -> one JSON line per case
(arrays = commet_readset_append from numpy memory, fasta = commet_readset_from_fasta on a file in /dev/shm)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import commet_amd  # noqa: E402
from commet_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
L = 100
b0, o0 = synth.synth_set(0, n, L)
b1, o1 = synth.synth_set(1, n, L)
path = ("/dev/shm" if os.access("/dev/shm", os.W_OK) else "/tmp") + f"/commet_ingest_bench_{os.getpid()}.fa"
synth.write_fasta_fast(path, b1, n, L)
ctx = commet_amd.Context(k=32, t=2)
for name, fn in (("arrays, cold pool", lambda: commet_amd.ReadSet.from_files(ctx, [(b0, o0)])),
                 ("arrays, warm pool", lambda: commet_amd.ReadSet.from_files(ctx, [(b1, o1)])),
                 ("arrays, warm pool again", lambda: commet_amd.ReadSet.from_files(ctx, [(b0, o0)])),
                 ("fasta file", lambda: commet_amd.ReadSet.from_fasta(ctx, [path])),
                 ("fasta file again", lambda: commet_amd.ReadSet.from_fasta(ctx, [path]))):
    t0 = time.perf_counter()
    rs = fn()
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"case": name, "reads": n, "seconds": round(dt, 4), "M_reads_per_s": round(n / dt / 1e6, 1),
                      "GB_per_s_of_bases": round(n * L / dt / 1e9, 2)}), flush=True)
    rs.close()
os.remove(path)
ctx.close()

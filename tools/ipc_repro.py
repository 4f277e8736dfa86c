"""Two processes on one device that both export K resident sets and then import one of the other's (the N x N driver's
device-to-device hand-over, stripped down): where does commet_readset_import stop returning?
  python tools/ipc_repro.py [reads] [K] [mode]      mode: both (default) | one (only process 0 imports) | probe+both (a tiny set exported / imported / freed first)"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys, time
sys.path.insert(0, {root!r})
if "preload" in sys.argv[4]:    # the fix: the library (and with it the system's ROCm runtime) before torch
    from commet_amd import lib as _lib
    _lib.load()
if "torch" in sys.argv[4]:      # the driver's ranks used to import torch (gloo) before the library was loaded
    import torch, torch.distributed
import commet_amd
from commet_amd import synth
me, n, K, mode, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
b, o = synth.synth_set(me, n, 100)
import numpy as np
with commet_amd.Context(k=32, t=2) as ctx:
    if "probe" in mode:      # what the driver does first: a tiny set exported, imported by the other, then FREED by its owner
        pb = np.frombuffer(b"ACGTTGCAACGTACGTTTGACCAGTACGATCGATCGGCTA" * 4, dtype=np.uint8)
        po = np.arange(5, dtype=np.uint64) * np.uint64(40)
        probe = commet_amd.ReadSet.from_files(ctx, [(pb, po)])
        open(os.path.join(d, f"p{{me}}_probe.tmp"), "wb").write(probe.export())
        os.rename(os.path.join(d, f"p{{me}}_probe.tmp"), os.path.join(d, f"p{{me}}_probe.blob"))
        while not os.path.exists(os.path.join(d, f"p{{1 - me}}_probe.blob")):
            time.sleep(0.01)
        g = commet_amd.ReadSet.import_(ctx, open(os.path.join(d, f"p{{1 - me}}_probe.blob"), "rb").read())
        g.close()
        open(os.path.join(d, f"p{{me}}_probed"), "w").write("x")
        while not os.path.exists(os.path.join(d, f"p{{1 - me}}_probed")):
            time.sleep(0.01)
        probe.close()
        print(json.dumps(dict(proc=me, probe="exported, imported by the other, freed")), flush=True)
    if "fork" in mode:      # child processes started (fork + exec) beside the parsing, as the driver starts filter_reads
        import subprocess, threading
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(3)
        forks = [pool.submit(subprocess.run, ["/bin/sh", "-c", "cat " + os.path.join(d, f"p{{me}}.fa") + " > /dev/null 2>&1; sleep 0.3"]) for _ in range(6)]
    if "fasta" in mode:     # parsed from a FASTA file by the host ingest threads, as the driver's sets are
        fa = os.path.join(d, f"p{{me}}.fa")
        synth.write_fasta_fast(fa, b, n, 100)
        if "thread" in mode:  # ... on the loader thread
            import threading
            sets = []
            th = threading.Thread(target=lambda: sets.extend(commet_amd.ReadSet.from_fasta(ctx, [fa]) for _ in range(K)))
            th.start()
            th.join()
        else:
            sets = [commet_amd.ReadSet.from_fasta(ctx, [fa]) for _ in range(K)]
    else:
        sets = [commet_amd.ReadSet.from_files(ctx, [(b, o)]) for _ in range(K)]
    if "fork" in mode:
        for f in forks:
            f.result()
        print(json.dumps(dict(proc=me, children="6 started beside the parsing, all done")), flush=True)
    for i, rs in enumerate(sets):
        open(os.path.join(d, f"p{{me}}_s{{i}}.tmp"), "wb").write(rs.export())
        os.rename(os.path.join(d, f"p{{me}}_s{{i}}.tmp"), os.path.join(d, f"p{{me}}_s{{i}}.blob"))
    print(json.dumps(dict(proc=me, exported=K, reads=n)), flush=True)
    other = 1 - me
    while not os.path.exists(os.path.join(d, f"p{{other}}_s{{K - 1}}.blob")):
        time.sleep(0.01)
    def imports():
        for i in range(min(K, 2)):
            t0 = time.perf_counter()
            r = commet_amd.ReadSet.import_(ctx, open(os.path.join(d, f"p{{other}}_s{{i}}.blob"), "rb").read())
            print(json.dumps(dict(proc=me, imported=i, seconds=round(time.perf_counter() - t0, 4), reads=r.num_reads)), flush=True)
            r.close()
    if "both" in mode or me == 0:
        if "thread" in mode:     # from a second host thread, as the driver's loader does, the main thread waiting
            import threading
            th = threading.Thread(target=imports)
            th.start()
            th.join()
        else:
            imports()
    open(os.path.join(d, f"p{{me}}.done"), "w").write("x")
    while not os.path.exists(os.path.join(d, f"p{{other}}.done")):      # the exported sets stay alive until the other is done
        time.sleep(0.01)
'''


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    mode = sys.argv[3] if len(sys.argv) > 3 else "both"
    d = tempfile.mkdtemp(prefix="commet_ipc_", dir="/dev/shm" if os.access("/dev/shm", os.W_OK) else None)
    code = CHILD.format(root=ROOT)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(p), str(n), str(K), mode, d]) for p in (0, 1)]
    t0 = time.time()
    rc = []
    for p in procs:
        try:
            rc.append(p.wait(timeout=max(1, 150 - (time.time() - t0))))
        except subprocess.TimeoutExpired:
            rc.append("hung")
    for p in procs:
        if p.poll() is None:
            p.kill()
    print(json.dumps(dict(reads=n, K=K, mode=mode, result=rc)), flush=True)
    for f in os.listdir(d):
        os.remove(os.path.join(d, f))
    os.rmdir(d)
    return 0 if all(r == 0 for r in rc) else 1


if __name__ == "__main__":
    sys.exit(main())

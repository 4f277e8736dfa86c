"""Write / read rates of packed read-set images (commet_readset_save / _load) through /dev/shm: what the ranks of a node
hand each other in the N x N driver.  python tools/image_bench.py [reads] [read_len]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import commet_amd  # noqa: E402
from commet_amd import synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    b, o = synth.synth_set(0, n, L)
    root = "/dev/shm" if os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    d = tempfile.mkdtemp(prefix="commet_img_", dir=root)
    path = os.path.join(d, "set.pk")
    try:
        with commet_amd.Context(k=32, t=2) as ctx:
            rs = commet_amd.ReadSet.from_files(ctx, [(b, o)])
            for rep in range(3):
                t0 = time.perf_counter()
                rs.save(path)
                ts = time.perf_counter() - t0
                size = os.path.getsize(path)
                t0 = time.perf_counter()
                r2 = commet_amd.ReadSet.load(ctx, path)
                tl = time.perf_counter() - t0
                same = r2.num_reads == rs.num_reads
                r2.close()
                print(json.dumps(dict(reads=n, image_GB=round(size / 1e9, 3), save_s=round(ts, 4), save_GBps=round(size / ts / 1e9, 1),
                                      load_s=round(tl, 4), load_GBps=round(size / tl / 1e9, 1), ok=same)), flush=True)
            # the file-less hand-over: this process exports, a child process (same device here) imports three times
            t0 = time.perf_counter()
            blob = rs.export()
            te = time.perf_counter() - t0
            bp = os.path.join(d, "set.blob")
            open(bp, "wb").write(blob)
            child = (f"import sys, time, json; sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r}); import commet_amd\n"
                     f"blob = open({bp!r}, 'rb').read()\n"
                     "with commet_amd.Context(k=32, t=2) as ctx:\n"
                     "    for rep in range(3):\n"
                     "        t0 = time.perf_counter(); r = commet_amd.ReadSet.import_(ctx, blob); dt = time.perf_counter() - t0\n"
                     f"        print(json.dumps(dict(import_s=round(dt, 4), import_GBps=round({size} / dt / 1e9, 1), reads=r.num_reads)), flush=True)\n"
                     "        r.close()\n")
            import subprocess
            print(json.dumps(dict(export_s=round(te, 5), blob_bytes=len(blob))), flush=True)
            subprocess.run([sys.executable, "-c", child], check=True)
    finally:
        for f in os.listdir(d):
            os.remove(os.path.join(d, f))
        os.rmdir(d)


if __name__ == "__main__":
    main()

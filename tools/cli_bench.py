"""End-to-end time of the drop-in CLI on BASELINE configs[1]-shaped FASTA files (what one Commet.py job costs):
  python tools/cli_bench.py [reads] [k]"""
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from commet_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    k = sys.argv[2] if len(sys.argv) > 2 else "32"
    work = tempfile.mkdtemp(prefix="commet_cli_")
    for s in (0, 1):
        b, _ = synth.synth_set(s, n, 100)
        synth.write_fasta_fast(os.path.join(work, f"s{s}.fa"), b, n, 100)
        with open(os.path.join(work, f"c{s}.txt"), "w") as fh:
            fh.write(f"set{s}: {work}/s{s}.fa\n")
    exe = os.path.join(HERE, "commet_amd", "bin", "index_and_search")
    env = dict(os.environ, COMMET_INGEST_VERBOSE="1")
    server = None
    if os.environ.get("CLI_SERVER"):          # the same job through a resident server (index_and_search --serve)
        sock = os.path.join(work, "s.sock")
        server = subprocess.Popen([exe, "--serve", sock], stderr=subprocess.DEVNULL)
        while not os.path.exists(sock):
            time.sleep(0.05)
        env["COMMET_SERVER"] = sock
    for rep in range(int(os.environ.get("CLI_REPS", "3"))):
        t0 = time.perf_counter()
        r = subprocess.run([exe, "-i", f"{work}/c0.txt", "-s", f"{work}/c1.txt", "-o", f"{work}/out", "-l", f"{work}/out", "-k", k, "-t", "2"],
                           capture_output=True, text=True, env=env)
        dt = time.perf_counter() - t0
        print(f"run {rep}: wall {dt:.3f} s rc={r.returncode}")
        for ln in (r.stdout + r.stderr).splitlines():
            if any(w in ln for w in ("time", "ingest", "indexed", "ms", " s")):
                print("   ", ln)
    if server:
        subprocess.run([exe, "--server-stats"], env=env)
        subprocess.run([exe, "--server-stop"], env=env)
        server.wait()
    subprocess.run(["rm", "-rf", work])


if __name__ == "__main__":
    main()

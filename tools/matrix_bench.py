"""N x N matrix on synthetic sets through the resident driver (BASELINE configs[2]/[3] shape):
  python tools/matrix_bench.py [n_sets] [reads_per_set] [k]
Under torch.distributed.run every rank takes its share of the pairs."""
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from commet_amd import matrix, sharding, synth  # noqa: E402


def main():
    n_sets = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    L = 100
    ranks = sharding.Ranks()
    work = os.environ.get("COMMET_BENCH_DIR") or os.path.join(tempfile.gettempdir(), f"commet_matrix_{n_sets}_{n}")
    if ranks.rank == 0:
        os.makedirs(work, exist_ok=True)
        t0 = time.time()
        with open(os.path.join(work, "sets.txt"), "w") as fh:
            for s in range(n_sets):
                p = os.path.join(work, f"set{s}.fa")
                if not os.path.exists(p):
                    b, _ = synth.synth_set(s, n, L)
                    synth.write_fasta_fast(p, b, n, L)
                fh.write(f"S{s}: {p}\n")
        print(f"generated {n_sets} x {n} reads in {time.time() - t0:.1f} s", file=sys.stderr)
    ranks.barrier()
    res = matrix.run(os.path.join(work, "sets.txt"), os.path.join(work, f"out_r{ranks.world}") + "/", k=k, t=2,
                     ranks=ranks, verbose=False)
    if ranks.rank == 0:
        res.pop("matrix")
        print(json.dumps(res))
        if not os.environ.get("COMMET_BENCH_KEEP"):
            shutil.rmtree(work, ignore_errors=True)
    ranks.close()


if __name__ == "__main__":
    main()

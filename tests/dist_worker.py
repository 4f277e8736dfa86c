"""Worker of tests/test_distributed_cpu.py: the ranks of a job (started by torch.distributed.run or by sharding.spawn_ranks), no GPU.
argv: out_dir n_sets [backend: tcp (the default) | gloo]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from commet_amd import sharding  # noqa: E402


def main():
    out_dir = sys.argv[1]
    n_sets = int(sys.argv[2])
    ranks = sharding.Ranks(backend=sys.argv[3] if len(sys.argv) > 3 else None)
    chains = sharding.pair_chains(n_sets)
    cost = [1.0 + (c % 3) for c in range(len(chains))]
    mine = sharding.assign_chains(chains, ranks.world, ranks.rank, cost)
    done = []

    def step():
        for c in mine:
            for job in chains[c]:
                done.append((c, job[0]))
        time.sleep(0.05 * (ranks.rank + 1))          # uneven ranks: MAX must pick the slowest

    elapsed = sharding.timed_region(ranks, lambda: None, step, 2)
    total_jobs = ranks.sum_int(len(done))
    everyone = ranks.gather_objects(mine)
    first = ranks.broadcast_object(f"from rank {ranks.rank}", src=1 % ranks.world)
    for _ in range(5):                                # (rounds of the store are dropped two behind)
        ranks.barrier()
    with open(os.path.join(out_dir, f"rank{ranks.rank}.json"), "w") as fh:
        json.dump(dict(rank=ranks.rank, world=ranks.world, mine=mine, elapsed=elapsed, total_jobs=total_jobs,
                       everyone=everyone, first=first, backend=ranks.backend, torch_loaded="torch" in sys.modules), fh)
    ranks.close()


if __name__ == "__main__":
    main()

"""Host-side sharding of the N x N comparison over ranks (one process per GPU).

The path has no exchange step: the unit of work is one
index_and_search invocation, and Commet.py's job DAG (Commet.py:186-240, 570-574) is
    J1(ref)          index S_ref, search S_{ref+1..N-1}
    J2(ref, i)       index S_i restricted to <F>_in_<S_ref>.bv, search S_ref      (needs J1(ref))
    J3(ref, i)       index S_ref restricted to <G>_in_<S_i>.bv, search S_i        (needs J2(ref, i); overwrites J1's output)
for ref < i.  J1(ref) is split per search set so that every (ref, i) pair is one
independent chain J1(ref,i) -> J2(ref,i) -> J3(ref,i): N(N-1)/2 chains, no
ordering between chains, only `.bv` bit-vectors flow inside a chain.  Chains
are dealt to ranks; the only cross-rank operations are a barrier and a MAX
reduction of the elapsed time (gloo on the host: no data-path collective)."""
import time


def commet_jobs(n_sets):
    """The N^2-1 invocations of Commet.py in its own order: (kind, index set, [search sets], restrict_to or None)."""
    jobs = []
    for ref in range(n_sets - 1):
        jobs.append(("J1", ref, list(range(ref + 1, n_sets)), None))
        for i in range(ref + 1, n_sets):
            jobs.append(("J2", i, [ref], ref))
            jobs.append(("J3", ref, [i], i))
    return jobs


def pair_chains(n_sets):
    """One chain per unordered pair (ref, i), ref < i: [J1(ref,i), J2(ref,i), J3(ref,i)]."""
    chains = []
    for ref in range(n_sets - 1):
        for i in range(ref + 1, n_sets):
            chains.append([("J1", ref, [i], None), ("J2", i, [ref], ref), ("J3", ref, [i], i)])
    return chains


def assign_chains(chains, world_size, rank, cost=None):
    """Longest-processing-time-first dealing of chains to ranks; deterministic on every rank."""
    if cost is None:
        cost = [1.0] * len(chains)
    order = sorted(range(len(chains)), key=lambda c: (-cost[c], c))
    load = [0.0] * world_size
    mine = []
    for c in order:
        r = min(range(world_size), key=lambda x: (load[x], x))
        load[r] += cost[c]
        if r == rank:
            mine.append(c)
    return sorted(mine)


def assign_pairs_contiguous(cost, world_size):
    """Cuts the row-major list of (ref, i) pairs into world_size CONTIGUOUS runs of about equal cost; returns
    [range(lo, hi)] per rank.  Contiguous runs keep a rank's pairs on few reference sets, so that J1's index of
    S_ref is built once per (rank, ref) and a rank touches few sets; boundaries fall where the running cost crosses
    r / world_size of the total (each run is within one pair's cost of the ideal share)."""
    n, total = len(cost), float(sum(cost))
    cuts, acc, c = [0], 0.0, 0
    for r in range(1, world_size):
        target = total * r / world_size
        while c < n and acc + cost[c] / 2.0 <= target:
            acc += cost[c]
            c += 1
        cuts.append(c)
    cuts.append(n)
    return [range(cuts[r], cuts[r + 1]) for r in range(world_size)]


class Ranks:
    """Barrier / MAX-of-elapsed over ranks.  world_size 1 needs no torch at all."""

    def __init__(self, backend="gloo"):
        import os
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                import datetime
                # a rank that dies must not leave the others waiting for long (default 30 min)
                tmo = datetime.timedelta(seconds=float(os.environ.get("COMMET_DIST_TIMEOUT_S", "600")))
                # gloo announces its connections on STDOUT ("[Gloo] Rank 0 is connected to ..."): keep them out of a
                # caller's result stream (bench.py prints one JSON line there) by lending fd 1 to stderr meanwhile
                import sys
                sys.stdout.flush()
                saved = os.dup(1)
                try:
                    os.dup2(2, 1)
                    dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world, timeout=tmo)
                    dist.barrier()          # the connections are made (and announced) at the first collective
                finally:
                    sys.stdout.flush()
                    os.dup2(saved, 1)
                    os.close(saved)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_seconds(self, seconds):
        if self.dist is None:
            return float(seconds)
        import torch
        t = torch.tensor([float(seconds)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def sum_int(self, v):
        if self.dist is None:
            return int(v)
        import torch
        t = torch.tensor([int(v)], dtype=torch.int64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t[0])

    def broadcast_object(self, obj, src=0):
        if self.dist is None:
            return obj
        box = [obj if self.rank == src else None]
        self.dist.broadcast_object_list(box, src=src)
        return box[0]

    def gather_objects(self, obj):
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None


def timed_region(ranks, sync, fn, steps):
    """barrier + device sync, `steps` calls of fn, device sync + barrier; returns MAX elapsed over ranks."""
    sync()
    ranks.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    ranks.barrier()
    return ranks.max_seconds(time.perf_counter() - t0)

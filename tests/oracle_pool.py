"""Runs the CPU checker's batch kernels (oracle_binding.Bloom) in worker PROCESSES, one index chunk each.
TEST INFRASTRUCTURE ONLY.  A chunk's filter depends on that chunk's reads alone and a read's final tag is the OR of
its per-chunk results (index_and_search.cpp:255-277: a read found in chunk c is merely skipped afterwards), so the
chunks of a large index set can be replayed side by side; each worker builds one 2^(k-1)-byte filter.

Workers are spawned (not forked: the parent holds a HIP context) and never touch the GPU; arrays travel as .npy
files in a scratch directory."""
import multiprocessing as mp
import os

import numpy as np


def chunks_from_counts(kc, max_kmer):
    """Chunk read ranges [a, e) of a sequence of reads with kc[i] complete k-mers each, the reference's way
    (index_reads.h:49,60): a chunk is full once its k-mers reach max_kmer; the look-ahead read e is dropped."""
    out, n, i = [], len(kc), 0
    pre = np.concatenate([[0], np.cumsum(kc, dtype=np.int64)])
    while i < n:
        e = int(np.searchsorted(pre, pre[i] + max_kmer, side="left"))     # first e with pre[e] - pre[i] >= max
        e = min(max(e, i + 1), n)
        out.append((i, e))
        i = e + 1
    return out


def _one_chunk(args):
    bases_npy, a, e, L, k, t, sample_npy, Ls = args
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import oracle_binding as ob
    bases = np.load(bases_npy, mmap_mode="r")
    sb = np.load(sample_npy)
    f = ob.Bloom(k)
    chunk = np.ascontiguousarray(bases[a * L: e * L])
    fed = f.index(chunk, np.arange(e - a + 1, dtype=np.uint64) * np.uint64(L))
    ns = sb.size // Ls
    found, _ = f.search(t, sb, np.arange(ns + 1, dtype=np.uint64) * np.uint64(Ls))
    f.close()
    return fed, found.tobytes()


def search_sample_over_chunks(scratch, tag, index_bases, L, chunks, k, t, sample_bases, workers=None, first_chunk=False,
                              sample_len=None):
    """index_bases: uint8[n * L] of the reads that are fed (already restricted to the selected reads, in order);
    chunks: [(a, e)] over those reads; sample_bases: uint8[m * sample_len] query reads (sample_len defaults to L).
    Returns (bool[m] found in any chunk, [k-mers fed per chunk]); with first_chunk also int[m], the first chunk that
    finds each read (-1: none) — what the reference's chunk loop would tag it in."""
    os.makedirs(scratch, exist_ok=True)
    bnpy, snpy = os.path.join(scratch, tag + "_index.npy"), os.path.join(scratch, tag + "_sample.npy")
    np.save(bnpy, np.asarray(index_bases, dtype=np.uint8))
    np.save(snpy, np.asarray(sample_bases, dtype=np.uint8))
    Ls = sample_len or L
    m = len(sample_bases) // Ls
    jobs = [(bnpy, a, e, L, k, t, snpy, Ls) for a, e in chunks]
    workers = workers or max(1, min(len(jobs), (os.cpu_count() or 2) - 1, 12))
    if len(jobs) == 1 or workers == 1:
        res = [_one_chunk(j) for j in jobs]
    else:
        with mp.get_context("spawn").Pool(workers) as pool:
            res = pool.map(_one_chunk, jobs, chunksize=1)
    found = np.zeros(m, dtype=bool)
    first = np.full(m, -1, dtype=np.int64)
    for ci, (_, fb) in enumerate(res):
        fc = np.unpackbits(np.frombuffer(fb, dtype=np.uint8), bitorder="little")[:m].astype(bool)
        first[fc & ~found] = ci
        found |= fc
    os.remove(bnpy)
    os.remove(snpy)
    if first_chunk:
        return found, [fed for fed, _ in res], first
    return found, [fed for fed, _ in res]


def chunk_loop_in_threads(k, t, ib, io, qb, qo, chunks, n_q, workers=None):
    """The reference's chunk loop (index_and_search.cpp:255-277) over `chunks` of the index batch (ib, io) for the query batch
    (qb, qo), on the CPU checker, the chunks dealt to threads in contiguous runs (the checker is called through ctypes, which
    releases the interpreter lock).  A read's tag is the OR of its per-chunk results and a read found by a chunk is merely skipped
    afterwards, so every run may start from empty tags.  Returns (found bits uint8[n_q // 8 + 1], reads the LAST chunk searched =
    the log line's "searched").  k >= 33: 4-8 GiB per filter, so four threads at most."""
    from concurrent.futures import ThreadPoolExecutor
    import oracle_binding as ob
    chunks = list(chunks)
    if workers is None:
        workers = 4 if k >= 33 else 8
    workers = max(1, min(workers, len(chunks)))
    per = -(-len(chunks) // workers)
    runs = [chunks[i: i + per] for i in range(0, len(chunks), per)]

    def run(my, is_last):
        found = np.zeros(n_q // 8 + 1, dtype=np.uint8)
        before_last = None
        for ci, (a, e) in enumerate(my):
            if is_last and ci == len(my) - 1:
                before_last = found.copy()
            f = ob.Bloom(k)
            f.index(ib[int(io[a]): int(io[e])], io[a: e + 1] - io[a])
            fnd, _ = f.search(t, qb, qo, ~found)
            found |= fnd
            f.close()
        return found, before_last

    with ThreadPoolExecutor(len(runs)) as pool:
        res = list(pool.map(lambda a: run(*a), [(r, i == len(runs) - 1) for i, r in enumerate(runs)]))
    found = np.zeros(n_q // 8 + 1, dtype=np.uint8)
    seen_before_last = res[-1][1].copy()
    for i, (f, _) in enumerate(res):
        found |= f
        if i < len(res) - 1:
            seen_before_last |= f
    pad = np.unpackbits(seen_before_last, bitorder="little")[:n_q]
    return found, int(n_q - pad.sum())

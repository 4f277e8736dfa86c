"""Derived data cached with resident read sets (the tiled search's query lists): accounted, held to a budget, given back
least-recently-used first, rebuilt on demand — and never a different result bit.  Packed images are byte-reproducible."""
import os

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def _sets(rng, n_sets, n, L):
    base = util.random_reads(rng, n, L, L, n_rate=0.002)
    out = [base]
    for _ in range(n_sets - 1):
        out.append(util.related_reads(rng, base, n, L, L, share=0.4, n_rate=0.002))
    return [util.to_batch(r) for r in out]


def test_query_lists_are_budgeted_and_rebuilt():
    import commet_amd
    rng = np.random.default_rng(3)
    k, t, n, L = 26, 2, 6000, 100
    batches = _sets(rng, 4, n, L)
    with commet_amd.Context(k=k, t=t) as ctx:
        ctx.set_option("tiled_search", 2)
        ctx.set_option("max_kmer", 150000)                                   # 3 chunks per index set ...
        ctx.set_option("chunk_group", 2)                                      # ... searched as a group of two and a single one
        rs = [commet_amd.ReadSet.from_files(ctx, [b]) for b in batches]
        assert all(r.cache_bytes == 0 for r in rs)
        ref = {}
        for j in (1, 2, 3):
            ref[j] = ctx.index_and_search(rs[0], [rs[j]])
            assert rs[j].cache_bytes > 0                                     # the set's list was made by its first scan ...
        one = rs[1].cache_bytes
        st = ctx.cache_stats()
        assert st["bytes"] == sum(r.cache_bytes for r in rs) and st["evictions"] == 0 and rs[0].cache_bytes == 0
        assert 5 * n * 30 < one < 8 * n * 80                                 # ~6 bytes per first-hit window (+ tile bounds)
        # a budget of two lists: the least recently used one (set 1's) goes
        ctx.set_option("query_list_budget_mb", 0)
        assert ctx.cache_stats()["bytes"] == 0 and ctx.cache_stats()["evictions"] == 3
        ctx.set_option("query_list_budget_mb", 1 + 2 * one // (1 << 20))
        for j in (1, 2, 3, 1):
            tags, stats, _ = ctx.index_and_search(rs[0], [rs[j]])
            assert np.array_equal(tags[0], ref[j][0][0]) and stats[0]["shared"] == ref[j][1][0]["shared"]
            assert ctx.cache_stats()["bytes"] <= ctx.cache_stats()["budget_bytes"]
        assert rs[1].cache_bytes > 0 and rs[3].cache_bytes > 0 and rs[2].cache_bytes == 0   # 2 was the least recently used
        assert ctx.cache_stats()["evictions"] >= 5
        # explicit release, and a job whose search sets' lists exceed the budget together: the running job's lists stay
        rs[1].drop_cache()
        assert rs[1].cache_bytes == 0
        ctx.set_option("query_list_budget_mb", 1)
        tags, stats, _ = ctx.index_and_search(rs[0], [rs[1], rs[2], rs[3]])
        for j in (1, 2, 3):
            assert np.array_equal(tags[j - 1], ref[j][0][0])
        ctx.set_option("tiled_search", 1)                                     # the gather kernels agree
        tags, _, _ = ctx.index_and_search(rs[0], [rs[1]])
        assert np.array_equal(tags[0], ref[1][0][0])


def test_packed_images_are_byte_reproducible(tmp_path):
    """the gap triples between reads are zeroed at creation: two images of the same reads are the same bytes, whatever
    the HBM held before (ragged reads: the host packer leaves gaps at staging-buffer boundaries)"""
    import commet_amd
    rng = np.random.default_rng(9)
    reads = util.random_reads(rng, 30000, 20, 300, n_rate=0.01)
    b, o = util.to_batch(reads)
    with commet_amd.Context(k=24, t=2) as ctx:
        junk = commet_amd.ReadSet.from_files(ctx, [util.to_batch(util.random_reads(rng, 40000, 250, 300))])
        junk.close()                                                          # leaves its bits in freed HBM
        a = commet_amd.ReadSet.from_files(ctx, [(b, o)])
        a.save(str(tmp_path / "a.pk"))
        c = commet_amd.ReadSet.from_files(ctx, [(b, o)])
        c.save(str(tmp_path / "c.pk"))
        d = commet_amd.ReadSet.load(ctx, str(tmp_path / "a.pk"))
        d.save(str(tmp_path / "d.pk"))
    A, Cc, D = (open(tmp_path / f, "rb").read() for f in ("a.pk", "c.pk", "d.pk"))
    assert A == Cc == D


def test_device_memory_is_kept_for_reuse_and_given_back():
    """blocks of 8 MiB or more that a context gives up are filed by the library and handed to the next allocation they fit (a
    hipMalloc of GBs costs 15-30 ms per GiB on this driver and now and then stalls behind a large hipFree): a second context gets
    the first one's filter and workspaces, results unchanged although the memory comes with the old bits in it; trimming gives it
    all back to the driver"""
    import commet_amd
    rng = np.random.default_rng(17)
    k, t, n, L = 28, 2, 150000, 100
    b0, b1 = _sets(rng, 2, n, L)
    commet_amd.device_cache_trim()
    assert commet_amd.device_cache_bytes(0) == 0
    runs = []
    for rep in range(3):
        before = commet_amd.device_cache_bytes(0)
        with commet_amd.Context(k=k, t=t) as ctx:
            during = commet_amd.device_cache_bytes(0)
            a, b = commet_amd.ReadSet.from_files(ctx, [b0]), commet_amd.ReadSet.from_files(ctx, [b1])
            ctx.set_option("index_mode", 2)                                   # the bucketed build: workspaces of tens of MB
            tags, stats, info = ctx.index_and_search(a, [b])
            runs.append((tags[0].copy(), stats[0]["shared"], info["n_chunks"]))
        after = commet_amd.device_cache_bytes(0)
        assert after >= (1 << (k - 1))                                        # the filter (2^(k-1) bytes) at least is filed
        if rep:
            assert during < before                                            # the new context took filed blocks (its filter first)
            assert after <= before + (64 << 20)                               # ... and the same blocks came back: nothing new was needed
    for r in runs[1:]:
        assert np.array_equal(r[0], runs[0][0]) and r[1:] == runs[0][1:]
    assert runs[0][1] > n // 4
    freed = commet_amd.device_cache_trim()
    assert freed >= (1 << (k - 1)) and commet_amd.device_cache_bytes(0) == 0
    with commet_amd.Context(k=k, t=t) as ctx:                                 # and a context on fresh memory agrees
        a, b = commet_amd.ReadSet.from_files(ctx, [b0]), commet_amd.ReadSet.from_files(ctx, [b1])
        tags, stats, _ = ctx.index_and_search(a, [b])
        assert np.array_equal(tags[0], runs[0][0]) and stats[0]["shared"] == runs[0][1]

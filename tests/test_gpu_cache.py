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


def test_pooled_blocks_and_the_export_of_a_set_that_lies_in_them(tmp_path):
    """COMMET_DEVMEM_POOL=1: new blocks above a threshold (256 MiB; 8 MiB here, COMMET_DEVMEM_POOL_MIN_MB) come from hipMallocAsync — no IPC handle exists
    for them: a set whose planes lie in such a block gives the same job results before and after commet_readset_export moved them
    into a hipMalloc block, a second process imports it and agrees with its own parse, pooled blocks are filed and reused like
    the others and trimming gives them back.  (A process of its own: the threshold is read once.)"""
    import subprocess
    import sys
    rng = np.random.default_rng(23)
    b0, b1 = _sets(rng, 2, 260000, 100)                                       # planes of ~10 MB
    np.save(tmp_path / "b0.npy", b0[0]), np.save(tmp_path / "o0.npy", b0[1])
    np.save(tmp_path / "b1.npy", b1[0]), np.save(tmp_path / "o1.npy", b1[1])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = f'''
import sys, subprocess, numpy as np
sys.path.insert(0, {root!r})
import commet_amd
d = {str(tmp_path)!r}
s0 = (np.load(d + "/b0.npy"), np.load(d + "/o0.npy"))
s1 = (np.load(d + "/b1.npy"), np.load(d + "/o1.npy"))
'''
    child = common + '''
with commet_amd.Context(k=25, t=2) as ctx:
    got = commet_amd.ReadSet.import_(ctx, open(d + "/set.blob", "rb").read())
    own, q = commet_amd.ReadSet.from_files(ctx, [s0]), commet_amd.ReadSet.from_files(ctx, [s1])
    r1, r2 = ctx.index_and_search(got, [q]), ctx.index_and_search(own, [q])
    assert np.array_equal(r1[0][0], r2[0][0]) and r1[1][0]["shared"] == r2[1][0]["shared"] > 1000
    np.save(d + "/child_tags.npy", r1[0][0])
print("imported ok")
'''
    open(tmp_path / "child.py", "w").write(child)
    parent = common + '''
commet_amd.device_cache_trim()
with commet_amd.Context(k=25, t=2) as ctx:                      # (a 16 MiB filter: pooled as well)
    a, q = commet_amd.ReadSet.from_files(ctx, [s0]), commet_amd.ReadSet.from_files(ctx, [s1])
    pooled = commet_amd.device_pooled_bytes(0)
    assert pooled >= 2 * 9000000 + (1 << 24), pooled
    before = ctx.index_and_search(a, [q])
    filed0, pooled = commet_amd.device_cache_bytes(0), commet_amd.device_pooled_bytes(0)   # (the job took workspaces from the pool too)
    open(d + "/set.blob", "wb").write(a.export())
    moved = pooled - commet_amd.device_pooled_bytes(0)
    assert moved >= 9000000, moved                                # the planes left the pool (the old block is filed)
    filed1 = commet_amd.device_cache_bytes(0)
    assert filed1 == filed0 + moved, (filed0, filed1, moved)
    after = ctx.index_and_search(a, [q])
    assert np.array_equal(before[0][0], after[0][0]) and before[1][0]["shared"] == after[1][0]["shared"] > 1000
    pooled = commet_amd.device_pooled_bytes(0)
    again = a.export()                                            # nothing left to move
    assert commet_amd.device_pooled_bytes(0) == pooled
    p = subprocess.run([sys.executable, d + "/child.py"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert p.returncode == 0 and b"imported ok" in p.stdout, p.stdout.decode()[-2000:]
    assert np.array_equal(np.load(d + "/child_tags.npy"), before[0][0])
    b = commet_amd.ReadSet.from_files(ctx, [s0])                  # the next set takes the filed (pooled) block of the first one's planes
    assert commet_amd.device_cache_bytes(0) <= filed1 - moved, (commet_amd.device_cache_bytes(0), filed1, moved)
assert commet_amd.device_pooled_bytes(0) == 0
filed = commet_amd.device_cache_bytes(0)
assert commet_amd.device_cache_trim() == filed > 0 and commet_amd.device_cache_bytes(0) == 0
print("parent ok")
'''
    env = dict(os.environ, COMMET_DEVMEM_POOL="1", COMMET_DEVMEM_POOL_MIN_MB="8")
    env.pop("COMMET_DEVMEM_CACHE", None)
    p = subprocess.run([sys.executable, "-c", parent], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0 and b"parent ok" in p.stdout, p.stdout.decode()[-3000:]

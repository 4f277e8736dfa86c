"""BASELINE.json's full sizes.  configs[1] (2 x 10 M x 100 bp, k=32, t=2) is too big to replay read by read
on one CPU core inside a test, so parity at this size is checked
  (1) bit-exactly on a random SAMPLE of query reads: the CPU checker builds the same two chunk filters from
      the full index set (chunk boundaries taken from its own k-mer counts, incl. the dropped look-ahead
      read) and replays search_reads on the sample only;
  (2) through size-independent properties: no false negatives (an indexed read with a clean run of
      >= t*k bases is always found), tags(t=2) subset of tags(t=1), counts = popcounts, the atomic and the
      bucketed index constructions give identical tags.
configs[4]'s shape (k=21, t=5, 150 bp: ~1900-read chunks, L2-resident filter) is replayed in full at reduced
read counts (the reference's geometry: 2^(k-1)-byte filter, SURVEY 8d note on the "8 GiB" figure)."""
import numpy as np
import pytest

import oracle_binding as ob
import util

pytestmark = pytest.mark.gpu


def _chunks_from_counts(kc, max_kmer):
    """chunk read ranges of an unfiltered single-file set (index_reads.h:49,60: look-ahead read dropped)"""
    out, n, i = [], len(kc), 0
    pre = np.concatenate([[0], np.cumsum(kc, dtype=np.int64)])
    while i < n:
        e = int(np.searchsorted(pre, pre[i] + max_kmer, side="left"))     # first e with pre[e] - pre[i] >= max
        e = min(max(e, i + 1), n)
        out.append((i, e))
        i = e + 1
    return out


@pytest.fixture(scope="module")
def c2():
    import commet_amd
    from commet_amd import synth
    n, L = 10_000_000, 100
    b0, o0 = synth.synth_set(0, n, L)
    b1, o1 = synth.synth_set(1, n, L)
    ctx = commet_amd.Context(k=32, t=2)
    irs = commet_amd.ReadSet.from_files(ctx, [(b0, o0)])
    qrs = commet_amd.ReadSet.from_files(ctx, [(b1, o1)])
    tags, stats, info = ctx.index_and_search(irs, [qrs])
    yield dict(ctx=ctx, irs=irs, qrs=qrs, b0=b0, o0=o0, b1=b1, o1=o1, tags=tags[0], stats=stats[0], info=info, n=n, L=L)
    ctx.close()


def test_c2_counts_are_consistent(c2):
    n = c2["n"]
    found = util.bools_from_bits(c2["tags"], n)
    assert int(found.sum()) == c2["stats"]["shared"]
    assert c2["info"]["n_chunks"] == 2                                   # 6.7e8 k-mers / max_kmer 5e8
    assert c2["stats"]["indexed"] == n - 1                               # one look-ahead read dropped
    assert 0 < c2["stats"]["searched"] <= n                               # reads scanned in the last pass only
    # 25 % of set 1 are (mutated) copies of set 0's first reads: nearly all of them are found
    assert found[: n // 4].mean() > 0.85 and found[n // 4:].mean() < 0.01


def test_c2_sample_is_bit_exact_against_cpu_checker(c2):
    k, t, n, L = 32, 2, c2["n"], c2["L"]
    kc = c2["irs"].kmer_counts()
    chunks = _chunks_from_counts(kc, ob.max_kmer(k))
    assert len(chunks) == c2["info"]["n_chunks"]
    rng = np.random.default_rng(7)
    sample = np.sort(np.concatenate([rng.choice(n // 4, 6000, replace=False), n // 4 + rng.choice(n - n // 4, 14000, replace=False)]))
    sb = c2["b1"].reshape(n, L)[sample].reshape(-1)
    so = np.arange(len(sample) + 1, dtype=np.uint64) * np.uint64(L)
    found = np.zeros(len(sample) // 8 + 1, dtype=np.uint8)
    for (a, e) in chunks:
        f = ob.Bloom(k)
        fed = f.index(c2["b0"][a * L: e * L], c2["o0"][a: e + 1] - c2["o0"][a])
        assert fed == int(kc[a:e].sum())
        active = ~found
        fnd, _ = f.search(t, sb, so, active)
        found |= fnd
        f.close()
    got = util.bools_from_bits(c2["tags"], n)[sample]
    assert np.array_equal(got, util.bools_from_bits(found, len(sample)))
    assert got.sum() > 4000                                               # the sample does contain shared reads


def test_c2_properties(c2):
    import commet_amd
    ctx, irs, qrs, n, L = c2["ctx"], c2["irs"], c2["qrs"], c2["n"], c2["L"]
    found2 = util.bools_from_bits(c2["tags"], n)
    # atomic construction of the filter gives the very same tags
    ctx.set_option("index_mode", 1)
    tags_a, st_a, _ = ctx.index_and_search(irs, [qrs])
    ctx.set_option("index_mode", 0)
    assert np.array_equal(tags_a[0], c2["tags"]) and st_a[0] == {**c2["stats"], "search_ms": st_a[0]["search_ms"]}
    # t = 1 finds a superset
    with commet_amd.Context(k=32, t=1) as c1:
        i1 = commet_amd.ReadSet.from_files(c1, [(c2["b0"], c2["o0"])])
        q1 = commet_amd.ReadSet.from_files(c1, [(c2["b1"], c2["o1"])])
        t1, _, _ = c1.index_and_search(i1, [q1])
        found1 = util.bools_from_bits(t1[0], n)
        assert not (found2 & ~found1).any() and found1.sum() > found2.sum()
        # no false negatives: set 0 searched in itself — every indexed read with >= 2*k clean bases is found
        self_tags, self_st, self_info = ctx.index_and_search(irs, [commet_amd.ReadSet.from_files(ctx, [(c2["b0"], c2["o0"])])])
        fs = util.bools_from_bits(self_tags[0], n)
        clean = (c2["b0"].reshape(n, L) != ord("N")).all(axis=1)
        dropped = np.zeros(n, dtype=bool)
        for (a, e) in _chunks_from_counts(irs.kmer_counts(), ob.max_kmer(32))[:-1]:
            dropped[e] = True
        assert fs[clean & ~dropped].all()


def test_k33_default_k_sample_is_bit_exact_against_cpu_checker(c2):
    """the reference's DEFAULT k (index_and_search.cpp:71, Commet.py:453) on configs[1]'s sets: 64-bit keys, a 4 GiB filter,
    one chunk (6.8e8 k-mers < max_kmer = 1e9), the tiled search on 64-bit keys (tq_*<uint64_t>); bit-exact on a sample,
    and the same bits from the plain search kernel"""
    import commet_amd
    k, t, n, L = 33, 2, c2["n"], c2["L"]
    with commet_amd.Context(k=k, t=t) as ctx:
        irs = commet_amd.ReadSet.from_files(ctx, [(c2["b0"], c2["o0"])])
        qrs = commet_amd.ReadSet.from_files(ctx, [(c2["b1"], c2["o1"])])
        kc = irs.kmer_counts()
        ctx.set_option("kernel_timing", 1)
        tags, stats, info = ctx.index_and_search(irs, [qrs])
        times = ctx.kernel_times()
        ctx.set_option("kernel_timing", 0)
        assert info["n_chunks"] == 1 and "tq_replay_kernel" in times and "search_kernel" not in times
        assert qrs.cache_bytes > 0
        ctx.set_option("tiled_search", 1)
        tags_g, stats_g, _ = ctx.index_and_search(irs, [qrs])
        assert np.array_equal(tags_g[0], tags[0])
        assert (stats_g[0]["indexed"], stats_g[0]["searched"], stats_g[0]["shared"]) == (stats[0]["indexed"], stats[0]["searched"], stats[0]["shared"])
    chunks = _chunks_from_counts(kc, ob.max_kmer(k))
    assert len(chunks) == 1 and stats[0]["indexed"] == n
    rng = np.random.default_rng(33)
    sample = np.sort(np.concatenate([rng.choice(n // 4, 6000, replace=False), n // 4 + rng.choice(n - n // 4, 14000, replace=False)]))
    sb = c2["b1"].reshape(n, L)[sample].reshape(-1)
    so = np.arange(len(sample) + 1, dtype=np.uint64) * np.uint64(L)
    f = ob.Bloom(k)
    fed = f.index(c2["b0"], c2["o0"])
    assert fed == int(kc.sum()) == info["kmers_indexed"]
    fnd, _ = f.search(t, sb, so, np.full(len(sample) // 8 + 1, 255, dtype=np.uint8))
    f.close()
    got = util.bools_from_bits(tags[0], n)[sample]
    assert np.array_equal(got, util.bools_from_bits(fnd, len(sample)))
    assert got.sum() > 4000 and stats[0]["shared"] == int(util.bools_from_bits(tags[0], n).sum())


def test_c2_sized_skewed_sets_sample_is_bit_exact_against_cpu_checker():
    """configs[1]'s size with 10 % of every set's reads low-complexity or repeated (poly-A, tandem repeats, a shared
    library of 1000 reads: synth.skew_set) — hot buckets in hist / scatter / build (split tiles), heavy scans in the
    search.  Bit-exact on a sample that holds 4000 of the replaced reads; atomic and bucketed constructions agree."""
    import commet_amd
    from commet_amd import synth
    k, t, n, L = 32, 2, 10_000_000, 100
    b0, o0 = synth.synth_set_skewed(0, n, L, 0.10)
    b1, o1 = synth.synth_set_skewed(1, n, L, 0.10)
    with commet_amd.Context(k=k, t=t) as ctx:
        irs = commet_amd.ReadSet.from_files(ctx, [(b0, o0)])
        qrs = commet_amd.ReadSet.from_files(ctx, [(b1, o1)])
        kc = irs.kmer_counts()
        tags, stats, info = ctx.index_and_search(irs, [qrs])
        ctx.set_option("index_mode", 1)
        ctx.set_option("tiled_search", 1)
        tags_a, stats_a, _ = ctx.index_and_search(irs, [qrs])            # atomic index kernel, gather search kernels
        assert np.array_equal(tags_a[0], tags[0]) and stats_a[0]["shared"] == stats[0]["shared"]
    chunks = _chunks_from_counts(kc, ob.max_kmer(k))
    assert len(chunks) == info["n_chunks"] == 2
    rng = np.random.default_rng(17)
    replaced = synth.skew_set(np.array(b1), n, L, 1, 0.10)               # (the same draw: which reads of set 1 were replaced)
    sample = np.unique(np.concatenate([rng.choice(n // 4, 5000, replace=False), n // 4 + rng.choice(n - n // 4, 11000, replace=False),
                                       rng.choice(replaced, 4000, replace=False)]))
    sb = np.ascontiguousarray(b1.reshape(n, L)[sample]).reshape(-1)
    so = np.arange(len(sample) + 1, dtype=np.uint64) * np.uint64(L)
    found = np.zeros(len(sample) // 8 + 1, dtype=np.uint8)
    for (a, e) in chunks:
        f = ob.Bloom(k)
        fed = f.index(b0[a * L: e * L], o0[a: e + 1] - o0[a])
        assert fed == int(kc[a:e].sum())
        fnd, _ = f.search(t, sb, so, ~found)
        found |= fnd
        f.close()
    got = util.bools_from_bits(tags[0], n)[sample]
    assert np.array_equal(got, util.bools_from_bits(found, len(sample)))
    # the replaced reads are mostly shared (every set holds poly-A reads, the same repeat units, the same library)
    is_rep = np.isin(sample, replaced)
    assert got[is_rep].mean() > 0.6 and got.sum() > 5000


@pytest.mark.parametrize("n_index,n_query", [(30000, 30000)])
def test_c5_shape_many_small_chunks(n_index, n_query):
    """k=21, t=5, 150 bp: max_kmer = 244 140 -> ~1880-read chunks, 1 MiB filter; full replay on the CPU checker"""
    import commet_amd
    from commet_amd import synth
    k, t, L = 21, 5, 150
    b0, o0 = synth.synth_set(0, n_index, L)
    b1, o1 = synth.synth_set(1, n_query, L, copy_frac=0.4)
    with commet_amd.Context(k=k, t=t) as ctx:
        irs = commet_amd.ReadSet.from_files(ctx, [(b0, o0)])
        qrs = commet_amd.ReadSet.from_files(ctx, [(b1, o1)])
        tags, stats, info = ctx.index_and_search(irs, [qrs])
        kc = irs.kmer_counts()
    chunks = _chunks_from_counts(kc, ob.max_kmer(k))
    assert info["n_chunks"] == len(chunks) >= 15
    found = np.zeros(n_query // 8 + 1, dtype=np.uint8)
    searched_last = 0
    for (a, e) in chunks:
        f = ob.Bloom(k)
        f.index(b0[a * L: e * L], o0[a: e + 1] - o0[a])
        active = ~found
        searched_last = int(util.bools_from_bits(active, n_query).sum())
        fnd, _ = f.search(t, b1, o1, active)
        found |= fnd
        f.close()
    assert np.array_equal(tags[0], found)
    assert stats[0]["shared"] == int(util.bools_from_bits(found, n_query).sum())
    assert stats[0]["searched"] == searched_last
    assert stats[0]["indexed"] == sum(e - a for a, e in chunks)

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


def _sample_of(rng, n, extra=None):
    parts = [rng.choice(n // 4, 6000, replace=False), n // 4 + rng.choice(n - n // 4, 14000, replace=False)]
    if extra is not None:
        parts.append(extra)
    return np.unique(np.concatenate(parts))


def _cpu_checker_on_sample(k, t, b0, o0, L, chunks, kc, sb):
    """the reference's chunk loop on the CPU checker: every chunk's filter from the whole index set, the sample searched
    against it (reads found in an earlier chunk are skipped, as FileManager's tags make the reference do)"""
    ns = sb.size // L
    so = np.arange(ns + 1, dtype=np.uint64) * np.uint64(L)
    found = np.zeros(ns // 8 + 1, dtype=np.uint8)
    for (a, e) in chunks:
        f = ob.Bloom(k)
        fed = f.index(b0[a * L: e * L], o0[a: e + 1] - o0[a])
        assert fed == int(kc[a:e].sum())
        fnd, _ = f.search(t, sb, so, ~found)
        found |= fnd
        f.close()
    return util.bools_from_bits(found, ns)


def _cpu_checker_on_ragged_sample(k, t, b0, o0, chunks, kc, sb, so):
    """_cpu_checker_on_sample for sets of many read lengths: (sb, so) = the sampled search reads as a batch"""
    ns = len(so) - 1
    found = np.zeros(ns // 8 + 1, dtype=np.uint8)
    for (a, e) in chunks:
        f = ob.Bloom(k)
        fed = f.index(b0[int(o0[a]): int(o0[e])], o0[a: e + 1] - o0[a])
        assert fed == int(kc[a:e].sum())
        fnd, _ = f.search(t, sb, so, ~found)
        found |= fnd
        f.close()
    return util.bools_from_bits(found, ns)


def _oracle_main_loop(d, k, t, index_fa, query_fa):
    """ok_index_and_search — the CPU checker's restatement of the tool's own main (set files, FastaFile iteration, the chunk loop
    of index_and_search.cpp:241-277) — with its chunk trace: (rc, results, chunks, k-mers, trace rows, .bv bits)"""
    import ctypes as C
    import os
    for name, fa in (("index", index_fa), ("search", query_fa)):
        with open(os.path.join(d, name + ".txt"), "w") as fh:
            fh.write(("ref:" if name == "index" else "qry:") + fa + "\n")
    os.makedirs(os.path.join(d, "o")), os.makedirs(os.path.join(d, "l"))
    trace = np.zeros((64, 4), dtype=np.uint64)
    ob.load().ok_trace_begin(trace.ctypes.data_as(C.c_void_p), len(trace))
    rc, res, chunks, kmers = ob.index_and_search(os.path.join(d, "index.txt"), os.path.join(d, "search.txt"), os.path.join(d, "o"),
                                                 os.path.join(d, "l"), k, t)
    nch = ob.load().ok_trace_end()
    _, nbits, bits = util.read_bv(os.path.join(d, "o", os.path.basename(query_fa) + "_in_ref.bv"))
    return dict(rc=rc, res=res, chunks=chunks, kmers=kmers, trace=trace[:nch].copy(), n=nbits, bits=np.array(bits))


@pytest.fixture(scope="module")
def cpu_runs(c2, tmp_path_factory):
    """The three full-size sample replays of this file (configs[1] at k = 32, the same sets at the reference's default
    k = 33, configs[1]'s size with skewed sets) need ~45 s of ONE host core each — the CPU checker is the reference's
    sequential algorithm — so the GPU jobs are run first and the three replays then run side by side in threads (the
    checker is called through ctypes, which releases the interpreter lock)."""
    import commet_amd
    from commet_amd import synth
    from concurrent.futures import ThreadPoolExecutor
    n, L, t = c2["n"], c2["L"], 2
    out = {}
    # (a) configs[1], k = 32: the module's GPU job
    kc32 = c2["irs"].kmer_counts()
    ch32 = _chunks_from_counts(kc32, ob.max_kmer(32))
    smp32 = _sample_of(np.random.default_rng(7), n)
    out["c2"] = dict(sample=smp32, chunks=ch32)
    # (b) the same sets at k = 33: 64-bit keys, a 4 GiB filter, one chunk, the tiled search on 64-bit keys
    with commet_amd.Context(k=33, t=t) as ctx:
        irs = commet_amd.ReadSet.from_files(ctx, [(c2["b0"], c2["o0"])])
        qrs = commet_amd.ReadSet.from_files(ctx, [(c2["b1"], c2["o1"])])
        kc33 = irs.kmer_counts()
        ctx.set_option("kernel_timing", 1)
        tags, stats, info = ctx.index_and_search(irs, [qrs])
        times = ctx.kernel_times()
        ctx.set_option("kernel_timing", 0)
        cache = qrs.cache_bytes
        ctx.set_option("tiled_search", 1)
        tags_g, stats_g, _ = ctx.index_and_search(irs, [qrs])
    smp33 = _sample_of(np.random.default_rng(33), n)
    out["k33"] = dict(tags=tags[0], stats=stats[0], info=info, times=times, cache=cache, tags_plain=tags_g[0], stats_plain=stats_g[0],
                      sample=smp33, chunks=_chunks_from_counts(kc33, ob.max_kmer(33)), kc=kc33)
    # (c) configs[1]'s size, 10 % of the reads low-complexity / repeated
    sb0, so0 = synth.synth_set_skewed(0, n, L, 0.10)
    sb1, so1 = synth.synth_set_skewed(1, n, L, 0.10)
    with commet_amd.Context(k=32, t=t) as ctx:
        irs = commet_amd.ReadSet.from_files(ctx, [(sb0, so0)])
        qrs = commet_amd.ReadSet.from_files(ctx, [(sb1, so1)])
        kcs = irs.kmer_counts()
        tags, stats, info = ctx.index_and_search(irs, [qrs])
        ctx.set_option("index_mode", 1)
        ctx.set_option("tiled_search", 1)
        tags_a, stats_a, _ = ctx.index_and_search(irs, [qrs])            # atomic index kernel, gather search kernels
    replaced = synth.skew_set(np.array(sb1), n, L, 1, 0.10)              # (the same draw: which reads of set 1 were replaced)
    smps = _sample_of(np.random.default_rng(17), n, np.random.default_rng(18).choice(replaced, 4000, replace=False))
    out["skew"] = dict(tags=tags[0], stats=stats[0], info=info, tags_atomic=tags_a[0], stats_atomic=stats_a[0], sample=smps,
                       replaced=replaced, chunks=_chunks_from_counts(kcs, ob.max_kmer(32)))

    # (e) configs[1]'s size on RAGGED sets (round 6): read lengths uniform in 50..150 — what a trimmed .fq.gz run looks like
    #     (fastq_file.h:139-190): the item list of hist / scatter1 (index_part.hpp, LIST), the query list sized by the set's real
    #     first-hit windows, the replay with three mask words; and the same job with the round planner and the gather kernels
    rb0, ro0 = synth.synth_set_ragged(0, n, 50, 150)
    rb1, ro1 = synth.synth_set_ragged(1, n, 50, 150)
    with commet_amd.Context(k=32, t=t) as ctx:
        irs = commet_amd.ReadSet.from_files(ctx, [(rb0, ro0)])
        qrs = commet_amd.ReadSet.from_files(ctx, [(rb1, ro1)])
        kcr = irs.kmer_counts()
        ctx.set_option("kernel_timing", 1)
        tags, stats, info = ctx.index_and_search(irs, [qrs])
        times = ctx.kernel_times()
        ctx.set_option("kernel_timing", 0)
        cache = qrs.cache_bytes
        ctx.set_option("part_list", 1)
        ctx.set_option("tiled_search", 1)
        tags_p, stats_p, _ = ctx.index_and_search(irs, [qrs])            # round planner, gather search kernels
    smpr = _sample_of(np.random.default_rng(61), n)
    lens1 = np.diff(ro1.astype(np.int64))
    rso = np.zeros(len(smpr) + 1, dtype=np.uint64)
    np.cumsum(lens1[smpr], out=rso[1:].view(np.int64))
    rsb = np.concatenate([rb1[int(ro1[i]): int(ro1[i + 1])] for i in smpr])
    out["ragged"] = dict(tags=tags[0], stats=stats[0], info=info, times=times, cache=cache, tags_plain=tags_p[0], stats_plain=stats_p[0],
                         sample=smpr, chunks=_chunks_from_counts(kcr, ob.max_kmer(32)), kc=kcr, lens1=lens1)

    def pick(b, smp):
        return np.ascontiguousarray(b.reshape(n, L)[smp]).reshape(-1)

    # (d) the checker's own MAIN LOOP over the whole index set of configs[1] (chunk boundaries from the restated tool, not from this
    #     file's numpy) and a search set of set 1's first reads
    md = tmp_path_factory.mktemp("c2main")
    m_q = 4000
    synth.write_fasta_fast(str(md / "s0.fa"), c2["b0"], n, L)
    synth.write_fasta_fast(str(md / "q.fa"), c2["b1"][: m_q * L], m_q, L)

    with ThreadPoolExecutor(5) as pool:
        fe = pool.submit(_cpu_checker_on_ragged_sample, 32, t, rb0, ro0, out["ragged"]["chunks"], kcr, rsb, rso)
        fd = pool.submit(_oracle_main_loop, str(md), 32, t, str(md / "s0.fa"), str(md / "q.fa"))
        fa = pool.submit(_cpu_checker_on_sample, 32, t, c2["b0"], c2["o0"], L, ch32, kc32, pick(c2["b1"], smp32))
        fb = pool.submit(_cpu_checker_on_sample, 33, t, c2["b0"], c2["o0"], L, out["k33"]["chunks"], kc33, pick(c2["b1"], smp33))
        fc = pool.submit(_cpu_checker_on_sample, 32, t, sb0, so0, L, out["skew"]["chunks"], kcs, pick(sb1, smps))
        out["c2"]["want"], out["k33"]["want"], out["skew"]["want"] = fa.result(), fb.result(), fc.result()
        out["c2"]["main"] = fd.result()
        out["ragged"]["want"] = fe.result()
    import os
    os.unlink(md / "s0.fa")
    return out


def test_c2_counts_are_consistent(c2):
    n = c2["n"]
    found = util.bools_from_bits(c2["tags"], n)
    assert int(found.sum()) == c2["stats"]["shared"]
    assert c2["info"]["n_chunks"] == 2                                   # 6.7e8 k-mers / max_kmer 5e8
    assert c2["stats"]["indexed"] == n - 1                               # one look-ahead read dropped
    assert 0 < c2["stats"]["searched"] <= n                               # reads scanned in the last pass only
    # 25 % of set 1 are (mutated) copies of set 0's first reads: nearly all of them are found
    assert found[: n // 4].mean() > 0.85 and found[n // 4:].mean() < 0.01


def test_c2_sample_is_bit_exact_against_cpu_checker(c2, cpu_runs):
    run = cpu_runs["c2"]
    assert len(run["chunks"]) == c2["info"]["n_chunks"]
    got = util.bools_from_bits(c2["tags"], c2["n"])[run["sample"]]
    assert np.array_equal(got, run["want"])
    assert got.sum() > 4000                                               # the sample does contain shared reads


def test_c2_chunks_and_first_reads_match_the_checkers_main_loop(c2, cpu_runs):
    """configs[1]'s whole index set through the CPU checker's restatement of the tool's main (FASTA parsing, FastaFile iteration,
    the max_kmer chunk loop): its chunk trace — first / last read, reads, k-mers of every chunk — is what the device's plan and
    this file's numpy boundaries say, and its .bv for a search set of set 1's first 4000 reads equals the job's tags for them"""
    main, n = cpu_runs["c2"]["main"], c2["n"]
    assert main["rc"] == 0 and main["chunks"] == c2["info"]["n_chunks"] == len(main["trace"])
    assert main["kmers"] == c2["info"]["kmers_indexed"]
    kc = c2["irs"].kmer_counts()
    for (a, e), row in zip(cpu_runs["c2"]["chunks"], main["trace"]):
        first, last, reads, kmers = (int(x) for x in row)
        assert (first, last, reads) == (a, e - 1, e - a) and kmers == int(kc[a:e].sum())
    assert main["res"][0]["indexed"] == c2["stats"]["indexed"] == n - 1
    m_q = main["n"]
    got = util.bools_from_bits(c2["tags"], n)[:m_q]
    want = util.bools_from_bits(main["bits"], m_q)
    assert np.array_equal(got, want) and main["res"][0]["shared"] == int(want.sum()) > 3000


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


def test_k33_default_k_sample_is_bit_exact_against_cpu_checker(c2, cpu_runs):
    """the reference's DEFAULT k (index_and_search.cpp:71, Commet.py:453) on configs[1]'s sets: 64-bit keys, a 4 GiB filter,
    one chunk (6.8e8 k-mers < max_kmer = 1e9), the tiled search on 64-bit keys (tq_*<uint64_t>); bit-exact on a sample,
    and the same bits from the plain search kernel"""
    run, n = cpu_runs["k33"], c2["n"]
    assert run["info"]["n_chunks"] == 1 == len(run["chunks"]) and run["stats"]["indexed"] == n
    assert "tq_replay_kernel" in run["times"] and "search_kernel" not in run["times"] and run["cache"] > 0
    assert np.array_equal(run["tags_plain"], run["tags"])
    assert [run["stats_plain"][f] for f in ("indexed", "searched", "shared")] == [run["stats"][f] for f in ("indexed", "searched", "shared")]
    assert run["info"]["kmers_indexed"] == int(run["kc"].sum())
    got = util.bools_from_bits(run["tags"], n)[run["sample"]]
    assert np.array_equal(got, run["want"])
    assert got.sum() > 4000 and run["stats"]["shared"] == int(util.bools_from_bits(run["tags"], n).sum())


def test_c2_sized_skewed_sets_sample_is_bit_exact_against_cpu_checker(c2, cpu_runs):
    """configs[1]'s size with 10 % of every set's reads low-complexity or repeated (poly-A, tandem repeats, a shared
    library of 1000 reads: synth.skew_set) — hot buckets in hist / scatter / build (split tiles, the aggregated atomics
    of scatter2), heavy scans in the search.  Bit-exact on a sample that holds 4000 of the replaced reads; atomic and
    bucketed constructions agree."""
    run, n = cpu_runs["skew"], c2["n"]
    assert np.array_equal(run["tags_atomic"], run["tags"]) and run["stats_atomic"]["shared"] == run["stats"]["shared"]
    assert len(run["chunks"]) == run["info"]["n_chunks"] == 2
    got = util.bools_from_bits(run["tags"], n)[run["sample"]]
    assert np.array_equal(got, run["want"])
    # the replaced reads are mostly shared (every set holds poly-A reads, the same repeat units, the same library)
    is_rep = np.isin(run["sample"], run["replaced"])
    assert is_rep.sum() >= 4000 and got[is_rep].mean() > 0.6 and got.sum() > 5000


def test_c2_sized_ragged_sets_sample_is_bit_exact_against_cpu_checker(c2, cpu_runs):
    """configs[1]'s size with read lengths uniform in 50..150 bp (the reference's real inputs are trimmed reads, fastq_file.h:139-190):
    the fast paths that used to be reserved for reads of one length — hist / scatter1 on the chunk's item list, the tiled search on a
    query list sized by the set's own first-hit windows — bit-exact on a sample replayed by the CPU checker, and equal to the same
    job through the round planner and the gather kernels"""
    run, n = cpu_runs["ragged"], c2["n"]
    assert len(run["chunks"]) == run["info"]["n_chunks"] == 2 and run["info"]["kmers_indexed"] == int(run["kc"].sum()) - int(run["kc"][run["chunks"][0][1]])
    assert "tq_replay_kernel" in run["times"] and "part_items_kernels" in run["times"] and run["cache"] > 0
    assert np.array_equal(run["tags_plain"], run["tags"])
    assert [run["stats_plain"][f] for f in ("indexed", "searched", "shared")] == [run["stats"][f] for f in ("indexed", "searched", "shared")]
    found = util.bools_from_bits(run["tags"], n)
    got = found[run["sample"]]
    assert np.array_equal(got, run["want"])
    assert got.sum() > 3500 and run["stats"]["shared"] == int(found.sum())
    # a read shorter than t * k bases cannot hold t non-overlapping k-mers
    assert not found[run["lens1"] < 64].any() and found[: n // 4][run["lens1"][: n // 4] >= 80].mean() > 0.85


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

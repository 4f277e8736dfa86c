"""BASELINE.json configs[2] (10 sets x 10 M reads, the full 10 x 10 matrix on one GPU) and a configs[3]-sized pair
(2 x 50 M reads: 7 index chunks, searched in ONE pass by search_group8_kernel) at full size, plus the
search_group8_kernel instantiations the smaller scenarios cannot reach (three mask words; 64-bit keys).

Full-size parity, as the CPU checker allows it:
  * bit-exact on a random SAMPLE of query reads, for every job of one pair chain J1 -> J2 -> J3 of the matrix
    (Commet.py:186-240): the CPU checker builds the chunk filters of the WHOLE index set (chunk boundaries from its
    own rule, index_reads.h:49,60; the chunks replayed in parallel processes, tests/oracle_pool.py) and replays
    search_reads on the sample;
  * size-independent properties: matrix invariants, J3 subset of J1, every grouping of the chunk filters
    (chunk_group 1 = the reference's order ... 8) gives the same bits.
"""
import os

import numpy as np
import pytest

import oracle_binding as ob
import oracle_pool
import util

pytestmark = pytest.mark.gpu


def _sample(rng, n, n_first, n_rest):
    return np.sort(np.concatenate([rng.choice(n // 4, n_first, replace=False),
                                   n // 4 + rng.choice(n - n // 4, n_rest, replace=False)]))


def _bits_at(bits, n, idx):
    return util.bools_from_bits(bits, n)[idx]


def _replay(scratch, tag, bases, L, sel_bools, kc, k, t, sample_bases):
    """CPU checker: index the selected reads of `bases` chunk by chunk, search the sample; returns bool[m]"""
    n = len(kc)
    if sel_bools is None:
        ib, ikc = bases, kc
    else:
        idx = np.flatnonzero(sel_bools)
        ib = np.ascontiguousarray(bases.reshape(n, L)[idx]).reshape(-1)
        ikc = kc[idx]
    chunks = oracle_pool.chunks_from_counts(ikc, ob.max_kmer(k))
    found, fed = oracle_pool.search_sample_over_chunks(scratch, tag, ib, L, chunks, k, t, sample_bases)
    assert fed == [int(ikc[a:e].sum()) for a, e in chunks]
    return found, len(chunks)


# ---------------------------------------------------------------------------------------------------------------
# configs[2]: 10 synthetic sets x 10 M reads, full 10 x 10 matrix, one GPU
# ---------------------------------------------------------------------------------------------------------------
C3_N, C3_L, C3_SETS, C3_PAIR = 10_000_000, 100, 10, (3, 7)


@pytest.fixture(scope="module")
def c3(tmp_path_factory):
    from commet_amd import build, matrix, synth
    build.build_lib()
    build.build_tools()
    d = tmp_path_factory.mktemp("c3")
    keep = {}
    import multiprocessing as mp
    with open(d / "sets.txt", "w") as fh:
        for s in range(C3_SETS):
            fh.write(f"S{s}: {d}/set{s}.fa\n")
    # the sets are generated side by side in worker processes (spawned: no GPU state is inherited); the two sets of
    # the pair that is replayed on the CPU checker are made here, where their bases are needed
    with mp.get_context("spawn").Pool(min(8, max(1, (os.cpu_count() or 2) - 1))) as pool:
        pending = pool.map_async(util.gen_set_fasta, [(s, C3_N, C3_L, str(d / f"set{s}.fa")) for s in range(C3_SETS)
                                                       if s not in C3_PAIR], chunksize=1)
        for s in C3_PAIR:
            keep[s], _ = synth.synth_set(s, C3_N, C3_L)
            synth.write_fasta_fast(str(d / f"set{s}.fa"), keep[s], C3_N, C3_L)
        pending.get()
    res = matrix.run(str(d / "sets.txt"), str(d / "out") + "/", k=32, t=2, verbose=False)
    yield dict(dir=d, res=res, bases=keep)
    for s in range(C3_SETS):
        os.remove(d / f"set{s}.fa")


def test_c3_matrix_invariants(c3):
    res, N = c3["res"], C3_SETS
    m, considered = res["matrix"], res["considered"]
    assert considered == [C3_N] * N                                # filter_reads with -l 0 -e 0 keeps every read
    for i in range(N):
        assert m[i][i] == considered[i]
        for j in range(N):
            assert 0 <= m[i][j] <= considered[i]
            if i != j:
                # a quarter of every set derives from set 0's first reads (1 % substitutions each): most of them are
                # shared, and almost nothing else is
                assert 0.50 * C3_N / 4 < m[i][j] < 1.05 * C3_N / 4, (i, j, m[i][j])
    out = str(c3["dir"] / "out")
    assert sum(1 for f in os.listdir(out) if "_in_" in f and f.endswith(".bv")) == N * (N - 1)
    # every matrix cell is the bit count of its .bv file (what Commet.py reads back through bvop -i, Commet.py:247-262)
    for (a, b) in [(0, 9), (9, 0), C3_PAIR, C3_PAIR[::-1], (5, 4)]:
        _, n, bits = util.read_bv(os.path.join(out, f"set{a}.fa_in_S{b}.bv"))
        assert n == C3_N and int(util.bools_from_bits(bits, n).sum()) == m[a][b]
    assert res["reads_searched"] == 3 * C3_N * N * (N - 1) // 2   # J1 + J2 + J3 of every pair, each over 10 M considered reads
    assert res["world"] == 1 and res["jobs_s"] > 0


def test_c3_pair_chain_sample_is_bit_exact_against_cpu_checker(c3, tmp_path):
    """J1, J2, J3 of one pair (ref, i), every job checked on a 20 000-read sample of its search set"""
    import commet_amd
    ref, i = C3_PAIR
    k, t, n, L = 32, 2, C3_N, C3_L
    b_ref, b_i = c3["bases"][ref], c3["bases"][i]
    out = str(c3["dir"] / "out")
    offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    with commet_amd.Context(k=k, t=t) as ctx:
        rs_ref = commet_amd.ReadSet.from_files(ctx, [(b_ref, offs)])
        rs_i = commet_amd.ReadSet.from_files(ctx, [(b_i, offs)])
        kc_ref, kc_i = rs_ref.kmer_counts(), rs_i.kmer_counts()
        tags1, st1, inf1 = ctx.index_and_search(rs_ref, [rs_i])          # J1(ref, i) on its own
    T1 = util.bools_from_bits(tags1[0], n)
    _, n2, bits2 = util.read_bv(os.path.join(out, f"set{ref}.fa_in_S{i}.bv"))      # J2: S_ref in (S_i restricted to T1)
    _, n3, bits3 = util.read_bv(os.path.join(out, f"set{i}.fa_in_S{ref}.bv"))      # J3: S_i in (S_ref restricted to T2)
    T2, T3 = util.bools_from_bits(bits2, n2), util.bools_from_bits(bits3, n3)
    assert n2 == n3 == n
    assert not (T3 & ~T1).any()                                          # J3 indexes a subset of what J1 indexed
    assert T3.sum() == c3["res"]["matrix"][i][ref] and T2.sum() == c3["res"]["matrix"][ref][i]
    rng = np.random.default_rng(11)
    smp_i, smp_ref = _sample(rng, n, 6000, 14000), _sample(rng, n, 6000, 14000)
    sb_i = np.ascontiguousarray(b_i.reshape(n, L)[smp_i]).reshape(-1)
    sb_ref = np.ascontiguousarray(b_ref.reshape(n, L)[smp_ref]).reshape(-1)
    scratch = str(tmp_path)
    # the three replays side by side (each one runs its chunks in worker processes; J2 / J3 take the GPU's T1 / T2 as their
    # index selections, so they do not wait for one another)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(3) as pool:
        r1 = pool.submit(_replay, scratch, "j1", b_ref, L, None, kc_ref, k, t, sb_i)
        r2 = pool.submit(_replay, scratch, "j2", b_i, L, T1, kc_i, k, t, sb_ref)      # index set restricted to J1's result
        r3 = pool.submit(_replay, scratch, "j3", b_ref, L, T2, kc_ref, k, t, sb_i)
        (f1, nch1), (f2, _), (f3, _) = r1.result(), r2.result(), r3.result()
    assert nch1 == inf1["n_chunks"] == 2
    assert np.array_equal(T1[smp_i], f1)
    assert np.array_equal(T2[smp_ref], f2)
    assert np.array_equal(T3[smp_i], f3)
    assert f1.sum() > 4000 and f2.sum() > 4000 and f3.sum() > 4000      # the samples do contain shared reads


# ---------------------------------------------------------------------------------------------------------------
# search_group8_kernel: three mask words (65..96 first-hit windows) and 64-bit keys, more than four chunks each
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k,t,L,max_kmer,n_idx,lo", [
    (20, 1, 110, 0, 9000, None),          # 91 first-hit windows -> <u32, 3>; the reference's own chunk size (122 070 k-mers)
    (24, 2, 140, 60000, 6000, None),      # 93 windows -> <u32, 3>
    (33, 2, 100, 50000, 6000, None),      # 35 windows -> <u64, 2>
    (34, 1, 128, 60000, 5000, None),      # 95 windows -> <u64, 3>
    (33, 3, 150, 40000, 4000, None),      # 52 windows, t = 3 -> <u64, 2>
    # round 6: four and six mask words (97..128, 129..192 first-hit windows: reads of up to 255 bases at k = 32, t = 2), ragged sets
    (20, 1, 140, 0, 11000, 40),           # 121 windows -> <u32, 4>
    (24, 2, 230, 60000, 4000, 60),        # 183 windows -> <u32, 6>
    (33, 1, 150, 40000, 4000, 50),        # 118 windows -> <u64, 4>
    (34, 2, 250, 60000, 3000, 70),        # 183 windows -> <u64, 6>
    (32, 2, 250, 60000, 3000, 250),       # 187 windows, reads of one length -> <u32, 6>
    (32, 2, 300, 80000, 2500, 120),       # 237 windows (MiSeq-length reads) -> <u32, 8>
    (33, 2, 318, 80000, 2500, 318),       # 253 windows -> <u64, 8>
])
def test_group8_instantiations_match_cpu_checker(k, t, L, max_kmer, n_idx, lo):
    """max_kmer != 0 uses the library's test hook (k-mers per chunk) so that k >= 33 gets more than four chunks from a
    few thousand reads; the CPU checker is then chunked with the same constant.  lo: read lengths uniform in [lo, L]
    (None: all reads of L bases)."""
    import commet_amd
    rng = np.random.default_rng(1000 * k + L)
    lo = L if lo is None else lo
    idx_reads = util.random_reads(rng, n_idx, lo, L, n_rate=0.002)
    q_reads = util.related_reads(rng, idx_reads, 12000, lo, L, share=0.5, n_rate=0.002)
    q_reads[0] = (q_reads[1] * 8)[:L]                                      # (the set's longest read has L bases whatever the draw)
    ib, io = util.to_batch(idx_reads)
    qb, qo = util.to_batch(q_reads)
    with commet_amd.Context(k=k, t=t) as ctx:
        if max_kmer:
            ctx.set_option("max_kmer", max_kmer)
        ctx.set_option("slice_mode", 1)                                    # k <= 24 with 8 chunks or more would take the bit-sliced regime
        irs = commet_amd.ReadSet.from_files(ctx, [(ib, io)])
        qrs = commet_amd.ReadSet.from_files(ctx, [(qb, qo)])
        kc = irs.kmer_counts()
        res = {}
        for group in (8, 4, 1):
            ctx.set_option("chunk_group", group)
            res[group] = ctx.index_and_search(irs, [qrs])
        if lo != L:
            # ragged sets: the first pass walks the set's reads in order of their window counts, segment by segment with the
            # narrowest masks a segment's reads fit (capi/search_dispatch.hpp, launch_search_group; `mask_split` 1 = one launch)
            ctx.set_option("chunk_group", 8)
            ctx.set_option("ordered_scan", 2)
            launches = {}
            for split_off in (0, 1):
                ctx.set_option("mask_split", split_off)
                ctx.set_option("kernel_timing", 1)
                res["split" if not split_off else "whole"] = ctx.index_and_search(irs, [qrs])
                launches[split_off] = ctx.kernel_times()["search_group8_kernel"][0]
                ctx.set_option("kernel_timing", 0)
            passes = (len(oracle_pool.chunks_from_counts(kc, max_kmer or ob.max_kmer(k))) + 7) // 8
            assert launches[1] == passes and launches[0] >= passes + 2, launches     # the first pass ran in three segments or more
    chunks = oracle_pool.chunks_from_counts(kc, max_kmer or ob.max_kmer(k))
    assert len(chunks) > 4
    tags, stats, info = res[8]
    assert info["n_chunks"] == len(chunks)
    assert info["search_launches"] == (len(chunks) + 7) // 8               # the eight-filter kernel did run
    found, searched_last = oracle_pool.chunk_loop_in_threads(k, t, ib, io, qb, qo, chunks, len(q_reads))
    for group in res:
        tg, sg, _ = res[group]
        assert np.array_equal(tg[0], found), group
        assert sg[0]["shared"] == int(util.bools_from_bits(found, len(q_reads)).sum())
        assert sg[0]["searched"] == searched_last and sg[0]["indexed"] == sum(e - a for a, e in chunks)
    assert stats[0]["shared"] > 2000


# ---------------------------------------------------------------------------------------------------------------
# configs[4]'s regime (k = 21, t = 5, 150-bp reads: 1 MiB filters, ~1 880-read chunks) on a 200 000 x 200 000-read slice
# ---------------------------------------------------------------------------------------------------------------
def test_c5_regime_200k_slice_matches_cpu_checker(tmp_path):
    """107 chunks; the bit-sliced regime (auto: 128 chunk filters per pass) and the slot path must both give the CPU
    checker's bits and log numbers.  The CPU checker replays every chunk against ALL query reads (workers in parallel);
    a read belongs to the first chunk that finds it."""
    import commet_amd
    from commet_amd import synth
    k, t, L, n = 21, 5, 150, 200_000
    b0, o0 = synth.synth_set(0, n, L)
    b1, o1 = synth.synth_set(1, n, L, copy_frac=0.4)
    res = {}
    with commet_amd.Context(k=k, t=t) as ctx:
        irs = commet_amd.ReadSet.from_files(ctx, [(b0, o0)])
        qrs = commet_amd.ReadSet.from_files(ctx, [(b1, o1)])
        kc = irs.kmer_counts()
        res["auto"] = ctx.index_and_search(irs, [qrs])
        for words in (1, 8):
            ctx.set_option("slice_words", words)
            res[words] = ctx.index_and_search(irs, [qrs])
        ctx.set_option("slice_words", 0)
        ctx.set_option("slice_wide", 2)                                    # wide rows (auto only above 256 chunks): one pass, LPR = 8
        res["wide"] = ctx.index_and_search(irs, [qrs])
        ctx.set_option("slice_wide", 0)
        ctx.set_option("slice_mode", 1)
        res["slots"] = ctx.index_and_search(irs, [qrs])
    chunks = oracle_pool.chunks_from_counts(kc, ob.max_kmer(k))
    assert len(chunks) > 100 and res["wide"][2]["search_launches"] == 1
    found, fed, first = oracle_pool.search_sample_over_chunks(str(tmp_path), "c5", b0, L, chunks, k, t, b1, first_chunk=True)
    assert fed == [int(kc[a:e].sum()) for a, e in chunks]
    searched_last = n - int(((first >= 0) & (first < len(chunks) - 1)).sum())
    assert res["auto"][2]["search_launches"] == 1 and res[1][2]["search_launches"] == (len(chunks) + 31) // 32
    for name, (tags, stats, info) in res.items():
        assert info["n_chunks"] == len(chunks), name
        assert np.array_equal(util.bools_from_bits(tags[0], n), found), name
        assert (stats[0]["indexed"], stats[0]["searched"], stats[0]["shared"]) == \
            (sum(e - a for a, e in chunks), searched_last, int(found.sum())), name
    assert found[: int(0.4 * n)].mean() > 0.5 and found.sum() > 50_000


# ---------------------------------------------------------------------------------------------------------------
# search_wide_kernel: every (lanes per read, pieces per lane) instantiation, one and several passes
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k,t,L,n_chunks,cap_words,inst", [
    (12, 1, 40, 300, 0, "8x1"),          # 2 groups of 256 chunks -> 16 words per row -> 8 lanes per read
    (14, 2, 60, 1500, 0, "16x1"),        # 48 words
    (16, 2, 80, 3000, 0, "32x1"),        # 96 words
    (15, 3, 90, 6000, 0, "64x1"),        # 192 words
    (13, 2, 50, 9000, 0, "64x2"),        # 288 words: two 16-byte pieces per lane
    (16, 2, 80, 2900, 48, "16x1"),       # 12 groups in 2 passes of 6 (rows capped at 48 words)
    (21, 5, 150, 700, 8, "8x1"),         # configs[4]'s k, t, read length; 3 passes of one group
])
def test_wide_rows_instantiations_match_cpu_checker(k, t, L, n_chunks, cap_words, inst):
    """max_kmer = 1 (the library's test hook) makes every chunk one read (plus the dropped look-ahead read), so
    a few thousand reads give the thousands of chunk filters the wide instantiations are chosen by; the CPU checker runs
    the reference's chunk loop with the same constant."""
    import commet_amd
    rng = np.random.default_rng(77 * k + n_chunks)
    idx_reads = util.random_reads(rng, 2 * n_chunks, L, L, n_rate=0.002)
    q_reads = util.related_reads(rng, idx_reads, 1200, L, L, share=0.5, n_rate=0.002)
    ib, io = util.to_batch(idx_reads)
    qb, qo = util.to_batch(q_reads)
    with commet_amd.Context(k=k, t=t) as ctx:
        ctx.set_option("max_kmer", 1)
        ctx.set_option("slice_mode", 2)
        ctx.set_option("slice_wide", 2)                                    # (auto would probe first and might take the narrow tables)
        ctx.set_option("slice_wide_words", cap_words)
        irs = commet_amd.ReadSet.from_files(ctx, [(ib, io)])
        qrs = commet_amd.ReadSet.from_files(ctx, [(qb, qo)])
        kc = irs.kmer_counts()
        ctx.set_option("kernel_timing", 1)
        tags, stats, info = ctx.index_and_search(irs, [qrs])
        times = ctx.kernel_times()
        ctx.set_option("slice_wide", 1)                                    # the narrow tables, 256 chunks per pass
        narrow = ctx.index_and_search(irs, [qrs])
    chunks = oracle_pool.chunks_from_counts(kc, 1)
    assert abs(len(chunks) - n_chunks) <= n_chunks // 50 and info["n_chunks"] == len(chunks)
    groups = (len(chunks) + 255) // 256
    passes = 1 if not cap_words else -(-groups * 8 // cap_words)
    assert info["search_launches"] == passes and times["search_wide_kernel"][0] == passes
    nw = -(-groups // passes) * 8
    pieces = nw // 4
    assert inst == f"{8 if pieces <= 8 else 16 if pieces <= 16 else 32 if pieces <= 32 else 64}x{1 if pieces <= 64 else 2}"
    found, searched_last = oracle_pool.chunk_loop_in_threads(k, t, ib, io, qb, qo, chunks, len(q_reads))
    for tg, sg in ((tags, stats), (narrow[0], narrow[1])):
        assert np.array_equal(tg[0], found)
        assert sg[0]["shared"] == int(util.bools_from_bits(found, len(q_reads)).sum())
        assert sg[0]["searched"] == searched_last and sg[0]["indexed"] == sum(e - a for a, e in chunks)
    assert stats[0]["shared"] > 60


# ---------------------------------------------------------------------------------------------------------------
# instantiations the randomised scenarios do not reach (tests/test_gpu_zz_dispatch_coverage.py keeps the list honest)
# ---------------------------------------------------------------------------------------------------------------
def _chunk_loop_on_cpu_checker(k, t, ib, io, qb, qo, chunks, n_q):
    return oracle_pool.chunk_loop_in_threads(k, t, ib, io, qb, qo, chunks, n_q)


@pytest.mark.parametrize("k,t,L,max_kmer,group", [(33, 2, 100, 40000, 2), (34, 2, 110, 30000, 4)])
def test_probe_counting_group_kernels_with_64_bit_keys(k, t, L, max_kmer, group):
    """search_group_kernel<uint64_t, 2 | 4, COUNT = true>: the builds that reproduce the reference's probe count, on 64-bit
    keys, for groups of chunk filters: the CPU checker's bits, and the probe count of the one-filter-at-a-time kernel"""
    import commet_amd
    rng = np.random.default_rng(5 * k + group)
    idx_reads = util.random_reads(rng, 3000, L, L, n_rate=0.002)
    q_reads = util.related_reads(rng, idx_reads, 4000, L - 20, L + 20, share=0.5, n_rate=0.002)
    ib, io = util.to_batch(idx_reads)
    qb, qo = util.to_batch(q_reads)
    res = {}
    with commet_amd.Context(k=k, t=t) as ctx:
        ctx.set_option("max_kmer", max_kmer)
        ctx.set_option("count_probes", 1)
        irs = commet_amd.ReadSet.from_files(ctx, [(ib, io)])
        qrs = commet_amd.ReadSet.from_files(ctx, [(qb, qo)])
        kc = irs.kmer_counts()
        for g in (group, 1):
            ctx.set_option("chunk_group", g)
            res[g] = ctx.index_and_search(irs, [qrs])
    chunks = oracle_pool.chunks_from_counts(kc, max_kmer)
    assert len(chunks) >= 4 and res[group][2]["search_launches"] == -(-len(chunks) // group) and res[1][2]["search_launches"] == len(chunks)
    found, searched_last = _chunk_loop_on_cpu_checker(k, t, ib, io, qb, qo, chunks, len(q_reads))
    for g in (group, 1):
        tags, stats, info = res[g]
        assert np.array_equal(tags[0], found), g
        assert (stats[0]["searched"], stats[0]["shared"]) == (searched_last, int(util.bools_from_bits(found, len(q_reads)).sum())), g
    assert res[group][2]["probes"] == res[1][2]["probes"] > 0
    assert res[1][1][0]["shared"] > 500


@pytest.mark.parametrize("k,t,L,max_kmer,lo,words", [(26, 1, 110, 90000, 110, 3), (33, 1, 120, 90000, 120, 3),
                                                      (26, 1, 150, 90000, 60, 4), (33, 1, 155, 90000, 155, 4),
                                                      (28, 2, 240, 120000, 80, 6), (34, 1, 220, 120000, 90, 6), (32, 2, 250, 150000, 100, 6),
                                                      (32, 2, 300, 150000, 150, 8), (33, 1, 287, 200000, 287, 8)])
def test_tiled_replay_three_to_eight_mask_words(k, t, L, max_kmer, lo, words):
    """tq_replay_kernel<W, 1 | 2, 3 | 4 | 6 | 8>: 65..96 / 97..128 / 129..192 / 193..255 first-hit windows per read (three to eight mask
    words; round 6: reads of up to 318 bases at k = 32, t = 2 keep the tiled search) against groups of two chunk filters and against
    single ones; lo < L: ragged sets"""
    import commet_amd
    rng = np.random.default_rng(11 * k)
    idx_reads = util.random_reads(rng, 4000, lo, L, n_rate=0.002)
    q_reads = util.related_reads(rng, idx_reads, 6000, lo, L, share=0.5, n_rate=0.002)
    q_reads[0] = (q_reads[1] * 8)[:L]                                      # (the set's longest read has L bases whatever the draw)
    ib, io = util.to_batch(idx_reads)
    qb, qo = util.to_batch(q_reads)
    fhw = L - t * k + 1
    assert words == (2 if fhw <= 64 else 3 if fhw <= 96 else 4 if fhw <= 128 else 6 if fhw <= 192 else 8)
    got = {}
    with commet_amd.Context(k=k, t=t) as ctx:
        ctx.set_option("max_kmer", max_kmer)
        ctx.set_option("tiled_search", 2)
        irs = commet_amd.ReadSet.from_files(ctx, [(ib, io)])
        qrs = commet_amd.ReadSet.from_files(ctx, [(qb, qo)])
        kc = irs.kmer_counts()
        for group in (2, 1):
            ctx.set_option("chunk_group", group)
            ctx.set_option("kernel_timing", 1)
            got[group] = ctx.index_and_search(irs, [qrs])
            times = ctx.kernel_times()
            ctx.set_option("kernel_timing", 0)
            assert "tq_replay_kernel" in times and "search_kernel" not in times and "search_group_kernel" not in times and qrs.cache_bytes > 0
    chunks = oracle_pool.chunks_from_counts(kc, max_kmer)
    assert len(chunks) >= 4
    found, searched_last = _chunk_loop_on_cpu_checker(k, t, ib, io, qb, qo, chunks, len(q_reads))
    for group in (2, 1):
        tags, stats, info = got[group]
        assert np.array_equal(tags[0], found), group
        assert (stats[0]["searched"], stats[0]["shared"]) == (searched_last, int(util.bools_from_bits(found, len(q_reads)).sum()))
    assert got[2][1][0]["shared"] > 1000


def test_auto_mode_probes_before_choosing_wide_rows_or_narrow_tables():
    """more than 256 chunk filters in auto mode: the first 256 are searched with the narrow tables against one read in
    sixteen; when most of those are found there (reads that will be found early whatever the plan) the job goes on with
    the narrow tables, group by group, else with the wide rows.  Either way the CPU checker's bits."""
    import commet_amd
    k, t, L, n_chunks = 16, 2, 80, 4200
    rng = np.random.default_rng(99)
    idx_reads = util.random_reads(rng, 2 * n_chunks, L, L, n_rate=0.002)
    early = [idx_reads[int(i)] for i in rng.integers(0, 500, size=4000)]        # copies of reads of the first 250 chunks
    late = util.random_reads(rng, 4000, L, L, n_rate=0.002)                      # reads that share nothing
    ib, io = util.to_batch(idx_reads)
    for name, q_reads, kernel in (("found early", early, "search_sliced_kernel"), ("not found", late, "search_wide_kernel")):
        qb, qo = util.to_batch(q_reads)
        with commet_amd.Context(k=k, t=t) as ctx:
            ctx.set_option("max_kmer", 1)
            irs = commet_amd.ReadSet.from_files(ctx, [(ib, io)])
            qrs = commet_amd.ReadSet.from_files(ctx, [(qb, qo)])
            kc = irs.kmer_counts()
            ctx.set_option("kernel_timing", 1)
            tags, stats, info = ctx.index_and_search(irs, [qrs])
            times = ctx.kernel_times()
        chunks = oracle_pool.chunks_from_counts(kc, 1)
        assert info["n_chunks"] == len(chunks) > 4000
        groups = (len(chunks) + 255) // 256
        # the probe is one launch of the narrow kernel; then either every group with it, or one wide pass
        if kernel == "search_sliced_kernel":
            assert "search_wide_kernel" not in times and times["search_sliced_kernel"][0] == 1 + groups, (name, times)
        else:
            assert times["search_sliced_kernel"][0] == 1 and times["search_wide_kernel"][0] == 1, (name, times)
        found, searched_last = _chunk_loop_on_cpu_checker(k, t, ib, io, qb, qo, chunks, len(q_reads))
        assert np.array_equal(tags[0], found), name
        assert (stats[0]["searched"], stats[0]["shared"]) == (searched_last, int(util.bools_from_bits(found, len(q_reads)).sum())), name
        assert (stats[0]["shared"] > 1500) == (kernel == "search_sliced_kernel"), (name, stats[0]["shared"])


# ---------------------------------------------------------------------------------------------------------------
# configs[4] AS STATED: k = 21, t = 5, 2 x 20 M x 150 bp reads -> ~10 400 chunk filters of 1 MiB, auto mode
# ---------------------------------------------------------------------------------------------------------------
def test_c5_full_size_sample_matches_cpu_checker(tmp_path):
    """BASELINE configs[4] at full size, on the code path a user gets (auto mode: the probe, then one pass of
    search_wide_kernel over the rows of ALL chunk filters — the 64-lane, two-pieces-per-lane instantiation).  The CPU
    checker builds every one of the ~10 400 chunk filters (index_reads.h:49-61 with the reference's max_kmer = 244 140 and
    the dropped look-ahead reads) in worker processes and runs search_reads (search_reads.h:45-83) of a 10 000-read sample
    of the query set against each of them; a read's tag is the OR over the chunks (index_and_search.cpp:255-277).  Beside
    the sample: the log numbers' size-independent relations at full size."""
    import commet_amd
    from commet_amd import synth
    from concurrent.futures import ThreadPoolExecutor
    k, t, L, n = 21, 5, 150, 20_000_000
    b0, o0 = synth.synth_set(0, n, L)
    b1, o1 = synth.synth_set(1, n, L)
    rng = np.random.default_rng(21)
    smp = _sample(rng, n, 4000, 6000)
    sb = np.ascontiguousarray(b1.reshape(n, L)[smp]).reshape(-1)
    with commet_amd.Context(k=k, t=t) as ctx, ThreadPoolExecutor(1) as cpu:
        irs = commet_amd.ReadSet.from_files(ctx, [(b0, o0)])
        qrs = commet_amd.ReadSet.from_files(ctx, [(b1, o1)])
        del o0, o1
        kc = irs.kmer_counts()
        chunks = oracle_pool.chunks_from_counts(kc, ob.max_kmer(k))
        assert ob.max_kmer(k) == 244_140 and 10_000 < len(chunks) < 10_800
        # the CPU replay runs beside the GPU job
        replay = cpu.submit(oracle_pool.search_sample_over_chunks, str(tmp_path), "c5full", b0, L, chunks, k, t, sb,
                            first_chunk=True, sample_len=L)
        ctx.set_option("kernel_timing", 1)
        tags, stats, info = ctx.index_and_search(irs, [qrs])
        times = ctx.kernel_times()
        found_cpu, fed, first = replay.result()
    assert info["n_chunks"] == len(chunks)
    assert times["search_wide_kernel"][0] == 1 and times["search_sliced_kernel"][0] == 1      # the probe, then ONE wide pass
    assert fed == [int(kc[a:e].sum()) for a, e in chunks] and info["kmers_indexed"] == sum(fed)
    found = util.bools_from_bits(tags[0], n)
    assert np.array_equal(found[smp], found_cpu)
    assert found_cpu[:4000].sum() > 2000                                   # the sample does contain shared reads ...
    assert len(np.unique(first[first >= 0])) > 1500                        # ... first found in chunk filters all over the job
    # log numbers at full size: indexed = every read but the dropped look-ahead ones; shared = the tags' bit count;
    # searched = the reads the LAST chunk's pass still scans = all but those found before it
    assert stats[0]["indexed"] == sum(e - a for a, e in chunks) == n - (len(chunks) - 1) - (0 if chunks[-1][1] == n else 1)
    assert stats[0]["shared"] == int(found.sum())
    assert n - stats[0]["shared"] <= stats[0]["searched"] <= n
    last_only = int((first == len(chunks) - 1).sum())                      # sample reads that only the last chunk finds: rare
    assert stats[0]["searched"] - (n - stats[0]["shared"]) <= max(1, last_only) * (n // len(smp)) * 4 + 4000
    assert found[: n // 4].mean() > 0.5 and found[n // 4:].mean() < 0.05


# ---------------------------------------------------------------------------------------------------------------
# configs[3] AS STATED, through the N x N driver: 10 sets x 50 M reads, all 99 jobs' worth on one GPU
# ---------------------------------------------------------------------------------------------------------------
C4_N, C4_L, C4_SETS, C4_PAIR = 50_000_000, 100, 10, (2, 6)


def _big_scratch(tmp_path_factory, need_bytes):
    """a directory for `need_bytes` of FASTA files: /dev/shm when it (and the host's memory) holds them with room to spare"""
    import shutil
    import tempfile
    try:
        free_shm = shutil.disk_usage("/dev/shm").free
        avail = next(int(ln.split()[1]) * 1024 for ln in open("/proc/meminfo") if ln.startswith("MemAvailable:"))
        if os.access("/dev/shm", os.W_OK) and free_shm > need_bytes * 1.2 and avail > need_bytes + (110 << 30):
            return tempfile.mkdtemp(prefix="commet_c4m_", dir="/dev/shm"), True
    except Exception:
        pass
    return str(tmp_path_factory.mktemp("c4m")), False


@pytest.fixture(scope="module")
def c4m(tmp_path_factory):
    """matrix.run on BASELINE configs[3] (the run `bench.py` times as its matrix leg), once for the module; the two sets of
    the pair that is replayed on the CPU checker are generated here, where their bases are needed"""
    import multiprocessing as mp
    import shutil
    from commet_amd import build, matrix, synth
    build.build_lib()
    build.build_tools()
    d, in_shm = _big_scratch(tmp_path_factory, C4_SETS * C4_N * (C4_L + 12))
    keep = {}
    try:
        with open(os.path.join(d, "sets.txt"), "w") as fh:
            for s in range(C4_SETS):
                fh.write(f"S{s}: {d}/set{s}.fa\n")
        with mp.get_context("spawn").Pool(4) as pool:                       # (a 50 M-read set is ~8 GB of generator state per worker)
            pending = pool.map_async(util.gen_set_fasta, [(s, C4_N, C4_L, os.path.join(d, f"set{s}.fa")) for s in range(C4_SETS)
                                                           if s not in C4_PAIR], chunksize=1)
            for s in C4_PAIR:
                keep[s], _ = synth.synth_set(s, C4_N, C4_L)
                synth.write_fasta_fast(os.path.join(d, f"set{s}.fa"), keep[s], C4_N, C4_L)
            pending.get()
        res = matrix.run(os.path.join(d, "sets.txt"), os.path.join(d, "out") + "/", k=32, t=2, verbose=False)
        for s in range(C4_SETS):                                            # 56 GB back before the CPU replays start
            os.remove(os.path.join(d, f"set{s}.fa"))
        yield dict(dir=d, res=res, bases=keep)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_c4_matrix_as_the_driver_runs_it(c4m, tmp_path):
    """The 10 x 50 M matrix as commet_amd.matrix produces it on one GPU — J1 of a reference set built once and shared by its
    targets (7 chunk filters per pass, search_group8_kernel), J2 / J3 on selection lists, 24 GB of packed sets resident, the
    sets loaded beside the jobs — against the CPU checker: one pair chain replayed on 20 000-read samples FROM THE DRIVER'S OWN
    .bv FILES (J2's and J3's outputs; J1's tags, which the driver never writes, recomputed through the API and then proved
    by J2's replay, whose index selection they are), plus the matrix's invariants."""
    import commet_amd
    from concurrent.futures import ThreadPoolExecutor
    res, N = c4m["res"], C4_SETS
    m, considered = res["matrix"], res["considered"]
    n, L, k, t = C4_N, C4_L, 32, 2
    assert considered == [n] * N and res["world"] == 1
    assert res["reads_searched"] == 3 * n * N * (N - 1) // 2
    for i in range(N):
        for j in range(N):
            if i != j:
                assert 0.50 * n / 4 < m[i][j] < 1.05 * n / 4, (i, j, m[i][j])
    out = os.path.join(c4m["dir"], "out")
    assert sum(1 for f in os.listdir(out) if "_in_" in f and f.endswith(".bv")) == N * (N - 1)
    ref, i = C4_PAIR
    b_ref, b_i = c4m["bases"][ref], c4m["bases"][i]
    offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    with commet_amd.Context(k=k, t=t) as ctx:
        rs_ref = commet_amd.ReadSet.from_files(ctx, [(b_ref, offs)])
        rs_i = commet_amd.ReadSet.from_files(ctx, [(b_i, offs)])
        kc_ref, kc_i = rs_ref.kmer_counts(), rs_i.kmer_counts()
        tags1, st1, inf1 = ctx.index_and_search(rs_ref, [rs_i])          # J1(ref, i) on its own: up to 8 chunk filters per pass
        assert inf1["n_chunks"] == 7 and inf1["search_launches"] == 1
        for group in (1, 4):                                             # the reference's order; the LDS-mask kernel (4 + 3 filters)
            ctx.set_option("chunk_group", group)
            tg, sg, ig = ctx.index_and_search(rs_ref, [rs_i])
            assert ig["search_launches"] == (7 if group == 1 else 2)
            assert np.array_equal(tg[0], tags1[0]), group
            assert (sg[0]["indexed"], sg[0]["searched"], sg[0]["shared"]) == (st1[0]["indexed"], st1[0]["searched"], st1[0]["shared"]), group
        ctx.set_option("chunk_group", 8)
        # the rest of the pair's chain through the API (Commet.py:186-240): the same bits as the driver's files
        tags2, st2, inf2 = ctx.index_and_search(rs_i, [rs_ref], index_select=tags1[0])
        tags3, st3, inf3 = ctx.index_and_search(rs_ref, [rs_i], index_select=tags2[0])
        assert inf2["n_chunks"] == 2 and inf3["n_chunks"] == 2           # a quarter of 50 M reads: 8.6e8 k-mers
    del offs
    T1 = util.bools_from_bits(tags1[0], n)
    assert st1[0]["shared"] == int(T1.sum()) and st1[0]["indexed"] == n - 6      # six look-ahead reads dropped
    assert T1[: n // 4].mean() > 0.4 and T1[n // 4:].mean() < 0.02      # (both sets' first quarters are mutated copies of set 0's)
    _, n2, bits2 = util.read_bv(os.path.join(out, f"set{ref}.fa_in_S{i}.bv"))      # J2: S_ref in (S_i restricted to T1)
    _, n3, bits3 = util.read_bv(os.path.join(out, f"set{i}.fa_in_S{ref}.bv"))      # J3: S_i in (S_ref restricted to T2)
    T2, T3 = util.bools_from_bits(bits2, n2), util.bools_from_bits(bits3, n3)
    assert n2 == n3 == n and inf1["n_chunks"] == 7
    assert np.array_equal(util.bools_from_bits(tags2[0], n), T2) and np.array_equal(util.bools_from_bits(tags3[0], n), T3)
    # (T3 within T1 is NOT an invariant: J3's two chunks each span reads that J1 had in several of its seven chunks, so two
    # hits that J1 saw in different filters can meet in one of J3's; the sample replays below are the check.  It is rare.)
    assert int((T3 & ~T1).sum()) < n // 1000
    assert T3.sum() == m[i][ref] and T2.sum() == m[ref][i]
    for (a, b) in [(0, 9), (9, 0), (5, 4)]:
        _, nn, bits = util.read_bv(os.path.join(out, f"set{a}.fa_in_S{b}.bv"))
        assert nn == n and int(util.bools_from_bits(bits, nn).sum()) == m[a][b]
    rng = np.random.default_rng(13)
    smp_i, smp_ref = _sample(rng, n, 6000, 14000), _sample(rng, n, 6000, 14000)
    sb_i = np.ascontiguousarray(b_i.reshape(n, L)[smp_i]).reshape(-1)
    sb_ref = np.ascontiguousarray(b_ref.reshape(n, L)[smp_ref]).reshape(-1)
    scratch = str(tmp_path)
    with ThreadPoolExecutor(3) as pool:   # 7 + 2 + 2 chunk filters, each built by a worker process of its own
        r1 = pool.submit(_replay, scratch, "m1", b_ref, L, None, kc_ref, k, t, sb_i)
        r2 = pool.submit(_replay, scratch, "m2", b_i, L, T1, kc_i, k, t, sb_ref)      # index set restricted to J1's result
        r3 = pool.submit(_replay, scratch, "m3", b_ref, L, T2, kc_ref, k, t, sb_i)
        (f1, nch1), (f2, nch2), (f3, nch3) = r1.result(), r2.result(), r3.result()
    assert (nch1, nch2, nch3) == (7, 2, 2)
    assert np.array_equal(T1[smp_i], f1)
    assert np.array_equal(T2[smp_ref], f2)           # the driver's J2 output (and with it the T1 it really used)
    assert np.array_equal(T3[smp_i], f3)             # the driver's J3 output
    assert f1.sum() > 4000 and f2.sum() > 4000 and f3.sum() > 4000

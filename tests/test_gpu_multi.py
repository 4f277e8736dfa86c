"""commet_index_many_and_search: several jobs that search the SAME read set (Commet.py's J2 jobs of a reference set, its J3 jobs of a
target, Commet.py:220, 233), their chunk filters side by side in one pass of search_group8_kernel — every job's tags and numbers must be
what commet_index_and_search gives for that job alone (which the other suites pin to the CPU checker); the N x N driver's use of the
call is checked against the checker itself in test_gpu_configs.py / test_gpu_matrix.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(rng, n, frac):
    sel = rng.random(n) < frac
    out = np.zeros(n // 8 + 1, dtype=np.uint8)
    pk = np.packbits(sel, bitorder="little")
    out[:pk.size] = pk
    return out


@pytest.mark.parametrize("k,t,L,max_kmer", [(32, 2, 100, 12_000_000), (33, 2, 100, 9_000_000), (26, 2, 130, 15_000_000), (28, 3, 110, 0),
                                             (32, 2, (50, 150), 12_000_000), (33, 2, (60, 140), 9_000_000), (27, 3, (90, 170), 14_000_000)])
def test_jobs_sharing_a_pass_equal_the_jobs_alone(k, t, L, max_kmer):
    """L = (lo, hi): ragged sets (round 6: index sets of many read lengths share passes, too)"""
    import commet_amd
    from commet_amd import synth
    rng = np.random.default_rng(k * 100 + t)
    n_i, n_s = 400_000, 600_000
    make = (lambda s, n: synth.synth_set_ragged(s, n, L[0], L[1])) if isinstance(L, tuple) else (lambda s, n: synth.synth_set(s, n, L))
    with commet_amd.Context(k=k, t=t) as ctx:
        ctx.set_option("index_mode", 2)                  # the bucketed construction whatever a chunk's size (the fast path's condition)
        ctx.set_option("max_kmer", max_kmer)             # several chunks per index set (test hook; both ways chunk alike)
        srs = commet_amd.ReadSet.from_files(ctx, [make(0, n_s)])
        irs = [commet_amd.ReadSet.from_files(ctx, [make(s, n_i)]) for s in (1, 2, 3, 4, 5)]
        sels = [None, _bits(rng, n_i, 0.5), _bits(rng, n_i, 0.2), _bits(rng, n_i, 0.9), None]
        # alone
        alone = [ctx.index_and_search(rs, [srs], index_select=sel) for rs, sel in zip(irs, sels)]
        # together
        ctx.set_option("kernel_timing", 1)
        tags, stats, info = ctx.index_many_and_search(irs, srs, index_selects=sels)
        times = ctx.kernel_times()
        ctx.set_option("kernel_timing", 0)
        chunks = [a[2]["n_chunks"] for a in alone]
        assert info["n_chunks"] == sum(chunks) and max(chunks) <= 8
        for j, a in enumerate(alone):
            assert np.array_equal(tags[j], a[0][0]), j
            assert {f: stats[j][f] for f in ("indexed", "searched", "shared")} == {f: a[1][0][f] for f in ("indexed", "searched", "shared")}, j
            assert a[1][0]["shared"] > n_i // 8 * (0.1 if sels[j] is not None else 0.5)
        assert info["kmers_indexed"] == sum(a[2]["kmers_indexed"] for a in alone)
        # the jobs did share passes: consecutive jobs while their chunks fit eight slots
        passes, g = 1, 0
        for c_ in chunks:
            if g + c_ > 8:
                passes, g = passes + 1, 0
            g += c_
        launches = times["search_group8_kernel"][0]
        if isinstance(L, tuple):       # a ragged search set: a pass runs over the set's length-ordered list, one launch per mask width (two here)
            assert launches == 2 * passes
        else:
            assert launches == passes
        assert passes < len(irs) and info["search_launches"] == passes
        # job by job on request, and whenever the fast path does not apply (a selection on the search set; a single job)
        ctx.set_option("multi_job", 1)
        t2, s2, i2 = ctx.index_many_and_search(irs, srs, index_selects=sels)
        ctx.set_option("multi_job", 0)
        assert i2["search_launches"] >= len(irs)
        ssel = _bits(rng, n_s, 0.3)
        t3, s3, i3 = ctx.index_many_and_search(irs[:2], srs, index_selects=sels[:2], search_select=ssel)
        for j in range(len(irs)):
            assert np.array_equal(t2[j], tags[j]) and s2[j]["shared"] == stats[j]["shared"]
        for j in range(2):
            a = ctx.index_and_search(irs[j], [srs], index_select=sels[j], search_selects=[ssel])
            assert np.array_equal(t3[j], a[0][0]) and s3[j]["shared"] == a[1][0]["shared"] and s3[j]["searched"] == a[1][0]["searched"]
        t4, s4, _ = ctx.index_many_and_search(irs[:1], srs, index_selects=sels[:1])
        assert np.array_equal(t4[0], tags[0]) and s4[0]["shared"] == stats[0]["shared"]


def test_shared_passes_fall_back_to_the_jobs_alone_when_eight_slots_do_not_fit():
    """commet_index_many_and_search on a device that has no room for the eight filter slots of a shared pass (16 GiB + 4 GiB of
    interleaved A planes at k = 32): the jobs run one after the other, as include/commet_hip.h promises — commet_index_and_search
    itself degrades the same way.  The device is filled with a read set of the right capacity (its planes are one allocation)."""
    import commet_amd
    from commet_amd import synth
    k, t, L, n_i, n_s = 32, 2, 100, 300_000, 400_000
    rng = np.random.default_rng(5)
    with commet_amd.Context(k=k, t=t) as ctx:
        ctx.set_option("index_mode", 2)
        ctx.set_option("max_kmer", 9_000_000)
        srs = commet_amd.ReadSet.from_files(ctx, [synth.synth_set(0, n_s, L)])
        irs = [commet_amd.ReadSet.from_files(ctx, [synth.synth_set(s, n_i, L)]) for s in (1, 2, 3)]
        sels = [None, _bits(rng, n_i, 0.5), None]
        alone = [ctx.index_and_search(rs, [srs], index_select=sel) for rs, sel in zip(irs, sels)]      # (two slots, workspaces: allocated now)
        commet_amd.device_cache_trim(ctx.device)
        free, total = ctx.device_memory()
        hog_bytes = free - (9 << 30)                     # leaves less than the 20 GiB a shared pass asks for
        assert hog_bytes > (64 << 30)
        hog = commet_amd.ReadSet(ctx, 1, int(hog_bytes / 0.375))      # 12 bytes of planes per 32 bases
        try:
            before = commet_amd.device_alloc_stats(ctx.device)
            tags, stats, info = ctx.index_many_and_search(irs, srs, index_selects=sels)
            assert info["search_launches"] >= len(irs)                 # no shared pass: every job searched on its own
            for j, a in enumerate(alone):
                assert np.array_equal(tags[j], a[0][0]), j
                assert {f: stats[j][f] for f in ("indexed", "searched", "shared")} == {f: a[1][0][f] for f in ("indexed", "searched", "shared")}, j
            assert commet_amd.device_alloc_stats(ctx.device)["calls"] > before["calls"]      # the driver WAS asked (and said no)
        finally:
            hog.close()
        # with the memory back the same call shares passes again
        ctx.set_option("kernel_timing", 1)
        t2, s2, i2 = ctx.index_many_and_search(irs, srs, index_selects=sels)
        assert "search_group8_kernel" in ctx.kernel_times() and i2["search_launches"] < len(irs)
        for j in range(len(irs)):
            assert np.array_equal(t2[j], tags[j])


@pytest.mark.parametrize("k,t,L", [(32, 2, 100), (33, 2, (60, 140)), (28, 1, 120)])
def test_single_chunk_jobs_go_through_the_tiled_search_in_twos(k, t, L):
    """commet_index_many_and_search, jobs of ONE chunk filter each on a search set that takes the tiled search (the J2 / J3 jobs of a
    matrix of 10 M-read sets): two jobs per scan — one probe, one replay that keeps the jobs apart (tq_replay_kernel, job_tag_words) —
    an odd job out on its own; every job's tags and numbers are those of the job alone"""
    import commet_amd
    from commet_amd import synth
    rng = np.random.default_rng(k * 10 + t)
    n_i, n_s = 250_000, 500_000
    make = (lambda s, n: synth.synth_set_ragged(s, n, L[0], L[1])) if isinstance(L, tuple) else (lambda s, n: synth.synth_set(s, n, L))
    with commet_amd.Context(k=k, t=t) as ctx:
        ctx.set_option("index_mode", 2)                  # the bucketed construction whatever a chunk's size (the shared paths' condition)
        ctx.set_option("tiled_search", 2)                # the tiled search whatever the set's size
        srs = commet_amd.ReadSet.from_files(ctx, [make(0, n_s)])
        irs = [commet_amd.ReadSet.from_files(ctx, [make(s, n_i)]) for s in (1, 2, 3, 4, 5)]
        sels = [_bits(rng, n_i, 0.3), None, _bits(rng, n_i, 0.2), _bits(rng, n_i, 0.6), None]
        alone = [ctx.index_and_search(rs, [srs], index_select=sel) for rs, sel in zip(irs, sels)]
        assert all(a[2]["n_chunks"] == 1 for a in alone)
        ctx.set_option("kernel_timing", 1)
        tags, stats, info = ctx.index_many_and_search(irs, srs, index_selects=sels)
        times = ctx.kernel_times()
        ctx.set_option("kernel_timing", 0)
        assert times["tq_replay_kernel"][0] == 3 and times["tq_probe_kernel"][0] == 3      # two scans of two jobs, one of the fifth
        assert "search_group8_kernel" not in times and info["search_launches"] == 3 and info["n_chunks"] == 5
        for j, a in enumerate(alone):
            assert np.array_equal(tags[j], a[0][0]), j
            assert {f: stats[j][f] for f in ("indexed", "searched", "shared")} == {f: a[1][0][f] for f in ("indexed", "searched", "shared")}, j
        assert stats[1]["shared"] > n_i // 8 * 0.5 and info["kmers_indexed"] == sum(a[2]["kmers_indexed"] for a in alone)
        # job by job on request: the same bits
        ctx.set_option("multi_job", 1)
        t2, s2, i2 = ctx.index_many_and_search(irs, srs, index_selects=sels)
        assert i2["search_launches"] == 5
        for j in range(5):
            assert np.array_equal(t2[j], tags[j]) and s2[j]["shared"] == stats[j]["shared"]
        ctx.set_option("multi_job", 0)
        # the same index set in both jobs of a scan, under two selections (one selection bitmap per set on the device)
        twice = [irs[0], irs[0], irs[2], irs[2]]
        sels2 = [sels[0], _bits(rng, n_i, 0.5), None, sels[2]]
        t3, s3, i3 = ctx.index_many_and_search(twice, srs, index_selects=sels2)
        assert i3["search_launches"] == 2
        for j, (rs, sel) in enumerate(zip(twice, sels2)):
            a = ctx.index_and_search(rs, [srs], index_select=sel)
            assert np.array_equal(t3[j], a[0][0]) and s3[j]["shared"] == a[1][0]["shared"] and s3[j]["indexed"] == a[1][0]["indexed"], j

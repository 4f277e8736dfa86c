"""Host-side chunk planner (commet_amd/csrc/read_iter.hpp, pure C++ — no GPU)
against the CPU checker's chunk trace: same chunks, same dropped reads, same
visited reads, for the generic iterator and for the prefix-sum fast path."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
import util
from conftest import ROOT
from scenarios import GoldenScenario, Scenario, run_oracle

# COMMET_PLAN_LIB: a prebuilt (e.g. sanitizer-instrumented, tests/test_sanitizers.py) planner library to test instead
PLAN_LIB = os.environ.get("COMMET_PLAN_LIB") or os.path.join(ROOT, "commet_amd", "libcommet_plan.so")
PLAN_SRC = os.path.join(ROOT, "commet_amd", "csrc", "host", "plan_capi.cpp")


@pytest.fixture(scope="module")
def plan():
    srcs = [PLAN_SRC, os.path.join(ROOT, "commet_amd", "csrc", "read_iter.hpp")]
    if os.environ.get("COMMET_PLAN_LIB"):
        pass
    elif not os.path.exists(PLAN_LIB) or any(os.path.getmtime(s) > os.path.getmtime(PLAN_LIB) for s in srcs):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", PLAN_LIB, PLAN_SRC], check=True)
    lib = C.CDLL(PLAN_LIB)
    lib.commet_plan_index.restype = C.c_uint64
    lib.commet_plan_index.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                      C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.commet_plan_search.restype = C.c_uint64
    lib.commet_plan_search.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int,
                                       C.c_void_p]
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def plan_index(lib, file_counts, select, kcnt, max_kmer, fast, empty=()):
    n = int(sum(file_counts))
    files = np.zeros(2 * len(file_counts), dtype=np.uint64)
    pos = 0
    for i, c in enumerate(file_counts):
        files[2 * i], files[2 * i + 1] = pos, c
        pos += c
    er = np.array(sorted(empty), dtype=np.uint64)
    cap = n + 8
    chunks = np.zeros((cap, 4), dtype=np.uint64)
    bits = np.zeros(n // 8 + 1, dtype=np.uint8)
    stats = np.zeros(2, dtype=np.uint64)
    kc = np.ascontiguousarray(kcnt, dtype=np.uint32)
    sel = None if select is None else util.bits_from_bools(select)
    nch = lib.commet_plan_index(_p(files), len(file_counts), _p(sel), _p(er), len(er), _p(kc), n, max_kmer, int(fast),
                                _p(chunks), cap, _p(bits), _p(stats))
    return chunks[:nch], util.bools_from_bits(bits, n), int(stats[0]), int(stats[1])


def _scenario_index_inputs(scn):
    files = scn.sets[scn.index_name]
    counts = [len(r) for _, _, r, _ in files]
    has_bv = any(bv for _, bv, _, _ in files)
    sel = np.concatenate([s for _, _, _, s in files]) if has_bv else None
    reads = [r for _, _, rr, _ in files for r in rr]
    # CRLF files keep the '\r' in the sequence: count k-mers on what the tools see
    parsed = [r for fa, _, _, _ in files for r in util.parse_reads(os.path.join(scn.dir, fa))]
    assert len(parsed) == len(reads)
    b, o = util.to_batch(parsed)
    return counts, sel, ob.kmer_counts(b, o, scn.k)


def _check_against_trace(lib, scn, tmp):
    trace = np.zeros((100000, 4), dtype=np.uint64)
    ob.load().ok_trace_begin(trace.ctypes.data_as(C.c_void_p), len(trace))
    rc, res, chunks, kmers = run_oracle(scn, os.path.join(tmp, "o"), os.path.join(tmp, "l"))
    nch = ob.load().ok_trace_end()
    assert rc == 0 and nch == chunks
    counts, sel, kcnt = _scenario_index_inputs(scn)
    got, bits, indexed, km = plan_index(lib, counts, sel, kcnt, ob.max_kmer(scn.k), fast=False)
    assert len(got) == nch
    for g, t in zip(got, trace[:nch]):
        assert g[2] == t[2] and g[3] == t[3]                    # reads, k-mers
        if t[2]:
            assert g[0] == t[0] and g[1] == t[1]                # first, last read number
    assert indexed == res[0]["indexed"] if res else True
    assert km == kmers
    # the fast path (when it applies) must agree with the generic one
    got_f, bits_f, indexed_f, km_f = plan_index(lib, counts, sel, kcnt, ob.max_kmer(scn.k), fast=True)
    assert np.array_equal(got, got_f) and np.array_equal(bits, bits_f) and (indexed, km) == (indexed_f, km_f)


@pytest.mark.parametrize("name", GoldenScenario.names())
def test_plan_matches_oracle_trace_golden(plan, tmp_path, name):
    _check_against_trace(plan, GoldenScenario(name), str(tmp_path))


@pytest.mark.parametrize("seed", range(3000, 3040))
def test_plan_matches_oracle_trace_random(plan, tmp_path, seed):
    _check_against_trace(plan, Scenario(str(tmp_path / "s"), seed, allow_bv=(seed % 2 == 0)), str(tmp_path))


@pytest.mark.parametrize("seed", range(30))
def test_fast_plan_equals_generic_plan(plan, seed):
    rng = np.random.default_rng(seed)
    nfiles = int(rng.integers(1, 4))
    counts = [int(rng.integers(1, 400)) for _ in range(nfiles)]
    n = sum(counts)
    kcnt = rng.integers(0, 90, size=n).astype(np.uint32)
    if seed % 3 == 0:
        kcnt[rng.random(n) < 0.3] = 0
    for max_kmer in (1, 7, 100, 1000, 10 ** 9):
        a = plan_index(plan, counts, None, kcnt, max_kmer, fast=False)
        b = plan_index(plan, counts, None, kcnt, max_kmer, fast=True)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:]


def test_search_plan_stops_at_empty_selection(plan):
    # a set whose next file has no selected read ends there; an empty FIRST file is skipped
    def visited(counts, sel, fast=False):
        n = sum(counts)
        files = np.zeros(2 * len(counts), dtype=np.uint64)
        pos = 0
        for i, c in enumerate(counts):
            files[2 * i], files[2 * i + 1] = pos, c
            pos += c
        bits = np.zeros(n // 8 + 1, dtype=np.uint8)
        s = None if sel is None else util.bits_from_bools(sel)
        nv = plan.commet_plan_search(_p(files), len(counts), _p(s), None, 0, n, int(fast), _p(bits))
        return nv, util.bools_from_bits(bits, n)

    sel = np.ones(9, dtype=bool)
    sel[3:6] = False                               # file 1 of 3 has nothing selected
    nv, v = visited([3, 3, 3], sel)
    assert nv == 3 and v.tolist() == [True] * 3 + [False] * 6
    sel = np.ones(9, dtype=bool)
    sel[0:3] = False                               # an empty first file is skipped
    nv, v = visited([3, 3, 3], sel)
    assert nv == 6 and v.tolist() == [False] * 3 + [True] * 6
    nv, v = visited([3, 3, 3], None, fast=True)
    assert nv == 9 and v.all()


@pytest.mark.parametrize("seed", range(60))
def test_selected_plans_equal_generic_plans(plan, seed):
    """input-filter fast paths (SelectedIterator / plan_search_select) vs the reference-shaped SetIterator"""
    rng = np.random.default_rng(1000 + seed)
    nfiles = int(rng.integers(1, 6))
    counts = [int(rng.integers(1, 300)) for _ in range(nfiles)]
    n = sum(counts)
    kcnt = rng.integers(0, 90, size=n).astype(np.uint32)
    sel = rng.random(n) < rng.uniform(0.05, 0.95)
    pos = 0
    for c in counts:                       # some files with nothing selected, at every position
        if rng.random() < 0.35:
            sel[pos:pos + c] = False
        pos += c
    for max_kmer in (0, 1, 50, 900, 10 ** 9):
        a = plan_index(plan, counts, sel, kcnt, max_kmer, fast=False)
        b = plan_index(plan, counts, sel, kcnt, max_kmer, fast=True)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:], (seed, max_kmer)
    files = np.zeros(2 * nfiles, dtype=np.uint64)
    pos = 0
    for i, c in enumerate(counts):
        files[2 * i], files[2 * i + 1] = pos, c
        pos += c
    s = util.bits_from_bools(sel)
    out = []
    for fast in (0, 1):
        bits = np.zeros(n // 8 + 1, dtype=np.uint8)
        nv = plan.commet_plan_search(_p(files), nfiles, _p(s), None, 0, n, fast, _p(bits))
        out.append((nv, bits.copy()))
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("seed", range(60))
def test_block_sum_selected_plan_equals_generic_plan(plan, seed):
    """selections are planned from per-block k-mer sums (the library gets them from the device) without per-read work
    over the set: same chunks, same dropped reads, same indexed bits as the reference-shaped iterator, for one and for
    many chunks, for every block size; selections that do not qualify (a file without a selected read) take the other
    planners"""
    rng = np.random.default_rng(5000 + seed)
    nfiles = int(rng.integers(1, 5))
    counts = [int(rng.integers(1, 400)) for _ in range(nfiles)]
    n = sum(counts)
    kcnt = rng.integers(0, 90, size=n).astype(np.uint32)
    if seed % 4 == 0:
        kcnt[rng.random(n) < 0.6] = 0
    if seed % 7 == 0:
        kcnt[:] = 0
    sel = rng.random(n) < rng.uniform(0.02, 0.95)
    pos = 0
    for c in counts:                       # mostly every file keeps a selected read (the qualifying shape)
        if not sel[pos:pos + c].any() and rng.random() < 0.85:
            sel[pos + int(rng.integers(0, c))] = True
        pos += c
    total = int(kcnt[sel].sum())
    for max_kmer in (1, 7, 60, 500, max(total, 1), total + 1, 10 ** 9):
        a = plan_index(plan, counts, sel, kcnt, max_kmer, fast=False)
        for log_bs in (0, 2, 5, 12):
            b = plan_index(plan, counts, sel, kcnt, max_kmer, fast=2 + log_bs)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:], (seed, max_kmer, log_bs)
    # no selection at all: the same planner with every read selected (what the library uses for whole sets)
    for max_kmer in (1, 7, 60, 500, 10 ** 9):
        a = plan_index(plan, counts, None, kcnt, max_kmer, fast=False)
        for log_bs in (0, 3, 12):
            b = plan_index(plan, counts, None, kcnt, max_kmer, fast=2 + log_bs)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:], (seed, max_kmer, log_bs)

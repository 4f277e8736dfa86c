"""Python mirror of the C ABI (include/commet_hip.h): same names, same argument
meaning, errors raised as CommetError with the library's message.  All compute
happens in the HIP library; numpy arrays only carry host buffers."""
import ctypes as C
import os
import weakref

import numpy as np

from . import lib as _l


class CommetError(RuntimeError):
    pass


def _err(lib):
    msg = lib.commet_last_error()
    return msg.decode() if msg else "unknown error"


def bits_nbytes(n):
    """bytes of a BooleanVector over n reads (boolean_vector.h:130)"""
    return n // 8 + 1


def _as_bits(arr, n, what):
    if arr is None:
        return None
    a = np.ascontiguousarray(arr, dtype=np.uint8)
    if a.size < bits_nbytes(n):
        raise CommetError(f"{what}: need {bits_nbytes(n)} bytes for {n} reads, got {a.size}")
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def device_count():
    """commet_device_count: HIP devices this process sees (0 without one; counting initialises nothing a later fork would mind)"""
    return int(_l.load().commet_device_count())


def device_cache_trim(device=-1):
    """commet_device_cache_trim: gives the device memory the library keeps for reuse back to the driver; returns the bytes released"""
    return int(_l.load().commet_device_cache_trim(int(device)))


def device_cache_bytes(device=0):
    """commet_device_cache_bytes: device memory filed for reuse on `device`"""
    return int(_l.load().commet_device_cache_bytes(int(device)))


def device_pooled_bytes(device=0):
    """commet_device_pooled_bytes: bytes in use on `device` that came from the stream-ordered pool (not shareable over HIP IPC)"""
    return int(_l.load().commet_device_pooled_bytes(int(device)))


def device_alloc_stats(device=-1):
    """commet_device_alloc_stats: what the library asked the driver for since the process started (blocks reused from its own
    cache do not count): {"wait_ms": host time inside hipMalloc, "fresh_bytes", "calls"}"""
    ms, by, n = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
    _l.load().commet_device_alloc_stats(int(device), C.byref(ms), C.byref(by), C.byref(n))
    return {"wait_ms": float(ms.value), "fresh_bytes": int(by.value), "calls": int(n.value)}


class Context:
    """commet_ctx: device, k, t, the 4-lane Bloom filter in HBM."""

    def __init__(self, k, t=2, device=0):
        self._lib = _l.load()
        self._h = self._lib.commet_create(int(device), int(k), int(t))
        if not self._h:
            raise CommetError(_err(self._lib))
        self.k = int(k)
        self.t = self._lib.commet_min_hits(self._h)
        self.device = int(device)
        self._readsets = weakref.WeakSet()

    def close(self):
        if getattr(self, "_h", None):
            for rs in list(self._readsets):      # read sets die with their context
                rs.close()
            self._lib.commet_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise CommetError(_err(self._lib))

    @property
    def max_kmer(self):
        return int(self._lib.commet_max_kmer(self._h))

    def set_option(self, name, value):
        self._check(self._lib.commet_set_option(self._h, name.encode(), int(value)))

    def synchronize(self):
        self._check(self._lib.commet_synchronize(self._h))

    def device_memory(self):
        """commet_device_memory: (free, total) bytes of the context's device; blocks the library keeps for reuse count as free"""
        f, t = C.c_uint64(0), C.c_uint64(0)
        self._check(self._lib.commet_device_memory(self._h, C.byref(f), C.byref(t)))
        return int(f.value), int(t.value)

    def filter_reset(self):
        self._check(self._lib.commet_filter_reset(self._h))

    def index_reads(self, rs, first=0, count=None, select_bits=None, want_kmers=True):
        n = rs.num_reads
        if count is None:
            count = n - first
        sel = _as_bits(select_bits, n, "select_bits")
        fed = C.c_uint64(0)
        self._check(self._lib.commet_index_reads(self._h, rs._h, first, count, _ptr(sel),
                                                 C.byref(fed) if want_kmers else None))
        return int(fed.value) if want_kmers else None

    def search_reads(self, rs, active_bits=None):
        n = rs.num_reads
        act = _as_bits(active_bits, n, "active_bits")
        found = np.zeros(bits_nbytes(n), dtype=np.uint8)
        scanned = C.c_uint64(0)
        nfound = C.c_uint64(0)
        self._check(self._lib.commet_search_reads(self._h, rs._h, _ptr(act), _ptr(found), C.byref(scanned),
                                                  C.byref(nfound)))
        return found, int(scanned.value), int(nfound.value)

    def index_and_search(self, index_rs, search_sets, index_select=None, search_selects=None):
        """The chunk loop on resident sets.  Returns (tags, stats, info):
        tags[i] = BooleanVector bytes of search set i, stats[i] = dict(indexed,
        searched, shared) — the numbers of the reference's log line."""
        ns = len(search_sets)
        isel = _as_bits(index_select, index_rs.num_reads, "index_select")
        ssel = [None] * ns
        if search_selects is not None:
            ssel = [_as_bits(s, rs.num_reads, "search_select") for s, rs in zip(search_selects, search_sets)]
        tags = [np.zeros(bits_nbytes(rs.num_reads), dtype=np.uint8) for rs in search_sets]
        rs_arr = (C.c_void_p * max(ns, 1))(*[rs._h for rs in search_sets])
        sel_arr = (C.c_void_p * max(ns, 1))(*[(_ptr(s).value if s is not None else None) for s in ssel])
        tag_arr = (C.c_void_p * max(ns, 1))(*[_ptr(t).value for t in tags])
        stats = (_l.PairStats * max(ns, 1))()
        info = _l.JobInfo()
        self._check(self._lib.commet_index_and_search(self._h, index_rs._h, _ptr(isel), ns, rs_arr, sel_arr, tag_arr,
                                                      stats, C.byref(info)))
        st = [dict(indexed=int(stats[i].indexed), searched=int(stats[i].searched), shared=int(stats[i].shared),
                   search_ms=float(stats[i].search_ms)) for i in range(ns)]
        inf = {f: getattr(info, f) for f, _ in _l.JobInfo._fields_}
        return tags, st, inf

    def index_many_and_search(self, index_sets, search_rs, index_selects=None, search_select=None):
        """commet_index_many_and_search: job j indexes index_sets[j] (restricted to index_selects[j]) and searches search_rs; the jobs
        share passes over the search set where the library can arrange it.  Returns (tags, stats, info): tags[j], stats[j] as
        index_and_search(index_sets[j], [search_rs], ...) gives them for job j alone."""
        nj = len(index_sets)
        isel = [None] * nj
        if index_selects is not None:
            isel = [_as_bits(s, rs.num_reads, "index_select") for s, rs in zip(index_selects, index_sets)]
        ssel = _as_bits(search_select, search_rs.num_reads, "search_select")
        tags = [np.zeros(bits_nbytes(search_rs.num_reads), dtype=np.uint8) for _ in range(nj)]
        rs_arr = (C.c_void_p * max(nj, 1))(*[rs._h for rs in index_sets])
        sel_arr = (C.c_void_p * max(nj, 1))(*[(_ptr(s).value if s is not None else None) for s in isel])
        tag_arr = (C.c_void_p * max(nj, 1))(*[_ptr(t).value for t in tags])
        stats = (_l.PairStats * max(nj, 1))()
        info = _l.JobInfo()
        self._check(self._lib.commet_index_many_and_search(self._h, nj, rs_arr, sel_arr, search_rs._h, _ptr(ssel), tag_arr, stats,
                                                           C.byref(info)))
        st = [dict(indexed=int(stats[i].indexed), searched=int(stats[i].searched), shared=int(stats[i].shared),
                   search_ms=float(stats[i].search_ms)) for i in range(nj)]
        return tags, st, {f: getattr(info, f) for f, _ in _l.JobInfo._fields_}

    def export_filter_reference(self):
        nbytes = int(2 ** (self.k - 1))
        out = np.zeros(max(nbytes, 1), dtype=np.uint8)
        self._check(self._lib.commet_filter_export_reference(self._h, _ptr(out), out.size))
        return out[:nbytes]

    def last_kernel_ms(self):
        i = C.c_double(0)
        s = C.c_double(0)
        self._check(self._lib.commet_last_kernel_ms(self._h, C.byref(i), C.byref(s)))
        return i.value, s.value

    def kernel_times(self):
        """{kernel: (launches, total_ms)} since set_option("kernel_timing", 1)"""
        arr = (_l.KernelTime * 64)()
        n = C.c_int(0)
        self._check(self._lib.commet_kernel_times(self._h, arr, 64, C.byref(n)))
        return {arr[i].name.decode(): (int(arr[i].launches), float(arr[i].total_ms)) for i in range(min(n.value, 64))}

    def cache_stats(self):
        """HBM held by the cached query lists of this context's read sets: dict(bytes, budget_bytes, evictions)"""
        b, g, e = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self._check(self._lib.commet_cache_stats(self._h, C.byref(b), C.byref(g), C.byref(e)))
        return dict(bytes=int(b.value), budget_bytes=int(g.value), evictions=int(e.value))

    def membench(self, atomic, table_bytes, n_access):
        ms = C.c_double(0)
        self._check(self._lib.commet_membench(self._h, int(atomic), int(table_bytes), int(n_access), C.byref(ms)))
        return ms.value

    def ldsbench(self, mode, n_words, n_access):
        ms = C.c_double(0)
        self._check(self._lib.commet_ldsbench(self._h, int(mode), int(n_words), int(n_access), C.byref(ms)))
        return ms.value


class ReadSet:
    """commet_readset: the reads of one set, packed and resident in HBM."""

    def __init__(self, ctx, max_reads, max_bases, _handle=None):
        self._ctx = ctx
        self._lib = ctx._lib
        self._h = _handle if _handle is not None else self._lib.commet_readset_create(ctx._h, int(max_reads), int(max_bases))
        if not self._h:
            raise CommetError(_err(self._lib))
        ctx._readsets.add(self)

    @classmethod
    def from_fasta(cls, ctx, paths):
        """Maps the FASTA files of one set (host parser in the library), streams them to HBM, finalizes."""
        arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        h = ctx._lib.commet_readset_from_fasta(ctx._h, arr, len(paths))
        if not h:
            raise CommetError(_err(ctx._lib))
        rs = cls(ctx, 0, 0, _handle=h)
        rs.finalize()
        return rs

    def save(self, path):
        """Writes the packed image of the set (commet_readset_save)."""
        self._check(self._lib.commet_readset_save(self._h, os.fsencode(path)))

    @classmethod
    def load(cls, ctx, path):
        """Loads a packed image written by save() — no parsing; finalizes."""
        h = ctx._lib.commet_readset_load(ctx._h, os.fsencode(path))
        if not h:
            raise CommetError(_err(ctx._lib))
        rs = cls(ctx, 0, 0, _handle=h)
        rs.finalize()
        return rs

    def export(self):
        """Descriptor + HIP IPC handles of the set's device buffers (bytes): another process of the node imports the set
        from it, device to device, without a file.  This set must stay alive until every importer has returned."""
        n = C.c_uint64(0)
        self._check(self._lib.commet_readset_export(self._h, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        self._check(self._lib.commet_readset_export(self._h, buf, n.value, C.byref(n)))
        return buf.raw[:n.value]

    @classmethod
    def import_(cls, ctx, blob):
        """A set exported by another process of the node (ReadSet.export), copied device to device; finalizes."""
        h = ctx._lib.commet_readset_import(ctx._h, blob, len(blob))
        if not h:
            raise CommetError(_err(ctx._lib))
        rs = cls(ctx, 0, 0, _handle=h)
        rs.finalize()
        return rs

    def file_reads(self):
        return [int(self._lib.commet_readset_file_reads(self._h, i)) for i in range(self.num_files)]

    @classmethod
    def from_files(cls, ctx, files):
        """files: list of (bases uint8[...], offsets uint64[n+1]) — one entry per file of the set."""
        nr = sum(len(o) - 1 for _, o in files)
        nb = sum(int(o[-1]) for _, o in files)
        rs = cls(ctx, nr, nb)
        for b, o in files:
            rs.add_file(b, o)
        rs.finalize()
        return rs

    @property
    def cache_bytes(self):
        """HBM held by data derived from the set and cached with it (the tiled search's query list)"""
        return int(self._lib.commet_readset_cache_bytes(self._h))

    def drop_cache(self):
        self._lib.commet_readset_drop_cache(self._h)

    def cache_estimate(self):
        """bytes of the set's query list for its context's (k, t); 0 = the set does not qualify for the tiled search"""
        return int(self._lib.commet_readset_cache_estimate(self._ctx._h, self._h))

    def reserve_cache(self):
        """commet_readset_reserve_cache: asks the driver NOW (from the calling thread) for the memory the set's query list will need, so
        that a list above the cap can be built later without an allocation on a job's path"""
        self._ctx._check(self._lib.commet_readset_reserve_cache(self._ctx._h, self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.commet_readset_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise CommetError(_err(self._lib))

    def add_file(self, bases, offsets):
        b = np.ascontiguousarray(bases, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        if o.size < 1 or int(o[-1]) > b.size:
            raise CommetError("offsets do not match bases")
        self._check(self._lib.commet_readset_begin_file(self._h))
        self._check(self._lib.commet_readset_append(self._h, _ptr(b), _ptr(o), o.size - 1))

    def add_file_staged(self, bases, offsets):
        """The same through the pinned staging API (commet_readset_stage_acquire / _commit): the caller's parser writes
        ASCII bases straight into pinned buffers, the device packs them (pack_reads_kernel)."""
        b = np.ascontiguousarray(bases, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = o.size - 1
        self._check(self._lib.commet_readset_begin_file(self._h))
        done = 0
        while done < n:
            hb, ho = _l.u8p(), _l.u64p()
            bcap, rcap = C.c_uint64(0), C.c_uint64(0)
            self._check(self._lib.commet_readset_stage_acquire(self._h, C.byref(hb), C.byref(bcap), C.byref(ho), C.byref(rcap)))
            b0 = int(o[done])
            take = 0
            while done + take < n and take < rcap.value and int(o[done + take + 1]) - b0 <= bcap.value:
                take += 1
            if take == 0:
                self._check(self._lib.commet_readset_stage_commit(self._h, 0))
                raise CommetError("a read does not fit the staging buffer")
            nb = int(o[done + take]) - b0
            C.memmove(hb, b[b0:b0 + nb].ctypes.data, nb)
            offs = (o[done:done + take + 1] - np.uint64(b0)).astype(np.uint64)
            C.memmove(ho, offs.ctypes.data, offs.size * 8)
            self._check(self._lib.commet_readset_stage_commit(self._h, take))
            done += take

    def finalize(self):
        self._check(self._lib.commet_readset_finalize(self._h))

    @property
    def num_reads(self):
        return int(self._lib.commet_readset_num_reads(self._h))

    @property
    def num_files(self):
        return int(self._lib.commet_readset_num_files(self._h))

    def kmer_counts(self):
        out = np.zeros(max(self.num_reads, 1), dtype=np.uint32)
        self._check(self._lib.commet_readset_kmer_counts(self._h, _ptr(out)))
        return out[:self.num_reads]

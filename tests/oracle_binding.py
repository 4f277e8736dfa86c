"""ctypes binding of oracle/_build/liboracle.so — the CPU checker.
TEST INFRASTRUCTURE ONLY: imported from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never from commet_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", "liboracle.so")
CLI = os.path.join(ORACLE_DIR, "_build", "oracle_cli")


class OkHash(C.Structure):
    _fields_ = [("a", C.c_uint64), ("b", C.c_uint64), ("c", C.c_uint64), ("d", C.c_uint64),
                ("mask", C.c_uint64), ("rv_mask", C.c_uint64), ("top", C.c_uint64), ("size", C.c_int)]


class OkBloom(C.Structure):
    _fields_ = [("vec", C.POINTER(C.c_uint8)), ("nbytes", C.c_uint64), ("probes", C.c_uint64)]


class OkSetResult(C.Structure):
    _fields_ = [("search_name", C.c_char * 256), ("indexed", C.c_uint64), ("searched", C.c_uint64),
                ("shared", C.c_uint64), ("probes", C.c_uint64)]


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True, stdout=subprocess.DEVNULL)
        lib = C.CDLL(LIB)
        lib.ok_bloom_new.restype = C.POINTER(OkBloom)
        lib.ok_bloom_new.argtypes = [C.c_int]
        lib.ok_bloom_free.argtypes = [C.POINTER(OkBloom)]
        lib.ok_max_kmer.restype = C.c_uint64
        lib.ok_max_kmer.argtypes = [C.c_int]
        lib.ok_set_max_kmer.restype = None
        lib.ok_set_max_kmer.argtypes = [C.c_uint64]
        lib.ok_index_batch.restype = C.c_uint64
        lib.ok_index_batch.argtypes = [C.POINTER(OkBloom), C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        lib.ok_search_batch.restype = C.c_uint64
        lib.ok_search_batch.argtypes = [C.POINTER(OkBloom), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64,
                                        C.c_void_p, C.c_void_p]
        lib.ok_keys_of_read.restype = C.c_uint64
        lib.ok_keys_of_read.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]
        lib.ok_bv_write.restype = C.c_int
        lib.ok_bv_write.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint64]
        lib.ok_bv_nb_one.restype = C.c_uint64
        lib.ok_bv_nb_one.argtypes = [C.c_void_p, C.c_uint64]
        lib.ok_index_and_search.restype = C.c_int
        lib.ok_index_and_search.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int,
                                            C.POINTER(OkSetResult), C.c_int, C.POINTER(C.c_int),
                                            C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]
        _lib = lib
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Bloom:
    """The reference's byte-array filter (bloom_filter.h) on the CPU."""

    def __init__(self, k):
        self.lib = load()
        self.k = k
        self.h = self.lib.ok_bloom_new(k)
        if not self.h:
            raise MemoryError("oracle filter allocation failed")

    def close(self):
        if self.h:
            self.lib.ok_bloom_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def bytes(self):
        n = int(self.h.contents.nbytes)
        return np.ctypeslib.as_array(self.h.contents.vec, shape=(max(n, 1),))[:n].copy()

    @property
    def probes(self):
        return int(self.h.contents.probes)

    def index(self, bases, offsets, select_bits=None):
        b = np.ascontiguousarray(bases, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        s = None if select_bits is None else np.ascontiguousarray(select_bits, dtype=np.uint8)
        return int(self.lib.ok_index_batch(self.h, self.k, _p(b), _p(o), o.size - 1, _p(s)))

    def search(self, t, bases, offsets, active_bits=None):
        b = np.ascontiguousarray(bases, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = o.size - 1
        a = None if active_bits is None else np.ascontiguousarray(active_bits, dtype=np.uint8)
        found = np.zeros(n // 8 + 1, dtype=np.uint8)
        nf = int(self.lib.ok_search_batch(self.h, self.k, t, _p(b), _p(o), n, _p(a), _p(found)))
        return found, nf


def keys_of_read(seq, k, reverse=False):
    lib = load()
    if isinstance(seq, str):
        seq = seq.encode()
    cap = max(len(seq), 1)
    keys = np.zeros((cap, 4), dtype=np.uint64)
    pos = np.zeros(cap, dtype=np.uint32)
    n = int(lib.ok_keys_of_read(seq, len(seq), k, int(reverse), _p(keys), _p(pos), cap))
    return keys[:n], pos[:n]


def kmer_counts(bases, offsets, k):
    """complete k-mers per read (what index_reads.h:55-57 feeds)"""
    b = np.asarray(bases, dtype=np.uint8).tobytes()
    out = np.zeros(len(offsets) - 1, dtype=np.uint32)
    for i in range(len(offsets) - 1):
        out[i] = len(keys_of_read(b[int(offsets[i]):int(offsets[i + 1])], k)[1])
    return out


def max_kmer(k):
    return int(load().ok_max_kmer(k))


def index_and_search(index_cfg, search_cfg, out_dir, log_dir, k, t, max_kmer=0):
    """Runs the restated tool in-process. Returns (rc, results, n_chunks, kmers).  max_kmer != 0: k-mers per chunk (the
    twin of the library's test hook `max_kmer`; 0 = the reference's constant)."""
    lib = load()
    lib.ok_set_max_kmer(int(max_kmer))
    res = (OkSetResult * 64)()
    n = C.c_int(0)
    chunks = C.c_uint64(0)
    kmers = C.c_uint64(0)
    rc = lib.ok_index_and_search(index_cfg.encode(), search_cfg.encode(), out_dir.encode(), log_dir.encode(), k, t,
                                 res, 64, C.byref(n), C.byref(chunks), C.byref(kmers), 1)
    lib.ok_set_max_kmer(0)
    out = [dict(name=res[i].search_name.decode(), indexed=int(res[i].indexed), searched=int(res[i].searched),
                shared=int(res[i].shared), probes=int(res[i].probes)) for i in range(n.value)]
    return rc, out, int(chunks.value), int(kmers.value)

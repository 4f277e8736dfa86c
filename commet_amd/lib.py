"""ctypes binding of include/commet_hip.h.

There is no CPU fallback: if libcommet_hip.so has not been built, or no HIP
device is usable, every entry point fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcommet_hip.so")

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f64p = C.POINTER(C.c_double)


class PairStats(C.Structure):
    _fields_ = [("indexed", C.c_uint64), ("searched", C.c_uint64), ("shared", C.c_uint64),
                ("search_ms", C.c_double)]


class JobInfo(C.Structure):
    _fields_ = [("n_chunks", C.c_uint64), ("kmers_indexed", C.c_uint64), ("reads_scanned", C.c_uint64),
                ("reads_indexed", C.c_uint64), ("index_launches", C.c_uint64), ("search_launches", C.c_uint64),
                ("probes", C.c_uint64), ("zero_ms", C.c_double), ("index_ms", C.c_double),
                ("index_kernel_ms", C.c_double), ("search_ms", C.c_double), ("total_ms", C.c_double)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double)]


# name -> (restype, argtypes); the test-suite checks that every symbol declared
# in include/commet_hip.h is exported by the library and listed here.
SIGNATURES = {
    "commet_version": (C.c_char_p, []),
    "commet_last_error": (C.c_char_p, []),
    "commet_device_count": (C.c_int, []),
    "commet_create": (C.c_void_p, [C.c_int, C.c_int, C.c_int]),
    "commet_destroy": (None, [C.c_void_p]),
    "commet_kmer_size": (C.c_int, [C.c_void_p]),
    "commet_min_hits": (C.c_int, [C.c_void_p]),
    "commet_max_kmer": (C.c_uint64, [C.c_void_p]),
    "commet_synchronize": (C.c_int, [C.c_void_p]),
    "commet_device_memory": (C.c_int, [C.c_void_p, u64p, u64p]),
    "commet_readset_create": (C.c_void_p, [C.c_void_p, C.c_uint64, C.c_uint64]),
    "commet_readset_destroy": (None, [C.c_void_p]),
    "commet_readset_begin_file": (C.c_int, [C.c_void_p]),
    "commet_readset_stage_acquire": (C.c_int, [C.c_void_p, C.POINTER(u8p), u64p, C.POINTER(u64p), u64p]),
    "commet_readset_stage_commit": (C.c_int, [C.c_void_p, C.c_uint64]),
    "commet_readset_append": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "commet_readset_from_fasta": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_char_p), C.c_int]),
    "commet_readset_from_buffers": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_char_p), u64p, C.c_int]),
    "commet_readset_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "commet_readset_load": (C.c_void_p, [C.c_void_p, C.c_char_p]),
    "commet_readset_export": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, u64p]),
    "commet_readset_import": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "commet_readset_file_reads": (C.c_uint64, [C.c_void_p, C.c_uint64]),
    "commet_readset_finalize": (C.c_int, [C.c_void_p]),
    "commet_readset_num_reads": (C.c_uint64, [C.c_void_p]),
    "commet_readset_num_files": (C.c_uint64, [C.c_void_p]),
    "commet_readset_kmer_counts": (C.c_int, [C.c_void_p, C.c_void_p]),
    "commet_readset_cache_bytes": (C.c_uint64, [C.c_void_p]),
    "commet_readset_drop_cache": (None, [C.c_void_p]),
    "commet_cache_stats": (C.c_int, [C.c_void_p, u64p, u64p, u64p]),
    "commet_readset_cache_estimate": (C.c_uint64, [C.c_void_p, C.c_void_p]),
    "commet_readset_reserve_cache": (C.c_int, [C.c_void_p, C.c_void_p]),
    "commet_device_cache_trim": (C.c_uint64, [C.c_int]),
    "commet_device_cache_bytes": (C.c_uint64, [C.c_int]),
    "commet_device_pooled_bytes": (C.c_uint64, [C.c_int]),
    "commet_device_alloc_stats": (C.c_int, [C.c_int, C.POINTER(C.c_double), u64p, u64p]),
    "commet_filter_reset": (C.c_int, [C.c_void_p]),
    "commet_index_reads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, u64p]),
    "commet_search_reads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, u64p, u64p]),
    "commet_index_and_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(PairStats),
                                          C.POINTER(JobInfo)]),
    "commet_index_many_and_search": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p,
                                               C.POINTER(C.c_void_p), C.POINTER(PairStats), C.POINTER(JobInfo)]),
    "commet_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "commet_filter_export_reference": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "commet_last_kernel_ms": (C.c_int, [C.c_void_p, f64p, f64p]),
    "commet_kernel_times": (C.c_int, [C.c_void_p, C.POINTER(KernelTime), C.c_int, C.POINTER(C.c_int)]),
    "commet_launched_kernels": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]),
    "commet_membench": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, f64p]),
    "commet_ldsbench": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, f64p]),
}

_lib = None


def load():
    """Loads the C-ABI library (once). Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m commet_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for this path.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_LOCAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib

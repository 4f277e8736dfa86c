// capi.hip — implementation of include/commet_hip.h on HIP (gfx950).
// Host side: context (filter slots, scratch of the bucketed index build), HBM
// residency of read sets, pinned multi-threaded ingest, exact chunk planning
// (read_iter.hpp) and the chunk loop of index_and_search on resident sets
// (chunks taken in groups, see search_group_kernel in kernels.hpp).
#include "../../include/commet_hip.h"

#include "kernels.hpp"
#include "index_part.hpp"
#include "slice_search.hpp"
#include "tile_search.hpp"
#include "read_iter.hpp"
#include "host/fasta_source.hpp"
#include "host/ingest_pack.hpp"

#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <atomic>
#include <memory>
#include <mutex>
#include <set>
#include <future>
#include <thread>
#include <string>
#include <vector>

using namespace commet;

namespace {

thread_local std::string g_err;

// Every kernel launch of the library notes its entry point here (the host-side handle hipLaunchKernel takes): the
// test-suite resolves the addresses against the library's symbol table and checks that every instantiation compiled
// into it was reached by a parity test (commet_launched_kernels, tests/test_gpu_zz_dispatch_coverage.py).
std::mutex g_launch_mu;
std::set<const void *> g_launched;
inline void note_launch(const void *entry)
{
    std::lock_guard<std::mutex> lk(g_launch_mu);
    g_launched.insert(entry);
}
#define COMMET_LAUNCH(kernel, ...)                    \
    do {                                               \
        note_launch((const void *) (kernel));          \
        hipLaunchKernelGGL(kernel, __VA_ARGS__);       \
    } while (0)

int fail(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

#define HIP_OK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define HIP_OK_NULL(expr)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);           \
            return nullptr;                                                                            \
        }                                                                                              \
    } while (0)

inline uint64_t bitmap_words(uint64_t n) { return n / 64 + 1; }
inline uint64_t bitmap_bytes_host(uint64_t n) { return n / 8 + 1; }   // boolean_vector.h:130

constexpr uint64_t STAGE_BASES = 64ull << 20;
constexpr uint64_t STAGE_READS = 1ull << 20;
constexpr int      N_COUNTERS = 8;

}  // namespace

struct commet_ctx {
    int device = 0;
    int k = 0, t = 0;
    hipStream_t stream = nullptr;
    uint32_t *filter = nullptr;       // 4 planes, contiguous
    uint64_t plane_words = 0;
    uint64_t filter_bytes = 0;
    unsigned long long *d_counters = nullptr;
    unsigned long long *h_counters = nullptr;   // pinned
    hipEvent_t ev_i0 = nullptr, ev_i1 = nullptr, ev_s0 = nullptr, ev_s1 = nullptr;
    bool have_index_ev = false, have_search_ev = false;
    bool count_probes = false;
    int index_mode = 0;               // 0 auto, 1 atomic kernel, 2 bucketed construction
    int part_b1 = 0;                  // override of the level-1 radix bits (0 = default split)
    int part_packed = 1;              // option: final buckets as groups of three 19-bit keys in 8 bytes (index_part.hpp)
    int part_no_uni = 0;              // option: never take the uniform-length fast path of hist / scatter1 (tests, A/B timing)
    int s2_swizzle = 128;             // scatter2 slab order: number of interleaved slab ranges (index_part.hpp), 0 = dispatch order
    uint64_t part_min_kmers = 8ull << 20;
    // workspaces of the bucketed construction (index_part.hpp): two, so that the chunks of a group can be built on two
    // streams at once (the compute-bound hist / scatter1 of one chunk overlap the HBM-bound scatter2 / build of another)
    struct PartWs {
        uint32_t *bufA = nullptr, *bufB = nullptr;
        uint64_t cap_keys = 0;
        uint32_t *hist = nullptr, *wl = nullptr;
        uint64_t *off = nullptr, *goff = nullptr;   // bucket offsets in keys / in 8-byte groups (packed final level)
        unsigned long long *cur2 = nullptr, *blockoff = nullptr;   // final-bucket cursors; scatter1 start positions [workgroup][coarse bucket]
        uint32_t *blockcnt = nullptr;                              // keys per [scatter1 workgroup][coarse bucket]
        uint32_t nb = 0;
        void release()
        {
            (void) hipFree(bufA); (void) hipFree(bufB); (void) hipFree(hist); (void) hipFree(wl); (void) hipFree(off); (void) hipFree(goff);
            (void) hipFree(cur2); (void) hipFree(blockoff); (void) hipFree(blockcnt);
            *this = PartWs();
        }
    } part[2];
    unsigned long long *d_jobcnt = nullptr;   // per (chunk, set) counters of commet_index_and_search, kept between calls
    uint32_t *d_ids = nullptr, *d_idblk = nullptr;   // read numbers of the index selection of the running job, in order (sel_ids_kernel)
    uint64_t ids_cap = 0, idblk_cap = 0;
    unsigned long long *d_plansum = nullptr;  // per-block k-mer sums of a selection (host planner input)
    uint64_t plansum_cap = 0;
    uint64_t jobcnt_cap = 0;
    hipStream_t aux_stream = nullptr;         // second lane of a chunk group's index phase
    hipStream_t load_stream = nullptr;        // everything that makes a read set (uploads, k-mer counts): a set may be loaded by one
                                              // host thread while another runs jobs on sets that are complete
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int index_lanes = 2;                      // option: 1 = build the chunks of a group one after the other

    int n_slots = 1;                  // filter slots allocated behind `filter` (chunk groups, kernels.hpp)
    int cur_slot = 0;                 // slot the index / search launch helpers work on
    uint32_t *il_a = nullptr;         // interleaved A planes of a chunk group
    int il_stride = 0;
    // option "kernel_timing": a hipEvent pair around every kernel launch of commet_index_and_search, on the stream the
    // kernel is launched on; per-kernel totals are read with commet_kernel_times (bench.py's roofline leg)
    struct KernelClock {
        struct Rec { const char *name; hipEvent_t a, b; };
        bool on = false;
        std::vector<Rec> open;                         // launches of the current call
        std::vector<hipEvent_t> spare;                 // events kept for the next call
        std::vector<std::string> names;                // totals, in first-seen order
        std::vector<uint64_t> launches;
        std::vector<double> total_ms;
        hipEvent_t get()
        {
            hipEvent_t e = nullptr;
            if (!spare.empty()) e = spare.back(), spare.pop_back();
            else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
            return e;
        }
        void collect()                                  // after the stream has been synchronised
        {
            for (Rec &r : open) {
                float ms = 0;
                if (r.a && r.b && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
                    size_t i = 0;
                    while (i < names.size() && names[i] != r.name) ++i;
                    if (i == names.size()) names.push_back(r.name), launches.push_back(0), total_ms.push_back(0);
                    launches[i] += 1;
                    total_ms[i] += ms;
                }
                if (r.a) spare.push_back(r.a);
                if (r.b) spare.push_back(r.b);
            }
            open.clear();
        }
        void reset() { names.clear(), launches.clear(), total_ms.clear(); }
        void release()
        {
            collect();
            for (hipEvent_t e : spare) (void) hipEventDestroy(e);
            spare.clear();
        }
    } kclock;
    // the many-small-chunks regime (slice_search.hpp): staging bit-planes, bit-sliced tables, chunk descriptors
    uint32_t *slice_stage = nullptr, *slice_tables = nullptr;
    SliceChunk *d_slice_chunks = nullptr;
    uint64_t slice_stage_words = 0, slice_table_words = 0, slice_chunks_cap = 0;
    uint8_t *d_qres = nullptr;        // tiled search (tile_search.hpp): one result byte per query record of the set being scanned
    uint64_t qres_cap = 0;
    int tiled_mode = 0;               // option "tiled_search": 0 auto (large sets, groups of 1 or 2 chunks), 1 never, 2 whenever possible
    // environment knobs of A/B runs, read ONCE in commet_create (nothing on the launch path calls getenv)
    int tq_sbits = 0;                 // COMMET_TQ_SBITS: log2 bits per address slice of the query list (0 = TQ_SBITS)
    int tq_parts = 1;                 // COMMET_TQ_PARTS / option "tq_parts": runs of pieces whose replay overlaps the next run's probe.  Off:
                                      // measured on configs[1] 19.76 ms per step in one part, 21.8 / 23.2 / 24.6 / 25.3 in 2 / 3 / 4 / 6 (the
                                      // replay of one part and the probe of the next contend for the same memory system, r03_parts_*.json)
    unsigned tq_wpx = 64;             // COMMET_TQ_WPX: probe workgroups per XCD (a multiple of the 32 CUs of an XCD keeps the sweep even;
                                      // measured: 32 or 64 (1 or 2 per CU) 2.3-2.6 ms, 128: 3.7, 256: 4.8)
    bool stage_reads = true;          // COMMET_NO_STAGE_READS: search_group_kernel without the LDS copy of the lanes' reads
    bool job_verbose = false;         // COMMET_JOB_VERBOSE: host-side phase times of every commet_index_and_search call on stderr
    bool ingest_verbose = false;      // COMMET_INGEST_VERBOSE
    int slice_mode = 0;               // option: 0 auto, 1 never, 2 whenever k allows it
    int slice_gw = 0;                 // option: words per bit-sliced entry (32 chunks each); 0 = by the number of chunks
    int slice_wide = 0;               // option: wide rows (all chunk filters side by side, slice_search.hpp): 0 auto (more than 256 chunks), 1 never, 2 whenever the regime applies
    uint32_t wide_cap_words = 0;      // option "slice_wide_words": at most this many words per row (tests: several passes); 0 = the budget decides
    uint32_t *wide_tables = nullptr;
    uint64_t wide_table_words = 0;
    uint64_t max_kmer_test = 0;       // option "max_kmer": chunk size override for tests (0 = the reference's constant)
    int chunk_group = 8;              // option: chunks searched per pass (1 = one pass per chunk; more than 4 only where group8_ok)
    // pinned / device staging buffers of the parallel host ingest, kept for the next read set (hipHostMalloc is slow)
    struct IngestBuf {
        uint32_t *h_planes = nullptr;
        uint64_t *h_goff = nullptr;
        hipEvent_t done = nullptr;
    };
    std::vector<IngestBuf> ingest_pool;

    // Derived data cached with the read sets (the tiled search's query lists, ~6 bytes per first-hit window: several times
    // the packed set itself) is accounted here and given back under pressure: least recently used lists first when the
    // budget is exceeded, every list that is not part of the running job when a device allocation fails.
    std::mutex ql_mu;                                 // guards the registry and every query list of the context
    std::vector<commet_readset *> sets;               // read sets alive on this context
    uint64_t ql_bytes = 0, ql_budget = 64ull << 30, ql_clock = 0, ql_evictions = 0;
    uint64_t ql_max_list = 4ull << 30;                // auto mode: sets whose list (8 bytes per first-hit window, estimated) is larger keep the gather kernels

    uint32_t *slot_ptr(int i) const { return filter + (uint64_t) i * 4 * plane_words; }
    FilterView view() const
    {
        uint32_t *base = slot_ptr(cur_slot);
        FilterView f;
        f.a = base;
        f.b = base + plane_words;
        f.c = base + 2 * plane_words;
        f.d = base + 3 * plane_words;
        return f;
    }
};

struct commet_readset {
    commet_ctx *ctx = nullptr;
    uint64_t max_reads = 0, max_bases = 0;
    uint64_t n_reads = 0, n_bases = 0;
    uint32_t *d_planes = nullptr;
    uint64_t *d_goff = nullptr;
    uint32_t *d_kcnt = nullptr;
    uint32_t *d_lenmm = nullptr;
    uint64_t *d_sel = nullptr, *d_tags = nullptr, *d_found = nullptr;   // bitmaps, bitmap_words(max_reads)
    struct Stage {
        uint8_t *h_bases = nullptr;
        uint64_t *h_offs = nullptr;
        uint8_t *d_bases = nullptr;
        uint64_t *d_offs = nullptr;
        hipEvent_t done = nullptr;
        bool inflight = false;
    } st[2];
    int cur = 0;
    bool acquired = false;
    uint64_t stage_bases = 0, stage_reads = 0;
    std::vector<FileSpan> files;
    std::vector<uint64_t> empty_reads;
    mutable std::vector<uint32_t> h_kcnt;      // host copy of d_kcnt, made on first use (host_counts)
    mutable std::vector<uint64_t> h_kprefix;   // prefix sums of h_kcnt (fast chunk planning)
    mutable bool have_host_counts = false;
    uint32_t uniform_len = 0;
    uint32_t max_kcnt = 0;
    uint32_t max_len = 0, min_len = 0;
    // query list of the tiled search (tile_search.hpp): the set's lane-a addresses sorted by address slice, made on first use
    struct QueryList {
        unsigned long long *d_tile_off = nullptr;
        uint32_t *d_qaddr = nullptr, *d_tstart = nullptr;
        uint16_t *d_qwho = nullptr, *d_tlen = nullptr;
        uint64_t n_records = 0;
        uint32_t n_slices = 0, n_pieces = 0;
        int sbits = 0;
        bool built = false, failed = false;
        uint64_t bytes = 0, last_use = 0;           // HBM held; the context's ql_clock at the last scan that used the list
        void release()
        {
            (void) hipFree(d_tile_off); (void) hipFree(d_qaddr); (void) hipFree(d_qwho); (void) hipFree(d_tstart); (void) hipFree(d_tlen);
            *this = QueryList();
        }
    };
    mutable QueryList ql;
    mutable bool in_job = false;                    // part of the commet_index_and_search call that is running: its list stays
    bool host_packed = false;                  // some reads were packed on the host (host/ingest_pack.hpp): counts come from kmer_counts_kernel
    uint32_t host_min_len = 0xFFFFFFFFu, host_max_len = 0;
    bool finalized = false;

    ReadsView view() const
    {
        ReadsView v;
        v.planes = d_planes;
        v.goff = d_goff;
        v.uniform_len = uniform_len;
        v.n = n_reads;
        return v;
    }
};

namespace {
// times one kernel launch when option "kernel_timing" is on (no-op otherwise)
struct KScope {
    commet_ctx::KernelClock &kc;
    hipStream_t stream;
    size_t idx = ~(size_t) 0;
    KScope(commet_ctx *c, const char *name, hipStream_t s) : kc(c->kclock), stream(s)
    {
        if (!kc.on) return;
        commet_ctx::KernelClock::Rec r{name, kc.get(), kc.get()};
        if (r.a) (void) hipEventRecord(r.a, stream);
        idx = kc.open.size();
        kc.open.push_back(r);
    }
    ~KScope()
    {
        if (idx != ~(size_t) 0 && kc.open[idx].b) (void) hipEventRecord(kc.open[idx].b, stream);
    }
};
}  // namespace

namespace {
// drops one set's query list (caller holds ql_mu); hipFree waits for the kernels that read it
void drop_query_list(commet_ctx *c, const commet_readset *rs)
{
    if (!rs->ql.built && !rs->ql.bytes) return;
    c->ql_bytes -= std::min(c->ql_bytes, rs->ql.bytes);
    rs->ql.release();
    ++c->ql_evictions;
}

// gives cached query lists back until at most `target` bytes of them are left: least recently used first, never a list
// of the running job unless `even_in_job` (the job thread itself is out of memory and holds no list between build and
// launch).  Returns the bytes released.  Caller holds ql_mu.
uint64_t shrink_query_lists(commet_ctx *c, uint64_t target, bool even_in_job)
{
    uint64_t freed = 0;
    while (c->ql_bytes > target) {
        const commet_readset *victim = nullptr;
        for (const commet_readset *rs : c->sets)
            if (rs->ql.built && (even_in_job || !rs->in_job) && (!victim || rs->ql.last_use < victim->ql.last_use)) victim = rs;
        if (!victim) break;
        freed += victim->ql.bytes;
        drop_query_list(c, victim);
    }
    return freed;
}

// hipMalloc that gives the cached query lists back and tries once more when the device is out of memory
hipError_t dev_alloc(commet_ctx *c, void **p, size_t bytes, bool job_thread)
{
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipErrorOutOfMemory) return e;
    (void) hipGetLastError();
    uint64_t freed;
    {
        std::lock_guard<std::mutex> lk(c->ql_mu);
        freed = shrink_query_lists(c, 0, job_thread);
    }
    if (!freed) return e;
    e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory) (void) hipGetLastError();
    return e;
}
}  // namespace

extern "C" {

const char *commet_version(void) { return "commet-amd 0.1 (gfx950)"; }
const char *commet_last_error(void) { return g_err.c_str(); }

int commet_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

commet_ctx *commet_create(int device, int kmer_size, int min_hits)
{
    if (kmer_size < 1 || kmer_size > 38) {
        fail("k-mer size %d out of range [1,38]", kmer_size);
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fail("no HIP device available: the index_and_search path has no CPU fallback");
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        fail("device %d out of range (have %d)", device, ndev);
        return nullptr;
    }
    HIP_OK_NULL(hipSetDevice(device));
    commet_ctx *c = new commet_ctx;
    c->device = device;
    c->k = kmer_size;
    c->t = min_hits < 1 ? 1 : min_hits;
    if (const char *e = getenv("COMMET_INDEX_LANES")) c->index_lanes = atoi(e) == 1 ? 1 : 2;   // 1: one kernel at a time (per-kernel profiles)
    if (const char *e = getenv("COMMET_TILED")) c->tiled_mode = std::max(0, std::min(2, atoi(e)));
    if (const char *e = getenv("COMMET_TQ_SBITS")) c->tq_sbits = atoi(e);
    if (const char *e = getenv("COMMET_TQ_WPX")) c->tq_wpx = (unsigned) std::max(1, atoi(e));
    if (const char *e = getenv("COMMET_TQ_PARTS")) c->tq_parts = std::max(1, std::min(16, atoi(e)));
    c->stage_reads = getenv("COMMET_NO_STAGE_READS") == nullptr;
    c->job_verbose = getenv("COMMET_JOB_VERBOSE") != nullptr;
    c->ingest_verbose = getenv("COMMET_INGEST_VERBOSE") != nullptr;
    if (const char *e = getenv("COMMET_QUERY_LIST_MAX_GB")) c->ql_max_list = (uint64_t) (std::max(0.0, atof(e)) * (double) (1ull << 30));
    if (const char *e = getenv("COMMET_QUERY_LIST_GB")) c->ql_budget = (uint64_t) (std::max(0.0, atof(e)) * (double) (1ull << 30));
    if (const char *e = getenv("COMMET_SLICE_MODE")) c->slice_mode = std::max(0, std::min(2, atoi(e)));     // A/B runs of bench.py
    if (const char *e = getenv("COMMET_SLICE_WIDE")) c->slice_wide = std::max(0, std::min(2, atoi(e)));     // A/B runs of bench.py
    if (const char *e = getenv("COMMET_SLICE_WORDS")) {
        const int v = atoi(e);
        if (v == 1 || v == 2 || v == 4 || v == 8) c->slice_gw = v;
    }
    // 2^k bits per plane, at least one word; 4 planes = 2^(k-1) bytes (bloom_filter.h:73)
    const uint64_t plane_bits = 1ull << kmer_size;
    c->plane_words = plane_bits < 32 ? 1 : plane_bits / 32;
    c->filter_bytes = 4 * c->plane_words * sizeof(uint32_t);
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **) &c->filter, c->filter_bytes);
    if (e != hipSuccess) {
        if (e == hipErrorOutOfMemory)
            fail("Index memory allocation impossible, try with a lower k value or with more RAM memory");
        else fail("context creation failed: %s", hipGetErrorString(e));
        commet_destroy(c);
        return nullptr;
    }
    e = hipMalloc((void **) &c->d_counters, N_COUNTERS * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipHostMalloc((void **) &c->h_counters, N_COUNTERS * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipEventCreate(&c->ev_i0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_i1);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_s0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_s1);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->load_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMemsetAsync(c->filter, 0, c->filter_bytes, c->stream);
    if (e != hipSuccess) {
        fail("context creation failed: %s", hipGetErrorString(e));
        commet_destroy(c);
        return nullptr;
    }
    return c;
}

void commet_destroy(commet_ctx *c)
{
    if (!c) return;
    (void) hipSetDevice(c->device);
    if (c->stream) (void) hipStreamSynchronize(c->stream);
    if (c->filter) (void) hipFree(c->filter);
    for (commet_ctx::IngestBuf &b : c->ingest_pool) {
        if (b.h_planes) (void) hipHostFree(b.h_planes);
        if (b.h_goff) (void) hipHostFree(b.h_goff);
        if (b.done) (void) hipEventDestroy(b.done);
    }
    c->kclock.release();
    (void) hipFree(c->d_qres);
    (void) hipFree(c->slice_stage);
    (void) hipFree(c->slice_tables);
    (void) hipFree(c->wide_tables);
    (void) hipFree(c->d_slice_chunks);
    (void) hipFree(c->il_a);
    (void) hipFree(c->d_jobcnt);
    (void) hipFree(c->d_plansum);
    (void) hipFree(c->d_ids);
    (void) hipFree(c->d_idblk);
    c->part[0].release();
    c->part[1].release();
    if (c->aux_stream) (void) hipStreamSynchronize(c->aux_stream), (void) hipStreamDestroy(c->aux_stream);
    if (c->load_stream) (void) hipStreamSynchronize(c->load_stream), (void) hipStreamDestroy(c->load_stream);
    if (c->ev_fork) (void) hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void) hipEventDestroy(c->ev_join);
    if (c->d_counters) (void) hipFree(c->d_counters);
    if (c->h_counters) (void) hipHostFree(c->h_counters);
    if (c->ev_i0) (void) hipEventDestroy(c->ev_i0);
    if (c->ev_i1) (void) hipEventDestroy(c->ev_i1);
    if (c->ev_s0) (void) hipEventDestroy(c->ev_s0);
    if (c->ev_s1) (void) hipEventDestroy(c->ev_s1);
    if (c->stream) (void) hipStreamDestroy(c->stream);
    delete c;
}

int commet_kmer_size(const commet_ctx *c) { return c->k; }
int commet_min_hits(const commet_ctx *c) { return c->t; }

uint64_t commet_max_kmer(const commet_ctx *c)
{
    if (c->max_kmer_test) return c->max_kmer_test;          // test hook, see commet_set_option
    return (uint64_t) (1000000000.0 / pow(2, 33 - c->k));   // index_and_search.cpp:73,146
}

int commet_device_memory(const commet_ctx *c, uint64_t *free_bytes, uint64_t *total_bytes)
{
    HIP_OK(hipSetDevice(c->device));
    size_t f = 0, t = 0;
    HIP_OK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return 0;
}

int commet_synchronize(commet_ctx *c)
{
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

/* ---- read sets ------------------------------------------------------------ */

commet_readset *commet_readset_create(commet_ctx *c, uint64_t max_reads, uint64_t max_bases)
{
    if (!c) {
        fail("null context");
        return nullptr;
    }
    HIP_OK_NULL(hipSetDevice(c->device));
    commet_readset *rs = new commet_readset;
    rs->ctx = c;
    rs->max_reads = max_reads;
    rs->max_bases = max_bases;
    rs->stage_bases = max_bases < STAGE_BASES ? (max_bases ? max_bases : 1) : STAGE_BASES;
    rs->stage_reads = max_reads < STAGE_READS ? (max_reads ? max_reads : 1) : STAGE_READS;
    const uint64_t triples = (max_bases >> 5) + max_reads + 1;
    const uint64_t bw = bitmap_words(max_reads);
    // (a set may be made by a second host thread while a job runs: that thread never takes a list of the running job)
    hipError_t e = dev_alloc(c, (void **) &rs->d_planes, triples * 3 * sizeof(uint32_t), false);
    if (e == hipSuccess) e = dev_alloc(c, (void **) &rs->d_goff, (max_reads + 1) * sizeof(uint64_t), false);
    if (e == hipSuccess) e = dev_alloc(c, (void **) &rs->d_kcnt, (max_reads + 1) * sizeof(uint32_t), false);
    if (e == hipSuccess) e = dev_alloc(c, (void **) &rs->d_lenmm, 3 * sizeof(uint32_t), false);
    if (e == hipSuccess) e = dev_alloc(c, (void **) &rs->d_sel, bw * 8, false);
    if (e == hipSuccess) e = dev_alloc(c, (void **) &rs->d_tags, bw * 8, false);
    if (e == hipSuccess) e = dev_alloc(c, (void **) &rs->d_found, bw * 8, false);
    // the gap triples between reads and the closing triple are never written by the host packer when the neighbours
    // come from different staging buffers; no kernel reads them, but a packed image (commet_readset_save) carries them
    if (e == hipSuccess) e = hipMemsetAsync(rs->d_planes, 0, triples * 3 * sizeof(uint32_t), c->load_stream);
    if (e == hipSuccess) {
        const uint32_t mm[3] = {0xFFFFFFFFu, 0u, 0u};
        e = hipMemcpyAsync(rs->d_lenmm, mm, sizeof mm, hipMemcpyHostToDevice, c->load_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->load_stream);
    }
    if (e != hipSuccess) {
        fail("read set allocation failed (%llu reads, %llu bases): %s", (unsigned long long) max_reads,
             (unsigned long long) max_bases, hipGetErrorString(e));
        commet_readset_destroy(rs);
        return nullptr;
    }
    {
        std::lock_guard<std::mutex> lk(c->ql_mu);
        c->sets.push_back(rs);
    }
    return rs;
}

void commet_readset_destroy(commet_readset *rs)
{
    if (!rs) return;
    (void) hipSetDevice(rs->ctx->device);
    (void) hipStreamSynchronize(rs->ctx->load_stream);
    (void) hipStreamSynchronize(rs->ctx->stream);   // (a job that still reads the set)
    {
        commet_ctx *c = rs->ctx;
        std::lock_guard<std::mutex> lk(c->ql_mu);
        c->ql_bytes -= std::min(c->ql_bytes, rs->ql.bytes);
        c->sets.erase(std::remove(c->sets.begin(), c->sets.end(), rs), c->sets.end());
    }
    (void) hipFree(rs->d_planes);
    (void) hipFree(rs->d_goff);
    (void) hipFree(rs->d_kcnt);
    (void) hipFree(rs->d_lenmm);
    (void) hipFree(rs->d_sel);
    (void) hipFree(rs->d_tags);
    (void) hipFree(rs->d_found);
    rs->ql.release();
    for (int i = 0; i < 2; ++i) {
        if (rs->st[i].h_bases) (void) hipHostFree(rs->st[i].h_bases);
        if (rs->st[i].h_offs) (void) hipHostFree(rs->st[i].h_offs);
        (void) hipFree(rs->st[i].d_bases);
        (void) hipFree(rs->st[i].d_offs);
        if (rs->st[i].done) (void) hipEventDestroy(rs->st[i].done);
    }
    delete rs;
}

int commet_readset_begin_file(commet_readset *rs)
{
    if (rs->finalized) return fail("read set already finalized");
    if (rs->acquired) return fail("begin_file with an uncommitted staging buffer");
    rs->files.push_back(FileSpan{rs->n_reads, 0});
    return 0;
}

int commet_readset_stage_acquire(commet_readset *rs, uint8_t **bases, uint64_t *bases_cap, uint64_t **offsets,
                                 uint64_t *reads_cap)
{
    if (rs->finalized) return fail("read set already finalized");
    if (rs->files.empty()) return fail("commet_readset_begin_file must be called first");
    if (rs->acquired) return fail("staging buffer already acquired");
    HIP_OK(hipSetDevice(rs->ctx->device));
    commet_readset::Stage &s = rs->st[rs->cur];
    if (!s.h_bases) {   // staging buffers are created on first use (commet_readset_from_fasta has its own)
        HIP_OK(hipHostMalloc((void **) &s.h_bases, rs->stage_bases));
        HIP_OK(hipHostMalloc((void **) &s.h_offs, (rs->stage_reads + 1) * sizeof(uint64_t)));
        HIP_OK(hipMalloc((void **) &s.d_bases, rs->stage_bases));
        HIP_OK(hipMalloc((void **) &s.d_offs, (rs->stage_reads + 1) * sizeof(uint64_t)));
        HIP_OK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
    if (s.inflight) {
        HIP_OK(hipEventSynchronize(s.done));
        s.inflight = false;
    }
    *bases = s.h_bases;
    *bases_cap = rs->stage_bases;
    *offsets = s.h_offs;
    *reads_cap = rs->stage_reads;
    rs->acquired = true;
    return 0;
}

int commet_readset_stage_commit(commet_readset *rs, uint64_t n)
{
    if (!rs->acquired) return fail("commit without acquire");
    rs->acquired = false;
    if (n == 0) return 0;
    commet_readset::Stage &s = rs->st[rs->cur];
    if (n > rs->stage_reads) return fail("too many reads in one staging batch");
    if (s.h_offs[0] != 0) return fail("offsets[0] must be 0");
    const uint64_t nbases = s.h_offs[n];
    if (nbases > rs->stage_bases) return fail("staging batch overflows its base buffer");
    if (rs->n_reads + n > rs->max_reads || rs->n_bases + nbases > rs->max_bases)
        return fail("read set capacity exceeded (%llu reads / %llu bases reserved)", (unsigned long long) rs->max_reads,
                    (unsigned long long) rs->max_bases);
    for (uint64_t i = 0; i < n; ++i) {
        if (s.h_offs[i + 1] < s.h_offs[i]) return fail("offsets must be non-decreasing");
        if (s.h_offs[i + 1] - s.h_offs[i] > 0x7FFFFFFFull) return fail("read longer than 2^31-1 bases");
        if (s.h_offs[i + 1] == s.h_offs[i]) rs->empty_reads.push_back(rs->n_reads + i);
    }
    commet_ctx *c = rs->ctx;
    HIP_OK(hipSetDevice(c->device));
    if (nbases) HIP_OK(hipMemcpyAsync(s.d_bases, s.h_bases, nbases, hipMemcpyHostToDevice, c->load_stream));
    HIP_OK(hipMemcpyAsync(s.d_offs, s.h_offs, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->load_stream));
    const unsigned grid = (unsigned) ((n + 1 + 255) / 256);
    COMMET_LAUNCH(pack_reads_kernel, dim3(grid), dim3(256), 0, c->load_stream, s.d_bases, s.d_offs, n, rs->n_reads,
                       rs->n_bases, rs->d_planes, rs->d_goff, rs->d_kcnt, rs->d_lenmm, c->k);
    HIP_OK(hipGetLastError());
    HIP_OK(hipEventRecord(s.done, c->load_stream));
    s.inflight = true;
    rs->n_reads += n;
    rs->n_bases += nbases;
    rs->files.back().count += n;
    rs->cur ^= 1;
    return 0;
}

}  // extern "C"

namespace {

// ---- host ingest: records are 2-bit packed by the ingest threads (host/ingest_pack.hpp) and uploaded as planes ----
#ifndef INGEST_STAGE_KIB
#define INGEST_STAGE_KIB 3072
#endif
#ifndef INGEST_STAGE_READS_LOG2
#define INGEST_STAGE_READS_LOG2 17
#endif
constexpr uint64_t INGEST_STAGE_BYTES = (uint64_t) INGEST_STAGE_KIB << 10;       // one pinned staging buffer of planes (12 bytes per triple): 8 M bases;
                                                          // small, because pinning memory costs ~0.2 ms per MiB on first use
constexpr uint64_t INGEST_STAGE_READS = 1ull << INGEST_STAGE_READS_LOG2;       // base offsets per staging buffer

// the upload side of host/ingest_pack.hpp: two pinned staging buffers per worker out of the context's pool; a flush
// queues hipMemcpyAsync of the planes (and the reads' base offsets) straight to their final place in the read set
struct HipPackSink {
    commet_readset *rs = nullptr;
    std::vector<int> cur;                    // which of its two buffers a worker fills next
    std::vector<char> inflight;              // per pool buffer

    bool prepare(commet_readset *set, int workers)
    {
        rs = set;
        commet_ctx *c = rs->ctx;
        if (hipSetDevice(c->device) != hipSuccess) return false;
        // the pool's entries exist up front (workers never resize it); their pinned memory is made by the worker that
        // first needs it, in acquire(): pinning costs ~0.2 ms per MiB, and paid here, on one thread before any packing,
        // it was 57 ms of the first set's 107 ms
        if (c->ingest_pool.size() < (size_t) workers * 2) c->ingest_pool.resize((size_t) workers * 2);
        cur.assign(workers, 0);
        inflight.assign((size_t) workers * 2, 0);
        return true;
    }
    bool acquire(int worker, commet_host::PackStage &st)
    {
        const size_t bi = (size_t) worker * 2 + cur[worker];
        commet_ctx::IngestBuf &b = rs->ctx->ingest_pool[bi];
        if (!b.done) {   // hipHostMalloc is slow: buffers stay with the context
            if (hipSetDevice(rs->ctx->device) != hipSuccess) return false;
            const bool ok = hipHostMalloc((void **) &b.h_planes, INGEST_STAGE_BYTES) == hipSuccess &&
                            hipHostMalloc((void **) &b.h_goff, INGEST_STAGE_READS * sizeof(uint64_t)) == hipSuccess &&
                            hipEventCreateWithFlags(&b.done, hipEventDisableTiming) == hipSuccess;
            if (!ok) {   // a half-made entry must not stay
                if (b.h_planes) (void) hipHostFree(b.h_planes);
                if (b.h_goff) (void) hipHostFree(b.h_goff);
                if (b.done) (void) hipEventDestroy(b.done);
                b = commet_ctx::IngestBuf();
                (void) hipGetLastError();
                return false;
            }
        }
        if (inflight[bi]) {
            if (hipEventSynchronize(b.done) != hipSuccess) return false;
            inflight[bi] = 0;
        }
        st.planes = b.h_planes;
        st.goff = b.h_goff;
        st.cap_triples = INGEST_STAGE_BYTES / 12;
        st.cap_reads = INGEST_STAGE_READS;
        return true;
    }
    bool flush(int worker, const commet_host::PackStage &st, uint64_t triple0, uint64_t n_triples, uint64_t read0, uint64_t n_reads)
    {
        commet_ctx *c = rs->ctx;
        const size_t bi = (size_t) worker * 2 + cur[worker];
        if (triple0 + n_triples > (rs->max_bases >> 5) + rs->max_reads + 1 || read0 + n_reads > rs->max_reads) return false;
        if (hipSetDevice(c->device) != hipSuccess) return false;
        if (n_triples && hipMemcpyAsync(rs->d_planes + 3 * triple0, st.planes, n_triples * 12, hipMemcpyHostToDevice, c->load_stream) != hipSuccess) return false;
        if (n_reads && hipMemcpyAsync(rs->d_goff + read0, st.goff, n_reads * sizeof(uint64_t), hipMemcpyHostToDevice, c->load_stream) != hipSuccess) return false;
        if (hipEventRecord(c->ingest_pool[bi].done, c->load_stream) != hipSuccess) return false;
        inflight[bi] = 1;
        cur[worker] ^= 1;
        return true;
    }
};

void absorb_summary(commet_readset *rs, const commet_host::PackSummary &sm)
{
    rs->host_packed = true;
    rs->host_min_len = std::min(rs->host_min_len, sm.min_len);
    rs->host_max_len = std::max(rs->host_max_len, sm.max_len);
    rs->empty_reads.insert(rs->empty_reads.end(), sm.empty_reads.begin(), sm.empty_reads.end());
}

}  // namespace

extern "C" {

int commet_readset_append(commet_readset *rs, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads)
{
    if (rs->finalized) return fail("read set already finalized");
    if (rs->files.empty()) return fail("commet_readset_begin_file must be called first");
    if (rs->acquired) return fail("append with an uncommitted staging buffer");
    if (n_reads == 0) return 0;
    if (offsets[0] != 0) return fail("offsets[0] must be 0");
    const uint64_t nbases = offsets[n_reads];
    if (rs->n_reads + n_reads > rs->max_reads || rs->n_bases + nbases > rs->max_bases)
        return fail("read set capacity exceeded (%llu reads / %llu bases reserved)", (unsigned long long) rs->max_reads,
                    (unsigned long long) rs->max_bases);
    HipPackSink sink;
    const int T = commet_host::ingest_threads();
    const bool verbose = rs->ctx->ingest_verbose;
    const auto tv0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tv0).count(); };
    if (!sink.prepare(rs, T)) return fail("cannot allocate the ingest staging buffers");
    if (verbose) fprintf(stderr, "[ingest] staging ready      %8.1f ms\n", since());
    commet_host::PackSummary sm;
    std::string err;
    const bool ok = commet_host::ingest_arrays(bases, offsets, n_reads, rs->n_reads, rs->n_bases, T, sink, sm, err);
    if (verbose) fprintf(stderr, "[ingest] packed + queued    %8.1f ms\n", since());
    // the staging buffers go back to the pool only once their copies are done
    const bool synced = hipStreamSynchronize(rs->ctx->load_stream) == hipSuccess;
    if (verbose) fprintf(stderr, "[ingest] uploaded           %8.1f ms\n", since());
    if (!synced && ok) return fail("upload failed: %s", hipGetErrorString(hipGetLastError()));
    if (!ok) return fail("%s", err.empty() ? "read set ingest failed" : err.c_str());
    absorb_summary(rs, sm);
    rs->n_reads += n_reads;
    rs->n_bases += nbases;
    rs->files.back().count += n_reads;
    return 0;
}

commet_readset *commet_readset_from_fasta(commet_ctx *c, const char *const *paths, int n_paths)
{
    std::vector<std::unique_ptr<commet_host::ReadFileData>> maps;
    std::vector<const char *> data;
    std::vector<uint64_t> sizes;
    // one thread per file maps it or, when gzipped, inflates it (a zlib stream is sequential; files are independent)
    std::vector<std::future<std::unique_ptr<commet_host::ReadFileData>>> opening;
    for (int i = 0; i < n_paths; ++i) {
        const std::string path = paths[i];
        opening.push_back(std::async(std::launch::async, [path]() {
            std::unique_ptr<commet_host::ReadFileData> f(new commet_host::ReadFileData);
            if (!f->open_file(path)) f.reset();
            return f;
        }));
    }
    for (int i = 0; i < n_paths; ++i) {
        std::unique_ptr<commet_host::ReadFileData> mf = opening[i].get();
        if (!mf) {
            fail("Cannot open file %s", paths[i]);
            return nullptr;
        }
        if (mf->format() == commet_host::ReadFormat::Unknown) {
            fail("Unknown format: %s", paths[i]);
            return nullptr;
        }
        data.push_back(mf->data());
        sizes.push_back(mf->size());
        maps.push_back(std::move(mf));
    }
    return commet_readset_from_buffers(c, data.data(), sizes.data(), n_paths);
}

commet_readset *commet_readset_from_buffers(commet_ctx *c, const char *const *data, const uint64_t *sizes, int n_paths)
{
    std::vector<const char *> d(data, data + n_paths);
    std::vector<size_t> n(sizes, sizes + n_paths);
    std::vector<commet_host::ReadFormat> fmts;
    for (int i = 0; i < n_paths; ++i) {
        fmts.push_back(commet_host::sniff_format(data[i], (size_t) sizes[i]));
        if (fmts.back() == commet_host::ReadFormat::Unknown) {
            fail("Unknown format: file %d of the set is neither FASTA nor FASTQ text", i);
            return nullptr;
        }
    }
    const bool verbose = c->ingest_verbose;
    const auto tv0 = std::chrono::steady_clock::now();
    commet_readset *rs = nullptr;
    HipPackSink sink;
    std::vector<uint64_t> file_reads;
    uint64_t total_reads = 0, total_bases = 0;
    commet_host::PackSummary sm;
    std::string err;
    const bool ok = commet_host::ingest_files<HipPackSink>(
        d, n, fmts, commet_host::ingest_threads(),
        [&](uint64_t reads, uint64_t bases, int workers) -> HipPackSink * {
            if (verbose)
                fprintf(stderr, "[ingest] counted            %8.1f ms\n",
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tv0).count());
            rs = commet_readset_create(c, reads, bases);
            if (!rs || !sink.prepare(rs, workers)) return nullptr;
            return &sink;
        },
        file_reads, total_reads, total_bases, sm, err);
    if (rs) (void) hipStreamSynchronize(c->load_stream);
    if (verbose)
        fprintf(stderr, "[ingest] packed + uploaded  %8.1f ms\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tv0).count());
    if (!ok) {
        if (rs) fail("%s", err.empty() ? "read set ingest failed" : err.c_str());   // (else the failing call has set the message)
        if (rs) commet_readset_destroy(rs);
        return nullptr;
    }
    uint64_t pos = 0;
    for (int f = 0; f < n_paths; ++f) {
        rs->files.push_back(FileSpan{pos, file_reads[f]});
        pos += file_reads[f];
    }
    rs->n_reads = total_reads;
    rs->n_bases = total_bases;
    absorb_summary(rs, sm);
    return rs;
}


uint64_t commet_readset_file_reads(const commet_readset *rs, uint64_t file_index)
{
    return file_index < rs->files.size() ? rs->files[file_index].count : 0;
}

// host copy of the per-read k-mer counts + prefix sums, on first need (a set that is only searched never needs them)
static int host_counts(const commet_readset *rs)
{
    if (rs->have_host_counts) return 0;
    HIP_OK(hipSetDevice(rs->ctx->device));
    rs->h_kcnt.resize(rs->n_reads);
    if (rs->n_reads) HIP_OK(hipMemcpy(rs->h_kcnt.data(), rs->d_kcnt, rs->n_reads * sizeof(uint32_t), hipMemcpyDeviceToHost));
    build_kmer_prefix(rs->h_kcnt.data(), rs->n_reads, rs->h_kprefix);
    rs->have_host_counts = true;
    return 0;
}

int commet_readset_finalize(commet_readset *rs)
{
    if (rs->finalized) return 0;
    if (rs->acquired) return fail("finalize with an uncommitted staging buffer");
    commet_ctx *c = rs->ctx;
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipStreamSynchronize(c->load_stream));
    rs->st[0].inflight = rs->st[1].inflight = false;
    // shortest / longest read: from the packing kernel (reads that came through the staging API) and from the host
    // packer (append / from_fasta); the host copy of the per-read counts and their prefix sums (chunk planning) are made
    // when the set is first used as an index set (host_counts)
    uint32_t mm[3] = {0xFFFFFFFFu, 0, 0};
    if (rs->n_reads) HIP_OK(hipMemcpy(mm, rs->d_lenmm, sizeof mm, hipMemcpyDeviceToHost));
    mm[0] = std::min(mm[0], rs->host_min_len);
    mm[1] = std::max(mm[1], rs->host_max_len);
    rs->uniform_len = (rs->n_reads && mm[0] == mm[1] && mm[0] != 0) ? mm[0] : 0;
    rs->max_len = rs->n_reads ? mm[1] : 0;
    rs->min_len = rs->n_reads ? mm[0] : 0;
    if (rs->host_packed && rs->n_reads) {
        // host-packed reads have no counts yet: complete k-mers of every read from its validity plane, on the device
        const uint64_t nb = rs->n_bases;
        HIP_OK(hipMemcpyAsync(rs->d_goff + rs->n_reads, &nb, sizeof nb, hipMemcpyHostToDevice, c->load_stream));   // closes the offsets
        COMMET_LAUNCH(kmer_counts_kernel, dim3((unsigned) ((rs->n_reads + 255) / 256)), dim3(256), 0, c->load_stream, rs->view(), c->k,
                           rs->d_kcnt, rs->d_lenmm);
        HIP_OK(hipGetLastError());
        HIP_OK(hipStreamSynchronize(c->load_stream));
        HIP_OK(hipMemcpy(mm, rs->d_lenmm, sizeof mm, hipMemcpyDeviceToHost));
    }
    rs->max_kcnt = rs->n_reads ? mm[2] : 0;
    std::sort(rs->empty_reads.begin(), rs->empty_reads.end());
    // the staging buffers are no longer needed: give the memory back
    for (int i = 0; i < 2; ++i) {
        if (rs->st[i].h_bases) (void) hipHostFree(rs->st[i].h_bases);
        if (rs->st[i].h_offs) (void) hipHostFree(rs->st[i].h_offs);
        (void) hipFree(rs->st[i].d_bases);
        (void) hipFree(rs->st[i].d_offs);
        rs->st[i].h_bases = nullptr;
        rs->st[i].h_offs = nullptr;
        rs->st[i].d_bases = nullptr;
        rs->st[i].d_offs = nullptr;
    }
    rs->finalized = true;
    return 0;
}

/* ---- packed images of a read set (k-independent): parse once, load everywhere ---------------------------------- */
namespace {
struct PackHeader {
    char     magic[8];          // "CMTPK01"
    uint64_t n_reads, n_bases, triples, n_files, n_empty;
    uint32_t uniform_len, min_len, max_len, pad;
};
inline uint64_t align64(uint64_t x) { return (x + 63) & ~63ull; }
struct PackLayout {
    uint64_t files_at, empty_at, planes_at, goff_at, total;
    PackLayout(const PackHeader &h)
    {
        files_at = align64(sizeof(PackHeader));
        empty_at = files_at + h.n_files * sizeof(FileSpan);
        planes_at = align64(empty_at + h.n_empty * 8);
        goff_at = align64(planes_at + h.triples * 12);
        total = goff_at + (h.uniform_len ? 0 : (h.n_reads + 1) * 8);
    }
};
}  // namespace

int commet_readset_save(const commet_readset *rs, const char *path)
{
    if (!rs->finalized) return fail("read set not finalized");
    commet_ctx *c = rs->ctx;
    HIP_OK(hipSetDevice(c->device));
    PackHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "CMTPK01", 8);
    h.n_reads = rs->n_reads, h.n_bases = rs->n_bases, h.triples = (rs->n_bases >> 5) + rs->n_reads + 1;
    h.n_files = rs->files.size(), h.n_empty = rs->empty_reads.size();
    h.uniform_len = rs->uniform_len, h.min_len = rs->min_len, h.max_len = rs->max_len;
    const PackLayout lay(h);
    const std::string tmp = std::string(path) + ".tmp";
    (void) unlink(tmp.c_str());                                   // (what an interrupted save may have left)
    const int fd = open(tmp.c_str(), O_RDWR | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);   // never through a link somebody else planted
    if (fd < 0) return fail("cannot create %s: %s", tmp.c_str(), strerror(errno));
    if (ftruncate(fd, (off_t) lay.total) != 0) {
        close(fd);
        return fail("cannot size %s to %llu bytes: %s", tmp.c_str(), (unsigned long long) lay.total, strerror(errno));
    }
    uint8_t *m = (uint8_t *) mmap(nullptr, lay.total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return fail("cannot map %s: %s", tmp.c_str(), strerror(errno));
    memcpy(m, &h, sizeof h);
    if (h.n_files) memcpy(m + lay.files_at, rs->files.data(), h.n_files * sizeof(FileSpan));
    if (h.n_empty) memcpy(m + lay.empty_at, rs->empty_reads.data(), h.n_empty * 8);
    hipError_t e = hipStreamSynchronize(c->load_stream);
    if (e == hipSuccess) e = hipMemcpy(m + lay.planes_at, rs->d_planes, h.triples * 12, hipMemcpyDeviceToHost);
    if (e == hipSuccess && !h.uniform_len) e = hipMemcpy(m + lay.goff_at, rs->d_goff, (h.n_reads + 1) * 8, hipMemcpyDeviceToHost);
    munmap(m, lay.total);
    if (e != hipSuccess) {
        unlink(tmp.c_str());
        return fail("read set download failed: %s", hipGetErrorString(e));
    }
    if (rename(tmp.c_str(), path) != 0) return fail("cannot rename %s: %s", tmp.c_str(), strerror(errno));
    return 0;
}

commet_readset *commet_readset_load(commet_ctx *c, const char *path)
{
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        fail("cannot open %s: %s", path, strerror(errno));
        return nullptr;
    }
    struct stat st;
    PackHeader h;
    if (fstat(fd, &st) != 0 || (uint64_t) st.st_size < sizeof h || pread(fd, &h, sizeof h, 0) != (ssize_t) sizeof h ||
        memcmp(h.magic, "CMTPK01", 8) != 0) {
        close(fd);
        fail("%s is not a packed read set", path);
        return nullptr;
    }
    // the counts are bounded by the file's own size before any arithmetic is done with them
    const uint64_t fsz = (uint64_t) st.st_size;
    if (h.n_files > fsz / sizeof(FileSpan) || h.n_empty > fsz / 8 || h.triples > fsz / 12 || h.n_reads > h.triples || (h.n_bases >> 5) > h.triples) {
        close(fd);
        fail("%s: inconsistent packed read set", path);
        return nullptr;
    }
    const PackLayout lay(h);
    if (h.triples != (h.n_bases >> 5) + h.n_reads + 1 || lay.total != (uint64_t) st.st_size) {
        close(fd);
        fail("%s: inconsistent packed read set", path);
        return nullptr;
    }
    const uint8_t *m = (const uint8_t *) mmap(nullptr, lay.total, PROT_READ, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
        fail("cannot map %s: %s", path, strerror(errno));
        return nullptr;
    }
    commet_readset *rs = commet_readset_create(c, h.n_reads, h.n_bases);
    if (!rs) {
        munmap((void *) m, lay.total);
        return nullptr;
    }
    const FileSpan *fs = (const FileSpan *) (m + lay.files_at);
    rs->files.assign(fs, fs + h.n_files);
    const uint64_t *er = (const uint64_t *) (m + lay.empty_at);
    rs->empty_reads.assign(er, er + h.n_empty);
    rs->n_reads = h.n_reads;
    rs->n_bases = h.n_bases;
    const uint32_t mm[3] = {h.n_reads ? h.min_len : 0xFFFFFFFFu, h.max_len, 0u};
    hipError_t e = hipMemcpyAsync(rs->d_lenmm, mm, sizeof mm, hipMemcpyHostToDevice, c->load_stream);
    // the image is pageable memory: the copies below are staged by the runtime and return when the source has been read
    if (e == hipSuccess) e = hipMemcpyAsync(rs->d_planes, m + lay.planes_at, h.triples * 12, hipMemcpyHostToDevice, c->load_stream);
    if (e == hipSuccess && !h.uniform_len)
        e = hipMemcpyAsync(rs->d_goff, m + lay.goff_at, (h.n_reads + 1) * 8, hipMemcpyHostToDevice, c->load_stream);
    if (e == hipSuccess && h.n_reads) {
        ReadsView v = rs->view();
        v.uniform_len = h.uniform_len;
        COMMET_LAUNCH(kmer_counts_kernel, dim3((unsigned) ((h.n_reads + 255) / 256)), dim3(256), 0, c->load_stream, v, c->k, rs->d_kcnt,
                           rs->d_lenmm);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->load_stream);
    munmap((void *) m, lay.total);
    if (e != hipSuccess) {
        fail("read set upload failed: %s", hipGetErrorString(e));
        commet_readset_destroy(rs);
        return nullptr;
    }
    return rs;
}

/* ---- a resident set handed to another process of the node without a file ---------------------------------------- */
namespace {
struct ExportTail {                 // behind PackHeader + file spans + empty reads, 8-byte aligned
    hipIpcMemHandle_t planes, goff;
    int32_t device, has_goff;
};
}  // namespace

int commet_readset_export(const commet_readset *rs, void *blob, uint64_t cap, uint64_t *blob_bytes)
{
    if (!rs->finalized) return fail("read set not finalized");
    commet_ctx *c = rs->ctx;
    HIP_OK(hipSetDevice(c->device));
    PackHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "CMTIPC1", 8);
    h.n_reads = rs->n_reads, h.n_bases = rs->n_bases, h.triples = (rs->n_bases >> 5) + rs->n_reads + 1;
    h.n_files = rs->files.size(), h.n_empty = rs->empty_reads.size();
    h.uniform_len = rs->uniform_len, h.min_len = rs->min_len, h.max_len = rs->max_len;
    const uint64_t files_at = align64(sizeof h), empty_at = files_at + h.n_files * sizeof(FileSpan);
    const uint64_t tail_at = align64(empty_at + h.n_empty * 8), total = tail_at + sizeof(ExportTail);
    if (blob_bytes) *blob_bytes = total;
    if (!blob || cap < total) return blob ? fail("export buffer too small (%llu bytes needed)", (unsigned long long) total) : 0;   // (size query)
    uint8_t *m = (uint8_t *) blob;
    memset(m, 0, total);
    memcpy(m, &h, sizeof h);
    if (h.n_files) memcpy(m + files_at, rs->files.data(), h.n_files * sizeof(FileSpan));
    if (h.n_empty) memcpy(m + empty_at, rs->empty_reads.data(), h.n_empty * 8);
    ExportTail t;
    memset(&t, 0, sizeof t);
    t.device = c->device, t.has_goff = h.uniform_len ? 0 : 1;
    HIP_OK(hipStreamSynchronize(c->load_stream));        // the planes are complete
    HIP_OK(hipIpcGetMemHandle(&t.planes, rs->d_planes));
    if (t.has_goff) HIP_OK(hipIpcGetMemHandle(&t.goff, rs->d_goff));
    memcpy(m + tail_at, &t, sizeof t);
    return 0;
}

commet_readset *commet_readset_import(commet_ctx *c, const void *blob, uint64_t blob_bytes)
{
    PackHeader h;
    if (!blob || blob_bytes < sizeof h) {
        fail("not an exported read set");
        return nullptr;
    }
    memcpy(&h, blob, sizeof h);
    const uint64_t files_at = align64(sizeof h);
    if (memcmp(h.magic, "CMTIPC1", 8) != 0 || h.n_files > blob_bytes / sizeof(FileSpan) || h.n_empty > blob_bytes / 8 ||
        h.triples != (h.n_bases >> 5) + h.n_reads + 1) {
        fail("not an exported read set");
        return nullptr;
    }
    const uint64_t empty_at = files_at + h.n_files * sizeof(FileSpan), tail_at = align64(empty_at + h.n_empty * 8);
    if (tail_at + sizeof(ExportTail) != blob_bytes) {
        fail("inconsistent exported read set");
        return nullptr;
    }
    const uint8_t *m = (const uint8_t *) blob;
    ExportTail t;
    memcpy(&t, m + tail_at, sizeof t);
    commet_readset *rs = commet_readset_create(c, h.n_reads, h.n_bases);
    if (!rs) return nullptr;
    const FileSpan *fs = (const FileSpan *) (m + files_at);
    rs->files.assign(fs, fs + h.n_files);
    const uint64_t *er = (const uint64_t *) (m + empty_at);
    rs->empty_reads.assign(er, er + h.n_empty);
    rs->n_reads = h.n_reads;
    rs->n_bases = h.n_bases;
    // the owner's buffers, mapped into this process (another device of the node: over xGMI), copied device to device
    void *src_planes = nullptr, *src_goff = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&src_planes, t.planes, hipIpcMemLazyEnablePeerAccess);
    if (e == hipSuccess && t.has_goff) e = hipIpcOpenMemHandle(&src_goff, t.goff, hipIpcMemLazyEnablePeerAccess);
    const uint32_t mm[3] = {h.n_reads ? h.min_len : 0xFFFFFFFFu, h.max_len, 0u};
    if (e == hipSuccess) e = hipMemcpyAsync(rs->d_lenmm, mm, sizeof mm, hipMemcpyHostToDevice, c->load_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(rs->d_planes, src_planes, h.triples * 12, hipMemcpyDeviceToDevice, c->load_stream);
    if (e == hipSuccess && t.has_goff) e = hipMemcpyAsync(rs->d_goff, src_goff, (h.n_reads + 1) * 8, hipMemcpyDeviceToDevice, c->load_stream);
    if (e == hipSuccess && h.n_reads) {
        ReadsView v = rs->view();
        v.uniform_len = h.uniform_len;
        COMMET_LAUNCH(kmer_counts_kernel, dim3((unsigned) ((h.n_reads + 255) / 256)), dim3(256), 0, c->load_stream, v, c->k, rs->d_kcnt,
                      rs->d_lenmm);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->load_stream);
    if (src_planes) (void) hipIpcCloseMemHandle(src_planes);
    if (src_goff) (void) hipIpcCloseMemHandle(src_goff);
    if (e != hipSuccess) {
        fail("read set import failed (device %d -> %d): %s", t.device, c->device, hipGetErrorString(e));
        (void) hipGetLastError();
        commet_readset_destroy(rs);
        return nullptr;
    }
    return rs;
}

uint64_t commet_readset_num_reads(const commet_readset *rs) { return rs->n_reads; }
uint64_t commet_readset_num_files(const commet_readset *rs) { return rs->files.size(); }

int commet_readset_kmer_counts(const commet_readset *rs, uint32_t *out)
{
    if (!rs->finalized) return fail("read set not finalized");
    if (host_counts(rs)) return 1;
    if (rs->n_reads) memcpy(out, rs->h_kcnt.data(), rs->n_reads * sizeof(uint32_t));
    return 0;
}

/* ---- kernels -------------------------------------------------------------- */

int commet_filter_reset(commet_ctx *c)
{
    HIP_OK(hipSetDevice(c->device));
    KScope ks(c, "filter_memset", c->stream);
    HIP_OK(hipMemsetAsync(c->slot_ptr(c->cur_slot), 0, c->filter_bytes, c->stream));
    return 0;
}

}  // extern "C"

namespace {

// uploads a host bit array (n/8+1 bytes) into a device bitmap of bitmap_words(n) words
int upload_bits(commet_ctx *c, uint64_t *d_bits, const uint8_t *h_bits, uint64_t n)
{
    HIP_OK(hipMemsetAsync(d_bits, 0, bitmap_words(n) * 8, c->stream));
    HIP_OK(hipMemcpyAsync(d_bits, h_bits, bitmap_bytes_host(n), hipMemcpyHostToDevice, c->stream));
    return 0;
}

int launch_index_atomic(commet_ctx *c, const commet_readset *rs, uint64_t first, uint64_t count, const uint64_t *d_sel,
                        unsigned long long *d_fed)
{
    if (count == 0) return 0;
    const uint64_t blocks = (count + 255) / 256;
    if (blocks >= (1ull << 24)) return fail("index launch too large (>= 2^32 reads in one chunk)");
    KScope ks(c, "index_kernel", c->stream);
    if (c->k <= 32)
        COMMET_LAUNCH(index_kernel<uint32_t>, dim3((unsigned) blocks), dim3(256), 0, c->stream, rs->view(), c->view(),
                           c->k, first, count, d_sel, d_fed);
    else
        COMMET_LAUNCH(index_kernel<uint64_t>, dim3((unsigned) blocks), dim3(256), 0, c->stream, rs->view(), c->view(),
                           c->k, first, count, d_sel, d_fed);
    HIP_OK(hipGetLastError());
    return 0;
}

bool partition_eligible(const commet_ctx *c, const commet_readset *rs)
{
    return c->k >= 20 && c->k <= 34 && (uint64_t) rs->max_kcnt * 4 <= S1_KEYS &&
           ((uint64_t) rs->max_len + 7) / 8 <= S1_ITEMS;
}

// Bucketed construction of the filter for one chunk (index_part.hpp).  The
// filter must have been zeroed on the stream before.  kmers = exact number of
// complete k-mers of the selected reads of [first, first+count).
// d_ids != nullptr (fixed-length sets only): the chunk's selected reads are ids[pos_first .. pos_first + pos_count) (sel_ids_kernel);
// hist and scatter1 then take the arithmetic item path over that list instead of planning rounds over the bitmap
int launch_index_partitioned(commet_ctx *c, const commet_readset *rs, uint64_t first, uint64_t count, const uint64_t *d_sel,
                             uint64_t kmers, bool additive, bool zero_fill, int lane = 0, const uint32_t *d_ids = nullptr,
                             uint64_t pos_first = 0, uint64_t pos_count = 0)
{
    if (count == 0 || kmers == 0) return 0;
    if (d_ids && rs->uniform_len != 0 && !c->part_no_uni && pos_count) first = pos_first, count = pos_count, d_sel = nullptr;
    else d_ids = nullptr;
    commet_ctx::PartWs &ws = c->part[lane];
    hipStream_t stream = lane ? c->aux_stream : c->stream;
    uint32_t *const slot = c->slot_ptr(c->cur_slot);
    PartGeom g = make_geom(c->k);
    g.xcd_swizzle = c->s2_swizzle;
    g.packed = c->part_packed;
    if (c->part_b1 > 0 && c->part_b1 < g.nb_bits && c->part_b1 <= 8 && g.nb_bits - c->part_b1 <= 9) {
        g.b1 = c->part_b1;
        g.b2 = g.nb_bits - g.b1;
        g.nb1 = 1u << g.b1;
    }
    if (g.b2 == 0) g.packed = 0;   // single level: scatter1 writes the final buckets itself, as plain keys
    const uint64_t total = 4 * kmers;
    if (ws.nb != g.nb) {
        (void) hipFree(ws.hist); (void) hipFree(ws.wl); (void) hipFree(ws.off); (void) hipFree(ws.goff);
        (void) hipFree(ws.cur2);
        ws.hist = ws.wl = nullptr; ws.off = ws.goff = nullptr; ws.cur2 = nullptr;
        HIP_OK(dev_alloc(c, (void **) &ws.hist, (g.nb + 1) * sizeof(uint32_t), true));
        HIP_OK(dev_alloc(c, (void **) &ws.wl, (g.nb + 1) * sizeof(uint32_t), true));
        HIP_OK(dev_alloc(c, (void **) &ws.off, (g.nb + 1) * sizeof(uint64_t), true));
        HIP_OK(dev_alloc(c, (void **) &ws.goff, (g.nb + 1) * sizeof(uint64_t), true));
        if (!ws.blockcnt) HIP_OK(dev_alloc(c, (void **) &ws.blockcnt, (size_t) S1_GRID_MAX * MAX_L1 * sizeof(uint32_t), true));
        if (!ws.blockoff) HIP_OK(dev_alloc(c, (void **) &ws.blockoff, (size_t) S1_GRID_MAX * MAX_L1 * sizeof(unsigned long long), true));
        HIP_OK(dev_alloc(c, (void **) &ws.cur2, g.nb * sizeof(unsigned long long), true));
        ws.nb = g.nb;
    }
    if (ws.cap_keys < total) {
        HIP_OK(hipStreamSynchronize(stream));
        (void) hipFree(ws.bufA); (void) hipFree(ws.bufB);
        ws.bufA = ws.bufB = nullptr;
        ws.cap_keys = 0;
        const uint64_t cap = total + total / 16 + (1ull << 20);   // bufB, packed: 2/3 + nsub/4096 words per key + a constant
        HIP_OK(dev_alloc(c, (void **) &ws.bufA, cap * sizeof(uint32_t), true));
        HIP_OK(dev_alloc(c, (void **) &ws.bufB, cap * sizeof(uint32_t), true));
        // first touch here, not inside the first scatter2 launch (measured: 15.6 ms instead of 2.5 ms for that one launch)
        HIP_OK(hipMemsetAsync(ws.bufA, 0, cap * sizeof(uint32_t), stream));
        HIP_OK(hipMemsetAsync(ws.bufB, 0, cap * sizeof(uint32_t), stream));
        ws.cap_keys = cap;
    }
    const bool wide = c->k > 32;
    // every read of one length and no selection bitmap: items by arithmetic, no round planning (index_part.hpp, UNI)
    const bool uni = rs->uniform_len != 0 && d_sel == nullptr && !c->part_no_uni;   // (d_ids: positions in the list of selected reads)
    HIP_OK(hipMemsetAsync(ws.hist, 0, (g.nb + 1) * sizeof(uint32_t), stream));
    // scatter1's grid fixes how the read range is cut; hist counts with the same cut, two ranges per workgroup
    const uint32_t grid1 = (uint32_t) std::min<uint64_t>(S1_GRID_MAX, (count + 63) / 64);
    {
        const unsigned grid = (grid1 + 1) / 2;
        const bool full = g.nb <= HIST_MAX_BUCKETS;
        // 32-bit keys (k <= 32): at most 2^15 buckets, the LDS histogram always covers them all (FULL); 64-bit keys: never.
        // Only those four instantiations exist (tests/test_gpu_zz_dispatch_coverage.py checks that each is reached).
        if (full == wide) return fail("internal error: histogram geometry (k = %d, %u buckets)", c->k, g.nb);
        const void *fn = wide ? (uni ? (const void *) part_hist_kernel<uint64_t, true, false> : (const void *) part_hist_kernel<uint64_t, false, false>)
                              : (uni ? (const void *) part_hist_kernel<uint32_t, true, true> : (const void *) part_hist_kernel<uint32_t, false, true>);
        for (uint32_t b_lo = 0; b_lo < g.nb; b_lo += HIST_MAX_BUCKETS) {
            const uint32_t n_b = std::min<uint32_t>(HIST_MAX_BUCKETS, g.nb - b_lo);
            const size_t lds = ((size_t) n_b + 2 * HIST_NT + 24) * 4 + (size_t) HIST_NT * 8;
            HIP_OK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
            ReadsView rv = rs->view();
            const uint32_t *kc = rs->d_kcnt;
            uint32_t *hist = ws.hist, *bcnt = ws.blockcnt;
            uint32_t nblk = grid1;
            void *args[] = {&rv, &kc, &d_sel, &first, &count, &g, &b_lo, (void *) &n_b, &hist, &nblk, &bcnt, &d_ids};
            KScope ks(c, "part_hist_kernel", stream);
            note_launch(fn);
            HIP_OK(hipLaunchKernel(fn, dim3(grid), dim3(HIST_NT), args, lds, stream));
        }
    }
    {
        const bool lds_hist = (size_t) g.nb * 4 <= (128u << 10);   // stage the histogram in LDS (coalesced loads) when it fits
        const size_t lds = lds_hist ? ((size_t) g.nb + g.nb / 32 + 1) * 4 : 0;   // (padded: see the kernel)
        if (lds) HIP_OK(hipFuncSetAttribute((const void *) part_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        KScope ks(c, "part_scan_kernel", stream);
        COMMET_LAUNCH(part_scan_kernel, dim3(1), dim3(1024), lds, stream, ws.hist, g, zero_fill ? 1 : 0, ws.off,
                           ws.cur2, ws.wl, ws.goff, lds_hist ? 1 : 0);
    }
    HIP_OK(hipGetLastError());
    {
        KScope ks(c, "part_blockoff_kernel", stream);
        COMMET_LAUNCH(part_blockoff_kernel, dim3(g.nb1), dim3(512), 0, stream, ws.blockcnt, ws.off, g, grid1,
                           ws.blockoff);
    }
    HIP_OK(hipGetLastError());
    // scatter 1 (straight into the final buckets when there is a single level)
    uint32_t *level1_out = g.b2 ? ws.bufA : ws.bufB;
    {
        const void *fn = wide ? (uni ? (const void *) part_scatter1_kernel<uint64_t, true> : (const void *) part_scatter1_kernel<uint64_t, false>)
                              : (uni ? (const void *) part_scatter1_kernel<uint32_t, true> : (const void *) part_scatter1_kernel<uint32_t, false>);
        ReadsView rv = rs->view();
        const uint32_t *kc = rs->d_kcnt;
        const unsigned long long *boff = ws.blockoff;
        void *args[] = {&rv, &kc, &d_sel, &first, &count, &g, &boff, &level1_out, &d_ids};
        KScope ks(c, "part_scatter1_kernel", stream);
        note_launch(fn);
        HIP_OK(hipLaunchKernel(fn, dim3(grid1), dim3(S1_NT), args, 0, stream));
    }
    if (COMMET_ABLATE & 31) return 0;   // ablation builds only: scatter1 left garbage in bufA, nothing downstream may consume it
    if (g.b2) {
        const uint64_t grid = (total + S2_KEYS - 1) / S2_KEYS;
        if (grid >= (1ull << 24)) return fail("scatter launch too large");
        {
            KScope ks(c, (g.packed && (1u << g.b2) <= S2P_MAX_SUB) ? "part_scatter2_packed_kernel" : "part_scatter2_kernel", stream);
            if (g.packed && (1u << g.b2) <= S2P_MAX_SUB)
                COMMET_LAUNCH(part_scatter2_packed_kernel, dim3((unsigned) grid), dim3(S2_NT), 0, stream, ws.bufA, (uint2 *) ws.bufB,
                                   ws.off, g, ws.cur2, total);
            else
                COMMET_LAUNCH(part_scatter2_kernel, dim3((unsigned) grid), dim3(S2_NT), 0, stream, ws.bufA, ws.bufB,
                                   ws.off, g, ws.cur2, total);
        }
        HIP_OK(hipGetLastError());
    }
    if (COMMET_ABLATE) return 0;   // ablation builds only: bufB holds garbage
    {
        const uint64_t grid = (uint64_t) g.nb + total / BUILD_CAP + 1;
        if (grid >= (1ull << 24)) return fail("build launch too large");
        if (zero_fill) {   // no memset happened: clear the tiles that several workgroups OR into
            KScope ks(c, "part_zero_split_kernel", stream);
            COMMET_LAUNCH(part_zero_split_kernel, dim3(g.nb), dim3(256), 0, stream, ws.wl, g, slot);
            HIP_OK(hipGetLastError());
        }
        HIP_OK(hipFuncSetAttribute((const void *) part_build_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int) (TILE_WORDS * sizeof(uint32_t))));
        {
            KScope ks(c, "part_build_kernel", stream);
            COMMET_LAUNCH(part_build_kernel, dim3((unsigned) grid), dim3(BUILD_NT), TILE_WORDS * sizeof(uint32_t), stream,
                               ws.bufB, g.packed ? ws.goff : ws.off, ws.wl, g, slot, additive ? 1 : 0, ws.cur2);
        }
        HIP_OK(hipGetLastError());
    }
    return 0;
}

// whether a launch of `kmers` complete k-mers takes the bucketed construction
bool would_partition(const commet_ctx *c, const commet_readset *rs, uint64_t kmers)
{
    if (kmers == ~0ull || kmers == 0) return false;
    if (c->index_mode == 2) return partition_eligible(c, rs);
    if (c->index_mode == 0) return partition_eligible(c, rs) && kmers >= c->part_min_kmers;
    return false;
}

// kmers: exact complete-k-mer count of the launch when known (enables the bucketed path), else ~0.
// fresh_filter: the filter holds nothing yet; filter_zeroed: the caller has zeroed it (if not, a bucketed build
// zero-fills what it does not set; the atomic kernel always needs a zeroed filter).
int launch_index(commet_ctx *c, const commet_readset *rs, uint64_t first, uint64_t count, const uint64_t *d_sel,
                 unsigned long long *d_fed, uint64_t kmers = ~0ull, bool fresh_filter = false, bool filter_zeroed = true,
                 int lane = 0, const uint32_t *d_ids = nullptr, uint64_t pos_first = 0, uint64_t pos_count = 0)
{
    if (c->index_mode == 2) {
        if (!partition_eligible(c, rs)) return fail("bucketed index construction needs 20 <= k <= 34 and reads of at most %u k-mers", S1_KEYS / 4);
        if (kmers == ~0ull) return fail("bucketed index construction needs the k-mer count of the launch");
    }
    if (!would_partition(c, rs, kmers)) {
        if (!filter_zeroed) return fail("internal error: atomic index launch on a filter that was not zeroed");
        return launch_index_atomic(c, rs, first, count, d_sel, d_fed);
    }
    if (d_fed) {
        // the count is known exactly on the host
        const unsigned long long v = kmers;
        HIP_OK(hipMemcpyAsync(d_fed, &v, sizeof v, hipMemcpyHostToDevice, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
    }
    return launch_index_partitioned(c, rs, first, count, d_sel, kmers, !fresh_filter, fresh_filter && !filter_zeroed, lane, d_ids, pos_first,
                                    pos_count);
}

// min_hits as the kernels get it: a read of max_len bases holds at most max_len / k non-overlapping k-mers, so every
// t above max_len / k + 1 behaves like that value (never found, same probes); clamping keeps (t - seen - 1) * k and
// last - (t - 1) * k inside 32-bit int whatever atoi handed to commet_create
inline int t_eff(const commet_ctx *c, const commet_readset *rs)
{
    return (int) std::min<uint64_t>((uint64_t) c->t, (uint64_t) rs->max_len / (uint64_t) c->k + 1);
}

int launch_search(commet_ctx *c, const commet_readset *rs, const uint64_t *d_sel, uint64_t *d_tags, uint64_t *d_found,
                  unsigned long long *d_counters, unsigned long long *d_probes = nullptr)
{
    if (rs->n_reads == 0) return 0;
    const uint64_t blocks = (rs->n_reads + 255) / 256;
    if (blocks >= (1ull << 24)) return fail("search launch too large (>= 2^32 reads in one set)");
    const dim3 g((unsigned) blocks), b(256);
    const bool cnt = d_probes != nullptr;
    KScope ks(c, "search_kernel", c->stream);
    if (c->k <= 32) {
        if (cnt)
            COMMET_LAUNCH((search_kernel<uint32_t, true>), g, b, 0, c->stream, rs->view(), c->view(), c->k, t_eff(c, rs), d_sel,
                               d_tags, d_found, d_counters, d_probes);
        else
            COMMET_LAUNCH((search_kernel<uint32_t, false>), g, b, 0, c->stream, rs->view(), c->view(), c->k, t_eff(c, rs), d_sel,
                               d_tags, d_found, d_counters, d_probes);
    } else {
        if (cnt)
            COMMET_LAUNCH((search_kernel<uint64_t, true>), g, b, 0, c->stream, rs->view(), c->view(), c->k, t_eff(c, rs), d_sel,
                               d_tags, d_found, d_counters, d_probes);
        else
            COMMET_LAUNCH((search_kernel<uint64_t, false>), g, b, 0, c->stream, rs->view(), c->view(), c->k, t_eff(c, rs), d_sel,
                               d_tags, d_found, d_counters, d_probes);
    }
    HIP_OK(hipGetLastError());
    return 0;
}

// makes `g` filter slots (+ the interleaved A planes with stride gs) available; slot contents are undefined after a grow
int ensure_slots(commet_ctx *c, int g, int gs)
{
    if (c->n_slots < g) {
        HIP_OK(hipStreamSynchronize(c->stream));
        uint32_t *nf = nullptr;
        hipError_t e = dev_alloc(c, (void **) &nf, (size_t) g * c->filter_bytes, true);
        if (e != hipSuccess) return fail("cannot allocate %d filter slots: %s", g, hipGetErrorString(e));
        (void) hipFree(c->filter);
        c->filter = nf;
        c->n_slots = g;
    }
    if (c->il_stride < gs) {
        HIP_OK(hipStreamSynchronize(c->stream));
        (void) hipFree(c->il_a);
        c->il_a = nullptr;
        c->il_stride = 0;
        HIP_OK(dev_alloc(c, (void **) &c->il_a, (size_t) gs * c->plane_words * sizeof(uint32_t), true));
        c->il_stride = gs;
    }
    return 0;
}

int launch_interleave(commet_ctx *c, int g, int gs)
{
    const uint64_t blocks = std::min<uint64_t>((c->plane_words + 255) / 256, 1u << 16);
    KScope ks(c, "interleave_a_kernel", c->stream);
    if (gs == 2)
        COMMET_LAUNCH(interleave_a_kernel<2>, dim3((unsigned) blocks), dim3(256), 0, c->stream, c->filter, 4 * c->plane_words,
                           c->plane_words, g, c->il_a);
    else if (gs == 4)
        COMMET_LAUNCH(interleave_a_kernel<4>, dim3((unsigned) blocks), dim3(256), 0, c->stream, c->filter, 4 * c->plane_words,
                           c->plane_words, g, c->il_a);
    else
        COMMET_LAUNCH(interleave_a_kernel<8>, dim3((unsigned) blocks), dim3(256), 0, c->stream, c->filter, 4 * c->plane_words,
                           c->plane_words, g, c->il_a);
    HIP_OK(hipGetLastError());
    return 0;
}

template <typename W, int GS>
int launch_search_group_t(commet_ctx *c, const commet_readset *rs, const FilterGroupView &fg, uint32_t nw_max, const uint64_t *d_sel,
                          uint64_t *d_tags, unsigned long long *d_counters, uint32_t cstride, unsigned long long *d_probes)
{
    const dim3 g((unsigned) ((rs->n_reads + 255) / 256)), b(256);
    size_t lds = (size_t) fg.g * 2 * nw_max * 256 * sizeof(uint32_t);
    // the lanes' reads staged in LDS too (3 * nw_max words each) when that still fits 64 KiB
    uint32_t rw_nw = 0;
    if (!d_probes && nw_max <= 8 && lds + (size_t) 3 * nw_max * 256 * sizeof(uint32_t) <= (64u << 10) && c->stage_reads) {
        rw_nw = nw_max;
        lds += (size_t) 3 * nw_max * 256 * sizeof(uint32_t);
    }
    KScope ks(c, "search_group_kernel", c->stream);
    if (d_probes) {
        HIP_OK(hipFuncSetAttribute((const void *) search_group_kernel<W, GS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        COMMET_LAUNCH((search_group_kernel<W, GS, true>), g, b, lds, c->stream, rs->view(), fg, c->k, t_eff(c, rs), nw_max, d_sel, d_tags,
                           d_counters, cstride, d_probes, rw_nw);
    } else {
        HIP_OK(hipFuncSetAttribute((const void *) search_group_kernel<W, GS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        COMMET_LAUNCH((search_group_kernel<W, GS, false>), g, b, lds, c->stream, rs->view(), fg, c->k, t_eff(c, rs), nw_max, d_sel, d_tags,
                           d_counters, cstride, d_probes, rw_nw);
    }
    HIP_OK(hipGetLastError());
    return 0;
}

// one pass of rs over the `g` chunk filters in slots 0..g-1 (A planes already interleaved with stride gs)
int launch_search_group(commet_ctx *c, const commet_readset *rs, int g, int gs, const uint64_t *d_sel, uint64_t *d_tags,
                        unsigned long long *d_counters, uint32_t cstride, unsigned long long *d_probes)
{
    if (rs->n_reads == 0) return 0;
    if ((rs->n_reads + 255) / 256 >= (1ull << 24)) return fail("search launch too large (>= 2^32 reads in one set)");
    FilterGroupView fg;
    fg.il_a = c->il_a;
    fg.slot0 = c->filter;
    fg.slot_words = 4 * c->plane_words;
    fg.plane_words = c->plane_words;
    fg.g = g;
    if (gs == 8) {   // register masks, no LDS (group8_ok)
        const dim3 grid((unsigned) ((rs->n_reads + 255) / 256)), block(256);
        const bool three = (int64_t) rs->max_len - (int64_t) t_eff(c, rs) * c->k + 1 > 64;   // mask words per strand and filter
        KScope ks(c, "search_group8_kernel", c->stream);
        if (c->k <= 32) {
            if (three)
                COMMET_LAUNCH((search_group8_kernel<uint32_t, 3>), grid, block, 0, c->stream, rs->view(), fg, c->k, t_eff(c, rs), d_sel,
                                   d_tags, d_counters, cstride);
            else
                COMMET_LAUNCH((search_group8_kernel<uint32_t, 2>), grid, block, 0, c->stream, rs->view(), fg, c->k, t_eff(c, rs), d_sel,
                                   d_tags, d_counters, cstride);
        } else {
            if (three)
                COMMET_LAUNCH((search_group8_kernel<uint64_t, 3>), grid, block, 0, c->stream, rs->view(), fg, c->k, t_eff(c, rs), d_sel,
                                   d_tags, d_counters, cstride);
            else
                COMMET_LAUNCH((search_group8_kernel<uint64_t, 2>), grid, block, 0, c->stream, rs->view(), fg, c->k, t_eff(c, rs), d_sel,
                                   d_tags, d_counters, cstride);
        }
        HIP_OK(hipGetLastError());
        return 0;
    }
    const uint32_t nw_max = (rs->max_len + 31) / 32;
    if (c->k <= 32) {
        if (gs == 2) return launch_search_group_t<uint32_t, 2>(c, rs, fg, nw_max, d_sel, d_tags, d_counters, cstride, d_probes);
        return launch_search_group_t<uint32_t, 4>(c, rs, fg, nw_max, d_sel, d_tags, d_counters, cstride, d_probes);
    }
    if (gs == 2) return launch_search_group_t<uint64_t, 2>(c, rs, fg, nw_max, d_sel, d_tags, d_counters, cstride, d_probes);
    return launch_search_group_t<uint64_t, 4>(c, rs, fg, nw_max, d_sel, d_tags, d_counters, cstride, d_probes);
}

bool group_searchable(const commet_ctx *c, const commet_readset *rs, int g)
{
    // LDS masks: g chunks x 2 strands x ceil(max_len/32) words per lane, at most 64 KiB per workgroup
    const uint64_t nw = ((uint64_t) rs->max_len + 31) / 32;
    return c->k >= 2 && nw >= 1 && (uint64_t) g * 2 * nw * 256 * 4 <= (64u << 10);
}

// groups of 5..8 chunk filters: search_group8_kernel keeps the gathered bits of at most 96 first-hit windows per read in
// registers (kernels.hpp); the probe-counting builds exist for groups of <= 4 only
bool group8_ok(const commet_ctx *c, const commet_readset *rs)
{
    const int64_t first_hit_windows = (int64_t) rs->max_len - (int64_t) t_eff(c, rs) * c->k + 1;
    return c->k >= 2 && !c->count_probes && first_hit_windows <= 96;
}

// ---- tiled search (tile_search.hpp) ----------------------------------------------------------------------------
constexpr int TQ_SBITS = 24;          // slice = 2^24 bits of plane A's address space: 2 MiB per chunk filter, 4 MiB for a group of two
                                      // (measured on configs[1]: 22 / 23 / 24 -> probe 2.43 / 2.56 / 2.35 ms, gpurun_out/r02_tq_ab2.log)

constexpr int TQ_MAX_K = 34;          // 64-bit keys from k = 33 (the reference's default k, index_and_search.cpp:71): 2^(k - 24) <= 1024 slices

bool tiled_ok(const commet_ctx *c, const commet_readset *rs, int g)
{
    if (c->tiled_mode == 1 || c->count_probes || rs->ql.failed) return false;
    if (c->k <= TQ_SBITS || c->k > TQ_MAX_K || g < 1 || g > 2) return false;
    const int64_t first_hit_windows = (int64_t) rs->max_len - (int64_t) t_eff(c, rs) * c->k + 1;
    if (first_hit_windows < 1 || first_hit_windows > TQ_MAX_WIN) return false;
    if (rs->n_reads >= (1ull << 32)) return false;
    if (c->tiled_mode == 2) return true;
    // auto: sets of a million reads or more whose query list (8 bytes per first-hit window) stays under 4 GiB.  Measured
    // against the fused kernels: configs[1] search 9.4 -> 8.6 ms, configs[2] jobs 1.92-1.96 -> 1.84 s.
    return rs->n_reads >= (1ull << 20) && rs->n_reads * (uint64_t) first_hit_windows * 8 <= c->ql_max_list &&
           rs->n_reads * (uint64_t) first_hit_windows < (1ull << 32);   // (record numbers are 32 bits)
}

// the set's query list for this context's (k, t): counted, scanned, filled; kept with the set
int build_query_list(commet_ctx *c, const commet_readset *rs)
{
    commet_readset::QueryList &ql = rs->ql;
    if (ql.built) {
        ql.last_use = ++c->ql_clock;
        return 0;
    }
    ql.sbits = TQ_SBITS;
    if (c->tq_sbits) ql.sbits = std::max(c->k - 10, std::min(c->k - 1, c->tq_sbits));   // A/B runs
    ql.n_slices = 1u << (c->k - ql.sbits);
    ql.n_pieces = (uint32_t) ((rs->n_reads + TQ_PIECE - 1) / TQ_PIECE);
    const uint64_t entries = (uint64_t) ql.n_slices * ql.n_pieces;
    const uint32_t nb = (uint32_t) ((entries + 4095) / 4096);
    unsigned long long *d_totals = nullptr;
    // (the caller holds ql_mu: on an allocation failure here the other sets' lists are given back directly)
    auto alloc = [&](void **ptr, size_t bytes) -> hipError_t {
        hipError_t ae = hipMalloc(ptr, bytes);
        if (ae != hipErrorOutOfMemory) return ae;
        (void) hipGetLastError();
        if (!shrink_query_lists(c, 0, false)) return ae;
        ae = hipMalloc(ptr, bytes);
        if (ae == hipErrorOutOfMemory) (void) hipGetLastError();
        return ae;
    };
    hipError_t e = alloc((void **) &ql.d_tile_off, (entries + 1) * sizeof(unsigned long long));
    if (e == hipSuccess) e = alloc((void **) &d_totals, ((size_t) nb + 1) * sizeof(unsigned long long));
    if (e == hipSuccess) {
        const int t = t_eff(c, rs);
        const size_t lds = (size_t) ql.n_slices * 4;
        {
            KScope ks(c, "tq_count_kernel", c->stream);
            if (c->k <= 32)
                COMMET_LAUNCH(tq_count_kernel<uint32_t>, dim3(ql.n_pieces), dim3(256), lds, c->stream, rs->view(), c->k, t, ql.sbits, ql.n_slices,
                              ql.n_pieces, ql.d_tile_off);
            else
                COMMET_LAUNCH(tq_count_kernel<uint64_t>, dim3(ql.n_pieces), dim3(256), lds, c->stream, rs->view(), c->k, t, ql.sbits, ql.n_slices,
                              ql.n_pieces, ql.d_tile_off);
        }
        {
            KScope ks(c, "tq_scan_kernels", c->stream);
            COMMET_LAUNCH(tq_scan_blocks_kernel, dim3(nb), dim3(1024), 0, c->stream, ql.d_tile_off, entries, d_totals);
            COMMET_LAUNCH(tq_scan_totals_kernel, dim3(1), dim3(1024), 0, c->stream, d_totals, nb, d_totals + nb);
            COMMET_LAUNCH(tq_scan_add_kernel, dim3(nb), dim3(1024), 0, c->stream, ql.d_tile_off, entries, d_totals, d_totals + nb);
        }
        e = hipGetLastError();
        unsigned long long total = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&total, d_totals + nb, sizeof total, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        ql.n_records = total;
        if (e == hipSuccess) e = alloc((void **) &ql.d_qaddr, std::max<uint64_t>(total, 1) * 4);
        if (e == hipSuccess && total >= (1ull << 32)) e = hipErrorOutOfMemory;   // tstart is 32 bits (never with the 4 GiB cap)
        if (e == hipSuccess) e = alloc((void **) &ql.d_qwho, std::max<uint64_t>(total, 1) * 2);
        if (e == hipSuccess) e = alloc((void **) &ql.d_tstart, std::max<uint64_t>(entries, 1) * 4);
        if (e == hipSuccess) e = alloc((void **) &ql.d_tlen, std::max<uint64_t>(entries, 1) * 2);
        if (e == hipSuccess) {
            KScope ks(c, "tq_bounds_kernel", c->stream);
            COMMET_LAUNCH(tq_bounds_kernel, dim3((unsigned) ((entries + 255) / 256)), dim3(256), 0, c->stream, ql.d_tile_off, ql.n_slices,
                               ql.n_pieces, ql.d_tstart, ql.d_tlen);
            e = hipGetLastError();
        }
        if (e == hipSuccess) {
            // reads sorted per round in LDS: as many as keep rpr * (first-hit windows per read) within TQ_FILL_CAP records
            const int64_t fhw = std::max<int64_t>(1, (int64_t) rs->max_len - (int64_t) t * c->k + 1);
            uint32_t rpr = TQ_PIECE;
            while (rpr > 1 && (uint64_t) rpr * (uint64_t) fhw > TQ_FILL_CAP) rpr /= 2;
            const size_t lds_fill = ((size_t) 4 * ql.n_slices + 2 * TQ_FILL_CAP) * 4;
            e = hipFuncSetAttribute(c->k <= 32 ? (const void *) tq_fill_kernel<uint32_t> : (const void *) tq_fill_kernel<uint64_t>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_fill);
            if (e == hipSuccess) {
                KScope ks(c, "tq_fill_kernel", c->stream);
                if (c->k <= 32)
                    COMMET_LAUNCH(tq_fill_kernel<uint32_t>, dim3(ql.n_pieces), dim3(256), lds_fill, c->stream, rs->view(), c->k, t, ql.sbits,
                                  ql.n_slices, ql.n_pieces, rpr, ql.d_tile_off, ql.d_qaddr, ql.d_qwho);
                else
                    COMMET_LAUNCH(tq_fill_kernel<uint64_t>, dim3(ql.n_pieces), dim3(256), lds_fill, c->stream, rs->view(), c->k, t, ql.sbits,
                                  ql.n_slices, ql.n_pieces, rpr, ql.d_tile_off, ql.d_qaddr, ql.d_qwho);
                e = hipGetLastError();
            }
        }
    }
    (void) hipFree(d_totals);
    if (e != hipSuccess) {   // no room for the list (or a launch failed): this set keeps the gather kernels
        (void) hipGetLastError();
        ql.release();
        ql.failed = true;
        return 1;
    }
    ql.built = true;
    ql.bytes = (entries + 1) * 8 + ql.n_records * 6 + entries * 6;
    ql.last_use = ++c->ql_clock;
    c->ql_bytes += ql.bytes;
    if (c->ql_bytes > c->ql_budget) (void) shrink_query_lists(c, c->ql_budget, false);   // least recently used first; never one of this job
    return 0;
}

// the scan's result bytes (one per record of the set's query list); no room = this set keeps the gather kernels
int ensure_query_results(commet_ctx *c, const commet_readset *rs)
{
    const uint64_t need = rs->ql.n_records;
    if (c->qres_cap >= need && c->d_qres) return 0;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return 1;
    (void) hipFree(c->d_qres);
    c->d_qres = nullptr, c->qres_cap = 0;
    hipError_t e = hipMalloc((void **) &c->d_qres, std::max<uint64_t>(need, 1));
    if (e == hipErrorOutOfMemory) {   // (the caller holds ql_mu) give back the lists of sets outside this job and try once more
        (void) hipGetLastError();
        if (shrink_query_lists(c, 0, false)) e = hipMalloc((void **) &c->d_qres, std::max<uint64_t>(need, 1));
    }
    if (e != hipSuccess) {
        (void) hipGetLastError();
        rs->ql.failed = true;
        return 1;
    }
    c->qres_cap = need;
    return 0;
}

// one pass of rs over the g <= 2 chunk filters in slots slot0 .. slot0 + g - 1 (g == 2: slots 0, 1 with interleaved A planes)
int launch_search_tiled(commet_ctx *c, const commet_readset *rs, int g, int slot0, const uint64_t *d_sel, uint64_t *d_tags,
                        unsigned long long *d_counters, uint32_t cstride)
{
    if (rs->n_reads == 0) return 0;
    const commet_readset::QueryList &q = rs->ql;
    if (c->qres_cap < q.n_records) return fail("internal error: tiled search without its result buffer");
    QueryListView v;
    v.tile_off = q.d_tile_off, v.qaddr = q.d_qaddr, v.qwho = q.d_qwho, v.tstart = q.d_tstart, v.tlen = q.d_tlen;
    v.n_slices = q.n_slices, v.n_pieces = q.n_pieces, v.sbits = q.sbits;
    FilterGroupView fg;
    fg.slot0 = c->slot_ptr(slot0);
    fg.il_a = g == 1 ? c->slot_ptr(slot0) : c->il_a;   // one filter: its own plane A (stride 1)
    fg.slot_words = 4 * c->plane_words;
    fg.plane_words = c->plane_words;
    fg.g = g;
    // The probe is bound by L2 gathers, the replay by L2-MISSING requests and bookkeeping: different walls.  The set is cut
    // into `parts` runs of pieces; part i's probe and replay go to stream i % 2, every probe waiting for the probe before it
    // (one slice sweep at a time keeps the slice's filter words in L2), so the replay of part i runs beside the probe of
    // part i + 1.  With per-kernel timing on (durations must add up) or a small set: one part, one stream.
    const bool three = (int64_t) rs->max_len - (int64_t) t_eff(c, rs) * c->k + 1 > 64;
    const int t = t_eff(c, rs);
    uint32_t parts = (c->kclock.on || q.n_pieces < 4096) ? 1u : (uint32_t) std::max(1, std::min(16, c->tq_parts));
    const unsigned wpx = c->tq_wpx;
    hipEvent_t ev_probe = c->ev_fork, ev_done = c->ev_join;
    for (uint32_t pi = 0; pi < parts; ++pi) {
        const uint32_t p0 = (uint32_t) ((uint64_t) q.n_pieces * pi / parts), p1 = (uint32_t) ((uint64_t) q.n_pieces * (pi + 1) / parts);
        hipStream_t st = (pi & 1u) ? c->aux_stream : c->stream;
        if (pi) HIP_OK(hipStreamWaitEvent(st, ev_probe, 0));      // behind the previous part's probe (and so behind the filter build)
        {
            KScope ks(c, "tq_probe_kernel", st);
            if (g == 1) COMMET_LAUNCH(tq_probe_kernel<1>, dim3(8 * wpx), dim3(256), 0, st, v, fg.il_a, c->d_qres, p0, p1);
            else COMMET_LAUNCH(tq_probe_kernel<2>, dim3(8 * wpx), dim3(256), 0, st, v, fg.il_a, c->d_qres, p0, p1);
        }
        HIP_OK(hipGetLastError());
        if (pi + 1 < parts) HIP_OK(hipEventRecord(ev_probe, st));
        {
            KScope ks(c, "tq_replay_kernel", st);
            const dim3 grid(p1 - p0), block(TQ_PIECE);
#define COMMET_TQ_REPLAY(W, GS, MW) COMMET_LAUNCH((tq_replay_kernel<W, GS, MW>), grid, block, 0, st, rs->view(), v, c->d_qres, fg, c->k, t, d_sel, d_tags, d_counters, cstride, p0)
            if (c->k <= 32) {
                if (g == 1) {
                    if (three) COMMET_TQ_REPLAY(uint32_t, 1, 3);
                    else COMMET_TQ_REPLAY(uint32_t, 1, 2);
                } else {
                    if (three) COMMET_TQ_REPLAY(uint32_t, 2, 3);
                    else COMMET_TQ_REPLAY(uint32_t, 2, 2);
                }
            } else {
                if (g == 1) {
                    if (three) COMMET_TQ_REPLAY(uint64_t, 1, 3);
                    else COMMET_TQ_REPLAY(uint64_t, 1, 2);
                } else {
                    if (three) COMMET_TQ_REPLAY(uint64_t, 2, 3);
                    else COMMET_TQ_REPLAY(uint64_t, 2, 2);
                }
            }
#undef COMMET_TQ_REPLAY
        }
        HIP_OK(hipGetLastError());
    }
    if (parts > 1) {   // the second stream's replays join the main stream (the even parts are on it already)
        HIP_OK(hipEventRecord(ev_done, c->aux_stream));
        HIP_OK(hipStreamWaitEvent(c->stream, ev_done, 0));
    }
    HIP_OK(hipGetLastError());
    return 0;
}

// words per bit-sliced entry (32 chunk filters per word) for a job of n_chunks chunks; 0 = the job takes the slot path
int slice_words(const commet_ctx *c, uint64_t n_chunks)
{
    if (c->slice_mode == 1 || c->count_probes) return 0;
    if (c->k < SLICE_MIN_K || c->k > SLICE_MAX_K || n_chunks == 0) return 0;
    if (c->slice_mode == 0 && n_chunks < 8) return 0;
    int gw = c->slice_gw ? c->slice_gw : n_chunks > 128 ? 8 : n_chunks > 64 ? 4 : n_chunks > 32 ? 2 : 1;
    while (gw > 1 && (((uint64_t) 16 * gw) << c->k) > (1ull << 30)) gw /= 2;   // the four tables: at most 1 GiB
    return gw;
}

int ensure_slice_buffers(commet_ctx *c, int gw, uint64_t n_chunks)
{
    const uint64_t G = 32ull * gw;
    const uint64_t stage_words = (G * 4) << (c->k - 5), table_words = ((uint64_t) 4 * gw) << c->k;
    if (c->slice_stage_words < stage_words || c->slice_table_words < table_words || c->slice_chunks_cap < n_chunks) {
        HIP_OK(hipStreamSynchronize(c->stream));
        if (c->slice_stage_words < stage_words) {
            (void) hipFree(c->slice_stage);
            c->slice_stage = nullptr, c->slice_stage_words = 0;
            HIP_OK(dev_alloc(c, (void **) &c->slice_stage, stage_words * 4, true));
            c->slice_stage_words = stage_words;
        }
        if (c->slice_table_words < table_words) {
            (void) hipFree(c->slice_tables);
            c->slice_tables = nullptr, c->slice_table_words = 0;
            HIP_OK(dev_alloc(c, (void **) &c->slice_tables, table_words * 4, true));
            c->slice_table_words = table_words;
        }
        if (c->slice_chunks_cap < n_chunks) {
            (void) hipFree(c->d_slice_chunks);
            c->d_slice_chunks = nullptr, c->slice_chunks_cap = 0;
            HIP_OK(dev_alloc(c, (void **) &c->d_slice_chunks, n_chunks * sizeof(SliceChunk), true));
            c->slice_chunks_cap = n_chunks;
        }
    }
    return 0;
}

// filters of chunks [ci, ci + g) of the plan -> bit-sliced tables (slice_search.hpp)
int launch_slice_build(commet_ctx *c, const commet_readset *rs, const uint64_t *d_sel, uint64_t ci, int g, int gw,
                       uint32_t *tables = nullptr, uint32_t row_words = 0, uint32_t col0 = 0)
{
    if (!tables) tables = c->slice_tables, row_words = (uint32_t) gw, col0 = 0;   // the table of one group (search_sliced_kernel)
    const int tile_bits = std::min(c->k, SLICE_TILE_BITS);
    const uint32_t tiles = 1u << (c->k - tile_bits);
    const size_t lds = (size_t) 4 << (tile_bits - 5);
    HIP_OK(hipFuncSetAttribute((const void *) slice_build_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    {
        KScope ks(c, "slice_build_kernel", c->stream);
        COMMET_LAUNCH(slice_build_kernel, dim3(4 * tiles, (unsigned) g), dim3(1024), lds, c->stream, rs->view(), d_sel,
                           c->d_slice_chunks + ci, c->k, tile_bits, tiles, c->slice_stage);
    }
    HIP_OK(hipGetLastError());
    const unsigned grid = (unsigned) ((((uint64_t) 4 << (c->k - 5)) + 255) / 256);
    {
        KScope ks(c, "slice_transpose_kernel", c->stream);
        switch (gw) {
        case 1: COMMET_LAUNCH(slice_transpose_kernel<1>, dim3(grid), dim3(256), 0, c->stream, c->slice_stage, c->k, g, tables, row_words, col0); break;
        case 2: COMMET_LAUNCH(slice_transpose_kernel<2>, dim3(grid), dim3(256), 0, c->stream, c->slice_stage, c->k, g, tables, row_words, col0); break;
        case 4: COMMET_LAUNCH(slice_transpose_kernel<4>, dim3(grid), dim3(256), 0, c->stream, c->slice_stage, c->k, g, tables, row_words, col0); break;
        default: COMMET_LAUNCH(slice_transpose_kernel<8>, dim3(grid), dim3(256), 0, c->stream, c->slice_stage, c->k, g, tables, row_words, col0); break;
        }
    }
    HIP_OK(hipGetLastError());
    return 0;
}

int launch_search_sliced(commet_ctx *c, const commet_readset *rs, int g, int gw, const uint64_t *d_sel, uint64_t *d_tags,
                         unsigned long long *d_counters, uint32_t cstride, uint32_t block_stride = 1)
{
    if (rs->n_reads == 0) return 0;
    if ((rs->n_reads + 255) / 256 >= (1ull << 24)) return fail("search launch too large (>= 2^32 reads in one set)");
    const uint64_t blocks = (rs->n_reads + 255) / 256;
    const dim3 grid((unsigned) ((blocks + block_stride - 1) / block_stride)), block(256);
    KScope ks(c, "search_sliced_kernel", c->stream);
    switch (gw) {
    case 1: COMMET_LAUNCH(search_sliced_kernel<1>, grid, block, 0, c->stream, rs->view(), c->slice_tables, c->k, t_eff(c, rs), g, d_sel, d_tags, d_counters, cstride, block_stride); break;
    case 2: COMMET_LAUNCH(search_sliced_kernel<2>, grid, block, 0, c->stream, rs->view(), c->slice_tables, c->k, t_eff(c, rs), g, d_sel, d_tags, d_counters, cstride, block_stride); break;
    case 4: COMMET_LAUNCH(search_sliced_kernel<4>, grid, block, 0, c->stream, rs->view(), c->slice_tables, c->k, t_eff(c, rs), g, d_sel, d_tags, d_counters, cstride, block_stride); break;
    default: COMMET_LAUNCH(search_sliced_kernel<8>, grid, block, 0, c->stream, rs->view(), c->slice_tables, c->k, t_eff(c, rs), g, d_sel, d_tags, d_counters, cstride, block_stride); break;
    }
    HIP_OK(hipGetLastError());
    return 0;
}

// ---- wide rows (slice_search.hpp): every chunk filter of the job — or as many as the table budget allows — in one table ----
struct WidePlan {
    uint32_t nw = 0;              // words per row that hold chunks (a multiple of WIDE_GROUP_WORDS); 0 = no wide pass
    uint32_t rw = 0;              // row stride in words (a multiple of 32: rows start on 128-byte lines)
    uint64_t chunks_per_pass = 0;
    int lpr = 0, np = 0;          // lanes per read, 16-byte pieces per lane
};

WidePlan wide_plan(const commet_ctx *c, uint64_t n_chunks, int slice_gw)
{
    WidePlan w;
    if (!slice_gw || c->slice_wide == 1) return w;
    if (c->slice_wide == 0 && n_chunks <= 256) return w;             // one table of the narrow kind holds them all
    const uint64_t groups = (n_chunks + 255) / 256;
    // four tables of 2^k rows: 16 bytes per row word and key; at most a third of what is free now, and 48 GiB
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return w;
    const uint64_t have = (uint64_t) c->wide_table_words * 4;        // (what the context already holds counts as free)
    const uint64_t budget = std::min<uint64_t>(((uint64_t) free_b + have) / 3, 48ull << 30);
    uint64_t cap = std::min<uint64_t>(WIDE_MAX_ROW_WORDS, budget / (16ull << c->k));
    if (c->wide_cap_words) cap = std::min<uint64_t>(cap, c->wide_cap_words);
    cap = cap / WIDE_GROUP_WORDS * WIDE_GROUP_WORDS;
    if (cap < WIDE_GROUP_WORDS) return w;
    const uint64_t passes = (groups * WIDE_GROUP_WORDS + cap - 1) / cap;
    const uint64_t groups_per_pass = (groups + passes - 1) / passes;
    w.nw = (uint32_t) (groups_per_pass * WIDE_GROUP_WORDS);
    w.rw = (w.nw + 31u) & ~31u;   // rows start on 128-byte lines: a row of 1312 bytes is 11 lines, never 12
    w.chunks_per_pass = groups_per_pass * 256;
    const uint32_t pieces = w.nw / 4;
    w.lpr = pieces <= 8 ? 8 : pieces <= 16 ? 16 : pieces <= 32 ? 32 : 64;
    w.np = pieces <= 64 ? 1 : 2;
    return w;
}

int ensure_wide_tables(commet_ctx *c, const WidePlan &w)
{
    const uint64_t words = ((uint64_t) 4 * w.rw) << c->k;
    if (c->wide_table_words >= words) return 0;
    HIP_OK(hipStreamSynchronize(c->stream));
    (void) hipFree(c->wide_tables);
    c->wide_tables = nullptr, c->wide_table_words = 0;
    if (dev_alloc(c, (void **) &c->wide_tables, words * 4, true) != hipSuccess) {
        (void) hipGetLastError();
        return 1;                                                    // the caller falls back to the narrow tables
    }
    c->wide_table_words = words;
    return 0;
}

int launch_search_wide(commet_ctx *c, const commet_readset *rs, const WidePlan &w, int g, const uint64_t *d_sel, uint64_t *d_tags,
                       unsigned long long *d_counters, uint32_t cstride)
{
    if (rs->n_reads == 0) return 0;
    const uint64_t reads_per_block = 4ull * (64 / w.lpr);
    const uint64_t blocks = (rs->n_reads + reads_per_block - 1) / reads_per_block;
    if (blocks >= (1ull << 31)) return fail("search launch too large");
    const dim3 grid((unsigned) blocks), block(256);
    const int t = t_eff(c, rs);
    KScope ks(c, "search_wide_kernel", c->stream);
#define COMMET_WIDE(LPR, NP) COMMET_LAUNCH((search_wide_kernel<LPR, NP>), grid, block, 0, c->stream, rs->view(), c->wide_tables, c->k, t, g, w.nw, w.rw, d_sel, d_tags, d_counters, cstride)
    if (w.np == 2) COMMET_WIDE(64, 2);
    else if (w.lpr == 64) COMMET_WIDE(64, 1);
    else if (w.lpr == 32) COMMET_WIDE(32, 1);
    else if (w.lpr == 16) COMMET_WIDE(16, 1);
    else COMMET_WIDE(8, 1);
#undef COMMET_WIDE
    HIP_OK(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" {

int commet_index_reads(commet_ctx *c, const commet_readset *rs, uint64_t first, uint64_t count,
                       const uint8_t *select_bits, uint64_t *kmers_fed)
{
    if (!rs->finalized) return fail("read set not finalized");
    if (rs->ctx != c) return fail("read set belongs to another context");
    if (first > rs->n_reads || count > rs->n_reads - first) return fail("index range out of bounds");
    HIP_OK(hipSetDevice(c->device));
    const uint64_t *d_sel = nullptr;
    if (select_bits) {
        if (upload_bits(c, rs->d_sel, select_bits, rs->n_reads)) return 1;
        d_sel = rs->d_sel;
    }
    unsigned long long *d_fed = nullptr;
    if (kmers_fed) {
        HIP_OK(hipMemsetAsync(c->d_counters, 0, sizeof(unsigned long long), c->stream));
        d_fed = c->d_counters;
    }
    HIP_OK(hipEventRecord(c->ev_i0, c->stream));
    // exact k-mer count of the launch (host copy of the per-read counts): lets the bucketed path run
    if (host_counts(rs)) return 1;
    uint64_t kmers = 0;
    for (uint64_t r = first; r < first + count; ++r)
        if (!select_bits || bit_at(select_bits, r)) kmers += rs->h_kcnt[r];
    if (launch_index(c, rs, first, count, d_sel, d_fed, kmers, false)) return 1;
    HIP_OK(hipEventRecord(c->ev_i1, c->stream));
    c->have_index_ev = true;
    if (kmers_fed) {
        HIP_OK(hipMemcpyAsync(c->h_counters, c->d_counters, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
        *kmers_fed = c->h_counters[0];
    }
    return 0;
}

int commet_search_reads(commet_ctx *c, const commet_readset *rs, const uint8_t *active_bits, uint8_t *found_bits,
                        uint64_t *n_scanned, uint64_t *n_found)
{
    if (!rs->finalized) return fail("read set not finalized");
    if (rs->ctx != c) return fail("read set belongs to another context");
    HIP_OK(hipSetDevice(c->device));
    const uint64_t *d_sel = nullptr;
    if (active_bits) {
        if (upload_bits(c, rs->d_sel, active_bits, rs->n_reads)) return 1;
        d_sel = rs->d_sel;
    }
    HIP_OK(hipMemsetAsync(c->d_counters, 0, 2 * sizeof(unsigned long long), c->stream));
    HIP_OK(hipMemsetAsync(rs->d_found, 0, bitmap_words(rs->n_reads) * 8, c->stream));
    HIP_OK(hipEventRecord(c->ev_s0, c->stream));
    if (launch_search(c, rs, d_sel, nullptr, rs->d_found, c->d_counters)) return 1;
    HIP_OK(hipEventRecord(c->ev_s1, c->stream));
    c->have_search_ev = true;
    HIP_OK(hipMemcpyAsync(c->h_counters, c->d_counters, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    if (found_bits)
        HIP_OK(hipMemcpyAsync(found_bits, rs->d_found, bitmap_bytes_host(rs->n_reads), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    if (n_scanned) *n_scanned = c->h_counters[0];
    if (n_found) *n_found = c->h_counters[1];
    return 0;
}

/* ---- the chunk loop (index_and_search.cpp:241-277) ------------------------ */

int commet_index_and_search(commet_ctx *c, const commet_readset *index_rs, const uint8_t *index_select, int n_search,
                            const commet_readset *const *search_rs, const uint8_t *const *search_select,
                            uint8_t *const *tags_out, commet_pair_stats *stats, commet_job_info *info)
{
    const auto wall0 = std::chrono::steady_clock::now();
    // host-side phase times of the call (COMMET_JOB_VERBOSE: one line per call on stderr)
    const bool job_verbose = c->job_verbose;
    auto lap_t = wall0;
    double ph_plan = 0, ph_upload = 0, ph_launch = 0, ph_wait = 0;
    auto lap = [&](double &acc) {
        const auto now = std::chrono::steady_clock::now();
        acc += std::chrono::duration<double, std::milli>(now - lap_t).count();
        lap_t = now;
    };
    if (!index_rs->finalized) return fail("index read set not finalized");
    if (index_rs->ctx != c) return fail("index read set belongs to another context");
    for (int s = 0; s < n_search; ++s) {
        if (!search_rs[s]->finalized) return fail("search read set %d not finalized", s);
        if (search_rs[s]->ctx != c) return fail("search read set %d belongs to another context", s);
        if (search_rs[s] == index_rs) return fail("a set cannot be searched against itself in one call");
        for (int q = 0; q < s; ++q)
            if (search_rs[q] == search_rs[s]) return fail("search read set listed twice");
    }
    HIP_OK(hipSetDevice(c->device));
    // the sets of this call keep their cached query lists whatever memory pressure another thread meets meanwhile
    struct InJob {
        commet_ctx *c;
        const commet_readset *index_rs;
        const commet_readset *const *srs;
        int n;
        void mark(bool v) const
        {
            std::lock_guard<std::mutex> lk(c->ql_mu);
            index_rs->in_job = v;
            for (int i = 0; i < n; ++i) srs[i]->in_job = v;
        }
        InJob(commet_ctx *c_, const commet_readset *i_, const commet_readset *const *s_, int n_) : c(c_), index_rs(i_), srs(s_), n(n_) { mark(true); }
        ~InJob() { mark(false); }
    } in_job(c, index_rs, search_rs, n_search);

    // an input filter that selects every read is no filter (Commet.py passes all-ones bvs when nothing was filtered)
    if (index_select && all_ones(index_select, index_rs->n_reads)) index_select = nullptr;
    // host plan: chunks of the index set, visited reads of each search set
    const uint64_t max_kmer = commet_max_kmer(c);
    // The plan is made from per-block k-mer sums computed on the device, where kcnt lives; the host walks only the
    // blocks in which a chunk starts or ends and fetches just those blocks' counts: no per-read loop over the set and
    // no host copy of its counts (a selection bitmap, when there is one, is uploaded first for the kernel to use).
    std::vector<uint64_t> blk_sums;
    if (index_rs->n_reads && plan_blocks_ok(index_rs->files, index_select, index_rs->empty_reads, max_kmer)) {
        const uint64_t nblk = (index_rs->n_reads + PLAN_BLOCK_READS - 1) / PLAN_BLOCK_READS;
        if (c->plansum_cap < nblk) {
            HIP_OK(hipStreamSynchronize(c->stream));
            (void) hipFree(c->d_plansum);
            c->d_plansum = nullptr;
            c->plansum_cap = 0;
            HIP_OK(dev_alloc(c, (void **) &c->d_plansum, nblk * sizeof(unsigned long long), true));
            c->plansum_cap = nblk;
        }
        if (index_select && upload_bits(c, index_rs->d_sel, index_select, index_rs->n_reads)) return 1;
        {
            KScope ks(c, "block_kmer_sums_kernel", c->stream);
            COMMET_LAUNCH(block_kmer_sums_kernel, dim3((unsigned) nblk), dim3(256), 0, c->stream, index_rs->d_kcnt,
                               index_select ? index_rs->d_sel : nullptr, index_rs->n_reads, c->d_plansum);
        }
        HIP_OK(hipGetLastError());
        blk_sums.resize(nblk);
        HIP_OK(hipMemcpyAsync(blk_sums.data(), c->d_plansum, nblk * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
    }
    // counts of one block of reads, fetched on demand (or taken from the host copy when somebody made one)
    std::vector<uint32_t> kblock(PLAN_BLOCK_READS);
    uint64_t kblock_no = ~0ull;
    bool kfetch_failed = false;
    auto kcnt_of = [&](uint64_t q) -> uint32_t {
        if (index_rs->have_host_counts) return index_rs->h_kcnt[q];
        const uint64_t blk = q / PLAN_BLOCK_READS;
        if (blk != kblock_no) {
            const uint64_t lo = blk * PLAN_BLOCK_READS, cnt = std::min<uint64_t>(PLAN_BLOCK_READS, index_rs->n_reads - lo);
            if (hipMemcpy(kblock.data(), index_rs->d_kcnt + lo, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) kfetch_failed = true;
            kblock_no = blk;
        }
        return kblock[q % PLAN_BLOCK_READS];
    };
    if (blk_sums.empty() && host_counts(index_rs)) return 1;   // the other planners read the counts on the host
    const IndexPlan plan = !blk_sums.empty() ? plan_index_blocks(index_select, kcnt_of, index_rs->n_reads, max_kmer,
                                                                 blk_sums.data(), PLAN_BLOCK_READS)
                           : plan_fast_ok(index_rs->files, index_select, index_rs->empty_reads, max_kmer)
                               ? plan_index_fast(index_rs->h_kprefix, index_rs->n_reads, max_kmer)
                           : (index_select && index_rs->empty_reads.empty())
                               ? plan_index_select(index_rs->files, index_select, index_rs->h_kcnt.data(), index_rs->n_reads, max_kmer)
                               : plan_index(index_rs->files, index_select, index_rs->empty_reads, index_rs->h_kcnt.data(),
                                            index_rs->n_reads, max_kmer);
    if (kfetch_failed) return fail("k-mer count fetch failed: %s", hipGetErrorString(hipGetLastError()));
    std::vector<uint64_t> visited(n_search, 0);
    std::vector<std::vector<uint8_t>> vis(n_search);
    std::vector<char> all_visited(n_search, 0);   // every read of the set is visited: the kernels take a null bitmap
    lap(ph_plan);
    // a dense plan indexes whole read ranges: no bitmap needed on the device
    if (!plan.dense && upload_bits(c, index_rs->d_sel, plan.indexed_bits.data(), index_rs->n_reads)) return 1;
    // a selection on a fixed-length set (Commet.py's J2 / J3 jobs): the selected reads' numbers as a list, so that the
    // bucketed build walks them arithmetically (index_part.hpp, sel_ids_kernel); chunk j's reads are the next n_reads of the list
    const uint32_t *d_ids = nullptr;
    std::vector<uint64_t> chunk_pos;
    uint64_t ids_expected = ~0ull;
    if (!plan.dense && index_rs->uniform_len != 0 && !c->part_no_uni && plan.indexed_reads && c->index_mode != 1) {
        const uint64_t n_words = bitmap_words(index_rs->n_reads), nb = (n_words + IDS_BLOCK_WORDS - 1) / IDS_BLOCK_WORDS;
        bool ok = true;
        if (c->ids_cap < plan.indexed_reads || c->idblk_cap < nb + 1) {
            HIP_OK(hipStreamSynchronize(c->stream));
            (void) hipFree(c->d_ids), (void) hipFree(c->d_idblk);
            c->d_ids = c->d_idblk = nullptr, c->ids_cap = c->idblk_cap = 0;
            const uint64_t cap = std::max<uint64_t>(plan.indexed_reads, index_rs->n_reads / 2);   // (grown rarely)
            ok = dev_alloc(c, (void **) &c->d_ids, cap * sizeof(uint32_t), true) == hipSuccess &&
                 dev_alloc(c, (void **) &c->d_idblk, (nb + 1) * sizeof(uint32_t), true) == hipSuccess;
            if (ok) c->ids_cap = cap, c->idblk_cap = nb + 1;
            else (void) hipGetLastError();              // no room: the round planner walks the bitmap, as before
        }
        if (ok) {
            KScope ks(c, "sel_ids_kernels", c->stream);
            COMMET_LAUNCH(sel_count_kernel, dim3((unsigned) nb), dim3(64), 0, c->stream, index_rs->d_sel, n_words, c->d_idblk);
            COMMET_LAUNCH(sel_scan_kernel, dim3(1), dim3(1024), 0, c->stream, c->d_idblk, (uint32_t) nb);
            COMMET_LAUNCH(sel_ids_kernel, dim3((unsigned) nb), dim3(64), 0, c->stream, index_rs->d_sel, n_words, c->d_idblk, c->d_ids);
            HIP_OK(hipGetLastError());
            // (the list must hold exactly the plan's indexed reads: checked when the job's stream is next synchronised)
            c->h_counters[N_COUNTERS - 1] = ~0ull;
            HIP_OK(hipMemcpyAsync(&c->h_counters[N_COUNTERS - 1], c->d_idblk + nb, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            ids_expected = plan.indexed_reads;
            d_ids = c->d_ids;
            uint64_t at = 0;
            for (const Chunk &ch : plan.chunks) chunk_pos.push_back(at), at += ch.n_reads;
        }
    }
    lap(ph_upload);
    for (int s = 0; s < n_search; ++s) {
        const commet_readset *rs = search_rs[s];
        const uint8_t *ssel = search_select ? search_select[s] : nullptr;
        if (ssel && all_ones(ssel, rs->n_reads)) ssel = nullptr;
        all_visited[s] = plan_fast_ok(rs->files, ssel, rs->empty_reads, 1);
        if (all_visited[s]) visited[s] = rs->n_reads;   // == plan_search_fast, whose bitmap nobody would read
        else
            vis[s] = (ssel && rs->empty_reads.empty()) ? plan_search_select(rs->files, ssel, rs->n_reads, &visited[s])
                                                       : plan_search(rs->files, ssel, rs->empty_reads, rs->n_reads, &visited[s]);
        lap(ph_plan);
        if (!all_visited[s] && upload_bits(c, rs->d_sel, vis[s].data(), rs->n_reads)) return 1;
        HIP_OK(hipMemsetAsync(rs->d_tags, 0, bitmap_words(rs->n_reads) * 8, c->stream));
        lap(ph_upload);
    }
    HIP_OK(hipStreamSynchronize(c->stream));   // the host bit arrays above are pageable
    lap(ph_upload);

    // per (chunk, set) counters {scanned, found}
    const uint64_t n_chunks = plan.chunks.size();
    const uint64_t n_cnt = 2 * n_chunks * (uint64_t) n_search + 1;   // last slot: probe counter
    std::vector<unsigned long long> h_cnt(n_cnt, 0);
    if (c->jobcnt_cap < n_cnt) {   // kept between calls: hipMalloc / hipFree per job cost more than the counters' kernels
        HIP_OK(hipStreamSynchronize(c->stream));
        (void) hipFree(c->d_jobcnt);
        c->d_jobcnt = nullptr;
        c->jobcnt_cap = 0;
        HIP_OK(dev_alloc(c, (void **) &c->d_jobcnt, std::max<uint64_t>(n_cnt, 64) * sizeof(unsigned long long), true));
        c->jobcnt_cap = std::max<uint64_t>(n_cnt, 64);
    }
    unsigned long long *const d_cnt = c->d_jobcnt;
    HIP_OK(hipMemsetAsync(d_cnt, 0, n_cnt * sizeof(unsigned long long), c->stream));

    // device timing: one event pair around all index work and one around all
    // search work would overlap; instead accumulate per phase with event pairs
    // on the (in-order) stream.
    std::vector<hipEvent_t> evs;
    auto new_event = [&](hipEvent_t *e) -> int {
        HIP_OK(hipEventCreate(e));
        evs.push_back(*e);
        return 0;
    };
    // the many-small-chunks regime: the chunk filters of a group live bit-sliced in one set of tables (slice_search.hpp)
    int slice_gw = slice_words(c, n_chunks);
    // no room for the staging planes / tables of that regime: the job takes the slot loop below (slower, same bits)
    if (slice_gw && ensure_slice_buffers(c, c->slice_wide == 1 || (c->slice_wide == 0 && n_chunks <= 256) ? slice_gw : 8, n_chunks)) {
        (void) hipGetLastError();
        slice_gw = 0;
    }
    const bool timed = (info != nullptr || stats != nullptr) &&
                       (slice_gw ? (n_chunks / (32 * slice_gw) + 1) * (uint64_t) (n_search + 2) : n_chunks * (uint64_t) (n_search + 4)) <= 16384;
    std::vector<hipEvent_t> e_idx0, e_idx1, e_zero0, e_zero1;
    std::vector<std::vector<hipEvent_t>> e_set(n_search);   // end of set s's search, per chunk
    uint64_t n_index_launches = 0, n_search_launches = 0;
    unsigned long long *d_probes = c->count_probes ? d_cnt + (n_cnt - 1) : nullptr;

    int rc = 0;
    // chunks are taken in groups of up to `chunk_group`: their filters are built into separate slots and every
    // search set is scanned ONCE per group (search_group_kernel) instead of once per chunk
    int group_cap = (c->k >= 2) ? std::max(1, std::min(8, c->chunk_group)) : 1;
    if (n_chunks < 2) group_cap = 1;
    if (group_cap > 4) {   // more than four filters per pass: every search set must qualify for the register-mask kernel
        bool ok8 = n_chunks > 4;
        for (int s = 0; s < n_search && ok8; ++s) ok8 = group8_ok(c, search_rs[s]);
        if (!ok8) group_cap = 4;
    }
    if (slice_gw) {
        std::vector<SliceChunk> hc(n_chunks);
        for (uint64_t i = 0; i < n_chunks; ++i) {
            const Chunk &ch = plan.chunks[i];
            hc[i].first = ch.first;
            hc[i].count = ch.n_reads ? ch.last - ch.first + 1 : 0;
        }
        WidePlan wide = wide_plan(c, n_chunks, slice_gw);
        if (wide.nw && ensure_wide_tables(c, wide)) wide = WidePlan();   // no room for the wide tables: groups of 256 chunks as before
        if (ensure_slice_buffers(c, wide.nw ? 8 : slice_gw, n_chunks)) rc = 1;   // (sized above already; a wide plan that fell back may need less)
        if (!rc && hipMemcpy(c->d_slice_chunks, hc.data(), n_chunks * sizeof(SliceChunk), hipMemcpyHostToDevice) != hipSuccess)
            rc = fail("chunk descriptor upload failed");
        // Wide rows or narrow tables?  The wide pass looks at EVERY chunk filter for every read; the narrow tables take 256
        // chunks per pass and skip, in later passes, the reads that earlier ones have found — 2.5x the cost per chunk and
        // read (configs[4]: 8.3 s against 2.6 s), but when most reads are found early there is little left to pay it on
        // (10 M x 100 bp reads, t = 2: k = 20 narrow 628 ms / wide 850 ms, k = 18 664 / 1391, k = 16 649 / 1709 — random
        // reads share that many short k-mers — but k = 22 491 / 384, k = 24 307 / 256).  In auto mode the first 64 chunk
        // filters are therefore searched with the narrow tables against a sample of every search set (one 64-read word in 128 of a large set); with
        // p = the share of them that a group of 256 chunks would find at that rate, the reads still unfound after g groups
        // are taken as (1 - p)^g of the set, a narrow pass is priced at 3.7x a wide one per chunk and read (the largest
        // ratio measured: reads that are found leave the narrow kernel early, too), and the cheaper plan runs.  The probe's
        // reads are searched for real (tags and counters): whichever plan follows skips the found ones and finds nothing
        // new in those chunks for the others.
        if (!rc && wide.nw && c->slice_wide == 0) {
            const int g0 = (int) std::min<uint64_t>(64, n_chunks);
            if (launch_slice_build(c, index_rs, plan.dense ? nullptr : index_rs->d_sel, 0, g0, 2)) rc = 1;
            n_index_launches += 2;
            uint64_t sampled = 0;
            std::vector<uint64_t> smp;
            for (int s = 0; s < n_search && !rc; ++s) {
                const commet_readset *rs = search_rs[s];
                if (!rs->n_reads) continue;
                const uint64_t nw64 = bitmap_words(rs->n_reads);
                smp.assign(nw64, 0);
                const uint64_t *vw = all_visited[s] ? nullptr : (const uint64_t *) vis[s].data();   // (n/8+1 bytes: the last word may be partial)
                const uint64_t stride = rs->n_reads >= (4ull << 20) ? 128 : rs->n_reads >= (1ull << 20) ? 32 : 8;   // >= ~16 k sampled reads
                // (a block of the kernel is 4 words: only the blocks that hold a sampled word are launched)
                for (uint64_t w = 0; w < nw64; w += stride) {
                    uint64_t bits = ~0ull;
                    if (vw) {
                        bits = 0;
                        const uint64_t nbytes = bitmap_bytes_host(rs->n_reads), o = w * 8;
                        memcpy(&bits, vis[s].data() + o, (size_t) std::min<uint64_t>(8, nbytes > o ? nbytes - o : 0));
                    }
                    if (w * 64 >= rs->n_reads) bits = 0;                                              // (bitmaps have a spare word)
                    else if (rs->n_reads - w * 64 < 64) bits &= (1ull << (rs->n_reads - w * 64)) - 1ull;   // reads past the end
                    smp[w] = bits;
                    sampled += (uint64_t) __builtin_popcountll(bits);
                }
                if (hipMemcpyAsync(rs->d_found, smp.data(), nw64 * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                    hipStreamSynchronize(c->stream) != hipSuccess) { rc = fail("probe bitmap upload failed"); break; }   // (smp is reused)
                if (launch_search_sliced(c, rs, g0, 2, rs->d_found, rs->d_tags, d_cnt + 2 * (uint64_t) s, (uint32_t) (2 * n_search), (uint32_t) (stride / 4))) { rc = 1; break; }
                ++n_search_launches;
            }
            std::vector<unsigned long long> pc((size_t) 2 * g0 * n_search);
            if (!rc && (hipMemcpyAsync(pc.data(), d_cnt, pc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                        hipStreamSynchronize(c->stream) != hipSuccess)) rc = fail("probe counter copy failed");
            if (!rc) {
                uint64_t found = 0;
                for (size_t i = 1; i < pc.size(); i += 2) found += pc[i];
                const double p0 = sampled ? std::min(1.0, (double) found / (double) sampled) : 0.0;   // found in g0 chunks
                const double pf = 1.0 - std::pow(1.0 - p0, 256.0 / (double) g0);                       // ... in a group of 256, at that rate
                const uint64_t groups = (n_chunks + 255) / 256;
                double left = 1.0, narrow_cost = 0.0;
                for (uint64_t gi = 0; gi < groups; ++gi) narrow_cost += 3.7 * 256.0 * left, left *= 1.0 - pf;
                if (narrow_cost < (double) n_chunks) wide = WidePlan();   // most reads are found early: the narrow tables, group by group
            }
        }
        // wide rows: the filters of a pass's chunks (all of them when the tables fit) are built 256 at a time into their
        // columns of the rows, then every search set is scanned ONCE per pass
        for (uint64_t c0 = 0; wide.nw && c0 < n_chunks && !rc; c0 += wide.chunks_per_pass) {
            const uint64_t c1 = std::min<uint64_t>(n_chunks, c0 + wide.chunks_per_pass);
            hipEvent_t a = nullptr, b = nullptr;
            if (timed) {
                if (new_event(&a) || new_event(&b)) { rc = 1; break; }
                (void) hipEventRecord(a, c->stream);
            }
            for (uint64_t ci = c0; ci < c1 && !rc; ci += 256) {
                const int g = (int) std::min<uint64_t>(256, c1 - ci);
                if (launch_slice_build(c, index_rs, plan.dense ? nullptr : index_rs->d_sel, ci, g, 8, c->wide_tables, wide.rw,
                                       (uint32_t) ((ci - c0) / 256 * WIDE_GROUP_WORDS))) rc = 1;
                n_index_launches += 2;
            }
            if (rc) break;
            if (timed) {
                (void) hipEventRecord(b, c->stream);
                e_idx0.push_back(a);
                e_idx1.push_back(b);
            }
            for (int s = 0; s < n_search && !rc; ++s) {
                const commet_readset *rs = search_rs[s];
                if (launch_search_wide(c, rs, wide, (int) (c1 - c0), all_visited[s] ? nullptr : rs->d_sel, rs->d_tags,
                                       d_cnt + 2 * (c0 * n_search + s), (uint32_t) (2 * n_search))) { rc = 1; break; }
                if (rs->n_reads) ++n_search_launches;
                if (timed) {
                    hipEvent_t d = nullptr;
                    if (new_event(&d)) { rc = 1; break; }
                    (void) hipEventRecord(d, c->stream);
                    e_set[s].push_back(d);
                }
            }
        }
        const uint64_t G = 32ull * slice_gw;
        for (uint64_t ci = 0; ci < n_chunks && !rc && !wide.nw; ci += G) {
            const int g = (int) std::min<uint64_t>(G, n_chunks - ci);
            hipEvent_t a = nullptr, b = nullptr;
            if (timed) {
                if (new_event(&a) || new_event(&b)) { rc = 1; break; }
                (void) hipEventRecord(a, c->stream);
            }
            if (launch_slice_build(c, index_rs, plan.dense ? nullptr : index_rs->d_sel, ci, g, slice_gw)) { rc = 1; break; }
            n_index_launches += 2;
            if (timed) {
                (void) hipEventRecord(b, c->stream);
                e_idx0.push_back(a);
                e_idx1.push_back(b);
            }
            for (int s = 0; s < n_search && !rc; ++s) {
                const commet_readset *rs = search_rs[s];
                if (launch_search_sliced(c, rs, g, slice_gw, all_visited[s] ? nullptr : rs->d_sel, rs->d_tags,
                                         d_cnt + 2 * (ci * n_search + s), (uint32_t) (2 * n_search))) { rc = 1; break; }
                if (rs->n_reads) ++n_search_launches;
                if (timed) {
                    hipEvent_t d = nullptr;
                    if (new_event(&d)) { rc = 1; break; }
                    (void) hipEventRecord(d, c->stream);
                    e_set[s].push_back(d);
                }
            }
        }
    }
    for (uint64_t ci = 0; ci < n_chunks && !rc && !slice_gw;) {
        int g = (int) std::min<uint64_t>((uint64_t) group_cap, n_chunks - ci);
        const int gs = g <= 2 ? 2 : g <= 4 ? 4 : 8;
        if (g > 1 && ensure_slots(c, g, gs)) {   // not enough memory for the group
            (void) hipGetLastError();
            if (g > 4) {                          // eight slots do not fit: groups of four
                group_cap = 4;
                continue;
            }
            g = 1;                                // one chunk at a time
            group_cap = 1;
        }
        hipEvent_t a = nullptr, b = nullptr;
        if (timed) {
            if (new_event(&a) || new_event(&b)) { rc = 1; break; }
            (void) hipEventRecord(a, c->stream);
        }
        // two lanes: when every chunk of the group takes the bucketed construction (which writes all of its filter
        // slot itself), odd chunks are built on the second stream with the second workspace, beside the even ones
        bool lanes = g > 1 && c->index_lanes > 1 && !c->kclock.on;   // per-kernel times are additive on one stream only
        for (int i = 0; i < g && lanes; ++i) {
            const Chunk &ch = plan.chunks[ci + i];
            lanes = ch.n_reads && would_partition(c, index_rs, ch.kmers);
        }
        if (lanes) {   // the second stream starts behind everything issued so far (the previous group's searches read the slots)
            if (hipEventRecord(c->ev_fork, c->stream) != hipSuccess || hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0) != hipSuccess) {
                rc = fail("stream fork failed");
                break;
            }
        }
        for (int i = 0; i < g && !rc; ++i) {
            const Chunk &ch = plan.chunks[ci + i];
            c->cur_slot = i;
            hipEvent_t z0 = nullptr, z1 = nullptr;
            if (timed) {
                if (new_event(&z0) || new_event(&z1)) { rc = 1; break; }
                (void) hipEventRecord(z0, c->stream);
            }
            // new BloomFilter per chunk: zero it, unless the bucketed build is going to write every tile anyway
            const bool self_zeroing = ch.n_reads && would_partition(c, index_rs, ch.kmers);
            if (!self_zeroing && commet_filter_reset(c)) { rc = 1; break; }
            if (timed) {
                (void) hipEventRecord(z1, c->stream);
                e_zero0.push_back(z0);
                e_zero1.push_back(z1);
            }
            if (ch.n_reads) {
                if (launch_index(c, index_rs, ch.first, ch.last - ch.first + 1, plan.dense ? nullptr : index_rs->d_sel, nullptr, ch.kmers, true, !self_zeroing,
                                 lanes ? (i & 1) : 0, d_ids, d_ids ? chunk_pos[ci + i] : 0, ch.n_reads)) { rc = 1; break; }
                ++n_index_launches;
            }
        }
        if (lanes && !rc) {
            if (hipEventRecord(c->ev_join, c->aux_stream) != hipSuccess || hipStreamWaitEvent(c->stream, c->ev_join, 0) != hipSuccess)
                rc = fail("stream join failed");
        }
        if (rc) break;
        if (g > 1 && launch_interleave(c, g, gs)) { rc = 1; break; }
        if (timed) {
            (void) hipEventRecord(b, c->stream);
            e_idx0.push_back(a);
            e_idx1.push_back(b);
        }
        for (int s = 0; s < n_search && !rc; ++s) {
            const commet_readset *rs = search_rs[s];
            unsigned long long *cnt = d_cnt + 2 * (ci * n_search + s);
            // the tiled search (tile_search.hpp) of one pass: 0 = launched, 1 = not for this set / group, 2 = error.  The set's
            // query list is made or found, and its kernels queued, under ql_mu: no other thread gives the list back in between
            auto try_tiled = [&](int tg, int slot0, unsigned long long *tcnt) -> int {
                std::lock_guard<std::mutex> qlk(c->ql_mu);
                if (!tiled_ok(c, rs, tg) || build_query_list(c, rs) != 0 || ensure_query_results(c, rs) != 0) return 1;
                return launch_search_tiled(c, rs, tg, slot0, all_visited[s] ? nullptr : rs->d_sel, rs->d_tags, tcnt, (uint32_t) (2 * n_search)) ? 2 : 0;
            };
            const int tiled2 = g == 2 ? try_tiled(2, 0, cnt) : 1;   // large set, two chunk filters: lane-a gathers served from L2, slice by slice
            if (tiled2 == 2) { rc = 1; break; }
            if (tiled2 == 0) {
                if (rs->n_reads) ++n_search_launches;
            } else if (g > 1 && (gs == 8 || group_searchable(c, rs, g))) {
                if (launch_search_group(c, rs, g, gs, all_visited[s] ? nullptr : rs->d_sel, rs->d_tags, cnt, (uint32_t) (2 * n_search), d_probes)) { rc = 1; break; }
                if (rs->n_reads) ++n_search_launches;
            } else {
                for (int i = 0; i < g && !rc; ++i) {
                    c->cur_slot = i;
                    const int tiled1 = try_tiled(1, i, cnt + 2 * (uint64_t) i * n_search);   // the same, one filter at a time
                    if (tiled1 == 2) rc = 1;
                    else if (tiled1 == 1 &&
                             launch_search(c, rs, all_visited[s] ? nullptr : rs->d_sel, rs->d_tags, nullptr, cnt + 2 * (uint64_t) i * n_search, d_probes)) rc = 1;
                    if (rs->n_reads) ++n_search_launches;
                }
            }
            if (timed && !rc) {
                hipEvent_t d = nullptr;
                if (new_event(&d)) { rc = 1; break; }
                (void) hipEventRecord(d, c->stream);
                e_set[s].push_back(d);
            }
        }
        c->cur_slot = 0;
        ci += (uint64_t) g;
    }
    c->cur_slot = 0;
    lap(ph_launch);
    if (!rc)
        if (hipMemcpyAsync(h_cnt.data(), d_cnt, n_cnt * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream) != hipSuccess)
            rc = fail("counter copy failed");
    for (int s = 0; s < n_search && !rc; ++s) {
        const commet_readset *rs = search_rs[s];
        if (tags_out && tags_out[s])
            if (hipMemcpyAsync(tags_out[s], rs->d_tags, bitmap_bytes_host(rs->n_reads), hipMemcpyDeviceToHost, c->stream) != hipSuccess)
                rc = fail("tag copy failed");
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = fail("stream synchronize failed: %s", hipGetErrorString(hipGetLastError()));
    c->kclock.collect();
    lap(ph_wait);
    if (!rc && d_ids && (c->h_counters[N_COUNTERS - 1] & 0xFFFFFFFFull) != (ids_expected & 0xFFFFFFFFull))
        rc = fail("internal error: the selection list holds %llu reads, the plan indexes %llu", (unsigned long long) (c->h_counters[N_COUNTERS - 1] & 0xFFFFFFFFull),
                  (unsigned long long) ids_expected);

    if (!rc) {
        uint64_t scans = 0;
        for (int s = 0; s < n_search; ++s) {
            uint64_t shared = 0, last_scanned = 0;
            for (uint64_t ci = 0; ci < n_chunks; ++ci) {
                const unsigned long long *p = &h_cnt[2 * (ci * n_search + s)];
                // an empty search set launches nothing: scanned = visited - found so far
                last_scanned = visited[s] - shared;
                scans += last_scanned;
                if (search_rs[s]->n_reads && !slice_gw && p[0] != last_scanned)   // (the sliced kernel counts found reads only)
                    rc = fail("internal error: device scanned %llu reads, host plan says %llu (chunk %llu, set %d)",
                              p[0], (unsigned long long) last_scanned, (unsigned long long) ci, s);
                shared += p[1];
            }
            if (stats) {
                stats[s].indexed = plan.indexed_reads;
                stats[s].searched = n_chunks ? last_scanned : 0;
                stats[s].shared = shared;
                stats[s].search_ms = 0;
            }
        }
        double idx_ms = 0, srch_ms = 0, zero_ms = 0;
        if (timed && !rc) {
            for (size_t i = 0; i < e_zero0.size(); ++i) {
                float ms = 0;
                if (hipEventElapsedTime(&ms, e_zero0[i], e_zero1[i]) == hipSuccess) zero_ms += ms;
            }
            for (size_t i = 0; i < e_idx0.size(); ++i) {
                float ms = 0;
                if (hipEventElapsedTime(&ms, e_idx0[i], e_idx1[i]) == hipSuccess) idx_ms += ms;
                for (int s = 0; s < n_search; ++s) {
                    if (i >= e_set[s].size()) continue;
                    hipEvent_t prev = s == 0 ? e_idx1[i] : e_set[s - 1][i];
                    if (hipEventElapsedTime(&ms, prev, e_set[s][i]) == hipSuccess) {
                        srch_ms += ms;
                        if (stats) stats[s].search_ms += ms;
                    }
                }
            }
        }
        if (info) {
            info->n_chunks = n_chunks;
            info->kmers_indexed = plan.kmers;
            info->reads_scanned = scans;
            info->reads_indexed = plan.indexed_reads;
            info->index_launches = n_index_launches;
            info->search_launches = n_search_launches;
            info->probes = h_cnt[n_cnt - 1];
            info->zero_ms = zero_ms;
            info->index_ms = idx_ms;
            info->index_kernel_ms = idx_ms - zero_ms;
            info->search_ms = srch_ms;
        }
    }
    for (hipEvent_t e : evs) (void) hipEventDestroy(e);
    if (job_verbose) {
        double ph_tail = 0;
        lap(ph_tail);
        fprintf(stderr, "[job] plan %.2f ms, bitmap upload %.2f ms, launches %.2f ms, wait + download %.2f ms, stats + cleanup %.2f ms\n",
                ph_plan, ph_upload, ph_launch, ph_wait, ph_tail);
    }
    if (info && !rc)
        info->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    return rc;
}

/* ---- hooks ---------------------------------------------------------------- */

uint64_t commet_readset_cache_bytes(const commet_readset *rs)
{
    std::lock_guard<std::mutex> lk(rs->ctx->ql_mu);
    return rs->ql.built ? rs->ql.bytes : 0;
}

void commet_readset_drop_cache(commet_readset *rs)
{
    commet_ctx *c = rs->ctx;
    (void) hipSetDevice(c->device);
    std::lock_guard<std::mutex> lk(c->ql_mu);
    if (rs->in_job) return;                              // (never under a running job)
    drop_query_list(c, rs);
    rs->ql.failed = false;                               // a list that did not fit once may fit now
}

int commet_cache_stats(commet_ctx *c, uint64_t *bytes, uint64_t *budget_bytes, uint64_t *evictions)
{
    std::lock_guard<std::mutex> lk(c->ql_mu);
    if (bytes) *bytes = c->ql_bytes;
    if (budget_bytes) *budget_bytes = c->ql_budget;
    if (evictions) *evictions = c->ql_evictions;
    return 0;
}

int commet_set_option(commet_ctx *c, const char *name, int64_t value)
{
    if (!strcmp(name, "query_list_max_mb")) {      // auto mode: largest list (estimated) a set may get; larger sets keep the gather kernels
        if (value < 0) return fail("query_list_max_mb must be >= 0");
        c->ql_max_list = (uint64_t) value << 20;
        return 0;
    }
    if (!strcmp(name, "query_list_budget_mb")) {   // HBM the cached query lists of this context's read sets may hold (default 64 GiB)
        if (value < 0) return fail("query_list_budget_mb must be >= 0");
        std::lock_guard<std::mutex> lk(c->ql_mu);
        c->ql_budget = (uint64_t) value << 20;
        (void) shrink_query_lists(c, c->ql_budget, false);
        return 0;
    }
    if (!strcmp(name, "count_probes")) {
        c->count_probes = value != 0;
        return 0;
    }
    if (!strcmp(name, "index_mode")) {        // 0 auto, 1 atomic kernel, 2 bucketed construction
        if (value < 0 || value > 2) return fail("index_mode must be 0, 1 or 2");
        c->index_mode = (int) value;
        return 0;
    }
    if (!strcmp(name, "chunk_group")) {       // chunk filters searched per pass over a set (1 = reference order)
        if (value < 1 || value > 8) return fail("chunk_group must be 1..8");
        c->chunk_group = (int) value;
        return 0;
    }
    if (!strcmp(name, "kernel_timing")) {     // 1: time every kernel of the following commet_index_and_search calls (totals reset)
        HIP_OK(hipSetDevice(c->device));
        HIP_OK(hipStreamSynchronize(c->stream));
        c->kclock.collect();
        c->kclock.on = value != 0;
        if (value) c->kclock.reset();
        return 0;
    }
    if (!strcmp(name, "tiled_search")) {      // 0 auto, 1 never, 2 whenever the set and the group allow it (tests)
        if (value < 0 || value > 2) return fail("tiled_search must be 0, 1 or 2");
        c->tiled_mode = (int) value;
        return 0;
    }
    if (!strcmp(name, "tq_parts")) {          // tiled search: parts of the set whose replay runs beside the next part's probe (1 = off)
        if (value < 1 || value > 16) return fail("tq_parts must be 1..16");
        c->tq_parts = (int) value;
        return 0;
    }
    if (!strcmp(name, "slice_mode")) {        // 0 auto (8 chunks or more, 12 <= k <= 24), 1 never, 2 whenever k allows it
        if (value < 0 || value > 2) return fail("slice_mode must be 0, 1 or 2");
        c->slice_mode = (int) value;
        return 0;
    }
    if (!strcmp(name, "slice_words")) {       // chunk filters per pass / 32 in the sliced regime: 0 auto, 1, 2, 4 or 8
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return fail("slice_words must be 0, 1, 2, 4 or 8");
        c->slice_gw = (int) value;
        return 0;
    }
    if (!strcmp(name, "slice_wide")) {        // wide rows in the many-small-chunks regime: 0 auto (more than 256 chunks), 1 never, 2 always
        if (value < 0 || value > 2) return fail("slice_wide must be 0, 1 or 2");
        c->slice_wide = (int) value;
        return 0;
    }
    if (!strcmp(name, "slice_wide_words")) {  // cap on the words per wide row (32 chunk filters each; tests: several passes); 0 = the budget decides
        if (value < 0 || value > 512 || value % 8) return fail("slice_wide_words must be a multiple of 8 in 0..512");
        c->wide_cap_words = (uint32_t) value;
        return 0;
    }
    if (!strcmp(name, "max_kmer")) {          // chunk size in k-mers (0 = the reference's constant); changes the chunking
        if (value < 0) return fail("max_kmer must be >= 0");
        c->max_kmer_test = (uint64_t) value;
        return 0;
    }
    if (!strcmp(name, "index_lanes")) {       // 1 = the chunks of a group are built one after the other
        if (value < 1 || value > 2) return fail("index_lanes must be 1 or 2");
        c->index_lanes = (int) value;
        return 0;
    }
    if (!strcmp(name, "part_packed")) {
        c->part_packed = value != 0;
        return 0;
    }
    if (!strcmp(name, "part_no_uni")) {
        c->part_no_uni = value != 0;
        return 0;
    }
    if (!strcmp(name, "part_b1")) {
        c->part_b1 = (int) value;
        return 0;
    }
    if (!strcmp(name, "s2_swizzle")) {
        c->s2_swizzle = (int) std::max<int64_t>(0, std::min<int64_t>(value, 1 << 20));
        return 0;
    }
    if (!strcmp(name, "part_min_kmers")) {    // auto mode: chunks with fewer k-mers use the atomic kernel
        c->part_min_kmers = (uint64_t) value;
        return 0;
    }
    return fail("unknown option '%s'", name);
}

int commet_filter_export_reference(commet_ctx *c, uint8_t *out, uint64_t out_bytes)
{
    const uint64_t nbytes = (uint64_t) pow(2, c->k - 1);   // bloom_filter.h:73
    if (out_bytes < nbytes) return fail("export buffer too small");
    if (nbytes == 0) return 0;
    HIP_OK(hipSetDevice(c->device));
    uint8_t *d_out = nullptr;
    HIP_OK(hipMalloc((void **) &d_out, nbytes));
    const uint64_t blocks = std::min<uint64_t>((nbytes + 255) / 256, 1u << 20);   // grid-stride beyond
    COMMET_LAUNCH(export_reference_kernel, dim3((unsigned) blocks), dim3(256), 0, c->stream, c->view(), c->k, nbytes, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, nbytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void) hipFree(d_out);
    if (e != hipSuccess) return fail("filter export failed: %s", hipGetErrorString(e));
    return 0;
}

int commet_launched_kernels(const void **out, int cap, int *n_out)
{
    std::lock_guard<std::mutex> lk(g_launch_mu);
    int i = 0;
    for (const void *e : g_launched) {
        if (i < cap && out) out[i] = e;
        ++i;
    }
    if (n_out) *n_out = i;
    return 0;
}

int commet_last_kernel_ms(commet_ctx *c, double *index_ms, double *search_ms)
{
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipStreamSynchronize(c->stream));
    float ms = 0;
    if (index_ms) {
        *index_ms = 0;
        if (c->have_index_ev) {
            HIP_OK(hipEventElapsedTime(&ms, c->ev_i0, c->ev_i1));
            *index_ms = ms;
        }
    }
    if (search_ms) {
        *search_ms = 0;
        if (c->have_search_ev) {
            HIP_OK(hipEventElapsedTime(&ms, c->ev_s0, c->ev_s1));
            *search_ms = ms;
        }
    }
    return 0;
}

int commet_kernel_times(commet_ctx *c, commet_kernel_time *out, int cap, int *n_out)
{
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipStreamSynchronize(c->stream));
    c->kclock.collect();
    const int n = (int) c->kclock.names.size();
    if (n_out) *n_out = n;
    for (int i = 0; i < n && i < cap; ++i) {
        snprintf(out[i].name, sizeof out[i].name, "%s", c->kclock.names[i].c_str());
        out[i].launches = c->kclock.launches[i];
        out[i].total_ms = c->kclock.total_ms[i];
    }
    return 0;
}

int commet_membench(commet_ctx *c, int atomic, uint64_t table_bytes, uint64_t n_access, double *ms_out)
{
    HIP_OK(hipSetDevice(c->device));
    if (atomic == 4 || atomic == 5) {   // streaming ceilings: 4 = device-to-device copy of table_bytes, 5 = fill
        uint8_t *a = nullptr, *b = nullptr;
        HIP_OK(hipMalloc((void **) &a, table_bytes));
        HIP_OK(hipMalloc((void **) &b, table_bytes));
        hipEvent_t e0, e1;
        HIP_OK(hipEventCreate(&e0));
        HIP_OK(hipEventCreate(&e1));
        HIP_OK(hipMemsetAsync(a, 1, table_bytes, c->stream));
        HIP_OK(hipMemsetAsync(b, 2, table_bytes, c->stream));
        for (int rep = 0; rep < 2; ++rep) {
            HIP_OK(hipEventRecord(e0, c->stream));
            if (atomic == 4) HIP_OK(hipMemcpyAsync(b, a, table_bytes, hipMemcpyDeviceToDevice, c->stream));
            else HIP_OK(hipMemsetAsync(b, 3, table_bytes, c->stream));
            HIP_OK(hipEventRecord(e1, c->stream));
        }
        HIP_OK(hipStreamSynchronize(c->stream));
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        if (ms_out) *ms_out = ms;
        (void) hipEventDestroy(e0);
        (void) hipEventDestroy(e1);
        (void) hipFree(a);
        (void) hipFree(b);
        return 0;
    }
    if (atomic >= 100) {   // windowed gathers: atomic = 100 + log2(window bytes), +1000 = XCD-aware sweep; n_access gathers in all
        const int xcd = atomic >= 1000 ? 1 : 0;
        const uint32_t win_words = (1u << ((atomic % 1000) - 100)) / 4;
        const uint64_t n_windows = table_bytes / 4 / win_words;
        uint32_t *table = nullptr, *sink = nullptr;
        HIP_OK(hipMalloc((void **) &table, n_windows * win_words * 4));
        HIP_OK(hipMalloc((void **) &sink, 4));
        HIP_OK(hipMemsetAsync(table, 0, n_windows * win_words * 4, c->stream));
        // one resident set of workgroups (8 per CU); every thread does `iters` gathers in each window of its XCD's eighth
        const uint32_t grid = 256 * 8;
        const uint32_t iters = (uint32_t) std::max<uint64_t>(1, n_access / (grid * 256ull) / std::max<uint64_t>(n_windows / 8, 1));
        hipEvent_t e0, e1;
        HIP_OK(hipEventCreate(&e0));
        HIP_OK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            HIP_OK(hipEventRecord(e0, c->stream));
            COMMET_LAUNCH(membench_window_kernel, dim3(grid), dim3(256), 0, c->stream, table, n_windows, win_words, iters, xcd, sink);
            HIP_OK(hipEventRecord(e1, c->stream));
        }
        HIP_OK(hipStreamSynchronize(c->stream));
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        if (ms_out) *ms_out = ms / ((double) iters * (double) (n_windows / 8) * grid * 256.0) * (double) n_access;
        (void) hipEventDestroy(e0);
        (void) hipEventDestroy(e1);
        (void) hipFree(table);
        (void) hipFree(sink);
        return 0;
    }
    uint64_t words = 1;
    while (words * 2 * 4 <= table_bytes) words *= 2;   // power of two words
    uint32_t *table = nullptr, *sink = nullptr;
    HIP_OK(hipMalloc((void **) &table, words * 4));
    HIP_OK(hipMalloc((void **) &sink, 4));
    HIP_OK(hipMemsetAsync(table, 0, words * 4, c->stream));
    const uint64_t threads = 256ull * 256 * 32;   // 32 blocks of 256 per CU
    const uint32_t iters = (uint32_t) std::max<uint64_t>(1, n_access / threads);
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {   // rep 0 warms up
        HIP_OK(hipEventRecord(e0, c->stream));
        const dim3 g((unsigned) (threads / 256)), b(256);
        switch (atomic) {
        case 1: COMMET_LAUNCH(membench_kernel<1>, g, b, 0, c->stream, table, words - 1, iters, sink); break;
        case 2: COMMET_LAUNCH(membench_kernel<2>, g, b, 0, c->stream, table, words - 1, iters, sink); break;
        case 3: COMMET_LAUNCH(membench_kernel<3>, g, b, 0, c->stream, table, words - 1, iters, sink); break;
        default: COMMET_LAUNCH(membench_kernel<0>, g, b, 0, c->stream, table, words - 1, iters, sink); break;
        }
        HIP_OK(hipEventRecord(e1, c->stream));
    }
    HIP_OK(hipStreamSynchronize(c->stream));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    if (ms_out) *ms_out = ms / ((double) iters * threads) * (double) n_access;   // scaled to n_access
    (void) hipEventDestroy(e0);
    (void) hipEventDestroy(e1);
    (void) hipFree(table);
    (void) hipFree(sink);
    return 0;
}

int commet_ldsbench(commet_ctx *c, int mode, uint32_t n_words, uint64_t n_access, double *ms_out)
{
    HIP_OK(hipSetDevice(c->device));
    if (n_words == 0 || (n_words & (n_words - 1)) || n_words > 32768) return fail("ldsbench: n_words must be a power of two <= 32768");
    if (mode < 0 || mode > 5) return fail("ldsbench: mode 0..5");
    uint32_t *sink = nullptr;
    HIP_OK(hipMalloc((void **) &sink, 4));
    const uint64_t threads = 512ull * 256 * 8;   // 8 workgroups of 512 per CU (LDS permitting)
    const uint32_t iters = (uint32_t) std::max<uint64_t>(1, n_access / threads);
    const size_t lds = (size_t) n_words * 4;
    const void *fns[6] = {(const void *) ldsbench_kernel<0>, (const void *) ldsbench_kernel<1>, (const void *) ldsbench_kernel<2>,
                          (const void *) ldsbench_kernel<3>, (const void *) ldsbench_kernel<4>, (const void *) ldsbench_kernel<5>};
    HIP_OK(hipFuncSetAttribute(fns[mode], hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {   // rep 0 warms up
        HIP_OK(hipEventRecord(e0, c->stream));
        uint32_t it = iters;
        void *args[] = {&n_words, &it, &sink};
        HIP_OK(hipLaunchKernel(fns[mode], dim3((unsigned) (threads / 512)), dim3(512), args, lds, c->stream));
        HIP_OK(hipEventRecord(e1, c->stream));
    }
    HIP_OK(hipStreamSynchronize(c->stream));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    if (ms_out) *ms_out = ms / ((double) iters * threads) * (double) n_access;
    (void) hipEventDestroy(e0);
    (void) hipEventDestroy(e1);
    (void) hipFree(sink);
    return 0;
}

}  // extern "C"

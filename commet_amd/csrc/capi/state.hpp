// capi/state.hpp — what every part of the C-ABI implementation shares: error reporting, launch bookkeeping, the context and
// read-set objects behind the opaque handles of include/commet_hip.h, the per-launch timing scope.
// (a part of the one translation unit capi.hip: included there first, after the kernel headers)
#pragma once

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
    // (the sliced / wide regimes issue thousands of launches per job, from two host threads: only an entry point this thread has
    // not noted lately takes the lock)
    thread_local const void *recent[16] = {nullptr};
    thread_local unsigned next = 0;
    for (const void *r : recent)
        if (r == entry) return;
    recent[next++ & 15u] = entry;
    std::lock_guard<std::mutex> lk(g_launch_mu);
    g_launched.insert(entry);
}
// a call site notes its kernel once (a template's call site: once per instantiation)
#define COMMET_LAUNCH(kernel, ...)                                   \
    do {                                                              \
        static std::atomic<bool> noted_{false};                       \
        if (!noted_.load(std::memory_order_relaxed)) {                \
            note_launch((const void *) (kernel));                     \
            noted_.store(true, std::memory_order_relaxed);            \
        }                                                             \
        hipLaunchKernelGGL(kernel, __VA_ARGS__);                      \
    } while (0)

// ---- device memory kept for reuse ----------------------------------------------------------------------------------------
// On this driver a hipMalloc of GBs costs 15-30 ms per GiB on some boxes (and ~0 on others: tools/exp/alloc_cost*.hip — where a
// 16 GiB hipMalloc took 483 ms, and 965 ms again after its hipFree, sixteen of 1 GiB took 0.3 ms and a hipMallocAsync of 16 GiB
// 14 ms), and one that follows the hipFree of tens of GB now and then blocks for 50-150 ms (seen up to 2 s): a context created after another one was closed, a query list built after one was dropped, a
// workspace that grows — all of them on some job's path (bench.py's configs[2] leg behind the headline: 1.50 s of device time
// against 1.15 s in a fresh process, the difference being the GPU idling behind such calls; profiles/r05_c2_variance).  So the
// library never gives a block of 8 MiB or more back while the process lives: dm_free waits for the device, as hipFree does (its
// callers count on that), and files the block; dm_malloc takes the smallest filed block of the current device that is large enough
// and at most a quarter larger, else asks hipMalloc — and when hipMalloc is out of memory, everything filed is given back and
// it is asked once more.  Blocks are never split (an exported set's IPC handle must be that of a whole allocation).
// COMMET_DEVMEM_POOL=1 (off by default): a new block of 256 MiB or more comes from hipMallocAsync on a stream of the cache's own,
// drained before the block is handed out — 0.4-0.9 ms per GiB on either kind of box in isolation, but on a box whose hipMalloc
// is free anyway the 10-set matrix of bench.py took 12.7 s with it against 10.9 s without (device time +4 %: kernels gather more
// slowly from pool memory; the loader thread's waits 1.5 s against 0.6 s; profiles/r05_pool), and the box of the other kind has
// not come up again to be measured.  No IPC handle can be had for such a block, so commet_readset_export first moves a set's
// planes into a hipMalloc block (dm_make_shareable).
// At most half the device (COMMET_DEVMEM_CACHE_GB) is kept, the largest blocks going first; commet_device_cache_trim gives all of
// it back.  COMMET_DEVMEM_CACHE=0: plain hipMalloc / hipFree.
struct DevMemCache {
    struct Block {
        int device;
        size_t bytes;                                              // as allocated
        bool pooled;                                               // from hipMallocAsync (no IPC handle can be had for it)
    };
    std::mutex mu;
    std::map<void *, Block> live;                                  // blocks handed out
    std::multimap<size_t, std::pair<void *, bool>> filed[16];      // per device, by size: pointer, pooled
    size_t filed_bytes[16] = {0};
    size_t cap[16] = {0};                                          // bytes kept at most per device (COMMET_DEVMEM_CACHE_GB; default: half the device)
    hipStream_t pool_stream[16] = {nullptr};                       // the stream the stream-ordered allocations are ordered on (always drained before use)
    int on = -1, pool_on = -1;
    size_t pool_min = 0;                                           // COMMET_DEVMEM_POOL_MIN_MB (tests: a set of a few MB in pooled blocks); never below 8 MiB
    // what the DRIVER was asked for, per device (commet_device_alloc_stats): time the calling thread spent inside hipMalloc /
    // hipMallocAsync, bytes and calls — a box that charges for a process's first use of device memory shows here, not in kernel time
    std::atomic<uint64_t> drv_ns[16], drv_bytes[16], drv_calls[16];
    std::atomic<uint64_t> trims{0};                                // times filed blocks went back to the driver (dm_trim): what was set aside before is gone
    DevMemCache()
    {
        for (int d = 0; d < 16; ++d) drv_ns[d] = 0, drv_bytes[d] = 0, drv_calls[d] = 0;
    }
    bool enabled()
    {
        if (on < 0) {
            const char *e = getenv("COMMET_DEVMEM_CACHE");
            on = !(e && atoi(e) == 0);
        }
        return on != 0;
    }
    int verb = -1;
    bool verbose()                                                 // COMMET_DEVMEM_VERBOSE=1: one line on stderr per block asked from or returned to the driver
    {
        if (verb < 0) {
            const char *e = getenv("COMMET_DEVMEM_VERBOSE");
            verb = e && atoi(e) != 0;
        }
        return verb != 0;
    }
    bool pooling()
    {
        if (pool_on < 0) {
            const char *e = getenv("COMMET_DEVMEM_POOL");
            pool_on = e && atoi(e) != 0;
            const char *m = getenv("COMMET_DEVMEM_POOL_MIN_MB");
            pool_min = m ? std::max<size_t>((size_t) atoll(m) << 20, (size_t) 8 << 20) : (size_t) 256 << 20;
        }
        return pool_on != 0;
    }
};
DevMemCache g_devmem;
constexpr size_t DEVMEM_MIN_FILED = (size_t) 8 << 20;

// one block from the driver: stream-ordered from 256 MiB on unless an IPC handle will be asked for it
hipError_t dm_driver_alloc_untimed(void **p, size_t bytes, int dev, bool shareable, bool *pooled);
hipError_t dm_driver_alloc(void **p, size_t bytes, int dev, bool shareable, bool *pooled)
{
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = dm_driver_alloc_untimed(p, bytes, dev, shareable, pooled);
    if (dev >= 0 && dev < 16) {
        g_devmem.drv_ns[dev] += (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        g_devmem.drv_calls[dev] += 1;
        if (e == hipSuccess) g_devmem.drv_bytes[dev] += bytes;
    }
    return e;
}
hipError_t dm_driver_alloc_untimed(void **p, size_t bytes, int dev, bool shareable, bool *pooled)
{
    *pooled = false;
    if (!shareable && g_devmem.pooling() && bytes >= g_devmem.pool_min) {
        hipStream_t st;
        {
            std::lock_guard<std::mutex> lk(g_devmem.mu);
            if (!g_devmem.pool_stream[dev] && hipStreamCreateWithFlags(&g_devmem.pool_stream[dev], hipStreamNonBlocking) != hipSuccess) {
                (void) hipGetLastError();
                g_devmem.pool_stream[dev] = nullptr;
            }
            st = g_devmem.pool_stream[dev];
        }
        if (st) {
            hipError_t e = hipMallocAsync(p, bytes, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e == hipSuccess) {
                *pooled = true;
                return e;
            }
            if (e != hipErrorOutOfMemory) (void) hipGetLastError();     // (not supported here, ...: the plain call below)
            else return e;
        }
    }
    return hipMalloc(p, bytes);
}

void dm_driver_free(void *q, int dev, bool pooled)
{
    if (!pooled) {
        (void) hipFree(q);
        return;
    }
    hipStream_t st = g_devmem.pool_stream[dev];
    (void) hipFreeAsync(q, st);
    (void) hipStreamSynchronize(st);
}

// gives every filed block of `device` (-1: all) back to the driver; returns the bytes released
size_t dm_trim(int device)
{
    g_devmem.trims += 1;
    std::vector<std::pair<int, std::pair<void *, bool>>> drop;
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> lk(g_devmem.mu);
        for (int d = 0; d < 16; ++d) {
            if (device >= 0 && d != device) continue;
            for (auto &b : g_devmem.filed[d]) drop.push_back({d, b.second}), bytes += b.first;
            g_devmem.filed[d].clear();
            g_devmem.filed_bytes[d] = 0;
        }
    }
    bool any_pooled = false;
    for (auto &q : drop) dm_driver_free(q.second.first, q.first, q.second.second), any_pooled |= q.second.second;
    if (any_pooled) {                                              // the pool itself keeps nothing either
        int cur = 0;
        const bool have = hipGetDevice(&cur) == hipSuccess;
        for (int d = 0; d < 16; ++d) {
            if ((device >= 0 && d != device) || !g_devmem.pool_stream[d]) continue;
            hipMemPool_t pool;
            if (hipSetDevice(d) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, d) == hipSuccess) (void) hipMemPoolTrimTo(pool, 0);
        }
        if (have) (void) hipSetDevice(cur);
        (void) hipGetLastError();
    }
    return bytes;
}

size_t dm_filed_bytes(int device)
{
    std::lock_guard<std::mutex> lk(g_devmem.mu);
    return device >= 0 && device < 16 ? g_devmem.filed_bytes[device] : 0;
}

size_t dm_pooled_bytes(int device)
{
    std::lock_guard<std::mutex> lk(g_devmem.mu);
    size_t n = 0;
    for (auto &b : g_devmem.live)
        if (b.second.device == device && b.second.pooled) n += b.second.bytes;
    return n;
}

// Size classes: a block of 8 MiB or more is asked for in steps of 1/16 .. 1/32 of its size (a request of 8.5 GB becomes one of 8.6 GB).
// The objects a context makes — scatter workspaces sized by a chunk's k-mer count, a set's planes, a query list — come out a few
// hundred bytes apart from one set to the next; filed under their exact sizes, a block that was 100 bytes short served nobody and
// the next context asked the driver for the same 17 GB again (the first process on a box that charges for fresh device memory pays
// 30 ms per GiB for that, DESIGN section 4: bench.py's matrix legs, commet_device_alloc_stats).
inline size_t dm_size_class(size_t bytes)
{
    if (bytes < DEVMEM_MIN_FILED) return bytes;
    const int lg = 63 - __builtin_clzll((unsigned long long) bytes);
    const size_t granule = (size_t) 1 << (lg - 4);
    return (bytes + granule - 1) & ~(granule - 1);
}

// `shareable`: the block may be exported to another process (commet_readset_export), so it comes from hipMalloc
hipError_t dm_malloc(void **p, size_t bytes, bool shareable = false)
{
    *p = nullptr;
    if (g_devmem.enabled()) bytes = dm_size_class(bytes);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipMalloc(p, bytes);
    if (!g_devmem.enabled()) {
        const auto t0 = std::chrono::steady_clock::now();
        const hipError_t e = hipMalloc(p, bytes);
        g_devmem.drv_ns[dev] += (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        g_devmem.drv_calls[dev] += 1;
        if (e == hipSuccess) g_devmem.drv_bytes[dev] += bytes;
        return e;
    }
    if (bytes >= DEVMEM_MIN_FILED) {
        std::lock_guard<std::mutex> lk(g_devmem.mu);
        for (auto it = g_devmem.filed[dev].lower_bound(bytes); it != g_devmem.filed[dev].end() && it->first <= bytes + bytes / 4; ++it) {
            if (shareable && it->second.second) continue;
            *p = it->second.first;
            g_devmem.filed_bytes[dev] -= it->first;
            g_devmem.live[*p] = {dev, it->first, it->second.second};
            g_devmem.filed[dev].erase(it);
            return hipSuccess;
        }
    }
    bool pooled = false;
    if (g_devmem.verbose()) {
        std::lock_guard<std::mutex> lk(g_devmem.mu);
        fprintf(stderr, "[devmem] driver alloc %.1f MiB (filed: %.1f GiB in %zu blocks)\n", bytes / 1048576.0, g_devmem.filed_bytes[dev] / 1073741824.0, g_devmem.filed[dev].size());
    }
    hipError_t e = dm_driver_alloc(p, bytes, dev, shareable, &pooled);
    if (e == hipErrorOutOfMemory && dm_trim(dev)) {
        (void) hipGetLastError();
        e = dm_driver_alloc(p, bytes, dev, shareable, &pooled);
    }
    if (e == hipSuccess && bytes >= DEVMEM_MIN_FILED) {
        std::lock_guard<std::mutex> lk(g_devmem.mu);
        g_devmem.live[*p] = {dev, bytes, pooled};
    }
    return e;
}

// a block of `bytes` asked from the driver NOW and filed at once, for an allocation that will come later on some job's path (a helper
// thread pays the driver's price instead of the job thread); never taken from the filed blocks themselves
hipError_t dm_reserve(size_t bytes)
{
    if (!g_devmem.enabled() || bytes < DEVMEM_MIN_FILED) return hipSuccess;
    bytes = dm_size_class(bytes);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    void *p = nullptr;
    bool pooled = false;
    const hipError_t e = dm_driver_alloc(&p, bytes, dev, false, &pooled);
    if (e != hipSuccess) {
        (void) hipGetLastError();
        return e;
    }
    std::lock_guard<std::mutex> lk(g_devmem.mu);
    g_devmem.filed[dev].emplace(bytes, std::make_pair(p, pooled));
    g_devmem.filed_bytes[dev] += bytes;
    return hipSuccess;
}

hipError_t dm_free(void *p)
{
    if (!p) return hipSuccess;
    if (g_devmem.enabled()) {
        DevMemCache::Block what{-1, 0, false};
        {
            std::lock_guard<std::mutex> lk(g_devmem.mu);
            auto it = g_devmem.live.find(p);
            if (it != g_devmem.live.end()) what = it->second, g_devmem.live.erase(it);
        }
        if (what.device >= 0) {
            const int d = what.device;
            int cur = d;
            const bool have_cur = hipGetDevice(&cur) == hipSuccess;
            if (have_cur && cur != d) (void) hipSetDevice(d);      // (the block's device, whatever device the calling thread is on)
            (void) hipDeviceSynchronize();                         // what hipFree does: no kernel still reads the block when somebody else gets it
            std::vector<std::pair<void *, bool>> over;             // kept within the cap (other processes may share the device): largest first
            {
                std::lock_guard<std::mutex> lk(g_devmem.mu);
                g_devmem.filed[d].emplace(what.bytes, std::make_pair(p, what.pooled));
                g_devmem.filed_bytes[d] += what.bytes;
                if (!g_devmem.cap[d]) {
                    size_t fr = 0, tot = 0;
                    const char *e = getenv("COMMET_DEVMEM_CACHE_GB");
                    if (e) g_devmem.cap[d] = (size_t) (std::max(0.0, atof(e)) * (double) (1ull << 30)) + 1;
                    else g_devmem.cap[d] = hipMemGetInfo(&fr, &tot) == hipSuccess ? tot / 2 : (size_t) 64 << 30;
                }
                while (g_devmem.filed_bytes[d] > g_devmem.cap[d] && !g_devmem.filed[d].empty()) {
                    auto last = std::prev(g_devmem.filed[d].end());
                    over.push_back(last->second);
                    g_devmem.filed_bytes[d] -= last->first;
                    g_devmem.filed[d].erase(last);
                }
            }
            if (!over.empty() && g_devmem.verbose()) fprintf(stderr, "[devmem] over the cap of %.1f GiB: %zu block(s) back to the driver\n", g_devmem.cap[d] / 1073741824.0, over.size());
            for (auto &q : over) dm_driver_free(q.first, d, q.second);
            if (have_cur && cur != d) (void) hipSetDevice(cur);
            return hipSuccess;
        }
    }
    return hipFree(p);
}

// the block at *p made exportable: one that came from the stream-ordered pool is copied into a hipMalloc block of its own (the
// device drained first; the caller sees to it that no other host thread uses the block meanwhile) and filed
hipError_t dm_make_shareable(void **p)
{
    if (!*p) return hipSuccess;
    DevMemCache::Block what{-1, 0, false};
    {
        std::lock_guard<std::mutex> lk(g_devmem.mu);
        auto it = g_devmem.live.find(*p);
        if (it != g_devmem.live.end()) what = it->second;
    }
    if (what.device < 0 || !what.pooled) return hipSuccess;
    void *q = nullptr;
    hipError_t e = dm_malloc(&q, what.bytes, true);
    if (e != hipSuccess) return e;
    e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(q, *p, what.bytes, hipMemcpyDeviceToDevice);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void) dm_free(q);
        return e;
    }
    (void) dm_free(*p);
    *p = q;
    return hipSuccess;
}

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
    int part_list = 0;                // option: 0 = ragged sets take the item list in hist / scatter1 (index_part.hpp, LIST), 1 = never (the round planner)
    int s2_swizzle = 128;             // scatter2 slab order: number of interleaved slab ranges (index_part.hpp), 0 = dispatch order
    uint64_t part_min_kmers = 8ull << 20;
    // workspaces of the bucketed construction (index_part.hpp): two, so that the chunks of a group can be built on two
    // streams at once (the compute-bound hist / scatter1 of one chunk overlap the HBM-bound scatter2 / build of another)
    struct PartWs {
        uint32_t *bufA = nullptr, *bufB = nullptr;
        uint64_t cap_keys = 0, cap_b = 0;                          // bufA in keys; bufB in words (fewer in the packed geometry)
        uint32_t *hist = nullptr, *wl = nullptr;
        uint64_t *off = nullptr, *goff = nullptr;   // bucket offsets in keys / in 8-byte groups (packed final level)
        unsigned long long *cur2 = nullptr, *blockoff = nullptr;   // final-bucket cursors; scatter1 start positions [workgroup][coarse bucket]
        uint32_t *blockcnt = nullptr;                              // keys per [scatter1 workgroup][coarse bucket]
        uint32_t *items = nullptr, *itemblk = nullptr;             // ragged sets: the chunk's item list and the builder's block sums (index_part.hpp, LIST)
        uint64_t items_cap = 0, itemblk_cap = 0;
        uint64_t items_set = 0, items_first = 0, items_count = 0;  // whose list the buffer holds: (set uid, read range), no selection — 0 = nobody's
        uint32_t items_nblk = 0;
        uint64_t hist_set = 0, hist_first = 0, hist_count = 0;     // (timing bound COMMET_HIST_REUSE only: whose histogram the workspace holds)
        uint32_t nb = 0;
        void release()
        {
            (void) dm_free(bufA); (void) dm_free(bufB); (void) dm_free(hist); (void) dm_free(wl); (void) dm_free(off); (void) dm_free(goff);
            (void) dm_free(cur2); (void) dm_free(blockoff); (void) dm_free(blockcnt); (void) dm_free(items); (void) dm_free(itemblk);
            *this = PartWs();
        }
    } part[2];
    unsigned long long *d_jobcnt = nullptr;   // per (chunk, set) counters of commet_index_and_search, kept between calls
    uint32_t *d_ids = nullptr, *d_idblk = nullptr;   // read numbers of the index selection of the running job, in order (sel_ids_kernel)
    uint64_t ids_cap = 0, idblk_cap = 0;
    uint32_t *d_ids2 = nullptr, *d_idblk2 = nullptr; // the same for the second of two jobs whose chunks are built side by side (multi.hpp, two jobs per tiled scan)
    uint64_t ids2_cap = 0, idblk2_cap = 0;
    uint32_t *d_act = nullptr, *d_actblk = nullptr;  // read numbers of a sparse search pass (sel & ~tags, in order) and the scan's block sums
    uint64_t act_cap = 0, actblk_cap = 0;
    uint64_t *d_mtags = nullptr;                     // found flags of the jobs of a commet_index_many_and_search pass (up to eight bitmaps over the search set)
    uint64_t mtags_cap = 0;
    int multi_job = 0;                               // option: 0 = commet_index_many_and_search shares passes between jobs where it can, 1 = job by job
    int sparse_search = 0;                           // option: 0 auto (a pass over less than half of a set's reads), 1 never, 2 whenever a selection applies
    unsigned long long *d_plansum = nullptr;  // per-block k-mer sums of a selection (host planner input)
    uint64_t plansum_cap = 0;
    uint64_t jobcnt_cap = 0;
    hipStream_t aux_stream = nullptr;         // second lane of a chunk group's index phase
    hipStream_t load_stream = nullptr;        // everything that makes a read set (uploads, k-mer counts): a set may be loaded by one
                                              // host thread while another runs jobs on sets that are complete
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int index_lanes = 2;                      // option: 1 = build the chunks of a group one after the other
    int lane_stagger = 1;                     // option / COMMET_LANE_STAGGER: the second lane's chunk starts behind the first lane's scatter1 (index_dispatch.hpp)
    bool stagger_armed = false;
    hipEvent_t ev_stagger = nullptr;

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
    hipStream_t list_stream = nullptr;        // query lists are built here, beside the job's index kernels (build_query_list)
    hipEvent_t ev_list = nullptr;
    unsigned long long *d_ql_totals = nullptr;   // scratch of the list build's scan, kept (no hipFree on the job path)
    uint64_t ql_totals_cap = 0;
    uint8_t *d_qres = nullptr;        // tiled search (tile_search.hpp): one result byte per query record of the set being scanned
    uint64_t qres_cap = 0;
    int tq_hit_cap = TQ_HIT_CAP;      // option "tq_hit_cap" (tests): full hits a piece of the replay may post before its scans walk their own candidates
    int ordered_scan = 0;             // option "ordered_scan": 0 = ragged sets of 2^16 reads and more are walked in order of their window counts by the
                                      // gather kernels' first pass, 1 = never, 2 = whenever the set is ragged (tests)
    int mask_split = 0;               // option "mask_split": 0 = a pass over that list runs segment by segment, each with the mask width its reads need, 1 = one launch
    unsigned long long *d_lo_cnt = nullptr;   // scratch of that list's counting sort (class-major block counts), kept
    uint64_t lo_cnt_cap = 0;
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
    uint64_t ql_bytes = 0, ql_budget = 64ull << 30, ql_clock = 0, ql_evictions = 0;   // (budget: half the device when that is more, commet_create)
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
    uint64_t uid = 0;                          // unique per read set of the process (a pointer may come back; a uid does not)
    uint64_t fhw_total = 0;                    // first-hit windows of all reads for the context's (k, t): sum of max(0, len - t k + 1) (finalize)
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
            (void) dm_free(d_tile_off); (void) dm_free(d_qaddr); (void) dm_free(d_qwho); (void) dm_free(d_tstart); (void) dm_free(d_tlen);
            *this = QueryList();
        }
    };
    mutable QueryList ql;
    // the set's reads in order of their first-hit window counts (tile_search.hpp, lo_*_kernel): ragged sets only, made on the first pass
    // of a gather kernel that visits the whole set; ids[n_reads] = n_reads closes the list
    mutable uint32_t *d_len_order = nullptr;
    mutable bool len_order_failed = false;
    // ... cut by mask width (round 6): the list starts with the reads of most windows, so the register-mask kernel runs segment by
    // segment with the narrowest masks the segment's reads fit (search_dispatch.hpp, launch_search_group); segment s = reads
    // [start, start + count) of the list, its count also at d_len_order[n_reads + 1 + s] for the kernel
    struct LenSeg { uint32_t start, count; int mw; };
    mutable LenSeg len_seg[5];
    mutable int n_len_seg = 0;
    mutable uint32_t ql_wanted = 0;                 // scans that would have taken the tiled search had the set's (large) list existed (tiled_ok)
    mutable std::atomic<uint64_t> ql_reserved_at{0};  // g_devmem.trims when the memory was set aside: a trim since then has given it back
    mutable std::atomic<bool> ql_reserved{false};   // the memory of the set's list waits in the library's device cache (commet_readset_reserve_cache): a list above the cap may be built
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

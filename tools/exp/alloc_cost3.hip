// Third look at what device memory costs where a 16 GiB hipMalloc takes 483 ms (and sixteen of 1 GiB 0.4 ms): (a) plain hipMalloc of fresh memory by SIZE —
// where does it start to cost; (b) one virtual range backed by physical chunks of 1 / 2 / 4 GiB (hipMemCreate per chunk, hipMemMap side by side);
// (c) how fast random 4-byte gathers run over 2 GiB of each kind of memory (the search kernels' access pattern: does the mapping cost TLB reach?).
//   hipcc --offload-arch=gfx950 -O2 -o tools/exp/alloc_cost3 tools/exp/alloc_cost3.hip && tools/exp/alloc_cost3
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const size_t GiB = 1ull << 30;

// a range of `bytes` backed by chunks of `chunk` bytes each
static void *map_chunked(size_t bytes, size_t chunk)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) return nullptr;
    void *va = nullptr;
    if (hipMemAddressReserve(&va, bytes, gran, nullptr, 0) != hipSuccess) return nullptr;
    for (size_t off = 0; off < bytes; off += chunk) {
        const size_t n = bytes - off < chunk ? bytes - off : chunk;
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, n, &prop, 0) != hipSuccess) return nullptr;
        if (hipMemMap((char *) va + off, n, 0, h, 0) != hipSuccess) return nullptr;
        if (hipMemRelease(h) != hipSuccess) return nullptr;          // (the mapping keeps the memory)
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(va, bytes, &acc, 1) != hipSuccess) return nullptr;
    return va;
}

__global__ __launch_bounds__(256) void gather_kernel(const uint32_t *__restrict__ table, uint32_t mask, int per_thread, uint32_t *__restrict__ sink)
{
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    for (int i = 0; i < per_thread; i += 4) {
        uint32_t a[4];
        for (int j = 0; j < 4; ++j) x = x * 1664525u + 1013904223u, a[j] = (x >> 3) & mask;
        for (int j = 0; j < 4; ++j) acc += table[a[j]];
    }
    if (acc == 0x12345678u) *sink = acc;
}

static double gather_rate(const void *table, size_t bytes, uint32_t *sink)
{
    const uint32_t mask = (uint32_t) (bytes / 4 - 1);
    const int per_thread = 256, grid = 256 * 32;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    gather_kernel<<<grid, 256>>>((const uint32_t *) table, mask, per_thread, sink);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) gather_kernel<<<grid, 256>>>((const uint32_t *) table, mask, per_thread, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return 5.0 * grid * 256.0 * per_thread / (ms * 1e6);      // G gathers per second
}

int main()
{
    hipSetDevice(0);
    hipFree(nullptr);
    size_t fr = 0, tot = 0;
    hipMemGetInfo(&fr, &tot);
    printf("free %.1f GiB of %.1f\n", fr / (double) GiB, tot / (double) GiB);
    auto timed = [&](const char *what, double gib, auto &&fn) {
        const double t0 = now_ms();
        void *p = fn();
        const double t1 = now_ms();
        printf("%-52s %8.1f ms  (%5.1f ms per GiB)%s\n", what, t1 - t0, (t1 - t0) / gib, p ? "" : "  FAILED");
        fflush(stdout);
        return p;
    };
    char name[128];
    std::vector<void *> held;
    for (double g : {1.0, 1.5, 2.0, 2.5, 3.0, 4.0, 6.0, 8.0, 12.0, 16.0}) {
        snprintf(name, sizeof name, "hipMalloc %.1f GiB (fresh)", g);
        held.push_back(timed(name, g, [&] { void *p = nullptr; return hipMalloc(&p, (size_t) (g * GiB)) == hipSuccess ? p : nullptr; }));
    }
    for (double ch : {1.0, 2.0, 4.0}) {
        snprintf(name, sizeof name, "16 GiB range of %.0f GiB chunks (fresh)", ch);
        void *p = timed(name, 16, [&] { return map_chunked(16 * GiB, (size_t) (ch * GiB)); });
        if (p) timed("   hipMemset of it", 16, [&] { hipMemset(p, 1, 16 * GiB); hipDeviceSynchronize(); return p; });
    }
    {   // 2.5 GiB in chunks of 1 GiB (a tail chunk), 8.5 GiB
        timed("2.5 GiB range of 1 GiB chunks", 2.5, [&] { return map_chunked((size_t) (2.5 * GiB), GiB); });
        timed("8.5 GiB range of 1 GiB chunks", 8.5, [&] { return map_chunked((size_t) (8.5 * GiB), GiB); });
    }
    // random gathers over 2 GiB of each kind
    uint32_t *sink = nullptr;
    hipMalloc((void **) &sink, 4);
    hipStream_t st;
    hipStreamCreate(&st);
    void *plain = nullptr, *pool = nullptr;
    hipMalloc(&plain, 2 * GiB);
    hipMallocAsync(&pool, 2 * GiB, st);
    hipStreamSynchronize(st);
    void *chunked = map_chunked(2 * GiB, GiB), *chunked_small = map_chunked(2 * GiB, 64 << 20);
    for (int rep = 0; rep < 2; ++rep) {
        if (plain) { hipMemset(plain, 0, 2 * GiB); printf("gathers over 2 GiB, hipMalloc:            %.1f G/s\n", gather_rate(plain, 2 * GiB, sink)); }
        if (pool) { hipMemset(pool, 0, 2 * GiB); printf("gathers over 2 GiB, hipMallocAsync:       %.1f G/s\n", gather_rate(pool, 2 * GiB, sink)); }
        if (chunked) { hipMemset(chunked, 0, 2 * GiB); printf("gathers over 2 GiB, 1 GiB chunks mapped:  %.1f G/s\n", gather_rate(chunked, 2 * GiB, sink)); }
        if (chunked_small) { hipMemset(chunked_small, 0, 2 * GiB); printf("gathers over 2 GiB, 64 MiB chunks mapped: %.1f G/s\n", gather_rate(chunked_small, 2 * GiB, sink)); }
    }
    hipMemGetInfo(&fr, &tot);
    printf("free now %.1f GiB of %.1f\n", fr / (double) GiB, tot / (double) GiB);
    return 0;
}

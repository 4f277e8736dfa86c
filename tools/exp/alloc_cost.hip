// What does a first-touch allocation of device memory cost on this driver, by API?  Every variant takes FRESH memory (the earlier ones are
// still held), 16 GiB each; then the same sizes again after freeing (memory the process has touched before).
//   hipcc --offload-arch=gfx950 -O2 -o tools/exp/alloc_cost tools/exp/alloc_cost.hip && tools/exp/alloc_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t GiB = 1ull << 30, SZ = 16 * GiB;
    hipSetDevice(0);
    hipFree(nullptr);
    size_t fr = 0, tot = 0;
    hipMemGetInfo(&fr, &tot);
    printf("free %.1f GiB of %.1f\n", fr / (double) GiB, tot / (double) GiB);
    std::vector<void *> held;
    auto timed = [&](const char *what, auto &&fn) {
        const double t0 = now_ms();
        void *p = fn();
        const double t1 = now_ms();
        printf("%-44s %8.1f ms  (%5.1f ms per GiB)%s\n", what, t1 - t0, (t1 - t0) / (SZ / (double) GiB), p ? "" : "  FAILED");
        fflush(stdout);
        return p;
    };
    // 1. plain hipMalloc, fresh
    void *a = timed("hipMalloc 16 GiB (fresh)", [&] { void *p = nullptr; return hipMalloc(&p, SZ) == hipSuccess ? p : nullptr; });
    // 2. 16 x 1 GiB hipMalloc, fresh
    std::vector<void *> small;
    timed("16 x hipMalloc 1 GiB (fresh)", [&] { for (int i = 0; i < 16; ++i) { void *p = nullptr; if (hipMalloc(&p, GiB) != hipSuccess) return (void *) nullptr; small.push_back(p); } return small[0]; });
    // 3. hipExtMallocWithFlags default / uncached
    void *b = timed("hipExtMallocWithFlags(Default) 16 GiB (fresh)", [&] { void *p = nullptr; return hipExtMallocWithFlags(&p, SZ, hipDeviceMallocDefault) == hipSuccess ? p : nullptr; });
    // 4. stream-ordered pool
    hipStream_t st;
    hipStreamCreate(&st);
    void *c = timed("hipMallocAsync 16 GiB + sync (fresh)", [&] { void *p = nullptr; if (hipMallocAsync(&p, SZ, st) != hipSuccess) return (void *) nullptr; hipStreamSynchronize(st); return p; });
    // 5. virtual memory management: create + reserve + map + set access
    void *d = timed("hipMemCreate + map 16 GiB (fresh)", [&]() -> void * {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) return nullptr;
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, SZ, &prop, 0) != hipSuccess) return nullptr;
        void *va = nullptr;
        if (hipMemAddressReserve(&va, SZ, gran, nullptr, 0) != hipSuccess) return nullptr;
        if (hipMemMap(va, SZ, 0, h, 0) != hipSuccess) return nullptr;
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        if (hipMemSetAccess(va, SZ, &acc, 1) != hipSuccess) return nullptr;
        return va;
    });
    // first touch by a kernel-side fill of one of them (is anything deferred to first use?)
    timed("hipMemset 16 GiB of the hipMalloc block", [&] { hipMemset(a, 1, SZ); hipDeviceSynchronize(); return a; });
    timed("hipMemset 16 GiB again", [&] { hipMemset(a, 2, SZ); hipDeviceSynchronize(); return a; });
    if (c) timed("hipMemset 16 GiB of the async block", [&] { hipMemset(c, 1, SZ); hipDeviceSynchronize(); return c; });
    if (d) timed("hipMemset 16 GiB of the mapped block", [&] { hipMemset(d, 1, SZ); hipDeviceSynchronize(); return d; });
    // free and allocate again: memory this process has touched
    timed("hipFree 16 GiB", [&] { hipFree(a); return (void *) 1; });
    a = timed("hipMalloc 16 GiB (touched before)", [&] { void *p = nullptr; return hipMalloc(&p, SZ) == hipSuccess ? p : nullptr; });
    timed("hipFree 16 x 1 GiB", [&] { for (void *p : small) hipFree(p); return (void *) 1; });
    timed("hipMalloc 16 GiB (over the 16 freed GiB)", [&] { void *p = nullptr; return hipMalloc(&p, SZ) == hipSuccess ? p : nullptr; });
    (void) b;
    return 0;
}

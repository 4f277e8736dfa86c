// Second look: the cost of a fresh hipMalloc by SIZE (1 .. 16 GiB, each followed by a fill: is anything deferred to the first touch?),
// hipMallocAsync by size, free-and-take-again for both, and whether an IPC handle can be had for stream-ordered memory.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t GiB = 1ull << 30;
    hipSetDevice(0);
    hipFree(nullptr);
    hipStream_t st;
    hipStreamCreate(&st);
    auto fill = [&](void *p, size_t n) { const double t0 = now_ms(); hipMemset(p, 1, n); hipDeviceSynchronize(); return now_ms() - t0; };
    const double sizes[] = {0.25, 0.5, 1, 1.5, 2, 2.4, 3, 4, 8, 16};
    printf("-- fresh hipMalloc by size (held) --\n");
    std::vector<void *> held;
    for (double g : sizes) {
        const size_t n = (size_t) (g * GiB);
        void *p = nullptr;
        const double t0 = now_ms();
        const hipError_t e = hipMalloc(&p, n);
        const double t1 = now_ms();
        printf("hipMalloc %5.2f GiB: %8.1f ms (%5.1f per GiB) %s; fill %.1f ms\n", g, t1 - t0, (t1 - t0) / g, e == hipSuccess ? "" : "FAILED", p ? fill(p, n) : 0.0);
        fflush(stdout);
        held.push_back(p);
    }
    printf("-- fresh hipMallocAsync by size (held) --\n");
    std::vector<void *> aheld;
    for (double g : sizes) {
        const size_t n = (size_t) (g * GiB);
        void *p = nullptr;
        const double t0 = now_ms();
        const hipError_t e = hipMallocAsync(&p, n, st);
        hipStreamSynchronize(st);
        const double t1 = now_ms();
        printf("hipMallocAsync %5.2f GiB: %8.1f ms (%5.1f per GiB) %s; fill %.1f ms\n", g, t1 - t0, (t1 - t0) / g, e == hipSuccess ? "" : "FAILED", p ? fill(p, n) : 0.0);
        fflush(stdout);
        aheld.push_back(p);
    }
    {   // IPC handle of stream-ordered memory?
        hipIpcMemHandle_t h;
        const hipError_t e = hipIpcGetMemHandle(&h, aheld[2]);
        printf("hipIpcGetMemHandle on hipMallocAsync memory: %s\n", hipGetErrorString(e));
        (void) hipGetLastError();
        const hipError_t e2 = hipIpcGetMemHandle(&h, held[2]);
        printf("hipIpcGetMemHandle on hipMalloc memory: %s\n", hipGetErrorString(e2));
    }
    printf("-- free the 8 and 16 GiB blocks and take them again --\n");
    {
        double t0 = now_ms();
        hipFree(held[8]), hipFree(held[9]);
        printf("hipFree 8 + 16 GiB: %.1f ms\n", now_ms() - t0);
        for (double g : {8.0, 16.0}) {
            void *p = nullptr;
            t0 = now_ms();
            hipMalloc(&p, (size_t) (g * GiB));
            printf("hipMalloc %4.1f GiB again: %8.1f ms\n", g, now_ms() - t0);
        }
        t0 = now_ms();
        hipFreeAsync(aheld[8], st), hipFreeAsync(aheld[9], st);
        hipStreamSynchronize(st);
        printf("hipFreeAsync 8 + 16 GiB + sync: %.1f ms\n", now_ms() - t0);
        for (double g : {8.0, 16.0}) {
            void *p = nullptr;
            t0 = now_ms();
            hipMallocAsync(&p, (size_t) (g * GiB), st);
            hipStreamSynchronize(st);
            printf("hipMallocAsync %4.1f GiB again: %8.1f ms\n", g, now_ms() - t0);
        }
    }
    size_t fr = 0, tot = 0;
    hipMemGetInfo(&fr, &tot);
    printf("free now %.1f GiB of %.1f\n", fr / (double) GiB, tot / (double) GiB);
    return 0;
}

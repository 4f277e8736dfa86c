// Experiment (not part of the product): how fast can a chip-wide set of workgroups append fixed-size units (32 / 64 / 128 B,
// aligned) to many open output streams each — the write pattern of a one-level radix scatter into 4096 buckets per plane.
//   scatter32_bench <unit_bytes> <n_buckets> <cap_units> <flushes_per_round> <lanes_per_unit> [nt]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int UNIT, int LANES, bool NT>
__global__ __launch_bounds__(1024) void append_kernel(uint4 *out, uint32_t nb, uint32_t cap, uint32_t rounds, uint32_t fpr)
{
    extern __shared__ uint32_t cur[];
    for (uint32_t i = threadIdx.x; i < nb; i += 1024) cur[i] = 0;
    __syncthreads();
    constexpr int V = UNIT / 16;            // 16-byte vectors per unit
    constexpr int VPL = V / LANES;          // per lane
    const uint32_t unit_lane = threadIdx.x % LANES, slot = threadIdx.x / LANES;
    for (uint32_t r = 0; r < rounds; ++r) {
        if (slot < fpr) {
            const uint32_t b = mix(r * 2654435761u + slot * 40503u + blockIdx.x * 97u) & (nb - 1);
            uint32_t pos = 0;
            if (unit_lane == 0) pos = atomicAdd(&cur[b], 1u);
            if (LANES > 1) pos = __shfl(pos, (threadIdx.x & 63) - unit_lane);
            if (pos < cap) {
                uint4 *p = out + (((size_t) blockIdx.x * nb + b) * cap + pos) * V + unit_lane * VPL;
                const uint4 v = make_uint4(r, b, pos, slot);
#pragma unroll
                for (int i = 0; i < VPL; ++i) {
                    if (NT) __builtin_nontemporal_store(*(const v4u *) &v, (v4u *) (p + i)); else p[i] = v;
                }
            }
        }
        __syncthreads();
    }
}

template <int UNIT, int LANES>
static void run(bool nt, uint4 *out, uint32_t grid, uint32_t nb, uint32_t cap, uint32_t rounds, uint32_t fpr, hipStream_t s)
{
    if (nt) hipLaunchKernelGGL((append_kernel<UNIT, LANES, true>), dim3(grid), dim3(1024), nb * 4, s, out, nb, cap, rounds, fpr);
    else hipLaunchKernelGGL((append_kernel<UNIT, LANES, false>), dim3(grid), dim3(1024), nb * 4, s, out, nb, cap, rounds, fpr);
}

int main(int argc, char **argv)
{
    const int unit = argc > 1 ? atoi(argv[1]) : 32;
    const uint32_t nb = argc > 2 ? atoi(argv[2]) : 4096;
    const uint32_t cap = argc > 3 ? atoi(argv[3]) : 160;
    const uint32_t fpr = argc > 4 ? atoi(argv[4]) : 683;
    const int lanes = argc > 5 ? atoi(argv[5]) : 1;
    const bool nt = argc > 6 && atoi(argv[6]);
    const uint32_t grid = 256;
    const size_t bytes = (size_t) grid * nb * cap * unit;
    uint4 *out;
    OK(hipMalloc((void **) &out, bytes));
    OK(hipMemset(out, 0, bytes));
    const uint32_t rounds = (uint32_t) ((uint64_t) nb * cap * 95 / 100 / fpr);   // ~95 % of the capacity, so that few appends are dropped
    hipStream_t s;
    OK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    OK(hipEventCreate(&e0));
    OK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        OK(hipEventRecord(e0, s));
#define GO(U, L) if (unit == U && lanes == L) run<U, L>(nt, out, grid, nb, cap, rounds, fpr, s)
        GO(32, 1); GO(32, 2); GO(64, 1); GO(64, 2); GO(64, 4); GO(128, 2); GO(128, 4); GO(128, 8); GO(16, 1);
        OK(hipEventRecord(e1, s));
        OK(hipStreamSynchronize(s));
        float ms;
        OK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double written = (double) grid * rounds * fpr * unit;
    printf("unit %d B, %u buckets, %u flushes/round x %d lanes, nt %d: %.1f GB in %.3f ms = %.2f TB/s (%.1f G units/s)\n", unit, nb, fpr, lanes, (int) nt,
           written / 1e9, best, written / best / 1e9, written / unit / best / 1e6);
    return 0;
}

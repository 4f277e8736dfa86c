// Experiment (not part of the product): the core of a one-level scatter — 2048 open buckets per workgroup kept as 64-byte
// units of 24 twenty-bit keys in LDS, claimed with LDS atomics, flushed as whole units — on synthetic keys.
//   one_scatter_proto [rounds] [valu_pad] [grid]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr uint32_t NLB = 2048, QCAP = 1024, PIECES = 256;

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ __launch_bounds__(1024) void scatter_proto(uint4 *units, unsigned long long *spill_n, uint32_t *spill, uint32_t rounds, uint32_t cap_units, int pad)
{
    extern __shared__ unsigned long long lds[];
    unsigned long long *stage = lds;                           // NLB * 8
    uint32_t *cntw = (uint32_t *) (stage + NLB * 8);           // NLB / 2 (two 16-bit fill counters a word)
    uint16_t *nunits = (uint16_t *) (cntw + NLB / 2);          // NLB
    uint16_t *flist = nunits + NLB;                            // NLB
    uint32_t *q = (uint32_t *) (flist + NLB);                  // 2 * QCAP
    uint32_t *fcnt = q + 2 * QCAP, *qcnt = fcnt + 2;
    for (uint32_t i = threadIdx.x; i < NLB * 8; i += 1024) stage[i] = 0;
    for (uint32_t i = threadIdx.x; i < NLB / 2; i += 1024) cntw[i] = 0;
    for (uint32_t i = threadIdx.x; i < NLB; i += 1024) nunits[i] = 0;
    if (threadIdx.x < 4) fcnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t piece = blockIdx.x % PIECES, cls = blockIdx.x / PIECES;
    uint32_t n_spill = 0;
    auto place = [&](uint32_t key31, bool second, uint32_t par) {
        const uint32_t lb = key31 >> 20, off = key31 & 0xFFFFFu, sh = (lb & 1u) * 16u;
        const uint32_t s = (atomicAdd(&cntw[lb >> 1], 1u << sh) >> sh) & 0xFFFFu;
        if (s < 24u) {
            const uint32_t g = (s * 11u) >> 5;
            atomicOr(&stage[lb * 8 + g], (unsigned long long) off << (20u * (s - 3u * g)));
            if (s == 23u) flist[atomicAdd(&fcnt[par], 1u)] = (uint16_t) lb;
        } else if (!second) {
            const uint32_t i = atomicAdd(&qcnt[par], 1u);
            if (i < QCAP) q[par * QCAP + i] = key31;
            else ++n_spill;
        } else ++n_spill;
    };
    uint32_t x = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u;
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t par = r & 1u;
        const uint32_t nq = min(qcnt[par ^ 1u], QCAP);
        if (threadIdx.x < nq) place(q[(par ^ 1u) * QCAP + threadIdx.x], true, par);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            x = mix(x + 0x9E3779B9u);
            uint32_t v = x;
            for (int p = 0; p < pad; ++p) v = v * 0x01000193u + (v >> 7);   // stand-in for the key arithmetic
            if (v >> 31) place((x ^ (v & 1u)) & 0x7FFFFFFFu, false, par);
        }
        __syncthreads();
        const uint32_t nf = fcnt[par];
        for (uint32_t e = threadIdx.x >> 2; e < nf; e += 256) {
            const uint32_t lb = flist[e], l4 = threadIdx.x & 3u;
            const unsigned long long v0 = stage[lb * 8 + 2 * l4], v1 = stage[lb * 8 + 2 * l4 + 1];
            stage[lb * 8 + 2 * l4] = 0, stage[lb * 8 + 2 * l4 + 1] = 0;
            uint32_t u = 0;
            if (l4 == 0) {
                u = nunits[lb];
                nunits[lb] = (uint16_t) (u + 1);
                atomicAnd(&cntw[lb >> 1], ~(0xFFFFu << ((lb & 1u) * 16u)));
            }
            u = __shfl(u, (threadIdx.x & 63u) & ~3u);
            if (u < cap_units) {
                const uint64_t gb = (uint64_t) cls * NLB + lb;
                units[((gb * cap_units + u) * PIECES + piece) * 4 + l4] = make_uint4((uint32_t) v0, (uint32_t) (v0 >> 32), (uint32_t) v1, (uint32_t) (v1 >> 32));
            } else if (l4 == 0) ++n_spill;
        }
        if (threadIdx.x == 0) fcnt[par ^ 1u] = 0, qcnt[par ^ 1u] = 0;
        __syncthreads();
    }
    if (n_spill) atomicAdd(spill_n, (unsigned long long) n_spill);
}

int main(int argc, char **argv)
{
    const uint32_t rounds = argc > 1 ? atoi(argv[1]) : 172;
    const int pad = argc > 2 ? atoi(argv[2]) : 0;
    const uint32_t grid = argc > 3 ? atoi(argv[3]) : 2048;
    const uint32_t mean_units = (uint32_t) ((uint64_t) rounds * 4096 / NLB / 24);
    const uint32_t cap_units = 2 * mean_units + 4;
    const size_t bytes = (size_t) (grid / PIECES) * NLB * cap_units * PIECES * 64;
    uint4 *units;
    unsigned long long *spill_n;
    OK(hipMalloc((void **) &units, bytes));
    OK(hipMemset(units, 0, bytes));
    OK(hipMalloc((void **) &spill_n, 8));
    OK(hipMemset(spill_n, 0, 8));
    const size_t lds = (size_t) NLB * 64 + NLB * 2 + NLB * 2 + NLB * 2 + 2 * QCAP * 4 + 32;
    OK(hipFuncSetAttribute((const void *) scatter_proto, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    hipStream_t s;
    OK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    OK(hipEventCreate(&e0));
    OK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        OK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(scatter_proto, dim3(grid), dim3(1024), lds, s, units, spill_n, (uint32_t *) nullptr, rounds, cap_units, pad);
        OK(hipEventRecord(e1, s));
        OK(hipStreamSynchronize(s));
        OK(hipGetLastError());
        float ms;
        OK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    unsigned long long sp = 0;
    OK(hipMemcpy(&sp, spill_n, 8, hipMemcpyDeviceToHost));
    const double keys = (double) grid * rounds * 4096;
    printf("rounds %u pad %d grid %u lds %zu B cap %u units (%.1f GB): %.3f ms, %.0f G kept keys/s, spilled %.4f %%\n", rounds, pad, grid, lds, cap_units,
           bytes / 1e9, best, keys / best / 1e6, 100.0 * sp / 3 / keys);
    return 0;
}

// Experiment (not part of the product): the write pattern of part_scatter1_kernel alone — 1024 workgroups (two per CU at a
// time), 256 output streams each, one 256-byte run per stream and round, bucket-major layout — against how the buffer was
// allocated and where the streams start.
//   s1_pattern_bench [n_allocs] 
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr uint32_t PIECES = 1024;
static uint32_t NB = 256, ROUNDS = 86, RUN_KEYS = 64;

// word offset of (bucket c, piece p, round r): c * bucket_stride + p * piece_stride + r * RUN_KEYS + skew
template <bool NT>
__global__ __launch_bounds__(512, 4) void pattern_kernel(uint32_t *out, uint64_t bucket_stride, uint64_t piece_stride, uint32_t skew_mask, uint32_t rounds,
                                                          uint32_t NB, uint32_t RUN_KEYS)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t piece = blockIdx.x;
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t lpr = RUN_KEYS / 4, rpw = 64 / lpr;
        for (uint32_t c = wave * rpw + lane / lpr; c < NB; c += 8 * rpw) {
            const uint32_t skew = ((c * 2654435761u + piece * 40503u) >> 16) & skew_mask;     // unaligned starts like the real runs (multiples of 4 bytes)
            uint32_t *p = out + (uint64_t) c * bucket_stride + (uint64_t) piece * piece_stride + (uint64_t) r * RUN_KEYS + skew;
            const uint32_t l = lane % lpr;
            // 64 keys = 16 lanes x 4 words; unaligned -> four 4-byte stores per lane when skewed, one 16-byte store when not
            if (skew_mask == 0) {
                if (NT) { v4u v = {r, c, piece, l}; __builtin_nontemporal_store(v, (v4u *) (p + 4 * l)); }
                else *(uint4 *) (p + 4 * l) = make_uint4(r, c, piece, l);
            }
            else { p[4 * l] = r; p[4 * l + 1] = c; p[4 * l + 2] = piece; p[4 * l + 3] = l; }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void fill_kernel(uint4 *out, uint64_t n)
{
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t) gridDim.x * 256) out[i] = make_uint4(1, 2, 3, 4);
}
__global__ __launch_bounds__(256) void sum_kernel(const uint4 *in, uint64_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t) gridDim.x * 256) { const uint4 v = in[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
static uint64_t g_words_left = 0;
static bool g_nt = false;   // words of the buffer from the pointer handed to run()

static float run(uint32_t *buf, uint64_t bs, uint64_t ps, uint32_t skew_mask, hipStream_t s)
{
    // the last word any lane can touch: refuse a geometry that does not fit the buffer
    const uint64_t last = (uint64_t) (NB - 1) * bs + (uint64_t) (PIECES - 1) * ps + (uint64_t) ROUNDS * RUN_KEYS + skew_mask + 64;
    if (last >= g_words_left) {
        fprintf(stderr, "geometry needs %llu words, buffer has %llu: skipped\n", (unsigned long long) last, (unsigned long long) g_words_left);
        return -1.f;
    }
    hipEvent_t e0, e1;
    OK(hipEventCreate(&e0));
    OK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        OK(hipEventRecord(e0, s));
        if (g_nt) hipLaunchKernelGGL(pattern_kernel<true>, dim3(PIECES), dim3(512), 0, s, buf, bs, ps, skew_mask, ROUNDS, NB, RUN_KEYS);
        else hipLaunchKernelGGL(pattern_kernel<false>, dim3(PIECES), dim3(512), 0, s, buf, bs, ps, skew_mask, ROUNDS, NB, RUN_KEYS);
        OK(hipEventRecord(e1, s));
        OK(hipStreamSynchronize(s));
        float ms;
        OK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    OK(hipEventDestroy(e0));
    OK(hipEventDestroy(e1));
    return best;
}

int main(int argc, char **argv)
{
    const int n_allocs = argc > 1 ? atoi(argv[1]) : 6;
    const uint64_t ps = (uint64_t) ROUNDS * RUN_KEYS + 8;        // words per (bucket, piece): like exact counts, not a power of two
    const uint64_t bs = ps * PIECES;
    size_t bytes = (size_t) NB * PIECES * 6 * 4096 + (64u << 20);   // room for the 4 KiB-padded variant, too
    if (getenv("S1_BYTES_MIB")) bytes = std::max<size_t>(bytes, (size_t) atoll(getenv("S1_BYTES_MIB")) << 20);
    hipStream_t s;
    OK(hipStreamCreate(&s));
    printf("buffer %.2f GB, piece stride %llu B, bucket stride %.2f MB\n", bytes / 1e9, (unsigned long long) ps * 4, bs * 4 / 1e6);
    std::vector<uint32_t *> bufs;
    g_words_left = bytes / 4;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    uint32_t *sink;
    OK(hipMalloc((void **) &sink, 4));
    auto stream_ms = [&](uint32_t *p, bool write) {
        hipEvent_t e0, e1;
        OK(hipEventCreate(&e0));
        OK(hipEventCreate(&e1));
        float best = 1e9f;
        const uint64_t n16 = (uint64_t) NB * bs / 4;        // the bytes the pattern writes
        for (int rep = 0; rep < 3; ++rep) {
            OK(hipEventRecord(e0, s));
            if (write) hipLaunchKernelGGL(fill_kernel, dim3(256 * 16), dim3(256), 0, s, (uint4 *) p, n16);
            else hipLaunchKernelGGL(sum_kernel, dim3(256 * 16), dim3(256), 0, s, (const uint4 *) p, n16, sink);
            OK(hipEventRecord(e1, s));
            OK(hipStreamSynchronize(s));
            float ms;
            OK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        return best;
    };
    auto report = [&](const char *what, int a, uint32_t *p) {
        printf("%s %d at %p: aligned runs %.3f ms, unaligned runs %.3f ms; streaming fill %.3f ms, streaming read %.3f ms\n", what, a, (void *) p,
               run(p, bs, ps, 0, s), run(p, bs, ps, 15u, s), stream_ms(p, true), stream_ms(p, false));
    };
    if (mode == 4) {   // map of the streaming-fill rate over one large allocation, block by block
        const size_t blk = 256ull << 20, nblk = argc > 3 ? (size_t) atoi(argv[3]) : 192;
        uint32_t *big;
        OK(hipMalloc((void **) &big, blk * nblk));
        OK(hipMemsetAsync(big, 0, blk * nblk, s));
        hipEvent_t e0, e1;
        OK(hipEventCreate(&e0));
        OK(hipEventCreate(&e1));
        printf("fill rate (TB/s) of %zu blocks of 256 MiB at %p:\n", nblk, (void *) big);
        for (size_t i = 0; i < nblk; ++i) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                OK(hipEventRecord(e0, s));
                hipLaunchKernelGGL(fill_kernel, dim3(256 * 16), dim3(256), 0, s, (uint4 *) (big + i * (blk / 4)), (uint64_t) (blk / 16));
                OK(hipEventRecord(e1, s));
                OK(hipStreamSynchronize(s));
                float ms;
                OK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("%.2f%s", blk / best / 1e9, (i % 16 == 15) ? "\n" : " ");
        }
        printf("\n");
        return 0;
    }
    if (mode == 1) {   // something small and something large allocated first, as in the library (filter, read sets)
        void *x, *y;
        OK(hipMalloc(&x, 2ull << 30));
        OK(hipMalloc(&y, 500ull << 20));
        OK(hipMemsetAsync(x, 0, 2ull << 30, s));
    }
    for (int a = 0; a < n_allocs; ++a) {
        uint32_t *b;
        OK(hipMalloc((void **) &b, bytes));
        OK(hipMemsetAsync(b, 0, bytes, s));
        bufs.push_back(b);
        report("alloc", a, b);
    }
    if (mode == 2) {   // free them all and allocate again
        OK(hipStreamSynchronize(s));
        for (uint32_t *p : bufs) OK(hipFree(p));
        bufs.clear();
        for (int a = 0; a < n_allocs; ++a) {
            uint32_t *b;
            OK(hipMalloc((void **) &b, bytes));
            OK(hipMemsetAsync(b, 0, bytes, s));
            bufs.push_back(b);
            report("again", a, b);
        }
    }
    if (mode == 3) {   // a larger allocation, the pattern run at several places inside it
        uint32_t *big;
        const size_t nbig = 5;
        OK(hipMalloc((void **) &big, bytes * nbig));
        OK(hipMemsetAsync(big, 0, bytes * nbig, s));
        for (size_t i = 0; i < nbig; ++i) report("inside one allocation of 5x the size, part", (int) i, big + i * (bytes / 4));
    }
    uint32_t *b = bufs[0];
    for (uint64_t off : {0ull, 64ull, 1024ull, 4096ull, 65536ull, 1048576ull, 2097152ull + 4096ull})
    {
        g_words_left = bytes / 4 - off / 4;
        printf("alloc 0 + %llu B: unaligned runs %.3f ms\n", (unsigned long long) off, run(b + off / 4, bs, ps, 15u, s));
    }
    g_words_left = bytes / 4;
    // other strides: piece regions padded to 4 KiB / a prime number of 256-byte lines; bucket stride a power of two
    for (int v = 0; v < 3; ++v) {   // fewer, longer runs: 128 coarse buckets x 128 keys, 64 x 256 (same bytes per round)
        NB = 256u >> v, RUN_KEYS = 64u << v;
        const uint64_t ps2 = (uint64_t) ROUNDS * RUN_KEYS + 8, bs2 = ps2 * PIECES;
        for (int a : {0, 1})
            printf("%u buckets x %u-key runs, alloc %d: aligned %.3f ms, unaligned %.3f ms\n", NB, RUN_KEYS, a, run(bufs[a], bs2, ps2, 0, s), run(bufs[a], bs2, ps2, 15u, s));
    }
    g_nt = true;
    for (int v = 0; v < 2; ++v) {
        NB = 256u >> v, RUN_KEYS = 64u << v;
        const uint64_t ps2 = (uint64_t) ROUNDS * RUN_KEYS + 8, bs2 = ps2 * PIECES;
        printf("non-temporal stores, %u buckets x %u-key runs, alloc 0: aligned %.3f ms\n", NB, RUN_KEYS, run(bufs[0], bs2, ps2, 0, s));
    }
    g_nt = false;
    NB = 256, RUN_KEYS = 64;
    printf("piece stride 4 KiB-padded: %.3f ms\n", run(b, ((ps * 4 + 4095) / 4096 * 1024) * PIECES, (ps * 4 + 4095) / 4096 * 1024, 15u, s));
    printf("bucket stride 2^k (32 MiB): %.3f ms\n", run(b, (32u << 20) / 4, ps, 15u, s));
    printf("piece-major layout (a piece's 256 runs next to each other): %.3f ms\n", run(b, ps, ps * NB, 15u, s));
    return 0;
}

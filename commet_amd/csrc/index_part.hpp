// index_part.hpp — bucketed construction of the 4-lane Bloom filter (gfx950).
//
// Why: setting 4 random bits per k-mer with global atomics runs at the memory
// system's atomic rate (~18 G atomic ORs/s on a 2 GiB table, measured,
// profiles/r01_membench_random_access.jsonl), i.e. ~150 ms for the 6.7e8 k-mers
// of BASELINE configs[1].  LDS atomics are two orders of magnitude faster, so
// the filter is built tile by tile in LDS instead:
//
//   tile     = 2^19 consecutive bits of one plane (64 KiB of LDS)
//   bucket   = (plane, key >> 19)                      NB = 4 * 2^(k-19) buckets
//   hist     : count the keys of every bucket           (LDS histogram, 1 pass over the reads)
//   scan     : exclusive scan -> exact bucket offsets, cursors, build work list
//   scatter1 : keys -> 2^b1 coarse buckets (top b1 bits of the bucket id); per
//              block an LDS counting sort, runs written coalesced     (bufA)
//   scatter2 : every coarse bucket -> its 2^b2 final buckets, same scheme (bufB)
//   build    : one workgroup per (tile, split): ds_or the bucket's keys into a
//              zeroed LDS tile, write the tile to the filter with plain
//              coalesced stores (tiles whose bucket is split over several
//              workgroups OR their non-zero words in with atomics instead)
//
// All traffic is streaming: 2 x 4 B per key and level, plus the filter once.
// Results are bit-identical to the atomic path (set semantics); which path runs
// is a host decision (capi.hip: use_partition()).
//
// Replaces: BloomFilter::feed (bloom_filter.h:112-118) applied by index_reads
// (index_reads.h:51-59) to a whole chunk.
#pragma once

#include "kernels.hpp"

namespace commet {

// Timing ablations (results become wrong on purpose) exist only in builds made with -DCOMMET_ABLATE=<mask>; the shipped
// library compiles them out.  scatter1: 1 no write-out, 2 no pass B, 4 no pass-A atomics; scatter2: 32 no placement,
// 64 no write-out, 128 no cursor reservation, 256 no counting.
#ifndef COMMET_ABLATE
#define COMMET_ABLATE 0
#endif

constexpr int      TILE_BITS = 19;
constexpr uint32_t TILE_MASK = (1u << TILE_BITS) - 1;
constexpr uint32_t TILE_WORDS = 1u << (TILE_BITS - 5);      // 16384 words = 64 KiB
#ifndef COMMET_S1_NT
#define COMMET_S1_NT 512
#endif
constexpr int      S1_NT = COMMET_S1_NT;                    // scatter-1 workgroup size
constexpr uint32_t S1_KEYS = 32u * COMMET_S1_NT;            // keys staged per scatter-1 round (64 KiB)
constexpr uint32_t S1_ITEMS = S1_NT;                        // octet items per round: one per thread, keys cached
#ifndef COMMET_S2_NT
#define COMMET_S2_NT 512   // (448 threads x 72 VGPRs = four workgroups per CU instead of three: 4.52 against 4.54 ms per configs[1] step; 384: 5.06)
#endif
constexpr int      S2_NT = COMMET_S2_NT;                    // scatter-2 workgroup size
constexpr uint32_t S2_KEYS = 16u * COMMET_S2_NT;            // keys per scatter-2 block (16 per thread)
constexpr uint32_t S2_PER_THREAD = S2_KEYS / S2_NT;
#ifndef S2P_WAVES
#define S2P_WAVES 6   // 80 VGPRs, three workgroups per CU; 8 (64 VGPRs, four per CU with S2P_MAX_SUB = 128) spills and measured 4.6 instead of 4.1 ms
#endif
constexpr uint32_t S2P_MAX_SUB = 256;                       // packed scatter-2: final buckets per coarse bucket (k <= 33: the reference's default k takes the packed
                                                            // geometry too; 41.9 KiB of LDS, three workgroups of 80 VGPRs per CU either way)
constexpr int      HIST_NT = 1024;                          // histogram workgroup size
#ifndef BUILD_NT
#define BUILD_NT 1024   // build workgroup: its loads in flight are what the kernel runs on (256 / 512 / 1024 threads: 2.75 / 2.52 / 2.36 ms per configs[1] step)
#endif
constexpr uint32_t BUILD_CAP = 1u << 17;                    // keys per build workgroup
constexpr uint32_t HIST_MAX_BUCKETS = 32768;                // LDS histogram capacity (128 KiB)
constexpr uint32_t MAX_SUB = 512;                           // 2^b2 upper bound
constexpr uint32_t MAX_L1 = 256;                            // 2^b1 upper bound
#ifndef HIST_ROLLING
#define HIST_ROLLING 1
#endif
#ifndef S1_ALIGNED
#define S1_ALIGNED 1
#endif
constexpr uint32_t S1_GRID_MAX = 1024;                       // pieces of the read range = scatter-1 workgroups (two run per CU at a time; 512 / 1024 / 2048 pieces: 4.46 / 4.37 / 4.33 ms per configs[1] step)

struct PartGeom {
    int      k;
    int      nb_bits;    // log2(number of buckets) = k - 17
    int      b1, b2;     // level-1 / level-2 radix bits (b2 may be 0)
    uint32_t nb;         // buckets
    uint32_t nb1;        // coarse buckets
    int      plane_shift;   // k - TILE_BITS: bucket = (plane << plane_shift) | (key >> TILE_BITS)
    int      xcd_swizzle;   // scatter2: slab order, number of interleaved slab ranges (speed only)
    int      packed;        // final buckets hold groups of three 19-bit keys in 8 bytes (two-level geometry only)
};

inline PartGeom make_geom(int k)
{
    PartGeom g;
    g.k = k;
    g.nb_bits = k - 17;
    // two levels: 2^b1 coarse buckets out of scatter1, 2^b2 final buckets per coarse one out of scatter2.  scatter1 is bound by its
    // write pattern — one run per coarse bucket, workgroup and round, 2^17 streams open at a time over ~6 GB: 2^8 buckets x 256-byte
    // runs 1.95 ms per configs[1] chunk for the writes alone, 2^7 x 512 bytes 1.59 ms (tools/exp/s1_pattern_bench.hip) — so 2^7
    // coarse buckets wherever the packed scatter2 still takes the rest (b2 <= 8): k = 32 is 7 + 8 (was 8 + 7: scatter1 -0.55 … -0.9 ms
    // per step, scatter2 +0.15 … +0.5)
    g.b1 = g.nb_bits <= 8 ? g.nb_bits : (g.nb_bits + 1) / 2;
    if (g.nb_bits > 8 && g.b1 > 7 && g.nb_bits - 7 <= 8) g.b1 = 7;   // (two-level geometries only: k = 25 keeps its single level of 2^8 buckets)
    if (g.b1 > 8) g.b1 = 8;
    g.b2 = g.nb_bits - g.b1;
    g.nb = 1u << g.nb_bits;
    g.nb1 = 1u << g.b1;
    g.plane_shift = k - TILE_BITS;
    g.xcd_swizzle = 0;
    g.packed = 0;
    return g;
}

// An item = 8 consecutive bases of a read ("octet" q covers bases 8q..8q+7).
// Octets below (k-1)>>3 hold no k-mer end and are never enumerated.
__device__ __forceinline__ uint32_t octets_of(uint32_t len, int k)
{
    return len >= (uint32_t) k ? ((len - 1) >> 3) - ((uint32_t) (k - 1) >> 3) + 1 : 0;
}

// calls f(plane, key) for the 4 forward keys of every complete k-mer ending in octet q of the read at p
// TOP: the plane-A key is only good for its bucket (psi_a_top)
// planes: bit p set = plane p's key is wanted (a histogram pass over part of the buckets needs only some planes)
template <typename W, bool TOP = false, typename F>
__device__ __forceinline__ void for_each_key(const uint32_t *p, uint32_t len, uint32_t q, int k, F &&f, uint32_t planes = 15u)
{
    using T = KeyTraits<W>;
    const uint32_t w = q >> 2;
    ItemWords<W> it;
    it.load(p, w);
    const W mask = (k == T::BITS) ? ~(W) 0 : (((W) 1 << k) - 1);
    const int sh = T::BITS - k;
    const uint32_t j0 = (q & 3u) * 8u;
#pragma unroll
    for (uint32_t jj = 0; jj < 8; ++jj) {
        const uint32_t j = j0 + jj;
        const uint32_t pos = 32u * w + j;
        if (pos + 1u < (uint32_t) k || pos >= len) continue;
        W wh, wl;
        if (!it.window(j, k, mask, wh, wl)) continue;
        const W ka = T::brev(wh) >> sh;
        const W kb = T::brev(wl) >> sh;
        if (planes & 1u) f(0u, TOP ? psi_a_top<W>(ka, k) : psi_a<W>(ka, k));   // plane A is stored strand-paired (kernels.hpp)
        if (planes & 2u) f(1u, kb);
        if (planes & 4u) f(2u, ka ^ kb);
        if (planes & 8u) f(3u, ka | kb);
    }
}

// The same for a bucket count (k <= 32): calls f(plane, bucket) with bucket = key >> TILE_BITS, the top nbits = k - TILE_BITS
// bits of the plane's key, for the complete k-mers ending in octet q.  Those bits roll from one position to the next:
//   top(keya) = the hi bits of the nbits OLDEST bases, oldest first   -> shift left, the next base enters at the bottom
//   top(keyb) = the same of the lo bits
//   top(psi_a) = top(keya) ^ ~(hi bits of the nbits NEWEST bases, newest first)    (psi_a_top: s = u ^ g(L))
//                                                                     -> shift right, the new base enters at the top
// so a position costs ~27 instructions instead of the ~52 of three 64-bit window extractions, two bit reversals and
// psi_a_top (the histogram kernel spends 82 % of the VALU's cycles).  The entering bases sit at fixed offsets from the
// octet's first position (k - nbits = TILE_BITS), one 64-bit shift per plane and octet.
template <typename F>
__device__ __forceinline__ void for_each_bucket32(const uint32_t *p, uint32_t len, uint32_t q, int k, F &&f)
{
    const uint32_t w = q >> 2, j0 = (q & 3u) * 8u;
    ItemWords<uint32_t> it;
    it.load(p, w);
    const int nbits = k - TILE_BITS;
    const uint32_t mask = (k == 32) ? ~0u : ((1u << k) - 1u), nmask = (1u << nbits) - 1u;
    const uint64_t hi64 = ((uint64_t) it.hi[1] << 32) | it.hi[0], lo64 = ((uint64_t) it.lo[1] << 32) | it.lo[0];
    const uint64_t va64 = ((uint64_t) it.va[1] << 32) | it.va[0];
    const uint32_t s = 33u + j0 - (uint32_t) k;   // the window ending at j0 (ItemWords::window)
    const uint32_t wh = (uint32_t) (hi64 >> s) & mask, wl = (uint32_t) (lo64 >> s) & mask, vw = (uint32_t) (va64 >> s) & mask;
    uint32_t a_top = __brev(wh) >> (32 - nbits), b_top = __brev(wl) >> (32 - nbits), w_top = wh >> (k - nbits);
    uint32_t run = (uint32_t) __clz((int) ~(vw << (32 - k)));   // valid bases in a row up to j0, counted up to k
    // bit jj of these = the base that enters at position j0 + jj: the oldest groups take base (position - TILE_BITS)
    const uint32_t cA = (uint32_t) (hi64 >> (32u - TILE_BITS + j0)), cB = (uint32_t) (lo64 >> (32u - TILE_BITS + j0));
    const uint32_t cW = it.hi[1] >> j0, cV = it.va[1] >> j0;
#pragma unroll
    for (uint32_t jj = 0; jj < 8; ++jj) {
        if (jj) {
            a_top = ((a_top << 1) | ((cA >> jj) & 1u)) & nmask;
            b_top = ((b_top << 1) | ((cB >> jj) & 1u)) & nmask;
            w_top = (w_top >> 1) | (((cW >> jj) & 1u) << (nbits - 1));
            run = ((cV >> jj) & 1u) ? run + 1u : 0u;
        }
        if (run < (uint32_t) k || 32u * w + j0 + jj >= len) continue;
        f(0u, a_top ^ w_top ^ nmask);
        f(1u, b_top);
        f(2u, a_top ^ b_top);
        f(3u, a_top | b_top);
    }
}

// The same for 64-bit keys (k = 33, 34; round 3): the rolling part only looks at the bases that enter — position - TILE_BITS
// for the oldest groups, the position itself for the newest — which lie in word triples w - 1 and w whatever k; only the
// octet's first window needs the three-word extraction.  planes: the planes whose buckets the pass counts (two passes of
// two planes at k = 33, four of one at k = 34: 2^16 / 2^17 counters do not fit the LDS).
template <typename F>
__device__ __forceinline__ void for_each_bucket64(const uint32_t *p, uint32_t len, uint32_t q, int k, uint32_t planes, F &&f)
{
    const uint32_t w = q >> 2, j0 = (q & 3u) * 8u;
    ItemWords<uint64_t> it;
    it.load(p, w);
    const int nbits = k - TILE_BITS;                      // 14, 15 <= k / 2
    const uint64_t mask = (1ull << k) - 1ull;
    const uint32_t nmask = (1u << nbits) - 1u;
    uint64_t wh, wl;
    (void) it.window(j0, k, mask, wh, wl);
    const uint64_t vw = ItemWords<uint64_t>::ext(it.va, 65u + j0 - (uint32_t) k, mask);
    uint32_t a_top = (uint32_t) (__brevll(wh) >> (64 - nbits)), b_top = (uint32_t) (__brevll(wl) >> (64 - nbits));
    uint32_t w_top = (uint32_t) (wh >> (k - nbits));
    uint32_t run = (uint32_t) __clzll((long long) ~(vw << (64 - k)));   // valid bases in a row up to j0, counted up to k
    const uint64_t hi64 = ((uint64_t) it.hi[2] << 32) | it.hi[1], lo64 = ((uint64_t) it.lo[2] << 32) | it.lo[1];
    const uint32_t cA = (uint32_t) (hi64 >> (32u - TILE_BITS + j0)), cB = (uint32_t) (lo64 >> (32u - TILE_BITS + j0));
    const uint32_t cW = it.hi[2] >> j0, cV = it.va[2] >> j0;
#pragma unroll
    for (uint32_t jj = 0; jj < 8; ++jj) {
        if (jj) {
            a_top = ((a_top << 1) | ((cA >> jj) & 1u)) & nmask;
            b_top = ((b_top << 1) | ((cB >> jj) & 1u)) & nmask;
            w_top = (w_top >> 1) | (((cW >> jj) & 1u) << (nbits - 1));
            run = ((cV >> jj) & 1u) ? run + 1u : 0u;
        }
        if (run < (uint32_t) k || 32u * w + j0 + jj >= len) continue;
        if (planes & 1u) f(0u, a_top ^ w_top ^ nmask);
        if (planes & 2u) f(1u, b_top);
        if (planes & 4u) f(2u, a_top ^ b_top);
        if (planes & 8u) f(3u, a_top | b_top);
    }
}

// ---------------------------------------------------------------------------
// block-level helpers (256 threads)
// ---------------------------------------------------------------------------
// exclusive scan of one value per thread; returns the exclusive prefix, *total = block sum
template <int NT>
__device__ __forceinline__ uint32_t block_scan(uint32_t v, uint32_t *wsum /* NT/64 words of LDS */, uint32_t *total)
{
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t n = __shfl_up(inc, o, 64);
        if (lane >= o) inc += n;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int i = 0; i < NW; ++i) {
        const uint32_t s = wsum[i];
        if (i < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// the same for 64-bit values (bucket offsets); wsum: NT/64 uint64 of LDS
template <int NT>
__device__ __forceinline__ uint64_t block_scan64(uint64_t v, uint64_t *wsum, uint64_t *total)
{
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t n = __shfl_up((unsigned long long) inc, o, 64);
        if (lane >= o) inc += n;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint64_t base = 0, tot = 0;
    for (int i = 0; i < NW; ++i) {
        const uint64_t s = wsum[i];
        if (i < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// exclusive scan of cnt[0..n) (n <= 1024) into base[]
template <int NT>
__device__ __forceinline__ void lds_scan(const uint32_t *cnt, uint32_t *base, uint32_t n, uint32_t *wsum)
{
    if (n <= (uint32_t) NT) {   // one counter per thread
        const uint32_t v = threadIdx.x < n ? cnt[threadIdx.x] : 0u;
        uint32_t tot;
        const uint32_t ex = block_scan<NT>(v, wsum, &tot);
        if (threadIdx.x < n) base[threadIdx.x] = ex;
        __syncthreads();
        return;
    }
    const uint32_t per = (n + NT - 1) / NT;
    const uint32_t b = threadIdx.x * per;
    uint32_t s = 0;
    for (uint32_t i = 0; i < per; ++i)
        if (b + i < n) s += cnt[b + i];
    uint32_t tot;
    uint32_t ex = block_scan<NT>(s, wsum, &tot);
    for (uint32_t i = 0; i < per; ++i)
        if (b + i < n) {
            base[b + i] = ex;
            ex += cnt[b + i];
        }
    __syncthreads();
}

// Copies one sorted run of n keys from LDS to out[dst ..) with 16-byte stores: the lanes g = 0..G-1 of a lane group
// share the run; up to 3 keys before the first 16-byte boundary and up to 3 after the last whole vector go out as
// single dwords.  Four runs per wave instruction (G = 16) quadruple the bytes per store instruction, which is what
// the write-out phase is bound by (store issue, ~7 B/clk/CU with one dword per lane).
__device__ __forceinline__ void write_run(uint32_t *__restrict__ out, unsigned long long dst, const uint32_t *sorted,
                                          uint32_t src, uint32_t n, uint32_t g, uint32_t G)
{
    const uint32_t head = min(n, (uint32_t) ((4u - (uint32_t) (dst & 3ull)) & 3u));
    if (g < head) out[dst + g] = sorted[src + g];
    const uint32_t nvec = (n - head) >> 2;
    const uint32_t s0 = src + head;
    uint4 *o4 = (uint4 *) (out + dst + head);
    for (uint32_t v = g; v < nvec; v += G) {
        const uint32_t *q = sorted + s0 + 4 * v;
        o4[v] = make_uint4(q[0], q[1], q[2], q[3]);
    }
    const uint32_t done = head + 4 * nvec;
    if (g < n - done) out[dst + done + g] = sorted[src + done + g];
}

// write_run for a run whose LDS copy starts at the same offset from a 16-byte boundary as its destination (src = dst mod 4
// words): the whole vectors are then single 16-byte LDS reads, too.
__device__ __forceinline__ void write_run_aligned(uint32_t *__restrict__ out, unsigned long long dst, const uint32_t *sorted,
                                                  uint32_t src, uint32_t n, uint32_t g, uint32_t G)
{
    const uint32_t head = min(n, (uint32_t) ((4u - (uint32_t) (dst & 3ull)) & 3u));
    if (g < head) out[dst + g] = sorted[src + g];
    const uint32_t nvec = (n - head) >> 2;
    const uint4 *s4 = (const uint4 *) (sorted + src + head);
    uint4 *o4 = (uint4 *) (out + dst + head);
    for (uint32_t v = g; v < nvec; v += G) o4[v] = s4[v];
    const uint32_t done = head + 4 * nvec;
    if (g < n - done) out[dst + done + g] = sorted[src + done + g];
}

// Packed form of write_run: the run's n keys (19 bits each) leave as ceil(n / 3) groups of 8 bytes, three keys per
// group, the last key repeated to fill the last group (setting a bit twice is harmless).  Groups are 8-byte aligned by
// construction, so there is no head / tail handling.  dst = group index.
__device__ __forceinline__ void write_run_p3(uint2 *__restrict__ out, unsigned long long dst, const uint32_t *sorted, uint32_t src,
                                             uint32_t n, uint32_t g, uint32_t G)
{
    const uint32_t ng = (n + 2) / 3;
    for (uint32_t v = g; v < ng; v += G) {
        const uint32_t i0 = 3 * v, i1 = min(i0 + 1, n - 1), i2 = min(i0 + 2, n - 1);
        const uint32_t k0 = sorted[src + i0], k1 = sorted[src + i1], k2 = sorted[src + i2];
        out[dst + v] = make_uint2(k0 | (k1 << 19), (k1 >> 13) | (k2 << 6));
    }
}

// Picks the next reads [r, r + R) of a block's range whose keys fit `cap`
// (R <= 256, at least 1 when the range is not empty) and builds the item
// table: istart[i] = first item of read i, one item per octet that can end a k-mer.
// kms = 4 * complete k-mers of read r+i (0 if not selected).
struct RoundPlan {
    uint32_t n_reads;
    uint32_t n_items;
};

template <int NT>
__device__ __forceinline__ RoundPlan plan_round(const ReadsView &rv, const uint32_t *__restrict__ kcnt,
                                                const uint64_t *__restrict__ sel, uint64_t r, uint64_t r_end,
                                                uint32_t key_cap, uint32_t item_cap, int k, uint32_t *istart /* NT+1 */,
                                                uint32_t *rd_len /* NT */, uint64_t *rd_t0 /* NT */, uint32_t *wsum,
                                                uint32_t *sh_n)
{
    const uint64_t me = r + threadIdx.x;
    uint32_t keys = 0, items = 0, len = 0;
    uint64_t t0 = 0;
    if (me < r_end) {
        const bool on = !sel || ((sel[me >> 6] >> (me & 63)) & 1ull);
        const uint32_t km = on ? kcnt[me] : 0;
        if (km) {
            read_extent(rv, me, t0, len);
            keys = 4 * km;
            items = octets_of(len, k);
        }
    }
    uint32_t tot, itot;
    const uint32_t kex = block_scan<NT>(keys, wsum, &tot);
    uint32_t iex = block_scan<NT>(items, wsum, &itot);
    // reads whose inclusive key / item prefixes fit (monotone in the thread index)
    const bool fits = (me < r_end) && (kex + keys <= key_cap) && (iex + items <= item_cap);
    if (threadIdx.x == 0) *sh_n = 0;
    __syncthreads();
    if (fits) atomicMax(sh_n, threadIdx.x + 1);
    __syncthreads();
    uint32_t R = *sh_n;
    if (R == 0 && r < r_end) R = 1;               // oversize read: the host never sends those here
    if (threadIdx.x >= R) items = 0;
    istart[threadIdx.x] = iex;
    rd_len[threadIdx.x] = len;
    rd_t0[threadIdx.x] = t0;
    if (threadIdx.x == R - 1) istart[NT] = iex + items;   // items of the round
    __syncthreads();
    RoundPlan p;
    p.n_reads = R;
    p.n_items = istart[NT];
    return p;
}

// item id -> (read slot, octet) by binary search in istart[0..R]
__device__ __forceinline__ void item_lookup(const uint32_t *istart, uint32_t R, uint32_t id, int k, uint32_t &slot, uint32_t &w)
{
    uint32_t lo = 0, hi = R;   // largest slot with istart[slot] <= id
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (istart[mid] <= id) lo = mid;
        else hi = mid;
    }
    slot = lo;
    w = id - istart[lo] + ((uint32_t) (k - 1) >> 3);
}

// ---------------------------------------------------------------------------
// Read numbers of the set bits of a selection bitmap, in order (round 3).  Commet.py's J2 / J3 jobs index a set restricted
// to the previous job's result (Commet.py:220, 233): ~22 % of the reads.  Walking such a selection with the round planner
// (plan_round: two block scans and barriers per 512 items, most of the reads it looks at not selected) made hist and
// scatter1 2.3x as expensive per indexed read as the arithmetic item path of unselected fixed-length sets.  With the
// selected reads' numbers in a list, item i of a piece is octet i % opr of read ids[i / opr]: the same arithmetic path.
// Three small launches per job: set bits per block of 4096 reads, their exclusive scan, the ids.
// ---------------------------------------------------------------------------
constexpr uint32_t IDS_BLOCK_WORDS = 64;                  // bitmap words (64 reads each) per block
// minus != nullptr: the reads whose bit is set there do not count (a search pass: selected and not yet tagged)
__global__ __launch_bounds__(64) void sel_count_kernel(const uint64_t *__restrict__ sel, uint64_t n_words, uint32_t *__restrict__ blk,
                                                       const uint64_t *__restrict__ minus = nullptr)
{
    const uint64_t w = (uint64_t) blockIdx.x * IDS_BLOCK_WORDS + threadIdx.x;
    uint32_t c = w < n_words ? (uint32_t) __popcll(sel[w] & ~(minus ? minus[w] : 0ull)) : 0u;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if (threadIdx.x == 0) blk[blockIdx.x] = c;
}

// in-place exclusive scan of blk[0 .. nb), blk[nb] = total; one workgroup, strips of 1024
__global__ __launch_bounds__(1024) void sel_scan_kernel(uint32_t *__restrict__ blk, uint32_t nb)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nb; b0 += 1024) {
        const uint32_t i = b0 + threadIdx.x;
        const uint32_t v = i < nb ? blk[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_scan<1024>(v, wsum, &tot);
        const uint32_t c = carry;
        if (i < nb) blk[i] = c + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) blk[nb] = carry;
}

__global__ __launch_bounds__(64) void sel_ids_kernel(const uint64_t *__restrict__ sel, uint64_t n_words, const uint32_t *__restrict__ blk,
                                                     uint32_t *__restrict__ ids, const uint64_t *__restrict__ minus = nullptr)
{
    const uint64_t w = (uint64_t) blockIdx.x * IDS_BLOCK_WORDS + threadIdx.x;
    uint64_t bits = w < n_words ? sel[w] & ~(minus ? minus[w] : 0ull) : 0ull;
    const uint32_t c = (uint32_t) __popcll(bits);
    uint32_t inc = c;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t x = __shfl_up(inc, o, 64);
        if ((int) threadIdx.x >= o) inc += x;
    }
    uint32_t at = blk[blockIdx.x] + inc - c;
    for (; bits; bits &= bits - 1ull) ids[at++] = (uint32_t) (w * 64ull + (uint64_t) (__ffsll((long long) bits) - 1));
}

// ---------------------------------------------------------------------------
// Item list of a chunk of RAGGED reads (round 6).  Real inputs are trimmed reads of many lengths (fastq_file.h:139-190): item i of a
// piece is then no arithmetic function of i, and the round planner (plan_round: per round two block scans, three barriers, the
// counts, the selection and two offsets of every read it looks at fetched on the spot, the item's words fetched behind them) made
// hist and scatter1 twice as expensive per read as on fixed-length sets.  Instead the chunk's items are written out once, one word
// each, in read order:
//     item = (triple << 4) | (octet in the word << 2) | min(word index in the read, 3)
// triple = the item's word triple counted from the chunk's first read (< 2^28: the host checks), and hist / scatter1 walk that list
// exactly as they walk a fixed-length set (LIST mode below): item i of a piece is items[i0 + i], the descriptor of the round after
// the coming one fetched a round ahead.  Reads without a complete k-mer (and reads a selection bitmap leaves out) have no items.
// What a k-mer may span is decided by the validity plane alone in this mode: bits past a read's end are zero in every packer
// (pack_reads_kernel, host/ingest_pack.hpp), and triples before the read's first are replaced by zeros through the word index.
// Two launches around a scan of the block sums: count, (sel_scan_kernel), fill.
// ---------------------------------------------------------------------------
constexpr uint32_t ITEMS_BLOCK = 1024;                   // reads per workgroup of the builder
constexpr uint32_t ITEM_MAX_TRIPLES = 1u << 28;
template <bool FILL>
__global__ __launch_bounds__(ITEMS_BLOCK) void part_items_kernel(ReadsView rv, const uint32_t *__restrict__ kcnt, const uint64_t *__restrict__ sel,
                                                                 uint64_t first, uint64_t count, int k, uint32_t *__restrict__ blk,
                                                                 uint32_t *__restrict__ items)
{
    __shared__ uint32_t wsum[ITEMS_BLOCK / 64];
    const uint64_t i = (uint64_t) blockIdx.x * ITEMS_BLOCK + threadIdx.x, r = first + i;
    uint32_t n = 0, len = 0;
    uint64_t t0 = 0;
    if (i < count) {
        const bool on = !sel || ((sel[r >> 6] >> (r & 63)) & 1ull);
        if (on && kcnt[r]) {
            read_extent(rv, r, t0, len);
            n = octets_of(len, k);
        }
    }
    uint32_t tot;
    const uint32_t ex = block_scan<ITEMS_BLOCK>(n, wsum, &tot);
    if (!FILL) {
        if (threadIdx.x == 0) blk[blockIdx.x] = tot;
        return;
    }
    if (!n) return;
    uint64_t tb;
    uint32_t l0;
    read_extent(rv, first, tb, l0);
    const uint32_t trel = (uint32_t) (t0 - tb), q_first = (uint32_t) (k - 1) >> 3;
    uint32_t at = blk[blockIdx.x] + ex;
    for (uint32_t q = q_first; q < q_first + n; ++q) items[at++] = ((trel + (q >> 2)) << 4) | ((q & 3u) << 2) | min(q >> 2, 3u);
}

// an item of the list -> what the key loops take: p = the read's first triple as far as they may look back (at most three words),
// q = the octet's number from there; the read's length is not known here (0xFFFFFFFF: the validity plane decides)
struct ListItem {
    const uint32_t *p;
    uint32_t q;
};
__device__ __forceinline__ ListItem list_item(const uint32_t *chunk_planes, uint32_t d)
{
    const uint32_t wc = d & 3u;
    ListItem it;
    it.p = chunk_planes + 3ull * ((uint64_t) (d >> 4) - wc);
    it.q = 4u * wc + ((d >> 2) & 3u);
    return it;
}

// ---------------------------------------------------------------------------
// hist: bucket histogram of the chunk, buckets [b_lo, b_lo + n_b) in LDS
// ---------------------------------------------------------------------------
// UNI: every read has rv.uniform_len bases and no selection bitmap applies.  Item i of a block's read range is then
// octet (i % opr) of read (i / opr) — no round planning, no barriers in the loop (reads without a complete k-mer just
// yield no key).
// The read range is cut exactly like scatter1 cuts it (n_blk1 workgroups of per1 = ceil(count / n_blk1) reads); one
// hist workgroup takes the ranges of scatter1 workgroups 2b and 2b+1 one after the other and also leaves their
// coarse-bucket counts in blockcnt[j * nb1 + c]: scatter1 then knows where each of its runs goes without reserving
// space with global atomics (their round trip used to sit in every round).
// FULL: the LDS histogram covers every bucket (b_lo == 0, n_b == nb; always the case for k <= 32): no range test.
// MODE: 0 = rounds planned over the reads (any set, any selection), 1 = UNI, 2 = LIST: the chunk's item list (part_items_kernel;
// items[0 .. *n_items)), cut into n_blk1 pieces of equal ITEM counts
template <typename W, int MODE, bool FULL>
__global__ __launch_bounds__(HIST_NT) void part_hist_kernel(ReadsView rv, const uint32_t *__restrict__ kcnt,
                                                            const uint64_t *__restrict__ sel, uint64_t first,
                                                            uint64_t count, PartGeom g, uint32_t b_lo, uint32_t n_b,
                                                            uint32_t *__restrict__ hist, uint32_t n_blk1,
                                                            uint32_t *__restrict__ blockcnt, const uint32_t *__restrict__ ids,
                                                            const uint32_t *__restrict__ items, const uint32_t *__restrict__ n_items)
{
    // UNI with ids != nullptr: [first, first + count) are POSITIONS in the list of selected reads (sel_ids_kernel)
    constexpr int NT = HIST_NT;
    constexpr bool UNI = MODE == 1, LIST = MODE == 2;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *h = smem;                            // n_b counters
    uint32_t *istart = h + n_b;                    // NT + 4
    uint32_t *rd_len = istart + NT + 4;            // NT
    uint32_t *wsum = rd_len + NT;                  // 16
    uint32_t *sh_n = wsum + 16;                    // 4
    uint64_t *rd_t0 = (uint64_t *) (sh_n + 4);     // NT
    for (uint32_t i = threadIdx.x; i < n_b; i += NT) h[i] = 0;
    __syncthreads();
    const uint64_t l_total = LIST ? (uint64_t) *n_items : 0;
    const uint64_t per1 = ((LIST ? l_total : count) + n_blk1 - 1) / n_blk1;
    const uint32_t *l_planes = rv.planes;                 // LIST: the chunk's first triple
    if (LIST) {
        uint64_t tb;
        uint32_t l0;
        read_extent(rv, first, tb, l0);
        l_planes += 3 * tb;
    }
    const uint32_t nsub = 1u << g.b2, c_lo = b_lo >> g.b2, n_c = n_b >> g.b2;   // coarse buckets of this pass
    uint32_t prev = 0;   // thread c < n_c: keys of coarse bucket c_lo + c counted before this half
    uint32_t planes = 0;  // planes with buckets in [b_lo, b_lo + n_b)
    for (uint32_t pl = 0; pl < 4; ++pl)
        if (((pl + 1) << g.plane_shift) > b_lo && (pl << g.plane_shift) < b_lo + n_b) planes |= 1u << pl;
    if (FULL) planes = 15u;
    // k <= 32 (always FULL): the buckets come from rolled top bits (for_each_bucket32)
    constexpr bool ROLL = FULL && sizeof(W) == 4 && HIST_ROLLING;
    constexpr bool ROLL64 = !FULL && sizeof(W) == 8 && HIST_ROLLING;       // k = 33, 34: the same rolling, pass by pass
    auto add_bucket = [&](uint32_t plane, uint32_t bucket) { atomicAdd(h + (plane << g.plane_shift) + bucket, 1u); };
    // (a pass of 64-bit keys covers whole planes: HIST_MAX_BUCKETS is a multiple of a plane's 2^(k - 19) buckets, so the
    // planes mask alone decides and the bucket's place in the pass is its number minus b_lo)
    auto add_bucket_rel = [&](uint32_t plane, uint32_t bucket) { atomicAdd(h + ((plane << g.plane_shift) + bucket - b_lo), 1u); };
    auto add = [&](uint32_t plane, W key) {
        if (FULL) {
            atomicAdd(h + (plane << g.plane_shift) + (uint32_t) (key >> TILE_BITS), 1u);
        } else {
            const uint32_t b = (plane << g.plane_shift) | (uint32_t) (key >> TILE_BITS);
            const uint32_t rel = b - b_lo;
            if (rel < n_b) atomicAdd(&h[rel], 1u);
        }
    };
    for (uint32_t half = 0; half < 2; ++half) {
        const uint64_t j = 2ull * blockIdx.x + half;
        if (j >= n_blk1) break;   // uniform
        uint64_t r = min(first + count, first + j * per1);
        const uint64_t r_end = min(first + count, r + per1);
        if (LIST) {
            const uint64_t i0 = min(l_total, j * per1), i1 = min(l_total, i0 + per1);
            uint64_t id = i0 + threadIdx.x;
            uint32_t d = id < i1 ? items[id] : 0u;
            for (; id < i1; id += NT) {
                const uint32_t dn = id + NT < i1 ? items[id + NT] : 0u;      // (the next item's descriptor travels while this one's keys are counted)
                const ListItem li = list_item(l_planes, d);
                if constexpr (ROLL) for_each_bucket32(li.p, 0xFFFFFFFFu, li.q, g.k, add_bucket);
                else if constexpr (ROLL64) for_each_bucket64(li.p, 0xFFFFFFFFu, li.q, g.k, planes, add_bucket_rel);
                else for_each_key<W, true>(li.p, 0xFFFFFFFFu, li.q, g.k, add, planes);
                d = dn;
            }
        } else if (UNI) {
            const uint32_t L = rv.uniform_len;
            const uint32_t opr = max(octets_of(L, g.k), 1u), q_first = (uint32_t) (g.k - 1) >> 3;
            const uint64_t total = (r < r_end && L >= (uint32_t) g.k) ? (r_end - r) * opr : 0;
            uint64_t rd = r + threadIdx.x / opr;
            uint32_t q = threadIdx.x % opr;
            const uint32_t dpos = NT / opr, dq = NT % opr;
            for (uint64_t id = threadIdx.x; id < total; id += NT) {
                const uint64_t rn = ids ? (uint64_t) ids[rd] : rd;          // the read's number
                const uint32_t *rp = rv.planes + 3 * (((rn * L) >> 5) + rn);
                if constexpr (ROLL) for_each_bucket32(rp, L, q + q_first, g.k, add_bucket);
                else if constexpr (ROLL64) for_each_bucket64(rp, L, q + q_first, g.k, planes, add_bucket_rel);
                else for_each_key<W, true>(rp, L, q + q_first, g.k, add, planes);
                rd += dpos, q += dq;
                if (q >= opr) q -= opr, ++rd;
            }
        } else {
            while (r < r_end) {
                const RoundPlan rp = plan_round<NT>(rv, kcnt, sel, r, r_end, 0xFFFFFFFFu, 0xFFFFFFFFu, g.k, istart, rd_len,
                                                    rd_t0, wsum, sh_n);
                for (uint32_t id = threadIdx.x; id < rp.n_items; id += NT) {
                    uint32_t slot, q;
                    item_lookup(istart, rp.n_reads, id, g.k, slot, q);
                    if constexpr (ROLL) for_each_bucket32(rv.planes + 3 * rd_t0[slot], rd_len[slot], q, g.k, add_bucket);
                    else if constexpr (ROLL64) for_each_bucket64(rv.planes + 3 * rd_t0[slot], rd_len[slot], q, g.k, planes, add_bucket_rel);
                    else for_each_key<W, true>(rv.planes + 3 * rd_t0[slot], rd_len[slot], q, g.k, add, planes);
                }
                __syncthreads();
                r += rp.n_reads;
            }
        }
        __syncthreads();
        if (threadIdx.x < n_c) {   // coarse bucket = nsub consecutive final buckets; rotated start: no bank conflicts
            uint32_t sum = 0;
            for (uint32_t i = 0; i < nsub; ++i) sum += h[(threadIdx.x << g.b2) + ((i + threadIdx.x) & (nsub - 1))];
            blockcnt[j * g.nb1 + c_lo + threadIdx.x] = sum - prev;
            prev = sum;
        }
        __syncthreads();
    }
    for (uint32_t i = threadIdx.x; i < n_b; i += NT) {
        const uint32_t v = h[i];
        if (v) atomicAdd(&hist[b_lo + i], v);
    }
}

// ---------------------------------------------------------------------------
// scan: bucket offsets, cursors, build work list.  One block of 1024 threads.
// ---------------------------------------------------------------------------
// fill_empty != 0: the filter was NOT zeroed beforehand, so empty buckets get a work item too (their tile is written
// as zeros) and part_zero_split_kernel clears the tiles that several workgroups will OR into.
__global__ __launch_bounds__(1024) void part_scan_kernel(const uint32_t *__restrict__ hist, PartGeom g, int fill_empty,
                                                         uint64_t *__restrict__ off /* nb+1 */,
                                                         unsigned long long *__restrict__ cursor2 /* nb */,
                                                         uint32_t *__restrict__ wl_off /* nb+1 */,
                                                         uint64_t *__restrict__ goff /* nb+1, packed geometry */,
                                                         int lds_hist /* nb * 4 bytes of dynamic LDS were given */)
{
    __shared__ uint64_t s_w64[16];
    __shared__ uint32_t s_w32[16];
    __shared__ uint64_t c_off[MAX_L1 + 1];               // key offsets of the coarse buckets
    extern __shared__ uint32_t h_lds[];                   // the histogram, staged with coalesced loads when it fits (lds_hist)
    // (one idle word per 32: thread x walks the counters x * per ... of its own, and with per = 32 every lane of a wave
    // would otherwise sit in the same LDS bank — 64 cycles per access instead of 2, 40 % of this kernel's 110 us)
    const bool staged = lds_hist != 0;
    if (staged) {
        for (uint32_t i = threadIdx.x; i < g.nb; i += 1024) h_lds[i + (i >> 5)] = hist[i];
        __syncthreads();
    }
    auto count_of = [&](uint32_t b) { return staged ? h_lds[b + (b >> 5)] : hist[b]; };
    const uint32_t per = (g.nb + 1023) / 1024;
    const uint32_t b0 = threadIdx.x * per;
    const uint32_t sub_mask = (1u << g.b2) - 1u;
    uint64_t s = 0;
    uint32_t wl = 0;
    for (uint32_t i = 0; i < per; ++i)
        if (b0 + i < g.nb) {
            const uint32_t c = count_of(b0 + i);
            s += c;
            wl += (c || !fill_empty) ? (c + BUILD_CAP - 1) / BUILD_CAP : 1u;   // empty buckets: no workgroup unless they must be zero-filled
        }
    uint64_t s_tot;
    uint32_t wl_tot;
    uint64_t ex = block_scan64<1024>(s, s_w64, &s_tot);
    uint32_t wex = block_scan<1024>(wl, s_w32, &wl_tot);
    for (uint32_t i = 0; i < per; ++i)
        if (b0 + i < g.nb) {
            const uint32_t b = b0 + i;
            const uint32_t c = count_of(b);
            off[b] = ex;
            if ((b & sub_mask) == 0) c_off[b >> g.b2] = ex;
            if (!g.packed) cursor2[b] = ex;
            wl_off[b] = wex;
            ex += c;
            wex += (c || !fill_empty) ? (c + BUILD_CAP - 1) / BUILD_CAP : 1u;
        }
    if (threadIdx.x == 1023) {
        off[g.nb] = s_tot;
        wl_off[g.nb] = wl_tot;
        c_off[g.nb1] = s_tot;
    }
    if (!g.packed) return;
    // Packed geometry: a final bucket is a sequence of 8-byte groups of three keys.  Every (slab piece, final bucket)
    // run of scatter2 is rounded up to whole groups (its last key repeated), so a bucket of c keys that receives runs
    // from P pieces needs at most c / 3 + P groups; P = the slabs of S2_KEYS keys that overlap its coarse bucket.
    __syncthreads();
    auto groups_of = [&](uint32_t b) -> uint64_t {
        const uint32_t c = count_of(b);
        const uint64_t lo = c_off[b >> g.b2], hi = c_off[(b >> g.b2) + 1];
        const uint64_t pieces = hi > lo ? (hi - 1) / S2_KEYS - lo / S2_KEYS + 1 : 0;
        return c ? c / 3 + pieces + 1 : 0;
    };
    uint64_t gs = 0;
    for (uint32_t i = 0; i < per; ++i)
        if (b0 + i < g.nb) gs += groups_of(b0 + i);
    uint64_t g_tot;
    uint64_t gex = block_scan64<1024>(gs, s_w64, &g_tot);
    for (uint32_t i = 0; i < per; ++i)
        if (b0 + i < g.nb) {
            const uint32_t b = b0 + i;
            goff[b] = gex;
            cursor2[b] = gex;
            gex += groups_of(b);
        }
    if (threadIdx.x == 1023) goff[g.nb] = g_tot;
}

// blockoff[j * nb1 + c] = where scatter1 workgroup j starts writing in coarse bucket c
//                        = off[c << b2] + keys of c counted for workgroups 0 .. j-1.   One workgroup per c;
// thread x takes the BLOCKOFF_PER consecutive workgroups j = x * BLOCKOFF_PER ... (n_blk1 <= S1_GRID_MAX).
constexpr uint32_t BLOCKOFF_PER = (S1_GRID_MAX + 511) / 512;
__global__ __launch_bounds__(512) void part_blockoff_kernel(const uint32_t *__restrict__ blockcnt,
                                                            const uint64_t *__restrict__ off, PartGeom g, uint32_t n_blk1,
                                                            unsigned long long *__restrict__ blockoff)
{
    __shared__ uint64_t s_sum[512];
    const uint32_t c = blockIdx.x;
    uint32_t cntv[BLOCKOFF_PER];
    uint64_t v = 0;
#pragma unroll
    for (uint32_t i = 0; i < BLOCKOFF_PER; ++i) {
        const uint32_t j = threadIdx.x * BLOCKOFF_PER + i;
        cntv[i] = j < n_blk1 ? blockcnt[(uint64_t) j * g.nb1 + c] : 0u;
        v += cntv[i];
    }
    s_sum[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t o = 1; o < 512; o <<= 1) {
        const uint64_t a = threadIdx.x >= o ? s_sum[threadIdx.x - o] : 0;
        __syncthreads();
        s_sum[threadIdx.x] += a;
        __syncthreads();
    }
    uint64_t at = off[(uint64_t) c << g.b2] + (s_sum[threadIdx.x] - v);
#pragma unroll
    for (uint32_t i = 0; i < BLOCKOFF_PER; ++i) {
        const uint32_t j = threadIdx.x * BLOCKOFF_PER + i;
        if (j < n_blk1) blockoff[(uint64_t) j * g.nb1 + c] = at;
        at += cntv[i];
    }
}

// ---------------------------------------------------------------------------
// scatter1: reads -> keys -> 2^b1 coarse buckets (LDS counting sort per round)
// payload = ((bucket & (2^b2 - 1)) << 19) | (key & TILE_MASK)
// Every thread owns one octet item per round (rounds are cut at S1_NT items or S1_KEYS keys,
// whichever comes first); for 32-bit keys the (keya, keyb) pairs are computed once and kept
// in registers for both passes.
// ---------------------------------------------------------------------------
// UNI (see part_hist_kernel): items come from arithmetic instead of plan_round, and the read words of the NEXT round's
// item are loaded right after this round's keys are made, so their latency hides behind the sort and the write-out.
// MODE as in part_hist_kernel; UNI below = "items come without planning" (modes 1 and 2), LIST = mode 2
template <typename W, int MODE>
__global__ __launch_bounds__(S1_NT, 4) void part_scatter1_kernel(ReadsView rv, const uint32_t *__restrict__ kcnt,
                                                              const uint64_t *__restrict__ sel, uint64_t first,
                                                              uint64_t count, PartGeom g,
                                                              const unsigned long long *__restrict__ blockoff,
                                                              uint32_t *__restrict__ out, const uint32_t *__restrict__ ids,
                                                              const uint32_t *__restrict__ items, const uint32_t *__restrict__ n_items)
{
    // mode 1 with ids != nullptr: [first, first + count) are POSITIONS in the list of selected reads (sel_ids_kernel)
    constexpr int NT = S1_NT;
    constexpr bool UNI = MODE != 0, LIST = MODE == 2;
    constexpr bool WIDE = sizeof(W) == 8;          // 33 <= k <= 34: keys of 33 / 34 bits
    constexpr uint32_t NTR = WIDE ? 3u : 2u;       // word triples a k-mer window can span
    using T = KeyTraits<W>;
    // ALIGNED: a run's LDS copy starts at the same offset from a 16-byte boundary as its place in bufA (up to 3 + 3 idle
    // slots per run), so that the write-out reads whole vectors from LDS.  Only where the LDS allows it beside two
    // workgroups per CU: the round planner's tables of the other variant take that room.
    constexpr bool ALIGNED = UNI && S1_ALIGNED;
    __shared__ __attribute__((aligned(16))) uint32_t sorted[S1_KEYS + (ALIGNED ? 6 * MAX_L1 : 0)];
    __shared__ uint32_t cnt[MAX_L1], base[MAX_L1];
    __shared__ unsigned long long gbase[MAX_L1], gcur[MAX_L1];   // this round's / the next round's output position per coarse bucket
    __shared__ uint32_t istart[UNI ? 4 : NT + 4], rd_len[UNI ? 1 : NT], wsum[16], sh_n[4];
    __shared__ uint64_t rd_t0[UNI ? 1 : NT];
    const uint64_t per = (count + gridDim.x - 1) / gridDim.x;   // the cut part_hist_kernel counted with
    uint64_t r = min(first + count, first + blockIdx.x * per);
    const uint64_t r_end = min(first + count, r + per);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const W mask = (g.k == T::BITS) ? ~(W) 0 : (((W) 1 << g.k) - 1);
    const int sh = T::BITS - g.k;
    if (threadIdx.x < g.nb1) gcur[threadIdx.x] = blockoff[(uint64_t) blockIdx.x * g.nb1 + threadIdx.x];   // first use is behind a barrier
    // coarse bucket of a key of plane p = p * nbp + (key >> sA); payload = the key's low sA bits
    // (= ((bucket & sub_mask) << TILE_BITS) | (key & TILE_MASK), the final-bucket bits sit right above the tile bits;
    // sA = TILE_BITS + b2 <= 28, so the payload always comes from the key's low word)
    const uint32_t sA = TILE_BITS + g.b2, nbp = g.nb1 >> 2, pay_mask = (1u << sA) - 1u;
    // UNI: this thread's item of the coming round = octet u_q + q_first of read u_rd
    const uint32_t q_first = (uint32_t) (g.k - 1) >> 3;
    const uint32_t opr = (UNI && !LIST) ? max(octets_of(rv.uniform_len, g.k), 1u) : 1u;
    // LIST: this workgroup's piece of the chunk's item list = items[l_i0 .. l_i0 + u_total)
    const uint64_t l_all = LIST ? (uint64_t) *n_items : 0, l_per = (l_all + gridDim.x - 1) / gridDim.x;
    const uint64_t l_i0 = min(l_all, (uint64_t) blockIdx.x * l_per);
    const uint32_t *l_planes = rv.planes;                 // LIST: the chunk's first triple
    if (LIST) {
        uint64_t tb;
        uint32_t l0;
        read_extent(rv, first, tb, l0);
        l_planes += 3 * tb;
    }
    const uint64_t u_total = LIST ? min(l_all, l_i0 + l_per) - l_i0
                                  : (UNI && r < r_end && rv.uniform_len >= (uint32_t) g.k) ? (r_end - r) * opr : 0;
    const uint32_t u_dpos = NT / opr, u_dq = NT % opr;
    uint64_t u_done = 0, u_rd = r + threadIdx.x / opr;
    uint32_t u_q = threadIdx.x % opr;
    uint32_t l_cur = 0, l_next = 0;                       // LIST: descriptors of this thread's item of the coming round / of the round after
    auto uni_ptr = [&](uint64_t rd) { return rv.planes + 3 * (((rd * rv.uniform_len) >> 5) + rd); };
    // prefetched word triples w-NTR+1 .. w of the coming item, RAW: no branch and no use between the loads and the
    // claim below, so that they really stay in flight (a triple before the read's first is read as triple 0 and
    // zeroed at use)
    uint32_t pre[3 * NTR];
#pragma unroll
    for (uint32_t i = 0; i < 3 * NTR; ++i) pre[i] = 0;
    auto pre_load = [&](const uint32_t *p, uint32_t w) {
#pragma unroll
        for (uint32_t t = 0; t < NTR; ++t) {
            const uint32_t back = NTR - 1 - t;
            const uint32_t *q = p + 3 * (w >= back ? w - back : 0u);
            pre[3 * t] = q[0], pre[3 * t + 1] = q[1], pre[3 * t + 2] = q[2];
        }
    };
    // vmcnt counts loads and stores in issue order, so a wait for these words that the compiler places behind the
    // write-out would also wait for that round's stores to be acknowledged (their whole HBM latency, every round).
    // Claiming the words right before the write-out costs nothing: the loads were issued half a round earlier and the
    // only stores still in flight are the previous round's.
    auto pre_claim = [&]() {
#pragma unroll
        for (uint32_t i = 0; i < 3 * NTR; ++i) asm volatile("" : "+v"(pre[i]));
    };
    // (threads without an item load the block's first triple instead: an unconditional load needs no register copies,
    // which the compiler would otherwise park right behind the load together with a wait)
    // ids: the read number of an item is a load of its own; the one of the item AFTER the coming one is fetched a round ahead
    // (nid), so that the prefetch of the words never waits for it
    auto read_no = [&](uint64_t pos) -> uint64_t { return ids ? (uint64_t) ids[pos] : pos; };
    auto advance = [&](uint64_t &rd, uint32_t &q) {
        rd += u_dpos, q += u_dq;
        if (q >= opr) q -= opr, ++rd;
    };
    uint64_t nid = 0;                                    // read number of the item of the round after the coming one
    if (LIST && u_total) {
        l_cur = threadIdx.x < u_total ? items[l_i0 + threadIdx.x] : 0u;          // (no item: the chunk's first triple, never used)
        const ListItem li = list_item(l_planes, l_cur);
        pre_load(li.p, li.q >> 2);
        pre_claim();
        l_next = (uint64_t) NT + threadIdx.x < u_total ? items[l_i0 + NT + threadIdx.x] : 0u;
    } else if (UNI && u_total) {
        const bool in = threadIdx.x < u_total;
        pre_load(uni_ptr(read_no(in ? u_rd : r)), in ? (u_q + q_first) >> 2 : 0u);
        pre_claim();
        uint64_t rd2 = u_rd;
        uint32_t q2 = u_q;
        advance(rd2, q2);
        nid = read_no((uint64_t) NT + threadIdx.x < u_total ? rd2 : r);
    }
    while (UNI ? u_done < u_total : r < r_end) {
        RoundPlan rp;
        rp.n_reads = 0, rp.n_items = 0;
        if (!UNI) rp = plan_round<NT>(rv, kcnt, sel, r, r_end, S1_KEYS, S1_ITEMS, g.k, istart, rd_len, rd_t0, wsum, sh_n);
        if (threadIdx.x < g.nb1) cnt[threadIdx.x] = 0;
        __syncthreads();
        // my item of the round: low words of keya, keyb, psi(keya) of its 8 positions, the bits above them (6 per
        // position, wide keys only), the ranks from the counting pass (two per word), valid positions
        uint32_t cka[8], ckb[8], cpa[8], crk[16], chi[2] = {0, 0}, cvalid = 0;
        bool ion;
        uint32_t iq = 0, ilen = 0;
        const uint32_t *ip = rv.planes;
        if (LIST) {
            ion = u_done + threadIdx.x < u_total;
            iq = 4u * (l_cur & 3u) + ((l_cur >> 2) & 3u), ilen = 0xFFFFFFFFu;      // (list_item's q: the validity plane knows the read's end)
        } else if (UNI) {
            ion = u_done + threadIdx.x < u_total;
            iq = u_q + q_first, ilen = rv.uniform_len;
        } else {
            ion = threadIdx.x < rp.n_items;
            if (ion) {
                uint32_t slot;
                item_lookup(istart, rp.n_reads, threadIdx.x, g.k, slot, iq);
                ip = rv.planes + 3 * rd_t0[slot];
                ilen = rd_len[slot];
            }
        }
        // pass A: keys, count per coarse bucket; the returned value is the key's rank inside its coarse bucket for
        // this round (< S1_KEYS <= 2^16)
        if (ion) {
            const uint32_t w = iq >> 2, j0 = (iq & 3u) * 8u;
            ItemWords<W> it;
            if (UNI) {
#pragma unroll
                for (uint32_t t = 0; t < NTR; ++t) {
                    const bool there = w >= NTR - 1 - t;
                    it.hi[t] = there ? pre[3 * t] : 0u, it.lo[t] = there ? pre[3 * t + 1] : 0u, it.va[t] = there ? pre[3 * t + 2] : 0u;
                }
            } else {
                it.load(ip, w);
            }
            uint32_t *const cnt_b = cnt + nbp, *const cnt_c = cnt + 2 * nbp, *const cnt_d = cnt + 3 * nbp;
#pragma unroll
            for (uint32_t jj = 0; jj < 8; ++jj) {
                const uint32_t j = j0 + jj, pos = 32u * w + j;
                W wh = 0, wl = 0;
                const bool ok = it.window(j, g.k, mask, wh, wl) && (pos + 1u >= (uint32_t) g.k) && (pos < ilen);
                const W ka = T::brev(wh) >> sh, kb = T::brev(wl) >> sh;
                const W pa = psi_a<W>(ka, g.k);
                cka[jj] = (uint32_t) ka;
                ckb[jj] = (uint32_t) kb;
                cpa[jj] = (uint32_t) pa;
                if (WIDE)   // k <= 34: at most two bits above the low word
                    chi[jj >> 2] |= ((uint32_t) ((uint64_t) ka >> 32) | ((uint32_t) ((uint64_t) kb >> 32) << 2) | ((uint32_t) ((uint64_t) pa >> 32) << 4))
                                    << (6u * (jj & 3u));
                if (ok && !(COMMET_ABLATE & 4)) {   // crk[] of other positions is never read
                    cvalid |= 1u << jj;
                    const uint32_t r0 = atomicAdd(cnt + (uint32_t) (pa >> sA), 1u);
                    const uint32_t r1 = atomicAdd(cnt_b + (uint32_t) (kb >> sA), 1u);
                    const uint32_t r2 = atomicAdd(cnt_c + (uint32_t) ((ka ^ kb) >> sA), 1u);
                    const uint32_t r3 = atomicAdd(cnt_d + (uint32_t) ((ka | kb) >> sA), 1u);
                    crk[2 * jj] = r0 | (r1 << 16);
                    crk[2 * jj + 1] = r2 | (r3 << 16);
                }
            }
        }
        if (LIST) {  // next round's item: its words travel while this round is sorted and written, the descriptor of the round after it too
            u_done += NT;
            const ListItem li = list_item(l_planes, l_next);
            pre_load(li.p, li.q >> 2);
            l_cur = l_next;
            l_next = u_done + NT + threadIdx.x < u_total ? items[l_i0 + u_done + NT + threadIdx.x] : 0u;
        } else if (UNI) {   // next round's item; its words travel while this round is sorted and written
            u_done += NT;
            advance(u_rd, u_q);
            const bool in = u_done + threadIdx.x < u_total;
            pre_load(uni_ptr(nid), in ? (u_q + q_first) >> 2 : 0u);          // (nid = the block's first read when there is no item)
            uint64_t rd2 = u_rd;
            uint32_t q2 = u_q;
            advance(rd2, q2);
            nid = read_no(u_done + NT + threadIdx.x < u_total ? rd2 : r);    // used a round from now
        }
        __syncthreads();
        if (!ALIGNED) {
            lds_scan<NT>(cnt, base, g.nb1, wsum);
            if (threadIdx.x < g.nb1) {   // exact positions: the hist pass counted this workgroup's keys per coarse bucket
                const unsigned long long at = gcur[threadIdx.x];
                gbase[threadIdx.x] = at;
                gcur[threadIdx.x] = at + cnt[threadIdx.x];
            }
        } else {   // the same, the run of bucket c starting at (a multiple of 4) + (its place in bufA mod 4)
            const bool mine = threadIdx.x < g.nb1;   // nb1 <= MAX_L1 < NT
            const uint32_t c = mine ? cnt[threadIdx.x] : 0u;
            const unsigned long long at = mine ? gcur[threadIdx.x] : 0ull;
            const uint32_t ph = (uint32_t) (at & 3ull);
            uint32_t tot;
            const uint32_t ex = block_scan<NT>(mine ? ((ph + c + 3u) & ~3u) : 0u, wsum, &tot);
            if (mine) {
                base[threadIdx.x] = ex + ph;
                gbase[threadIdx.x] = at;
                gcur[threadIdx.x] = at + c;
            }
            __syncthreads();
        }
        // pass B: place every key at its bucket's base + its rank
        if (ion && !(COMMET_ABLATE & 2)) {
            const uint32_t *const base_b = base + nbp, *const base_c = base + 2 * nbp, *const base_d = base + 3 * nbp;
#pragma unroll
            for (uint32_t jj = 0; jj < 8; ++jj) {
                if (!((cvalid >> jj) & 1u)) continue;
                const uint32_t ka = cka[jj], kb = ckb[jj], pa = cpa[jj], kc = ka ^ kb, kd = ka | kb;
                uint32_t i0 = pa >> sA, i1 = kb >> sA, i2 = kc >> sA, i3 = kd >> sA;
                if (WIDE) {   // the bits above the low word belong to the bucket number
                    const uint32_t h6 = chi[jj >> 2] >> (6u * (jj & 3u));
                    const uint32_t ha = h6 & 3u, hb = (h6 >> 2) & 3u, hp = (h6 >> 4) & 3u, up = 32u - sA;
                    i0 |= hp << up, i1 |= hb << up, i2 |= (ha ^ hb) << up, i3 |= (ha | hb) << up;
                }
                const uint32_t p0 = base[i0] + (crk[2 * jj] & 0xFFFFu);
                const uint32_t p1 = base_b[i1] + (crk[2 * jj] >> 16);
                const uint32_t p2 = base_c[i2] + (crk[2 * jj + 1] & 0xFFFFu);
                const uint32_t p3 = base_d[i3] + (crk[2 * jj + 1] >> 16);
                // p < S1_KEYS: a round holds at most S1_ITEMS * 32 = S1_KEYS keys
                sorted[p0] = pa & pay_mask;
                sorted[p1] = kb & pay_mask;
                sorted[p2] = kc & pay_mask;
                sorted[p3] = kd & pay_mask;
            }
        }
        __syncthreads();
        if (UNI) pre_claim();
        // write-out: one wave per run, consecutive lanes -> consecutive addresses
        if (!(COMMET_ABLATE & 1))
        for (uint32_t c1 = wave * 4 + (lane >> 4); c1 < g.nb1; c1 += (NT / 64) * 4)
        {
            if (ALIGNED) write_run_aligned(out, gbase[c1], sorted, base[c1], cnt[c1], lane & 15u, 16u);
            else write_run(out, gbase[c1], sorted, base[c1], cnt[c1], lane & 15u, 16u);
        }
        __syncthreads();
        r += rp.n_reads;
    }
}

// ---------------------------------------------------------------------------
// scatter2: coarse buckets -> final buckets.  Flat grid over bufA; a slab that
// straddles coarse buckets is processed segment by segment.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(S2_NT, S2P_WAVES) void part_scatter2_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                              const uint64_t *__restrict__ off, PartGeom g,
                                                              unsigned long long *__restrict__ cursor2, uint64_t total)
{
    constexpr int NT = S2_NT;
    __shared__ uint32_t sorted[S2_KEYS];
    __shared__ uint32_t cnt[MAX_SUB], base[MAX_SUB], fill[MAX_SUB];
    __shared__ unsigned long long gbase[MAX_SUB];
    __shared__ uint32_t wsum[16];
    const uint32_t nsub = 1u << g.b2;
    // Slab order (option "s2_swizzle" = G, 0 = dispatch order): workgroups b with equal b % G walk one of G contiguous
    // ranges of slabs, so the workgroups that run at the same time work on G different places of bufA, i.e. on
    // different coarse buckets, instead of all reserving space through the same 2^b2 cursors.  G = 8 is the
    // XCD-contiguous order (all writers of a final bucket behind one L2).  Placement never affects results.
    // Bijective for any grid size: range x owns q (+1 if x < rem) slabs starting at x*q + min(x, rem).
    const uint32_t G = (uint32_t) g.xcd_swizzle;
    uint64_t slab = blockIdx.x;
    if (G > 1) {
        const uint32_t q = gridDim.x / G, rem = gridDim.x % G, x = blockIdx.x % G;
        slab = (uint64_t) x * q + min(x, rem) + blockIdx.x / G;
    }
    const uint64_t s0 = slab * S2_KEYS;
    if (s0 >= total) return;
    const uint64_t s1 = min(total, s0 + S2_KEYS);
    // coarse bucket containing s0: largest c with off[c << b2] <= s0
    uint32_t lo = 0, hi = g.nb1;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[(uint64_t) mid << g.b2] <= s0) lo = mid;
        else hi = mid;
    }
    uint32_t c1 = lo;
    uint64_t pos = s0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    while (pos < s1) {
        const uint64_t c_end = off[(uint64_t) (c1 + 1) << g.b2];
        const uint64_t seg_end = min(s1, c_end);
        if (seg_end <= pos) {   // empty coarse bucket
            ++c1;
            continue;
        }
        const uint32_t n = (uint32_t) (seg_end - pos);
        for (uint32_t i = threadIdx.x; i < nsub; i += NT) cnt[i] = 0, fill[i] = 0;
        __syncthreads();
        uint32_t key[S2_PER_THREAD];
        const bool whole = n == S2_KEYS;   // (uniform) a whole slab inside one coarse bucket, the usual case: no per-key bounds tests
        if (whole) {
            // (pos = slab * S2_KEYS here: 16-byte aligned; which thread sorts which key does not matter)
            const uint4 *in4 = (const uint4 *) (in + pos);
#pragma unroll
            for (uint32_t q = 0; q < S2_PER_THREAD / 4; ++q) {
                const uint4 v = in4[threadIdx.x + NT * q];
                key[4 * q] = v.x, key[4 * q + 1] = v.y, key[4 * q + 2] = v.z, key[4 * q + 3] = v.w;
            }
#pragma unroll
            for (uint32_t q = 0; q < S2_PER_THREAD; ++q)
                if (!(COMMET_ABLATE & 256)) atomicAdd(&cnt[key[q] >> TILE_BITS], 1u);
        } else {
#pragma unroll
            for (uint32_t q = 0; q < S2_PER_THREAD; ++q) {
                const uint32_t i = threadIdx.x + NT * q;
                key[q] = i < n ? in[pos + i] : 0xFFFFFFFFu;
            }
#pragma unroll
            for (uint32_t q = 0; q < S2_PER_THREAD; ++q) {
                const uint32_t i = threadIdx.x + NT * q;
                if (i < n && !(COMMET_ABLATE & 256)) atomicAdd(&cnt[key[q] >> TILE_BITS], 1u);
            }
        }
        __syncthreads();
        lds_scan<NT>(cnt, base, nsub, wsum);
        for (uint32_t i = threadIdx.x; i < nsub; i += NT) {
            const uint32_t c = cnt[i];
            const unsigned long long want = g.packed ? (c + 2) / 3 : c;   // packed: whole groups of three keys
            gbase[i] = (c && !(COMMET_ABLATE & 128)) ? atomicAdd(&cursor2[((uint64_t) c1 << g.b2) + i], want) : 0ull;
        }
        if (whole) {
#pragma unroll
            for (uint32_t q = 0; q < S2_PER_THREAD; ++q)
                if (!(COMMET_ABLATE & 32)) {
                    const uint32_t sb = key[q] >> TILE_BITS;
                    sorted[base[sb] + atomicAdd(&fill[sb], 1u)] = key[q] & TILE_MASK;
                }
        } else {
#pragma unroll
            for (uint32_t q = 0; q < S2_PER_THREAD; ++q) {
                const uint32_t i = threadIdx.x + NT * q;
                if (i < n && !(COMMET_ABLATE & 32)) {
                    const uint32_t sb = key[q] >> TILE_BITS;
                    sorted[base[sb] + atomicAdd(&fill[sb], 1u)] = key[q] & TILE_MASK;
                }
            }
        }
        __syncthreads();
        if (COMMET_ABLATE & 64) {
        } else if (g.packed) {
            for (uint32_t sb = wave * 2 + (lane >> 5); sb < nsub; sb += (NT / 64) * 2)
                write_run_p3((uint2 *) out, gbase[sb], sorted, base[sb], cnt[sb], lane & 31u, 32u);
        } else {
            for (uint32_t sb = wave * 4 + (lane >> 4); sb < nsub; sb += (NT / 64) * 4)
                write_run(out, gbase[sb], sorted, base[sb], cnt[sb], lane & 15u, 16u);
        }
        __syncthreads();
        pos = seg_end;
        if (pos >= c_end) ++c1;
    }
}

// ---------------------------------------------------------------------------
// scatter2, packed geometry (final buckets = groups of three 19-bit keys in 8 bytes), 2^b2 <= S2P_MAX_SUB.
//
// Count, scan the runs' GROUP counts, place — a run's keys start at slot 3 * (its first group), so the slab is one flat
// sequence of groups; gid[] = the run of a group, noted by the thread that places the group's first key — pad every
// run's last group with its last key (setting a bit twice is harmless), then thread f packs and stores groups f,
// f + NT, ... whatever run they belong to.  (Walking the runs one by one, 32 lanes per run of which ~22 have a group,
// spent more instructions on the runs' bookkeeping than on their keys.)
// A slab costs its workgroup a chain of dependent steps (keys in, count, scan, cursors, place, out: ~10 us with three
// workgroups per CU), and the kernel's time is that chain, not a throughput — measured on configs[1], per step:
// 4.84 ms as runs, 4.15 ms like this; fixed bins per final bucket (one atomic and one store per key, no counting pass)
// halve the LDS work but expose the cursors' round trip: 4.7 ms; persistent workgroups that fetch the next slab's keys
// while this one is written out: 6.1-6.7 ms (static shares lose more than the prefetch hides).  So: as few barriers
// as possible, the bucket cursors reserved (one atomic per run, whole groups, as part_scan_kernel's bound on the groups
// of a bucket assumes) while the keys are placed, and the dispatcher balancing one slab per workgroup.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(S2_NT, S2P_WAVES) void part_scatter2_packed_kernel(const uint32_t *__restrict__ in, uint2 *__restrict__ out,
                                                                     const uint64_t *__restrict__ off, PartGeom g,
                                                                     unsigned long long *__restrict__ cursor2, uint64_t total)
{
    constexpr int NT = S2_NT;
    __shared__ uint32_t sorted[S2_KEYS + 2 * S2P_MAX_SUB];
    __shared__ uint32_t cnt[S2P_MAX_SUB], fill[S2P_MAX_SUB];
    __shared__ unsigned long long gbase[S2P_MAX_SUB];
    __shared__ uint32_t wsum[16];
    __shared__ uint8_t gid[S2_KEYS / 3 + S2P_MAX_SUB + 4];
    const uint32_t nsub = 1u << g.b2;   // <= S2P_MAX_SUB < NT
    // Slab order (option "s2_swizzle" = G, 0 = dispatch order): see part_scatter2_kernel
    const uint32_t G = (uint32_t) g.xcd_swizzle;
    uint64_t slab = blockIdx.x;
    if (G > 1) {
        const uint32_t q = gridDim.x / G, rem = gridDim.x % G, x = blockIdx.x % G;
        slab = (uint64_t) x * q + min(x, rem) + blockIdx.x / G;
    }
    const uint64_t s0 = slab * S2_KEYS;
    if (s0 >= total) return;
    const uint64_t s1 = min(total, s0 + S2_KEYS);
    uint32_t lo = 0, hi = g.nb1;   // coarse bucket containing s0: largest c with off[c << b2] <= s0
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[(uint64_t) mid << g.b2] <= s0) lo = mid;
        else hi = mid;
    }
    uint32_t c1 = lo;
    uint64_t pos = s0;
    while (pos < s1) {
        const uint64_t c_end = off[(uint64_t) (c1 + 1) << g.b2];
        const uint64_t seg_end = min(s1, c_end);
        if (seg_end <= pos) {   // empty coarse bucket
            ++c1;
            continue;
        }
        const uint32_t n = (uint32_t) (seg_end - pos);
        const bool whole = n == S2_KEYS;   // (uniform) a whole slab inside one coarse bucket, the usual case: no per-key bounds tests
        uint32_t key[S2_PER_THREAD];
        if (whole) {   // (pos = slab * S2_KEYS here: 16-byte aligned; which thread sorts which key does not matter)
            const uint4 *in4 = (const uint4 *) (in + pos);
#pragma unroll
            for (uint32_t q = 0; q < S2_PER_THREAD / 4; ++q) {
                const uint4 v = in4[threadIdx.x + NT * q];
                key[4 * q] = v.x, key[4 * q + 1] = v.y, key[4 * q + 2] = v.z, key[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (uint32_t q = 0; q < S2_PER_THREAD; ++q) {
                const uint32_t i = threadIdx.x + NT * q;
                key[q] = i < n ? in[pos + i] : 0xFFFFFFFFu;
            }
        }
        if (threadIdx.x < nsub) cnt[threadIdx.x] = 0;
        // Hot final buckets (poly-A reads, tandem repeats: every k-mer of a read makes the same key, so its keys lie in a
        // row in the coarse bucket): LDS atomics of one wave instruction on ONE counter run one lane after the other
        // (0.5 per clock and CU against 5-7 on spread addresses, commet_ldsbench with a 1-word table), and the slabs of a
        // hot coarse bucket are on few CUs at any time — 12.1 instead of 3.9 ms per step with 3 % of the keys poly-A.
        // A slab whose first four keys of some thread all go to one final bucket takes the aggregated path below: the
        // lanes of a wave whose four keys share a bucket elect one lane per bucket value that adds for all of them.
        // i.i.d. keys never qualify (2^-21 per thread), so the usual path pays three compares per slab and thread.
        bool hot = false;
        if (whole) {
            const uint32_t a0 = key[0] >> TILE_BITS;
            hot = (a0 == (key[1] >> TILE_BITS)) & (a0 == (key[2] >> TILE_BITS)) & (a0 == (key[3] >> TILE_BITS));
        }
        hot = __syncthreads_or(hot);
        const uint32_t lane = threadIdx.x & 63u;
        if (COMMET_ABLATE & 256) {
        } else if (hot) {
#pragma unroll
            for (uint32_t q = 0; q < S2_PER_THREAD; q += 4) {
                const uint32_t b0 = key[q] >> TILE_BITS, b1 = key[q + 1] >> TILE_BITS, b2 = key[q + 2] >> TILE_BITS, b3 = key[q + 3] >> TILE_BITS;
                const bool run = (b0 == b1) & (b0 == b2) & (b0 == b3);
                for (uint64_t rem = __ballot(run); rem;) {          // (uniform over the wave)
                    const int l = __ffsll((long long) rem) - 1;
                    const uint32_t v = (uint32_t) __shfl((int) b0, l, 64);
                    const uint64_t same = __ballot(run && b0 == v);
                    if ((int) lane == l) atomicAdd(&cnt[v], 4u * (uint32_t) __popcll(same));
                    rem &= ~same;
                }
                if (!run) atomicAdd(&cnt[b0], 1u), atomicAdd(&cnt[b1], 1u), atomicAdd(&cnt[b2], 1u), atomicAdd(&cnt[b3], 1u);
            }
        } else if (whole) {
#pragma unroll
            for (uint32_t q = 0; q < S2_PER_THREAD; ++q) atomicAdd(&cnt[key[q] >> TILE_BITS], 1u);
        } else {
#pragma unroll
            for (uint32_t q = 0; q < S2_PER_THREAD; ++q)
                if (threadIdx.x + NT * q < n) atomicAdd(&cnt[key[q] >> TILE_BITS], 1u);
        }
        __syncthreads();
        // ex = first GROUP of the run (exclusive scan of the runs' group counts); its keys start at slot 3 * ex, which is
        // where the run's fill counter starts: the placement's atomic returns the key's slot
        uint32_t n_groups = 0;
        const uint32_t c = threadIdx.x < nsub ? cnt[threadIdx.x] : 0u;
        const uint32_t ex = block_scan<NT>((c + 2) / 3, wsum, &n_groups);
        // the run's place in its final bucket: reserved now, used after the placement (the atomic's round trip runs beside it)
        unsigned long long at = 0;
        if (threadIdx.x < nsub) {
            fill[threadIdx.x] = 3u * ex;
            if (c && !(COMMET_ABLATE & 128)) at = atomicAdd(&cursor2[((uint64_t) c1 << g.b2) + threadIdx.x], (unsigned long long) ((c + 2) / 3));
        }
        __syncthreads();
        auto place = [&](uint32_t kq) {
            const uint32_t sb = kq >> TILE_BITS;
            const uint32_t slot = atomicAdd(&fill[sb], 1u), grp = slot / 3u;
            sorted[slot] = kq & TILE_MASK;
            if (slot == 3u * grp) gid[grp] = (uint8_t) sb;
        };
        if (COMMET_ABLATE & 32) {
        } else if (hot) {
#pragma unroll
            for (uint32_t q = 0; q < S2_PER_THREAD; q += 4) {
                const uint32_t b0 = key[q] >> TILE_BITS, b1 = key[q + 1] >> TILE_BITS, b2 = key[q + 2] >> TILE_BITS, b3 = key[q + 3] >> TILE_BITS;
                const bool run = (b0 == b1) & (b0 == b2) & (b0 == b3);
                for (uint64_t rem = __ballot(run); rem;) {
                    const int l = __ffsll((long long) rem) - 1;
                    const uint32_t v = (uint32_t) __shfl((int) b0, l, 64);
                    const uint64_t same = __ballot(run && b0 == v);
                    uint32_t first = 0;
                    if ((int) lane == l) first = atomicAdd(&fill[v], 4u * (uint32_t) __popcll(same));
                    first = (uint32_t) __shfl((int) first, l, 64);
                    if (run && b0 == v) {                            // this lane's four slots, in lane order
                        const uint32_t slot0 = first + 4u * (uint32_t) __popcll(same & ((1ull << lane) - 1ull));
#pragma unroll
                        for (uint32_t u = 0; u < 4; ++u) {
                            const uint32_t slot = slot0 + u;
                            sorted[slot] = key[q + u] & TILE_MASK;
                            if (slot % 3u == 0u) gid[slot / 3u] = (uint8_t) v;
                        }
                    }
                    rem &= ~same;
                }
                if (!run) place(key[q]), place(key[q + 1]), place(key[q + 2]), place(key[q + 3]);
            }
        } else if (whole) {
#pragma unroll
            for (uint32_t q = 0; q < S2_PER_THREAD; ++q) place(key[q]);
        } else {
#pragma unroll
            for (uint32_t q = 0; q < S2_PER_THREAD; ++q)
                if (threadIdx.x + NT * q < n) place(key[q]);
        }
        __syncthreads();
        if (threadIdx.x < nsub) {
            gbase[threadIdx.x] = at - ex;   // where group 0 of the SLAB would go if the run's groups were numbered like the slab's
            const uint32_t end = 3u * ex + c;
            if (c % 3u) {
                const uint32_t last = sorted[end - 1];
                sorted[end] = last;
                if (c % 3u == 1u) sorted[end + 1] = last;
            }
        }
        __syncthreads();
        for (uint32_t f = threadIdx.x; f < n_groups && !(COMMET_ABLATE & 64); f += NT) {
            const uint32_t k0 = sorted[3u * f], k1 = sorted[3u * f + 1], k2 = sorted[3u * f + 2];
            out[gbase[gid[f]] + f] = make_uint2(k0 | (k1 << 19), (k1 >> 13) | (k2 << 6));
        }
        __syncthreads();
        pos = seg_end;
        if (pos >= c_end) ++c1;
    }
}

// zeroes the tiles of buckets that are split over several build workgroups (they merge with atomic ORs)
__global__ __launch_bounds__(256) void part_zero_split_kernel(const uint32_t *__restrict__ wl_off, PartGeom g,
                                                              uint32_t *__restrict__ filter)
{
    const uint32_t b = blockIdx.x;
    if (b >= g.nb || wl_off[b + 1] - wl_off[b] <= 1) return;
    uint4 *d4 = (uint4 *) (filter + (uint64_t) b * TILE_WORDS);
    for (uint32_t i = threadIdx.x; i < TILE_WORDS / 4; i += 256) d4[i] = make_uint4(0, 0, 0, 0);
}

// ---------------------------------------------------------------------------
// build: (tile, split) work items -> LDS tile -> filter
// ---------------------------------------------------------------------------
// packed geometry: off = goff (group offsets), gend = the final scatter2 cursors: bucket b's groups are
// [off[b], gend[b]); its n_split workgroups take equal shares of them
__global__ __launch_bounds__(BUILD_NT) void part_build_kernel(const uint32_t *__restrict__ keys,
                                                         const uint64_t *__restrict__ off,
                                                         const uint32_t *__restrict__ wl_off, PartGeom g,
                                                         uint32_t *__restrict__ filter, int additive,
                                                         const unsigned long long *__restrict__ gend)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t tile[];   // TILE_WORDS
    const uint32_t total_items = wl_off[g.nb];
    if (blockIdx.x >= total_items) return;
    // bucket with wl_off[b] <= blockIdx.x < wl_off[b+1]
    uint32_t lo = 0, hi = g.nb;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (wl_off[mid] <= blockIdx.x) lo = mid;
        else hi = mid;
    }
    const uint32_t b = lo;
    const uint32_t split = blockIdx.x - wl_off[b];
    const uint32_t n_split = wl_off[b + 1] - wl_off[b];
    const uint64_t k0 = off[b] + (uint64_t) split * BUILD_CAP;
    const uint64_t k1 = min(off[b + 1], k0 + BUILD_CAP);
    uint4 *t4 = (uint4 *) tile;
    for (uint32_t i = threadIdx.x; i < TILE_WORDS / 4; i += BUILD_NT) t4[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (g.packed) {
        const uint64_t g0 = off[b], gn = gend[b] - g0;            // groups of the bucket
        const uint64_t share = (gn + n_split - 1) / n_split;
        const uint64_t v0 = min(gn, (uint64_t) split * share), v1 = min(gn, v0 + share);
        const uint2 *gv = (const uint2 *) keys + g0;
        auto put = [&](uint2 x) {
            const uint32_t a = x.x & TILE_MASK, bq = ((x.x >> 19) | (x.y << 13)) & TILE_MASK, cq = (x.y >> 6) & TILE_MASK;
            atomicOr(&tile[a >> 5], 1u << (a & 31u));
            atomicOr(&tile[bq >> 5], 1u << (bq & 31u));
            atomicOr(&tile[cq >> 5], 1u << (cq & 31u));
        };
        uint64_t v = v0 + threadIdx.x;
        for (; v + 3 * BUILD_NT < v1; v += 4 * BUILD_NT) {
            const uint2 x0 = gv[v], x1 = gv[v + BUILD_NT], x2 = gv[v + 2 * BUILD_NT], x3 = gv[v + 3 * BUILD_NT];
            put(x0), put(x1), put(x2), put(x3);
        }
        for (; v < v1; v += BUILD_NT) put(gv[v]);
    } else {
        // head up to 16-byte alignment, then 4 keys per lane per load, 4 loads in flight
        uint64_t a0 = (k0 + 3) & ~3ull;
        if (a0 > k1) a0 = k1;
        for (uint64_t i = k0 + threadIdx.x; i < a0; i += BUILD_NT) {
            const uint32_t key = keys[i];
            atomicOr(&tile[key >> 5], 1u << (key & 31u));
        }
        const uint64_t nvec = (k1 - a0) >> 2;
        const uint4 *kv = (const uint4 *) (keys + a0);
        uint64_t v = threadIdx.x;
        for (; v + 3 * BUILD_NT < nvec; v += 4 * BUILD_NT) {
            const uint4 x0 = kv[v], x1 = kv[v + BUILD_NT], x2 = kv[v + 2 * BUILD_NT], x3 = kv[v + 3 * BUILD_NT];
            const uint4 xs[4] = {x0, x1, x2, x3};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                atomicOr(&tile[xs[u].x >> 5], 1u << (xs[u].x & 31u));
                atomicOr(&tile[xs[u].y >> 5], 1u << (xs[u].y & 31u));
                atomicOr(&tile[xs[u].z >> 5], 1u << (xs[u].z & 31u));
                atomicOr(&tile[xs[u].w >> 5], 1u << (xs[u].w & 31u));
            }
        }
        for (; v < nvec; v += BUILD_NT) {
            const uint4 x = kv[v];
            atomicOr(&tile[x.x >> 5], 1u << (x.x & 31u));
            atomicOr(&tile[x.y >> 5], 1u << (x.y & 31u));
            atomicOr(&tile[x.z >> 5], 1u << (x.z & 31u));
            atomicOr(&tile[x.w >> 5], 1u << (x.w & 31u));
        }
        for (uint64_t i = a0 + 4 * nvec + threadIdx.x; i < k1; i += BUILD_NT) {
            const uint32_t key = keys[i];
            atomicOr(&tile[key >> 5], 1u << (key & 31u));
        }
    }
    __syncthreads();
    // bucket b = (plane << plane_shift) | tile index: its words sit at b * TILE_WORDS of the 4-plane array
    uint32_t *dst = filter + (uint64_t) b * TILE_WORDS;
    if (n_split == 1) {
        uint4 *d4 = (uint4 *) dst;
        if (additive) {   // the filter already holds bits (commet_index_reads called again without a reset)
            for (uint32_t i = threadIdx.x; i < TILE_WORDS / 4; i += BUILD_NT) {
                uint4 o = d4[i];
                const uint4 n = t4[i];
                o.x |= n.x, o.y |= n.y, o.z |= n.z, o.w |= n.w;
                d4[i] = o;
            }
        } else {
            for (uint32_t i = threadIdx.x; i < TILE_WORDS / 4; i += BUILD_NT) d4[i] = t4[i];
        }
    } else {
        for (uint32_t i = threadIdx.x; i < TILE_WORDS; i += BUILD_NT) {
            const uint32_t v = tile[i];
            if (v) (void) __hip_atomic_fetch_or(dst + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace commet

// kernels.hpp — CDNA4 (gfx950) device code of the index_and_search hot path.
//
// Data layout in HBM:
//   reads   : per read, ceil(len/32) word-triples {hi, lo, valid}; bit j of a
//             word = base 32*w+j.  hi = 1 for G/T, lo = 1 for C/T, valid = 1
//             for ACGTacgt.  Read r starts at triple index (goff[r] >> 5) + r,
//             goff = cumulative base offsets (n+1 entries).
//   filter  : four bit-planes A,B,C,D of 2^k bits each (plane p at words
//             [p*plane_words, (p+1)*plane_words) of a filter slot).  Bit `key`
//             of planes B,C,D = lane b,c,d of the reference filter at that
//             key; plane A holds lane a of `key` at bit psi_a(key) (strand-
//             paired layout, below).  Up to 4 slots hold the filters of a
//             group of chunks; their A planes are also kept word-interleaved.
//   bitmaps : 64 reads per uint64 word, LSB-first (== BooleanVector bytes).
//
// Hash structure used everywhere below (hash_key.h:63-123):
//   with the window W_x of plane x held LSB = oldest base of the k-mer,
//     forward  keya = bitreverse_k(W_hi), keyb = bitreverse_k(W_lo)
//     reverse  keya = ~W_hi & mask,       keyb = ~W_lo & mask
//     keyc = keya ^ keyb, keyd = keya | keyb   (both strands)
//   so one left-to-right rolling window serves both strands.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace commet {

// Scalar-register budget of the latency-bound kernels.  A CU admits floor(800 / (ceil(sgpr / 16) * 16 + 16)) workgroups of 256
// threads (MI355X_MICROARCH.md, "Residency and cooperative launch": <= 80 SGPRs: 8 workgroups, 82-96: 7, 98-112: 6), whatever
// the LDS and the vector registers would allow; left alone the compiler takes 92-106 for the search kernels (kernel arguments
// and loop-invariant addresses), i.e. 6-7 workgroups.  These kernels are chains of dependent memory round trips and run on
// workgroups in flight: tq_replay_kernel 5.91 -> 5.20 ms per configs[1] step with the cap alone (tools/kernel_resources.py lists
// every kernel's registers and what a CU admits).  What does not fit is kept in lanes of a vector register (v_writelane).
#ifndef COMMET_SGPR_CAP
#define COMMET_SGPR_CAP 80
#endif
#if COMMET_SGPR_CAP
#define COMMET_SGPRS __attribute__((amdgpu_num_sgpr(COMMET_SGPR_CAP)))
#else
#define COMMET_SGPRS
#endif

struct ReadsView {
    const uint32_t *planes;      // word triples
    const uint64_t *goff;        // n+1 cumulative base offsets (unused when uniform_len != 0)
    uint32_t        uniform_len; // != 0: every read has this length, goff[r] = r * uniform_len
    uint64_t        n;
};

struct FilterView {
    uint32_t *a, *b, *c, *d;     // bit-planes, 2^k bits each
};

template <typename W> struct KeyTraits;
template <> struct KeyTraits<uint32_t> {
    static constexpr int BITS = 32;
    __device__ static __forceinline__ uint32_t brev(uint32_t x) { return __brev(x); }
};
template <> struct KeyTraits<uint64_t> {
    static constexpr int BITS = 64;
    __device__ static __forceinline__ uint64_t brev(uint64_t x) { return __brevll(x); }
};

__device__ __forceinline__ void read_extent(const ReadsView &rv, uint64_t r, uint64_t &triple0, uint32_t &len)
{
    uint64_t o;
    if (rv.uniform_len) {
        o = r * (uint64_t) rv.uniform_len;
        len = rv.uniform_len;
    } else {
        o = rv.goff[r];
        len = (uint32_t) (rv.goff[r + 1] - o);
    }
    triple0 = (o >> 5) + r;
}

template <typename W>
__device__ __forceinline__ uint32_t test_bit(const uint32_t *plane, W key)
{
    return (plane[key >> 5] >> ((uint32_t) key & 31u)) & 1u;
}

template <typename W>
__device__ __forceinline__ void set_bit(uint32_t *plane, W key)
{
    // result unused -> global_atomic_or without return, executed at L2
    (void) __hip_atomic_fetch_or(plane + (key >> 5), 1u << ((uint32_t) key & 31u), __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
}

// The (scanned, found) counters of a search launch: every wave adding its two popcounts to the same global words
// serialises at L2 (milliseconds on a 10 M-read set), so the waves of a workgroup add into LDS first and the workgroup
// adds once.  Every thread of the workgroup must call this (it holds barriers); wg_cnt: 2 * n_chunks words of LDS.
__device__ __forceinline__ void add_chunk_counters(unsigned long long *__restrict__ counters, uint32_t cstride, int n_chunks,
                                                   bool active, int found_chunk, unsigned int *wg_cnt)
{
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 2u * (unsigned) n_chunks) wg_cnt[threadIdx.x] = 0;
    __syncthreads();
    // chunk i: scanned = active reads not found in an earlier chunk of the group, found = found in chunk i
    for (int i = 0; i < n_chunks; ++i) {
        const uint64_t sc = __ballot(active && (found_chunk < 0 || found_chunk >= i));
        const uint64_t fd = __ballot(found_chunk == i);
        if (lane == 0) {
            if (sc) atomicAdd(&wg_cnt[2 * i], (unsigned int) __popcll(sc));
            if (fd) atomicAdd(&wg_cnt[2 * i + 1], (unsigned int) __popcll(fd));
        }
    }
    __syncthreads();
    if (threadIdx.x < 2u * (unsigned) n_chunks && wg_cnt[threadIdx.x])
        atomicAdd(&counters[(uint64_t) (threadIdx.x >> 1) * cstride + (threadIdx.x & 1u)], (unsigned long long) wg_cnt[threadIdx.x]);
}

// ---------------------------------------------------------------------------
// Which read a search thread works on.  Usually thread i of the launch takes read i and the bitmaps decide (sel: reads to
// search, tags: reads found by an earlier chunk).  A pass over FEW of a set's reads — Commet.py's third job searches a set
// restricted to the first job's result, ~22 % of its reads (Commet.py:233) — gets the numbers of those reads as a list instead
// (ActiveList: sel & ~tags, in order; sel_ids_kernel): thread i takes read ids[i], every lane of a wave has work, and the
// pass costs what its reads cost, not what the set's do (a wave of the bitmap form waits for its busiest lane through all of
// a read's dependent round trips whether 14 or 64 of its lanes are active: 46 ms against 12 for 11 of 50 M reads).
// ---------------------------------------------------------------------------
struct ActiveList {
    const uint32_t *ids;   // nullptr: bitmap form
    const uint32_t *n;     // number of listed reads (device memory: the launch is sized by a bound the host knows)
};

struct SearchLane {
    uint64_t r, word, tagw;
    bool     active, in_range;
};

__device__ __forceinline__ SearchLane search_lane(const ReadsView &rv, const ActiveList &al, const uint64_t *__restrict__ sel,
                                                  const uint64_t *__restrict__ tags)
{
    SearchLane me;
    const uint64_t i = blockIdx.x * 256ull + threadIdx.x;
    if (al.ids) {
        me.active = i < (uint64_t) *al.n;
        me.r = me.active ? (uint64_t) al.ids[i] : rv.n;      // (rv.n: no read)
        me.word = me.r >> 6, me.tagw = 0, me.in_range = false;
        return me;
    }
    me.r = i;
    me.word = i >> 6;
    const int lane = threadIdx.x & 63;
    me.in_range = (me.word << 6) < rv.n;
    uint64_t selw = ~0ull;
    me.tagw = 0;
    if (me.in_range) {
        if (sel) selw = sel[me.word];
        if (tags) me.tagw = tags[me.word];
    }
    me.active = (i < rv.n) && ((selw >> lane) & 1ull) && !((me.tagw >> lane) & 1ull);
    return me;
}

// the found flags of the launch: one ballot word per 64 consecutive reads (bitmap form) or one atomic OR per found read (list form)
__device__ __forceinline__ void publish_found(const ActiveList &al, const SearchLane &me, bool found, uint64_t *__restrict__ tags)
{
    if (al.ids) {
        if (found && tags) (void) __hip_atomic_fetch_or(tags + me.word, 1ull << (me.r & 63ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const uint64_t fb = __ballot(found);
    if ((threadIdx.x & 63) == 0 && me.in_range && tags) tags[me.word] = me.tagw | fb;
}

// ---------------------------------------------------------------------------
// Strand-paired layout of plane A.
// The forward key of a window W is kf = bitreverse_k(W) and its reverse-complement
// key is kr = ~W & mask, so kr = T(kf) with the involution T(x) = ~bitreverse_k(x).
// A non-shared read probes plane A at BOTH kf and kr for every window (forward
// pass, then reverse pass): the filter layout is free, so plane A stores
// bit `key` at address psi(key), a bijection of [0, 2^k) with
//                     psi(T(key)) = psi(key) ^ 1
// (same 32-bit word, neighbouring bit).  One 4-byte load of the forward pass then
// returns the reverse strand's lane-a bit as well, and the reverse pass needs no
// plane-A traffic: ~half of the L2-missing requests of the search kernel go away.
//
// Construction (h = k/2): key = [H | m | L] with h-bit halves (m = middle bit, odd
// k only).  T maps (H, m, L) to (g(L), ~m, g(H)), g(x) = ~bitreverse_h(x).  With
// u = H, v = g(L), T is the swap (u, v) -> (v, u) (and m -> ~m).  s = u ^ v is
// invariant.  For s != 0 let b = lowest set bit of s and pi_s the GF(2)-linear
// bijection  y = u ^ (u_b ? s & ~(1<<b) : 0), then swap bits 0 and b of y:
// pi_s(u ^ s) = pi_s(u) ^ 1.  Address = [s | m ^ (pi&1) | pi]  (T flips bit 0 only).
// For s == 0 (u == v): even k: T(key) = key, address = [0 | u]; odd k: T flips m,
// address = [0 | u | m].
// ---------------------------------------------------------------------------
template <typename W>
__device__ __forceinline__ W psi_a(W key, int k, bool &self_paired)
{
    // branch-free: for s == 0 the general formula with b = 0 gives y = u, i.e. [0 | u] (even k)
    const int h = k >> 1;
    const bool odd = k & 1;
    const uint32_t hmask = (1u << h) - 1u;                     // h <= 19
    const uint32_t L = (uint32_t) key & hmask;
    const uint32_t m = odd ? (uint32_t) (key >> h) & 1u : 0u;
    const uint32_t u = (uint32_t) (key >> (h + (odd ? 1 : 0))) & hmask;
    const uint32_t v = (~(__brev(L) >> (32 - h))) & hmask;     // g(L); h >= 1 for k >= 2
    const uint32_t s = u ^ v;
    self_paired = (s == 0) && !odd;
    const uint32_t b = s ? (uint32_t) __ffs((int) s) - 1u : 0u;
    uint32_t y = u ^ ((0u - ((u >> b) & 1u)) & s & ~(1u << b));
    const uint32_t d = ((y >> b) ^ y) & 1u;                    // swap bits 0 and b
    y ^= d | (d << b);
    if (odd) {
        const uint32_t m2 = m ^ (y & 1u);
        const W nz = ((W) s << (h + 1)) | ((W) m2 << h) | (W) y;
        const W z = (W) ((u << 1) | m);
        return s ? nz : z;
    }
    return ((W) s << h) | (W) y;
}

// psi_a(key) with the bits below its s field cleared: s fills the address from bit h (+1 for odd k) upwards and
// h (+1) <= TILE_BITS for every k the bucketed construction takes, so  psi_a_top(key) >> TILE_BITS == psi_a(key) >> TILE_BITS
// — all a bucket count needs, at a quarter of the arithmetic.
template <typename W>
__device__ __forceinline__ W psi_a_top(W key, int k)
{
    const int h = k >> 1, odd = k & 1;
    const uint32_t hmask = (1u << h) - 1u;
    const uint32_t L = (uint32_t) key & hmask;
    const uint32_t u = (uint32_t) (key >> (h + odd)) & hmask;
    const uint32_t v = (~(__brev(L) >> (32 - h))) & hmask;
    return (W) (u ^ v) << (h + odd);
}

template <typename W>
__device__ __forceinline__ W psi_a(W key, int k)
{
    bool sp;
    return psi_a<W>(key, k, sp);
}

// ---------------------------------------------------------------------------
// direct window extraction: the k-mer ending at base p = 32*w + j of a read,
// from the word triples w-2, w-1, w (no serial rolling: any lane, any position)
// ---------------------------------------------------------------------------
template <typename W> struct ItemWords;
template <> struct ItemWords<uint32_t> {
    uint32_t hi[2], lo[2], va[2];   // [0] = word w-1, [1] = word w
    __device__ __forceinline__ void load(const uint32_t *p, uint32_t w)
    {
        hi[1] = p[3 * w], lo[1] = p[3 * w + 1], va[1] = p[3 * w + 2];
        if (w) hi[0] = p[3 * w - 3], lo[0] = p[3 * w - 2], va[0] = p[3 * w - 1];
        else hi[0] = lo[0] = va[0] = 0;
    }
    // the same through get(i) = word i of the read's triples (a copy of the read kept in LDS, see ReadWords)
    template <typename G> __device__ __forceinline__ void load_with(G &&get, uint32_t w)
    {
        hi[1] = get(3 * w), lo[1] = get(3 * w + 1), va[1] = get(3 * w + 2);
        if (w) hi[0] = get(3 * w - 3), lo[0] = get(3 * w - 2), va[0] = get(3 * w - 1);
        else hi[0] = lo[0] = va[0] = 0;
    }
    // window of the k bases ending at bit j of word w; false if any is not ACGT
    __device__ __forceinline__ bool window(uint32_t j, int k, uint32_t mask, uint32_t &wh, uint32_t &wl) const
    {
        const uint32_t s = 33u + j - (uint32_t) k;   // 1..32
        const uint32_t v = (uint32_t) ((((uint64_t) va[1] << 32) | va[0]) >> s) & mask;
        wh = (uint32_t) ((((uint64_t) hi[1] << 32) | hi[0]) >> s) & mask;
        wl = (uint32_t) ((((uint64_t) lo[1] << 32) | lo[0]) >> s) & mask;
        return v == mask;
    }
};
template <> struct ItemWords<uint64_t> {
    uint32_t hi[3], lo[3], va[3];   // words w-2, w-1, w
    __device__ __forceinline__ void load(const uint32_t *p, uint32_t w)
    {
        hi[2] = p[3 * w], lo[2] = p[3 * w + 1], va[2] = p[3 * w + 2];
        if (w) hi[1] = p[3 * w - 3], lo[1] = p[3 * w - 2], va[1] = p[3 * w - 1];
        else hi[1] = lo[1] = va[1] = 0;
        if (w > 1) hi[0] = p[3 * w - 6], lo[0] = p[3 * w - 5], va[0] = p[3 * w - 4];
        else hi[0] = lo[0] = va[0] = 0;
    }
    template <typename G> __device__ __forceinline__ void load_with(G &&get, uint32_t w)
    {
        hi[2] = get(3 * w), lo[2] = get(3 * w + 1), va[2] = get(3 * w + 2);
        if (w) hi[1] = get(3 * w - 3), lo[1] = get(3 * w - 2), va[1] = get(3 * w - 1);
        else hi[1] = lo[1] = va[1] = 0;
        if (w > 1) hi[0] = get(3 * w - 6), lo[0] = get(3 * w - 5), va[0] = get(3 * w - 4);
        else hi[0] = lo[0] = va[0] = 0;
    }
    __device__ __forceinline__ static uint64_t ext(const uint32_t *x, uint32_t s, uint64_t mask)
    {
        const uint64_t lo64 = ((uint64_t) x[1] << 32) | x[0];
        return ((lo64 >> s) | ((uint64_t) x[2] << (64 - s))) & mask;   // 27 <= s <= 63 for 33 <= k <= 38
    }
    __device__ __forceinline__ bool window(uint32_t j, int k, uint64_t mask, uint64_t &wh, uint64_t &wl) const
    {
        const uint32_t s = 65u + j - (uint32_t) k;
        wh = ext(hi, s, mask);
        wl = ext(lo, s, mask);
        return ext(va, s, mask) == mask;
    }
};

// ---------------------------------------------------------------------------
// pack: ASCII -> {hi, lo, valid} planes, per-read complete-k-mer counts.
// One lane per read.  Replaces the per-char work of Alphabet::is_in
// (alphabet.h:44-58) and HashKey::add's base classes (hash_key.h:72-88).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_reads_kernel(const uint8_t *__restrict__ bases,
                                                         const uint64_t *__restrict__ offs, uint64_t n_batch,
                                                         uint64_t read0, uint64_t base0,
                                                         uint32_t *__restrict__ planes, uint64_t *__restrict__ goff,
                                                         uint32_t *__restrict__ kcnt, uint32_t *__restrict__ len_minmax,
                                                         int k)
{
    const uint64_t i = blockIdx.x * 256ull + threadIdx.x;
    if (i > n_batch) return;
    const uint64_t o = offs[i];
    goff[read0 + i] = base0 + o;             // entry n_batch closes the batch
    if (i == n_batch) return;
    const uint32_t len = (uint32_t) (offs[i + 1] - o);
    const uint8_t *s = bases + o;
    uint32_t *dst = planes + 3 * (((base0 + o) >> 5) + read0 + i);
    uint32_t run = 0, cnt = 0;
    for (uint32_t w = 0; w * 32u < len; ++w) {
        uint32_t hi = 0, lo = 0, va = 0;
        const uint32_t nb = min(32u, len - w * 32u);
        for (uint32_t j = 0; j < nb; ++j) {
            const uint32_t ch = s[w * 32u + j];
            const uint32_t u = ch & 0xDFu;   // fold case
            const uint32_t v = (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T');
            const uint32_t h = (ch >> 2) & 1u;                 // G/T
            const uint32_t l = ((ch >> 1) ^ (ch >> 2)) & 1u;   // C/T
            hi |= (v & h) << j;
            lo |= (v & l) << j;
            va |= v << j;
            run = v ? run + 1 : 0;
            cnt += (run >= (uint32_t) k);
        }
        dst[3 * w + 0] = hi;
        dst[3 * w + 1] = lo;
        dst[3 * w + 2] = va;
    }
    kcnt[read0 + i] = cnt;
    atomicMin(&len_minmax[0], len);
    atomicMax(&len_minmax[1], len);
    atomicMax(&len_minmax[2], cnt);   // largest per-read k-mer count of the set
}

// complete k-mers per read from the validity plane alone: a packed read set is independent of k, the counts are not
// (commet_readset_load); also the set's largest count (len_minmax[2], as pack_reads_kernel leaves it)
__global__ __launch_bounds__(256) void kmer_counts_kernel(ReadsView rv, int k, uint32_t *__restrict__ kcnt,
                                                          uint32_t *__restrict__ len_minmax)
{
    const uint64_t r = blockIdx.x * 256ull + threadIdx.x;
    if (r >= rv.n) return;
    uint64_t t0;
    uint32_t len;
    read_extent(rv, r, t0, len);
    const uint32_t *p = rv.planes + 3 * t0;
    uint32_t run = 0, cnt = 0;
    for (uint32_t w = 0; w * 32u < len; ++w) {
        const uint32_t va = p[3 * w + 2];
        const uint32_t nb = min(32u, len - w * 32u);
        for (uint32_t j = 0; j < nb; ++j) {
            run = ((va >> j) & 1u) ? run + 1 : 0;
            cnt += (run >= (uint32_t) k);
        }
    }
    kcnt[r] = cnt;
    atomicMax(&len_minmax[2], cnt);
}

// *sum += over the reads of max(0, len - tk + 1): the first-hit windows of a ragged set (tk = t * k), an upper bound on the records of
// its query list (tile_search.hpp; windows that hold a non-ACGT base make no record)
__global__ __launch_bounds__(256) void first_hit_windows_kernel(ReadsView rv, uint32_t tk, unsigned long long *__restrict__ sum)
{
    __shared__ unsigned long long part[4];
    unsigned long long s = 0;
    for (uint64_t r = blockIdx.x * 256ull + threadIdx.x; r < rv.n; r += (uint64_t) gridDim.x * 256ull) {
        const uint64_t len = rv.goff[r + 1] - rv.goff[r];
        if (len >= tk) s += len - tk + 1;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0 && part[0] + part[1] + part[2] + part[3]) atomicAdd(sum, part[0] + part[1] + part[2] + part[3]);
}

// sums[b] = k-mers of the selected reads (bitmap sel, 64 reads per word) among reads [b * 4096, (b + 1) * 4096):
// what the host's selection planner walks instead of the reads (read_iter.hpp, plan_index_blocks)
constexpr uint32_t PLAN_BLOCK_READS = 4096;
__global__ __launch_bounds__(256) void block_kmer_sums_kernel(const uint32_t *__restrict__ kcnt, const uint64_t *__restrict__ sel,
                                                              uint64_t n, unsigned long long *__restrict__ sums)
{
    __shared__ unsigned long long part[4];
    const uint64_t r0 = (uint64_t) blockIdx.x * PLAN_BLOCK_READS;
    unsigned long long s = 0;
    for (uint32_t i = threadIdx.x; i < PLAN_BLOCK_READS; i += 256) {
        const uint64_t r = r0 + i;
        if (r < n && (!sel || ((sel[r >> 6] >> (r & 63)) & 1ull))) s += kcnt[r];   // sel == nullptr: every read
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// ---------------------------------------------------------------------------
// index: one lane per read, rolling forward keys, 4 atomic ORs per k-mer.
// Replaces index_reads.h:51-59 + BloomFilter::feed (bloom_filter.h:112-118).
// ---------------------------------------------------------------------------
template <typename W>
__global__ __launch_bounds__(256) void index_kernel(ReadsView rv, FilterView f, int k, uint64_t first, uint64_t count,
                                                    const uint64_t *__restrict__ sel,
                                                    unsigned long long *__restrict__ kmers_fed)
{
    using T = KeyTraits<W>;
    const uint64_t i = blockIdx.x * 256ull + threadIdx.x;
    const uint64_t r = first + i;
    bool act = i < count;
    if (act && sel) act = (sel[r >> 6] >> (r & 63)) & 1ull;
    uint32_t fed = 0;
    if (act) {
        uint64_t t0;
        uint32_t len;
        read_extent(rv, r, t0, len);
        const uint32_t *p = rv.planes + 3 * t0;
        W wh = 0, wl = 0;
        uint32_t run = 0;
        const int sh = T::BITS - k;
        for (uint32_t w = 0; w * 32u < len; ++w) {
            const uint32_t hi = p[3 * w], lo = p[3 * w + 1], va = p[3 * w + 2];
            const uint32_t nb = min(32u, len - w * 32u);
            for (uint32_t j = 0; j < nb; ++j) {
                wh = (wh >> 1) | ((W) ((hi >> j) & 1u) << (k - 1));
                wl = (wl >> 1) | ((W) ((lo >> j) & 1u) << (k - 1));
                run = ((va >> j) & 1u) ? run + 1 : 0;
                if (run >= (uint32_t) k) {
                    const W ka = T::brev(wh) >> sh;
                    const W kb = T::brev(wl) >> sh;
                    set_bit<W>(f.a, k >= 2 ? psi_a<W>(ka, k) : ka);
                    set_bit<W>(f.b, kb);
                    set_bit<W>(f.c, ka ^ kb);
                    set_bit<W>(f.d, ka | kb);
                    ++fed;
                }
            }
        }
    }
    if (kmers_fed) {
        // wave reduction, one atomic per wave
        for (int o = 32; o > 0; o >>= 1) fed += __shfl_down(fed, o, 64);
        if ((threadIdx.x & 63) == 0 && fed) atomicAdd(kmers_fed, (unsigned long long) fed);
    }
}

// ---------------------------------------------------------------------------
// search: one lane per read, exact reference control flow per lane:
// forward scan, greedy non-overlapping hits, stop at t; reverse scan only if
// the forward one failed (search_reads.h:45-83); 4-lane probe short-circuits
// a -> b -> c -> d (bloom_filter.h:124-131).  64 found flags leave the wave as
// one __ballot word = 8 bytes of the BooleanVector.
// ---------------------------------------------------------------------------
constexpr uint32_t SEARCH_MASK_WORDS = 8;   // reverse-strand lane-a bits remembered for the first 256 bases of a read

template <typename W, bool COUNT>
__global__ __launch_bounds__(256) COMMET_SGPRS void search_kernel(ReadsView rv, FilterView f, int k, int t,
                                                     const uint64_t *__restrict__ sel, uint64_t *__restrict__ tags,
                                                     uint64_t *__restrict__ found_out,
                                                     unsigned long long *__restrict__ counters,
                                                     unsigned long long *__restrict__ probe_counter, ActiveList al)
{
    using T = KeyTraits<W>;
    // per lane: lane-a bit of the reverse-complement key of every window probed in the forward pass, and which
    // windows were probed; [word][thread] so that a wave's accesses are conflict-free
    __shared__ uint32_t rc_bits[SEARCH_MASK_WORDS][256];
    __shared__ uint32_t rc_known[SEARCH_MASK_WORDS][256];
    const SearchLane me = search_lane(rv, al, sel, tags);
    const uint64_t r = me.r, word = me.word;
    const int lane = threadIdx.x & 63;
    const bool in_range = me.in_range, active = me.active;
    const bool paired = k >= 2;                 // plane A is stored strand-paired (psi_a)
    bool found = false;
    uint32_t probes = 0;   // filter words the REFERENCE control flow loads (COUNT builds only)
    if (active) {
        uint64_t t0;
        uint32_t len;
        read_extent(rv, r, t0, len);
        const uint32_t *p = rv.planes + 3 * t0;
        const int sh = T::BITS - k;
        const W mask = (k == T::BITS) ? ~(W) 0 : (((W) 1 << k) - 1);
        // Pruning (exact): a hit at window end q can still lead to t hits only if the remaining t-seen-1
        // non-overlapping windows fit behind it, q + (t-seen-1)*k <= len-1.  Past that point the reference keeps
        // probing but can no longer tag the read, so the scan of the strand stops (not in COUNT builds, which
        // reproduce the reference's probe count).
        const int last = (int) len - 1;
        uint32_t mask_words = 0;   // words of rc_bits / rc_known the forward pass has written for this lane
        for (int strand = 0; strand < 2 && !found; ++strand) {
            W wh = 0, wl = 0;
            uint32_t run = 0;
            int seen = 0;
            bool dead = false;   // no room left for the missing hits on this strand
            for (uint32_t w = 0; w * 32u < len && !found && !dead; ++w) {
                const uint32_t hi = p[3 * w], lo = p[3 * w + 1], va = p[3 * w + 2];
                const uint32_t nb = min(32u, len - w * 32u);
                uint32_t rcw = 0, knw = 0;
                if (strand == 1 && w < mask_words) {
                    rcw = rc_bits[w][threadIdx.x];
                    knw = rc_known[w][threadIdx.x];
                }
                for (uint32_t j = 0; j < nb && !found && !dead; ++j) {
                    wh = (wh >> 1) | ((W) ((hi >> j) & 1u) << (k - 1));
                    wl = (wl >> 1) | ((W) ((lo >> j) & 1u) << (k - 1));
                    run = ((va >> j) & 1u) ? run + 1 : 0;
                    if (!COUNT && (int) (32u * w + j) + (t - seen - 1) * k > last) {
                        dead = true;
                        break;
                    }
                    if (run >= (uint32_t) k) {
                        W ka, kb;
                        bool hit;
                        if (strand == 0) {
                            ka = T::brev(wh) >> sh;
                            kb = T::brev(wl) >> sh;
                            if (paired) {
                                // one load: lane-a bit of this key and of its reverse-complement partner
                                bool selfp;
                                const W addr = psi_a<W>(ka, k, selfp);
                                const uint32_t fw = f.a[addr >> 5];
                                const uint32_t bit = (uint32_t) addr & 31u;
                                hit = (fw >> bit) & 1u;
                                const uint32_t rcb = selfp ? (uint32_t) hit : ((fw >> (bit ^ 1u)) & 1u);
                                rcw |= rcb << j;
                                knw |= 1u << j;
                            } else {
                                hit = test_bit<W>(f.a, ka);
                            }
                        } else {
                            ka = ~wh & mask;
                            kb = ~wl & mask;
                            if (paired && ((knw >> j) & 1u)) hit = (rcw >> j) & 1u;       // remembered from the forward pass
                            else hit = test_bit<W>(f.a, paired ? psi_a<W>(ka, k) : ka);   // window skipped by the forward pass
                        }
                        if (COUNT) ++probes;
                        if (hit) {
                            hit = test_bit<W>(f.b, kb);
                            if (COUNT) ++probes;
                            if (hit) {
                                hit = test_bit<W>(f.c, ka ^ kb);
                                if (COUNT) ++probes;
                                if (hit) {
                                    hit = test_bit<W>(f.d, ka | kb);
                                    if (COUNT) ++probes;
                                }
                            }
                        }
                        if (hit) {
                            ++seen;
                            run = 0;                       // hash.clear(), search_reads.h:60
                            if (seen >= t) found = true;
                        }
                    }
                }
                if (strand == 0 && w < SEARCH_MASK_WORDS) {
                    rc_bits[w][threadIdx.x] = rcw;
                    rc_known[w][threadIdx.x] = knw;
                    mask_words = w + 1;
                }
            }
        }
    }
    publish_found(al, me, found, tags);
    if (found_out) {                            // (commet_search_reads: bitmap form only)
        const uint64_t fb = __ballot(found);
        if (lane == 0 && in_range) found_out[word] = fb;
    }
    if (counters) {
        __shared__ unsigned int wg_cnt[2];
        add_chunk_counters(counters, 0, 1, active, found ? 0 : -1, wg_cnt);
    }
    if (COUNT && probe_counter) {
        for (int o = 32; o > 0; o >>= 1) probes += __shfl_down(probes, o, 64);
        if (lane == 0 && probes) atomicAdd(probe_counter, (unsigned long long) probes);
    }
}

// ---------------------------------------------------------------------------
// windows per request of a cooperative tail fetch (powers of two <= 32).  More per request wastes L2 misses behind the hit
// that ends the scan; fewer costs round trips.  search_group_kernel on configs[1]: 4 / 8 / 16 / 32 -> 9.29 / 9.25 / 9.25 / 9.37 ms
// (9.39-9.65 with per-thread tails); search_group8_kernel on a 2 x 50 M-read pair: 8 / 16 / 32 -> 127 / 120.5 / 118.3 ms (136.6).
#ifndef GROUP_TAIL_WIN
#define GROUP_TAIL_WIN 16
#endif
#ifndef GROUP8_ABLATE
#define GROUP8_ABLATE 0   // timing ablations (wrong results): 1 no replay, 2 no tails
#endif
#ifndef G8_HEAVY
#define G8_HEAVY 20   // (8 / 12 / 16 / 20 / 24 / 28: 140 / 130 / 105 / 104 / 101 / 104 ms on a 2 x 50 M-read pair) search_group8_kernel: a scan with more lane-a candidates than this walks them itself
#endif
#ifndef G8_WAVES
#define G8_WAVES 1   // search_group8_kernel: waves per SIMD the register allocation is held to (1 = whatever 88 VGPRs allow: five workgroups per CU; 3 / 4 / 5 / 7 / 8: 105.6 / 105.7 / 103.7-106.5 / 108.3 / 110.5 ms per 50 M-read target)
#endif
#ifndef GROUP8_TAIL_WIN
#define GROUP8_TAIL_WIN 32
#endif
// ---------------------------------------------------------------------------
// search against a GROUP of chunk filters in one pass over the reads.
// The reference re-scans the search set once per index chunk (index_and_search.cpp:
// 255-277); a read's result is found_1 | found_2 | ... with found_c depending on
// chunk c's filter only.  With g <= GS chunk filters resident, their A planes are
// word-interleaved (il_a[w * GS + i] = plane A word w of chunk i), so ONE vector
// load yields the lane-a bits of a window for every chunk of the group and both
// strands (psi_a).  Per lane: (1) gather the lane-a bits of all complete windows
// into LDS masks, (2) replay the reference control flow chunk by chunk, strand by
// strand, on the masks, probing planes B, C, D of that chunk only on lane-a hits.
// counters: per chunk i of the group {scanned_i, found_i} at counters[i * cstride].
// ---------------------------------------------------------------------------
struct FilterGroupView {
    const uint32_t *il_a;        // interleaved A planes, stride GS words
    const uint32_t *slot0;       // first filter slot (4 planes); slot i at slot0 + i * slot_words
    uint64_t        slot_words;  // words per slot (4 * plane_words)
    uint64_t        plane_words;
    int             g;           // chunks in the group (<= GS)
};

template <int GS> struct GroupWords;
template <> struct GroupWords<2> {
    uint32_t x[2];
    __device__ __forceinline__ void load(const uint32_t *q)
    {
        const uint2 v = *(const uint2 *) q;
        x[0] = v.x, x[1] = v.y;
    }
};
template <> struct GroupWords<4> {
    uint32_t x[4];
    __device__ __forceinline__ void load(const uint32_t *q)
    {
        const uint4 v = *(const uint4 *) q;
        x[0] = v.x, x[1] = v.y, x[2] = v.z, x[3] = v.w;
    }
};

template <typename W, int GS, bool COUNT>
__global__ __launch_bounds__(256) COMMET_SGPRS void search_group_kernel(ReadsView rv, FilterGroupView fg, int k, int t, uint32_t nw_max,
                                                           const uint64_t *__restrict__ sel, uint64_t *__restrict__ tags,
                                                           unsigned long long *__restrict__ counters, uint32_t cstride,
                                                           unsigned long long *__restrict__ probe_counter, uint32_t rw_nw, ActiveList al)
{
    using T = KeyTraits<W>;
    extern __shared__ uint32_t gmask[];   // [chunk][strand][word][thread]
    // rw_nw != 0: the lane's read (3 * rw_nw words) is copied to LDS once, [word][thread]; the gather and every window
    // extraction of the replay then read LDS.  The kernel is bound by memory REQUESTS (L2 hits count too, at ~270 G/s):
    // ~100 of the ~150 requests per read were re-reads of these 12 words.
    uint32_t *const rw = gmask + (uint32_t) fg.g * 2u * nw_max * 256u;
    auto mask_at = [&](int i, int strand, uint32_t w) -> uint32_t & {
        return gmask[(((uint32_t) i * 2u + (uint32_t) strand) * nw_max + w) * 256u + threadIdx.x];
    };
    const SearchLane me = search_lane(rv, al, sel, tags);
    const uint64_t r = me.r;
    const int lane = threadIdx.x & 63;
    const bool active = me.active;
    bool found = false;
    int found_chunk = -1;
    uint32_t probes = 0;
    uint64_t t0 = 0;
    uint32_t len = 0;
    if (r < rv.n) read_extent(rv, r, t0, len);
    // every thread's read extent, for the threads that fetch its tail windows (the cooperative tails): on a set of many read lengths
    // the owner's extent is otherwise two more loads (goff) in front of each such fetch
    __shared__ unsigned long long wg_t0[256];
    __shared__ uint32_t wg_len[256];
    wg_t0[threadIdx.x] = t0, wg_len[threadIdx.x] = len;      // (first read behind a barrier)
    const uint32_t *p = rv.planes + 3 * t0;
    const int sh = T::BITS - k;
    const W mask = (k == T::BITS) ? ~(W) 0 : (((W) 1 << k) - 1);
    const bool staged = rw_nw && !COUNT;
    auto getw = [&](uint32_t i) -> uint32_t { return staged ? rw[i * 256u + threadIdx.x] : p[i]; };
    // Pruning (exact, see search_kernel): with `seen` hits a window ending at q matters only if
    // q + (t-seen-1)*k <= len-1.  The gather therefore covers the windows that can be a FIRST hit,
    // q <= pe = len-1-(t-1)*k (all of them in COUNT builds); later windows are needed only after a real
    // 4-lane hit and are then fetched by the whole workgroup (below) or, in COUNT builds, one by one in the replay.
    const int last = (int) len - 1;
    const int pe = COUNT ? last : last - (t - 1) * k;
    if (active) {
        if (staged) {
            const uint32_t n3 = 3u * ((len + 31u) >> 5);
            for (uint32_t i = 0; i < n3; ++i) rw[i * 256u + threadIdx.x] = p[i];
        }
        // (1) gather: lane-a bits of the complete windows ending at or before pe, all chunks, both strands
        {
            W wh = 0;
            uint32_t run = 0;
            for (uint32_t w = 0; w * 32u < len && (int) (w * 32u) <= pe; ++w) {
                const uint32_t hi = getw(3 * w), va = getw(3 * w + 2);
                const uint32_t nb = min(min(32u, len - w * 32u), (uint32_t) (pe - (int) (w * 32u) + 1));
                uint32_t fm[GS], rm[GS];
#pragma unroll
                for (int i = 0; i < GS; ++i) fm[i] = 0, rm[i] = 0;
                for (uint32_t j = 0; j < nb; ++j) {
                    wh = (wh >> 1) | ((W) ((hi >> j) & 1u) << (k - 1));
                    run = ((va >> j) & 1u) ? run + 1 : 0;
                    if (run >= (uint32_t) k) {
                        bool selfp;
                        const W addr = psi_a<W>(T::brev(wh) >> sh, k, selfp);
                        GroupWords<GS> gw;
                        gw.load(fg.il_a + (uint64_t) (addr >> 5) * GS);
                        const uint32_t bit = (uint32_t) addr & 31u;
#pragma unroll
                        for (int i = 0; i < GS; ++i) {
                            const uint32_t fb = (gw.x[i] >> bit) & 1u;
                            const uint32_t rb = selfp ? fb : ((gw.x[i] >> (bit ^ 1u)) & 1u);
                            fm[i] |= fb << j;
                            rm[i] |= rb << j;
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < GS; ++i) {
                    if (i < fg.g) {   // the dynamic LDS holds fg.g slots (launch_search_group_t), not GS
                        mask_at(i, 0, w) = fm[i];
                        mask_at(i, 1, w) = rm[i];
                    }
                }
            }
        }
    }
    if constexpr (!COUNT) {
        // (2) Sparse replay of the reference control flow per chunk (search_reads.h:45-83): the reference probes a window
        // iff its k bases are ACGT (that is what a gathered mask bit stands on) and it ends at least k bases after the
        // strand's last full hit.  A window whose lane-a bit is clear can never be a hit, so only the set mask bits are
        // visited, in order, with that rule; their keys come straight from the read's words (no rolling over the bases
        // in between).  Every thread walks the loops (the ones without work only for the barriers of the tails).
        __shared__ uint32_t tail_req[256], tail_bits[256];
        __shared__ uint32_t tail_n;
        for (int i = 0; i < fg.g; ++i) {   // (uniform)
            const uint32_t *pb = fg.slot0 + (uint64_t) i * fg.slot_words + fg.plane_words;
            const uint32_t *pc = pb + fg.plane_words;
            const uint32_t *pd = pc + fg.plane_words;
            for (int strand = 0; strand < 2; ++strand) {
                int seen = 0, next_ok = 0;
                bool dead = !active || found;
                auto probe_bcd = [&](W wh, W wl) -> bool {
                    W ka, kb;
                    if (strand == 0) ka = T::brev(wh) >> sh, kb = T::brev(wl) >> sh;
                    else ka = ~wh & mask, kb = ~wl & mask;
                    return test_bit<W>(pb, kb) && test_bit<W>(pc, ka ^ kb) && test_bit<W>(pd, ka | kb);
                };
                for (uint32_t w = 0; (int) (w * 32u) <= pe && !found && !dead; ++w) {
                    uint32_t m = mask_at(i, strand, w);
                    if (!m) continue;
                    ItemWords<W> it;
                    it.load_with(getw, w);
                    while (m && !found) {
                        const uint32_t j = (uint32_t) __ffs((int) m) - 1u;
                        m &= m - 1u;
                        const int q = (int) (32u * w + j);
                        if (q < next_ok) continue;
                        if (q + (t - seen - 1) * k > last) {
                            dead = true;
                            break;
                        }
                        W wh, wl;
                        (void) it.window(j, k, mask, wh, wl);   // valid: the gather saw a complete window here
                        if (probe_bcd(wh, wl)) {
                            ++seen;
                            next_ok = q + k;
                            if (seen >= t) found = true;
                        }
                    }
                }
                // Windows behind the gathered ones matter only after a first full hit (pruning, see above).  Their lane-a
                // bits are fetched by the whole workgroup, GROUP_TAIL_WIN windows per request: the threads that need a tail
                // post (read, first window end), thread p takes window p % GROUP_TAIL_WIN of request p / GROUP_TAIL_WIN —
                // one round trip with every lane busy (see tq_replay_kernel); then only the set ones are probed.
                for (int qb = max(pe + 1, next_ok);; qb += GROUP_TAIL_WIN) {   // (uniform trip count)
                    const bool want = !found && !dead && seen >= 1 && qb <= last && qb + (t - seen - 1) * k <= last;
                    if (threadIdx.x == 0) tail_n = 0;
                    if (!__syncthreads_or(want)) break;
                    if (want) {
                        tail_req[atomicAdd(&tail_n, 1u)] = threadIdx.x | ((uint32_t) qb << 8);
                        tail_bits[threadIdx.x] = 0;
                    }
                    __syncthreads();
                    const uint32_t n_pairs = tail_n * (uint32_t) GROUP_TAIL_WIN;
                    for (uint32_t pr = threadIdx.x; pr < n_pairs; pr += 256) {
                        const uint32_t rq = tail_req[pr / GROUP_TAIL_WIN], owner = rq & 255u, wi = pr % GROUP_TAIL_WIN;
                        const int q = (int) (rq >> 8) + (int) wi;
                        if (q >= (int) wg_len[owner]) continue;
                        const uint32_t *op = rv.planes + 3 * wg_t0[owner];
                        ItemWords<W> it;
                        it.load_with([&](uint32_t x) -> uint32_t { return staged ? rw[x * 256u + owner] : op[x]; }, (uint32_t) q >> 5);
                        W wh, wl;
                        if (!it.window((uint32_t) q & 31u, k, mask, wh, wl)) continue;   // a base that is not ACGT: no k-mer here
                        const W ka = strand ? (W) (~wh & mask) : (W) (T::brev(wh) >> sh);
                        const W addr = psi_a<W>(ka, k);
                        const uint32_t v = fg.il_a[(uint64_t) (addr >> 5) * GS + i];
                        if ((v >> ((uint32_t) addr & 31u)) & 1u) atomicOr(&tail_bits[owner], 1u << wi);
                    }
                    __syncthreads();
                    if (want) {
                        uint32_t m = tail_bits[threadIdx.x];
                        while (m && !found) {
                            const int q = qb + (__ffs((int) m) - 1);
                            m &= m - 1u;
                            if (q < next_ok) continue;
                            if (q + (t - seen - 1) * k > last) {
                                dead = true;
                                break;
                            }
                            ItemWords<W> it;
                            it.load_with(getw, (uint32_t) q >> 5);
                            W wh, wl;
                            (void) it.window((uint32_t) q & 31u, k, mask, wh, wl);
                            if (probe_bcd(wh, wl)) {
                                ++seen;
                                next_ok = q + k;   // the next complete window ends k bases later
                                if (seen >= t) found = true;
                            }
                        }
                    }
                }
            }
            if (found && found_chunk < 0) found_chunk = i;
        }
    } else if (active) {
        // (2, COUNT builds) the reference's walk base by base, every probe counted
        for (int i = 0; i < fg.g && !found; ++i) {
            const uint32_t *pb = fg.slot0 + (uint64_t) i * fg.slot_words + fg.plane_words;
            const uint32_t *pc = pb + fg.plane_words;
            const uint32_t *pd = pc + fg.plane_words;
            for (int strand = 0; strand < 2 && !found; ++strand) {
                W wh = 0, wl = 0;
                uint32_t run = 0;
                int seen = 0;
                bool dead = false;
                for (uint32_t w = 0; w * 32u < len && !found && !dead; ++w) {
                    const uint32_t hi = p[3 * w], lo = p[3 * w + 1], va = p[3 * w + 2];
                    const uint32_t nb = min(32u, len - w * 32u);
                    const uint32_t am = ((int) (w * 32u) <= pe) ? mask_at(i, strand, w) : 0u;
                    for (uint32_t j = 0; j < nb && !found && !dead; ++j) {
                        wh = (wh >> 1) | ((W) ((hi >> j) & 1u) << (k - 1));
                        wl = (wl >> 1) | ((W) ((lo >> j) & 1u) << (k - 1));
                        run = ((va >> j) & 1u) ? run + 1 : 0;
                        const int q = (int) (32u * w + j);
                        if (!COUNT && q + (t - seen - 1) * k > last) {
                            dead = true;
                            break;
                        }
                        if (run >= (uint32_t) k) {
                            W ka, kb;
                            if (strand == 0) {
                                ka = T::brev(wh) >> sh;
                                kb = T::brev(wl) >> sh;
                            } else {
                                ka = ~wh & mask;
                                kb = ~wl & mask;
                            }
                            bool hit;
                            if (q <= pe) hit = (am >> j) & 1u;          // gathered
                            else {                                      // behind a real hit: probe chunk i's A plane
                                const W addr = psi_a<W>(ka, k);
                                hit = (fg.il_a[(uint64_t) (addr >> 5) * GS + i] >> ((uint32_t) addr & 31u)) & 1u;
                            }
                            if (COUNT) ++probes;
                            if (hit) {
                                hit = test_bit<W>(pb, kb);
                                if (COUNT) ++probes;
                                if (hit) {
                                    hit = test_bit<W>(pc, ka ^ kb);
                                    if (COUNT) ++probes;
                                    if (hit) {
                                        hit = test_bit<W>(pd, ka | kb);
                                        if (COUNT) ++probes;
                                    }
                                }
                            }
                            if (hit) {
                                ++seen;
                                run = 0;
                                if (seen >= t) found = true;
                            }
                        }
                    }
                }
            }
            if (found) found_chunk = i;
        }
    }
    publish_found(al, me, found, tags);
    if (counters) {
        __shared__ unsigned int wg_cnt[2 * GS];
        add_chunk_counters(counters, cstride, fg.g, active, found_chunk, wg_cnt);
    }
    if (COUNT && probe_counter) {
        for (int o = 32; o > 0; o >>= 1) probes += __shfl_down(probes, o, 64);
        if (lane == 0 && probes) atomicAdd(probe_counter, (unsigned long long) probes);
    }
}

// ---------------------------------------------------------------------------
// Up to EIGHT chunk filters in one pass (A planes interleaved with stride 8: the eight words of a window are 32
// contiguous bytes of one 64-byte sector, still one request).  For read sets whose first-hit windows number at most
// 32 * MW per read (len - t*k + 1; MW = 2 or 3: 100-bp reads at k = 32, t = 2 have 37, 150-bp reads 87) the gathered
// lane-a bits live in REGISTERS,
// bit = window end - (k-1): no LDS at all, so the occupancy does not fall with the number of filters.  Replay = the
// sparse replay of search_group_kernel, unrolled over the filters.  Sets of more than four chunks need half the passes.
// ---------------------------------------------------------------------------
template <typename W, int MW>
__global__ __launch_bounds__(256, G8_WAVES) COMMET_SGPRS void search_group8_kernel(ReadsView rv, FilterGroupView fg, int k, int t,
                                                            const uint64_t *__restrict__ sel, uint64_t *__restrict__ tags,
                                                            unsigned long long *__restrict__ counters, uint32_t cstride, ActiveList al,
                                                            uint32_t job_mask, uint64_t job_tag_words)
{
    // job_mask != 0: the filters of the pass belong to SEVERAL jobs that search this one set (Commet.py's J2 jobs of a reference set,
    // its J3 jobs of a target: each against two chunk filters of another restricted index set, commet_index_many_and_search): bit i
    // set = filter i is the first of a job.  The lane-a gather — 37 of a J2 / J3 job's ~55 requests per read — then serves every job
    // of the pass; everything behind it runs job by job: `found` starts afresh at a job's first filter, job j's found flags go to
    // tags + j * job_tag_words (zeroed by the host: a job never spans passes), its counters to its own chunks' slots.
    using T = KeyTraits<W>;
    constexpr int GS = 8;
    // tails (the windows behind the first-hit ones, for a scan that has a hit but not yet t of them) are fetched by the
    // whole workgroup: see tq_replay_kernel (tile_search.hpp), where doing so took 1.8 ms of tails to 1.1
    __shared__ uint32_t tail_req[256], tail_bits[256];
    __shared__ uint32_t tail_n;
    // the first-hit candidates of a scan are probed by the whole workgroup as well (see (2) below)
    __shared__ uint16_t cand[256 * G8_HEAVY];            // thread (8 bits) | window (8 bits: <= 255 first-hit windows)
    __shared__ uint32_t full_hit[MW][256];
    __shared__ uint32_t cand_n;
    const bool multi = job_mask != 0;
    const SearchLane me = search_lane(rv, al, sel, multi ? nullptr : tags);
    const uint64_t r = me.r;
    const bool active = me.active;
    __shared__ unsigned int wg_cnt[2 * GS];
    int job = -1, job_first = 0;
    bool found = false;
    int found_chunk = -1;
    uint64_t t0 = 0;
    uint32_t len = 0;
    if (r < rv.n) read_extent(rv, r, t0, len);
    // every thread's read extent, for the threads that probe its candidates and tail windows: on a set of many read lengths the owner's
    // extent is otherwise two more loads (goff) in front of each of those probes' chains
    __shared__ unsigned long long wg_t0[256];
    __shared__ uint32_t wg_len[256];
    wg_t0[threadIdx.x] = t0, wg_len[threadIdx.x] = len;      // (first read behind a barrier)
    const uint32_t *p = rv.planes + 3 * t0;
    const int sh = T::BITS - k;
    const W mask = (k == T::BITS) ? ~(W) 0 : (((W) 1 << k) - 1);
    const int last = (int) len - 1;
    const int pe = last - (t - 1) * k;     // last window that can be a first hit; the host guarantees pe - (k-1) < 32 * MW (MW = 2, 3, 4, 6 or 8: mask_words, capi/search_dispatch.hpp)
    const int q0 = k - 1;
    uint32_t fm[MW][GS], rm[MW][GS];       // [word of the 32 * MW relative positions][filter]
#pragma unroll
    for (int h = 0; h < MW; ++h)
#pragma unroll
        for (int i = 0; i < GS; ++i) fm[h][i] = 0, rm[h][i] = 0;
    // (1) gather (one window per round trip on purpose: with 2 / 4 / 8 windows' loads in flight per thread the kernel took 105.5 / 104.5 /
    // 124.2 instead of 101.3 ms per 50 M-read target, and search_group_kernel 48.4 / 49.3 / 49.4 instead of 47.0 — at the request
    // ceiling more requests in flight per workgroup only lengthen the queues)
    if (active) {
        W wh = 0;
        uint32_t run = 0, cw = ~0u, hi = 0, va = 0;
        auto roll = [&](int pos) {
            const uint32_t w = (uint32_t) pos >> 5, j = (uint32_t) pos & 31u;
            if (w != cw) hi = p[3 * w], va = p[3 * w + 2], cw = w;
            wh = (wh >> 1) | ((W) ((hi >> j) & 1u) << (k - 1));
            run = ((va >> j) & 1u) ? run + 1 : 0;
        };
        for (int pos = 0; pos < q0 && pos <= pe; ++pos) roll(pos);
#pragma unroll
        for (int h = 0; h < MW; ++h) {
            for (int jj = 0; jj < 32; ++jj) {
                const int q = q0 + 32 * h + jj;
                if (q > pe) break;
                roll(q);
                if (run >= (uint32_t) k) {
                    bool selfp;
                    const W addr = psi_a<W>(T::brev(wh) >> sh, k, selfp);
                    const uint32_t *src = fg.il_a + (uint64_t) (addr >> 5) * GS;
                    const uint4 v = *(const uint4 *) src, u = *(const uint4 *) (src + 4);
                    const uint32_t x[GS] = {v.x, v.y, v.z, v.w, u.x, u.y, u.z, u.w};
                    const uint32_t bit = (uint32_t) addr & 31u;
#pragma unroll
                    for (int i = 0; i < GS; ++i) {
                        const uint32_t fb = (x[i] >> bit) & 1u;
                        const uint32_t rb = selfp ? fb : ((x[i] >> (bit ^ 1u)) & 1u);
                        fm[h][i] |= fb << jj;
                        rm[h][i] |= rb << jj;
                    }
                }
            }
        }
    }
    // (2) sparse replay, filter by filter (unrolled: the masks are registers); every thread walks the loop, the ones without
    // work only for its barriers
    if (GROUP8_ABLATE & 1) {   // keep the gather alive
        uint32_t n = 0;
#pragma unroll
        for (int h = 0; h < MW; ++h)
#pragma unroll
            for (int i = 0; i < GS; ++i) n += __popc(fm[h][i]) + 3 * __popc(rm[h][i]);
        if (n == (uint32_t) t) found = true;   // (never with the t the bench uses... wrong results anyway)
    }
    // (several jobs) what a job leaves behind: its found flags, one ballot word per 64 reads, and its chunks' counters
    auto end_job = [&](int first, int end) {
        if (al.ids) {                                     // list form (a ragged set's reads in order of their window counts): one atomic OR per found read
            if (found && tags) (void) __hip_atomic_fetch_or(tags + (uint64_t) job * job_tag_words + me.word, 1ull << (me.r & 63ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const uint64_t fb = __ballot(found);
            if ((threadIdx.x & 63) == 0 && me.in_range && tags) tags[(uint64_t) job * job_tag_words + me.word] = fb;
        }
        if (counters) add_chunk_counters(counters + (uint64_t) first * cstride, cstride, end - first, active, found_chunk < 0 ? -1 : found_chunk - first, wg_cnt);
    };
#pragma unroll
    for (int i = 0; i < GS; ++i) {
        if (i >= fg.g) continue;   // (uniform)
        if (multi && ((job_mask >> i) & 1u)) {   // (uniform) filter i opens a job
            if (job >= 0) end_job(job_first, i);
            ++job, job_first = i;
            found = false, found_chunk = -1;
        }
        const uint32_t *pb = fg.slot0 + (uint64_t) i * fg.slot_words + fg.plane_words;
        const uint32_t *pc = pb + fg.plane_words;
        const uint32_t *pd = pc + fg.plane_words;
#pragma unroll
        for (int strand = 0; strand < 2; ++strand) {
            int seen = 0, next_ok = 0;
            bool dead = !active || found || (GROUP8_ABLATE & 1);
            auto probe_bcd = [&](W wh, W wl) -> bool {
                W ka, kb;
                if (strand == 0) ka = T::brev(wh) >> sh, kb = T::brev(wl) >> sh;
                else ka = ~wh & mask, kb = ~wl & mask;
                return test_bit<W>(pb, kb) && test_bit<W>(pc, ka ^ kb) && test_bit<W>(pd, ka | kb);
            };
            // The scan's lane-a candidates (~3 of a read that shares nothing with the chunk, a different number in every
            // lane) are posted in LDS and dealt out evenly: thread p probes candidate p, p + 256, ... through planes B, C, D
            // and marks the full hits in full_hit[]; walking them per thread kept a wave in as many dependent round trips as
            // its busiest lane has candidates.  A scan with more than G8_HEAVY candidates is a read that shares sequence
            // with the chunk: it walks its own candidates and stops at t hits, as the reference does.
            uint32_t mm[MW], ncand = 0;
#pragma unroll
            for (int h = 0; h < MW; ++h) mm[h] = dead ? 0u : (strand ? rm[h][i] : fm[h][i]), ncand += __popc(mm[h]);
            const bool self = ncand > G8_HEAVY;
            if (threadIdx.x == 0) cand_n = 0;
#pragma unroll
            for (int h = 0; h < MW; ++h) full_hit[h][threadIdx.x] = 0;
            __syncthreads();
            if (!self && ncand) {
                uint32_t at = atomicAdd(&cand_n, ncand);   // <= 256 * G8_HEAVY in all
#pragma unroll
                for (int h = 0; h < MW; ++h)
                    for (uint32_t m = mm[h]; m; m &= m - 1u) cand[at++] = (uint16_t) (threadIdx.x | ((32u * h + (uint32_t) __ffs((int) m) - 1u) << 8));
            }
            __syncthreads();
            {
                const uint32_t n_cand = cand_n;
                constexpr int U = 2;   // candidates per thread and round: their probes are in flight together (1 / 2 / 4: no difference)
                for (uint32_t c0 = threadIdx.x; c0 < n_cand; c0 += U * 256) {
                    uint32_t owner[U], wq[U], vb[U];
                    W ka[U], kb[U];
                    bool have[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint32_t ci = c0 + (uint32_t) u * 256u;
                        have[u] = ci < n_cand;
                        owner[u] = 0, wq[u] = 0, vb[u] = 0, ka[u] = 0, kb[u] = 0;
                        if (!have[u]) continue;
                        const uint32_t e = cand[ci];
                        owner[u] = e & 255u, wq[u] = e >> 8;
                        const int q = q0 + (int) wq[u];
                        ItemWords<W> it;
                        it.load(rv.planes + 3 * wg_t0[owner[u]], (uint32_t) q >> 5);
                        W wh, wl;
                        (void) it.window((uint32_t) q & 31u, k, mask, wh, wl);   // complete: the gather saw it
                        if (strand == 0) ka[u] = T::brev(wh) >> sh, kb[u] = T::brev(wl) >> sh;
                        else ka[u] = ~wh & mask, kb[u] = ~wl & mask;
                        vb[u] = pb[kb[u] >> 5];
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (!have[u] || !((vb[u] >> ((uint32_t) kb[u] & 31u)) & 1u)) continue;
                        const W kc = ka[u] ^ kb[u], kd = ka[u] | kb[u];
                        const uint32_t vc = pc[kc >> 5], vd = pd[kd >> 5];
                        if ((vc >> ((uint32_t) kc & 31u)) & (vd >> ((uint32_t) kd & 31u)) & 1u) atomicOr(&full_hit[wq[u] >> 5][owner[u]], 1u << (wq[u] & 31u));
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int h = 0; h < MW; ++h) {
                uint32_t m = self ? mm[h] : full_hit[h][threadIdx.x];
                while (m && !found && !dead) {
                    const uint32_t jj = (uint32_t) __ffs((int) m) - 1u;
                    m &= m - 1u;
                    const int q = q0 + 32 * h + (int) jj;
                    if (q < next_ok) continue;
                    if (q + (t - seen - 1) * k > last) {
                        dead = true;
                        break;
                    }
                    if (self) {
                        ItemWords<W> it;
                        it.load(p, (uint32_t) q >> 5);
                        W wh, wl;
                        (void) it.window((uint32_t) q & 31u, k, mask, wh, wl);
                        if (!probe_bcd(wh, wl)) continue;
                    }
                    ++seen;
                    next_ok = q + k;
                    if (seen >= t) found = true;
                }
            }
            // windows behind the gathered ones, after a first full hit only, GROUP8_TAIL_WIN at a time
            for (int qb = max(pe + 1, next_ok);; qb += GROUP8_TAIL_WIN) {   // (uniform trip count: every thread takes part in the barriers)
                const bool want = !(GROUP8_ABLATE & 2) && !found && !dead && seen >= 1 && qb <= last && qb + (t - seen - 1) * k <= last;
                if (threadIdx.x == 0) tail_n = 0;
                if (!__syncthreads_or(want)) break;
                if (want) {
                    tail_req[atomicAdd(&tail_n, 1u)] = threadIdx.x | ((uint32_t) qb << 8);
                    tail_bits[threadIdx.x] = 0;
                }
                __syncthreads();
                const uint32_t n_pairs = tail_n * (uint32_t) GROUP8_TAIL_WIN;
                for (uint32_t pr = threadIdx.x; pr < n_pairs; pr += 256) {
                    const uint32_t rq = tail_req[pr / GROUP8_TAIL_WIN], owner = rq & 255u, w = pr % GROUP8_TAIL_WIN;
                    const int q = (int) (rq >> 8) + (int) w;
                    if (q >= (int) wg_len[owner]) continue;
                    ItemWords<W> it;
                    it.load(rv.planes + 3 * wg_t0[owner], (uint32_t) q >> 5);
                    W wh, wl;
                    if (!it.window((uint32_t) q & 31u, k, mask, wh, wl)) continue;   // a base that is not ACGT: no k-mer here
                    const W ka = strand ? (W) (~wh & mask) : (W) (T::brev(wh) >> sh);
                    const W addr = psi_a<W>(ka, k);
                    const uint32_t v = fg.il_a[(uint64_t) (addr >> 5) * GS + i];
                    if ((v >> ((uint32_t) addr & 31u)) & 1u) atomicOr(&tail_bits[owner], 1u << w);
                }
                __syncthreads();
                if (want) {
                    uint32_t m = tail_bits[threadIdx.x];
                    while (m && !found) {
                        const int q = qb + (__ffs((int) m) - 1);
                        m &= m - 1u;
                        if (q < next_ok) continue;
                        if (q + (t - seen - 1) * k > last) {
                            dead = true;
                            break;
                        }
                        ItemWords<W> it;
                        it.load(p, (uint32_t) q >> 5);
                        W wh, wl;
                        (void) it.window((uint32_t) q & 31u, k, mask, wh, wl);
                        if (probe_bcd(wh, wl)) {
                            ++seen;
                            next_ok = q + k;
                            if (seen >= t) found = true;
                        }
                    }
                }
            }
        }
        if (found && found_chunk < 0) found_chunk = i;
    }
    if (multi) {
        if (job >= 0) end_job(job_first, fg.g);
        return;
    }
    publish_found(al, me, found, tags);
    if (counters) add_chunk_counters(counters, cstride, fg.g, active, found_chunk, wg_cnt);
}

// il_a[w * GS + i] = plane A word w of filter slot i (0 for i >= g)
template <int GS>
__global__ __launch_bounds__(256) void interleave_a_kernel(const uint32_t *__restrict__ slot0, uint64_t slot_words,
                                                           uint64_t plane_words, int g, uint32_t *__restrict__ il_a)
{
    const uint64_t stride = (uint64_t) gridDim.x * 256ull;
    for (uint64_t w = blockIdx.x * 256ull + threadIdx.x; w < plane_words; w += stride) {
        uint32_t x[GS];
#pragma unroll
        for (int i = 0; i < GS; ++i) x[i] = i < g ? slot0[(uint64_t) i * slot_words + w] : 0u;
        if (GS == 2) *(uint2 *) (il_a + w * 2) = make_uint2(x[0], x[1]);
        else if (GS == 4) *(uint4 *) (il_a + w * 4) = make_uint4(x[0], x[1 % GS], x[2 % GS], x[3 % GS]);
        else {
            *(uint4 *) (il_a + w * 8) = make_uint4(x[0], x[1 % GS], x[2 % GS], x[3 % GS]);
            *(uint4 *) (il_a + w * 8 + 4) = make_uint4(x[4 % GS], x[5 % GS], x[6 % GS], x[7 % GS]);
        }
    }
}

// ---------------------------------------------------------------------------
// filter -> reference byte layout (bloom_filter.h:63-70,114-117); tests only.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void export_reference_kernel(FilterView f, int k, uint64_t nbytes, uint8_t *__restrict__ out)
{
    const uint64_t stride = (uint64_t) gridDim.x * 256ull;
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < nbytes; i += stride) {
        const uint64_t k0 = 2 * i, k1 = 2 * i + 1;
        uint32_t b = 0;
        b |= test_bit<uint64_t>(f.a, k >= 2 ? psi_a<uint64_t>(k0, k) : k0) << 7;
        b |= test_bit<uint64_t>(f.b, k0) << 6;
        b |= test_bit<uint64_t>(f.c, k0) << 5;
        b |= test_bit<uint64_t>(f.d, k0) << 4;
        b |= test_bit<uint64_t>(f.a, k >= 2 ? psi_a<uint64_t>(k1, k) : k1) << 3;
        b |= test_bit<uint64_t>(f.b, k1) << 2;
        b |= test_bit<uint64_t>(f.c, k1) << 1;
        b |= test_bit<uint64_t>(f.d, k1);
        out[i] = (uint8_t) b;
    }
}

// ---------------------------------------------------------------------------
// random-access microbenchmarks (practical ceilings)
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// MODE 0: plain 4-byte gather, 1: atomic OR, 2: non-temporal gather, 3: agent-scope (sc1, L1-bypassing) gather
template <int MODE>
__global__ __launch_bounds__(256) void membench_kernel(uint32_t *__restrict__ table, uint64_t word_mask, uint32_t iters,
                                                       uint32_t *__restrict__ sink)
{
    const uint64_t tid = blockIdx.x * 256ull + threadIdx.x;
    uint32_t acc = 0;
    uint64_t s = tid * 0x100000001B3ull + 12345;
    for (uint32_t i = 0; i < iters; ++i) {
        s = splitmix64(s);
        const uint64_t idx = s & word_mask;
        if (MODE == 1) (void) __hip_atomic_fetch_or(table + idx, 1u << (s >> 59), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 2) acc ^= __builtin_nontemporal_load(table + idx);
        else if (MODE == 3) acc ^= __hip_atomic_load(table + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else acc ^= table[idx];
    }
    if (MODE != 1 && acc == 0x12345678u) sink[0] = acc;   // keep the loads alive
}

// Windowed gathers: what the tiled search's probe pass would see.  The table is cut into windows of win_words words;
// workgroup b works on XCD b % 8 (workgroups are dealt to the XCDs round-robin) and sweeps that XCD's eighth of the
// windows in order, so that at any time the workgroups of one XCD gather from one or two windows — which then live in
// that XCD's L2.  `xcd_aware` = 0 lets consecutive workgroups take consecutive windows instead (windows spread over the XCDs).
__global__ __launch_bounds__(256) void membench_window_kernel(const uint32_t *__restrict__ table, uint64_t n_windows, uint32_t win_words,
                                                              uint32_t per_window, int xcd_aware, uint32_t *__restrict__ sink)
{
    // persistent sweep: the workgroups of XCD x (b % 8 == x) walk windows x * n/8 .. (x+1) * n/8 - 1 together, every
    // thread doing per_window gathers in each; xcd_aware = 0: workgroup b starts b windows further on (no sharing)
    const uint64_t per_xcd = n_windows / 8;
    uint64_t s = (blockIdx.x * 256ull + threadIdx.x) * 0x100000001B3ull + 777;
    uint32_t acc = 0;
    for (uint64_t i = 0; i < per_xcd; ++i) {
        const uint64_t win = xcd_aware ? (blockIdx.x % 8) * per_xcd + i : (blockIdx.x * 7919ull + i) % n_windows;
        const uint32_t *w = table + win * win_words;
        for (uint32_t j = 0; j < per_window; ++j) {
            s = splitmix64(s);
            acc ^= w[s % win_words];
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// LDS microbenchmark: what one CU's LDS pipeline does per cycle for this path's access shapes (index_part.hpp).
// MODE 0: atomic add, no return; 1: atomic add, rank returned; 2: atomic OR, no return; 3: plain store; 4: plain load;
// 5: atomic add with return, conflict-free addresses (lane-private counters).  Addresses: uniform over n_words.
template <int MODE>
__global__ __launch_bounds__(512) void ldsbench_kernel(uint32_t n_words, uint32_t iters, uint32_t *__restrict__ sink)
{
    extern __shared__ uint32_t lds_tab[];
    for (uint32_t i = threadIdx.x; i < n_words; i += 512) lds_tab[i] = 0;
    __syncthreads();
    uint32_t x = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    const uint32_t m = n_words - 1;   // power of two
    for (uint32_t i = 0; i < iters; ++i) {
        x ^= x << 13, x ^= x >> 17, x ^= x << 5;   // xorshift32
        uint32_t a = (x >> 7) & m;
        if (MODE == 5) a = ((a & ~63u) | (threadIdx.x & 63u)) & m;
        if (MODE == 0) atomicAdd(&lds_tab[a], 1u);
        else if (MODE == 1 || MODE == 5) acc += atomicAdd(&lds_tab[a], 1u);
        else if (MODE == 2) atomicOr(&lds_tab[a], 1u << (x & 31u));
        else if (MODE == 3) lds_tab[a] = x;
        else acc += lds_tab[a];
    }
    __syncthreads();
    if (acc == 0x12345678u || lds_tab[threadIdx.x & m] == 0xFFFFFFFFu) sink[0] = acc;
}

}  // namespace commet

// plan_capi.cpp — C entry points of the host-side chunk planner (read_iter.hpp)
// for CPU-only tests (no HIP): libcommet_plan.so.  The HIP library uses the same
// header directly.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../read_iter.hpp"

using namespace commet;

extern "C" {

// files: n_files pairs (first, count). Returns the number of chunks; chunk_out
// gets 4 values per chunk (first, last, n_reads, kmers) up to cap chunks;
// indexed_bits_out (n_reads/8+1 bytes) the reads fed; stats_out = {indexed_reads, kmers}.
uint64_t commet_plan_index(const uint64_t *files, int n_files, const uint8_t *select, const uint64_t *empty_reads,
                           uint64_t n_empty, const uint32_t *kcnt, uint64_t n_reads, uint64_t max_kmer, int fast,
                           uint64_t *chunk_out, uint64_t cap, uint8_t *indexed_bits_out, uint64_t *stats_out)
{
    std::vector<FileSpan> fs;
    for (int i = 0; i < n_files; ++i) fs.push_back(FileSpan{files[2 * i], files[2 * i + 1]});
    std::vector<uint64_t> er(empty_reads, empty_reads + n_empty);
    std::vector<uint64_t> prefix;
    IndexPlan plan;
    if (fast >= 2 && n_reads && plan_blocks_ok(fs, select, er, max_kmer)) {   // fast = 2 + log2(block size); the library sums on the device
        const uint64_t bs = 1ull << (fast - 2);
        std::vector<uint64_t> sums((n_reads + bs - 1) / bs, 0);
        for (uint64_t r = 0; r < n_reads; ++r)
            if (!select || bit_at(select, r)) sums[r / bs] += kcnt[r];
        plan = plan_index_blocks(select, [&](uint64_t q) { return kcnt[q]; }, n_reads, max_kmer, sums.data(), bs);
    } else if (fast && plan_fast_ok(fs, select, er, max_kmer)) {
        build_kmer_prefix(kcnt, n_reads, prefix);
        plan = plan_index_fast(prefix, n_reads, max_kmer);
    } else if (fast && select && er.empty()) {
        plan = plan_index_select(fs, select, kcnt, n_reads, max_kmer);
    } else {
        plan = plan_index(fs, select, er, kcnt, n_reads, max_kmer);
    }
    for (uint64_t c = 0; c < plan.chunks.size() && c < cap; ++c) {
        chunk_out[4 * c + 0] = plan.chunks[c].first;
        chunk_out[4 * c + 1] = plan.chunks[c].last;
        chunk_out[4 * c + 2] = plan.chunks[c].n_reads;
        chunk_out[4 * c + 3] = plan.chunks[c].kmers;
    }
    if (indexed_bits_out) memcpy(indexed_bits_out, plan.indexed_bits.data(), plan.indexed_bits.size());
    if (stats_out) {
        stats_out[0] = plan.indexed_reads;
        stats_out[1] = plan.kmers;
    }
    return plan.chunks.size();
}

uint64_t commet_plan_search(const uint64_t *files, int n_files, const uint8_t *select, const uint64_t *empty_reads,
                            uint64_t n_empty, uint64_t n_reads, int fast, uint8_t *visited_bits_out)
{
    std::vector<FileSpan> fs;
    for (int i = 0; i < n_files; ++i) fs.push_back(FileSpan{files[2 * i], files[2 * i + 1]});
    std::vector<uint64_t> er(empty_reads, empty_reads + n_empty);
    uint64_t n = 0;
    std::vector<uint8_t> bits;
    if (fast && plan_fast_ok(fs, select, er, 1)) bits = plan_search_fast(n_reads, &n);
    else if (fast && select && er.empty()) bits = plan_search_select(fs, select, n_reads, &n);
    else bits = plan_search(fs, select, er, n_reads, &n);
    memcpy(visited_bits_out, bits.data(), bits.size());
    return n;
}

}  // extern "C"

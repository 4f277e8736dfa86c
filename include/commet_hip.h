/*
 * commet_hip.h — C ABI of the MI355X-native index_and_search hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 * The reference (pierrepeterlongo/commet) has no FFI for this path: its
 * boundary is the two free functions
 *     BloomFilter * index_reads (FileManager*, k, min_hits, max_kmer, &nb_indexed_reads)   include/index_reads.h:41
 *     unsigned long search_reads(const BloomFilter*, FileManager*, k, min_hits, &nb_searched) include/search_reads.h:34
 * plus the chunk loop of main() (src/index_and_search.cpp:241-277).  Those are
 * pull-based on a std::string iterator, which a device path cannot use, so the
 * entry points below are the batch-based equivalents.  Each
 * declaration cites the reference interface it replaces.  INTEGRATION.md shows
 * the binding a Commet maintainer would add in src/index_and_search.cpp.
 *
 * Conventions
 *   - every function returning int returns 0 on success, non-zero on error;
 *     commet_last_error() then gives a message (thread-local).
 *   - bit arrays are LSB-first per byte, exactly BooleanVector's layout
 *     (include/boolean_vector.h:73-80, 222-233): read i = byte i/8, mask 1<<(i%8).
 *     A bit array over n reads has n/8+1 bytes (boolean_vector.h:130).
 *   - one ctx per (device, stream); calls on one ctx are serialised by the
 *     caller; several ctxs may be used from several host threads.  One exception:
 *     read sets are made on a stream of their own, so ONE host thread may build read
 *     sets of a ctx (commet_readset_create ... commet_readset_finalize,
 *     commet_readset_from_*, commet_readset_load) while another runs
 *     commet_index_and_search on sets that are complete (commet_amd/matrix.py does).
 *   - there is NO CPU fallback: without a usable HIP device commet_create fails.
 */
#ifndef COMMET_HIP_H_
#define COMMET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct commet_ctx     commet_ctx;
typedef struct commet_readset commet_readset;

/* ---- library / device ---------------------------------------------------- */
const char *commet_version(void);
const char *commet_last_error(void);
/* number of HIP devices visible (0 if none / no driver) */
int         commet_device_count(void);

/* ---- context: k, t, the Bloom filter in HBM ------------------------------ */
/* Replaces `HashKey hash(kmer_size)` + `new BloomFilter(kmer_size)`
 * (hash_key.h:40-49, bloom_filter.h:61-81): binds a device, creates a stream,
 * allocates the 4-lane filter (2^(k-1) bytes of HBM, laid out as four bit-planes
 * of 2^k bits).  1 <= k <= 38; t < 1 behaves as t = 1 (search_reads.h:55).
 * Returns NULL on error. */
commet_ctx *commet_create(int device, int kmer_size, int min_hits);
void        commet_destroy(commet_ctx *ctx);
int         commet_kmer_size(const commet_ctx *ctx);
int         commet_min_hits(const commet_ctx *ctx);
/* max_kmer = (unsigned long)(1e9 / 2^(33-k))  (src/index_and_search.cpp:73,146) */
uint64_t    commet_max_kmer(const commet_ctx *ctx);
/* free / total memory of the context's device (a resident server decides from it which sets to keep) */
int         commet_device_memory(const commet_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes);
/* blocks until everything queued on the ctx's stream has finished */
int         commet_synchronize(commet_ctx *ctx);

/* ---- read sets resident in HBM ------------------------------------------- */
/* Replaces FileManager + ReadFile as the *source of reads* (file_manager.h:38-49,
 * read_file.h:34-47): a read set is the virtual concatenation of its files;
 * reads are 2-bit packed on the device into bit-planes (+ a validity plane for
 * non-ACGT, alphabet.h:44-58).  max_reads / max_bases are capacity bounds
 * (e.g. #'>' lines and file size). */
commet_readset *commet_readset_create(commet_ctx *ctx, uint64_t max_reads, uint64_t max_bases);
void            commet_readset_destroy(commet_readset *rs);
/* Marks the start of a new file of the set (FileManager::addFile,
 * file_manager.h:117-171).  Must be called before the first append of each file. */
int             commet_readset_begin_file(commet_readset *rs);
/* Pinned staging, double-buffered.  acquire() hands out a pinned host buffer
 * (waiting for the copy that last used it); the parser writes sequence bytes
 * (ASCII, no separators) into *bases and read boundaries into *offsets
 * (offsets[0] = 0 ... offsets[n_reads] = bytes used); commit() queues the
 * hipMemcpyAsync + the packing kernel and returns at once. */
int             commet_readset_stage_acquire(commet_readset *rs, uint8_t **bases, uint64_t *bases_cap,
                                             uint64_t **offsets, uint64_t *reads_cap);
int             commet_readset_stage_commit(commet_readset *rs, uint64_t n_reads);
/* Convenience: copies (bases, offsets[n_reads+1]) through the staging buffers. */
int             commet_readset_append(commet_readset *rs, const uint8_t *bases, const uint64_t *offsets,
                                      uint64_t n_reads);
/* Host ingest of a whole set: opens the read files (FASTA or FASTQ, plain or gzip,
 * sniffed like file_manager.h:125-157; one per file of the set, in order), counts
 * their records, creates the read set and streams every record through the pinned
 * staging buffers (records as the reference reads them, fasta_file.h:155-175,
 * fastq_file.h:139-190).  Replaces FileManager::addFile + the ReadFile parsers for
 * resident sets (file_manager.h:117-171).  The set is NOT finalized.  NULL on error. */
commet_readset *commet_readset_from_fasta(commet_ctx *ctx, const char *const *paths, int n_paths);
/* Same, from file contents already in memory (plain FASTA / FASTQ text, one buffer per file of the set;
 * the buffers must stay valid until the call returns).  Records are parsed by several host threads
 * (COMMET_INGEST_THREADS, default 8) and uploaded out of order at their final positions. */
commet_readset *commet_readset_from_buffers(commet_ctx *ctx, const char *const *data, const uint64_t *sizes, int n_files);
/* reads of file `file_index` of the set (all records, selected or not) */
uint64_t        commet_readset_file_reads(const commet_readset *rs, uint64_t file_index);
/* Ends loading: waits for the uploads, fetches the per-read counts of complete
 * k-mers (what index_reads.h:55-57 would feed) needed for exact chunking. */
int             commet_readset_finalize(commet_readset *rs);
/* Packed image of a finalized read set (the bit-planes as they lie in HBM + file spans; independent of k): written
 * once by whoever parsed the files, loaded by every other context that needs the set — other GPUs of the node in the
 * N x N driver (one parse per set instead of one per GPU), or a later process.  load returns a set that still has to
 * be finalized; its per-read k-mer counts are recomputed on the device for the loading context's k.  The reference
 * has no counterpart: every index_and_search process re-parses its files (file_manager.h:117-171). */
int             commet_readset_save(const commet_readset *rs, const char *path);
commet_readset *commet_readset_load(commet_ctx *ctx, const char *path);
/* The same hand-over WITHOUT a file, for the processes of one node: export writes a descriptor (file spans, lengths, and the
 * HIP IPC handles of the set's device buffers — dmabuf mode, HSA_ENABLE_IPC_MODE_LEGACY=0) into blob (blob == NULL: only
 * *blob_bytes, the size needed); import, in another process on any device of the node that reaches the owner's (xGMI), or
 * on the same device, opens the handles, copies the planes device to device into a set of its own (2.4 GB of a 50 M-read
 * set: a few tens of ms, against 0.5 s to write and 0.2 s to read the image through /dev/shm) and closes them.  The
 * imported set still has to be finalized.  The OWNER'S SET MUST STAY ALIVE until every importer has returned. */
int             commet_readset_export(const commet_readset *rs, void *blob, uint64_t cap, uint64_t *blob_bytes);
commet_readset *commet_readset_import(commet_ctx *ctx, const void *blob, uint64_t blob_bytes);
uint64_t        commet_readset_num_reads(const commet_readset *rs);
uint64_t        commet_readset_num_files(const commet_readset *rs);
/* kmers_out[n_reads]: complete k-mers of each read (valid after finalize) */
int             commet_readset_kmer_counts(const commet_readset *rs, uint32_t *kmers_out);
/* Derived data cached WITH a resident set: the query list of the tiled search (the set's lane-a addresses sorted by
 * address slice; depends on (k, t) and the set only, made on the set's first scan, ~6 bytes per first-hit window — 2.2 GB
 * for 10 M x 100 bp reads at k = 32, t = 2, against 0.5 GB for the packed set).  The lists of a context are held to a
 * budget (option "query_list_budget_mb", COMMET_QUERY_LIST_GB; default: half the device's memory, at least 64 GiB): the least recently used ones are given
 * back first, and all of them (but the running job's) when a device allocation of the context fails, which is then
 * tried again.  cache_bytes: HBM the set's list holds now; drop_cache: give it back (rebuilt on the next scan that
 * wants it; no-op while a job uses the set); cache_stats: the context's totals.  No counterpart in the reference. */
uint64_t        commet_readset_cache_bytes(const commet_readset *rs);
void            commet_readset_drop_cache(commet_readset *rs);
int             commet_cache_stats(commet_ctx *ctx, uint64_t *bytes, uint64_t *budget_bytes, uint64_t *evictions);
/* Device memory the library keeps for reuse.  On this driver a hipMalloc of GBs costs 15-30 ms per GiB and now and then blocks
 * for 50-150 ms behind the hipFree of tens of GB, so blocks of 8 MiB or more that a context, a read set or a cache gives up are
 * filed per device (at most half the device, COMMET_DEVMEM_CACHE_GB) and handed to the next allocation they fit; they go back to
 * the driver when an allocation is out of memory, or here.  commet_device_cache_trim(device; -1 = every device) returns the bytes
 * released, commet_device_cache_bytes(device) what is filed now.  COMMET_DEVMEM_CACHE=0 turns the mechanism off.  Nothing in the
 * reference corresponds to it (its filters are plain `new char[]`, bloom_filter.h:73). */
uint64_t        commet_device_cache_trim(int device);
/* A query list above the cap (option "query_list_max_mb": 4 GiB, sets of ~15 M reads) costs its first user 15-30 ms per GiB of
 * driver time on a box whose device memory has not been touched yet — on the job's path.  A driver that knows a set will be scanned
 * again and again (the N x N matrix: a 50 M-read set's list is 11 GB and saves 12 ms per scan) calls commet_readset_reserve_cache
 * from a helper thread while its jobs run: the blocks the list will need (for the context's k and t) are asked from the driver there
 * and filed in the device cache; from then on the set may get its list whatever the cap says (built, as every list above 4 GiB,
 * for its second eligible scan).  _estimate returns the bytes such a list takes (0: the set does not qualify for the tiled search).
 * No-ops when the list exists, when the device cache is off, or when there is no room. */
uint64_t        commet_readset_cache_estimate(commet_ctx *ctx, const commet_readset *rs);
int             commet_readset_reserve_cache(commet_ctx *ctx, const commet_readset *rs);
uint64_t        commet_device_cache_bytes(int device);
/* COMMET_DEVMEM_POOL=1 (off by default): new blocks of 256 MiB or more come from the driver's stream-ordered pool (hipMallocAsync
 * on a stream of the library's own, drained before the block is used): 0.4-0.9 ms per GiB on every box, where hipMalloc takes
 * 30-60 ms per GiB on some and nothing on others (tools/exp/alloc_cost*.hip) — but kernels gather ~4 % more slowly from pool
 * memory and on a box of the second kind the 10-set matrix lost 1.8 s of 10.9 s with it (profiles/r05_pool), hence off.  No HIP
 * IPC handle exists for such memory: commet_readset_export moves the set's buffers into hipMalloc blocks first (once per set, a
 * device-to-device copy), and a caller that shares library memory by other means checks here.
 * commet_device_pooled_bytes(device): bytes of such blocks in use now. */
uint64_t        commet_device_pooled_bytes(int device);
/* What the library asked the DRIVER for on `device` (-1: every device) since the process started: host time its threads spent
 * inside hipMalloc / hipMallocAsync (ms), the bytes they returned and the number of calls — blocks taken from the library's own
 * cache do not count.  A box that charges a process for its first use of device memory (15-30 ms per GiB) shows
 * HERE and not in any kernel's time; the N x N driver reports the difference over a matrix leg as alloc_wait_ms / fresh_device_bytes.
 * Nothing in the reference corresponds to it. */
int             commet_device_alloc_stats(int device, double *wait_ms, uint64_t *fresh_bytes, uint64_t *calls);

/* ---- the two kernels ------------------------------------------------------ */
/* Replaces `new BloomFilter` per chunk (index_and_search.cpp:256-262,
 * bloom_filter.h:73-76): zeroes the filter (asynchronous). */
int commet_filter_reset(commet_ctx *ctx);

/* Replaces the body of index_reads (index_reads.h:49-61) for reads
 * [first, first+count) of rs whose select bit is 1 (select_bits indexed by the
 * set-wide read number, NULL = all): every complete k-mer sets its 4 lane bits
 * (bloom_filter.h:112-118).  Chunking (max_kmer, the dropped look-ahead read)
 * is decided by the caller / commet_index_and_search.  *kmers_fed (optional)
 * receives the number of k-mers fed; asking for it synchronises. */
int commet_index_reads(commet_ctx *ctx, const commet_readset *rs, uint64_t first, uint64_t count,
                       const uint8_t *select_bits, uint64_t *kmers_fed);

/* Replaces search_reads (search_reads.h:34-87) for the reads of rs whose
 * active bit is 1 (NULL = all): a read is found iff it has >= t non-overlapping
 * k-mers present in all 4 lanes, scanning forward keys first and
 * reverse-complement keys only if the forward scan failed.  found_bits
 * (n/8+1 bytes, caller-allocated) gets 1 for found reads and 0 for all others.
 * n_scanned = active reads, n_found = found reads (both optional). */
int commet_search_reads(commet_ctx *ctx, const commet_readset *rs, const uint8_t *active_bits,
                        uint8_t *found_bits, uint64_t *n_scanned, uint64_t *n_found);

/* ---- the chunk loop on resident sets --------------------------------------- */
typedef struct {
    uint64_t indexed;       /* nb_indexed_reads  (index_and_search.cpp:286: excludes dropped reads) */
    uint64_t searched;      /* nb_searched_reads of the LAST search pass (search_reads.h:39)        */
    uint64_t shared;        /* nb_found_reads summed over chunks                                    */
    double   search_ms;     /* device time of this set's search kernels (search_times[set_pos], :272) */
} commet_pair_stats;

typedef struct {
    uint64_t n_chunks;          /* filters built                          */
    uint64_t kmers_indexed;     /* k-mers fed over all chunks             */
    uint64_t reads_scanned;     /* search-read scans over all chunks/sets */
    uint64_t reads_indexed;     /* reads fed to a filter (dropped look-ahead reads excluded) */
    uint64_t index_launches;    /* index kernel launches                  */
    uint64_t search_launches;   /* search kernel launches                 */
    uint64_t probes;            /* filter words loaded by the search kernels = P_ref of the reference's
                                   control flow; 0 unless option "count_probes" is set */
    double   zero_ms;           /* device time: filter zeroing (hipEvents on the ctx stream) */
    double   index_ms;          /* device time: filter zeroing + index kernels */
    double   index_kernel_ms;   /* device time: index kernels only        */
    double   search_ms;         /* device time: search kernels            */
    double   total_ms;          /* host wall time of the call             */
} commet_job_info;

/* Replaces the while loop of main() (index_and_search.cpp:241-277) together
 * with the FileManager iteration rules it relies on (file_manager.h:88-112:
 * input-filter bits, skipping of already tagged reads, file switching;
 * index_reads.h:49,60: the look-ahead read that is dropped when a chunk fills):
 *   while reads remain in index_rs: build the filter of the next chunk, search
 *   every search set against it, OR the found bits into that set's tags.
 * index_select / search_select[i]: per-file input-filter bits concatenated over
 * the set (ReadFile::bv), NULL = all ones.  tags_out[i]: n_i/8+1 bytes, the
 * FileManager::file_bvs of search set i at the end (what save_bv writes).
 * stats[i] / info are optional. */
int commet_index_and_search(commet_ctx *ctx,
                            const commet_readset *index_rs, const uint8_t *index_select,
                            int n_search, const commet_readset *const *search_rs,
                            const uint8_t *const *search_select,
                            uint8_t *const *tags_out, commet_pair_stats *stats,
                            commet_job_info *info);
/* Several such jobs that search the SAME read set — Commet.py's J2 jobs of a reference set (Commet.py:220: for every other set S_i,
 * "S_ref in (S_i restricted to J1's result)") and its J3 jobs of a target (Commet.py:233) — in one call: job j indexes index_rs[j]
 * (restricted to index_select[j], may be NULL) and searches search_rs (search_select as above); tags_out[j], stats[j] are what
 * commet_index_and_search(index_rs[j], ..., 1, &search_rs, ...) gives for job j alone, bit for bit.  Where the jobs allow it
 * (index sets whose chunks, at most eight per job, take the bucketed construction; a search set that is visited
 * whole) the chunk filters of several jobs share a pass over the search set: the lane-a gathers of its reads, two thirds of a
 * job's memory requests, are then made once per pass instead of once per job: up to eight chunk filters per pass of the gather
 * kernel; on a search set that takes the tiled search, jobs of one chunk filter each two per scan (one probe of the set's query list,
 * one replay that keeps the two jobs apart).  Otherwise (and with option "multi_job" = 1, or when the device has no room for the
 * slots of a shared pass) the jobs run one after the other.  info (may be NULL) sums over the jobs. */
int commet_index_many_and_search(commet_ctx *ctx, int n_jobs, const commet_readset *const *index_rs,
                                 const uint8_t *const *index_select, const commet_readset *search_rs,
                                 const uint8_t *search_select, uint8_t *const *tags_out, commet_pair_stats *stats,
                                 commet_job_info *info);

/* ---- test / measurement hooks --------------------------------------------- */
/* Tunables / diagnostics, by name.  Unknown names are an error.  None of them changes a result bit, except the test
 * hook max_kmer.
 *   count_probes (0/1)   the search kernels count the filter words the REFERENCE flow loads (P_ref)
 *   index_mode (0/1/2)   0 auto, 1 atomic-OR kernel, 2 bucketed (LDS-tile) construction
 *   part_min_kmers       auto mode: chunks with fewer k-mers take the atomic kernel
 *   index_lanes (1/2)    2 = the chunks of a group are built on two streams (default)
 *   lane_stagger (0/1)   two lanes: the second lane's chunk starts when the first lane's scatter1 is through (default 1), so that
 *                        its VALU-bound phases run beside the first chunk's HBM-bound ones; 0 = both chunks start together
 *   drop_workspaces      frees the scatter workspaces (the next bucketed index build allocates them again)
 *   chunk_group (1..8)   chunk filters searched per pass over a set (1 = the reference's order; 5..8 only
 *                        for read sets with at most 255 first-hit windows per read — reads of up to 318 bases at k = 32, t = 2 —, else 4)
 *   tiled_search (0/1/2) large search sets against 1 or 2 chunk filters (25 <= k <= 34): lane-a gathers served from L2 slice
 *                        by slice from the set's cached query list; 0 = sets of 2^20 reads or more whose list fits 4 GiB,
 *                        1 = never, 2 = whenever possible
 *   slice_mode (0/1/2)   many-small-chunks regime (12 <= k <= 24): the filters of 32..256 chunks bit-sliced in one table
 *                        set and searched in ONE pass; 0 = from 8 chunks on, 1 = never, 2 = always
 *   slice_words          chunk filters per pass / 32 in that regime (0 auto, 1, 2, 4, 8)
 *   slice_wide (0/1/2)   that regime with EVERY chunk filter (up to 16 384 per pass) side by side in rows of one table and a
 *                        group of lanes per read (search_wide_kernel): 0 = jobs of more than 256 chunks, after a probe of the
 *                        first 64 chunk filters on a sample of the reads (few reads found early: wide rows; most: the narrow
 *                        tables, whose later passes skip the reads already found), 1 = never, 2 = always
 *   slice_wide_words     cap on the words per wide row (a multiple of 8, 32 chunk filters per word; 0 = by the memory free)
 *   query_list_budget_mb HBM the cached query lists of the context's read sets may hold (see commet_readset_cache_bytes)
 *   query_list_max_mb    auto mode of tiled_search: largest list (estimated) a set may get, default 4096 (sets of up to ~15 M reads;
 *                        larger lists — a 50 M-read set's is 11 GB — pay in long-lived contexts only: allocating them costs
 *                        15-30 ms per GiB; lists of more than 4 GiB are built for a set's second eligible scan)
 *   multi_job (0/1)      commet_index_many_and_search: 0 = chunk filters of several jobs in one pass where possible, 1 = job by job
 *   sparse_search (0/1/2) a pass over a SELECTION of a search set (a filter bv that leaves few reads: file_manager.h:88-112 skips the
 *                        others) walks the list of the selected, not yet tagged reads instead of the set's bitmap, so that every
 *                        lane of a wave has a read: 0 = when the host plan visits less than half of the set's reads, 1 = never,
 *                        2 = whenever a selection applies (tests)
 *   ordered_scan (0/1/2) ragged sets (reads of several lengths): the first pass of a gather kernel over a whole set walks its reads in
 *                        order of their first-hit window counts (a list made once per set: a workgroup lives as long as its longest
 *                        read, its other lanes idle meanwhile): 0 = sets of 2^16 reads and more, 1 = never, 2 = any ragged set (tests)
 *   mask_split (0/1)     a pass of the register-mask gather kernel over that list: 0 = segment by segment, each launched with the
 *                        narrowest masks its reads fit (the list starts with the reads of most windows), 1 = one launch at the set's width
 *   tq_hit_cap           TEST HOOK of the tiled search's replay: full hits a piece of 256 reads may post for its owners (default and
 *                        at most 1024; beyond it every scan of the piece walks its own candidates — same bits, slower); 0 forces that path
 *   tq_parts (1..16)     tiled search in parts, the replay of one beside the probe of the next (default 1: measured slower)
 *   part_no_uni (0/1)    1 = never take the fixed-read-length fast path of hist / scatter1 (nor the item list of ragged sets)
 *   part_list (0/1)      ragged sets (reads of several lengths): 0 = hist / scatter1 walk the chunk's item list (written out once per
 *                        chunk; the words of the coming round's items prefetched as on fixed-length sets), 1 = the round planner
 *   part_b1, s2_swizzle  radix split / scatter2 slab order of the bucketed construction
 *   kernel_timing (0/1)  time every kernel launch of commet_index_and_search (commet_kernel_times)
 *   max_kmer             TEST HOOK: k-mers per index chunk instead of the reference's constant (0 = reference; the
 *                        results are then those of a reference built with that constant, index_and_search.cpp:73) */
int commet_set_option(commet_ctx *ctx, const char *name, int64_t value);
/* Copies the filter to the host in the REFERENCE byte layout (byte key/2,
 * even keys 0x80/40/20/10, odd keys 0x08/04/02/01 for a/b/c/d,
 * bloom_filter.h:63-70,114-117); out has 2^(k-1) bytes.  For parity tests. */
int commet_filter_export_reference(commet_ctx *ctx, uint8_t *out, uint64_t out_bytes);
/* Device time in ms of the most recent index / search kernel launch on this
 * ctx, measured with hipEvents on the ctx's stream (synchronises). */
int commet_last_kernel_ms(commet_ctx *ctx, double *index_ms, double *search_ms);
/* Per-kernel device times of the commet_index_and_search calls made since option "kernel_timing" was set to 1:
 * a hipEvent pair around every launch, on the stream the kernel runs on (the chunks of a group are then built on
 * one stream, so that the durations add up).  Fills at most cap entries, *n_out = kernels seen.  Synchronises. */
typedef struct {
    char     name[48];
    uint64_t launches;
    double   total_ms;
} commet_kernel_time;
int commet_kernel_times(commet_ctx *ctx, commet_kernel_time *out, int cap, int *n_out);
/* Entry points (host-side kernel handles, addresses inside this library) of every kernel the library has launched in
 * this process so far, whatever the context.  The test-suite resolves them against the library's symbol table to check
 * that every kernel instantiation compiled into the library is reached by a parity test.  Fills at most cap entries,
 * *n_out = how many there are. */
int commet_launched_kernels(const void **out, int cap, int *n_out);
/* Random 4-byte-gather / atomic-OR microbenchmarks over a table of
 * table_bytes (practical random-access ceilings): n_access
 * accesses, returns elapsed device ms in *ms.  atomic: 0 plain gather, 1 atomic
 * OR, 2 non-temporal gather, 3 agent-scope (L1-bypassing) gather; 4 / 5 time a
 * device-to-device copy / a fill of table_bytes instead (streaming ceilings). */
int commet_membench(commet_ctx *ctx, int atomic, uint64_t table_bytes, uint64_t n_access, double *ms);
/* LDS microbenchmark (what bounds the bucketed index construction): n_access
 * operations on uniformly random words of an n_words-word LDS table (power of
 * two, <= 32768) per workgroup of 512.  mode 0 atomic add, 1 atomic add with
 * the old value used (rank), 2 atomic OR, 3 store, 4 load, 5 as 1 but on
 * lane-private words (no two lanes share an address). */
int commet_ldsbench(commet_ctx *ctx, int mode, uint32_t n_words, uint64_t n_access, double *ms);

#ifdef __cplusplus
}
#endif
#endif /* COMMET_HIP_H_ */

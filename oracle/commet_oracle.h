/*
 * commet_oracle.h — CPU restatement of Commet's index_and_search hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under commet_amd/ or include/ may include,
 * link or call this.  Allowed users: tests/, __graft_entry__.smoke(), and the
 * cpu_baseline leg of bench.py.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_*.py)
 *   - against golden vectors produced by the compiled reference
 *     (tests/golden/, generator tests/golden/make_golden.py), and
 *   - live against oracle/_ref/index_and_search (the reference's own sources
 *     compiled in place by oracle/Makefile) whenever that binary is present.
 *
 * Every function cites the reference file:line it restates
 * (paths relative to the reference root, pierrepeterlongo/commet @ v1).
 */
#ifndef COMMET_ORACLE_H_
#define COMMET_ORACLE_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- HashKey (include/hash_key.h:31-131) -------------------------------- */
typedef struct {
    uint64_t a, b, c, d;      /* _keya.._keyd */
    uint64_t mask;            /* mask_size_kmer    = 2^k - 1     (hash_key.h:45) */
    uint64_t rv_mask;         /* rv_mask_size_kmer = 2^(k-1) - 1 (hash_key.h:47) */
    uint64_t top;             /* bloom_size        = 2^(k-1)     (hash_key.h:43) */
    int      size;            /* hash_size */
} ok_hash;

void ok_hash_init(ok_hash *h, int k);          /* hash_key.h:40-49  */
void ok_hash_clear(ok_hash *h);                /* hash_key.h:52-56  */
int  ok_hash_add(ok_hash *h, char aa);         /* hash_key.h:63-89  */
int  ok_hash_rv_add(ok_hash *h, char aa);      /* hash_key.h:97-123 */
int  ok_alpha_is_in(unsigned char c);          /* alphabet.h:44-58 (+ SURVEY Q9) */

/* Key-level KAT helper: all complete k-mers of one read.  out_* hold 4 keys
 * (a,b,c,d) per valid window, in position order; returns the window count.
 * end_pos[i] is the 0-based index of the last base of window i. */
uint64_t ok_keys_of_read(const char *seq, uint64_t len, int k, int reverse,
                         uint64_t *out_keys4, uint32_t *end_pos, uint64_t cap);

/* ---- BloomFilter (include/bloom_filter.h:45-132) ------------------------ */
typedef struct {
    uint8_t *vec;             /* bloom_vector, 2^(k-1) bytes (bloom_filter.h:73-76) */
    uint64_t nbytes;
    uint64_t probes;          /* instrumentation: byte loads done by is_found */
} ok_bloom;

ok_bloom *ok_bloom_new(int k);                          /* bloom_filter.h:61-81   */
void      ok_bloom_free(ok_bloom *f);                   /* bloom_filter.h:83-85   */
void      ok_bloom_feed(ok_bloom *f, const ok_hash *h); /* bloom_filter.h:112-118 */
int       ok_bloom_is_found(ok_bloom *f, const ok_hash *h); /* bloom_filter.h:124-131 */

/* max_kmer = (unsigned long)(1e9 / 2^(33-k))  (src/index_and_search.cpp:73,146) */
uint64_t ok_max_kmer(int k);

/* Test hook, no counterpart in the reference: k-mers per chunk of the following ok_index_and_search calls (0: the
 * reference's constant again) — the twin of the library option `max_kmer`, so that both sides chunk alike. */
void ok_set_max_kmer(uint64_t max_kmer);

/* ---- batch forms of the two kernels (what the HIP path is compared with) - */
/* Feeds every complete k-mer of the reads whose select bit is 1 (select ==
 * NULL: all reads) — the body of the while loop of index_reads.h:49-61 without
 * the max_kmer stop (chunking is host-level).  Returns k-mers fed. */
uint64_t ok_index_batch(ok_bloom *f, int k, const uint8_t *bases,
                        const uint64_t *offsets, uint64_t n_reads,
                        const uint8_t *select_bits);

/* search_reads.h:45-83 applied to each read whose active bit is 1 (NULL: all).
 * Sets found_bits (LSB-first, BooleanVector order boolean_vector.h:73-80) for
 * the reads found; bits of inactive reads are left untouched.  Returns the
 * number found.  f->probes accumulates P_ref. */
uint64_t ok_search_batch(ok_bloom *f, int k, int t, const uint8_t *bases,
                         const uint64_t *offsets, uint64_t n_reads,
                         const uint8_t *active_bits, uint8_t *found_bits);

/* ---- BooleanVector file I/O (include/boolean_vector.h:302-414) ---------- */
/* Writes "<comment>\n#<n>\n" + n/8+1 raw bytes, mode 0600, truncating. */
int ok_bv_write(const char *path, const char *comment, const uint8_t *bits, uint64_t n);
/* Reads a .bv; *bits is malloc'd (n/8+1 bytes). Returns 0 on success. */
int ok_bv_read(const char *path, uint8_t **bits, uint64_t *n);
uint64_t ok_bv_nb_one(const uint8_t *bits, uint64_t n);      /* boolean_vector.h:236-264 */

/* ---- whole tool (src/index_and_search.cpp:56-401, non -f mode) ---------- */
typedef struct {
    char     search_name[256];
    uint64_t indexed, searched, shared;  /* the "[indexed X, searched Y, shared Z]" line */
    uint64_t probes;                     /* P_ref over all search passes of this set */
} ok_set_result;

/* Runs the reference algorithm end to end: parses the two set-configs
 * (set_parser.h:46-102), opens FASTA files (+ optional filter .bv), runs the
 * chunk loop, writes OUT/<basename>_in_<index>.bv and LOG/<search>_in_<index>.log
 * (timing lines are written as 0 s; only the 4th line is comparable).
 * results: caller array of cap entries; *n_results = number of search sets.
 * quiet != 0 suppresses the stdout banners.  Returns the tool's exit code. */
int ok_index_and_search(const char *index_cfg, const char *search_cfg,
                        const char *out_dir, const char *log_dir, int k, int t,
                        ok_set_result *results, int cap, int *n_results,
                        uint64_t *n_chunks, uint64_t *kmers_indexed, int quiet);

/* Chunk trace of the next ok_index_and_search call: 4 values per chunk
 * (first, last set-wide read number, reads, k-mers).  ok_trace_end returns the
 * number of chunks seen. */
void     ok_trace_begin(uint64_t *buf, uint64_t cap_chunks);
uint64_t ok_trace_end(void);

#ifdef __cplusplus
}
#endif
#endif

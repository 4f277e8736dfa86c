/*
 * commet_oracle.c — CPU restatement of Commet's index_and_search hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see commet_oracle.h).  Plain C, single-threaded,
 * same data structures and control flow as the reference so that probe counts,
 * chunk boundaries, false positives and the log line all come out identical.
 * Parity status: PINNED (golden vectors from the compiled reference +
 * live comparison with oracle/_ref/index_and_search).
 */
#define _GNU_SOURCE
#include "commet_oracle.h"

#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>
#include <zlib.h>

/* ======================================================================== */
/* HashKey — include/hash_key.h                                             */
/* ======================================================================== */

void ok_hash_init(ok_hash *h, int k)
{
    /* hash_key.h:43  bloom_size = (unsigned long) pow(2, kmer_size - 1) */
    h->top = (uint64_t) pow(2, k - 1);
    h->mask = (2 * h->top) - 1;      /* hash_key.h:45 */
    h->rv_mask = h->top - 1;         /* hash_key.h:47 */
    ok_hash_clear(h);
}

void ok_hash_clear(ok_hash *h)
{
    h->size = 0;                     /* hash_key.h:54-55 */
    h->a = h->b = h->c = h->d = 0;
}

int ok_hash_add(ok_hash *h, char aa)
{
    /* hash_key.h:65-89: shift left, mask to k bits, OR 1 per base class */
    h->size++;
    h->a = (h->a << 1) & h->mask;
    h->b = (h->b << 1) & h->mask;
    h->c = (h->c << 1) & h->mask;
    h->d = (h->d << 1) & h->mask;
    if (aa == 'C' || aa == 'c') {
        h->b |= 1; h->c |= 1; h->d |= 1;
    } else if (aa == 'G' || aa == 'g') {
        h->a |= 1; h->c |= 1; h->d |= 1;
    } else if (aa == 'T' || aa == 't') {
        h->a |= 1; h->b |= 1; h->d |= 1;
    }
    return h->size;
}

int ok_hash_rv_add(ok_hash *h, char aa)
{
    /* hash_key.h:99-123: shift right, mask to k-1 bits, OR top bit with the
     * COMPLEMENT base's classes (A->T: a,b,d; C->G: a,c,d; G->C: b,c,d; T->A: -) */
    h->size++;
    h->a = (h->a >> 1) & h->rv_mask;
    h->b = (h->b >> 1) & h->rv_mask;
    h->c = (h->c >> 1) & h->rv_mask;
    h->d = (h->d >> 1) & h->rv_mask;
    if (aa == 'A' || aa == 'a') {
        h->a |= h->top; h->b |= h->top; h->d |= h->top;
    } else if (aa == 'C' || aa == 'c') {
        h->a |= h->top; h->c |= h->top; h->d |= h->top;
    } else if (aa == 'G' || aa == 'g') {
        h->b |= h->top; h->c |= h->top; h->d |= h->top;
    }
    return h->size;
}

int ok_alpha_is_in(unsigned char c)
{
    /* alphabet.h:44-58.  The reference indexes a char[255] with a (possibly
     * negative) char: bytes >= 0x80 are UB there; SURVEY Q9 fixes them as
     * "not in alphabet". */
    switch (c) {
    case 'A': case 'a': case 'C': case 'c':
    case 'G': case 'g': case 'T': case 't':
        return 1;
    default:
        return 0;
    }
}

uint64_t ok_keys_of_read(const char *seq, uint64_t len, int k, int reverse,
                         uint64_t *out_keys4, uint32_t *end_pos, uint64_t cap)
{
    ok_hash h;
    uint64_t n = 0;
    ok_hash_init(&h, k);
    for (uint64_t i = 0; i < len; i++) {
        if (!ok_alpha_is_in((unsigned char) seq[i])) {
            ok_hash_clear(&h);
            continue;
        }
        int sz = reverse ? ok_hash_rv_add(&h, seq[i]) : ok_hash_add(&h, seq[i]);
        if (sz >= k) {
            if (n < cap) {
                out_keys4[4 * n + 0] = h.a;
                out_keys4[4 * n + 1] = h.b;
                out_keys4[4 * n + 2] = h.c;
                out_keys4[4 * n + 3] = h.d;
                end_pos[n] = (uint32_t) i;
            }
            n++;
        }
    }
    return n;
}

/* ======================================================================== */
/* BloomFilter — include/bloom_filter.h                                     */
/* ======================================================================== */

/* bloom_filter.h:63-70 */
#define MASK_A_EVEN 128
#define MASK_B_EVEN 64
#define MASK_C_EVEN 32
#define MASK_D_EVEN 16
#define MASK_A_ODD  8
#define MASK_B_ODD  4
#define MASK_C_ODD  2
#define MASK_D_ODD  1

ok_bloom *ok_bloom_new(int k)
{
    ok_bloom *f = (ok_bloom *) calloc(1, sizeof(ok_bloom));
    if (!f) return NULL;
    f->nbytes = (uint64_t) pow(2, k - 1);          /* bloom_filter.h:73 */
    f->vec = (uint8_t *) calloc(f->nbytes ? f->nbytes : 1, 1);  /* :76 */
    if (!f->vec) {
        fprintf(stderr, "Index memory allocation impossible, try with a lower k value or with more RAM memory\n");
        free(f);
        return NULL;
    }
    return f;
}

void ok_bloom_free(ok_bloom *f)
{
    if (!f) return;
    free(f->vec);
    free(f);
}

void ok_bloom_feed(ok_bloom *f, const ok_hash *h)
{
    /* bloom_filter.h:114-117 */
    f->vec[h->a / 2] |= (h->a % 2 ? MASK_A_ODD : MASK_A_EVEN);
    f->vec[h->b / 2] |= (h->b % 2 ? MASK_B_ODD : MASK_B_EVEN);
    f->vec[h->c / 2] |= (h->c % 2 ? MASK_C_ODD : MASK_C_EVEN);
    f->vec[h->d / 2] |= (h->d % 2 ? MASK_D_ODD : MASK_D_EVEN);
}

int ok_bloom_is_found(ok_bloom *f, const ok_hash *h)
{
    /* bloom_filter.h:126-130, short-circuit a -> b -> c -> d; one byte load
     * per evaluated term (P_ref of SURVEY §8d). */
    f->probes++;
    if (!(f->vec[h->a / 2] & (h->a % 2 ? MASK_A_ODD : MASK_A_EVEN))) return 0;
    f->probes++;
    if (!(f->vec[h->b / 2] & (h->b % 2 ? MASK_B_ODD : MASK_B_EVEN))) return 0;
    f->probes++;
    if (!(f->vec[h->c / 2] & (h->c % 2 ? MASK_C_ODD : MASK_C_EVEN))) return 0;
    f->probes++;
    if (!(f->vec[h->d / 2] & (h->d % 2 ? MASK_D_ODD : MASK_D_EVEN))) return 0;
    return 1;
}

uint64_t ok_max_kmer(int k)
{
    /* src/index_and_search.cpp:73,146 */
    return (uint64_t) (1000000000.0 / pow(2, 33 - k));
}

/* Test hook (no counterpart in the reference): the chunk size of the next ok_index_and_search calls, so that the
 * checker can be chunked like a library context whose `max_kmer` option is set (many chunks from small sets). */
static uint64_t g_max_kmer_override = 0;
void ok_set_max_kmer(uint64_t max_kmer) { g_max_kmer_override = max_kmer; }

/* ======================================================================== */
/* per-read bodies of index_reads / search_reads                            */
/* ======================================================================== */

/* index_reads.h:51-59: returns k-mers fed for this read */
static uint64_t index_one_read(ok_bloom *f, ok_hash *h, int k, const char *seq, uint64_t len)
{
    uint64_t fed = 0;
    ok_hash_clear(h);
    for (int i = 0; i < (int) len; i++) {
        if (!ok_alpha_is_in((unsigned char) seq[i])) {
            ok_hash_clear(h);
        } else if (ok_hash_add(h, seq[i]) >= k) {
            ok_bloom_feed(f, h);
            fed++;
        }
    }
    return fed;
}

/* search_reads.h:45-83: returns 1 when the read is found */
static int search_one_read(ok_bloom *f, ok_hash *h, int k, int t, const char *seq, uint64_t len)
{
    int seen = 0;
    int found = 0;
    ok_hash_clear(h);
    for (long i = 0; i < (int) len && !found; i++) {
        if (!ok_alpha_is_in((unsigned char) seq[i])) {
            ok_hash_clear(h);
        } else if (ok_hash_add(h, seq[i]) >= k) {
            if (ok_bloom_is_found(f, h)) {
                seen++;
                if (seen >= t) found = 1;
                ok_hash_clear(h);            /* search_reads.h:60 */
            }
        }
    }
    if (!found) {                            /* search_reads.h:65-83 */
        seen = 0;
        ok_hash_clear(h);
        for (long i = 0; i < (int) len && !found; i++) {
            if (!ok_alpha_is_in((unsigned char) seq[i])) {
                ok_hash_clear(h);
            } else if (ok_hash_rv_add(h, seq[i]) >= k) {
                if (ok_bloom_is_found(f, h)) {
                    seen++;
                    if (seen >= t) found = 1;
                    ok_hash_clear(h);
                }
            }
        }
    }
    return found;
}

static inline int bit_get(const uint8_t *bits, uint64_t i)
{
    return (bits[i >> 3] >> (i & 7)) & 1;    /* boolean_vector.h:222-225 */
}

static inline void bit_set(uint8_t *bits, uint64_t i)
{
    bits[i >> 3] |= (uint8_t) (1u << (i & 7));   /* boolean_vector.h:230-233 */
}

uint64_t ok_index_batch(ok_bloom *f, int k, const uint8_t *bases,
                        const uint64_t *offsets, uint64_t n_reads,
                        const uint8_t *select_bits)
{
    ok_hash h;
    uint64_t fed = 0;
    ok_hash_init(&h, k);
    for (uint64_t r = 0; r < n_reads; r++) {
        if (select_bits && !bit_get(select_bits, r)) continue;
        fed += index_one_read(f, &h, k, (const char *) bases + offsets[r], offsets[r + 1] - offsets[r]);
    }
    return fed;
}

uint64_t ok_search_batch(ok_bloom *f, int k, int t, const uint8_t *bases,
                         const uint64_t *offsets, uint64_t n_reads,
                         const uint8_t *active_bits, uint8_t *found_bits)
{
    ok_hash h;
    uint64_t nfound = 0;
    ok_hash_init(&h, k);
    for (uint64_t r = 0; r < n_reads; r++) {
        if (active_bits && !bit_get(active_bits, r)) continue;
        if (search_one_read(f, &h, k, t, (const char *) bases + offsets[r], offsets[r + 1] - offsets[r])) {
            bit_set(found_bits, r);
            nfound++;
        }
    }
    return nfound;
}

/* ======================================================================== */
/* BooleanVector I/O — include/boolean_vector.h                             */
/* ======================================================================== */

int ok_bv_write(const char *path, const char *comment, const uint8_t *bits, uint64_t n)
{
    /* boolean_vector.h:302-346: header "comment\n#N\n", then N/8+1 raw bytes,
     * O_CREAT|O_TRUNC, mode 0600 */
    char hdr[64];
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, (mode_t) 0600);
    if (fd == -1) {
        fprintf(stderr, "Error opening file %s -> exit\n", path);
        return 1;
    }
    snprintf(hdr, sizeof hdr, "\n#%lu\n", (unsigned long) n);
    uint64_t nbytes = n / 8 + 1;
    int ok = 1;
    ok &= write(fd, comment, strlen(comment)) == (ssize_t) strlen(comment);
    ok &= write(fd, hdr, strlen(hdr)) == (ssize_t) strlen(hdr);
    ok &= write(fd, bits, nbytes) == (ssize_t) nbytes;
    close(fd);
    return ok ? 0 : 1;
}

int ok_bv_read(const char *path, uint8_t **bits, uint64_t *n)
{
    /* boolean_vector.h:353-414: comment = bytes before the first '#', size =
     * atoi(text up to '\n'), then size/8+1 raw bytes */
    FILE *fp = fopen(path, "rb");
    if (!fp) {
        fprintf(stderr, "Error opening file %s -> exit\n", path);
        return 1;
    }
    fseek(fp, 0, SEEK_END);
    long fsz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    char *buf = (char *) malloc(fsz > 0 ? fsz : 1);
    if (fread(buf, 1, fsz, fp) != (size_t) fsz) { fclose(fp); free(buf); return 1; }
    fclose(fp);
    long i = 0;
    while (i < fsz && buf[i] != '#') i++;
    i++;
    char num[64];
    int nn = 0;
    while (i < fsz && buf[i] != '\n') {
        if (nn < 63) num[nn++] = buf[i];
        i++;
    }
    num[nn] = 0;
    i++;
    if (nn == 0) {
        fprintf(stderr, "Error, boolean vector does not contain its size\n");
        free(buf);
        return 1;
    }
    uint64_t size = (uint64_t) atoi(num);          /* boolean_vector.h:398 */
    uint64_t nbytes = size / 8 + 1;
    uint8_t *out = (uint8_t *) calloc(nbytes, 1);
    uint64_t avail = (i < fsz) ? (uint64_t) (fsz - i) : 0;
    memcpy(out, buf + i, avail < nbytes ? avail : nbytes);
    free(buf);
    *bits = out;
    *n = size;
    return 0;
}

uint64_t ok_bv_nb_one(const uint8_t *bits, uint64_t n)
{
    /* boolean_vector.h:236-264: popcount of all n/8+1 bytes, capped at n */
    uint64_t res = 0;
    for (uint64_t i = 0; i < n / 8 + 1; i++) res += (uint64_t) __builtin_popcount(bits[i]);
    if (res > n) res = n;
    return res;
}

/* ======================================================================== */
/* FastaFile — include/fasta_file.h:38-314 (records parsed up front, the    */
/* pull iterator keeps the reference's pos / first_read / cnt bookkeeping)  */
/* ======================================================================== */

typedef struct {
    char     *fname;
    uint64_t  nb_reads;      /* lines whose first char is '>' (fasta_file.h:61-68) */
    char     *seqs;          /* concatenated sequences */
    uint64_t *seq_off;       /* nb_reads + 1 */
    uint8_t  *bv;            /* ReadFile::bv, the input filter */
    uint64_t  nb_valid;      /* _nb_valid_reads = bv.nb_one() */
    uint64_t  cnt_valid;     /* _cnt_valid_reads */
    uint64_t  pos;           /* current_read_pos */
    int       first_read;
    uint8_t  *tags;          /* FileManager::file_bvs[i] (output bv) */
} ok_file;

static void ok_file_rewind(ok_file *f)
{
    /* fasta_file.h:258-264 */
    f->cnt_valid = 0;
    f->pos = 0;
    f->first_read = 1;
}

/* whole file in memory; gzip (or anything zlib passes through) when the first
 * byte is neither '>' nor '@' (file_manager.h:125-157) */
static char *slurp(const char *path, long *size_out)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return NULL;
    fseek(fp, 0, SEEK_END);
    long fsz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    char *buf = (char *) malloc((size_t) fsz + 1);
    if (fread(buf, 1, (size_t) fsz, fp) != (size_t) fsz) { fclose(fp); free(buf); return NULL; }
    fclose(fp);
    buf[fsz] = 0;
    if (fsz > 0 && (buf[0] == '>' || buf[0] == '@')) {
        *size_out = fsz;
        return buf;
    }
    free(buf);
    gzFile g = gzopen(path, "r");
    if (!g) return NULL;
    size_t cap = 1 << 20, n = 0;
    buf = (char *) malloc(cap + 1);
    for (;;) {
        if (cap - n < (1 << 16)) {
            cap *= 2;
            buf = (char *) realloc(buf, cap + 1);
        }
        int got = gzread(g, buf + n, (unsigned) (cap - n));
        if (got <= 0) break;
        n += (size_t) got;
    }
    gzclose(g);
    buf[n] = 0;
    *size_out = (long) n;
    return buf;
}

/* FASTQ records (fastq_file.h:60-67, 139-190): #records = non-empty lines / 4;
 * header (blank lines before it skipped), the next line verbatim = sequence,
 * '+' line and quality line (blank lines before each skipped). */
static int ok_file_parse_fastq(ok_file *f, char *buf, long fsz)
{
    uint64_t lines = 0;
    for (long i = 0; i < fsz;) {
        char *nl = (char *) memchr(buf + i, '\n', (size_t) (fsz - i));
        long e = nl ? (nl - buf) : fsz;
        if (e > i) lines++;
        i = nl ? e + 1 : fsz;
    }
    uint64_t n = lines / 4;
    f->nb_reads = n;
    f->seqs = (char *) malloc((size_t) fsz + 1);
    f->seq_off = (uint64_t *) calloc(n + 1, sizeof(uint64_t));
    uint64_t w = 0;
    long i = 0;
    for (uint64_t r = 0; r < n; r++) {
        long b, e;
        int ok;
#define NEXT_LINE() (i < fsz ? (b = i, e = (memchr(buf + i, '\n', (size_t) (fsz - i)) ? (char *) memchr(buf + i, '\n', (size_t) (fsz - i)) - buf : fsz), i = (e < fsz ? e + 1 : fsz), 1) : 0)
        while ((ok = NEXT_LINE()) && e == b) {}
        f->seq_off[r] = w;
        if (!ok) { for (uint64_t q = r; q <= n; q++) f->seq_off[q] = w; return 0; }
        if (NEXT_LINE()) {
            memcpy(f->seqs + w, buf + b, (size_t) (e - b));
            w += (uint64_t) (e - b);
        }
        while ((ok = NEXT_LINE()) && e == b) {}
        if (ok) while ((ok = NEXT_LINE()) && e == b) {}
#undef NEXT_LINE
    }
    f->seq_off[n] = w;
    return 0;
}

static int ok_file_parse(ok_file *f, const char *path)
{
    long fsz = 0;
    char *buf = slurp(path, &fsz);
    if (!buf) return 1;
    if (fsz > 0 && buf[0] == '@') {
        int rc = ok_file_parse_fastq(f, buf, fsz);
        free(buf);
        return rc;
    }
    if (!(fsz > 0 && buf[0] == '>')) {
        fprintf(stderr, "Unknown format: %s -> ignore\n", path);
        free(buf);
        return 1;
    }
    /* pass 1: count records (lines starting with '>') */
    uint64_t n = 0;
    for (long i = 0; i < fsz;) {
        if (buf[i] == '>') n++;
        char *nl = (char *) memchr(buf + i, '\n', (size_t) (fsz - i));
        i = nl ? (nl - buf) + 1 : fsz;
    }
    f->nb_reads = n;
    f->seqs = (char *) malloc((size_t) fsz + 1);
    f->seq_off = (uint64_t *) calloc(n + 1, sizeof(uint64_t));
    /* pass 2: a record = header line, then every following line up to the
     * next line that starts with '>' ; non-empty lines are concatenated
     * (fasta_file.h:155-175).  '\r' and any other byte stay in the sequence. */
    uint64_t r = 0, w = 0;
    long i = 0;
    while (i < fsz) {
        /* header */
        char *nl = (char *) memchr(buf + i, '\n', (size_t) (fsz - i));
        i = nl ? (nl - buf) + 1 : fsz;
        f->seq_off[r] = w;
        while (i < fsz && buf[i] != '>') {
            nl = (char *) memchr(buf + i, '\n', (size_t) (fsz - i));
            long e = nl ? (nl - buf) : fsz;
            memcpy(f->seqs + w, buf + i, (size_t) (e - i));
            w += (uint64_t) (e - i);
            i = nl ? e + 1 : fsz;
        }
        r++;
    }
    f->seq_off[r] = w;
    for (uint64_t q = r + 1; q <= n; q++) f->seq_off[q] = w;
    free(buf);
    return 0;
}

/* FastaFile::get_next_read (fasta_file.h:132-183).  Returns the sequence of
 * the next selected read (len 0 = the empty-string EOF sentinel). */
static const char *ok_file_next(ok_file *f, uint64_t *len)
{
    if (f->first_read) f->first_read = 0;
    else f->pos++;
    *len = 0;
    if (f->cnt_valid < f->nb_valid) {
        while (f->pos < f->nb_reads && !bit_get(f->bv, f->pos)) f->pos++;   /* :143-152 */
        if (f->pos < f->nb_reads) {
            *len = f->seq_off[f->pos + 1] - f->seq_off[f->pos];
            if (*len) f->cnt_valid++;                                        /* :178-180 */
            return f->seqs + f->seq_off[f->pos];
        }
    }
    return f->seqs;
}

/* ======================================================================== */
/* FileManager — include/file_manager.h                                     */
/* ======================================================================== */

typedef struct {
    char     nickname[256];
    ok_file *files;
    int      nfiles;
    int      current_file;
    uint64_t nb_seen;        /* nb_seen_reads */
} ok_fm;

static uint64_t ok_fm_total_valid(const ok_fm *m)
{
    uint64_t s = 0;                                   /* file_manager.h:268-274 */
    for (int i = 0; i < m->nfiles; i++) s += m->files[i].nb_valid;
    return s;
}

static void ok_fm_rewind(ok_fm *m)
{
    m->current_file = 0;                              /* file_manager.h:223-229 */
    m->nb_seen = 0;
    for (int i = 0; i < m->nfiles; i++) ok_file_rewind(&m->files[i]);
}

/* FileManager::get_next_read_to_compare (file_manager.h:88-112).  The
 * reference indexes files[current_file] past the end in some corner cases
 * (UB); here that state yields the empty sentinel. */
static const char *ok_fm_next(ok_fm *m, uint64_t *len)
{
    static const char empty[1] = "";
    *len = 0;
    if (m->current_file >= m->nfiles) { m->nb_seen++; return empty; }
    const char *r = ok_file_next(&m->files[m->current_file], len);
    if (*len == 0) {
        m->current_file++;
        if (m->current_file >= m->nfiles) {
            m->nb_seen++;
            return ok_file_next(&m->files[m->current_file - 1], len);       /* :94-95 */
        }
        r = ok_file_next(&m->files[m->current_file], len);
    }
    while (bit_get(m->files[m->current_file].tags, m->files[m->current_file].pos)) {   /* :99 */
        r = ok_file_next(&m->files[m->current_file], len);
        if (*len == 0) {
            m->current_file++;
            if (m->current_file >= m->nfiles) break;
            r = ok_file_next(&m->files[m->current_file], len);
        }
    }
    m->nb_seen++;
    return r;
}

static void ok_fm_tag_current(ok_fm *m)
{
    ok_file *f = &m->files[m->current_file];          /* file_manager.h:254-257 */
    bit_set(f->tags, f->pos);
}

/* ======================================================================== */
/* set_parser — include/set_parser.h                                        */
/* ======================================================================== */

typedef struct {
    char  *tag;
    char **files;
    char **bvs;       /* "" when absent */
    int    n;
} ok_set;

static void remove_spaces(char *s)
{
    /* set_parser.h:32-40: strips ' ' (only) at both ends */
    size_t b = 0, e = strlen(s);
    while (s[b] == ' ') b++;
    while (e > b && s[e - 1] == ' ') e--;
    memmove(s, s + b, e - b);
    s[e - b] = 0;
}

static void set_push(ok_set *s, const char *item)
{
    char *fname = strdup(item);
    char *bv = strdup("");
    remove_spaces(fname);
    char *comma = strchr(fname, ',');                 /* set_parser.h:74-79 */
    if (comma) {
        free(bv);
        bv = strdup(comma + 1);
        remove_spaces(bv);
        *comma = 0;
        remove_spaces(fname);
    }
    s->files = (char **) realloc(s->files, sizeof(char *) * (size_t) (s->n + 1));
    s->bvs = (char **) realloc(s->bvs, sizeof(char *) * (size_t) (s->n + 1));
    s->files[s->n] = fname;
    s->bvs[s->n] = bv;
    s->n++;
}

static int set_cmp(const void *x, const void *y)
{
    return strcmp(((const ok_set *) x)->tag, ((const ok_set *) y)->tag);
}

/* read_sets (set_parser.h:46-102); result sorted by tag like std::map,
 * a duplicate tag replaces the earlier entry. Returns -1 if unreadable. */
static int ok_read_sets(const char *path, ok_set **out)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) {
        fprintf(stderr, "Cannot read file %s\n", path);
        return -1;
    }
    ok_set *sets = NULL;
    int nsets = 0, nb_sets = 0;
    char *line = NULL;
    size_t cap = 0;
    ssize_t got;
    while ((got = getline(&line, &cap, fp)) >= 0) {
        if (got > 0 && line[got - 1] == '\n') line[--got] = 0;
        if (got == 0) continue;                       /* set_parser.h:61 */
        nb_sets++;
        ok_set s;
        memset(&s, 0, sizeof s);
        char *rest = line;
        char *colon = strchr(line, ':');
        if (colon) {                                  /* tag is NOT trimmed (:64-66) */
            *colon = 0;
            s.tag = strdup(line);
            rest = colon + 1;
        } else {
            char tmp[32];
            snprintf(tmp, sizeof tmp, "SET%d", nb_sets);
            s.tag = strdup(tmp);
        }
        /* set_parser.h:72-97: split on ';' ; the last piece is always pushed */
        char *p = rest;
        for (;;) {
            char *semi = (*p) ? strchr(p, ';') : NULL;
            if (!semi) break;
            *semi = 0;
            set_push(&s, p);
            p = semi + 1;
        }
        set_push(&s, p);
        int dup = -1;
        for (int i = 0; i < nsets; i++) if (strcmp(sets[i].tag, s.tag) == 0) dup = i;
        if (dup >= 0) sets[dup] = s;
        else {
            sets = (ok_set *) realloc(sets, sizeof(ok_set) * (size_t) (nsets + 1));
            sets[nsets++] = s;
        }
    }
    free(line);
    fclose(fp);
    qsort(sets, (size_t) nsets, sizeof(ok_set), set_cmp);
    *out = sets;
    return nsets;
}

/* ======================================================================== */
/* index_reads / search_reads over a FileManager                            */
/* ======================================================================== */

/* ---- optional chunk trace (for testing the host-side chunk planner) ---- */
static uint64_t *g_trace = NULL;      /* 4 values per chunk: first, last, n_reads, kmers (set-wide read numbers) */
static uint64_t g_trace_cap = 0, g_trace_n = 0;

void ok_trace_begin(uint64_t *buf, uint64_t cap_chunks)
{
    g_trace = buf;
    g_trace_cap = cap_chunks;
    g_trace_n = 0;
}

uint64_t ok_trace_end(void)
{
    g_trace = NULL;
    return g_trace_n;
}

static uint64_t fm_current_read_number(const ok_fm *m)
{
    int cf = m->current_file < m->nfiles ? m->current_file : m->nfiles - 1;
    uint64_t base = 0;
    for (int i = 0; i < cf; i++) base += m->files[i].nb_reads;
    return base + m->files[cf].pos;
}

/* index_reads.h:41-63 */
static ok_bloom *ok_index_reads(ok_fm *m, int k, uint64_t max_kmer, uint64_t *nb_indexed_reads, uint64_t *kmers_total)
{
    uint64_t nb_indexed_kmers = 0;
    ok_bloom *f = ok_bloom_new(k);
    ok_hash h;
    if (!f) exit(1);
    ok_hash_init(&h, k);
    uint64_t len;
    uint64_t tr_first = 0, tr_last = 0, tr_n = 0;
    const char *read = ok_fm_next(m, &len);
    while (len != 0 && nb_indexed_kmers < max_kmer) {
        (*nb_indexed_reads)++;
        if (g_trace) {
            tr_last = fm_current_read_number(m);
            if (tr_n == 0) tr_first = tr_last;
            tr_n++;
        }
        nb_indexed_kmers += index_one_read(f, &h, k, read, len);
        read = ok_fm_next(m, &len);      /* look-ahead fetch BEFORE the test (Q1) */
    }
    if (g_trace && g_trace_n < g_trace_cap) {
        g_trace[4 * g_trace_n + 0] = tr_first;
        g_trace[4 * g_trace_n + 1] = tr_last;
        g_trace[4 * g_trace_n + 2] = tr_n;
        g_trace[4 * g_trace_n + 3] = nb_indexed_kmers;
    }
    if (g_trace) g_trace_n++;
    if (kmers_total) *kmers_total += nb_indexed_kmers;
    return f;
}

/* search_reads.h:34-87 */
static uint64_t ok_search_reads(ok_bloom *f, ok_fm *m, int k, int t, uint64_t *nb_searched_reads)
{
    ok_hash h;
    ok_hash_init(&h, k);
    *nb_searched_reads = 0;
    uint64_t nb_found = 0;
    ok_fm_rewind(m);
    uint64_t len;
    const char *read = ok_fm_next(m, &len);
    while (len != 0) {
        (*nb_searched_reads)++;
        if (search_one_read(f, &h, k, t, read, len)) {
            ok_fm_tag_current(m);
            nb_found++;
        }
        read = ok_fm_next(m, &len);
    }
    return nb_found;
}

/* ======================================================================== */
/* main of index_and_search (non -f mode)                                   */
/* ======================================================================== */

static int ok_fm_add_file(ok_fm *m, const char *fname, const char *bvname)
{
    /* file_manager.h:117-216 + FastaFile ctors fasta_file.h:49-116 */
    FILE *fp = fopen(fname, "rb");
    if (!fp) {
        fprintf(stderr, "Cannot open file %s -> ignore\n", fname);
        return bvname[0] ? 0 : 1;
    }
    fclose(fp);
    m->files = (ok_file *) realloc(m->files, sizeof(ok_file) * (size_t) (m->nfiles + 1));
    ok_file *f = &m->files[m->nfiles];
    memset(f, 0, sizeof *f);
    f->fname = strdup(fname);
    if (ok_file_parse(f, fname)) return 1;
    uint64_t nbytes = f->nb_reads / 8 + 1;
    if (bvname[0]) {
        uint64_t n;
        if (ok_bv_read(bvname, &f->bv, &n)) return 1;
        if (n != f->nb_reads) {                       /* fasta_file.h:108-111 */
            fprintf(stderr, "Number of reads in %s and boolean vector size are not equal -> quit\n", fname);
            return 1;
        }
    } else {
        /* init_true: all ones, padding bits cleared (boolean_vector.h:148-164) */
        f->bv = (uint8_t *) malloc(nbytes);
        memset(f->bv, 255, nbytes);
        for (uint64_t i = f->nb_reads; i < nbytes * 8; i++) f->bv[i >> 3] &= (uint8_t) ~(1u << (i & 7));
    }
    f->nb_valid = ok_bv_nb_one(f->bv, f->nb_reads);
    f->tags = (uint8_t *) calloc(nbytes + 64, 1);     /* file_manager.h:168-169 (+slack: pos may run past the end) */
    ok_file_rewind(f);
    m->nfiles++;
    if (m->current_file < 0) m->current_file = 0;
    return 0;
}

static const char *base_name(const char *p)
{
    const char *s = strrchr(p, '/');                  /* file_manager.h:247 */
    return s ? s + 1 : p;
}

static void ensure_dir(const char *p)
{
    struct stat info;                                 /* index_and_search.cpp:178-191 */
    if (stat(p, &info) != 0) mkdir(p, S_IRWXU | S_IRGRP | S_IXGRP);
    else if (!(info.st_mode & S_IFDIR)) {
        fprintf(stderr, "Error: %s already exists and is not a directory\n", p);
        exit(1);
    }
}

int ok_index_and_search(const char *index_cfg, const char *search_cfg,
                        const char *out_dir, const char *log_dir, int k, int t,
                        ok_set_result *results, int cap, int *n_results,
                        uint64_t *n_chunks, uint64_t *kmers_indexed, int quiet)
{
    uint64_t max_kmer = g_max_kmer_override ? g_max_kmer_override : ok_max_kmer(k);
    ensure_dir(log_dir);
    ensure_dir(out_dir);

    ok_set *isets = NULL, *ssets = NULL;
    int ni = ok_read_sets(index_cfg, &isets);
    if (ni < 0) return 1;
    if (ni != 1) {                                    /* index_and_search.cpp:197-200 */
        fprintf(stderr, "Only one set of files is allowed for indexing\n");
        return 1;
    }
    ok_fm index_set;
    memset(&index_set, 0, sizeof index_set);
    index_set.current_file = -1;
    snprintf(index_set.nickname, sizeof index_set.nickname, "%s", isets[0].tag);
    for (int i = 0; i < isets[0].n; i++) {
        if (!quiet) {
            if (isets[0].bvs[i][0]) printf("open %s,%s\n", isets[0].files[i], isets[0].bvs[i]);
            else printf("open %s\n", isets[0].files[i]);
        }
        if (ok_fm_add_file(&index_set, isets[0].files[i], isets[0].bvs[i])) return 1;
    }
    int ns = ok_read_sets(search_cfg, &ssets);
    if (ns < 0) return 1;
    ok_fm *search_sets = (ok_fm *) calloc((size_t) (ns > 0 ? ns : 1), sizeof(ok_fm));
    for (int s = 0; s < ns; s++) {
        search_sets[s].current_file = -1;
        snprintf(search_sets[s].nickname, sizeof search_sets[s].nickname, "%s", ssets[s].tag);
        for (int i = 0; i < ssets[s].n; i++) {
            if (!quiet) {
                if (ssets[s].bvs[i][0]) printf("open %s,%s\n", ssets[s].files[i], ssets[s].bvs[i]);
                else printf("open %s\n", ssets[s].files[i]);
            }
            if (ok_fm_add_file(&search_sets[s], ssets[s].files[i], ssets[s].bvs[i])) return 1;
        }
    }

    /* chunk loop, index_and_search.cpp:241-277 */
    uint64_t nb_reads_to_index = ok_fm_total_valid(&index_set);
    uint64_t nb_indexed_reads = 0;
    uint64_t *nb_found = (uint64_t *) calloc((size_t) (ns > 0 ? ns : 1), sizeof(uint64_t));
    uint64_t *nb_searched = (uint64_t *) calloc((size_t) (ns > 0 ? ns : 1), sizeof(uint64_t));
    uint64_t *probes = (uint64_t *) calloc((size_t) (ns > 0 ? ns : 1), sizeof(uint64_t));
    uint64_t chunks = 0, kmers = 0;
    while (index_set.nb_seen < nb_reads_to_index) {
        ok_bloom *index = ok_index_reads(&index_set, k, max_kmer, &nb_indexed_reads, &kmers);
        chunks++;
        for (int s = 0; s < ns; s++) {
            if (!quiet) {
                printf("\n------------------------------------------------------------------\n");
                printf("finding reads from {%s} present in raw {%s}\n", search_sets[s].nickname, index_set.nickname);
                printf("------------------------------------------------------------------\n");
            }
            index->probes = 0;
            nb_found[s] += ok_search_reads(index, &search_sets[s], k, t, &nb_searched[s]);
            probes[s] += index->probes;
        }
        ok_bloom_free(index);
    }
    if (n_chunks) *n_chunks = chunks;
    if (kmers_indexed) *kmers_indexed = kmers;

    for (int s = 0; s < ns; s++) {
        if (!quiet) {
            printf("\n------------------------------------------------------------------\n");
            printf("Reads from {%s} present in raw {%s}\n", search_sets[s].nickname, index_set.nickname);
            printf("------------------------------------------------------------------\n");
            printf("[indexed %lu, searched %lu, shared %lu]\n", (unsigned long) nb_indexed_reads,
                   (unsigned long) nb_searched[s], (unsigned long) nb_found[s]);
        }
        char path[4096];
        snprintf(path, sizeof path, "%s/%s_in_%s.log", log_dir, search_sets[s].nickname, index_set.nickname);
        FILE *lf = fopen(path, "w");
        if (!lf) {
            fprintf(stderr, "Cannot open log file : %s\n", path);
            return 1;
        }
        fprintf(lf, "Index  time: 0 s\nSearch time: 0 s\nTotal  time: 0 s\n");
        fprintf(lf, "[indexed %lu, searched %lu, shared %lu]\n", (unsigned long) nb_indexed_reads,
                (unsigned long) nb_searched[s], (unsigned long) nb_found[s]);
        fclose(lf);
        if (results && s < cap) {
            snprintf(results[s].search_name, sizeof results[s].search_name, "%s", search_sets[s].nickname);
            results[s].indexed = nb_indexed_reads;
            results[s].searched = nb_searched[s];
            results[s].shared = nb_found[s];
            results[s].probes = probes[s];
        }
    }
    if (n_results) *n_results = ns;

    /* save_bv, index_and_search.cpp:397-399 + file_manager.h:245-252 */
    for (int s = 0; s < ns; s++) {
        for (int i = 0; i < search_sets[s].nfiles; i++) {
            ok_file *f = &search_sets[s].files[i];
            char path[4096], comment[4096];
            snprintf(path, sizeof path, "%s/%s_in_%s.bv", out_dir, base_name(f->fname), isets[0].tag);
            snprintf(comment, sizeof comment, "%s in %s", f->fname, isets[0].tag);
            if (ok_bv_write(path, comment, f->tags, f->nb_reads)) return 1;
        }
    }
    return 0;
}

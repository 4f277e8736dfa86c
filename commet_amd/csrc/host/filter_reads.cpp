// filter_reads — per-read length / #non-ACGT / Shannon-index filter producing a
// `.bv`, drop-in for Commet's tool (reference: src/filter_reads.cpp:48-306).
// Commet.py runs it on every input file when no filter bvs are given
// (Commet.py:103-121, 557-562).  O(bases) streaming over the mapped file.
#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <ctime>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "bv_file.hpp"
#include "fasta_source.hpp"

using namespace commet_host;

static const std::string version = "2.1";

static void print_usage()
{
    std::cout << "\nfilter_reads v" << version << "\n";
    std::cout << "Usage:\n\t./filter_reads <input_file> [options]\n";
    std::cout << "Mandatory:\n";
    std::cout << "\t<input_file>\t: file containing reads, in fasta or fastq format, gzipped or not\n";
    std::cout << "Options:\n";
    std::cout << "\t -o string\t: file where the boolean vector will be written [default=input_file.bv]\n";
    std::cout << "\t -l int\t\t: minimal length a read should have to be kept. [default=0]\n";
    std::cout << "\t -n int\t\t: maximal number of Ns a read should contain to be kept. [default=any]\n";
    std::cout << "\t -e float\t: minimal Shannon index a read should have to be kept. [default=0]\n";
    std::cout << "\t -m int\t\t: maximum number of selected reads [default=all]\n";
    std::cout << "\t -c string\t: the given string will be written in the header of the output file. [default=command line]\n";
    std::cout << "\t -h\t\t: prints this help\n";
    std::cout << "\t -v\t\t: prints the version number.\n\n";
}

// per-read statistics gathered while scanning a record's lines
struct ReadStats {
    uint64_t len = 0;
    uint64_t cnt[5] = {0, 0, 0, 0, 0};   // A C G T other (case-folded, filter_reads.cpp:277-295); other == Alphabet::is_in false (249-259)
    static const uint8_t *classes()
    {
        static uint8_t lut[256];
        static const bool init = [] {
            for (int i = 0; i < 256; ++i) lut[i] = 4;
            lut['A'] = lut['a'] = 0, lut['C'] = lut['c'] = 1, lut['G'] = lut['g'] = 2, lut['T'] = lut['t'] = 3;
            return true;
        }();
        (void) init;
        return lut;
    }
    // count = false: no criterion looks at the bases (no -n, -e 0: what Commet.py passes by default) — only the length
    void add(const char *s, size_t n, bool count = true)
    {
        len += n;
        if (!count) return;
        const uint8_t *lut = classes();
        for (size_t i = 0; i < n; ++i) ++cnt[lut[(uint8_t) s[i]]];
    }
    uint64_t non_acgt() const { return cnt[4]; }
};

// filter_reads.cpp:265-306, same float / double mix:  index (float) += f * log(f) / log(2)  with f = (float) count / (float) len.
// The double term depends on (count, len) only and is remembered per worker for the read lengths that occur.
struct Shannon {
    std::vector<std::vector<double>> memo;   // memo[len][count], NaN = not computed yet
    float operator()(const ReadStats &st)
    {
        float index = 0;
        for (int i = 0; i < 5; ++i) {
            const float f = (float) st.cnt[i] / (float) st.len;
            if (f == 0) continue;
            double term;
            if (st.len <= 1024) {
                if (memo.size() <= st.len) memo.resize(st.len + 1);
                std::vector<double> &row = memo[st.len];
                if (row.empty()) row.assign(st.len + 1, std::nan(""));
                double &slot = row[st.cnt[i]];
                if (slot != slot) slot = f * log(f) / log(2);
                term = slot;
            } else {
                term = f * log(f) / log(2);
            }
            index += term;
        }
        return fabs(index);
    }
};

// what the reference's loop does with a read (filter_reads.cpp:186-200), decided from its statistics alone
enum Verdict : uint8_t { KEEP = 0, RM_LENGTH = 1, RM_N = 2, RM_SHANNON = 3, EMPTY = 4 };

int main(int argc, char **argv)
{
    const clock_t begin_time = clock();
    std::string in_name, out_name;
    int min_size = 0, max_N = INT_MAX;
    float min_shannon = 0.0;
    std::stringstream comment;
    long max_reads = -1, nb_selected = 0;
    for (int i = 1; i < argc; ++i) {
        const std::string flag = argv[i];
        auto arg = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (flag.empty() || flag[0] != '-') {
            if (in_name.empty()) in_name = flag;
            else if (out_name.empty()) out_name = flag;
            else std::cout << "The mandatory files are already set, unknown file " << flag << " -> ignore\n";
        } else if (flag == "-o") out_name = arg();
        else if (flag == "-l") min_size = atoi(arg());
        else if (flag == "-n") max_N = atoi(arg());
        else if (flag == "-m") max_reads = atoi(arg());
        else if (flag == "-e") min_shannon = (float) atof(arg());
        else if (flag == "-c") comment << arg() << "\n";
        else if (flag == "-h") { print_usage(); return 0; }
        else if (flag == "-v") { std::cout << "\nfilter_reads version " << version << "\n"; return 0; }
        else {
            std::cerr << "Unknown option " << flag << "\n";
            print_usage();
            return 1;
        }
    }
    if (in_name.empty()) {
        std::cerr << "Error: An input file name is needed -> exit\n";
        print_usage();
        return 0;
    }
    std::string output_message;
    if (out_name.empty()) {
        output_message = "No output file name given, results will be written in " + in_name + ".bv\n";
        out_name = in_name + ".bv";
    }
    ReadFileData mf;
    if (!mf.open_file(in_name)) {
        std::cerr << "Cannot open file " << in_name << " -> quit\n";
        return 1;
    }
    if (mf.format() == ReadFormat::Unknown) {
        std::cerr << "Unknown format: " << in_name << " -> quit\n";
        return 1;
    }
    comment << "----------------\n";
    comment << "Reference file\n";
    const size_t slash = in_name.rfind("/");
    if (slash > 0 && slash < in_name.size()) comment << "  " << in_name.substr(slash + 1) << "\n";
    else comment << "  " << in_name << "\n";
    comment << "Filter Options\n";
    comment << "  min read size     : " << min_size << "\n";
    if (max_N == INT_MAX) comment << "  max number of N   : infinite\n";
    else comment << "  max number of N   : " << max_N << "\n";
    comment << "  min shannon index : " << min_shannon << "\n";

    const char *d = mf.data();
    const size_t n = mf.size();
    BitVector bv;
    const bool fastq = mf.format() == ReadFormat::Fastq;
    if (fastq) bv.init_true(count_records(mf.format(), d, n));   // FASTA: the classification below counts the records
    long rm_length = 0, rm_N = 0, rm_shannon = 0;
    const bool need_bases = max_N != INT_MAX || min_shannon > 0;   // else the verdict depends on the length alone
    auto classify = [&](const ReadStats &st, Shannon &sh) -> uint8_t {
        if (st.len == 0) return EMPTY;
        if ((int) st.len < min_size) return RM_LENGTH;
        if ((long) st.non_acgt() > (long) max_N) return RM_N;
        // the index is |...| >= 0: only a positive threshold can remove a read (the usual -e 0 never computes it)
        if (min_shannon > 0 && sh(st) < min_shannon) return RM_SHANNON;
        return KEEP;
    };
    // FASTA: pieces of whole records (cut at lines starting with '>', the reference's own record rule) are classified
    // by several threads; FASTQ stays one piece ('@' may also start a quality line).  The reference's sequential loop
    // (stop at an empty sequence or at the -m cap, counters, bits) then runs over the verdicts.
    std::vector<std::vector<uint8_t>> verdicts;
    if (fastq) {
        verdicts.resize(1);
        verdicts[0].reserve(bv.size);
        Shannon sh;
        for_each_fastq_record(d, n, bv.size, [&](const char *s, size_t len) {
            ReadStats st;
            st.add(s, len, need_bases);
            verdicts[0].push_back(classify(st, sh));
        });
    } else {
        std::vector<size_t> cuts(1, 0);
        size_t target = 8u << 20;
        if (const char *e = getenv("COMMET_FILTER_PIECE_BYTES")) target = (size_t) std::max(64L, atol(e));   // tests: many pieces from small files
        for (size_t at = target; at < n; at += target) {
            const char *q = d + at;
            while (true) {   // next line start that begins with '>'
                q = (const char *) memchr(q, '\n', (size_t) (d + n - q));
                if (!q || q + 1 >= d + n) { q = nullptr; break; }
                ++q;
                if (*q == '>') break;
            }
            if (!q) break;
            const size_t c = (size_t) (q - d);
            if (c > cuts.back()) cuts.push_back(c);
            at = std::max(at, c);
        }
        cuts.push_back(n);
        verdicts.resize(cuts.size() - 1);
        auto work = [&](size_t pi) {
            Shannon sh;
            std::vector<uint8_t> &out = verdicts[pi];
            size_t i = cuts[pi];
            const size_t end = cuts[pi + 1];
            while (i < end) {
                const char *nl = (const char *) memchr(d + i, '\n', end - i);   // header
                size_t j = nl ? (size_t) (nl - d) + 1 : end;
                ReadStats st;
                while (j < end && d[j] != '>') {
                    nl = (const char *) memchr(d + j, '\n', end - j);
                    const size_t e = nl ? (size_t) (nl - d) : end;
                    st.add(d + j, e - j, need_bases);
                    j = nl ? e + 1 : end;
                }
                i = j;
                out.push_back(classify(st, sh));
            }
        };
        unsigned nt = std::thread::hardware_concurrency();
        if (const char *e = getenv("COMMET_INGEST_THREADS")) nt = (unsigned) std::max(1, atoi(e));
        nt = std::max(1u, std::min<unsigned>(std::min(nt, 16u), (unsigned) verdicts.size()));
        std::atomic<size_t> next(0);
        auto loop = [&]() {
            for (size_t pi = next.fetch_add(1); pi < verdicts.size(); pi = next.fetch_add(1)) work(pi);
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; ++t) th.emplace_back(loop);
        loop();
        for (std::thread &t : th) t.join();
    }
    if (!fastq) {              // one verdict per line starting with '>' == the reference's read count (fasta_file.h:61-68)
        uint64_t total = 0;
        for (const std::vector<uint8_t> &piece : verdicts) total += piece.size();
        bv.init_true(total);
    }
    if (max_reads == -1) max_reads = (long) bv.size;
    uint64_t pos = 0;          // current_read_pos
    bool stopped = false;      // the reference iterator stops at an empty sequence / at the -m cap
    for (const std::vector<uint8_t> &piece : verdicts) {
        for (const uint8_t v : piece) {
            if (nb_selected >= max_reads || v == EMPTY) {   // loop condition of filter_reads.cpp:186
                stopped = true;
                break;
            }
            if (v == KEEP) ++nb_selected;
            else {
                bv.bytes[pos >> 3] &= (uint8_t) ~(1u << (pos & 7));
                if (v == RM_LENGTH) ++rm_length;
                else if (v == RM_N) ++rm_N;
                else ++rm_shannon;
            }
            ++pos;             // the look-ahead get_next_read (filter_reads.cpp:200)
        }
        if (stopped) break;
    }
    if (nb_selected >= max_reads)   // untag_last_reads: everything from the look-ahead read on
        for (uint64_t r = pos; r < bv.size; ++r) bv.bytes[r >> 3] &= (uint8_t) ~(1u << (r & 7));
    bv.comment = comment.str();
    if (!write_bv(out_name, bv)) return 1;

    std::cout << "Length filter [" << min_size << "]: " << rm_length << " reads removed\n";
    if (max_N == INT_MAX) std::cout << "Number of N filter [infinite]: " << rm_N << " reads removed\n";
    else std::cout << "Number of N filter [" << max_N << "]: " << rm_N << " reads removed\n";
    std::cout << "Shannon filter [" << min_shannon << "]: " << rm_shannon << " reads removed\n";
    std::cout << "Number of selected reads = " << nb_selected << "\n";
    if (!output_message.empty()) std::cout << output_message;
    std::cout << "Total  time : " << float(clock() - begin_time) / CLOCKS_PER_SEC << " s\n";
    return 0;
}

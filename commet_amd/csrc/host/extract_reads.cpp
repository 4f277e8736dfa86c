// extract_reads — writes the records of a read file whose bit is set in a `.bv`,
// drop-in for Commet's tool (reference: src/extract_reads.cpp:47-188).  Host-only
// (I/O): one pass over the mapped / inflated file, record text rebuilt with the
// per-format rules of the reference's parsers:
//   FASTA        header + '\n', then every non-empty line + '\n'        (fasta_file.h:155-175)
//   gzip FASTA   header (+ '\n'), then the raw bytes up to the next '>'  (fasta_file.h:417-436)
//   FASTQ        4 lines + '\n' each, blank lines skipped before the
//                header, '+' and quality lines                          (fastq_file.h:151-200)
//   gzip FASTQ   4 lines as stored, '\n' appended when missing           (fastq_file.h:466-526)
// Iteration ends at the first selected record with an empty sequence, or once
// nb_one() records have been written (fasta_file.h:142, 178-182).
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>

#include "bv_file.hpp"
#include "fasta_source.hpp"

using namespace commet_host;

static const std::string version = "2.1";

static void print_usage()
{
    std::cout << "\nextract_reads v" << version << "\n";
    std::cout << "Usage:\n\t./extract_reads <input_file> <bv_file> [options]\n";
    std::cout << "Mandatory:\n";
    std::cout << "\t<input_file>\t: file containing reads, in fasta or fastq format, gzipped or not\n";
    std::cout << "\t<bv_file>\t: associated boolean vector file\n";
    std::cout << "Options:\n";
    std::cout << "\t -o string: Output results in the given file [default=stdout]\n";
    std::cout << "\t -h: Prints this message and exit\n";
    std::cout << "\t -v: prints the version number.\n\n";
    exit(0);
}

namespace {

// Cursor over the file bytes yielding one record (sequence length + record text) per call.
class RecordCursor {
public:
    RecordCursor(ReadFormat fmt, bool gz, const char *d, size_t n) : fmt_(fmt), gz_(gz), d_(d), n_(n) {}

    // Consumes the next record.  With keep, `data` receives its text and `seq_len` the
    // number of sequence characters; without, the record is skipped.
    void next(bool keep, std::string &data, size_t &seq_len)
    {
        data.clear();
        seq_len = 0;
        if (fmt_ == ReadFormat::Fastq) fastq(keep, data, seq_len);
        else if (gz_) gz_fasta(keep, data, seq_len);
        else fasta(keep, data, seq_len);
    }

private:
    bool line(size_t &b, size_t &e) { return next_line(d_, n_, i_, b, e); }

    void fasta(bool keep, std::string &data, size_t &seq_len)
    {
        size_t b, e;
        if (!line(b, e)) return;
        if (d_[b] != '>') {   // fasta_file.h:157-160, 196-199
            std::cerr << "Error in Fasta format !!\n";
            exit(1);
        }
        if (keep) data.append(d_ + b, e - b).push_back('\n');
        while (i_ < n_ && d_[i_] != '>') {
            line(b, e);
            if (keep && e > b) {
                data.append(d_ + b, e - b).push_back('\n');
                seq_len += e - b;
            }
        }
    }

    void gz_fasta(bool keep, std::string &data, size_t &seq_len)
    {
        size_t b, e;
        if (!line(b, e)) return;
        if (d_[b] != '>') {   // fasta_file.h:418-421, 455-458
            std::cerr << "Error in Fasta format !!\n";
            exit(1);
        }
        if (keep) data.append(d_ + b, e - b).push_back('\n');
        if (!keep) {          // flush_next_read walks whole lines (fasta_file.h:459-465)
            while (i_ < n_ && d_[i_] != '>') line(b, e);
            return;
        }
        const char *gt = (const char *) memchr(d_ + i_, '>', n_ - i_);
        const size_t end = gt ? (size_t) (gt - d_) : n_;
        data.append(d_ + i_, end - i_);
        for (size_t p = i_; p < end; ++p) seq_len += d_[p] != '\n';
        i_ = end;
    }

    // next line; plain FASTQ skips blank lines where the reference's getline loops do
    bool line_nb(size_t &b, size_t &e, bool skip_blank)
    {
        bool ok;
        while ((ok = line(b, e)) && e == b && skip_blank) {}
        return ok;
    }

    void fastq(bool keep, std::string &data, size_t &seq_len)
    {
        const bool skip = !gz_;   // gzgets keeps the '\n', so "empty" never triggers (fastq_file.h:468-470)
        size_t b, e, sb = 0, se = 0;
        if (!line_nb(b, e, skip)) return;                       // header
        if (keep) data.append(d_ + b, e - b).push_back('\n');
        // sequence: verbatim when kept (fastq_file.h:166-170), blank-skipping when flushed (:222-227)
        if (!line_nb(sb, se, skip && !keep)) {
            data.clear();
            return;
        }
        if (keep) data.append(d_ + sb, se - sb).push_back('\n');
        if (!line_nb(b, e, skip)) {                             // '+' line
            data.clear();
            return;
        }
        if (keep) {
            if (!gz_ && (e == b || d_[b] != '+')) std::cerr << "Error\n";   // fastq_file.h:180-182
            data.append(d_ + b, e - b).push_back('\n');
        }
        if (line_nb(b, e, skip)) {                              // quality line
            if (keep) data.append(d_ + b, e - b).push_back('\n');
        } else if (keep && !gz_ && n_ && d_[n_ - 1] == '\n') {
            data.push_back('\n');                               // stream still good(): empty line appended (:192-196)
        } else {
            data.clear();
        }
        if (!data.empty()) seq_len = se - sb;
    }

    ReadFormat fmt_;
    bool gz_;
    const char *d_;
    size_t n_;
    size_t i_ = 0;
};

}  // namespace

int main(int argc, char **argv)
{
    if (argc < 3) print_usage();

    std::string input, bv_name, output;
    for (int a = 1; a < argc; ++a) {
        const std::string flag = argv[a];
        if (flag.empty() || flag[0] != '-') {
            if (input.empty()) input = flag;
            else if (bv_name.empty()) bv_name = flag;
            else std::cerr << "The mandatory files are already set, unknown file " << flag << " -> ignore\n";
        } else if (flag == "-o") {
            if (++a < argc) output = argv[a];
        } else if (flag == "-h") {
            print_usage();
        } else if (flag == "-v") {
            std::cout << "\nextract_reads version " << version << "\n";
            return 0;
        } else {
            std::cerr << "Unknown option " << flag << "\n";
            print_usage();
        }
    }
    if (input.empty()) {
        std::cerr << "Error: An input file name is needed -> exit\n";
        print_usage();
    } else if (bv_name.empty()) {
        std::cerr << "Error: A bv file name is needed -> exit\n";
        print_usage();
    }

    ReadFileData file;
    if (!file.open_file(input)) {
        std::cerr << "Cannot open file file " << input << " -> ignore\n";   // extract_reads.cpp:112-114
        std::cerr << "Cannot open file " << input << " -> ignore\n";        // :127-130
        return 1;
    }
    if (file.format() == ReadFormat::Unknown) {
        std::cerr << "Unknown format: " << input << " -> ignore\n";        // :141-143 (the reference then dereferences NULL)
        return 1;
    }
    const uint64_t nb_reads = count_records(file.format(), file.data(), file.size());
    BitVector bv;
    if (!read_bv(bv_name, bv)) return 1;
    if (nb_reads != bv.size) {   // fasta_file.h:108-111
        std::cerr << "Number of reads in " << input << " and boolean vector size are not equal -> quit\n";
        return 1;
    }

    // sinks (extract_reads.cpp:151-188): gzip input -> gzip output at level 6, which needs -o
    gzFile gz_out = nullptr;
    std::ofstream file_out;
    if (file.gzipped()) {
        if (output.empty()) {
            std::cerr << "Error, try to compress results but no output file name is given\n";
            return 1;
        }
        gz_out = gzopen(output.c_str(), "w6");
        if (!gz_out) {
            std::cerr << "Error, cannot open file " << output << "\n";
            return 1;
        }
    } else if (!output.empty()) {
        file_out.open(output.c_str());
        if (!file_out.good()) {
            std::cerr << "Cannot write on file " << output << "\n";
            return 1;
        }
    }

    RecordCursor cur(file.format(), file.gzipped(), file.data(), file.size());
    const uint64_t nb_valid = bv.nb_one();
    uint64_t written = 0;
    std::string data;
    size_t seq_len = 0;
    for (uint64_t pos = 0; pos < nb_reads && written < nb_valid; ++pos) {
        const bool keep = bv.get(pos);
        cur.next(keep, data, seq_len);
        if (!keep) continue;
        if (seq_len == 0) break;   // empty sequence = end-of-file sentinel (fasta_file.h:178-182)
        ++written;
        if (gz_out) {
            // gzprintf("%s"): stops at a NUL, and zlib drops a formatted string that does not fit its
            // 8192-byte buffer (zlib 1.2.11 gzwrite.c gzvprintf)
            const size_t len = strlen(data.c_str());
            if (len < 8192) gzwrite(gz_out, data.data(), (unsigned) len);
        } else if (file_out.is_open()) {
            file_out << data;
        } else {
            std::cout << data;
        }
    }
    if (gz_out) gzclose(gz_out);
    if (file_out.is_open()) file_out.close();
    return 0;
}

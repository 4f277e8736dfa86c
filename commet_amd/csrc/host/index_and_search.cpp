// index_and_search — drop-in replacement of Commet's tool of the same name
// (reference: src/index_and_search.cpp:56-422), driving the MI355X path through
// the C ABI of include/commet_hip.h.  Same flags, same set-config grammar, same
// stdout banners, same OUT/<file>_in_<set>.bv bytes and LOG/<s>_in_<i>.log
// format, so Commet.py (Commet.py:197,220,233) can call it unchanged.
//
// Host side = this file: argv, set-configs, FASTA / FASTQ (plain or gzip) -> pinned staging batches
// (fasta_source.hpp), filter / output .bv files (bv_file.hpp).  Everything
// between "reads are resident" and "tag bits are back" runs in the library.
//
// Extra (non-reference) controls: env COMMET_DEVICE=<n> picks the GPU.

#include <sys/stat.h>
#include <sys/types.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <future>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/commet_hip.h"
#include "bv_file.hpp"
#include "fasta_source.hpp"
#include "set_config.hpp"

using namespace commet_host;

static const std::string version = "2.1";   // index_and_search.cpp:44 (interface version we mirror)

static void print_usage()
{
    // index_and_search.cpp:407-422
    std::cerr << "\nindex_and_search, version " << version << "\n";
    std::cerr << "Usage : ./index_and_search -i <file> -s <file> [options]\n";
    std::cerr << "Mandatory:\n";
    std::cerr << "\t -i <file>: A file containing the list of files to index - MANDATORY\n";
    std::cerr << "\t -s <file>: A file containing the list of files to search - MANDATORY\n";
    std::cerr << "\t            Each line of the file corresponds to a set of files to search\n";
    std::cerr << "Options:\n";
    std::cerr << "\t -l </.../>: ABSOLUTE path to log folder\n";
    std::cerr << "\t -o </.../>: ABSOLUTE path to output folder\n";
    std::cerr << "\t -k <value>: Size of k-mers (value of k). [default=33]\n";
    std::cerr << "\t -t <value>: Number of shared k-mers. [default=2]\n";
    std::cerr << "\t -f: Full comparison of index set and the first searched set [default=false]\n";
    std::cerr << "\t -h: Prints this message\n";
    std::cerr << "\t -v: Prints the version number\n";
}

static void ensure_dir(const std::string &p)
{
    struct stat info;   // index_and_search.cpp:178-191
    if (stat(p.c_str(), &info) != 0) mkdir(p.c_str(), S_IRWXU | S_IRGRP | S_IXGRP);
    else if (!(info.st_mode & S_IFDIR)) {
        std::cerr << "Error: " << p << " already exists and is not a directory\n";
        exit(1);
    }
}

struct LoadedFile {
    std::string name;
    uint64_t nb_reads = 0;
    BitVector filter;     // ReadFile::bv
};

struct LoadedSet {
    std::string nickname;
    std::vector<LoadedFile> files;
    commet_readset *rs = nullptr;
    std::vector<uint8_t> select;   // set-wide input-filter bits
    bool any_bv = false;
    uint64_t n_reads = 0;
};

// FileManager::addFile for every entry of one set (file_manager.h:117-216),
// FastaFile ctors (fasta_file.h:49-116), then streams the reads to HBM.
static void load_set(commet_ctx *ctx, const std::string &nickname, const std::vector<SetEntry> &entries, LoadedSet &out)
{
    out.nickname = nickname;
    std::vector<std::unique_ptr<ReadFileData>> maps;
    std::vector<bool> bv_given;
    // the files of a set are opened (mapped, or inflated when gzipped: one zlib stream each) by one thread per file;
    // messages and error handling stay in file order
    std::vector<std::future<std::unique_ptr<ReadFileData>>> opening;
    for (const SetEntry &en : entries) {
        const std::string path = en.file;
        opening.push_back(std::async(std::launch::async, [path]() {
            std::unique_ptr<ReadFileData> f(new ReadFileData);
            if (!f->open_file(path)) f.reset();
            return f;
        }));
    }
    size_t entry_no = 0;
    for (const SetEntry &en : entries) {
        if (en.bv.empty()) std::cout << "open " << en.file << "\n";
        else std::cout << "open " << en.file << "," << en.bv << "\n";
        std::unique_ptr<ReadFileData> mf = opening[entry_no++].get();
        if (!mf) {
            if (en.bv.empty()) {
                std::cerr << "Cannot open file file " << en.file << " -> ignore\n";   // file_manager.h:121-123
                std::cerr << "Cannot open file " << en.file << " -> ignore\n";        // gz path, :144-147
                exit(1);
            }
            std::cerr << "Cannot open file " << en.file << " -> ignore\n";            // file_manager.h:177-180
            continue;
        }
        if (mf->format() == ReadFormat::Unknown) {   // neither '>' nor '@', plain or gzipped (file_manager.h:154-156)
            std::cerr << "Unknown format: " << en.file << " -> ignore\n";
            exit(1);
        }
        LoadedFile lf;
        lf.name = en.file;
        if (!en.bv.empty()) {
            if (!read_bv(en.bv, lf.filter)) exit(1);
            out.any_bv = true;
        }
        out.files.push_back(lf);
        bv_given.push_back(!en.bv.empty());
        maps.push_back(std::move(mf));
    }
    // parse + upload all files of the set (several host threads, pinned staging, hipMemcpyAsync + packing kernel)
    std::vector<const char *> data;
    std::vector<uint64_t> sizes;
    for (const std::unique_ptr<ReadFileData> &mf : maps) {
        data.push_back(mf->data());
        sizes.push_back(mf->size());
    }
    out.rs = commet_readset_from_buffers(ctx, data.data(), sizes.data(), (int) maps.size());
    if (!out.rs) {
        std::cerr << "Error: " << commet_last_error() << "\n";
        exit(1);
    }
    if (commet_readset_finalize(out.rs)) {
        std::cerr << "Error: " << commet_last_error() << "\n";
        exit(1);
    }
    out.n_reads = commet_readset_num_reads(out.rs);
    for (size_t i = 0; i < out.files.size(); ++i) {
        LoadedFile &lf = out.files[i];
        lf.nb_reads = commet_readset_file_reads(out.rs, i);
        if (!bv_given[i]) lf.filter.init_true(lf.nb_reads);
        else if (lf.nb_reads != lf.filter.size) {   // fasta_file.h:108-111
            std::cerr << "Number of reads in " << lf.name << " and boolean vector size are not equal -> quit\n";
            exit(1);
        }
    }
    // set-wide select bits = the per-file filters, concatenated
    out.select.assign(out.n_reads / 8 + 1, 0);
    if (!out.any_bv) {   // every read of every file: whole bytes at once
        std::fill(out.select.begin(), out.select.begin() + out.n_reads / 8, (uint8_t) 0xFF);
        for (uint64_t i = (out.n_reads / 8) * 8; i < out.n_reads; ++i) out.select[i >> 3] |= (uint8_t) (1u << (i & 7));
        return;
    }
    uint64_t pos = 0;
    for (const LoadedFile &lf : out.files) {
        for (uint64_t i = 0; i < lf.nb_reads; ++i)
            if (lf.filter.get(i)) out.select[(pos + i) >> 3] |= (uint8_t) (1u << ((pos + i) & 7));
        pos += lf.nb_reads;
    }
}

int main(int argc, char **argv)
{
    std::string search_file_list, index_file_list;
    int kmer_size = 33;   // index_and_search.cpp:71-72
    int min_hits = 2;
    std::string log_path = ".", out_path = ".";
    bool full = false;

    if (argc == 1) {
        print_usage();
        return 0;
    }
    int arg_pos = 1;
    while (arg_pos < argc) {   // index_and_search.cpp:85-172
        const std::string flag = argv[arg_pos];
        auto need_arg = [&]() {
            ++arg_pos;
            if (arg_pos >= argc) {
                std::cerr << "Error, flag " << argv[arg_pos - 1] << " needs an argument\n";
                print_usage();
                exit(1);
            }
        };
        if (flag == "-i") {
            need_arg();
            if (!index_file_list.empty()) std::cerr << "index files already given (-i) -> ignore";
            else index_file_list = argv[arg_pos];
        } else if (flag == "-s") {
            need_arg();
            if (!search_file_list.empty()) std::cerr << "search files already given (-s) -> ignore";
            else search_file_list = argv[arg_pos];
        } else if (flag == "-l") {
            need_arg();
            log_path = argv[arg_pos];
        } else if (flag == "-o") {
            need_arg();
            out_path = argv[arg_pos];
        } else if (flag == "-k") {
            need_arg();
            kmer_size = atoi(argv[arg_pos]);
            std::cout << "k-mer size (-k) = " << kmer_size << "\n";
        } else if (flag == "-t") {
            need_arg();
            min_hits = atoi(argv[arg_pos]);
            std::cout << "min hits (-t) = " << min_hits << "\n";
        } else if (flag == "-f") {
            full = true;
        } else if (flag == "-h") {
            print_usage();
            return 0;
        } else if (flag == "-v") {
            std::cout << "\nindex_and_search version " << version << "\n";
            return 0;
        } else {
            std::cerr << "Unknown option " << flag << "\n";
            print_usage();
            return 0;
        }
        ++arg_pos;
    }
    ensure_dir(log_path);
    ensure_dir(out_path);

    const auto start_time = std::chrono::steady_clock::now();

    SetMap index_sets, search_sets;
    if (!read_sets(index_file_list, index_sets)) exit(1);
    if (index_sets.size() != 1) {   // index_and_search.cpp:197-200
        std::cerr << "Only one set of files is allowed for indexing\n";
        exit(1);
    }

    // COMMET_INGEST_VERBOSE: wall time of the tool's phases on stderr
    const bool phase_verbose = getenv("COMMET_INGEST_VERBOSE") != nullptr;
    auto phase_t = std::chrono::steady_clock::now();
    auto phase = [&](const char *what) {
        const auto now = std::chrono::steady_clock::now();
        if (phase_verbose) fprintf(stderr, "[tool] %-24s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - phase_t).count());
        phase_t = now;
    };
    const char *dev_env = getenv("COMMET_DEVICE");
    commet_ctx *ctx = commet_create(dev_env ? atoi(dev_env) : 0, kmer_size, min_hits);
    phase("context (HIP start-up)");
    if (!ctx) {
        std::cerr << commet_last_error() << "\n";
        exit(1);
    }

    LoadedSet index_set;
    load_set(ctx, index_sets.begin()->first, index_sets.begin()->second, index_set);

    if (!read_sets(search_file_list, search_sets)) exit(1);
    // -f: only the first search set is opened (index_and_search.cpp:231-233)
    std::vector<LoadedSet> searches(full && !search_sets.empty() ? 1 : search_sets.size());
    {
        size_t s = 0;
        for (SetMap::iterator it = search_sets.begin(); it != search_sets.end() && s < searches.size(); ++it, ++s)
            load_set(ctx, it->first, it->second, searches[s]);
    }
    phase("load sets");
    if (searches.empty()) {
        // the reference dereferences search_sets[0] here (index_and_search.cpp:247): undefined
        std::cerr << "Error: no set to search\n";
        exit(1);
    }

    // the chunk loop (index_and_search.cpp:241-277) on the device
    const int ns = (int) searches.size();
    std::vector<const commet_readset *> rs(ns);
    std::vector<const uint8_t *> sel(ns);
    std::vector<std::vector<uint8_t>> tags(ns);
    std::vector<uint8_t *> tag_ptr(ns);
    for (int s = 0; s < ns; ++s) {
        rs[s] = searches[s].rs;
        sel[s] = searches[s].any_bv ? searches[s].select.data() : nullptr;
        tags[s].assign(searches[s].n_reads / 8 + 1, 0);
        tag_ptr[s] = tags[s].data();
    }
    std::vector<commet_pair_stats> stats(ns);
    commet_job_info info;
    if (commet_index_and_search(ctx, index_set.rs, index_set.any_bv ? index_set.select.data() : nullptr, ns, rs.data(),
                                sel.data(), tag_ptr.data(), stats.data(), &info)) {
        std::cerr << "Error: " << commet_last_error() << "\n";
        exit(1);
    }

    phase("index_and_search");
    // per-chunk banners (index_and_search.cpp:267-269)
    for (uint64_t c = 0; c < info.n_chunks; ++c)
        for (int s = 0; s < ns; ++s) {
            std::cout << "\n------------------------------------------------------------------\n";
            std::cout << "finding reads from {" << searches[s].nickname << "} present in raw {" << index_set.nickname
                      << "}\n";
            std::cout << "------------------------------------------------------------------\n";
        }

    const float index_s = (float) (info.index_ms / 1000.0);
    for (int s = 0; s < ns; ++s) {   // index_and_search.cpp:278-301
        const float search_s = (float) (stats[s].search_ms / 1000.0);
        auto total_s = [&]() {
            return (float) std::chrono::duration<double>(std::chrono::steady_clock::now() - start_time).count();
        };
        std::cout << "\n------------------------------------------------------------------\n";
        std::cout << "Reads from {" << searches[s].nickname << "} present in raw {" << index_set.nickname << "}\n";
        std::cout << "------------------------------------------------------------------\n";
        std::cout << "Index  time: " << index_s << " s\n";
        std::cout << "Search time: " << search_s << " s\n";
        std::cout << "Total  time: " << total_s() << " s\n";
        std::cout << "[indexed " << stats[s].indexed << ", searched " << stats[s].searched << ", shared "
                  << stats[s].shared << "]\n";
        const std::string fname = log_path + "/" + searches[s].nickname + "_in_" + index_set.nickname + ".log";
        std::ofstream log_file(fname.c_str());
        if (!log_file.good()) {
            std::cerr << "Cannot open log file : " << fname << "\n";
            exit(1);
        }
        log_file << "Index  time: " << index_s << " s\n";
        log_file << "Search time: " << search_s << " s\n";
        log_file << "Total  time: " << total_s() << " s\n";
        log_file << "[indexed " << stats[s].indexed << ", searched " << stats[s].searched << ", shared "
                 << stats[s].shared << "]\n";
        log_file.close();
    }

    // FileManager::save_bv (file_manager.h:245-252): one .bv per file of `set`, bits taken from set-wide tags
    auto save_bv = [&](const LoadedSet &set, const std::vector<uint8_t> &set_tags, const std::string &suffix) {
        uint64_t pos = 0;
        for (const LoadedFile &lf : set.files) {
            BitVector bv;
            bv.init_false(lf.nb_reads);
            for (uint64_t i = 0; i < lf.nb_reads; ++i)
                if ((set_tags[(pos + i) >> 3] >> ((pos + i) & 7)) & 1) bv.set(i);
            pos += lf.nb_reads;
            const std::string base = lf.name.substr(lf.name.rfind("/") + 1);
            bv.comment = lf.name + " in " + suffix;
            if (!write_bv(out_path + "/" + base + "_in_" + suffix + ".bv", bv)) exit(1);
        }
    };

    if (full) {
        // Full comparison on the first search set (index_and_search.cpp:304-391): A = index set, B = search set.
        //   pass 1 (above)  B in A                                   -> T1
        //   pass 2          A in (B restricted to T1)                -> T2, written as <A files>_in_<B>.bv
        //   pass 3          (B restricted to T1) in (A restricted to T2) -> written as <B files>_in_<A>.bv
        LoadedSet &A = index_set, &B = searches[0];
        auto popcount = [](const std::vector<uint8_t> &bits, uint64_t n) {
            uint64_t c = 0;
            for (uint64_t i = 0; i < n; ++i) c += (bits[i >> 3] >> (i & 7)) & 1;
            return c;
        };
        const uint64_t nb_reads_A = popcount(A.select, A.n_reads), nb_reads_B = popcount(B.select, B.n_reads);
        const std::vector<uint8_t> T1 = tags[0];
        auto one_pass = [&](const LoadedSet &idx, const uint8_t *idx_sel, const LoadedSet &srch, const uint8_t *srch_sel,
                            std::vector<uint8_t> &out_tags, const std::string &banner, const std::string &log_name,
                            uint64_t denom, bool save_now, const std::string &save_suffix) {
            std::ofstream log_file(log_name.c_str());
            if (!log_file.good()) {
                std::cerr << "Cannot open log file " << log_name << " -> exit\n";
                exit(1);
            }
            std::cout << "\n------------------------------------------------------------------\n";
            std::cout << banner << "\n";
            std::cout << "------------------------------------------------------------------\n";
            const auto t0 = std::chrono::steady_clock::now();
            const commet_readset *q = srch.rs;
            out_tags.assign(srch.n_reads / 8 + 1, 0);
            uint8_t *tp = out_tags.data();
            commet_pair_stats st;
            commet_job_info inf;
            if (commet_index_and_search(ctx, idx.rs, idx_sel, 1, &q, &srch_sel, &tp, &st, &inf)) {
                std::cerr << "Error: " << commet_last_error() << "\n";
                exit(1);
            }
            if (save_now) save_bv(srch, out_tags, save_suffix);
            const float tot = (float) std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            for (std::ostream *os : {(std::ostream *) &std::cout, (std::ostream *) &log_file}) {
                *os << "Index  time: " << (float) (inf.index_ms / 1000.0) << " s\n";
                *os << "Search time: " << (float) (st.search_ms / 1000.0) << " s\n";
                *os << "Total  time: " << tot << " s\n";
                *os << "[indexed " << st.indexed << ", searched " << st.searched << ", shared " << st.shared << "]\n"
                    << 100 * (float) st.shared / (float) denom << "%\n";
            }
        };
        std::vector<uint8_t> T2, T3;
        one_pass(B, T1.data(), A, A.select.data(), T2,
                 "finding reads from {" + A.nickname + "} present in {raw {" + B.nickname + "} present in raw {" + A.nickname + "}}",
                 log_path + "/" + A.nickname + "_in_" + B.nickname + ".log", nb_reads_A, true, B.nickname);
        one_pass(A, T2.data(), B, T1.data(), T3,
                 "finding reads from {" + B.nickname + "} present in {raw {" + A.nickname + "} present in {raw {" + B.nickname +
                     "} present in raw {" + A.nickname + "}}}",
                 log_path + "/" + B.nickname + "_in_" + A.nickname + ".log", nb_reads_B, true, A.nickname);
        tags[0] = T3;
    }

    // save_bv (index_and_search.cpp:397-399)
    for (int s = 0; s < ns; ++s) save_bv(searches[s], tags[s], index_set.nickname);
    phase("logs + .bv files");

    // (ending the process without this teardown was tried: the driver then reclaims the memory while the next job's
    // HIP start-up waits for it — same wall time per job)
    for (LoadedSet &ls : searches) commet_readset_destroy(ls.rs);
    commet_readset_destroy(index_set.rs);
    commet_destroy(ctx);
    phase("teardown");
    return 0;
}

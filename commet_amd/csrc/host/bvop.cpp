// bvop — operations on `.bv` bit-vector files, drop-in for Commet's tool of the
// same name (reference: src/bvop.cpp:54-173).  Commet.py refuses to start without it
// (Commet.py:486-488) and reads the matrices through `bvop <file> -i`, parsing the
// "  X / Y reads selected" line (Commet.py:256-257, 269).  Pure host tool: byte loops.
#include <iostream>
#include <string>

#include "bv_file.hpp"

using namespace commet_host;

static const std::string version = "2.1";

static void print_usage()
{
    std::cout << "\nbvop, version " << version << "\n";
    std::cout << "Usage : ./bvop <file1.bv> [options]\n";
    std::cout << "Mandatory:\n";
    std::cout << "\t<file1.bv>\t: file containing a boolean vector\n";
    std::cout << "Options:\n";
    std::cout << "\t -n             : performs NOT on file1.bv\n";
    std::cout << "\t -a <file2.bv>  : performs file1.bv AND file2.bv\n";
    std::cout << "\t -o <file2.bv>  : performs file1.bv OR file2.bv\n";
    std::cout << "\t -d <file2.bv>  : performs file1.bv AND (NOT file2.bv)\n";
    std::cout << "\t -p <output.bv> : print result in file output.bv [Default=stdout]\n";
    std::cout << "\t -i             : print information about file1.bv\n";
    std::cout << "\t -h             : Prints this message and exit\n";
    std::cout << "\t -v             : Prints the version number and exit\n";
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        std::cerr << "A boolean vector file must be provided, see usage\n";
        print_usage();
        return 1;
    }
    std::string file1, file2, out_name;
    bool to_file = false, info = false;
    char op = 'u';
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a[0] == '-') {
            const char flag = a.size() > 1 ? a[1] : '\0';
            if (flag == 'a' || flag == 'o' || flag == 'd') {
                if (i + 1 >= argc) { print_usage(); return 0; }
                file2 = argv[++i];
                op = flag;
            } else if (flag == 'n') {
                op = 'n';
            } else if (flag == 'p') {
                if (i + 1 >= argc) { print_usage(); return 0; }
                out_name = argv[++i];
                to_file = true;
            } else if (flag == 'i') {
                info = true;
            } else if (flag == 'v') {
                std::cout << "compare_reads version " << version << "\n";   // sic (bvop.cpp:105)
                return 0;
            } else {
                print_usage();
                return 0;
            }
        } else if (file1.empty()) {
            file1 = a;
        } else {
            std::cerr << "One input file is mandatory\n";
            print_usage();
            return 0;
        }
    }
    BitVector bv1;
    if (!read_bv(file1, bv1)) return 1;
    std::string comment;
    if (op == 'a' || op == 'o' || op == 'd') {
        BitVector bv2;
        if (!read_bv(file2, bv2)) return 1;
        if (bv2.size != bv1.size) {   // boolean_vector.h:419-422
            std::cerr << "Error: the two vectors are not the same size -> exit\n";
            return 1;
        }
        for (size_t b = 0; b < bv1.bytes.size(); ++b) {   // every byte, padding bits included
            if (op == 'a') bv1.bytes[b] &= bv2.bytes[b];
            else if (op == 'o') bv1.bytes[b] |= bv2.bytes[b];
            else bv1.bytes[b] &= (uint8_t) ~bv2.bytes[b];
        }
        comment = file1 + (op == 'a' ? " AND " : op == 'o' ? " OR " : " AND (NOT ") + file2 + (op == 'd' ? ")\n" : "\n");
    } else if (op == 'n') {
        for (uint8_t &b : bv1.bytes) b = (uint8_t) ~b;
        comment = "NOT " + file1 + "\n";
    }
    if (info) {
        std::cout << bv1.comment;
        std::cout << "\nReads:\n";
        std::cout << "  " << bv1.nb_one() << " / " << bv1.size << " reads selected\n";
    }
    if (op == 'u') return 0;
    bv1.comment = comment;
    if (to_file) return write_bv(out_name, bv1) ? 0 : 1;
    std::cout << bv1.comment << "\n#" << bv1.size << "\n";   // BooleanVector::print(), boolean_vector.h:287-295
    std::cout.write((const char *) bv1.bytes.data(), (std::streamsize) bv1.bytes.size());
    return 0;
}

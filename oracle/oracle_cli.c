/*
 * oracle_cli.c — command-line front end of the CPU restatement, same flags as
 * the reference tool (src/index_and_search.cpp:85-172, non -f mode).
 * TEST INFRASTRUCTURE ONLY (see commet_oracle.h).
 */
#include "commet_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv)
{
    const char *icfg = NULL, *scfg = NULL, *out = ".", *log = ".";
    int k = 33, t = 2;                      /* index_and_search.cpp:71-72 */
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "-i") && i + 1 < argc) icfg = argv[++i];
        else if (!strcmp(argv[i], "-s") && i + 1 < argc) scfg = argv[++i];
        else if (!strcmp(argv[i], "-o") && i + 1 < argc) out = argv[++i];
        else if (!strcmp(argv[i], "-l") && i + 1 < argc) log = argv[++i];
        else if (!strcmp(argv[i], "-k") && i + 1 < argc) { k = atoi(argv[++i]); printf("k-mer size (-k) = %d\n", k); }
        else if (!strcmp(argv[i], "-t") && i + 1 < argc) { t = atoi(argv[++i]); printf("min hits (-t) = %d\n", t); }
        else { fprintf(stderr, "Unknown option %s\n", argv[i]); return 0; }
    }
    if (!icfg || !scfg) { fprintf(stderr, "usage: oracle_cli -i <cfg> -s <cfg> [-o dir -l dir -k K -t T]\n"); return 1; }
    ok_set_result res[64];
    int n = 0;
    uint64_t chunks = 0, kmers = 0;
    int rc = ok_index_and_search(icfg, scfg, out, log, k, t, res, 64, &n, &chunks, &kmers, 0);
    for (int s = 0; s < n && s < 64; s++)
        fprintf(stderr, "oracle: %s probes=%lu chunks=%lu kmers=%lu\n", res[s].search_name,
                (unsigned long) res[s].probes, (unsigned long) chunks, (unsigned long) kmers);
    return rc;
}

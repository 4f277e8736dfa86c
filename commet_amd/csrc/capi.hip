// capi.hip — implementation of include/commet_hip.h on HIP (gfx950).
// Host side: context (filter slots, scratch of the bucketed index build), HBM
// residency of read sets, pinned multi-threaded ingest, exact chunk planning
// (read_iter.hpp) and the chunk loop of index_and_search on resident sets
// (chunks taken in groups, see search_group_kernel in kernels.hpp).
#include "../../include/commet_hip.h"

#include "kernels.hpp"
#include "index_part.hpp"
#include "slice_search.hpp"
#include "tile_search.hpp"
#include "read_iter.hpp"
#include "host/fasta_source.hpp"
#include "host/ingest_pack.hpp"

#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <future>
#include <thread>
#include <string>
#include <vector>

// The implementation, part by part (one translation unit: the kernels above are templates and inline device code, and the
// dispatch-coverage test resolves every launched entry point against this library's own symbol table).
#include "capi/state.hpp"            // errors, launch bookkeeping, commet_ctx, commet_readset
#include "capi/cache.hpp"            // query-list cache, allocations under pressure, scatter workspaces
#include "capi/context.hpp"          // commet_create / _destroy
#include "capi/readset.hpp"          // resident read sets, host ingest
#include "capi/images.hpp"           // packed images, HIP IPC hand-over
#include "capi/index_dispatch.hpp"   // index construction: which path, its launches
#include "capi/search_dispatch.hpp"  // search regimes: which one, its launches
#include "capi/job.hpp"              // commet_index_reads / _search_reads / _index_and_search
#include "capi/multi.hpp"            // commet_index_many_and_search: several jobs on one search set, their filters in one pass
#include "capi/options.hpp"          // commet_set_option, measurement hooks
#include "capi/microbench.hpp"       // commet_membench / _ldsbench

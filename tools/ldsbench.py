"""LDS operation rates of one MI355X for the access shapes of the bucketed index construction
(index_part.hpp): uniformly random words of a per-workgroup table.  Usage: python tools/ldsbench.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import commet_amd  # noqa: E402

MODES = {0: "atomic_add", 1: "atomic_add_rtn", 2: "atomic_or", 3: "store", 4: "load", 5: "atomic_add_rtn_lane_private"}


def main():
    n = 1 << 32
    with commet_amd.Context(k=10) as ctx:
        for mode, name in MODES.items():
            for words in (64, 256, 1024, 16384, 32768):
                ms = ctx.ldsbench(mode, words, n)
                print(json.dumps(dict(op=name, lds_words=words, ms=round(ms, 3), Gops_per_s=round(n / ms / 1e6, 1),
                                      ops_per_clk_per_CU=round(n / ms / 1e6 / 256 / 2.4, 2))), flush=True)


if __name__ == "__main__":
    main()

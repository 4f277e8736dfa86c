"""Random-access ceilings of the MI355X memory system for this path's access
shape (one random 4-byte probe / atomic OR per lane): the practical roofline of
SURVEY §8d next to the 8 TB/s streaming peak.  Usage: python tools/membench.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import commet_amd  # noqa: E402


def main():
    n_access = 1 << 31
    rows = []
    with commet_amd.Context(k=10) as ctx:
        for atomic in (0, 1):
            for mib in (1, 4, 32, 128, 256, 512, 1024, 2048, 4096, 8192):
                ms = ctx.membench(atomic, mib << 20, n_access)
                rate = n_access / ms / 1e6          # G accesses / s
                rows.append(dict(kind="atomic_or" if atomic else "gather4", table_MiB=mib, ms=round(ms, 3),
                                 Gaccess_per_s=round(rate, 2), GBps_64B_sectors=round(rate * 64, 1)))
                print(json.dumps(rows[-1]), flush=True)
    return rows


if __name__ == "__main__":
    main()

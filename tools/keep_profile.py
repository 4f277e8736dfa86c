"""Copies a profile directory made by tools/profile_bench.sh from gpurun_out/ (scratch) into profiles/ (tracked), without the
runtime's own copy / fill dispatches (thousands of rows that no figure is taken from):
    python tools/keep_profile.py gpurun_out/r03_c5 [profiles/r03_c5]"""
import csv
import os
import shutil
import sys


def main(src, dst=None):
    dst = dst or os.path.join("profiles", os.path.basename(src.rstrip("/")))
    os.makedirs(dst, exist_ok=True)
    for f in sorted(os.listdir(src)):
        p = os.path.join(src, f)
        if not os.path.isfile(p) or f.endswith((".log", ".err")):
            continue
        if f.startswith("pmc_") and f.endswith(".csv"):
            rows = list(csv.reader(open(p)))
            ki = rows[0].index("Kernel_Name")
            keep = [rows[0]] + [r for r in rows[1:] if not r[ki].startswith("__amd_rocclr")]
            csv.writer(open(os.path.join(dst, f), "w", newline="")).writerows(keep)
        else:
            shutil.copy(p, os.path.join(dst, f))
    print(dst, sorted(os.listdir(dst)))


if __name__ == "__main__":
    main(*sys.argv[1:3])

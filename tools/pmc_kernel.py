"""Per-kernel sums of rocprofv3 --pmc counter_collection.csv files:  python tools/pmc_kernel.py FILE.csv [kernel substring]"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("commet::", "")
        if sub not in name:
            continue
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[name].add(r["Dispatch_Id"])
    for name, d in acc.items():
        n = len(calls[name])
        print(name, "calls", n)
        for c, v in sorted(d.items()):
            print(f"   {c:28s} {v / n:16.0f} per call")


if __name__ == "__main__":
    main()

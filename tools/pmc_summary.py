#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel (dm2::* kernels only) from a counter_collection.csv."""
import collections
import csv
import glob
import sys


def main(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "dm2::" not in k:
                continue
            k = k.split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(acc.items()):
        print(k)
        for c, v in sorted(cs.items()):
            print(f"   {c:32s} n={len(v):4d} mean={sum(v)/len(v):.6g}")


if __name__ == "__main__":
    main(sys.argv[1])

#!/usr/bin/env python3
"""Shorten rocprofv3 kernel_stats.csv kernel names (template noise) so the summary fits in profiles/."""
import csv
import glob
import os
import re
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)              # drop the argument list
    name = re.sub(r"<.*", "<...>", name)           # drop template arguments
    return name[:96]


def main(src, dst=None):
    if os.path.isdir(src):                         # a rocprofv3 -d directory: its kernel_stats.csv
        found = sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True))
        if not found:
            raise SystemExit(f"no *kernel_stats.csv under {src}")
        src = found[0]
    rows = list(csv.reader(open(src)))
    f = open(dst, "w", newline="") if dst else sys.stdout
    w = csv.writer(f)
    w.writerow(rows[0])
    for r in rows[1:]:
        r[0] = short(r[0])
        w.writerow(r)


if __name__ == "__main__":
    main(*sys.argv[1:3])

#!/usr/bin/env python3
"""Shorten rocprofv3 kernel_stats.csv kernel names (template noise) so the summary fits in profiles/."""
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)              # drop the argument list
    name = re.sub(r"<.*", "<...>", name)           # drop template arguments
    return name[:96]


def main(src, dst):
    rows = list(csv.reader(open(src)))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            r[0] = short(r[0])
            w.writerow(r)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

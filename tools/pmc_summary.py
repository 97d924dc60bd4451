#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv (one counter per pass)."""
import csv
import collections
import re
import sys


def main(path, pattern="dgn::"):
    acc = collections.defaultdict(lambda: [0, 0.0])
    name_key = None
    for r in csv.DictReader(open(path)):
        if name_key is None:
            name_key = "Kernel_Name" if "Kernel_Name" in r else "Kernel Name"
        n = r[name_key]
        if pattern not in n:
            continue
        m = re.search(r"dgn::(?:\(anonymous namespace\)::)?(\w+(?:<[^>]*>)?)", n)
        key = (m.group(1) if m else n[:60], r["Counter_Name"])
        acc[key][0] += 1
        acc[key][1] += float(r["Counter_Value"])
    for (k, c), (cnt, tot) in sorted(acc.items()):
        print(f"{k:50s} {c:12s} launches {cnt:5d}  avg {tot / cnt:16.1f}")


if __name__ == "__main__":
    main(*sys.argv[1:])

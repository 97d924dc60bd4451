#!/usr/bin/env python3
"""Per-kernel averages per launch of every counter of a rocprofv3 --pmc pass:  tools/pmc_raw_summary.py DIR [name-filter]"""
import csv, sys, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
flt = sys.argv[2] if len(sys.argv) > 2 else "dgn::"
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if flt not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dgn::", "")[:46]
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k, d in sorted(rows.items(), key=lambda kv: -max(kv[1].values()))[:14]:
    print(f"{k:46s} " + " ".join(f"{c.replace('SQ_', '')}={v / cnt[k][c]:.4g}" for c, v in sorted(d.items())))

#!/usr/bin/env python3
"""SQ counters of a rocprofv3 --pmc pass as fractions of SQ_WAVE_CYCLES per kernel (WAIT_ANY / WAIT_INST_ANY / ACTIVE_INST_ANY / MFMA busy).
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES ... --output-format csv -d DIR -- cmd"""
import csv, sys, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, d in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:16]:
    wc = d.get("SQ_WAVE_CYCLES", 1)
    print(f"{k:60s} n={cnt[k]:4d} " + " ".join(f"{c[3:]}={v / wc:.3f}" for c, v in sorted(d.items()) if c != "SQ_WAVE_CYCLES") + f" wave_cyc/launch={wc / max(cnt[k], 1):.3g}")

#!/usr/bin/env python3
"""Kernel timeline of the LAST bench step from a rocprofv3 --kernel-trace csv: start offset, duration, queue; shows what overlapped.
   python tools/trace_overlap.py <kernel_trace.csv> [n_last]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step = the kernels before the last agg-leg launches: find the last bd_wgrad and take the n kernels ending there
idx = max(i for i, r in enumerate(rows) if "bd_wgrad<" in r["Kernel_Name"] or "bd_wgrad_finalize" in r["Kernel_Name"])
sel = rows[max(0, idx - n + 1): idx + 1]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dgn::", "").replace("lin::", "")[:44]
    print(f"{s / 1e3:9.1f} us +{(e - s) / 1e3:7.1f}  q={r.get('Queue_Id', '?'):>3}  {name}")
print("span", (int(sel[-1]["End_Timestamp"]) - t0) / 1e3, "us; sum of durations", sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel) / 1e3)

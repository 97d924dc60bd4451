#!/usr/bin/env python3
"""Per kernel: GRBM_GUI_ACTIVE / duration = shader clock while it ran; SQ_VALU_MFMA_BUSY_CYCLES / (GUI_ACTIVE * 1024 SIMDs) = MFMA pipe use.
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d DIR -- cmd ;  pmc_clock.py DIR"""
import csv, glob, sys, collections
d = sys.argv[1]
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
acc = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        did = r["Dispatch_Id"]
        if did not in dur: continue
        k = dur[did][1].split("(")[0][:64]
        a = acc[k]
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": a[0] += 1; a[1] += dur[did][0]; a[2] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES": a[3] += float(r["Counter_Value"])
for k, (n, ns, gui, mf) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
    if n: print(f"{k:64s} n={n:4d} avg_us={ns / n / 1e3:9.1f} clock_GHz={gui / ns:6.3f} mfma_pipe_use={mf / max(gui, 1) / 1024:6.3f}")

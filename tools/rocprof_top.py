#!/usr/bin/env python3
"""Print the per-kernel table (calls, total us, average us, share) of a rocprofv3 --kernel-trace run from its results.db (ROCm 7.2 writes
sqlite by default).  Usage: rocprof_top.py <dir or .db> [n]"""
import glob
import os
import sqlite3
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
db = sqlite3.connect(path)
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print(f"{'kernel':70s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'%':>6s}")
for name, calls, total, avg, pct in rows[:n]:
    print(f"{name[:70]:70s} {calls:7d} {total:12.1f} {avg:10.2f} {pct:6.2f}")

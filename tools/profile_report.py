#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (tools/gpu_profile.sh) into the committed evidence under profiles/:
   profiles/<round>_<tag>_kernel_stats.txt   rocprofv3 --kernel-trace --stats summary
   profiles/pmc_traffic.json                 per-kernel FETCH_SIZE / WRITE_SIZE averages (KiB) per workload tag
"""
import collections
import csv
import io
import json
import os
import re
import sys
from contextlib import redirect_stdout

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import prof_summary  # noqa: E402


def pmc_avgs(path):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        n = r.get("Kernel_Name") or r.get("Kernel Name")
        if "dgn::" not in n:
            continue
        m = re.search(r"dgn::(?:lin::|dc::|gemm::)?(?:\(anonymous namespace\)::)?(\w+(?:<[\w, ]+>)?)", n)
        key = m.group(1) if m.group(1).startswith(("ts_", "dc_")) else m.group(1).split("<")[0]
        acc[key][0] += 1
        acc[key][1] += float(r["Counter_Value"])
    return {k: v[1] / v[0] for k, v in acc.items()}


def main(tag, rnd="r01"):
    src = f"gpurun_out/prof_{tag}"
    buf = io.StringIO()
    with redirect_stdout(buf):
        prof_summary.main(f"{src}/trace/{tag}_kernel_stats.csv", 30)
    os.makedirs("profiles", exist_ok=True)
    lines = [l for l in open(f"{src}/trace.log").read().splitlines() if l.startswith("{")] if os.path.exists(f"{src}/trace.log") else []
    cmd = lines[-1][:400] if lines else ""
    with open(f"profiles/{rnd}_{tag}_kernel_stats.txt", "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py (workload tag {tag}); avg_us = average launch duration\n")
        f.write(buf.getvalue())
        f.write("\n# bench line of the profiled run (durations inside include profiler overhead):\n# " + cmd + "\n")
    if not os.path.exists(f"{src}/pmc_FETCH_SIZE/{tag}_counter_collection.csv"):   # trace-only tag
        print(open(f"profiles/{rnd}_{tag}_kernel_stats.txt").read()[:1200])
        return
    fetch = pmc_avgs(f"{src}/pmc_FETCH_SIZE/{tag}_counter_collection.csv")
    write = pmc_avgs(f"{src}/pmc_WRITE_SIZE/{tag}_counter_collection.csv")
    path = "profiles/pmc_traffic.json"
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[tag] = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (kernel-trace only); values are "
                         "per-launch averages in KiB as reported; on gfx950 FETCH_SIZE counts 64 B per 128-B request for "
                         "wide coalesced reads, so bytes_read = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM section); "
                         "WRITE_SIZE * 1024 = bytes written (checked against the exact output size of C5)",
                 "kernels": {k: {"FETCH_SIZE_KiB": fetch.get(k), "WRITE_SIZE_KiB": write.get(k)} for k in sorted(set(fetch) | set(write))}}
    json.dump(data, open(path, "w"), indent=1)
    print(open(f"profiles/{rnd}_{tag}_kernel_stats.txt").read()[:1500])
    print(json.dumps(data[tag]["kernels"], indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:])

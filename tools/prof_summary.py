#!/usr/bin/env python3
"""Condense a rocprofv3 *_kernel_stats.csv into a short table (kernel names truncated)."""
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"^(?:void )?([\w:]+)", name)
    base = m.group(1) if m else name[:60]
    tmpl = ""
    if base.startswith("dgn::") or "Cijk" in base:
        t = re.search(r"<([^>]{0,60})>", name)
        tmpl = f"<{t.group(1)}>" if t else ""
    for key in ("MulFunctor", "CUDAFunctor_add", "leaky_relu_backward", "leaky_relu", "FillFunctor", "direct_copy",
                "sum_functor", "threshold", "batch_norm", "CatArray", "normal_kernel", "index"):
        if key in name and key not in base:
            tmpl += f"[{key}]"
            break
    return (base + tmpl)[:100]


def main(path, top=40):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"{'kernel':100s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s} {'%':>6s}")
    for r in rows[:top]:
        print(f"{short(r['Name']):100s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e3:10.1f} "
              f"{float(r['AverageNs']) / 1e3:9.2f} {100 * float(r['TotalDurationNs']) / tot:6.2f}")
    print(f"total kernel time {tot / 1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)} launches")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)

#!/usr/bin/env python3
"""Registers / occupancy / LDS of the kernels whose mangled name matches a pattern, from the build's .remarks files.
usage: tools/kernel_regs.py <substring> [<substring> ...]"""
import glob, os, re, sys
here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dgn_amd", "csrc")
pats = sys.argv[1:]
for f in sorted(glob.glob(os.path.join(here, "*.remarks"))):
    cur = None
    for line in open(f, errors="replace"):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = dict(name=m.group(1)); continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|TotalSGPRs): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split()[0]] = int(m.group(2))
            if m.group(1).startswith("LDS"):
                if all(p in cur["name"] for p in pats):
                    print(os.path.basename(f)[:-8], cur["name"][:150], {k: v for k, v in cur.items() if k != "name"})
                cur = None

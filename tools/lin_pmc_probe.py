#!/usr/bin/env python3
"""A few launches of the posttrans-shaped and mixing-shaped ts_linear kernels, to be run under rocprofv3 --pmc (LDS / MFMA / wait
counters; DESIGN.md section 4 quotes the result)."""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from dgn_amd import ops, _lib
L = _lib.load()
M = 275167
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(5, M, 84, device="cuda", generator=g); w = torch.randn(5, 42, 84, device="cuda", generator=g)
xm = torch.randn(1, M, 70, device="cuda", generator=g); wm = torch.randn(1, 70, 70, device="cuda", generator=g)
for _ in range(5):
    ops._lin_fwd(L, x, w, False, None, 42)
    ops._lin_fwd(L, xm, wm, False, None, 70)
torch.cuda.synchronize()

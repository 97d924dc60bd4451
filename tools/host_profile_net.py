#!/usr/bin/env python3
"""cProfile of the host side of the eager NET training step (bench.run_net's eager loop)."""
import argparse, cProfile, os, pstats, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
pr = cProfile.Profile()
orig = bench.time.perf_counter
state = {"n": 0}
def pc():
    # the eager timed loop of run_net is bracketed by its first two perf_counter calls
    state["n"] += 1
    if state["n"] == 1:
        pr.enable()
    elif state["n"] == 2:
        pr.disable()
    return orig()
bench.time.perf_counter = pc
steps = 100
r = bench.run_net(argparse.Namespace(net_capture=False), torch.device("cuda"), steps=steps, warmup=15)
print("eager ms/step", r["ms_per_step"])
st = pstats.Stats(pr)
rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:32]
for (fn, line, func), (cc, nc, tt, ct_, callers) in rows:
    print(f"{tt / steps * 1e6:8.1f} us  calls/step {nc / steps:6.1f}  {os.path.basename(fn)}:{line} {func}")

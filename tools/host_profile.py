#!/usr/bin/env python3
"""Where the HOST time of an eager layer step goes: cProfile over bench.run_layer_workload for one workload.
   python tools/host_profile.py c3 [steps]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
import argparse
args = argparse.Namespace(steps=20, warmup=5, workload=name, scaling="weak", hipgraph=False, no_cpu_baseline=True, no_extras=True,
                          aggregators=None, scalers=None, gemm_tuning="off", cpu_sample_graphs=0, gpus=1)
dev = torch.device("cuda")
wl = dict(bench.WORKLOADS[name])
bench.run_layer_workload(args, wl, 0, 1, dev, steps=20, warmup=5, tag=name)     # warm
pr = cProfile.Profile()
pr.enable()
res, _ = bench.run_layer_workload(args, wl, 0, 1, dev, steps=steps, warmup=5, tag=name)
pr.disable()
print("ms_per_step", res["ms_per_step"])
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)

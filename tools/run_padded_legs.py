#!/usr/bin/env python3
"""The captured legs over padded static buffers (bench.py: run_bucketed, run_net) on their own."""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
args = argparse.Namespace(net_capture=True)
dev = torch.device("cuda")
for name, fn in (("c2_b128_bucketed", lambda: bench.run_bucketed(args, dev)), ("zinc_json_b128_bucketed", lambda: bench.run_bucketed(args, dev, workload="zinc_json_b128")),
                 ("zinc_net_b128", lambda: bench.run_net(args, dev))):
    r = fn()
    print(name, "ms_per_step", round(r["ms_per_step"], 4), "captured", json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in (r.get("captured") or {}).items() if k in ("ms_per_step", "replay_only_ms", "error")}))
